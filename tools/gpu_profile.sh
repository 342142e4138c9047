# usage (on the GPU box, via gpurun): bash tools/gpu_profile.sh <tag>
# rocprofv3 kernel-trace summary + PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_MFMA: each on its own, no trace options beside
# --pmc) + the default bench line of the SAME build -> gpurun_out/<tag>/.  The PMC record is installed as
# profiles/pmc_dominant.json BEFORE the bench line is taken, so that the line carries the counter traffic of this build
# (copy bench.json, kernel_stats.*, pmc_dominant.json, pmc_summary.txt to profiles/ afterwards).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-rX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
nproc > $OUT/nproc.txt
SHORT="--no-cpu-baseline --no-sub-records --repeats 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python bench.py --steps 7 --warmup 2 $SHORT > $OUT/bench_prof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv $OUT/kernel_stats.csv --top 80 > $OUT/kernel_stats.txt
[ -n "$DB" ] && python tools/step_timeline.py $DB > $OUT/step_timeline.txt
rm -rf $OUT/prof
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_MFMA; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//; s/sq_insts_//')
  timeout 300 rocprofv3 --pmc $c -d $OUT/pmc_$d --output-format csv -- python bench.py --steps 3 --warmup 1 $SHORT > /dev/null 2> $OUT/pmc_$d.err
done
python tools/pmc_bench.py $OUT $OUT/pmc_dominant.json > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
cp $OUT/pmc_dominant.json profiles/pmc_dominant.json
find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma -name '*.csv' -size +2000k -delete
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
head -30 $OUT/kernel_stats.txt | cut -c1-90,112-160
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'))
print(json.dumps(d.get('north_star_kernels'), indent=0)[:1500])
for s in d.get('configs', []): print(s.get('workload'), s.get('matrix_mode'), s.get('value'), s.get('ms_per_step'), s.get('error'))
"
