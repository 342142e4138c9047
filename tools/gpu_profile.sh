# usage (on the GPU box, via gpurun): bash tools/gpu_profile.sh <tag>
# bench line + rocprofv3 kernel-trace summary + PMC traffic passes of the SAME build -> gpurun_out/<tag>/
# (copy bench.json, kernel_stats.*, pmc_dominant.json to profiles/ afterwards)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-rX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
nproc > $OUT/nproc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-records --repeats 1 > $OUT/bench_prof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv $OUT/kernel_stats.csv --top 80 > $OUT/kernel_stats.txt
rm -rf $OUT/prof
# PMC passes on their own (no trace options next to --pmc)
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --repeats 1 > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --repeats 1 > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA -d $OUT/pmc_mfma --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --repeats 1 > /dev/null 2> $OUT/pmc_mfma.err
python tools/pmc_bench.py $OUT $OUT/pmc_dominant.json > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma -name '*.csv' -size +2000k -delete
head -30 $OUT/kernel_stats.txt | cut -c1-90,112-160
cat $OUT/bench.json
