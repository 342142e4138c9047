# rocprofv3 kernel stats of the other workloads' train steps + the un-profiled wall time beside them (kernel sum vs wall)
# usage (via gpurun): bash tools/gpu_workload_profile.sh <tag> [workloads...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-rX}; shift
for w in ${@:-md17_l2 md17_l3}; do
  O=gpurun_out/${TAG}_wlprof_$w; mkdir -p $O
  timeout 120 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records --repeats 1 > $O/bench_wall.json 2> $O/err_wall.txt
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-sub-records --repeats 1 > $O/bench.json 2> $O/err.txt
  DB=$(find $O/prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv $O/kernel_stats.csv --top 60 > $O/kernel_stats.txt
  rm -rf $O/prof
  python - $O <<'PY'
import json, sys, csv
o = sys.argv[1]
wall = json.loads(open(o + "/bench_wall.json").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(o + "/kernel_stats.csv")))
tot = sum(float(r["total_us"]) for r in rows); calls = sum(int(r["calls"]) for r in rows)
steps = 6
print(f"{wall['config']['workload'][:60]}: wall {wall['ms_per_step']:.2f} ms/step; kernels {tot / steps / 1000:.2f} ms/step in {calls / steps:.0f} launches/step (6 profiled steps, start-up copies included)")
PY
  head -12 $O/kernel_stats.txt | cut -c1-70,100-150
done
