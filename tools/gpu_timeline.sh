# usage (on the GPU box, via gpurun): bash tools/gpu_timeline.sh <tag> [bench args]
# rocprofv3 kernel trace of a short bench run -> per-kernel stats + the ordered timeline of the last step
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-rX_tl}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-records --repeats 1 "$@" > $OUT/bench_prof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv $OUT/kernel_stats.csv --top 80 > $OUT/kernel_stats.txt
[ -n "$DB" ] && python tools/step_timeline.py $DB > $OUT/step_timeline.txt
rm -rf $OUT/prof
tail -3 $OUT/step_timeline.txt
