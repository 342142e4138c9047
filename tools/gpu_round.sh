# usage: bash tools/gpu_round.sh <tag> [quick]
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-rX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
nproc > $OUT/nproc.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "sfc or separable" > $OUT/pytest_quick.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest_quick.log
tail -15 $OUT/pytest_quick.log
if [ $rc -ne 0 ]; then exit 0; fi
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-records --repeats 1 > $OUT/bench_prof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv $OUT/kernel_stats.csv --top 70 > $OUT/kernel_stats.txt
find $OUT/prof -name '*.db' -delete
head -30 $OUT/kernel_stats.txt | cut -c1-90,112-160
