set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r1b
nproc > gpurun_out/r1b/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r1b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1b/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r1b/bench.json 2> gpurun_out/r1b/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r1b/prof -o r1b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r1b/bench_prof.json 2> gpurun_out/r1b/prof.err
ls -R gpurun_out/r1b/prof | head -30
DB=$(find gpurun_out/r1b/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv gpurun_out/r1b/kernel_stats.csv --top 70 > gpurun_out/r1b/kernel_stats.txt
find gpurun_out/r1b/prof -name '*.db' -delete
tail -3 gpurun_out/r1b/pytest.log; cat gpurun_out/r1b/bench.json
