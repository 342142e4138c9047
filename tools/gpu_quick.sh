cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2w; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2w/prof -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
cd $GRAFT_REPO_ROOT; DB=$(find gpurun_out/r2w/prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB --csv gpurun_out/r2w/kernel_stats.csv --top 60 > gpurun_out/r2w/kernel_stats.txt; find gpurun_out/r2w/prof -name '*.db' -delete
grep -i "gate\|alpha\|attn" gpurun_out/r2w/kernel_stats.txt | cut -c1-60,112-160
