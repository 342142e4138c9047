cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2g; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | tail -2 | tee gpurun_out/r2g/bench.json
