cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2a; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-180
