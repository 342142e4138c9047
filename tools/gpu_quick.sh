cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2k; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "radius or csr or geometry" 2>&1 | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
