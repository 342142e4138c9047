cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2b; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "sfc or separable" 2>&1 | tail -3
timeout 200 python tools/bench_sfc.py 2>&1 | tee gpurun_out/r2b/bench_sfc.txt
