cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2q; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "sfc or separable" 2>&1 | tail -3
timeout 300 python tools/bench_sfc.py 25354 2>&1 | grep order | tee gpurun_out/r2q/sfc_bench.txt
