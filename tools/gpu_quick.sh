cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z; export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "sfc or separable" 2>&1 | tail -2
timeout 60 python tools/sfc_exp.py 2>&1 | grep "fwd exp= 0" | tee gpurun_out/r2z/x6_shared.txt
