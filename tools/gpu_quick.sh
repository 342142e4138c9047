cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2n; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -s -k "pbc or oc20" 2>&1 | tail -25
