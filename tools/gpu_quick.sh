cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2s; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "linear_message" 2>&1 | tail -8
