cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2m; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "second_order or l3_full" 2>&1 | tail -25
