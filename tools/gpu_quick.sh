cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "qm9_forward_backward" 2>&1 | tail -5
