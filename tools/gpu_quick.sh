cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -s 2>&1 | tail -25
