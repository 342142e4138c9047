cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['north_star_kernels']['scatter'])"
