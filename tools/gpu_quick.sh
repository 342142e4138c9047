cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2r; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
timeout 300 python bench.py --batch 4 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
