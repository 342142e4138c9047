cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2u; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/gemm_exp.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2u/gemm_exp2.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
