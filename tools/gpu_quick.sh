cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2t; export TMPDIR=/tmp
timeout 600 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2t/configs.txt
