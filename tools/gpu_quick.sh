cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2h; export TMPDIR=/tmp
timeout 300 python tools/bench_sfc.py 25354 2>&1 | tee gpurun_out/r2h/sfc_phases.txt
