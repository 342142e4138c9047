cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ag; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/bench_sfc.py > $O/bench_sfc.txt 2>&1
grep -h "sfcx mode [01] *bwd_data" $O/bench_sfc.txt
