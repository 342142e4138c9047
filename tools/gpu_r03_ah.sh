cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ah; mkdir -p $O; export TMPDIR=/tmp
cp equiformer_amd/libequiformer_hip.so /tmp/new.so
cp equiformer_amd/libequiformer_hip_old.so equiformer_amd/libequiformer_hip.so
timeout 300 python tools/bench_sfc.py > $O/bench_sfc_tileio.txt 2>&1
cp /tmp/new.so equiformer_amd/libequiformer_hip.so
timeout 300 python tools/bench_sfc.py > $O/bench_sfc_base.txt 2>&1
grep -h "sfcx mode [01] *bwd_data" $O/bench_sfc_tileio.txt; echo ---; grep -h "sfcx mode [01] *bwd_data" $O/bench_sfc_base.txt
