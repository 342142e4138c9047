cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_n; mkdir -p $O; export TMPDIR=/tmp
cp equiformer_amd/libequiformer_hip.so /tmp/new.so
cp equiformer_amd/libequiformer_hip_old.so equiformer_amd/libequiformer_hip.so
timeout 300 python tools/bench_sfc.py > $O/bench_sfc_onepass.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_sfcx.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_onepass.txt
cp /tmp/new.so equiformer_amd/libequiformer_hip.so
timeout 300 python tools/bench_sfc.py > $O/bench_sfc_twopass.txt 2>&1
cat $O/pytest_onepass.txt; grep -h "sfcx.*bwd_data" $O/bench_sfc_onepass.txt; echo ---; grep -h "sfcx.*bwd_data" $O/bench_sfc_twopass.txt
