cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_o; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sfcx.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.txt
timeout 300 python tools/bench_sfc.py > $O/bench_sfc_1wave.txt 2>&1
cp equiformer_amd/libequiformer_hip.so /tmp/new.so
cp equiformer_amd/libequiformer_hip_old.so equiformer_amd/libequiformer_hip.so
timeout 300 python tools/bench_sfc.py > $O/bench_sfc_2wave_spill.txt 2>&1
cp /tmp/new.so equiformer_amd/libequiformer_hip.so
cat $O/pytest.txt; grep -h "sfcx mode [01] *\(fwd\|bwd_data\)" $O/bench_sfc_1wave.txt; echo ---; grep -h "sfcx mode [01] *\(fwd\|bwd_data\)" $O/bench_sfc_2wave_spill.txt
