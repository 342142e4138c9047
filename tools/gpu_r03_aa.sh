cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_af; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sfcx.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.txt
timeout 300 python tools/bench_sfc.py > $O/bench_sfc.txt 2>&1
cp equiformer_amd/libequiformer_hip.so /tmp/new.so
cp equiformer_amd/libequiformer_hip_old.so equiformer_amd/libequiformer_hip.so
timeout 300 python tools/sfcx_trace.py sep_act 0 bwd > $O/trace_bwd_sep_act.txt 2>&1
cp /tmp/new.so equiformer_amd/libequiformer_hip.so
cat $O/pytest.txt; grep -h "sfcx mode [01] *bwd_data" $O/bench_sfc.txt; tail -5 $O/trace_bwd_sep_act.txt
