# in-kernel clock traces (dev build of sfcx.hip with -DEQF_XTRACE=1 installed as equiformer_amd/libequiformer_hip_old.so)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_aj; mkdir -p $O; export TMPDIR=/tmp
cp equiformer_amd/libequiformer_hip.so /tmp/new.so
cp equiformer_amd/libequiformer_hip_old.so equiformer_amd/libequiformer_hip.so
timeout 300 python tools/sfcx_trace.py sep_act 0 bwd > $O/trace_bwd_sep_act.txt 2>&1
timeout 300 python tools/sfcx_trace.py sep_value 0 bwd > $O/trace_bwd_sep_value.txt 2>&1
cp /tmp/new.so equiformer_amd/libequiformer_hip.so
cat $O/trace_bwd_sep_act.txt; tail -6 $O/trace_bwd_sep_value.txt
