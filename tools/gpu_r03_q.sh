cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_s; mkdir -p $O; export TMPDIR=/tmp
cp equiformer_amd/libequiformer_hip.so /tmp/new.so
cp equiformer_amd/libequiformer_hip_old.so equiformer_amd/libequiformer_hip.so
timeout 300 python tools/sfcx_trace.py sep_act 0 > $O/trace_sep_act.txt 2>&1
timeout 300 python tools/sfcx_trace.py sep_value 0 > $O/trace_sep_value.txt 2>&1
timeout 300 python tools/sfcx_trace.py sep_act 1 > $O/trace_sep_act_bf16.txt 2>&1
cp /tmp/new.so equiformer_amd/libequiformer_hip.so
cat $O/trace_sep_act.txt; tail -8 $O/trace_sep_value.txt
