set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_f}
mkdir -p $OUT
timeout 300 python tools/bench_sfcx.py 25354 0,1 > $OUT/bench_sfcx.txt 2>&1
grep bwd_data $OUT/bench_sfcx.txt
cp equiformer_amd/libequiformer_hip.so /tmp/lib_product.so
EQF_EXTRA_FLAGS="-DEQF_XTRACE=1" python -m equiformer_amd.build --force > $OUT/build.log 2>&1
timeout 120 python tools/sfcx_trace2.py sep_act 0 > $OUT/trace_sep_act.txt 2>&1
timeout 120 python tools/sfcx_trace2.py sep_value 0 > $OUT/trace_sep_value.txt 2>&1
cp /tmp/lib_product.so equiformer_amd/libequiformer_hip.so
grep -v "^/opt" $OUT/trace_sep_act.txt | head -9 | cut -c1-1500
grep -A20 "mean cycles" $OUT/trace_sep_act.txt
