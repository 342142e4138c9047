# dev build with in-kernel marks of the multi-wave data gradient, on the box (the product library is left alone)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_trace}
mkdir -p $OUT
cp equiformer_amd/libequiformer_hip.so /tmp/lib_product.so
EQF_EXTRA_FLAGS="-DEQF_XTRACE=1" python -m equiformer_amd.build --force > $OUT/build.log 2>&1
timeout 120 python tools/sfcx_trace2.py sep_act 0 > $OUT/trace_sep_act.txt 2>&1
timeout 120 python tools/sfcx_trace2.py sep_value 0 > $OUT/trace_sep_value.txt 2>&1
cp /tmp/lib_product.so equiformer_amd/libequiformer_hip.so
grep -v "^/opt" $OUT/trace_sep_act.txt | head -60
