# usage (on the GPU box, via gpurun): bash tools/gpu_trace.sh <tag>
# In-kernel clock samples of the sfcx forward (serialising, per step) and phase marks of the data gradient (tools/sfcx_trace.py).
# Needs a development build of the library next to the product one, made here (hipcc is on the box):
#   -DEQF_XTRACE=1 build of csrc/sfcx.hip linked with the product objects -> installed for the two trace runs only.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-rX_trace}
O=gpurun_out/$TAG
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -fno-slp-vectorize -DEQF_XTRACE=1 \
      -Iinclude -Iequiformer_amd/csrc -c equiformer_amd/csrc/sfcx.hip -o /tmp/sfcx_trace.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtrace.so $(ls equiformer_amd/csrc/*.o | grep -v sfcx.o) /tmp/sfcx_trace.o || exit 1
cp equiformer_amd/libequiformer_hip.so /tmp/product.so
cp /tmp/libtrace.so equiformer_amd/libequiformer_hip.so
for shape in sep_act sep_value; do
  timeout 300 python tools/sfcx_trace.py $shape 0 fwd > $O/trace_fwd_$shape.txt 2>&1
  timeout 300 python tools/sfcx_trace.py $shape 0 bwd > $O/trace_bwd_$shape.txt 2>&1
done
cp /tmp/product.so equiformer_amd/libequiformer_hip.so
tail -8 $O/trace_fwd_sep_act.txt $O/trace_bwd_sep_act.txt
