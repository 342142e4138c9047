cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_w; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from equiformer_amd import ops
from equiformer_amd.layout import DtpTable, RowLayout
dev = torch.device("cuda:0")
irr, sh = "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e"
table, lay = DtpTable(irr, sh, irr), RowLayout("224x0e+64x1e+32x2e")
spec = ops.SfcSpec(table, lay, n2=128)
E = 25354
x = torch.randn(E, table.layout_in.dim, device=dev); M = torch.randn(E, table.m_numel, device=dev)
w = torch.randn(E, table.weight_numel, device=dev)
weight = torch.randn(spec.weight_numel, device=dev); weight2 = torch.randn(spec.weight2_numel, device=dev)
d1 = torch.randn(E, lay.dim, device=dev); d2 = torch.randn(E, 128, device=dev)
dW = torch.zeros_like(weight); dW2 = torch.zeros_like(weight2)
which = sys.argv[1]
for _ in range(10):
    if which == "pack": ops._sfc_pack(weight, weight2, spec, 0)
    if which == "fwd":
        packed = ops._sfc_pack(weight, weight2, spec, 0) if _ == 0 else packed
        ops._sfc_fwd(x, M, w, weight, None, weight2, None, spec, 0, packed)
    if which == "fp32fwd": ops._sfc_fwd(x, M, w, weight, None, weight2, None, spec, None, None)
    if which == "wgrad": ops._sfc_bwd_weight(x, M, w, d1, d2, spec, dW, dW2, 0)
    if which == "ln":
        torch.empty(4, device=dev).zero_()
torch.cuda.synchronize()
PY
for k in pack fwd fp32fwd wgrad ln; do
  rm -rf $O/tr_$k; timeout 120 rocprofv3 --kernel-trace --stats -d $O/tr_$k -o t --output-format csv -- python /tmp/one.py $k > /dev/null 2> $O/tr_$k.err
  f=$(find $O/tr_$k -name '*kernel_stats.csv' | head -1)
  echo "== $k"; cut -d, -f1,2 $f | head -6
done
rm -rf $O/tr_*/
