#!/usr/bin/env python
"""Development aid (GPU): where does the OC20 auxiliary head on an E(3) feature leave the oracle?  Runs the oracle (fp64, CPU)
and the product (cuda:0) on the fixture of tests/test_gpu_oc20_heads.py::test_oc20_aux_head_on_e3_feature and compares the
output of every sub-module in execution order through statistics that do not depend on the channel layout inside a row
(per-row sum of squares and per-row sum; the product stores [2l+1][mul], e3nn [mul][2l+1])."""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ["EQF_ALLOW_E3_AUX"] = "1"
import make_golden as mg  # noqa: E402
from weights import fill_deterministic  # noqa: E402
from oracle import nets as onets  # noqa: E402
from test_gpu_oc20_heads import _slab  # noqa: E402
from equiformer_amd.nets.graph_attention_transformer_oc20 import GraphAttentionTransformerOC20  # noqa: E402

VARIANTS = {
    "e3_missing_0o": dict(irreps_node_embedding="32x0e+16x0o+16x1e+16x1o", irreps_sh="1x0e+1x1o",
                          irreps_feature="64x0e+16x1e+16x1o", irreps_head="8x0e+4x0o+4x1e+4x1o",
                          irreps_mlp_mid="64x0e+16x0o+32x1e+16x1o"),
    "e3_complete": dict(irreps_node_embedding="32x0e+16x0o+16x1e+16x1o", irreps_sh="1x0e+1x1o",
                        irreps_feature="64x0e+16x0o+16x1e+16x1o", irreps_head="8x0e+4x0o+4x1e+4x1o",
                        irreps_mlp_mid="64x0e+16x0o+32x1e+16x1o"),
}


GRAPH = {}


def node_level(t, dst, N):
    """[N, D] as is; [E, D] summed over the edges of every destination (independent of the edge order)"""
    t = t.detach().double().cpu()
    if t.dim() == 1:
        t = t[:, None]
    t = t.reshape(t.shape[0], -1)
    if t.shape[0] == N or dst is None or t.shape[0] != dst.numel():
        return t
    return torch.zeros(N, t.shape[1], dtype=torch.float64).index_add_(0, dst.long().cpu(), t)


def capture(model, store, order, irr):
    hs = []
    for name, m in model.named_modules():
        if not name:
            continue

        def hook(mod, inp, out, name=name):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for k, o in enumerate(outs):
                if torch.is_tensor(o) and o.is_floating_point():
                    key = "%s[%d]" % (name, k)
                    if key not in store:
                        order.append(key)
                    store[key] = o.detach()
                    if k == 0 and hasattr(mod, "irreps_out"):
                        irr[key] = str(mod.irreps_out)
        hs.append(m.register_forward_hook(hook))
    return hs


def seg_errors(a, b, irreps_str):
    """a: oracle rows (e3nn layout), b: product rows (channel-fastest layout) -> {irrep: rel err} or None"""
    from equiformer_amd.irreps import Irreps
    from equiformer_amd.layout import RowLayout
    try:
        lay = RowLayout(Irreps(irreps_str).simplify())
    except Exception:
        return None
    if lay.dim != a.shape[1] or a.shape != b.shape:
        return None
    b = b[:, lay.perm_to_e3nn()]
    out = {}
    scale = a.abs().max().clamp_min(1e-30)
    for (mul, l), par, off in zip(lay.segs, lay.par, lay.offsets):
        d = mul * (2 * l + 1)
        out["%d%s" % (l, "e" if par == 1 else "o")] = ((a[:, off:off + d] - b[:, off:off + d]).abs().max() / scale).item()
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "e3_missing_0o"
    extra = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        extra[k] = (v == "1")
    cfg = dict(mg.SMALL_OC20, number_of_basis=32, use_auxiliary_task=True, **VARIANTS[which])
    cfg.update(extra)
    dev = torch.device("cuda:0")
    ref = fill_deterministic(onets.GraphAttentionTransformerOC20(**cfg), 21).double().eval()
    mod = GraphAttentionTransformerOC20(None, None, 1, **cfg)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    mod = mod.to(dev).eval()
    pos, batch, Z, tags, ei, off = _slab(2, 24, seed=3)
    sr, so, orr, oo, irr, irr2 = {}, {}, [], [], {}, {}
    capture(ref, sr, orr, irr)
    capture(mod, so, oo, irr2)
    from equiformer_amd import graph as _graph
    init0 = _graph.EdgeGraph.__init__

    def init1(self, *a, **kw):
        init0(self, *a, **kw)
        GRAPH["g"] = self
    _graph.EdgeGraph.__init__ = init1
    with torch.no_grad():
        out_r = ref(Z, tags, pos.double(), batch, edge_index=ei, offsets=off.double())
        data = SimpleNamespace(pos=pos.to(dev), batch=batch.to(dev), atomic_numbers=Z.to(dev), tags=tags.to(dev),
                               edge_index=ei.to(dev), offsets=off.to(dev))
        out = mod(data)
    N = pos.shape[0]
    dst_r, dst_p = ei[1], GRAPH["g"].dst
    print("variant", which, extra, "N", N, "E oracle", ei.shape[1], "E product", GRAPH["g"].E)
    for k, (a, b) in enumerate(zip(out_r, out)):
        a, b = a.double(), b.double().cpu()
        print("output %d: rel err %.3e" % (k, ((a - b).abs().max() / a.abs().max()).item()))
    print("%-52s %-12s %s" % ("module (oracle execution order)", "shape", "error per irrep (or layout-free statistics), node level"))
    for key in orr:
        if key not in so:
            continue
        a, b = node_level(sr[key], dst_r, N), node_level(so[key], dst_p, N)
        if a.shape != b.shape:
            print("%-52s shapes differ %s %s" % (key, tuple(a.shape), tuple(b.shape)))
            continue
        errs = seg_errors(a, b, irr[key]) if key in irr else None
        if errs is None:
            e2 = ((a.pow(2).sum(1) - b.pow(2).sum(1)).abs().max() / a.pow(2).sum(1).abs().max().clamp_min(1e-30)).item()
            txt = "sumsq %.2e" % e2
            bad = e2 > 1e-4
        else:
            txt = "  ".join("%s %.1e" % kv for kv in errs.items())
            bad = max(errs.values()) > 1e-4
        print("%-52s %-12s %s%s" % (key, tuple(sr[key].shape), txt, "   <--" if bad else ""))


if __name__ == "__main__":
    main()
