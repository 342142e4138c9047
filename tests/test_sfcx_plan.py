"""CPU check of the split-precision SeparableFCTP kernels' index arithmetic (csrc/sfcx.hip) without a GPU.

`eqf_sfcx_dev_plan` (host-only, include/equiformer_hip_dev.h) returns the argument tables the three launches would use.
This file replays the kernels' LANE-LEVEL algorithm on those tables in numpy -- the weight-plane packing in MFMA fragment
order, the workgroup -> (tile, item) order, every lane's loads, the fragment <-> (row, k) / accumulator <-> (row, column)
maps of v_mfma_f32_32x32x16_bf16, the register epilogues -- and compares the results with the operator's definition

    mid[e,(p,u),m3] = w[e,p,u] * sum_i M_p[e][i,m3] * x[e,l1(p),i,u]
    out[e,l3,m3,n]  = sum_{(p,u) -> l3} mid[e,(p,u),m3] * W_l3[(p,u),n]

and its gradients.  What it pins: the C++ planners and the layout conventions the HIP code transcribes; what it cannot
pin: the HIP source itself (tests/test_gpu_sfcx.py does that on the device, against the exact-fp32 kernels and the oracle).
Plane splitting is left out (x = plane1 + plane2 + ... exactly, and the plane products only change rounding), and so is the
route an operand takes from memory to its lane since round 3 (row-major tiles through LDS: tests/test_sfcx_tiles.py pins
those index maps); the replay reads the element each lane ends up with."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib, ops  # noqa: E402
from equiformer_amd.layout import DtpTable, RowLayout  # noqa: E402

LANES = np.arange(64)
R, HI = LANES & 31, LANES >> 5
Q = np.arange(16)
ROW_OF = (Q[None, :] & 3) + 8 * (Q[None, :] >> 2) + 4 * HI[:, None]  # [lane, q] accumulator register -> row of the tile


def _plan(kind, table, lay, n2, E, mode=0):
    L = lib.load()
    L.eqf_sfcx_dev_plan.restype = ctypes.c_int
    L.eqf_sfcx_dev_plan.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_char_p, ctypes.c_int]
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.eqf_sfcx_dev_plan(kind, ctypes.cast(table.c_ref, ctypes.c_void_p), ctypes.cast(lay.c_ref, ctypes.c_void_p), n2, E, mode,
                            buf, len(buf))
    assert n > 0, n
    return [ln.split() for ln in buf.value.decode().splitlines()]


def _order_xy(b, nx, ny, per_xcd, mode=1):
    """sfc_common.h order_xy, mode 1 (XCD-aware) / mode 3 (XCD owns the tiles 8 xi + k, item-major inside the XCD)"""
    if mode == 3:
        k, s = b & 7, b >> 3
        y = s // per_xcd
        x = 8 * (s - y * per_xcd) + k
        return (x, y) if (x < nx and y < ny) else None
    if mode == 4:  # mode 3 in batches of 8 tiles per XCD
        k, s = b & 7, b >> 3
        bt, r = divmod(s, 8 * ny)
        y = r // 8
        x = 8 * (bt * 8 + (r - y * 8)) + k
        return (x, y) if (x < nx and y < ny) else None
    Lg = (b & 7) * per_xcd + (b >> 3)
    if Lg >= nx * ny:
        return None
    return Lg // ny, Lg % ny


def _mfma(acc, a, b):
    """acc[lane, q] += A.B with A[row = lane & 31][k = 8 (lane >> 5) + j] = a[lane, j], B[k][col = lane & 31] = b[lane, j]"""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for j in range(8):
        A[R, 8 * HI + j] = a[:, j]
        B[8 * HI + j, R] = b[:, j]
    C = A @ B
    acc += C[ROW_OF, R[:, None]]


def _pack(Wcat):
    """eqf_sfcx_pack for one degree (one plane = the value): Pf[kt, ct, lane, j], Pb[s, nt, lane, j]"""
    K, N = Wcat.shape
    Pf = np.zeros((K // 16, N // 32, 64, 8))
    Pb = np.zeros((K // 32, N // 16, 64, 8))
    for kt in range(K // 16):
        for ct in range(N // 32):
            for j in range(8):
                Pf[kt, ct, :, j] = Wcat[16 * kt + 8 * HI + j, 32 * ct + R]
    for s in range(K // 32):
        for nt in range(N // 16):
            for j in range(8):
                Pb[s, nt, :, j] = Wcat[32 * s + R, 16 * nt + 8 * HI + j]
    return Pf, Pb


class Problem:
    def __init__(self, irr, sh, out_irr, n2, use_w, E, seed=0):
        self.table = DtpTable(irr, sh, irr)
        self.lay = RowLayout(out_irr)
        self.spec = ops.SfcSpec(self.table, self.lay, n2=n2)
        assert self.spec.supported
        self.n2, self.E = n2, E
        g = np.random.default_rng(seed)
        t = self.table
        self.x = g.standard_normal((E, t.layout_in.dim))
        self.M = g.standard_normal((E, t.m_numel))
        self.w = g.standard_normal((E, t.weight_numel)) if use_w else None
        self.Wcat = {}  # l3 -> [K, N1 + n2']
        for (l3, K, N1, ncat) in self.spec.degs:
            self.Wcat[l3] = g.standard_normal((K, ncat)) / np.sqrt(K)
        self.bias = g.standard_normal(self.lay.mul_of(0))
        self.bias2 = g.standard_normal(n2) if n2 else None
        self.d1 = g.standard_normal((E, self.lay.dim))
        self.d2 = g.standard_normal((E, n2)) if n2 else None
        self.out_off = {l: off for (mul, l), off in zip(self.lay.segs, self.lay.offsets)}

    # ------------------------------------------------------------------ definition
    def mid(self, l3):
        """[E, K(l3), d3]"""
        d3 = 2 * l3 + 1
        K = [k for (l, k, _, _) in self.spec.degs if l == l3][0]
        out = np.zeros((self.E, K, d3))
        for p in self.table.paths:
            if p["l3"] != l3:
                continue
            d1, mul = 2 * p["l1"] + 1, p["mul"]
            xs = self.x[:, p["in_off"]:p["in_off"] + d1 * mul].reshape(self.E, d1, mul)
            Mp = self.M[:, p["m_off"]:p["m_off"] + d1 * d3].reshape(self.E, d1, d3)
            v = np.einsum("eim,eiu->eum", Mp, xs)
            if self.w is not None:
                v = v * self.w[:, p["w_off"]:p["w_off"] + mul, None]
            out[:, p["out_ch"]:p["out_ch"] + mul, :] = v
        return out

    def forward_ref(self):
        o1 = np.zeros((self.E, self.lay.dim))
        o2 = np.zeros((self.E, self.n2)) if self.n2 else None
        for (l3, K, N1, ncat) in self.spec.degs:
            d3 = 2 * l3 + 1
            r = np.einsum("ekm,kn->emn", self.mid(l3), self.Wcat[l3])
            main = r[:, :, :N1].copy()
            if l3 == 0:
                main += self.bias
                if self.n2:
                    o2[:] = r[:, 0, N1:] + self.bias2
            o1[:, self.out_off[l3]:self.out_off[l3] + d3 * N1] = main.reshape(self.E, d3 * N1)
        return o1, o2

    def d_mid(self, l3):
        """[E, K, d3] = sum_n d_out[e, m3, n] Wcat[k, n]"""
        d3 = 2 * l3 + 1
        (K, N1) = [(k, n) for (l, k, n, _) in self.spec.degs if l == l3][0]
        do = self.d1[:, self.out_off[l3]:self.out_off[l3] + d3 * N1].reshape(self.E, d3, N1)
        if l3 == 0 and self.n2:
            do = np.concatenate([do, self.d2[:, None, :]], axis=2)
        return np.einsum("emn,kn->ekm", do, self.Wcat[l3]), do

    def backward_ref(self):
        t = self.table
        dx = np.zeros_like(self.x)
        dw = np.zeros_like(self.w) if self.w is not None else None
        dM = np.zeros_like(self.M)
        dW = {}
        for (l3, K, N1, ncat) in self.spec.degs:
            d3 = 2 * l3 + 1
            dm, do = self.d_mid(l3)
            dW[l3] = np.einsum("ekm,emn->kn", self.mid(l3), do)
            for p in t.paths:
                if p["l3"] != l3:
                    continue
                d1, mul = 2 * p["l1"] + 1, p["mul"]
                xs = self.x[:, p["in_off"]:p["in_off"] + d1 * mul].reshape(self.E, d1, mul)
                Mp = self.M[:, p["m_off"]:p["m_off"] + d1 * d3].reshape(self.E, d1, d3)
                g = dm[:, p["out_ch"]:p["out_ch"] + mul, :]  # [E, mul, d3]
                wp = self.w[:, p["w_off"]:p["w_off"] + mul] if self.w is not None else np.ones((self.E, mul))
                if dw is not None:
                    dw[:, p["w_off"]:p["w_off"] + mul] = np.einsum("eum,eim,eiu->eu", g, Mp, xs)
                gw = g * wp[:, :, None]
                dx[:, p["in_off"]:p["in_off"] + d1 * mul] += np.einsum("eum,eim->eiu", gw, Mp).reshape(self.E, d1 * mul)
                dM[:, p["m_off"]:p["m_off"] + d1 * d3] += np.einsum("eum,eiu->eim", gw, xs).reshape(self.E, d1 * d3)
        return dx, dw, dM, dW


# ---------------------------------------------------------------------------------------------------- kernel replays
def replay_fwd(P, mode=0):
    plan = _plan(0, P.table, P.lay, P.n2, P.E, mode)
    hdr = dict(zip(plan[0][1::2], map(int, plan[0][2::2])))
    items = [tuple(map(int, ln[1:])) for ln in plan if ln[0] == "item"]
    degs, cur = [], None
    for ln in plan[1:]:
        v = list(map(int, ln[1:]))
        if ln[0] == "deg":
            cur = dict(d3=v[0], N1=v[1], Ncat=v[2], out1_off=v[3], cttot=v[4], nseg=v[5], pf=v[6], segs=[])
            degs.append(cur)
        elif ln[0] == "seg":
            cur["segs"].append(dict(x_off=v[0], mul=v[1], d1=v[2], npath=v[3], m_len=v[4], m_off=v[5], paths=[]))
        elif ln[0] == "path":
            cur["segs"][-1]["paths"].append(dict(w_off=v[0], kbase=v[1], m_rel=v[2]))
    assert hdr["ny"] == len(items) and hdr["nx"] == -(-P.E // 32)
    l3s = [l3 for (l3, _, _, _) in P.spec.degs]
    packs = [_pack(P.Wcat[l3])[0] for l3 in l3s]
    # packed-buffer offsets of the plan = what the pack kernel uses (per degree: Pf then Pb, K * Ncat * NPW each)
    npw = 1 if mode == 1 else 3
    off = 0
    for D, (l3, K, N1, ncat) in zip(degs, P.spec.degs):
        assert D["pf"] == off and D["Ncat"] == ncat and D["N1"] == N1
        off += 2 * K * ncat * npw
    assert off == P.spec.packed_numel(mode)
    o1 = np.full((P.E, P.lay.dim), np.nan)
    o2 = np.full((P.E, P.n2), np.nan) if P.n2 else None
    seen = set()
    ctmax = {1: 3, 3: 2, 5: 1, 7: 1}
    for b in range(hdr["nblk"]):
        xy = _order_xy(b, hdr["nx"], hdr["ny"], hdr["per_xcd"], hdr["mode"])
        if xy is None:
            continue
        tile, y = xy
        assert (tile, y) not in seen
        seen.add((tile, y))
        di, ct0, CT = items[y]
        D = degs[di]
        D3, CTM = D["d3"], ctmax[D["d3"]]
        assert CT <= CTM
        e0 = tile * 32
        valid = e0 + R < P.E
        er = np.where(valid, e0 + R, P.E - 1)
        acc = np.zeros((D3, CTM, 64, 16))
        for S in D["segs"]:
            rows = np.minimum(e0 + np.arange(32), P.E - 1)
            Mt = P.M[rows][:, S["m_off"]:S["m_off"] + S["m_len"]]
            for c in range(0, S["mul"], 16):
                xf = np.stack([P.x[er[:, None], S["x_off"] + i * S["mul"] + c + 8 * HI[:, None] + np.arange(8)[None, :]]
                               for i in range(S["d1"])])
                for Pth in S["paths"]:
                    if P.w is not None:
                        wf = P.w[er[:, None], Pth["w_off"] + c + 8 * HI[:, None] + np.arange(8)[None, :]] * valid[:, None]
                    else:
                        wf = np.ones((64, 8)) * valid[:, None]
                    kt = (Pth["kbase"] + c) >> 4
                    for m3 in range(D3):
                        a = np.zeros((64, 8))
                        for i in range(S["d1"]):
                            a += Mt[R, Pth["m_rel"] + i * D3 + m3][:, None] * xf[i]
                        a *= wf
                        for ct in range(CTM):
                            cc = ct0 + (ct if ct < CT else CT - 1)
                            _mfma(acc[m3, ct], a, packs[di][kt, cc])
        for ct in range(CT):
            c = (ct0 + ct) * 32 + R
            main = (ct0 + ct) * 32 < D["N1"]
            for m3 in range(D3):
                for q in range(16):
                    row = ROW_OF[:, q]
                    ok = e0 + row < P.E
                    val = acc[m3, ct, :, q]
                    if D3 == 1:
                        val = val + (P.bias[c] if main else P.bias2[c - D["N1"]])
                    if main:
                        o1[(e0 + row)[ok], (D["out1_off"] + m3 * D["N1"] + c)[ok]] = val[ok]
                    else:
                        o2[(e0 + row)[ok], (c - D["N1"])[ok]] = val[ok]
    assert len(seen) == hdr["nx"] * hdr["ny"]
    return o1, o2


def replay_bwd(P, mode=0):
    plan = _plan(1, P.table, P.lay, P.n2, P.E, mode)
    hdr = dict(zip(plan[0][1::2], map(int, plan[0][2::2])))
    degs = [dict(zip(("d3", "N1", "Ncat", "out1_off", "nt", "pb"), map(int, ln[1:]))) for ln in plan if ln[0] == "deg"]
    grps = []
    for ln in plan[1:]:
        v = list(map(int, ln[1:]))
        if ln[0] == "grp":
            grps.append(dict(x_off=v[0], mul=v[1], d1=v[2], npath=v[3], paths=[]))
        elif ln[0] == "path":
            grps[-1]["paths"].append(dict(deg=v[0], mlen=v[1], krow=v[2], w_off=v[3], m_off=v[4]))
    l3s = [l3 for (l3, _, _, _) in P.spec.degs]
    packs = [_pack(P.Wcat[l3])[1] for l3 in l3s]
    npw = 1 if mode == 1 else 3
    off = 0
    for D, (l3, K, N1, ncat) in zip(degs, P.spec.degs):
        assert D["pb"] == off + K * ncat * npw and D["nt"] == ncat // 16
        off += 2 * K * ncat * npw
    dx = np.full_like(P.x, np.nan)
    dw = np.full_like(P.w, np.nan) if P.w is not None else None
    dM = np.zeros_like(P.M)
    CH = (Q[None, :] & 3) + 8 * (Q[None, :] >> 2) + 4 * HI[:, None]  # [lane, q] -> channel of the slab
    seen = set()
    for b in range(hdr["nblk"]):
        xy = _order_xy(b, hdr["nx"], hdr["ny"], hdr["per_xcd"], hdr["mode"])
        if xy is None:
            continue
        tile, gi = xy
        assert (tile, gi) not in seen
        seen.add((tile, gi))
        G = grps[gi]
        D1, mul = G["d1"], G["mul"]
        e0 = tile * 32
        valid = e0 + R < P.E
        er = np.where(valid, e0 + R, P.E - 1)
        xv = np.stack([P.x[er[:, None], G["x_off"] + i * mul + CH] for i in range(D1)])  # [i, lane, q]
        gx = np.zeros_like(xv)
        for Pth in G["paths"]:
            D = degs[Pth["deg"]]
            D3, N1 = D["d3"], D["N1"]
            assert Pth["mlen"] == D1 * D3
            Mr = P.M[er][:, Pth["m_off"]:Pth["m_off"] + D1 * D3]  # [lane, i * D3 + m3] (via the staged block, row r)
            wv = P.w[er[:, None], Pth["w_off"] + CH] if P.w is not None else np.ones((64, 16))
            acc = np.zeros((D3, 64, 16))
            for nt in range(D["nt"]):
                aw = packs[Pth["deg"]][Pth["krow"] >> 5, nt]
                n0 = 16 * nt
                for m3 in range(D3):
                    cols = 8 * HI[:, None] + np.arange(8)[None, :]
                    if n0 < N1:
                        bfrag = P.d1[er[:, None], D["out1_off"] + n0 + m3 * N1 + cols]
                    else:
                        bfrag = P.d2[er[:, None], n0 - N1 + cols]
                    _mfma(acc[m3], aw, bfrag)
            # accumulator: row = channel of the slab (ROW_OF = CH), column = edge (lane & 31)
            gw = np.zeros((64, 16))
            dMa = np.zeros((64, D1 * D3))
            for m3 in range(D3):
                dm = acc[m3]
                dmw = dm * wv
                tm = np.zeros((64, 16))
                for i in range(D1):
                    m = Mr[:, i * D3 + m3][:, None]
                    tm += m * xv[i]
                    gx[i] += m * dmw
                    dMa[:, i * D3 + m3] += (dmw * xv[i]).sum(1)
                gw += dm * tm
            if dw is not None:
                ok = np.broadcast_to(valid[:, None], (64, 16))
                dw[er[:, None].repeat(16, 1)[ok], (Pth["w_off"] + CH)[ok]] = gw[ok]
            v = dMa + dMa[LANES ^ 32]
            for ln in range(32):
                if valid[ln]:
                    dM[er[ln], Pth["m_off"]:Pth["m_off"] + D1 * D3] += v[ln]
        ok = np.broadcast_to(valid[:, None], (64, 16))
        for i in range(D1):
            dx[er[:, None].repeat(16, 1)[ok], (G["x_off"] + i * mul + CH)[ok]] = gx[i][ok]
    assert len(seen) == hdr["nx"] * hdr["ny"]
    return dx, dw, dM


def replay_wgrad(P, mode=0):
    plan = _plan(2, P.table, P.lay, P.n2, P.E, mode)
    hdr = dict(zip(plan[0][1::2], map(int, plan[0][2::2])))
    degs = [dict(zip(("d3", "N1", "N2", "out1_off"), map(int, ln[1:]))) for ln in plan if ln[0] == "deg"]
    items = [dict(zip(("slab", "ct0", "ct", "x_off", "w_off", "m_off", "x_mul", "d1", "deg", "sid"), map(int, ln[1:])))
             for ln in plan if ln[0] == "item"]
    l3s = [l3 for (l3, _, _, _) in P.spec.degs]
    dW = {l3: np.zeros_like(P.Wcat[l3]) for l3 in l3s}
    ctmax = {1: 3, 3: 2, 5: 1, 7: 1}
    ech = hdr["echunk"]
    assert ech % 32 == 0
    seen = set()
    for b in range(hdr["nblk"]):
        xy = _order_xy(b, hdr["nx"], hdr["ny"], hdr["per_xcd"], hdr["mode"])
        if xy is None:
            continue
        chunk, it = xy
        assert (chunk, it) not in seen
        seen.add((chunk, it))
        I = items[it]
        D = degs[I["deg"]]
        D1, D3, CTM, CT = I["d1"], D["d3"], ctmax[D["d3"]], I["ct"]
        LEN = D1 * D3
        ebeg, eend = chunk * ech, min(P.E, chunk * ech + ech)
        if ebeg >= eend:
            continue
        acc = np.zeros((CTM, 64, 16))
        for eb in range(ebeg, eend, 32):
            rows = np.minimum(eb + np.arange(32), eend - 1)
            Mw = P.M[rows][:, I["m_off"]:I["m_off"] + LEN]
            for half in range(2):
                if eb + 16 * half >= eend:
                    break
                ec = eb + 16 * half + 8 * HI
                a_all = np.zeros((D3, 64, 8))
                eo = np.zeros((64, 8), dtype=int)
                for j in range(8):
                    v = ec + j < eend
                    eo[:, j] = np.where(v, ec + j, eend - 1)
                    wq = (P.w[eo[:, j], I["w_off"] + R] if P.w is not None else np.ones(64)) * v
                    mrow = Mw[16 * half + 8 * HI + j]  # [lane, LEN]
                    for m3 in range(D3):
                        t = np.zeros(64)
                        for i in range(D1):
                            t += mrow[:, i * D3 + m3] * P.x[eo[:, j], I["x_off"] + i * I["x_mul"] + R]
                        a_all[m3, :, j] = t * wq
                for m3 in range(D3):
                    for ct in range(CTM):
                        c0 = (I["ct0"] + (ct if ct < CT else CT - 1)) * 32
                        if c0 < D["N1"]:
                            bf = P.d1[eo, D["out1_off"] + c0 + m3 * D["N1"] + R[:, None]]
                        else:
                            bf = P.d2[eo, c0 - D["N1"] + R[:, None]]
                        _mfma(acc[ct], a_all[m3], bf)
        l3 = l3s[I["deg"]]
        for ct in range(CT):
            c0 = (I["ct0"] + ct) * 32
            for q in range(16):
                ch = I["sid"] * 32 + ROW_OF[:, q]
                np.add.at(dW[l3], (ch, c0 + R), acc[ct, :, q])
    assert len(seen) == hdr["nx"] * hdr["ny"]
    return dW


CASES = {
    "qm9_sep_act": ("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True),
    "qm9_sep_value": ("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False),
    "oc20_l1": ("256x0e+128x1e", "1x0e+1x1e", "256x0e+128x1e", 0, True),
}


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("case", sorted(CASES))
def test_forward_replay_equals_the_definition(case):
    P = Problem(*CASES[case], E=70)  # three tiles, the last one with 6 edges
    o1, o2 = replay_fwd(P)
    r1, r2 = P.forward_ref()
    assert not np.isnan(o1).any() and _rel(o1, r1) < 1e-12
    if P.n2:
        assert not np.isnan(o2).any() and _rel(o2, r2) < 1e-12


@pytest.mark.parametrize("case", sorted(CASES))
def test_data_gradient_replay_equals_the_definition(case):
    P = Problem(*CASES[case], E=45)
    dx, dw, dM = replay_bwd(P)
    rx, rw, rM, _ = P.backward_ref()
    assert not np.isnan(dx).any() and _rel(dx, rx) < 1e-12
    assert _rel(dM, rM) < 1e-12
    if P.w is not None:
        assert not np.isnan(dw).any() and _rel(dw, rw) < 1e-12


@pytest.mark.parametrize("case", sorted(CASES))
def test_weight_gradient_replay_equals_the_definition(case):
    P = Problem(*CASES[case], E=300)  # several edge chunks of 128, the last one partial
    dW = replay_wgrad(P)
    _, _, _, rW = P.backward_ref()
    for l3 in rW:
        assert _rel(dW[l3], rW[l3]) < 1e-12, l3


@pytest.mark.parametrize("case,E", [("qm9_sep_act", 25354), ("qm9_sep_value", 9000), ("oc20_l1", 100000), ("qm9_sep_act", 70)])
def test_multi_wave_weight_gradient_plan_covers_every_slab_tile_and_edge_once(case, E):
    """csrc/sfcw.hip (round 6): workgroup types = (output degree, group of <= 4 slabs, column group), heaviest first; workgroup
    (chunk, type) works on the edges [chunk * echunk, +echunk).  Every (slab of the degree, 32-column tile) must be owned by exactly
    one type, no wave may own more d_out tiles than the kernel has slots for ((tiles + 2) // 3), chunks are whole 32-edge blocks
    and the grid enumerates every (chunk, type) once."""
    irr, sh, out_irr, n2, _ = CASES[case]
    table, lay = DtpTable(irr, sh, irr), RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay, n2=n2)
    plan = _plan(3, table, lay, n2, E, 0)
    hdr = dict(zip(plan[0][1::2], map(int, plan[0][2::2])))
    types = [ln[1:] for ln in plan if ln[0] == "type"]
    assert hdr["ny"] == len(types) and hdr["echunk"] % 32 == 0 and hdr["nx"] * hdr["echunk"] >= E > (hdr["nx"] - 1) * hdr["echunk"]
    assert hdr["lds"] <= 80 * 1024
    owned = {}
    for t in types:
        deg, d3, slab0, nsl, ct0, ct = map(int, t[:6])
        slabs = [tuple(map(int, q.split(":"))) for q in t[6:]]
        assert len(slabs) == nsl and 1 <= nsl <= 4 and 1 <= ct <= {1: 3, 3: 2, 5: 1}[d3]
        assert -(-d3 * ct // nsl) <= (d3 * ct + 2) // 3
        for sid, d1 in slabs:
            for c in range(ct0, ct0 + ct):
                assert (deg, sid, c) not in owned
                owned[(deg, sid, c)] = 1
    want = 0
    for di, (l3, K, N1, ncat) in enumerate(spec.degs):
        want += (K // 32) * (ncat // 32)
        for sid in range(K // 32):
            for c in range(ncat // 32):
                assert (di, sid, c) in owned, (di, sid, c)
    assert len(owned) == want
    seen = set()
    for b in range(hdr["nblk"]):
        xy = _order_xy(b, hdr["nx"], hdr["ny"], hdr["per_xcd"], hdr["mode"])
        if xy is not None:
            assert xy not in seen
            seen.add(xy)
    assert len(seen) == hdr["nx"] * hdr["ny"]


def test_l3_plans():
    """MD17 L_max = 3 (config #4): forward, weight gradient and -- since the output degrees are processed in chunks of m3 --
    the data gradient are planned (d1, d3 up to 7); the replays equal the definition."""
    irr = "128x0e+64x1e+64x2e+32x3e"
    P = Problem(irr, "1x0e+1x1e+1x2e+1x3e", "128x0e+64x1e+64x2e+32x3e", 0, False, E=40)
    o1, _ = replay_fwd(P)
    r1, _ = P.forward_ref()
    assert _rel(o1, r1) < 1e-12
    assert P.spec.x_ok and P.spec.x_mask(0) == 7
    got = replay_bwd(P)
    want = P.backward_ref()
    for a, b in zip(got, want):
        if a is not None and b is not None:
            assert _rel(a, b) < 1e-12


def test_planner_verdicts_drive_the_fallback():
    """eqf_sfcx_supported (host only) is the planners' own verdict: all three launches for every registered configuration; a
    256-wide L = 2 trunk keeps forward + weight gradient on the split-precision kernels and sends the data gradient (14 input
    slabs, table of 12) to the exact-fp32 kernel; a 3x-wide L = 3 trunk exceeds all three tables and runs entirely on the exact-fp32 kernels (mode None)."""
    from equiformer_amd import ops
    from equiformer_amd.layout import DtpTable, RowLayout
    irr = "128x0e+64x1e+32x2e"
    spec = ops.SfcSpec(DtpTable(irr, "1x0e+1x1e+1x2e", irr), RowLayout("224x0e+64x1e+32x2e"), n2=128)
    assert [spec.x_mask(m) for m in (0, 1, 2)] == [7, 7, 7]
    wide = "256x0e+128x1e+64x2e"
    spec = ops.SfcSpec(DtpTable(wide, "1x0e+1x1e+1x2e", wide), RowLayout(wide), n2=0)
    assert spec.supported and spec.x_mask(0) == 5
    l3 = "384x0e+192x1e+192x2e+96x3e"
    spec = ops.SfcSpec(DtpTable(l3, "1x0e+1x1e+1x2e+1x3e", l3), RowLayout(l3), n2=128)
    if spec.supported:
        assert spec.x_mask(0) == 0 and ops._sfc_mode(spec) is None
