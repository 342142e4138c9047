"""ORACLE (test infrastructure, NOT product code) -- e3nn-0.4.4 semantics restated in plain torch.

DEPENDENCY RESTATEMENT: the reference's arithmetic primitives live in the un-vendored dependency e3nn==0.4.4
(env/env_equiformer.yml:358), which is not installable in this container.  This file restates e3nn's *published*
algorithms for the things the hot path uses, anchored on the reference's call sites.  The reference's MODEL code is
executed on top of these primitives by oracle/refshim (tests/test_reference_pin.py); the primitives themselves are
pinned by answers that do not come from this code base (tests/test_independent_kat.py) -- see oracle/__init__.py.

  * ``o3.Irreps`` algebra                       (call sites: nets/graph_attention_transformer.py:765-779)
  * ``o3.wigner_3j`` (real basis)               (used inside o3.TensorProduct; nets/tensor_product_rescale.py:33-37)
  * ``o3.spherical_harmonics(normalize=True, normalization='component')``
                                                (nets/graph_attention_transformer.py:869-870)
  * ``o3.TensorProduct(path_normalization='none')`` in 'uvw' / 'uvu' / 'uuu' connection modes
                                                (nets/tensor_product_rescale.py:33-37, nets/fast_activation.py:122)
  * ``e3nn.math.normalize2mom``                 (nets/fast_activation.py:25)

Known-answer tests that pin these conventions live in tests/test_oracle_kat.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import math
import re
from fractions import Fraction
from functools import lru_cache

import torch


# ----------------------------------------------------------------------------------------------
# Irreps algebra  (e3nn.o3.Irrep / Irreps)
# ----------------------------------------------------------------------------------------------
class Irrep(tuple):
    """(l, p) with p = +1 ('e') or -1 ('o')."""

    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                m = re.fullmatch(r"\s*(\d+)([eo])\s*", l)
                assert m, l
                return tuple.__new__(cls, (int(m.group(1)), 1 if m.group(2) == "e" else -1))
            l, p = l
        assert p in (1, -1) and l >= 0
        return tuple.__new__(cls, (int(l), int(p)))

    @property
    def l(self):
        return self[0]

    @property
    def p(self):
        return self[1]

    @property
    def dim(self):
        return 2 * self[0] + 1

    def __mul__(self, other):
        """Selection rule: |l1-l2| <= l <= l1+l2, parity multiplies."""
        other = Irrep(other)
        return [Irrep(l, self.p * other.p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]

    def __repr__(self):
        return "{}{}".format(self.l, "e" if self.p == 1 else "o")


class Irreps(tuple):
    """Ordered list of (mul, Irrep)."""

    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return irreps
        out = []
        if irreps is None:
            pass
        elif isinstance(irreps, str):
            if irreps.strip() != "":
                for tok in irreps.split("+"):
                    tok = tok.strip()
                    if "x" in tok:
                        mul, ir = tok.split("x")
                        out.append((int(mul), Irrep(ir)))
                    else:
                        out.append((1, Irrep(tok)))
        else:
            for item in irreps:
                mul, ir = item
                out.append((int(mul), Irrep(ir)))
        return tuple.__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax):
        return Irreps([(1, (l, (-1) ** l)) for l in range(lmax + 1)])

    @property
    def dim(self):
        return sum(mul * ir.dim for mul, ir in self)

    @property
    def num_irreps(self):
        return sum(mul for mul, _ in self)

    @property
    def lmax(self):
        return max(ir.l for _, ir in self)

    def slices(self):
        s, i = [], 0
        for mul, ir in self:
            s.append(slice(i, i + mul * ir.dim))
            i += mul * ir.dim
        return s

    def simplify(self):
        out = []
        for mul, ir in self:
            if mul == 0:
                continue
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            else:
                out.append((mul, ir))
        return Irreps(out)

    def __contains__(self, ir):
        ir = Irrep(ir)
        return any(ir == ir2 for _, ir2 in self)

    def __add__(self, other):
        return Irreps(tuple.__add__(self, Irreps(other)))

    def __mul__(self, n):
        return Irreps(tuple.__mul__(self, n))

    def __rmul__(self, n):
        return Irreps(tuple.__mul__(self, n))

    def __repr__(self):
        return "+".join("{}x{}".format(mul, ir) for mul, ir in self)


def sort_irreps_even_first(irreps):
    """nets/tensor_product_rescale.py:224-231: sort by (l, even-before-odd, creation index)."""
    irreps = Irreps(irreps)
    out = sorted((ir.l, -ir.p, i, mul) for i, (mul, ir) in enumerate(irreps))
    inv = tuple(i for _, _, i, _ in out)
    p = [0] * len(inv)
    for new, old in enumerate(inv):
        p[old] = new
    return Irreps([(mul, (l, -mp)) for l, mp, _, mul in out]), tuple(p), inv


# ----------------------------------------------------------------------------------------------
# Wigner 3j in e3nn's real basis (e3nn/o3/_wigner.py: _su2_clebsch_gordan, change_basis_real_to_complex,
# _so3_clebsch_gordan)
# ----------------------------------------------------------------------------------------------
def _f(n):
    return math.factorial(int(round(n)))


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3):
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))
    C = math.sqrt(
        (2.0 * j3 + 1.0)
        * Fraction(
            _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3) * _f(j3 + m3) * _f(j3 - m3),
            _f(j1 + j2 + j3 + 1) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2),
        )
    )
    S = 0
    for v in range(vmin, vmax + 1):
        S += (-1) ** int(v + j2 + m2) * Fraction(
            _f(j2 + j3 + m1 - v) * _f(j1 - m1 + v), _f(v) * _f(j3 - j1 + j2 - v) * _f(j3 + m3 - v) * _f(v + j1 - j2 - m3)
        )
    return C * float(S)


def _su2_cg(j1, j2, j3):
    mat = torch.zeros(2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1, dtype=torch.float64)
    for m1 in range(-j1, j1 + 1):
        for m2 in range(-j2, j2 + 1):
            if abs(m1 + m2) <= j3:
                mat[j1 + m1, j2 + m2, j3 + m1 + m2] = _su2_cg_coeff(j1, m1, j2, m2, j3, m1 + m2)
    return mat


def _real_to_complex(l):
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    s = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = s
        q[l + m, l - abs(m)] = -1j * s
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m * s
        q[l + m, l - abs(m)] = 1j * (-1) ** m * s
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1, l2, l3):
    """Real-basis Wigner 3j, unit Frobenius norm, fp64 tensor [2l1+1, 2l2+1, 2l3+1]."""
    assert abs(l2 - l3) <= l1 <= l2 + l3
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    C = _su2_cg(l1, l2, l3).to(torch.complex128)
    C = torch.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, torch.conj(Q3.T), C)
    assert C.imag.abs().max() < 1e-9
    C = C.real
    return (C / C.norm()).contiguous()


# ----------------------------------------------------------------------------------------------
# Spherical harmonics (e3nn/o3/_spherical_harmonics.py, x-y-z order with y the polar axis)
# ----------------------------------------------------------------------------------------------
def spherical_harmonics(lmax, vec, normalize=True, normalization="component"):
    """Y^0..Y^lmax concatenated, [..., (lmax+1)^2].  lmax <= 3 (all BASELINE configs)."""
    assert lmax <= 3 and normalization == "component"
    if normalize:
        vec = torch.nn.functional.normalize(vec, dim=-1)  # e3nn: F.normalize(x, dim=-1), eps=1e-12
    x, y, z = vec[..., 0], vec[..., 1], vec[..., 2]
    out = [torch.ones_like(x)]
    if lmax >= 1:
        out += [x, y, z]
    if lmax >= 2:
        s3 = math.sqrt(3.0)
        y2 = y * y
        x2z2 = x * x + z * z
        sh20 = s3 * x * z
        sh21 = s3 * x * y
        sh22 = y2 - 0.5 * x2z2
        sh23 = s3 * y * z
        sh24 = (s3 / 2.0) * (z * z - x * x)
        out += [sh20, sh21, sh22, sh23, sh24]
    if lmax >= 3:
        out += [
            math.sqrt(5.0 / 6.0) * (sh20 * z + sh24 * x),
            math.sqrt(5.0) * sh20 * y,
            math.sqrt(3.0 / 8.0) * (4.0 * y2 - x2z2) * x,
            0.5 * y * (2.0 * y2 - 3.0 * x2z2),
            math.sqrt(3.0 / 8.0) * z * (4.0 * y2 - x2z2),
            math.sqrt(5.0) * sh24 * y,
            math.sqrt(5.0 / 6.0) * (sh24 * z - sh20 * x),
        ]
    sh = torch.stack(out, dim=-1)
    scale = torch.cat([torch.full((2 * l + 1,), math.sqrt(2 * l + 1), dtype=sh.dtype, device=sh.device)
                       for l in range(lmax + 1)])
    return sh * scale


# ----------------------------------------------------------------------------------------------
# normalize2mom (e3nn/math/_normalize_activation.py): Monte-Carlo second-moment constant
# ----------------------------------------------------------------------------------------------
def normalize2mom_const(f):
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
    with torch.no_grad():
        return f(z).pow(2).mean().pow(-0.5).item()


# ----------------------------------------------------------------------------------------------
# o3.TensorProduct(path_normalization='none', irrep_normalization='component')
# ----------------------------------------------------------------------------------------------
class TensorProduct(torch.nn.Module):
    """Instructions are (i_in1, i_in2, i_out, mode, has_weight[, path_weight]).

    out[z, i_out] += sqrt(2 l_out + 1) * path_weight * sum_{ij} w * C_ijk * x1 * x2     (e3nn codegen)
      'uvw': einsum('zuvw,ijk,zui,zvj->zwk');  'uvu': einsum('zuv,ijk,zui,zvj->zuk');  'uuu': einsum('zu,ijk,zui,zuj->zuk')
    Flat weight = instruction weights concatenated in instruction order, each of shape
    (mul1, mul2, mul_out) / (mul1, mul2) / (mul,).
    """

    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, internal_weights=None, shared_weights=None):
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        self.irreps_out = Irreps(irreps_out)
        ins = []
        for x in instructions:
            x = tuple(x)
            if len(x) == 5:
                x = x + (1.0,)
            ins.append(x)
        self.instructions = ins
        if shared_weights is False and internal_weights is None:
            internal_weights = False
        if shared_weights is None:
            shared_weights = True
        if internal_weights is None:
            internal_weights = shared_weights and any(i[4] for i in ins)
        self.internal_weights = internal_weights
        self.shared_weights = shared_weights
        self.weight_shapes = []
        for i1, i2, io, mode, has_w, pw in ins:
            m1, m2, mo = self.irreps_in1[i1][0], self.irreps_in2[i2][0], self.irreps_out[io][0]
            shape = {"uvw": (m1, m2, mo), "uvu": (m1, m2), "uuu": (m1,)}[mode] if has_w else ()
            self.weight_shapes.append(shape)
        self.weight_numel = sum(math.prod(s) for s in self.weight_shapes if s)
        if self.internal_weights and self.weight_numel > 0:
            self.weight = torch.nn.Parameter(torch.randn(self.weight_numel))
        else:
            self.register_buffer("weight", torch.Tensor())

    def weight_views(self, weight=None):
        weight = self.weight if weight is None else weight
        off, views = 0, []
        for s in self.weight_shapes:
            n = math.prod(s) if s else 0
            if s:
                views.append(weight.narrow(-1, off, n).reshape(weight.shape[:-1] + s))
            off += n
        return views

    def forward(self, x1, x2, weight=None):
        if weight is None:
            weight = self.weight
        shared = weight.dim() == 1
        z = x1.shape[0]
        s1, s2 = self.irreps_in1.slices(), self.irreps_in2.slices()
        outs = [None] * len(self.irreps_out)
        off = 0
        for (i1, i2, io, mode, has_w, pw), shape in zip(self.instructions, self.weight_shapes):
            (m1, ir1), (m2, ir2), (mo, iro) = self.irreps_in1[i1], self.irreps_in2[i2], self.irreps_out[io]
            a = x1[:, s1[i1]].reshape(z, m1, ir1.dim)
            b = x2[:, s2[i2]].reshape(z, m2, ir2.dim)
            C = wigner_3j(ir1.l, ir2.l, iro.l).to(x1.dtype)
            coef = math.sqrt(iro.dim) * pw
            if has_w:
                n = math.prod(shape)
                w = weight.narrow(-1, off, n)
                off += n
                w = w.reshape(shape) if shared else w.reshape((z,) + shape)
            if mode == "uvw":
                assert has_w
                r = torch.einsum("uvw,ijk,zui,zvj->zwk" if shared else "zuvw,ijk,zui,zvj->zwk", w, C, a, b)
            elif mode == "uvu":
                if has_w:
                    r = torch.einsum("uv,ijk,zui,zvj->zuk" if shared else "zuv,ijk,zui,zvj->zuk", w, C, a, b)
                else:
                    r = torch.einsum("ijk,zui,zvj->zuk", C, a, b)
            elif mode == "uuu":
                if has_w:
                    r = torch.einsum("u,ijk,zui,zuj->zuk" if shared else "zu,ijk,zui,zuj->zuk", w, C, a, b)
                else:
                    r = torch.einsum("ijk,zui,zuj->zuk", C, a, b)
            else:
                raise NotImplementedError(mode)
            r = (coef * r).reshape(z, mo * iro.dim)
            outs[io] = r if outs[io] is None else outs[io] + r
        for k, (mo, iro) in enumerate(self.irreps_out):
            if outs[k] is None:
                outs[k] = x1.new_zeros(z, mo * iro.dim)
        return torch.cat(outs, dim=1)
