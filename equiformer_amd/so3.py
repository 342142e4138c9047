"""SO(3) constants consumed by the HIP kernels: real-basis Wigner-3j tables (with the e3nn 'component'
path normalisation folded in) and the second-moment activation constants.

Conventions are those of e3nn 0.4.4, which the reference pins (env/env_equiformer.yml:358): real basis ordered
m = -l..l with y as the polar axis, wigner_3j of unit Frobenius norm obtained from the SU(2) Clebsch-Gordan
coefficients through the real<->complex change of basis.  Independent of `oracle/` (which restates the same
published formulas separately); tests/test_so3.py cross-checks the two and the analytic known answers.
"""
import functools
import math

import numpy as np

# e3nn.math.normalize2mom Monte-Carlo constants (1e6 fp64 normal samples, generator seed 0); recomputed and
# compared in tests/test_so3.py.  [ref: nets/fast_activation.py:25, nets/graph_attention_transformer.py:463-464]
C_SILU = 1.6791767923989418
C_SIGMOID = 1.8467055342154763
C_SMOOTH_LEAKY_RELU_02 = 1.531320475574866


def _fact(n):
    return math.factorial(int(n))


def _su2_cg(j1, j2, j3):
    """<j1 m1 j2 m2 | j3 m3> as an array [2j1+1, 2j2+1, 2j3+1] (integer j only)."""
    out = np.zeros((2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1))
    pref_num = (2 * j3 + 1) * _fact(j3 + j1 - j2) * _fact(j3 - j1 + j2) * _fact(j1 + j2 - j3)
    pref_den = _fact(j1 + j2 + j3 + 1)
    for m1 in range(-j1, j1 + 1):
        for m2 in range(-j2, j2 + 1):
            m3 = m1 + m2
            if abs(m3) > j3:
                continue
            num = pref_num * _fact(j3 + m3) * _fact(j3 - m3)
            den = pref_den * _fact(j1 - m1) * _fact(j1 + m1) * _fact(j2 - m2) * _fact(j2 + m2)
            s = 0.0
            for v in range(max(-j1 + j2 + m3, -j1 + m1, 0), min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3) + 1):
                s += (-1.0) ** (v + j2 + m2) * (
                    _fact(j2 + j3 + m1 - v) * _fact(j1 - m1 + v)
                    / (_fact(v) * _fact(j3 - j1 + j2 - v) * _fact(j3 + m3 - v) * _fact(v + j1 - j2 - m3))
                )
            out[j1 + m1, j2 + m2, j3 + m3] = math.sqrt(num / den) * s
    return out


def _real_to_complex(l):
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    r = 1.0 / math.sqrt(2.0)
    for m in range(1, l + 1):
        q[l - m, l + m] = r
        q[l - m, l - m] = -1j * r
        q[l + m, l + m] = (-1) ** m * r
        q[l + m, l - m] = 1j * (-1) ** m * r
    q[l, l] = 1.0
    return (-1j) ** l * q


@functools.lru_cache(maxsize=None)
def wigner_3j(l1, l2, l3):
    """Real-basis Wigner 3j symbol, shape [2l1+1, 2l2+1, 2l3+1], unit Frobenius norm (numpy fp64)."""
    if not abs(l2 - l3) <= l1 <= l2 + l3:
        raise ValueError("triangle rule violated for (%d,%d,%d)" % (l1, l2, l3))
    c = _su2_cg(l1, l2, l3).astype(np.complex128)
    q1, q2, q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    c = np.einsum("ij,kl,mn,ikn->jlm", q1, q2, np.conj(q3.T), c)
    assert np.abs(c.imag).max() < 1e-9
    c = np.ascontiguousarray(c.real)
    return c / np.linalg.norm(c)


def path_table(l1, l2, l3):
    """Dense CG table of one 'uvu'/'uvw' path with the e3nn codegen normalisation for
    irrep_normalization='component', path_normalization='none': sqrt(2*l3+1) * w3j."""
    return math.sqrt(2 * l3 + 1) * wigner_3j(l1, l2, l3)
