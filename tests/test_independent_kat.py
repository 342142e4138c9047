"""Known-answer tests whose expected values do NOT come from this repository's own SO(3) code (the KATs of
tests/test_oracle_kat.py pin conventions against values derived in the same container).  Everything here is built from
scipy's complex spherical harmonics (Condon-Shortley phase, z polar) and numerical quadrature:

  * the real harmonics of e3nn 0.4.4 [dependency of the reference, called at nets/graph_attention_transformer.py:869]
    are the textbook real harmonics Y_lm, m = -l..l, with the axes relabelled (x_std, y_std, z_std) = (z, x, y)
    ("y is the polar axis") and 'component' normalisation (sum_m Y_lm^2 = 2l+1).  The oracle's and the HIP kernels'
    closed-form polynomials (l <= 3) must reproduce them;
  * the real Wigner-3j tensors of unit Frobenius norm are, for l1+l2+l3 even, the normalised real Gaunt integrals
    int Y_l1i Y_l2j Y_l3k dOmega (sign included); for l1+l2+l3 odd they are antisymmetric invariant tensors, checked
    through invariance under rotations whose Wigner-D matrices are again obtained from the scipy harmonics;
  * e3nn's normalize2mom constants are Monte-Carlo estimates (1e6 samples) of (E f(z)^2)^-1/2, z ~ N(0,1): the exact
    value by Gauss-Hermite quadrature must lie within the Monte-Carlo error of the constants the kernels use.
"""
import math

import numpy as np
import pytest
import torch
from scipy import special

from equiformer_amd import so3
from oracle import e3

SL = [slice(0, 1), slice(1, 4), slice(4, 9), slice(9, 16)]


def _real_sh_scipy(lmax, xyz):
    """[n, (lmax+1)^2] real harmonics in e3nn's axis convention and 'component' normalisation, from scipy only."""
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    xs, ys, zs = z, x, y                      # (x_std, y_std, z_std) = (z, x, y)
    r = np.sqrt(xs * xs + ys * ys + zs * zs)
    theta = np.arccos(np.clip(zs / r, -1.0, 1.0))   # polar angle from z_std
    phi = np.arctan2(ys, xs)                        # azimuth
    cols = []
    for l in range(lmax + 1):
        for m in range(-l, l + 1):
            Y = special.sph_harm_y(l, abs(m), theta, phi)  # complex, orthonormal on the sphere
            if m == 0:
                v = Y.real
            elif m > 0:
                v = math.sqrt(2.0) * (-1.0) ** m * Y.real
            else:
                v = math.sqrt(2.0) * (-1.0) ** m * Y.imag
            cols.append(v * math.sqrt(4.0 * math.pi))    # orthonormal -> component normalisation
    return np.stack(cols, 1)


def _sphere_quadrature(n_theta=24, n_phi=48):
    """Gauss-Legendre in cos(theta) x uniform in phi: exact for polynomials of degree < 2 n_theta; weights sum to 1."""
    ct, wt = np.polynomial.legendre.leggauss(n_theta)
    phi = (np.arange(n_phi) + 0.5) * 2.0 * math.pi / n_phi
    st = np.sqrt(1.0 - ct * ct)
    pts = np.stack([np.outer(st, np.cos(phi)).ravel(), np.outer(st, np.sin(phi)).ravel(), np.outer(ct, np.ones(n_phi)).ravel()], 1)
    w = np.outer(wt, np.ones(n_phi)).ravel() / (2.0 * n_phi)
    return pts, w


def test_spherical_harmonics_match_scipy():
    g = np.random.default_rng(0)
    xyz = g.standard_normal((200, 3))
    want = _real_sh_scipy(3, xyz)
    got = e3.spherical_harmonics(3, torch.from_numpy(xyz)).numpy()
    assert np.abs(got - want).max() < 1e-12


def test_scipy_harmonics_are_orthonormal_under_the_quadrature():
    pts, w = _sphere_quadrature()
    Y = _real_sh_scipy(3, pts)
    gram = (Y * w[:, None]).T @ Y
    assert np.abs(gram - np.eye(16)).max() < 1e-12   # component normalisation: mean over the sphere of Y_i Y_j = delta_ij


@pytest.mark.parametrize("l1,l2,l3", [(0, 0, 0), (0, 1, 1), (0, 2, 2), (0, 3, 3), (1, 1, 0), (1, 1, 2), (1, 2, 1), (1, 2, 3),
                                      (1, 3, 2), (2, 0, 2), (2, 1, 1), (2, 1, 3), (2, 2, 0), (2, 2, 2), (2, 3, 1), (2, 3, 3),
                                      (3, 1, 2), (3, 2, 1), (3, 2, 3), (3, 3, 0), (3, 3, 2)])
def test_wigner_3j_is_the_normalised_real_gaunt_tensor(l1, l2, l3):
    pts, w = _sphere_quadrature()
    Y = _real_sh_scipy(3, pts)
    G = np.einsum("z,zi,zj,zk->ijk", w, Y[:, SL[l1]], Y[:, SL[l2]], Y[:, SL[l3]])
    G /= np.linalg.norm(G)
    assert np.abs(e3.wigner_3j(l1, l2, l3).numpy() - G).max() < 1e-12          # oracle
    assert np.abs(np.asarray(so3.wigner_3j(l1, l2, l3)) - G).max() < 1e-12     # tables the HIP kernels consume


def _wigner_d_from_scipy(R):
    g = np.random.default_rng(1)
    x = g.standard_normal((96, 3))
    Y, YR = _real_sh_scipy(3, x), _real_sh_scipy(3, x @ R.T)
    return [np.linalg.lstsq(Y[:, SL[l]], YR[:, SL[l]], rcond=None)[0].T for l in range(4)]


@pytest.mark.parametrize("l1,l2,l3", [(1, 1, 1), (1, 2, 2), (2, 1, 2), (2, 2, 1), (2, 2, 3), (2, 3, 2), (3, 2, 2), (1, 3, 3),
                                      (3, 1, 3), (3, 3, 1), (3, 3, 3), (2, 3, 3)])
def test_odd_wigner_3j_are_invariant_unit_antisymmetric_tensors(l1, l2, l3):
    """l1+l2+l3 odd: no Gaunt integral (it vanishes).  The space of rotation-invariant tensors in l1 x l2 x l3 is one
    dimensional, so invariance + unit norm fix the tensor up to its sign; (-1)^(l1+l2+l3) symmetry under exchanging equal
    degrees and the e3nn sign of the Levi-Civita case (1,1,1) are checked on top."""
    g = np.random.default_rng(2)
    q, r = np.linalg.qr(g.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    D = _wigner_d_from_scipy(q)
    for C in (e3.wigner_3j(l1, l2, l3).numpy(), np.asarray(so3.wigner_3j(l1, l2, l3))):
        assert abs(np.linalg.norm(C) - 1.0) < 1e-12
        Cr = np.einsum("ia,jb,kc,abc->ijk", D[l1], D[l2], D[l3], C)
        assert np.abs(C - Cr).max() < 1e-9
        if l1 == l2:
            assert np.abs(C + C.transpose(1, 0, 2)).max() < 1e-12   # odd sum: antisymmetric in the equal pair
    if (l1, l2, l3) == (1, 1, 1):
        # e3nn: w3j(1,1,1)[i,j,k] = eps_ijk / sqrt(6) in its (x, y, z) component order (the cross product)
        eps = np.zeros((3, 3, 3))
        for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
            eps[i, j, k], eps[j, i, k] = 1.0, -1.0
        assert np.abs(e3.wigner_3j(1, 1, 1).numpy() - eps / math.sqrt(6.0)).max() < 1e-12


def test_w3j_relative_signs_across_index_permutations():
    """e3nn builds every real w3j from the same SU(2) coefficients, so tensors of permuted degrees are transposes of one
    another up to (-1)^(l1+l2+l3): w3j(l1,l2,l3)[i,j,k] = (-1)^(l1+l2+l3) w3j(l2,l1,l3)[j,i,k] = w3j(l3,l1,l2)[k,i,j]."""
    for (a, b, c) in [(1, 2, 2), (1, 2, 3), (2, 3, 3), (1, 1, 2), (2, 2, 3)]:
        s = (-1.0) ** (a + b + c)
        C = e3.wigner_3j(a, b, c).numpy()
        assert np.abs(C - s * e3.wigner_3j(b, a, c).numpy().transpose(1, 0, 2)).max() < 1e-12
        assert np.abs(C - e3.wigner_3j(c, a, b).numpy().transpose(1, 2, 0)).max() < 1e-12


@pytest.mark.parametrize("name,fn,const", [
    ("silu", lambda z: z / (1.0 + np.exp(-z)), so3.C_SILU),
    ("sigmoid", lambda z: 1.0 / (1.0 + np.exp(-z)), so3.C_SIGMOID),
    ("smooth_leaky_relu_0.2", lambda z: 0.6 * z + 0.4 * z * (2.0 / (1.0 + np.exp(-z)) - 1.0), so3.C_SMOOTH_LEAKY_RELU_02)])
def test_normalize2mom_constants_against_gauss_hermite(name, fn, const):
    """(E f(z)^2)^-1/2 by 200-point Gauss-Hermite quadrature; e3nn's constant is a 1e6-sample Monte-Carlo estimate of it:
    relative standard error ~ sqrt(Var f^2 / 1e6) / (2 E f^2), a few 1e-3."""
    t, w = np.polynomial.hermite_e.hermegauss(200)   # weight exp(-z^2/2)
    w = w / math.sqrt(2.0 * math.pi)
    m2 = float((w * fn(t) ** 2).sum())
    m4 = float((w * fn(t) ** 4).sum())
    exact = m2 ** -0.5
    rel_se = math.sqrt(max(m4 - m2 * m2, 0.0) / 1e6) / (2.0 * m2)
    assert abs(const - exact) / exact < 5.0 * rel_se + 1e-6, (name, const, exact, rel_se)
    assert abs(const - exact) / exact > 1e-5   # and it really is the Monte-Carlo value, not the exact one
