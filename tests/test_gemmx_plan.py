"""Index logic of csrc/gemmx.hip restated in numpy (CPU; constants parsed from the source so that the two cannot drift apart):

  * the two-level row index (k // d, k % d) that LoaderKS WALKS (+1 per row, +adv per step) instead of dividing,
  * the clamped addresses of its unconditional loads stay inside the matrix and the values zeroed in commit() are exactly those
    outside it,
  * the one-dimensional grid: workgroup -> problem through the prefix table (sign-bit count), -> (split, tile m, tile n), every
    (problem, split, tile) exactly once, no empty workgroup,
  * the K split of the weight-gradient kernel covers every K step exactly once.

The kernels themselves are compared with fp32 / fp64 references on the GPU (tests/test_gpu_gemmx.py)."""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = open(os.path.join(HERE, "..", "equiformer_amd", "csrc", "gemmx.hip")).read()


def const(name):
    m = re.search(r"constexpr int %s = (\d+);" % name, SRC)
    assert m, name
    return int(m.group(1))


GX_BK, GX_T, GX_MAXP = const("GX_BK"), const("GX_T"), const("GX_MAXP")
MINSTEPS = int(re.search(r"static int g_gemmx_tn_minsteps = (\d+);", SRC).group(1))


def test_constants_are_the_documented_ones():
    assert (GX_BK, GX_T, GX_MAXP, MINSTEPS) == (32, 64, 24, 8)
    # the kernarg segment holds the whole group (4 KB limit): the static_assert of the source, restated
    assert "sizeof(GXP) * GX_MAXP + 4 * (2 * GX_MAXP + 3) <= 4000" in SRC


def walk_ks(K, d, ld, inner, k_first, adv, steps):
    """LoaderKS::init + issue for every thread group kg = 0..3 -> per step: (element offsets of the 8 loads, kleft)."""
    out = []
    for kg in range(4):
        kb = k_first + 8 * kg
        q, rem = kb // d, kb % d
        per_step = []
        for _ in range(steps):
            nv = K - kb
            base = q * ld + rem * inner if nv > 0 else 0
            wrap = ld - d * inner
            rel, r, offs = 0, rem, []
            for j in range(8):
                offs.append(base + (rel if j < nv else 0))
                rel += inner
                r += 1
                if r == d:
                    r, rel = 0, rel + wrap
            per_step.append((offs, nv, kb))
            q32 = adv // d
            r32 = adv - q32 * d
            kb, q, rem = kb + adv, q + q32, rem + r32
            if rem >= d:
                rem, q = rem - d, q + 1
        out.append(per_step)
    return out


def test_walked_two_level_index_equals_division():
    rng = np.random.default_rng(0)
    for d in (1, 3, 5, 7, 9):
        for adv in (GX_BK, 4 * GX_BK):
            inner = int(rng.integers(8, 200)) * 4
            ld = d * inner + int(rng.integers(0, 50)) * 4  # a row = d components of `inner` floats + other segments
            K = int(rng.integers(1, 700))
            k_first = int(rng.integers(0, 5)) * GX_BK
            steps = (K + adv - 1) // adv + 2  # walks past the end: the loads stay inside, nothing past K counts
            last = (K - 1) // d * ld + (K - 1) % d * inner
            for kg, per_step in enumerate(walk_ks(K, d, ld, inner, k_first, adv, steps)):
                for s, (offs, nv, kb) in enumerate(per_step):
                    assert kb == k_first + 8 * kg + s * adv
                    for j, off in enumerate(offs):
                        k = kb + j
                        assert 0 <= off <= last  # every load is inside the matrix, valid or not
                        if j < nv:  # what commit() keeps
                            assert k < K and off == (k // d) * ld + (k % d) * inner
                        else:       # zeroed: exactly the rows past the end
                            assert k >= K


def flat_problem(woff, b):
    """flat_problem() of the source: sum over j = 1 .. GX_MAXP - 1 of the sign bit of woff[j] - 1 - b (32-bit)."""
    pi = 0
    for j in range(1, GX_MAXP):
        v = np.int32(woff[j]) - np.int32(1) - np.int32(b)
        pi += int(np.uint32(v) >> np.uint32(31))
    return pi


def plan_rows(shapes):
    woff, wg = [], 0
    for M, N in shapes:
        woff.append(wg)
        wg += -(-M // GX_T) * -(-N // GX_T)
    n = len(shapes)
    return woff + [wg] + [np.iinfo(np.int32).max] * (GX_MAXP - n), wg


def test_flat_grid_of_the_rows_kernels_covers_every_tile_once():
    rng = np.random.default_rng(1)
    for n in (1, 3, 7, GX_MAXP):
        shapes = [(int(rng.integers(1, 12000)), int(rng.integers(1, 1000))) for _ in range(n)]
        woff, total = plan_rows(shapes)
        assert len(woff) == GX_MAXP + 1
        seen = set()
        step = max(1, total // 3000)
        for b in list(range(0, total, step)) + [total - 1] + [w for w in woff[1:n + 1] if w < total] + [w - 1 for w in woff[1:n + 1]]:
            pi = flat_problem(woff, b)
            assert pi == np.searchsorted(np.array(woff[1:n + 1]), b, side="right")
            M, N = shapes[pi]
            local = b - woff[pi]
            tiles_n = -(-N // GX_T)
            mt, nt = local // tiles_n, local % tiles_n
            assert 0 <= mt * GX_T < M and 0 <= nt * GX_T < N  # no empty workgroup
            seen.add((pi, mt, nt))
        if step == 1:
            assert len(seen) == total == sum(-(-M // GX_T) * -(-N // GX_T) for M, N in shapes)


def plan_tn(M, N, K, minsteps=MINSTEPS):
    tiles = -(-M // GX_T) * -(-N // GX_T)
    total_steps = -(-K // GX_BK)
    ksplit = max(1, min(1024 // max(tiles, 1), -(-total_steps // minsteps)))
    steps_per_split = -(-total_steps // ksplit)
    ksplit = -(-total_steps // steps_per_split)
    return tiles, total_steps, ksplit, steps_per_split


def test_weight_gradient_split_covers_every_k_step_once():
    rng = np.random.default_rng(2)
    cases = [(128, 128, 2304), (32, 32, 11520), (64, 960, 25354), (384, 128, 2304), (1, 1, 1), (64, 64, 33)]
    cases += [(int(rng.integers(1, 500)), int(rng.integers(1, 1000)), int(rng.integers(1, 40000))) for _ in range(40)]
    for M, N, K in cases:
        tiles, total_steps, ksplit, sps = plan_tn(M, N, K)
        assert 1 <= ksplit <= max(1, 1024 // tiles) or tiles > 1024
        cover = np.zeros(total_steps, dtype=int)
        for bz in range(ksplit):
            s_beg, s_end = bz * sps, min(total_steps, (bz + 1) * sps)
            assert s_beg < s_end  # no split without work (the kernel returns early otherwise; none is launched)
            cover[s_beg:s_end] += 1
        assert (cover == 1).all()
        if total_steps >= MINSTEPS and ksplit > 1:
            assert sps >= MINSTEPS  # at least g_gemmx_tn_minsteps steps per workgroup
        # workgroup -> (split, tile): tile fastest
        wg = tiles * ksplit
        tiles_n = -(-N // GX_T)
        decoded = {(loc // tiles, (loc % tiles) // tiles_n, (loc % tiles) % tiles_n) for loc in range(0, wg, max(1, wg // 500))}
        assert all(bz < ksplit and mt * GX_T < M and nt * GX_T < N for bz, mt, nt in decoded)
