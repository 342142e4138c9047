"""Index maps of the row-major operand tiles of csrc/sfcx.hip (tile_fetch / tile_to_frag / tile_store, the d_out tiles of
xb_pair_mma, tile16_fetch / _put / _get), restated in numpy: (i) global -> LDS -> registers composes to exactly the
per-lane elements the kernels used to load directly, (ii) every global instruction covers whole 128-byte lines (8 rows x
128 bytes), (iii) the fragment-wise LDS reads are free of bank conflicts and the row-wise writes at most 2-way (model: 64
banks of 4 bytes; a 16-byte access is served per group of 16 lanes, a 4-byte access per 64 lanes).
The constants below mirror the kernel source; tests/test_gpu_sfcx.py is what checks the kernels themselves."""
import re
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "equiformer_amd", "csrc", "sfcx_common.h")).read()
XT_LD = int(re.search(r"constexpr int XT_LD = (\d+);", SRC).group(1))
LANES = np.arange(64)
R, HI = LANES & 31, LANES >> 5
C8, RR = LANES & 7, LANES >> 3


def _conflict_b128(addr_words):
    """addr_words[lane] = first 4-byte word of a 16-byte access; worst number of different words on one bank within a group
    of 16 lanes (1 = conflict free)"""
    worst = 1
    for g in range(4):
        words = (addr_words[16 * g:16 * g + 16, None] + np.arange(4)[None, :]).ravel()
        banks = {}
        for w in set(words.tolist()):
            banks[w % 64] = banks.get(w % 64, 0) + 1
        worst = max(worst, max(banks.values()))
    return worst


def _banks_ok_b128(addr_words):
    return _conflict_b128(addr_words) == 1


def _banks_ok_b32(addr_words):
    """one 4-byte word per lane: distinct banks, or the same word (broadcast)"""
    seen = {}
    for a in addr_words:
        b = a % 64
        if b in seen and seen[b] != a:
            return False
        seen[b] = a
    return True


def test_constants_match_the_source():
    assert XT_LD == 36 and "16 * XT_LD" in SRC and "32 * XT_LD" in SRC


def test_tile32_global_instructions_cover_whole_lines_and_compose_to_the_fragment_layout():
    ld = 480  # floats per row of the source tensor
    src = np.arange(40 * ld, dtype=np.int64).reshape(40, ld)  # element ids
    e0, col0 = 3, 64  # tile = rows e0 .. e0+31, columns col0 .. col0+31
    T = np.full(32 * XT_LD, -1, dtype=np.int64)
    for it in range(4):  # tile_fetch + the LDS write of tile_to_frag / xb_pair_mma
        rows = RR + 8 * it
        first = rows * ld + 4 * C8 + col0  # relative element of each lane's 16 bytes
        # 8 consecutive lanes cover one row's 128 bytes
        for row in np.unique(rows):
            lanes = np.where(rows == row)[0]
            assert sorted(4 * C8[lanes]) == list(range(0, 32, 4))
        wr = rows * XT_LD + 4 * C8
        assert _conflict_b128(wr) <= 2  # row pitch 36: the second row of a 16-lane group wraps onto 4 banks of the first
        for lane in LANES:
            T[wr[lane]:wr[lane] + 4] = src[e0 + rows[lane], col0 + 4 * C8[lane]:col0 + 4 * C8[lane] + 4]
    # C-fragment layout of x / w / dx / dw (tile_to_frag): lane (r, hi) holds row r, columns 8 g4 + 4 hi + j
    for g4 in range(4):
        rd = R * XT_LD + 8 * g4 + 4 * HI
        assert _banks_ok_b128(rd)
        for lane in LANES:
            got = T[rd[lane]:rd[lane] + 4]
            want = src[e0 + R[lane], col0 + 8 * g4 + 4 * HI[lane]:col0 + 8 * g4 + 4 * HI[lane] + 4]
            assert (got == want).all()
    # B-fragment layout of the d_out pair (xb_pair_mma): block h = columns 16 h .. 16 h + 15, lane supplies k = 8 hi .. 8 hi + 7
    for h in range(2):
        for half in range(2):
            rd = R * XT_LD + 16 * h + 8 * HI + 4 * half
            assert _banks_ok_b128(rd)
            for lane in LANES:
                got = T[rd[lane]:rd[lane] + 4]
                c = col0 + 16 * h + 8 * HI[lane] + 4 * half
                assert (got == src[e0 + R[lane], c:c + 4]).all()


def test_tile16_of_the_weight_gradient():
    ld = 576
    src = np.arange(64 * ld, dtype=np.int64).reshape(64, ld)
    e_lo, col0, elast = 20, 96, 29  # rows past elast re-read row elast
    T = np.full(16 * XT_LD, -1, dtype=np.int64)
    for k in range(2):  # t0, t1
        rows = RR + 8 * k
        wr = rows * XT_LD + 4 * C8
        assert _conflict_b128(wr) <= 2
        for lane in LANES:
            e = min(e_lo + rows[lane], elast)
            T[wr[lane]:wr[lane] + 4] = src[e, col0 + 4 * C8[lane]:col0 + 4 * C8[lane] + 4]
    for j in range(8):  # tile16_get: lane (r, hi) = column r of the rows 8 hi + j  (what `cb[ct][eo[j] * ld + r]` loaded)
        rd = (8 * HI + j) * XT_LD + R
        assert _banks_ok_b32(rd)
        for lane in LANES:
            e = min(e_lo + 8 * HI[lane] + j, elast)
            assert T[rd[lane]] == src[e, col0 + R[lane]]
