// Tables shared by the fused SeparableFCTP kernels (sfc.hip: exact-fp32 MFMA generation; sfcx.hip: split-precision bf16
// matrix cores): degree / slab descriptors built from the public path table, workgroup ordering, flop / byte counts.
#pragma once
#include "common.h"
#include <cstring>

namespace sfc {

constexpr int SFC_MAX_DEG = 4;
constexpr int SFC_MAX_SLABS = 72;   // 32-channel slabs over all output degrees (DTP width <= 3072 channels)
constexpr int SFC_MAX_D1 = 7;       // l1 <= 3
constexpr int SFC_LDS_LIMIT = 160 * 1024;

struct SfcSlab {
  int x_off;     // offset of (input segment l1, channel c0) in the x row
  int w_off;     // offset of the slab's 32 weights in the w row
  short m_off;   // offset of the path's coupling matrix in the coupling row
  short x_mul;   // multiplicity of the input segment (stride between components i)
  short d1;      // 2*l1+1
  short deg;     // index into deg[]
};

struct SfcDeg {
  const float* W;   // [K, N1] row-major: main consumer
  const float* W2;  // [K, N2] row-major: second scalar consumer (degree 0 only), may be null
  float* dW;        // weight-gradient targets (same shapes), accumulated
  float* dW2;
  int l3, d3, K, N1, N2, Ncat;
  int out1_off;    // offset of the degree segment inside an out1 row
  int m_base, m_len;  // block of the coupling row holding the matrices of all paths into l3
  int slab0, nslab;
};

struct SfcCommon {
  const float* x;
  const float* coupling;
  const float* w;  // may be null (unit path weights)
  int x_ld, m_ld, w_ld, E;
  float* o1;  // out1 (forward, written) / d_out1 (backward, read)
  int ld1;
  float* o2;  // out2 / d_out2, may be null
  int ld2;
  int ndeg;
  SfcDeg deg[SFC_MAX_DEG];
  SfcSlab slab[SFC_MAX_SLABS];
};

// Workgroup ordering.  All three kernels launch a 1-D grid of nx * ny workgroups, where the ny workgroups that share
// an x (an edge tile / edge chunk) re-read the same rows of x, w, coupling and d_out.  The hardware deals consecutive
// workgroup ids round-robin over the 8 XCDs, each with its own L2, so with order 1 the launch id b is first turned
// into a logical id that runs fastest *inside* an XCD (b % 8 = XCD, b / 8 = position), and logical neighbours -- the ny
// sharers of one x -- land on the same L2 at about the same time: their re-reads are L2 hits instead of trips to
// HBM / infinity cache.  order 0 = plain x-fastest enumeration (first version), order 2 = y-fastest without the XCD
// step (kept for A/B measurements).
struct SfcOrder {
  int mode, nx, ny, per_xcd;
};
#ifndef X_LPT_BATCH
#define X_LPT_BATCH 8  // tiles per XCD and batch of order mode 4
#endif
// false: surplus workgroup of the padded grid
__device__ __forceinline__ bool order_xy(const SfcOrder& o, int b, int& x, int& y) {
  if (o.mode == 0) {
    y = b / o.nx, x = b - y * o.nx;
    return true;
  }
  if (o.mode == 3) {
    // XCD k = b % 8 owns the tiles x = 8 xi + k (all items of a tile still share one L2); inside an XCD the launch ids run
    // ITEM-major, and the host sorts the items by cost, heaviest first: the long items of every tile start in the first round
    // and the short ones fill the slots they free (round 6: with y fastest the last tiles' long items started last and the
    // launch ended in a tail at low occupancy -- the forward's items take 23 / 50 / 55 us in a 150-us launch).  per_xcd =
    // tiles per XCD.
    const int k = b & 7, s = b >> 3;
    y = s / o.per_xcd;
    x = 8 * (s - y * o.per_xcd) + k;
    return x < o.nx && y < o.ny;
  }
  if (o.mode == 4) {
    // As mode 3, in BATCHES of X_LPT_BATCH tiles per XCD: heaviest items first inside a batch, batch after batch.  With the
    // whole launch item-major (mode 3) the seven data-gradient items of a tile ran at unrelated times and each re-read the
    // tile's d_out rows from HBM: 728 MB per launch against 332 MB tile-major (rocprofv3 FETCH_SIZE, profiles/r06/r06_u); a
    // batch keeps the rows of its tiles in the XCD's L2 while its items run, and the tail of short items still forms at the
    // end of every batch.
    const int k = b & 7, s = b >> 3;
    const int per_batch = X_LPT_BATCH * o.ny;
    const int bt = s / per_batch, r = s - bt * per_batch;
    y = r / X_LPT_BATCH;
    x = 8 * (bt * X_LPT_BATCH + (r - y * X_LPT_BATCH)) + k;
    return x < o.nx && y < o.ny;
  }
  const int L = (o.mode == 1) ? (b & 7) * o.per_xcd + (b >> 3) : b;
  if (L >= o.nx * o.ny) return false;
  x = L / o.ny, y = L - x * o.ny;
  return true;
}


// Fill the degree / slab tables.  o1_irreps: one segment per output degree; n2 extra scalar columns on degree 0.
inline int build_common(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* P,
                 const float* const* Wl, const float* W2, float* const* dWl, float* dW2, float* o1,
                 const eqf_irreps* o1_irreps, float* o2, int n2, int E, SfcCommon& C) {
  if (!x || !coupling || !P || !o1 || !o1_irreps) return EQF_E_BADARG;
  if (o1_irreps->nseg < 1 || o1_irreps->nseg > SFC_MAX_DEG || P->npaths < 1 || P->npaths > EQF_MAX_PATHS)
    return EQF_E_BADARG;
  if ((n2 > 0) != (o2 != nullptr)) return EQF_E_BADARG;
  memset(&C, 0, sizeof C);
  C.x = x, C.coupling = coupling, C.w = w;
  C.x_ld = P->in_dim, C.m_ld = P->m_numel, C.w_ld = P->w_numel, C.E = E;
  C.o1 = o1, C.ld1 = irreps_dim(*o1_irreps);
  C.o2 = o2, C.ld2 = n2;
  C.ndeg = o1_irreps->nseg;
  int off = 0, nslab = 0;
  bool have0 = false;
  for (int s = 0; s < o1_irreps->nseg; ++s) {
    SfcDeg& D = C.deg[s];
    D.l3 = o1_irreps->l[s], D.d3 = 2 * D.l3 + 1;
    D.N1 = o1_irreps->mul[s];
    D.N2 = (D.l3 == 0) ? n2 : 0;
    have0 |= D.l3 == 0;
    D.Ncat = D.N1 + D.N2;
    D.out1_off = off;
    off += D.N1 * D.d3;
    if (D.l3 > 3 || D.Ncat % 32 != 0 || D.N1 % 32 != 0) return EQF_E_UNSUPPORTED;
    D.W = Wl ? Wl[D.l3] : nullptr;
    D.dW = dWl ? dWl[D.l3] : nullptr;
    D.W2 = (D.l3 == 0) ? W2 : nullptr;
    D.dW2 = (D.l3 == 0) ? dW2 : nullptr;
    int K = 0, m_lo = 1 << 30, m_hi = 0;
    for (int p = 0; p < P->npaths; ++p)
      if (P->l3[p] == D.l3) {
        K = P->out_k[p];
        const int len = (2 * P->l1[p] + 1) * D.d3;
        if (P->m_off[p] < m_lo) m_lo = P->m_off[p];
        if (P->m_off[p] + len > m_hi) m_hi = P->m_off[p] + len;
      }
    if (K == 0) return EQF_E_BADARG;  // an output degree nothing feeds
    if (K % 32 != 0) return EQF_E_UNSUPPORTED;
    D.K = K, D.m_base = m_lo, D.m_len = m_hi - m_lo;
    D.slab0 = nslab, D.nslab = K / 32;
    if (nslab + D.nslab > SFC_MAX_SLABS) return EQF_E_UNSUPPORTED;
    for (int q = 0; q < D.nslab; ++q) C.slab[nslab + q].d1 = 0;
    for (int p = 0; p < P->npaths; ++p) {
      if (P->l3[p] != D.l3) continue;
      if (P->mul[p] % 32 != 0 || P->out_ch[p] % 32 != 0 || P->l1[p] > 3) return EQF_E_UNSUPPORTED;
      if (P->m_off[p] + 49 > 32767 || P->mul[p] > 32767) return EQF_E_UNSUPPORTED;
      for (int c = 0; c < P->mul[p]; c += 32) {
        SfcSlab& S = C.slab[nslab + (P->out_ch[p] + c) / 32];
        S.d1 = (short)(2 * P->l1[p] + 1);
        S.x_off = P->in_off[p] + c;
        S.x_mul = (short)P->mul[p];
        S.w_off = P->w_off[p] + c;
        S.m_off = (short)P->m_off[p];
        S.deg = (short)s;
      }
    }
    for (int q = 0; q < D.nslab; ++q)
      if (C.slab[nslab + q].d1 == 0) return EQF_E_BADARG;
    nslab += D.nslab;
  }
  if (n2 > 0 && !have0) return EQF_E_BADARG;
  return 0;
}

inline int max_d1(const SfcCommon& C) {
  int m = 1;
  for (int d = 0; d < C.ndeg; ++d)
    for (int q = 0; q < C.deg[d].nslab; ++q) m = C.slab[C.deg[d].slab0 + q].d1 > m ? C.slab[C.deg[d].slab0 + q].d1 : m;
  return m;
}

inline double sfc_flops(const SfcCommon& C) {
  double f = 0;
  for (int d = 0; d < C.ndeg; ++d) f += 2.0 * C.E * C.deg[d].d3 * (double)C.deg[d].K * C.deg[d].Ncat;
  return f;
}
inline double sfc_bytes(const SfcCommon& C) {  // x, w, coupling in; out rows out; weights
  double b = 4.0 * C.E * ((double)C.x_ld + (C.w ? C.w_ld : 0) + C.m_ld + C.ld1 + C.ld2);
  for (int d = 0; d < C.ndeg; ++d) b += 4.0 * C.deg[d].K * C.deg[d].Ncat;
  return b;
}


}  // namespace sfc
