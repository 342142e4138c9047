// Dense contractions of the Equiformer hot path on the gfx950 matrix cores.
//
// Everything here is exact fp32 (v_mfma_f32_32x32x2_f32: 157 TFLOP/s peak on MI355X, bit-equal to an fmaf
// chain), because the parity bar against the reference is 1e-4 relative in fp32.
//
// One kernel family, two shapes:
//   rows kernel : C[i,n] = sum_k A[i,k] B[k,n]     i = feature rows (nodes / edges x (2l+1)), large
//                 A operand either read from memory (two-level rows, k contiguous) or GENERATED on the fly by
//                 the depth-wise tensor product (x, coupling, w) -> the 3136-wide DTP output never exists in HBM.
//                 B operand either [K,N] (forward) or [N,K] (data gradient).
//   tn kernel   : C[m,n] += sum_i A[i,m] B[i,n]    weight gradients; reduction over the rows, split over the
//                 grid and over the 4 waves of a block, fp32 atomics into C.
// LDS tiles are stored k-major (T[k][x]) so that the MFMA operand fetch (lane = x, one k per half-wave) is a
// conflict-free ds_read_b32; sources that are contiguous along k are transposed while staging (odd row stride),
// sources contiguous along x are staged with ds_write_b128 (stride = 4 mod 8 floats).
// The K loop is software pipelined: the global loads of step k+1 are issued before the MFMAs of step k, so that
// HBM/L2 latency hides behind the 64-cycle fp32 MFMAs even at one or two workgroups per CU (graphs here are small:
// ~25 k edges per 128-molecule batch, i.e. only a few hundred row tiles per launch).
#include "common.h"
#include "prof.h"
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

int g_gemm_exp = 0;

// Development switches (phases of a kernel switched off, per-phase cycle counters) cost scalar instructions inside the hot
// loops: they are compiled in only with -DEQF_DEV_SWITCHES=1 (EQF_EXTRA_FLAGS="-DEQF_DEV_SWITCHES=1" python -m
// equiformer_amd.build); in the product build the eqf_*_debug_exp bits that act inside kernels are no-ops.
#ifndef EQF_DEV_SWITCHES
#define EQF_DEV_SWITCHES 0
#endif
#if EQF_DEV_SWITCHES
#define GEMM_OFF(g, bit) ((g).exp & (bit))
#else
#define GEMM_OFF(g, bit) false
#endif

namespace {

constexpr int BK = 32;
constexpr int NTHREADS = 256;
constexpr int MAX_SLABS = 32;   // K (channels of one DTP output degree) <= 1024
constexpr int ROWS_PAIRS = 8;   // (edge, channel) pairs generated per thread and K step in the rows kernel (<= 64 edges / tile)
constexpr int TN_PAIRS = 4;     // ... in the tn kernel (<= 32 edges per reduction step)
constexpr int MAX_MTILE = 6656; // floats of coupling staged in LDS per row tile

struct Rows {
  const float* base;
  int d, ld, inner;
};

struct DtpSlab {  // one 32-channel slab of the generated A operand
  int d1;         // 2*l1+1
  int x_off;      // offset of (segment l1, channel u0) in the x row
  int x_mul;      // multiplicity of that segment (stride between components i)
  int w_off;      // offset of the slab's weights in the w row
  int m_off;      // offset of the path's coupling matrix inside the degree-l3 block of the coupling row
};

struct DtpA {
  const float* x;
  const float* coupling;
  const float* w;  // may be null
  int x_ld, m_ld, w_ld;
  int d3;      // 2*l3+1
  int ept;     // edges per tile (rows kernel: per M tile; tn kernel: per reduction step)
  int m_base;  // start of the degree-l3 block in the coupling row
  int m_len;   // its length
  DtpSlab slabs[MAX_SLABS];
};

// ------------------------------------------------------------------------------------------------
// operand loaders: issue() = global -> registers (for the NEXT K step), commit() = registers -> LDS
// ------------------------------------------------------------------------------------------------
// source rows run over the tile's x index and are contiguous along k  ->  T[k][x], SX odd
template <int BX, int SX>
struct LoaderContigK {
  float4 v[BX / 32];
  long roff[BX / 32];  // row offsets of this thread's rows: the same for every K step, computed once (init)
  __device__ __forceinline__ void init(const Rows& R, int x0, int xcnt) {
    const int xr0 = threadIdx.x >> 3;
#pragma unroll
    for (int pass = 0; pass < BX / 32; ++pass) {
      const int xr = xr0 + pass * 32;
      roff[pass] = row_off2(x0 + (xr < xcnt ? xr : 0), R.d, R.ld, R.inner);
    }
  }
  __device__ __forceinline__ void issue(const Rows& R, int x0, int xcnt, int k0, int K, bool vec) {
    const int t = threadIdx.x;
    const int kq = t & 7, xr0 = t >> 3;
#pragma unroll
    for (int pass = 0; pass < BX / 32; ++pass) {
      const int xr = xr0 + pass * 32;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      const int krem = K - (k0 + kq * 4);
      if (xr < xcnt && krem > 0) {
        const float* p = R.base + roff[pass] + k0 + kq * 4;
        if (vec && krem >= 4) {
          r = *reinterpret_cast<const float4*>(p);
        } else {
          r.x = p[0];
          if (krem > 1) r.y = p[1];
          if (krem > 2) r.z = p[2];
          if (krem > 3) r.w = p[3];
        }
      }
      v[pass] = r;
    }
  }
  __device__ __forceinline__ void commit(float* __restrict__ T) const {
    const int t = threadIdx.x;
    const int kq = t & 7, xr0 = t >> 3;
#pragma unroll
    for (int pass = 0; pass < BX / 32; ++pass) {
      float* q = T + (kq * 4) * SX + xr0 + pass * 32;
      q[0] = v[pass].x;
      q[SX] = v[pass].y;
      q[2 * SX] = v[pass].z;
      q[3 * SX] = v[pass].w;
    }
  }
};

// source rows run over the reduction index k (two-level) and are contiguous along x  ->  T[k][x], SX % 4 == 0
template <int BX, int SX>
struct LoaderNatural {
  static constexpr int XQ = BX / 4;
  static constexpr int RPP = NTHREADS / XQ;
  static constexpr int NP = BK / RPP;
  float4 v[NP];
  __device__ __forceinline__ void issue(const Rows& R, int k0, int kcnt, int x0, int X, bool vec) {
    const int t = threadIdx.x;
    const int xq = t % XQ, kr0 = t / XQ;
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int kr = kr0 + pass * RPP;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      const int x = x0 + xq * 4;
      const int xrem = X - x;
      if (kr < kcnt && xrem > 0) {
        const float* p = R.base + row_off2(k0 + kr, R.d, R.ld, R.inner) + x;
        if (vec && xrem >= 4) {
          r = *reinterpret_cast<const float4*>(p);
        } else {
          r.x = p[0];
          if (xrem > 1) r.y = p[1];
          if (xrem > 2) r.z = p[2];
          if (xrem > 3) r.w = p[3];
        }
      }
      v[pass] = r;
    }
  }
  __device__ __forceinline__ void commit(float* __restrict__ T) const {
    const int t = threadIdx.x;
    const int xq = t % XQ, kr0 = t / XQ;
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) *reinterpret_cast<float4*>(T + (kr0 + pass * RPP) * SX + xq * 4) = v[pass];
  }
};

// DTP-generated operand: thread owns channel u = t & 31 of the slab and the edges el = (t >> 5) + 8 p.
// issue(): loads w[e, slab, u] and x[e, l1, 0..d1), u] of every owned edge (coalesced over u).
// commit(): out[el, m3] = w * sum_i M[el][i, m3] * x[i], with the tile's coupling block M read from LDS.
template <int NP>
struct LoaderDtp {
  float w[NP];
  float xv[NP][7];
  DtpSlab s;
  __device__ __forceinline__ void issue(const DtpA& D, int slab, int e0, int ecnt) {
    s = D.slabs[slab];
    const int u = threadIdx.x & 31, g = threadIdx.x >> 5;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int el = g + 8 * p;
      if (el < ecnt) {
        const long e = e0 + el;
        w[p] = D.w ? D.w[e * D.w_ld + s.w_off + u] : 1.0f;
        const float* xp = D.x + e * D.x_ld + s.x_off + u;
#pragma unroll
        for (int i = 0; i < 7; ++i) xv[p][i] = (i < s.d1) ? xp[i * s.x_mul] : 0.f;
      }
    }
  }
  // rows kernel: As[u][el*d3 + m3]   (row_stride = 1, col_stride = SA)
  // tn kernel  : As[el*d3 + m3][sl*32 + u]   (row_stride = SA, col_stride = 1, col0 = sl*32)
  __device__ __forceinline__ void commit(float* __restrict__ T, const float* __restrict__ Mt, int m_stride, int d3,
                                         int ecnt, int row_stride, int col_stride, int col0) const {
    const int u = threadIdx.x & 31, g = threadIdx.x >> 5;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int el = g + 8 * p;
      if (el < ecnt) {
        const float* mp = Mt + (long)el * m_stride + s.m_off;
        float* q = T + (col0 + u) * col_stride + (el * d3) * row_stride;
        for (int m3 = 0; m3 < d3; ++m3) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < 7; ++i)
            if (i < s.d1) acc = fmaf(mp[i * d3 + m3], xv[p][i], acc);
          q[m3 * row_stride] = acc * w[p];
        }
      }
    }
  }
};

__device__ __forceinline__ void stage_coupling(float* __restrict__ Mt, const DtpA& D, int e0, int ecnt) {
  const int n = ecnt * D.m_len;
  for (int i = threadIdx.x; i < n; i += NTHREADS) {
    const int el = i / D.m_len, j = i - el * D.m_len;
    Mt[i] = D.coupling[(long)(e0 + el) * D.m_ld + D.m_base + j];
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA over one staged K step
// ------------------------------------------------------------------------------------------------
template <int TM, int TN, int SA, int SB>
__device__ __forceinline__ void mma_step(const float* __restrict__ As, const float* __restrict__ Bs, int wm0, int wn0,
                                         int kbeg, int kend, f32x16 (&acc)[TM][TN]) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hi = lane >> 5;
#pragma unroll 8
  for (int kk = kbeg; kk < kend; kk += 2) {
    float a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = As[(kk + hi) * SA + wm0 + i * 32 + r];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + hi) * SB + wn0 + j * 32 + r];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------
// rows kernel
// ------------------------------------------------------------------------------------------------
struct RowsArgs {
  Rows A, B, C;
  const float* bias;
  int M, N, K;
  int rows_per_tile;
  int accumulate;
  int vecA, vecB;
  int exp;  // development aid (eqf_gemm_debug_exp): 1 no stores, 2 no MFMA
  DtpA dtp;
};

enum { A_MEM = 0, A_DTP = 1 };
enum { B_KN = 0, B_NK = 1 };

struct RowsP {  // one problem of a grouped launch (A operand from memory)
  Rows A, B, C;
  const float* bias;
  int M, N, K;
  int rows_per_tile;
  int accumulate;
  int vecA, vecB;
  int exp;
};
constexpr int MAX_GROUP = 8;
struct RowsGroup {
  int n;
  RowsP p[MAX_GROUP];
};

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE, class ArgsT>
__device__ __forceinline__ void gemm_rows_body(const ArgsT& g, const int bx, const int by) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
  constexpr int SA = BM + 1;
  constexpr int SB = (BMODE == B_KN) ? BN + 4 : BN + 1;
  __shared__ __attribute__((aligned(16))) float As[BK * SA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * SB];
  __shared__ __attribute__((aligned(16))) float Mt[(AMODE == A_DTP) ? MAX_MTILE : 4];

  const int m0 = bx * g.rows_per_tile;
  const int n0 = by * BN;
  const int mcnt = min(g.rows_per_tile, g.M - m0);
  const int ncnt = min(BN, g.N - n0);
  const int wave = threadIdx.x >> 6;
  const int wm0 = (wave / WN) * (TM * 32), wn0 = (wave % WN) * (TN * 32);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  LoaderContigK<BM, SA> la;
  LoaderDtp<ROWS_PAIRS> ld;
  LoaderNatural<BN, SB> lbn;
  LoaderContigK<BN, SB> lbk;
  int e0 = 0, ecnt = 0;
  if constexpr (AMODE == A_DTP) {
    e0 = bx * g.dtp.ept;
    ecnt = mcnt / g.dtp.d3;
    stage_coupling(Mt, g.dtp, e0, ecnt);
    ld.issue(g.dtp, 0, e0, ecnt);
  } else {
    la.init(g.A, m0, mcnt);
    la.issue(g.A, m0, mcnt, 0, g.K, g.vecA);
  }
  if (BMODE == B_KN) {
    lbn.issue(g.B, 0, min(BK, g.K), n0, g.N, g.vecB);
  } else {
    lbk.init(g.B, n0, ncnt);
    lbk.issue(g.B, n0, ncnt, 0, g.K, g.vecB);
  }
  if (AMODE == A_DTP) __syncthreads();  // coupling tile visible

  for (int k0 = 0; k0 < g.K; k0 += BK) {
    if constexpr (AMODE == A_DTP)
      ld.commit(As, Mt, g.dtp.m_len, g.dtp.d3, ecnt, 1, SA, 0);
    else
      la.commit(As);
    if (BMODE == B_KN)
      lbn.commit(Bs);
    else
      lbk.commit(Bs);
    __syncthreads();
    const int k1 = k0 + BK;
    if (k1 < g.K) {
      if constexpr (AMODE == A_DTP)
        ld.issue(g.dtp, k1 / BK, e0, ecnt);
      else
        la.issue(g.A, m0, mcnt, k1, g.K, g.vecA);
      if (BMODE == B_KN)
        lbn.issue(g.B, k1, min(BK, g.K - k1), n0, g.N, g.vecB);
      else
        lbk.issue(g.B, n0, ncnt, k1, g.K, g.vecB);
    }
    if (!GEMM_OFF(g, 2)) mma_step<TM, TN, SA, SB>(As, Bs, wm0, wn0, 0, BK, acc);
    __syncthreads();
  }
  if (GEMM_OFF(g, 1)) return;

  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hi = lane >> 5;
  const bool flat_c = g.C.d == 1;  // wave-uniform: plain rows need no two-level split
  const SmallDiv cdiv(g.C.d);
  float* const cbase = const_cast<float*>(g.C.base);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // rows of this lane in the 32-row tile: rb + (q & 3) + 8 (q >> 2); one exact division for rb, SmallDiv for the rest
    const int rb = wm0 + i * 32 + 4 * hi;
    long off0;
    int rem0 = 0;
    if (flat_c) {
      off0 = (long)(m0 + rb) * g.C.ld;
    } else {
      const int qb = (m0 + rb) / g.C.d;
      rem0 = (m0 + rb) - qb * g.C.d;
      off0 = (long)qb * g.C.ld;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + j * 32 + r;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int dr = (q & 3) + 8 * (q >> 2);
        if (rb + dr < mcnt) {
          long off;
          if (flat_c) {
            off = off0 + (long)dr * g.C.ld;
          } else {
            const int t = rem0 + dr, dq = cdiv.div(t);
            off = off0 + (long)dq * g.C.ld + (long)(t - dq * g.C.d) * g.C.inner;
          }
          float* p = cbase + off + col;
          float v = acc[i][j][q] + bv;
          if (g.accumulate) v += *p;
          *p = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE>
__global__ __launch_bounds__(NTHREADS) void gemm_rows_kernel(const RowsArgs g) {
  gemm_rows_body<BM, BN, WM, WN, AMODE, BMODE>(g, blockIdx.x, blockIdx.y);
}

// up to MAX_GROUP independent problems (the per-degree GEMMs of one irreps linear) in ONE launch: blockIdx.z picks the
// problem; the node-level linears are launch / latency bound (2304 rows), so 3-4x fewer launches and 3-4x more
// workgroups in flight per launch is what matters for them
template <int BM, int BN, int WM, int WN, int BMODE>
__global__ __launch_bounds__(NTHREADS) void gemm_rows_group_kernel(const RowsGroup g) {
  const RowsP& P = g.p[blockIdx.z];
  if ((int)blockIdx.x * P.rows_per_tile >= P.M || (int)blockIdx.y * BN >= P.N) return;
  gemm_rows_body<BM, BN, WM, WN, A_MEM, BMODE>(P, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// tn kernel (weight gradients): C[m,n] += sum_rows A[row,m] B[row,n]
// ------------------------------------------------------------------------------------------------
struct TnArgs {
  Rows A, B;
  float* C;
  int ldc;
  int M, N, R;
  int rows_per_step;    // reduction rows staged per K step (<= BK)
  int steps_per_split;  // K steps handled by one blockIdx.z
  int vecA, vecB;
  float* csA;  // optional: csA[m] += sum_rows A[row, m]  (bias gradient when A is dy), may be null
  float* csB;  // optional: csB[n] += sum_rows B[row, n]  (bias gradient when B is dy), may be null
  DtpA dtp;
};

struct TnP {  // one problem of a grouped launch (A operand from memory)
  Rows A, B;
  float* C;
  int ldc;
  int M, N, R;
  int rows_per_step;
  int steps_per_split;
  int vecA, vecB;
  float* csA;
  float* csB;
};
struct TnGroup {
  int n;
  TnP p[MAX_GROUP];
};

template <int BM, int BN, int WM, int WN, int WK, int AMODE, class ArgsT>
__device__ __forceinline__ void gemm_tn_body(const ArgsT& g, const int bx, const int by, const int bz) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN * WK == 4 && TM >= 1 && TN >= 1, "4 waves");
  constexpr int SA = BM + 4, SB = BN + 4;
  constexpr int SL = BM / 32;
  __shared__ __attribute__((aligned(16))) float As[BK * SA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * SB];

  const int m0 = bx * BM, n0 = by * BN;
  const int wave = threadIdx.x >> 6;
  const int wk = wave % WK;
  const int wmn = wave / WK;
  const int wm0 = (wmn / WN) * (TM * 32), wn0 = (wmn % WN) * (TN * 32);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int total_steps = (g.R + g.rows_per_step - 1) / g.rows_per_step;
  const int s_beg = bz * g.steps_per_split;
  const int s_end = min(total_steps, s_beg + g.steps_per_split);
  const int nslab = min(BM, g.M - m0) / 32;

  LoaderNatural<BM, SA> la;
  LoaderDtp<TN_PAIRS> ld[SL];
  LoaderNatural<BN, SB> lb;
  const bool do_csB = g.csB != nullptr && bx == 0;
  const bool do_csA = AMODE == A_MEM && g.csA != nullptr && by == 0;
  float cs = 0.f;
  static_assert(BM + BN <= NTHREADS, "disjoint thread ranges for the two column sums");

  // in DTP mode the (tiny) coupling rows are read straight from global memory (L1 resident, lane-uniform)
  auto issue = [&](int s) {
    const int r0 = s * g.rows_per_step;
    const int rcnt = min(g.rows_per_step, g.R - r0);
    if constexpr (AMODE == A_MEM) {
      la.issue(g.A, r0, rcnt, m0, g.M, g.vecA);
    } else {
#pragma unroll
      for (int sl = 0; sl < SL; ++sl)
        if (sl < nslab) ld[sl].issue(g.dtp, m0 / 32 + sl, s * g.dtp.ept, rcnt / g.dtp.d3);
    }
    lb.issue(g.B, r0, rcnt, n0, g.N, g.vecB);
  };

  if constexpr (AMODE == A_DTP) {
    for (int i = threadIdx.x; i < BK * SA; i += NTHREADS) As[i] = 0.f;  // padding rows / absent slabs stay zero
    __syncthreads();
  }
  if (s_beg < s_end) issue(s_beg);
  for (int s = s_beg; s < s_end; ++s) {
    const int r0 = s * g.rows_per_step;
    const int rcnt = min(g.rows_per_step, g.R - r0);
    if constexpr (AMODE == A_MEM) {
      la.commit(As);
    } else {
      const int ecnt = rcnt / g.dtp.d3;
      const float* Mg = g.dtp.coupling + (long)(s * g.dtp.ept) * g.dtp.m_ld + g.dtp.m_base;
      // rows of edges that are not part of this (last, partial) step must be zero
      if (ecnt < g.dtp.ept) {
        for (int i = threadIdx.x; i < (g.dtp.ept - ecnt) * g.dtp.d3 * BM; i += NTHREADS) {
          const int rr = ecnt * g.dtp.d3 + i / BM, cc = i % BM;
          As[rr * SA + cc] = 0.f;
        }
      }
#pragma unroll
      for (int sl = 0; sl < SL; ++sl)
        if (sl < nslab) ld[sl].commit(As, Mg, g.dtp.m_ld, g.dtp.d3, ecnt, SA, 1, sl * 32);
    }
    lb.commit(Bs);
    __syncthreads();
    if (s + 1 < s_end) issue(s + 1);
    // bias gradients ride along: the staged tiles already hold the rows whose column sums a separate colsum launch
    // would re-read (rows past rcnt are zero in the tiles); one tile column per thread, first tile row / column only
    if (do_csB && (int)threadIdx.x < BN) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) cs += Bs[k * SB + threadIdx.x];
    } else if (do_csA && (int)threadIdx.x >= NTHREADS - BM) {
      const int c = threadIdx.x - (NTHREADS - BM);
#pragma unroll 8
      for (int k = 0; k < BK; ++k) cs += As[k * SA + c];
    }
    mma_step<TM, TN, SA, SB>(As, Bs, wm0, wn0, wk * (BK / WK), (wk + 1) * (BK / WK), acc);
    __syncthreads();
  }
  if (do_csB && (int)threadIdx.x < BN && n0 + (int)threadIdx.x < g.N) atomicAdd(g.csB + n0 + threadIdx.x, cs);
  if (do_csA && !(do_csB && (int)threadIdx.x < BN) && (int)threadIdx.x >= NTHREADS - BM &&
      m0 + (int)threadIdx.x - (NTHREADS - BM) < g.M)
    atomicAdd(g.csA + m0 + threadIdx.x - (NTHREADS - BM), cs);

  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + j * 32 + r;
      if (col >= g.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi;
        if (row < g.M) atomicAdd(g.C + (long)row * g.ldc + col, acc[i][j][q]);
      }
    }
}

template <int BM, int BN, int WM, int WN, int WK, int AMODE>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(const TnArgs g) {
  gemm_tn_body<BM, BN, WM, WN, WK, AMODE>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// grouped weight gradients: grid.z = sum of the problems' split counts; zoff[] maps blockIdx.z to (problem, split)
struct TnGroupLaunch {
  TnGroup g;
  int zoff[MAX_GROUP + 1];
};
template <int BM, int BN, int WM, int WN, int WK>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_group_kernel(const TnGroupLaunch L) {
  int pi = 0;
  while (pi + 1 < L.g.n && (int)blockIdx.z >= L.zoff[pi + 1]) ++pi;
  const TnP& P = L.g.p[pi];
  if ((int)blockIdx.x * BM >= P.M || (int)blockIdx.y * BN >= P.N) return;
  gemm_tn_body<BM, BN, WM, WN, WK, A_MEM>(P, blockIdx.x, blockIdx.y, (int)blockIdx.z - L.zoff[pi]);
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool rows_vec_ok(const float* base, const eqf_rows& r) {
  return aligned16(base) && (r.ld % 4 == 0) && (r.inner % 4 == 0);
}

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE>
void launch_rows_cfg(const RowsArgs& a, hipStream_t st) {
  dim3 grid(eqf_cdiv(a.M, a.rows_per_tile), eqf_cdiv(a.N, BN));
  hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, WM, WN, AMODE, BMODE>), grid, dim3(NTHREADS), 0, st, a);
}

template <int AMODE, int BMODE>
int launch_rows(RowsArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return 0;
  const int bn = (a.N > 64 && !(g_gemm_exp & 4)) ? 128 : (a.N > 32 ? 64 : 32);
  // 64-row tiles when 128-row tiles would leave most of the 256 CUs without a workgroup
  const int d = AMODE == A_DTP ? a.dtp.d3 : 1;
  const long tiles128 = (long)eqf_cdiv(a.M, (128 / d) * d) * eqf_cdiv(a.N, bn);
  int bm = (bn >= 64 && tiles128 < 4096) ? 64 : 128;
  if (AMODE == A_DTP) {
    if (bn >= 64 && ((128 / d) > 8 * ROWS_PAIRS || (128 / d) * a.dtp.m_len > MAX_MTILE)) bm = 64;
    a.dtp.ept = bm / d;
    a.rows_per_tile = a.dtp.ept * d;
    if (a.dtp.ept > 8 * ROWS_PAIRS || a.dtp.ept * a.dtp.m_len > MAX_MTILE) return EQF_E_UNSUPPORTED;
  } else {
    a.rows_per_tile = bm;
  }
  a.exp = g_gemm_exp;
  char name[96];
  snprintf(name, sizeof name, "gemm_rows_%dx%d_%s_%s", bm, bn, AMODE == A_DTP ? "dtp" : "mem",
           BMODE == B_KN ? "kn" : "nk");
  // algorithmic bytes: A (or the DTP inputs x,w,coupling) + B + C, each once
  const double a_bytes = AMODE == A_DTP
                             ? 4.0 * (double)(a.M / a.dtp.d3) * (a.dtp.x_ld + (a.dtp.w ? a.dtp.w_ld : 0) + a.dtp.m_ld)
                             : 4.0 * (double)a.M * a.K;
  const int pid = eqf_prof_begin(name, st, 2.0 * a.M * (double)a.N * a.K,
                                 a_bytes + 4.0 * (double)a.K * a.N + 4.0 * (double)a.M * a.N);
  if (bm == 128) {
    if (bn == 128)
      launch_rows_cfg<128, 128, 2, 2, AMODE, BMODE>(a, st);
    else if (bn == 64)
      launch_rows_cfg<128, 64, 2, 2, AMODE, BMODE>(a, st);
    else
      launch_rows_cfg<128, 32, 4, 1, AMODE, BMODE>(a, st);
  } else {
    if (bn == 128)
      launch_rows_cfg<64, 128, 2, 2, AMODE, BMODE>(a, st);
    else
      launch_rows_cfg<64, 64, 2, 2, AMODE, BMODE>(a, st);
  }
  eqf_prof_end(pid, st);
  EQF_CHECK_LAUNCH();
  return 0;
}

template <int AMODE>
int launch_tn(TnArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || a.R <= 0) return 0;
  const bool bm64 = !(a.M % 64 != 0 && a.M < 256);
  const bool bn64 = !(a.N % 64 != 0 && a.N < 256);
  const int BMv = bm64 ? 64 : 32, BNv = bn64 ? 64 : 32;
  const int tiles = eqf_cdiv(a.M, BMv) * eqf_cdiv(a.N, BNv);
  const int total_steps = eqf_cdiv(a.R, a.rows_per_step);
  int ksplit = 2048 / tiles;
  if (ksplit < 1) ksplit = 1;
  int max_split = eqf_cdiv(total_steps, 8);  // at least 8 K steps per block
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1) ksplit = 1;
  a.steps_per_split = eqf_cdiv(total_steps, ksplit);
  ksplit = eqf_cdiv(total_steps, a.steps_per_split);
  dim3 grid(eqf_cdiv(a.M, BMv), eqf_cdiv(a.N, BNv), ksplit);
  char name[96];
  snprintf(name, sizeof name, "gemm_tn_%dx%d_%s", BMv, BNv, AMODE == A_DTP ? "dtp" : "mem");
  const double a_bytes = AMODE == A_DTP
                             ? 4.0 * (double)(a.R / a.dtp.d3) * (a.dtp.x_ld + (a.dtp.w ? a.dtp.w_ld : 0) + a.dtp.m_ld)
                             : 4.0 * (double)a.R * a.M;
  const int pid = eqf_prof_begin(name, st, 2.0 * a.M * (double)a.N * a.R,
                                 a_bytes + 4.0 * (double)a.R * a.N + 4.0 * (double)a.M * a.N);
  if (bm64 && bn64)
    hipLaunchKernelGGL((gemm_tn_kernel<64, 64, 2, 2, 1, AMODE>), grid, dim3(NTHREADS), 0, st, a);
  else if (bm64)
    hipLaunchKernelGGL((gemm_tn_kernel<64, 32, 2, 1, 2, AMODE>), grid, dim3(NTHREADS), 0, st, a);
  else if (bn64)
    hipLaunchKernelGGL((gemm_tn_kernel<32, 64, 1, 2, 2, AMODE>), grid, dim3(NTHREADS), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<32, 32, 1, 1, 4, AMODE>), grid, dim3(NTHREADS), 0, st, a);
  eqf_prof_end(pid, st);
  EQF_CHECK_LAUNCH();
  return 0;
}

// Build the per-slab table of output degree l3; returns K (channels of that degree) or <0 on error.
// The coupling row is laid out degree-major (layout.DtpTable), so the matrices of all paths that feed l3 form one
// contiguous block [m_base, m_base + m_len).
int build_dtp(const eqf_dtp_paths* P, int l3, const float* x, const float* coupling, const float* w, DtpA& D) {
  D.x = x;
  D.coupling = coupling;
  D.w = w;
  D.x_ld = P->in_dim;
  D.m_ld = P->m_numel;
  D.w_ld = P->w_numel;
  D.d3 = 2 * l3 + 1;
  int K = 0, m_lo = 1 << 30, m_hi = 0;
  for (int p = 0; p < P->npaths; ++p)
    if (P->l3[p] == l3) {
      K = P->out_k[p];
      const int len = (2 * P->l1[p] + 1) * (2 * l3 + 1);
      if (P->m_off[p] < m_lo) m_lo = P->m_off[p];
      if (P->m_off[p] + len > m_hi) m_hi = P->m_off[p] + len;
    }
  if (K == 0) return 0;
  if (K % 32 != 0 || K / 32 > MAX_SLABS) return EQF_E_UNSUPPORTED;
  D.m_base = m_lo;
  D.m_len = m_hi - m_lo;
  int m_sum = 0;
  for (int s = 0; s < K / 32; ++s) D.slabs[s].d1 = 0;
  for (int p = 0; p < P->npaths; ++p) {
    if (P->l3[p] != l3) continue;
    if (P->mul[p] % 32 != 0 || P->out_ch[p] % 32 != 0) return EQF_E_UNSUPPORTED;
    if (P->l1[p] > 3) return EQF_E_UNSUPPORTED;
    m_sum += (2 * P->l1[p] + 1) * (2 * l3 + 1);
    for (int c = 0; c < P->mul[p]; c += 32) {
      DtpSlab& s = D.slabs[(P->out_ch[p] + c) / 32];
      s.d1 = 2 * P->l1[p] + 1;
      s.x_off = P->in_off[p] + c;
      s.x_mul = P->mul[p];
      s.w_off = P->w_off[p] + c;
      s.m_off = P->m_off[p] - m_lo;
    }
  }
  if (m_sum != D.m_len) return EQF_E_BADARG;  // coupling row not degree-major
  for (int s = 0; s < K / 32; ++s)
    if (D.slabs[s].d1 == 0) return EQF_E_BADARG;
  return K;
}

}  // namespace

extern "C" {

int eqf_gemm_debug_exp(int mask) {
  g_gemm_exp = mask;
  return 0;
}

int eqf_gemm_nn(const float* A, eqf_rows ra, const float* B, int ldb, float* C, eqf_rows rc, const float* bias, int M,
                int N, int K, int accumulate, void* stream) {
  if (!A || !B || !C || ra.d < 1 || rc.d < 1) return EQF_E_BADARG;
  RowsArgs a{};
  a.A = {A, ra.d, ra.ld, ra.inner};
  a.B = {B, 1, ldb, 0};
  a.C = {C, rc.d, rc.ld, rc.inner};
  a.bias = bias;
  a.M = M, a.N = N, a.K = K, a.accumulate = accumulate;
  a.vecA = rows_vec_ok(A, ra);
  a.vecB = aligned16(B) && ldb % 4 == 0;
  return launch_rows<A_MEM, B_KN>(a, (hipStream_t)stream);
}

int eqf_gemm_nt(const float* A, eqf_rows ra, const float* B, int ldb, float* C, eqf_rows rc, const float* bias, int M,
                int N, int K, int accumulate, void* stream) {
  if (!A || !B || !C || ra.d < 1 || rc.d < 1) return EQF_E_BADARG;
  RowsArgs a{};
  a.A = {A, ra.d, ra.ld, ra.inner};
  a.B = {B, 1, ldb, 0};
  a.C = {C, rc.d, rc.ld, rc.inner};
  a.bias = bias;
  a.M = M, a.N = N, a.K = K, a.accumulate = accumulate;
  a.vecA = rows_vec_ok(A, ra);
  a.vecB = aligned16(B) && ldb % 4 == 0;
  return launch_rows<A_MEM, B_NK>(a, (hipStream_t)stream);
}

int eqf_gemm_tn(const float* A, eqf_rows ra, const float* B, eqf_rows rb, float* C, int ldc, int M, int N, int R,
                void* stream) {
  return eqf_gemm_tn_colsum(A, ra, B, rb, C, ldc, M, N, R, nullptr, nullptr, stream);
}

int eqf_gemm_tn_colsum(const float* A, eqf_rows ra, const float* B, eqf_rows rb, float* C, int ldc, int M, int N, int R,
                       float* colsum_a, float* colsum_b, void* stream) {
  if (!A || !B || !C || ra.d < 1 || rb.d < 1) return EQF_E_BADARG;
  TnArgs a{};
  a.csA = colsum_a, a.csB = colsum_b;
  a.A = {A, ra.d, ra.ld, ra.inner};
  a.B = {B, rb.d, rb.ld, rb.inner};
  a.C = C, a.ldc = ldc, a.M = M, a.N = N, a.R = R;
  a.rows_per_step = BK;
  a.vecA = rows_vec_ok(A, ra);
  a.vecB = rows_vec_ok(B, rb);
  return launch_tn<A_MEM>(a, (hipStream_t)stream);
}

int eqf_gemm_group(const eqf_gemm_desc* d, int n, void* stream) {
  if (!d || n < 1 || n > MAX_GROUP) return EQF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  // rows problems (kind 0: C = A B, kind 1: C = A B^T) share one launch per kind; tn problems (kind 2) another
  for (int kind = 0; kind < 2; ++kind) {
    RowsGroup G{};
    int maxm = 0, maxn = 0, bn_need = 32;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
      if (d[i].kind != kind) continue;
      if (!d[i].A || !d[i].B || !d[i].C || d[i].ra.d < 1 || d[i].rc.d < 1) return EQF_E_BADARG;
      if (d[i].M <= 0 || d[i].N <= 0) continue;
      RowsP& P = G.p[G.n++];
      P.A = {d[i].A, d[i].ra.d, d[i].ra.ld, d[i].ra.inner};
      P.B = {d[i].B, 1, d[i].ldb, 0};
      P.C = {d[i].C, d[i].rc.d, d[i].rc.ld, d[i].rc.inner};
      P.bias = d[i].bias;
      P.M = d[i].M, P.N = d[i].N, P.K = d[i].K, P.accumulate = d[i].accumulate;
      P.vecA = rows_vec_ok(d[i].A, d[i].ra);
      P.vecB = aligned16(d[i].B) && d[i].ldb % 4 == 0;
      P.exp = 0;
      P.rows_per_tile = 64;
      if (P.N > 32) bn_need = 64;
      flops += 2.0 * P.M * (double)P.N * P.K;
      bytes += 4.0 * ((double)P.M * P.K + (double)P.K * P.N + (double)P.M * P.N);
    }
    if (G.n == 0) continue;
    const int rpt = bn_need == 64 ? 64 : 128;
    for (int i = 0; i < G.n; ++i) {
      G.p[i].rows_per_tile = rpt;
      const int tm = eqf_cdiv(G.p[i].M, rpt), tn = eqf_cdiv(G.p[i].N, bn_need);
      if (tm > maxm) maxm = tm;
      if (tn > maxn) maxn = tn;
    }
    dim3 grid(maxm, maxn, G.n);
    const int pid = eqf_prof_begin(kind == 0 ? "gemm_group_kn" : "gemm_group_nk", st, flops, bytes);
    if (kind == 0) {
      if (bn_need == 64)
        hipLaunchKernelGGL((gemm_rows_group_kernel<64, 64, 2, 2, B_KN>), grid, dim3(NTHREADS), 0, st, G);
      else
        hipLaunchKernelGGL((gemm_rows_group_kernel<128, 32, 4, 1, B_KN>), grid, dim3(NTHREADS), 0, st, G);
    } else {
      if (bn_need == 64)
        hipLaunchKernelGGL((gemm_rows_group_kernel<64, 64, 2, 2, B_NK>), grid, dim3(NTHREADS), 0, st, G);
      else
        hipLaunchKernelGGL((gemm_rows_group_kernel<128, 32, 4, 1, B_NK>), grid, dim3(NTHREADS), 0, st, G);
    }
    eqf_prof_end(pid, st);
    EQF_CHECK_LAUNCH();
  }
  {
    TnGroupLaunch L{};
    int maxm = 0, maxn = 0, z = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
      if (d[i].kind != 2 && d[i].kind != 3) continue;
      if (!d[i].A || !d[i].B || !d[i].C || d[i].ra.d < 1 || d[i].rc.d < 1) return EQF_E_BADARG;
      if (d[i].M <= 0 || d[i].N <= 0 || d[i].K <= 0) continue;
      // kind 2:  C[M,N] (plain, leading dimension ldb) += sum over K rows of A[row, 0:M]^T B'[row, 0:N], B' given as (C-field rows rc)
      TnP& P = L.g.p[L.g.n];
      P.A = {d[i].A, d[i].ra.d, d[i].ra.ld, d[i].ra.inner};
      P.B = {d[i].B, d[i].rc.d, d[i].rc.ld, d[i].rc.inner};
      P.C = d[i].C, P.ldc = d[i].ldb, P.M = d[i].M, P.N = d[i].N, P.R = d[i].K;
      P.rows_per_step = BK;
      P.vecA = rows_vec_ok(d[i].A, d[i].ra);
      P.vecB = rows_vec_ok(d[i].B, d[i].rc);
      // kind 2: `bias` = accumulator of the column sums of B'; kind 3: of A (the operand that is dy; may be null)
      P.csA = d[i].kind == 3 ? const_cast<float*>(d[i].bias) : nullptr;
      P.csB = d[i].kind == 2 ? const_cast<float*>(d[i].bias) : nullptr;
      const int tiles = eqf_cdiv(P.M, 64) * eqf_cdiv(P.N, 64);
      const int total_steps = eqf_cdiv(P.R, BK);
      int ksplit = 1024 / (tiles > 0 ? tiles : 1);
      const int max_split = eqf_cdiv(total_steps, 8);
      if (ksplit > max_split) ksplit = max_split;
      if (ksplit < 1) ksplit = 1;
      P.steps_per_split = eqf_cdiv(total_steps, ksplit);
      ksplit = eqf_cdiv(total_steps, P.steps_per_split);
      L.zoff[L.g.n] = z;
      z += ksplit;
      L.g.n++;
      if (eqf_cdiv(P.M, 64) > maxm) maxm = eqf_cdiv(P.M, 64);
      if (eqf_cdiv(P.N, 64) > maxn) maxn = eqf_cdiv(P.N, 64);
      flops += 2.0 * P.M * (double)P.N * P.R;
      bytes += 4.0 * ((double)P.R * P.M + (double)P.R * P.N + (double)P.M * P.N);
    }
    if (L.g.n > 0) {
      L.zoff[L.g.n] = z;
      dim3 grid(maxm, maxn, z);
      const int pid = eqf_prof_begin("gemm_group_tn", st, flops, bytes);
      hipLaunchKernelGGL((gemm_tn_group_kernel<64, 64, 2, 2, 1>), grid, dim3(NTHREADS), 0, st, L);
      eqf_prof_end(pid, st);
      EQF_CHECK_LAUNCH();
    }
  }
  return 0;
}

int eqf_dtp_linear_fwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                       const float* const* Wl, const float* bias0, float* out, const eqf_irreps* out_irreps, int E,
                       void* stream) {
  if (!x || !coupling || !paths || !Wl || !out || !out_irreps) return EQF_E_BADARG;
  const int Dout = irreps_dim(*out_irreps);
  int off = 0;
  for (int s = 0; s < out_irreps->nseg; ++s) {
    const int l3 = out_irreps->l[s], N = out_irreps->mul[s], d3 = 2 * l3 + 1;
    RowsArgs a{};
    const int K = build_dtp(paths, l3, x, coupling, w, a.dtp);
    if (K < 0) return K;
    if (K == 0) return EQF_E_BADARG;  // an output degree nothing feeds
    a.B = {Wl[l3], 1, N, 0};
    a.C = {out + off, d3, Dout, N};
    a.bias = (l3 == 0) ? bias0 : nullptr;
    a.M = E * d3, a.N = N, a.K = K, a.accumulate = 0;
    a.vecA = 0;
    a.vecB = aligned16(Wl[l3]) && N % 4 == 0;
    int rc = launch_rows<A_DTP, B_KN>(a, (hipStream_t)stream);
    if (rc) return rc;
    off += N * d3;
  }
  return 0;
}

int eqf_dtp_linear_wgrad(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                         const float* d_out, const eqf_irreps* out_irreps, float* const* dWl, int E, void* stream) {
  if (!x || !coupling || !paths || !d_out || !dWl || !out_irreps) return EQF_E_BADARG;
  const int Dout = irreps_dim(*out_irreps);
  int off = 0;
  for (int s = 0; s < out_irreps->nseg; ++s) {
    const int l3 = out_irreps->l[s], N = out_irreps->mul[s], d3 = 2 * l3 + 1;
    TnArgs a{};
    const int K = build_dtp(paths, l3, x, coupling, w, a.dtp);
    if (K < 0) return K;
    if (K == 0) return EQF_E_BADARG;
    a.dtp.ept = BK / d3;
    a.B = {d_out + off, d3, Dout, N};
    a.C = dWl[l3], a.ldc = N, a.M = K, a.N = N, a.R = E * d3;
    a.rows_per_step = a.dtp.ept * d3;
    a.vecA = 0;
    a.vecB = aligned16(d_out + off) && Dout % 4 == 0 && N % 4 == 0;
    int rc = launch_tn<A_DTP>(a, (hipStream_t)stream);
    if (rc) return rc;
    off += N * d3;
  }
  return 0;
}

}  // extern "C"
