// Per-degree / dense linears of the hot path on the bf16 matrix cores (split precision), grouped launches.
//
//   kind 0:  C[i,n] (=|+=) sum_k A[i,k] B[k,n] (+ bias[n])       LinearRS forward (weight [K,N]); nn.Linear data gradient
//   kind 1:  C[i,n] (=|+=) sum_k A[i,k] B[n,k] (+ bias[n])       LinearRS data gradient; nn.Linear forward (weight [N,K])
//   kind 2:  C[m,n] += sum_i A[i,m] B[i,n], colsum(B) optional    LinearRS weight gradient (+ bias gradient)
//   kind 3:  same, colsum(A) optional                            nn.Linear weight gradient (dW[out,in] = dy^T x, db = colsum dy)
//
// [ref: LinearRS / FullyConnectedTensorProductRescale, nets/tensor_product_rescale.py:125-136,171-174; torch.nn.Linear inside
//  RadialProfile, nets/radial_func.py:46-49; their autograd backward]
//
// Same contract as gemm.hip's eqf_gemm_group (exact-fp32 MFMA, kept as the cross-check and the `fp32` matrix mode); here the
// matrix steps run on v_mfma_f32_32x32x16_bf16 -- 16 k per instruction instead of 2, on the matrix pipe instead of the VALU's
// FMA lanes -- with fp32 operands split into bf16 planes exactly as in sfcx.hip:
//   mode 0 (split)   activations 2 planes, weights 3 planes, 5 products (two activation operands: 2 + 2 planes, 3 products)
//   mode 1 (bf16)    1 + 1 planes: plain bf16 operands, fp32 accumulation (BASELINE config #2 / the reference's AMP linears)
//   mode 2 (split6)  3 + 3 planes
// Why: the node-row linears (2 304 rows) are latency chains -- ~7 us fixed + 1.45 us per 32-deep K step with the fp32 MFMA
// (profiles/r03/r03_u_*), 104 launches = 2.7 ms of the 13 ms QM9 step; a K step here is two matrix instructions deep.
//
// Tiling: 64 x 64 output tile per workgroup of four waves (32 x 32 each), K in steps of 32 through a double-buffered LDS
// image [plane][row / column][k] with k contiguous (80-byte rows: conflict-free 16-byte fragment reads); the global loads of
// step s + 1 are in flight during the matrix instructions of step s (registers), one barrier per step.  Operands are split
// where they are staged (VALU, once per tile element).  Loaders:
//   kc  source contiguous along k (A rows; [N,K] weights): 16-byte loads along k, 8-byte LDS writes per plane
//   ks  source contiguous along the tile's row / column index (weights [K,N]; both operands of the weight gradient, whose
//       reduction index is the feature row): lane = column, eight 4-byte loads down the reduction index (each a coalesced
//       256-byte run), one 16-byte LDS write per plane; the bias gradient's column sums are taken from the fp32 registers.
// Compiled with -fno-slp-vectorize (no packed-FP32 VALU beside bf16 MFMAs, as sfcx.hip).
#include <climits>
#include "sfcx_common.h"

namespace {

constexpr int GX_BK = 32;
constexpr int GX_LDK = 40;       // bf16 elements per LDS row: 32 + 8 (80 bytes)
constexpr int GX_T = 64;         // tile edge (rows and columns)
constexpr int GX_ROW = GX_T * GX_LDK;
constexpr int GX_MAXP = 24;     // problems per launch (the argument block stays under the 4 KB kernarg limit)

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

struct GRows {
  const float* base;
  int d, ld, inner;
};

struct GXP {          // one problem
  GRows A, B, C;      // rows kinds: A rows x K, B plain (ld = ldb), C rows x N.  tn kinds: A: K rows x M, B: K rows x N
  const float* bias;  // rows kinds: bias[N] or null
  float* cs;          // tn kinds: column-sum accumulator (of B for kind 2, of A for kind 3) or null
  float* Cw;          // tn kinds: C plain [M,N]
  int ldc;
  int M, N, K;        // tn kinds: K = number of reduction rows
  int accumulate, kind;
  int vecA, vecB;
  int steps_per_split;
};
static_assert(sizeof(GXP) * GX_MAXP + 4 * (2 * GX_MAXP + 3) <= 4000, "kernarg segment");
struct GXGroup {
  int n;
  int zoff[GX_MAXP + 1];  // one-wave-per-tile kernels (development switch): blockIdx.z -> (problem, K chunk)
  // LDS-tiled kernels: ONE-dimensional grid, workgroup b belongs to problem i with woff[i] <= b < woff[i + 1] (entries past n
  // are INT_MAX).  A (max tiles m, max tiles n, problems) grid launched mostly EMPTY workgroups when the problems of a group
  // differ in shape -- 23 k workgroups for the ~1 200 of a 24-problem weight-gradient group, 146 us per launch.
  int woff[GX_MAXP + 1];
  GXP p[GX_MAXP];
};

__device__ __forceinline__ int flat_problem(const GXGroup& g, int b) {  // branch-free: the table comes in a few wide scalar loads
  int pi = 0;
#pragma unroll
  for (int j = 1; j < GX_MAXP; ++j) pi += (int)((unsigned)(g.woff[j] - 1 - b) >> 31);  // sign bit of woff[j] - 1 - b: b >= woff[j]
  return pi;
}

template <int NP>
__device__ __forceinline__ void split_n(const float* v, int n, __bf16 (*p)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j >= n) break;
    float r = v[j];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const __bf16 h = (__bf16)r;
      p[q][j] = h;
      if (q + 1 < NP) r = r - (float)h;
    }
  }
}

// ---- kc: 64 tile rows (two-level row index x0 + xr), 32 k contiguous in memory; thread = (row xr0 + 32 pass, k quad kq)
template <int NP>
struct LoaderKC {
  float4 v[2];
  long roff[2];
  bool rv[2];
  __device__ __forceinline__ void init(const GRows& R, int x0, int X) {
    const int xr0 = threadIdx.x >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int x = x0 + xr0 + 32 * pass;
      rv[pass] = x < X;
      roff[pass] = row_off2(rv[pass] ? x : x0, R.d, R.ld, R.inner);
    }
  }
  // Every load is UNCONDITIONAL (address clamped into the row; commit() zeroes what lies outside): a load behind a per-lane
  // branch is waited for at the end of that branch, which serialises the round trips of a step (measured on the weight-gradient
  // kernel: 16 branched loads per step = 2.5 us per step, profiles/r04/r04_y_tn_minsteps.txt).  The zeroing is in commit() so
  // that nothing between issue() and commit() -- the MFMAs of the previous step -- depends on the loads.
  int krem;  // of the step in flight: K - (k0 + 4 kq)
  template <bool VEC>  // a template parameter, not a kernel argument: the loads of a UNIFORM branch are waited for at its end too
  __device__ __forceinline__ void issue(const GRows& R, int k0, int K) {
    const int k = k0 + 4 * (threadIdx.x & 7);
    krem = K - k;
    if constexpr (VEC) {  // rows 16-byte aligned and K % 4 == 0: a quad is whole or absent
      const int kc = krem > 0 ? k : 0;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) v[pass] = *reinterpret_cast<const float4*>(R.base + roff[pass] + kc);
    } else {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const float* const p = R.base + roff[pass];
        v[pass] = make_float4(p[min(k, K - 1)], p[min(k + 1, K - 1)], p[min(k + 2, K - 1)], p[min(k + 3, K - 1)]);
      }
    }
  }
  __device__ __forceinline__ void commit(__bf16* __restrict__ T) const {  // T: [NP][64][GX_LDK]
    const int kq = threadIdx.x & 7, xr0 = threadIdx.x >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float f[4] = {v[pass].x, v[pass].y, v[pass].z, v[pass].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(rv[pass] && e < krem)) f[e] = 0.f;
      __bf16 p[NP][8];
      split_n<NP>(f, 4, p);
#pragma unroll
      for (int q = 0; q < NP; ++q)
        *reinterpret_cast<bf16x4_t*>(T + q * GX_ROW + (xr0 + 32 * pass) * GX_LDK + 4 * kq) =
            bf16x4_t{p[q][0], p[q][1], p[q][2], p[q][3]};
    }
  }
};

// ---- ks: 32 reduction rows (two-level index k0 + k) x 64 tile columns contiguous in memory; thread = (column x, k group kg)
// The two-level row index (k / d, k % d) is WALKED, not divided: one exact division in init(), then +1 per row and +32 per step
// with a wrap.  (A division per load -- 16 per step and thread -- made the K loop of the weight-gradient kernel ~860 VALU
// instructions per step with one wave per SIMD: 1.7 us per step whatever else the step did, profiles/r04/r04_aa_*.)
template <int NP>
struct LoaderKS {
  float v[8];
  float cs, cc;  // running column sum of the fp32 values this thread staged (bias gradients), Kahan-compensated: a bias gradient
                 // is a sum of thousands of cancelling terms, and its rounding noise sat AT the 1e-4 bar of the MD17 second-order
                 // test (profiles/r05/r05_r_l2_second_order_by_matrix_mode.txt)
  int kleft;  // of the step in flight: valid reduction rows of this thread's group of 8 (may be <= 0); < 0 for a column past X
  int kb, q, rem;  // this thread's first reduction row of the NEXT step, and its two-level index
  __device__ __forceinline__ void init(const GRows& R, int k_first) {
    cs = 0.f, cc = 0.f;
    kb = k_first + 8 * (threadIdx.x >> 6);
    q = kb / R.d;
    rem = kb - q * R.d;
  }
  // loads the step at kb (unconditional loads, as above) and moves on by GX_BK rows
  __device__ __forceinline__ void issue(const GRows& R, int K, int x0, int X) {
    const int x = x0 + (threadIdx.x & 63);
    const bool xv = x < X;
    const int nv = K - kb;  // rows of the group inside the matrix
    kleft = xv ? nv : -1;
    // a group that starts past the end reads row 0 (never used)
    const float* const p = R.base + (xv ? x : x0) + (nv > 0 ? (long)q * R.ld + (long)rem * R.inner : 0L);
    const int wrap = R.ld - R.d * R.inner;
    int rel = 0, r = rem;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = p[j < nv ? rel : 0];
      rel += R.inner;
      if (++r == R.d) r = 0, rel += wrap;
    }
    const int q32 = GX_BK / R.d, r32 = GX_BK - q32 * R.d;  // uniform
    kb += GX_BK, q += q32, rem += r32;
    if (rem >= R.d) rem -= R.d, ++q;
  }
  __device__ __forceinline__ void commit(__bf16* __restrict__ T, bool want_cs) {
    const int x = threadIdx.x & 63, kg = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j >= kleft) v[j] = 0.f;
    if (want_cs) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float y = v[j] - cc, t = cs + y;
        cc = (t - cs) - y, cs = t;
      }
    }
    __bf16 p[NP][8];
    split_n<NP>(v, 8, p);
#pragma unroll
    for (int q_ = 0; q_ < NP; ++q_)
      *reinterpret_cast<bf16x8*>(T + q_ * GX_ROW + x * GX_LDK + 8 * kg) =
          bf16x8{p[q_][0], p[q_][1], p[q_][2], p[q_][3], p[q_][4], p[q_][5], p[q_][6], p[q_][7]};
  }
};

template <int NA, int NB>
__device__ __forceinline__ void gx_mma(const __bf16* __restrict__ As, const __bf16* __restrict__ Bs, int wm0, int wn0,
                                       f32x16& acc) {
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int kt = 0; kt < GX_BK / 16; ++kt) {
    bf16x8 a[NA], b[NB];
#pragma unroll
    for (int q = 0; q < NA; ++q) a[q] = *reinterpret_cast<const bf16x8*>(As + q * GX_ROW + (wm0 + r) * GX_LDK + 16 * kt + 8 * hi);
#pragma unroll
    for (int q = 0; q < NB; ++q) b[q] = *reinterpret_cast<const bf16x8*>(Bs + q * GX_ROW + (wn0 + r) * GX_LDK + 16 * kt + 8 * hi);
    mma_terms<NA, NB>(a, b, acc);
  }
}

// Output tile of one wave (32 x 32 accumulator fragment: register q of lane (r, hi) = row (q & 3) + 8 (q >> 2) + 4 hi, column r)
// -> memory as 16-byte stores of whole 128-byte lines, 16 rows at a time through a wave-private LDS tile: 4 store instructions per
// tile instead of 16 that each write two half lines (round 6: the radial bank's widest layer is bound by its 681 MB of writes).
// Needs 16-byte aligned rows (checked by the caller); bias already added.  rowptr(global row) -> first element of the row.
constexpr int GX_ST_LD = 36;
constexpr int GX_ST_FLOATS = 16 * GX_ST_LD;  // per wave
template <typename RowPtr>
__device__ __forceinline__ void gx_store_tile16(const f32x16& acc, float* __restrict__ T, const int row0, const int M, const int col0,
                                                const int N, const bool accumulate, const int lane, RowPtr rowptr) {
  const int r = lane & 31, hi = lane >> 5, c4 = 4 * (lane & 7), rr = lane >> 3;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) T[((q8 & 3) + 8 * (q8 >> 2) + 4 * hi) * GX_ST_LD + r] = acc[8 * half + q8];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int lr = rr + 8 * k, grow = row0 + 16 * half + lr;
      f32x4 v = *reinterpret_cast<const f32x4*>(T + lr * GX_ST_LD + c4);
      if (grow < M && col0 + c4 < N) {
        float* const p = rowptr(grow) + (col0 + c4);
        if (accumulate) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(p);
          v[0] += o[0], v[1] += o[1], v[2] += o[2], v[3] += o[3];
        }
        *reinterpret_cast<f32x4*>(p) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ rows kernels (kinds 0, 1)
template <int MODE, int BKIND, bool VEC>
__global__ __launch_bounds__(256) void gemmx_rows_kernel(const GXGroup g_byval) {
  KERNARG_IN_PLACE(GXGroup);
  constexpr int NA = Planes<MODE>::A, NB = Planes<MODE>::W;
  const int pi = flat_problem(g, (int)blockIdx.x);
  const GXP& P = g.p[pi];
  const int local = (int)blockIdx.x - g.woff[pi], tiles_n = (P.N + GX_T - 1) / GX_T;
  const int mt = local / tiles_n;
  const int m0 = mt * GX_T, n0 = (local - mt * tiles_n) * GX_T;
  __shared__ __attribute__((aligned(16))) __bf16 As[2][NA * GX_ROW];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][NB * GX_ROW];
  const int wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;

  LoaderKC<NA> la;
  LoaderKC<NB> lbk;
  LoaderKS<NB> lbs;
  la.init(P.A, m0, P.M);
  la.template issue<VEC>(P.A, 0, P.K);
  if constexpr (BKIND == 1) {
    lbk.init(P.B, n0, P.N);
    lbk.template issue<VEC>(P.B, 0, P.K);
  } else {
    lbs.init(P.B, 0);
    lbs.issue(P.B, P.K, n0, P.N);
  }
  la.commit(As[0]);
  if constexpr (BKIND == 1) lbk.commit(Bs[0]);
  else lbs.commit(Bs[0], false);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < P.K; k0 += GX_BK) {
    const bool more = k0 + GX_BK < P.K;
    if (more) {
      la.template issue<VEC>(P.A, k0 + GX_BK, P.K);
      if constexpr (BKIND == 1) lbk.template issue<VEC>(P.B, k0 + GX_BK, P.K);
      else lbs.issue(P.B, P.K, n0, P.N);
    }
    gx_mma<NA, NB>(As[cur], Bs[cur], wm0, wn0, acc);
    if (more) {
      la.commit(As[cur ^ 1]);
      if constexpr (BKIND == 1) lbk.commit(Bs[cur ^ 1]);
      else lbs.commit(Bs[cur ^ 1], false);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: accumulator register q of lane (r, hi) = tile row (q & 3) + 8 (q >> 2) + 4 hi, column r
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int col = n0 + wn0 + r;
  const float bv = (P.bias && col < P.N) ? P.bias[col] : 0.f;
  const bool flat_c = P.C.d == 1;
  const SmallDiv cdiv(P.C.d);
  float* const cbase = const_cast<float*>(P.C.base);
  const int rb = m0 + wm0 + 4 * hi;
  static_assert(sizeof(As) >= 4 * GX_ST_FLOATS * sizeof(float), "the operand stages double as the store tiles");
  if (((P.C.ld | P.C.inner | P.N) & 3) == 0 && (reinterpret_cast<size_t>(cbase) & 15) == 0) {  // uniform: 16-byte stores
    // (the last barrier of the K loop is behind every wave: the operand stages are free)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += bv;
    gx_store_tile16(acc, reinterpret_cast<float*>(&As[0][0]) + wave * GX_ST_FLOATS, m0 + wm0, P.M, n0 + wn0, P.N, P.accumulate != 0,
                    lane, [&](const int grow) {
                      if (flat_c) return cbase + (long)grow * P.C.ld;
                      const int qd = grow / P.C.d;
                      return cbase + (long)qd * P.C.ld + (long)(grow - qd * P.C.d) * P.C.inner;
                    });
    return;
  }
  if (col >= P.N) return;
  long off0;
  int rem0 = 0;
  if (flat_c) {
    off0 = (long)rb * P.C.ld;
  } else {
    const int qb = rb / P.C.d;
    rem0 = rb - qb * P.C.d;
    off0 = (long)qb * P.C.ld;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int dr = (q & 3) + 8 * (q >> 2);
    if (rb + dr < P.M) {
      long off;
      if (flat_c) {
        off = off0 + (long)dr * P.C.ld;
      } else {
        const int t = rem0 + dr, dq = cdiv.div(t);
        off = off0 + (long)dq * P.C.ld + (long)(t - dq * P.C.d) * P.C.inner;
      }
      float* p = cbase + off + col;
      float v = acc[q] + bv;
      if (P.accumulate) v += *p;
      *p = v;
    }
  }
}

// ------------------------------------------------------------------------------------- short-K rows kernel (kind 1, K <= 64)
// The radial MLPs' last layers: E rows x 64 -> N (960 per module, 7 modules side by side).  With the generic kernel every 64 x 64
// output tile is a workgroup of its own that loads and splits its A rows again (15 times per row tile) and lives for two K
// steps: 41 685 short-lived workgroups, 415 us for 681 MB of output (tools/gemm_shapes.py).  Here a workgroup keeps the planes
// of its 64 A rows in LDS and walks ALL column tiles of the problem: the next tile's weights are in flight (registers) while the
// current tile multiplies and stores.
constexpr int GW_K = 64;
constexpr int GW_LDK = GW_K + 8;
constexpr int GW_ROW = GX_T * GW_LDK;

template <int NP>
struct LoaderWide {  // 64 rows x K <= 64 floats, k contiguous: thread = (row t >> 2, quarter t & 3), 16-byte loads at k = 4 (q + 4 j)
  float4 v[GW_K / 16];
  int kv;  // valid k of this thread's row (0 for a row past X); unconditional loads, zeroing in commit (see LoaderKC)
  __device__ __forceinline__ void issue(const GRows& R, int x0, int X, int K, bool two_level) {
    const int row = x0 + (threadIdx.x >> 2), q = threadIdx.x & 3;
    const bool rv = row < X;
    kv = rv ? K : 0;
    const long off = two_level ? row_off2(rv ? row : x0, R.d, R.ld, R.inner) : (long)(rv ? row : x0) * R.ld;
#pragma unroll
    for (int j = 0; j < GW_K / 16; ++j) {  // K % 4 == 0 here
      const int k = 4 * (q + 4 * j);
      v[j] = *reinterpret_cast<const float4*>(R.base + off + (k < K ? k : 0));
    }
  }
  __device__ __forceinline__ void commit(__bf16* __restrict__ T) const {
    const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
#pragma unroll
    for (int j = 0; j < GW_K / 16; ++j) {
      const bool ok = 4 * (q + 4 * j) < kv;
      const float f[4] = {ok ? v[j].x : 0.f, ok ? v[j].y : 0.f, ok ? v[j].z : 0.f, ok ? v[j].w : 0.f};
      __bf16 p[NP][8];
      split_n<NP>(f, 4, p);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        *reinterpret_cast<bf16x4_t*>(T + pl * GW_ROW + row * GW_LDK + 4 * (q + 4 * j)) = bf16x4_t{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
    }
  }
};

template <int MODE>
__global__ __launch_bounds__(256) void gemmx_rows_wide_kernel(const GXGroup g_byval) {
  KERNARG_IN_PLACE(GXGroup);
  constexpr int NA = Planes<MODE>::A, NB = Planes<MODE>::W;
  const GXP& P = g.p[blockIdx.z];
  const int m0 = blockIdx.x * GX_T;
  if (m0 >= P.M) return;
  __shared__ __attribute__((aligned(16))) __bf16 As[NA * GW_ROW];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[NB * GW_ROW];
  __shared__ __attribute__((aligned(16))) float St[4 * GX_ST_FLOATS];  // output tiles on their way to 16-byte stores
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  LoaderWide<NA> la;
  LoaderWide<NB> lb;
  la.issue(P.A, m0, P.M, P.K, true);
  lb.issue(P.B, 0, P.N, P.K, false);
  la.commit(As);
  // this lane's output rows (fixed for the whole walk): tile row (q & 3) + 8 (q >> 2) + 4 hi
  const bool flat_c = P.C.d == 1;
  const SmallDiv cdiv(P.C.d);
  float* const cbase = const_cast<float*>(P.C.base);
  const int rb = m0 + wm0 + 4 * hi;
  long off0;
  int rem0 = 0;
  if (flat_c) {
    off0 = (long)rb * P.C.ld;
  } else {
    const int qb = rb / P.C.d;
    rem0 = rb - qb * P.C.d;
    off0 = (long)qb * P.C.ld;
  }
  const int nkt = (P.K + 15) / 16;
  const bool vec_c = ((P.C.ld | P.C.inner | P.N) & 3) == 0 && (reinterpret_cast<size_t>(cbase) & 15) == 0;  // uniform
  for (int n0 = 0; n0 < P.N; n0 += GX_T) {
    __syncthreads();  // the previous tile's fragments are read (and, first time round, As is complete after the next barrier)
    lb.commit(Bs);
    __syncthreads();
    if (n0 + GX_T < P.N) lb.issue(P.B, n0 + GX_T, P.N, P.K, false);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int kt = 0; kt < GW_K / 16; ++kt) {
      if (kt < nkt) {  // uniform
        bf16x8 a[NA], b[NB];
#pragma unroll
        for (int q = 0; q < NA; ++q) a[q] = *reinterpret_cast<const bf16x8*>(As + q * GW_ROW + (wm0 + r) * GW_LDK + 16 * kt + 8 * hi);
#pragma unroll
        for (int q = 0; q < NB; ++q) b[q] = *reinterpret_cast<const bf16x8*>(Bs + q * GW_ROW + (wn0 + r) * GW_LDK + 16 * kt + 8 * hi);
        mma_terms<NA, NB>(a, b, acc);
      }
    }
    const int col = n0 + wn0 + r;
    if (vec_c) {
      const float bv = (P.bias && col < P.N) ? P.bias[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] += bv;
      gx_store_tile16(acc, St + wave * GX_ST_FLOATS, m0 + wm0, P.M, n0 + wn0, P.N, P.accumulate != 0, lane, [&](const int grow) {
        if (flat_c) return cbase + (long)grow * P.C.ld;
        const int qd = grow / P.C.d;
        return cbase + (long)qd * P.C.ld + (long)(grow - qd * P.C.d) * P.C.inner;
      });
    } else if (col < P.N) {
      const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int dr = (q & 3) + 8 * (q >> 2);
        if (rb + dr < P.M) {
          long off;
          if (flat_c) {
            off = off0 + (long)dr * P.C.ld;
          } else {
            const int t = rem0 + dr, dq = cdiv.div(t);
            off = off0 + (long)dq * P.C.ld + (long)(t - dq * P.C.d) * P.C.inner;
          }
          float* p = cbase + off + col;
          float v = acc[q] + bv;
          if (P.accumulate) v += *p;
          *p = v;
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------- weight gradients (kinds 2, 3)
template <int MODE>
__global__ __launch_bounds__(256) void gemmx_tn_kernel(const GXGroup g_byval) {
  KERNARG_IN_PLACE(GXGroup);
  constexpr int NA = Planes<MODE>::A;  // both operands are activations
  const int pi = flat_problem(g, (int)blockIdx.x);
  const GXP& P = g.p[pi];
  const int local = (int)blockIdx.x - g.woff[pi], tiles_n = (P.N + GX_T - 1) / GX_T;
  const int tiles = ((P.M + GX_T - 1) / GX_T) * tiles_n;
  const int bz = local / tiles, tile = local - bz * tiles;  // tile fastest: the splits of a problem start together
  const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
  const int m0 = mt * GX_T, n0 = nt * GX_T;
  __shared__ __attribute__((aligned(16))) __bf16 As[2][NA * GX_ROW];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][NA * GX_ROW];
  const int wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const int total = (P.K + GX_BK - 1) / GX_BK;
  const int s_beg = bz * P.steps_per_split;
  const int s_end = min(total, s_beg + P.steps_per_split);
  if (s_beg >= s_end) return;
  const bool csA = P.kind == 3 && P.cs != nullptr && nt == 0;
  const bool csB = P.kind == 2 && P.cs != nullptr && mt == 0;
  LoaderKS<NA> la, lb;
  la.init(P.A, s_beg * GX_BK);
  lb.init(P.B, s_beg * GX_BK);
  la.issue(P.A, P.K, m0, P.M);
  lb.issue(P.B, P.K, n0, P.N);
  la.commit(As[0], csA);
  lb.commit(Bs[0], csB);
  __syncthreads();
  int cur = 0;
  for (int s = s_beg; s < s_end; ++s) {
    const bool more = s + 1 < s_end;
    if (more) {
      la.issue(P.A, P.K, m0, P.M);
      lb.issue(P.B, P.K, n0, P.N);
    }
    gx_mma<NA, NA>(As[cur], Bs[cur], wm0, wn0, acc);
    if (more) {
      la.commit(As[cur ^ 1], csA);
      lb.commit(Bs[cur ^ 1], csB);
    }
    __syncthreads();
    cur ^= 1;
  }
  // column sums: the four k groups of a column add up through atomics (exact fp32 values, not the planes)
  {
    const int x = threadIdx.x & 63;
    if (csA && m0 + x < P.M) atomicAdd(P.cs + m0 + x, la.cs);
    if (csB && n0 + x < P.N) atomicAdd(P.cs + n0 + x, lb.cs);
  }
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int col = n0 + wn0 + r;
  if (col >= P.N) return;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = m0 + wm0 + (q & 3) + 8 * (q >> 2) + 4 * hi;
    if (row < P.M) atomicAdd(P.Cw + (long)row * P.ldc + col, acc[q]);
  }
}

// ------------------------------------------------------------------------------------------------ direct kernels (node rows)
// The node-row linears (2 304 rows x (2l+1)) are not matrix work: ~12 us per launch whatever the matrix rate (fp32 MFMA 12.8 us,
// split planes 12.3 us, tools/gemm_shapes.py).  Experiment of round 4, kept behind eqf_gemmx_dev_set(0, 0): ONE WAVE owns a
// 32 x 32 output tile and issues the loads of a whole 128-deep K chunk (operand fragments straight in MFMA layout, no LDS, no
// barrier) before it multiplies -- one round trip per chunk, ~1 000 independent waves per launch.  Measured: the same 12.7 us
// (with loads behind per-lane bounds branches: 16 us, hipcc waits for them at the end of each branch); the floor of these
// launches is neither the K loop nor the barriers.
constexpr int GD_KC = 128;  // K chunk held in registers (8 fragments of 16)

// Preconditions (checked on the host, else the LDS-tiled kernel runs): K a multiple of 16 and 16-byte-aligned rows, so that every
// load is unconditional (rows / columns past the edge are clamped to a valid one and never stored): a load behind a per-lane
// branch makes hipcc wait for it at the end of the branch, and the "all loads of the chunk in flight" is gone.
template <int MODE, int BKIND>
__global__ __launch_bounds__(64, 2) void gemmx_rows_direct_kernel(const GXGroup g_byval) {
  KERNARG_IN_PLACE(GXGroup);
  constexpr int NA = Planes<MODE>::A, NB = Planes<MODE>::W;
  const GXP& P = g.p[blockIdx.z];
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  if (m0 >= P.M || n0 >= P.N) return;
  const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
  const int row = min(m0 + r, P.M - 1), col = min(n0 + r, P.N - 1);
  const float* const ap = P.A.base + row_off2(row, P.A.d, P.A.ld, P.A.inner) + 8 * hi;
  const float* const bp = BKIND == 1 ? P.B.base + (long)col * P.B.ld + 8 * hi : P.B.base + (long)(8 * hi) * P.B.ld + col;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int kc = 0; kc < P.K; kc += GD_KC) {
    float a[GD_KC / 16][8], b[GD_KC / 16][8];
    const int kend = min(P.K - kc, GD_KC);
#pragma unroll
    for (int kt = 0; kt < GD_KC / 16; ++kt) {
      if (16 * kt < kend) {  // uniform
        const int k = kc + 16 * kt;
        const float4 u0 = *reinterpret_cast<const float4*>(ap + k);
        const float4 u1 = *reinterpret_cast<const float4*>(ap + k + 4);
        a[kt][0] = u0.x, a[kt][1] = u0.y, a[kt][2] = u0.z, a[kt][3] = u0.w;
        a[kt][4] = u1.x, a[kt][5] = u1.y, a[kt][6] = u1.z, a[kt][7] = u1.w;
        if constexpr (BKIND == 1) {  // B [N, K]: k contiguous
          const float4 w0 = *reinterpret_cast<const float4*>(bp + k);
          const float4 w1 = *reinterpret_cast<const float4*>(bp + k + 4);
          b[kt][0] = w0.x, b[kt][1] = w0.y, b[kt][2] = w0.z, b[kt][3] = w0.w;
          b[kt][4] = w1.x, b[kt][5] = w1.y, b[kt][6] = w1.z, b[kt][7] = w1.w;
        } else {  // B [K, N]: lane = column, eight rows down k (each a 128-byte run over the 32 columns)
#pragma unroll
          for (int j = 0; j < 8; ++j) b[kt][j] = bp[(unsigned)(k + j) * (unsigned)P.B.ld];  // (32-bit offsets: one address register pair)
        }
      }
    }
#pragma unroll
    for (int kt = 0; kt < GD_KC / 16; ++kt) {
      if (16 * kt < kend) {  // uniform
        bf16x8 pa[NA], pb[NB];
        split_planes<NA>(a[kt], pa);
        split_planes<NB>(b[kt], pb);
        mma_terms<NA, NB>(pa, pb, acc);
      }
    }
  }
  if (n0 + r >= P.N) return;
  const int ccol = n0 + r;
  const float bv = P.bias ? P.bias[ccol] : 0.f;
  const bool flat_c = P.C.d == 1;
  const SmallDiv cdiv(P.C.d);
  float* const cbase = const_cast<float*>(P.C.base);
  const int rb = m0 + 4 * hi;
  long off0;
  int rem0 = 0;
  if (flat_c) {
    off0 = (long)rb * P.C.ld;
  } else {
    const int qb = rb / P.C.d;
    rem0 = rb - qb * P.C.d;
    off0 = (long)qb * P.C.ld;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int dr = (q & 3) + 8 * (q >> 2);
    if (rb + dr < P.M) {
      long off;
      if (flat_c) {
        off = off0 + (long)dr * P.C.ld;
      } else {
        const int t = rem0 + dr, dq = cdiv.div(t);
        off = off0 + (long)dq * P.C.ld + (long)(t - dq * P.C.d) * P.C.inner;
      }
      float* p = cbase + off + ccol;
      float v = acc[q] + bv;
      if (P.accumulate) v += *p;
      *p = v;
    }
  }
}

// weight gradients: one wave per (32 x 32 tile of C, chunk of GD_KC reduction rows); both operands "lane = column, eight rows
// down the reduction index"; column sums (bias gradients) from the fp32 registers; fp32 atomics into C.  Loads unconditional
// (clamped row / column), rows past the end zeroed by a select.
template <int MODE>
__global__ __launch_bounds__(64, 2) void gemmx_tn_direct_kernel(const GXGroup g_byval) {
  KERNARG_IN_PLACE(GXGroup);
  constexpr int NA = Planes<MODE>::A;
  int pi = 0;
  while (pi + 1 < g.n && (int)blockIdx.z >= g.zoff[pi + 1]) ++pi;
  const GXP& P = g.p[pi];
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  if (m0 >= P.M || n0 >= P.N) return;
  const int i0 = ((int)blockIdx.z - g.zoff[pi]) * GD_KC;
  if (i0 >= P.K) return;
  const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
  const int mcol = min(m0 + r, P.M - 1), ncol = min(n0 + r, P.N - 1);
  const int kend = min(P.K - i0, GD_KC);
  float a[GD_KC / 16][8], b[GD_KC / 16][8];
  // two-level reduction rows i = q d + rem: (q, rem) of this lane's first row, then stepped (the two operands share d only by
  // convention, so each keeps its own)
  const int ifirst = i0 + 8 * hi;
  int qa = ifirst / P.A.d, ra = ifirst - qa * P.A.d;
  int qb = ifirst / P.B.d, rb = ifirst - qb * P.B.d;
#pragma unroll
  for (int kt = 0; kt < GD_KC / 16; ++kt) {
    if (16 * kt < kend) {  // uniform
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = ifirst + 16 * kt + j;
        const bool iv = i < P.K;
        const long oa = iv ? (long)qa * P.A.ld + (long)ra * P.A.inner : 0;
        const long ob = iv ? (long)qb * P.B.ld + (long)rb * P.B.inner : 0;
        const float va = P.A.base[oa + mcol], vb = P.B.base[ob + ncol];
        a[kt][j] = iv ? va : 0.f;
        b[kt][j] = iv ? vb : 0.f;
        ++ra, ++rb;
        if (ra == P.A.d) ra = 0, ++qa;
        if (rb == P.B.d) rb = 0, ++qb;
      }
      // the other half-wave's eight rows lie between this step's and the next step's
      {
        const int sa = ra + 8, sb = rb + 8;
        const int da = sa / P.A.d, db = sb / P.B.d;
        qa += da, ra = sa - da * P.A.d;
        qb += db, rb = sb - db * P.B.d;
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float csa = 0.f, csb = 0.f, cca = 0.f, ccb = 0.f;  // (Kahan-compensated column sums, as in LoaderKS)
#pragma unroll
  for (int kt = 0; kt < GD_KC / 16; ++kt) {
    if (16 * kt < kend) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ya = a[kt][j] - cca, ta = csa + ya;
        cca = (ta - csa) - ya, csa = ta;
        const float yb = b[kt][j] - ccb, tb = csb + yb;
        ccb = (tb - csb) - yb, csb = tb;
      }
      bf16x8 pa[NA], pb[NA];
      split_planes<NA>(a[kt], pa);
      split_planes<NA>(b[kt], pb);
      mma_terms<NA, NA>(pa, pb, acc);
    }
  }
  const bool mv = m0 + r < P.M, nv = n0 + r < P.N;
  if (P.cs != nullptr) {
    if (P.kind == 3 && blockIdx.y == 0) {
      csa += __shfl_xor(csa, 32);
      if (hi == 0 && mv) atomicAdd(P.cs + mcol, csa);
    }
    if (P.kind == 2 && blockIdx.x == 0) {
      csb += __shfl_xor(csb, 32);
      if (hi == 0 && nv) atomicAdd(P.cs + ncol, csb);
    }
  }
  if (!nv) return;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = m0 + (q & 3) + 8 * (q >> 2) + 4 * hi;
    if (row < P.M) atomicAdd(P.Cw + (long)row * P.ldc + ncol, acc[q]);
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool rows_vec_ok(const float* base, const eqf_rows& r) { return aligned16(base) && (r.ld % 4 == 0) && (r.inner % 4 == 0); }

}  // namespace

// development switch (eqf_gemmx_dev_set key 0): 0 = the one-wave-per-tile kernels for node-row problems, anything else = the
// LDS-tiled kernels for every problem (the default: the two measure the same, tools/gemm_shapes.py, profiles/r04/r04_j_*)
static int g_gemmx_no_direct = 1;
static int g_gemmx_no_wide = 0;
static int g_gemmx_tn_minsteps = 8;   // key 2: K steps per workgroup of a weight gradient at least (sweep: profiles/r04/r04_y_*)
// key 1: 1 = the generic tiled kernel for the short-K, many-row problems too (A/B)

extern "C" {

int eqf_gemmx_dev_set(int key, int value) {
  if (key == 0) {
    g_gemmx_no_direct = value;
    return 0;
  }
  if (key == 1) {
    g_gemmx_no_wide = value;
    return 0;
  }
  if (key == 2 && value >= 1) {
    g_gemmx_tn_minsteps = value;
    return 0;
  }
  return EQF_E_BADARG;
}

int eqf_gemmx_group(const eqf_gemm_desc* d, int n, int mode, void* stream) {
  if (!d || n < 1 || n > GX_MAXP || mode < 0 || mode > 2) return EQF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    if (d[i].kind < 0 || d[i].kind > 3) return EQF_E_BADARG;
    if (!d[i].A || !d[i].B || !d[i].C || d[i].ra.d < 1 || d[i].rc.d < 1) return EQF_E_BADARG;
  }
  static thread_local GXGroup G;
  for (int kind = 0; kind < 2; ++kind) {
    memset(&G, 0, sizeof G);
    int maxm = 0, maxn = 0, big = 0, direct_ok = 1, wide_ok = 1, vec_ok = 1, wg = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
      if (d[i].kind != kind || d[i].M <= 0 || d[i].N <= 0) continue;
      GXP& P = G.p[G.n];
      G.woff[G.n++] = wg;
      wg += eqf_cdiv(d[i].M, GX_T) * eqf_cdiv(d[i].N, GX_T);
      P.A = {d[i].A, d[i].ra.d, d[i].ra.ld, d[i].ra.inner};
      P.B = {d[i].B, 1, d[i].ldb, 0};
      P.C = {d[i].C, d[i].rc.d, d[i].rc.ld, d[i].rc.inner};
      P.bias = d[i].bias;
      P.M = d[i].M, P.N = d[i].N, P.K = d[i].K, P.accumulate = d[i].accumulate, P.kind = kind;
      P.vecA = rows_vec_ok(d[i].A, d[i].ra);
      P.vecB = aligned16(d[i].B) && d[i].ldb % 4 == 0;
      if (P.K % 16 != 0 || !P.vecA || (kind == 1 && !P.vecB)) direct_ok = 0;
      if (P.K % 4 != 0 || !P.vecA || (kind == 1 && !P.vecB)) vec_ok = 0;  // one unaligned problem: scalar loads for the group
      if (!(kind == 1 && P.K <= GW_K && P.K % 4 == 0 && P.vecA && P.vecB && P.M >= 8192 && P.N >= 2 * GX_T)) wide_ok = 0;
      if (eqf_cdiv(P.M, GX_T) > maxm) maxm = eqf_cdiv(P.M, GX_T);
      if (eqf_cdiv(P.N, GX_T) > maxn) maxn = eqf_cdiv(P.N, GX_T);
      if (P.M >= 32768 / 2 + 1) big = 1;  // more than 16 k rows: edge rows
      flops += 2.0 * P.M * (double)P.N * P.K;
      bytes += 4.0 * ((double)P.M * P.K + (double)P.K * P.N + (double)P.M * P.N);
    }
    if (G.n == 0) continue;
    // few rows (node-level linears): one wave per 32 x 32 tile, whole K chunks in flight; many rows (edge-level: the radial
    // MLPs): the LDS-tiled kernel.  Timed under different names.
    const bool direct = !big && direct_ok && !g_gemmx_no_direct;
    const bool wide = kind == 1 && wide_ok && !g_gemmx_no_wide;
    for (int i = G.n; i <= GX_MAXP; ++i) G.woff[i] = i == G.n ? wg : INT_MAX;
    const dim3 grid = wide ? dim3(maxm, 1, G.n) : direct ? dim3(2 * maxm, 2 * maxn, G.n) : dim3(wg, 1, 1);
    const int pid = eqf_prof_begin(kind == 0 ? (big ? "gemmx_group_kn_edge" : "gemmx_group_kn_node")
                                             : (big ? "gemmx_group_nk_edge" : "gemmx_group_nk_node"), st, flops, bytes);
#define GX_ROWS(M_)                                                                                              \
  do {                                                                                                           \
    if (wide) {                                                                                                  \
      hipLaunchKernelGGL((gemmx_rows_wide_kernel<M_>), grid, dim3(256), 0, st, G);                              \
    } else if (direct) {                                                                                         \
      if (kind == 0) hipLaunchKernelGGL((gemmx_rows_direct_kernel<M_, 0>), grid, dim3(64), 0, st, G);           \
      else hipLaunchKernelGGL((gemmx_rows_direct_kernel<M_, 1>), grid, dim3(64), 0, st, G);                     \
    } else if (kind == 0) {                                                                                      \
      if (vec_ok) hipLaunchKernelGGL((gemmx_rows_kernel<M_, 0, true>), grid, dim3(256), 0, st, G);              \
      else hipLaunchKernelGGL((gemmx_rows_kernel<M_, 0, false>), grid, dim3(256), 0, st, G);                    \
    } else {                                                                                                     \
      if (vec_ok) hipLaunchKernelGGL((gemmx_rows_kernel<M_, 1, true>), grid, dim3(256), 0, st, G);              \
      else hipLaunchKernelGGL((gemmx_rows_kernel<M_, 1, false>), grid, dim3(256), 0, st, G);                    \
    }                                                                                                            \
  } while (0)
    if (mode == 0) GX_ROWS(0);
    else if (mode == 1) GX_ROWS(1);
    else GX_ROWS(2);
#undef GX_ROWS
    eqf_prof_end(pid, st);
    EQF_CHECK_LAUNCH();
  }
  {
    memset(&G, 0, sizeof G);
    int maxm = 0, maxn = 0, z = 0, big = 0, zd = 0, wg = 0;
    int zoff_d[GX_MAXP + 1];
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
      if (d[i].kind != 2 && d[i].kind != 3) continue;
      if (d[i].M <= 0 || d[i].N <= 0 || d[i].K <= 0) continue;
      GXP& P = G.p[G.n];
      P.A = {d[i].A, d[i].ra.d, d[i].ra.ld, d[i].ra.inner};
      P.B = {d[i].B, d[i].rc.d, d[i].rc.ld, d[i].rc.inner};
      P.Cw = d[i].C, P.ldc = d[i].ldb, P.M = d[i].M, P.N = d[i].N, P.K = d[i].K, P.kind = d[i].kind;
      P.cs = const_cast<float*>(d[i].bias);
      const int tiles = eqf_cdiv(P.M, GX_T) * eqf_cdiv(P.N, GX_T);
      const int total_steps = eqf_cdiv(P.K, GX_BK);
      int ksplit = 1024 / (tiles > 0 ? tiles : 1);
      // at least g_gemmx_tn_minsteps K steps per workgroup.  Longer walks are NOT cheaper: the time of these launches follows
      // the steps per workgroup (8 / 16 / 24 / 48 / 96 steps: 25 / 40 / 57 / 101 / 192 us for the 480 x 480 node-row gradient),
      // the 64 x 64 atomics per split do not show
      const int max_split = eqf_cdiv(total_steps, g_gemmx_tn_minsteps);
      if (ksplit > max_split) ksplit = max_split;
      if (ksplit < 1) ksplit = 1;
      P.steps_per_split = eqf_cdiv(total_steps, ksplit);
      ksplit = eqf_cdiv(total_steps, P.steps_per_split);
      G.zoff[G.n] = z;
      z += ksplit;
      G.woff[G.n] = wg;
      wg += tiles * ksplit;
      zoff_d[G.n] = zd;
      zd += eqf_cdiv(P.K, GD_KC);
      G.n++;
      if (eqf_cdiv(P.M, GX_T) > maxm) maxm = eqf_cdiv(P.M, GX_T);
      if (eqf_cdiv(P.N, GX_T) > maxn) maxn = eqf_cdiv(P.N, GX_T);
      if (P.K >= 32768 / 2 + 1) big = 1;
      flops += 2.0 * P.M * (double)P.N * P.K;
      bytes += 4.0 * ((double)P.K * P.M + (double)P.K * P.N + (double)P.M * P.N);
    }
    if (G.n > 0) {
      G.zoff[G.n] = z;
      const bool direct = !big && !g_gemmx_no_direct;
      if (direct) {
        zoff_d[G.n] = zd;
        for (int i = 0; i <= G.n; ++i) G.zoff[i] = zoff_d[i];
      }
      for (int i = G.n; i <= GX_MAXP; ++i) G.woff[i] = i == G.n ? wg : INT_MAX;
      const dim3 grid = direct ? dim3(2 * maxm, 2 * maxn, zd) : dim3(wg, 1, 1);
      const int pid = eqf_prof_begin(big ? "gemmx_group_tn_edge" : "gemmx_group_tn_node", st, flops, bytes);
      if (direct) {
        if (mode == 0) hipLaunchKernelGGL((gemmx_tn_direct_kernel<0>), grid, dim3(64), 0, st, G);
        else if (mode == 1) hipLaunchKernelGGL((gemmx_tn_direct_kernel<1>), grid, dim3(64), 0, st, G);
        else hipLaunchKernelGGL((gemmx_tn_direct_kernel<2>), grid, dim3(64), 0, st, G);
      } else if (mode == 0) hipLaunchKernelGGL((gemmx_tn_kernel<0>), grid, dim3(256), 0, st, G);
      else if (mode == 1) hipLaunchKernelGGL((gemmx_tn_kernel<1>), grid, dim3(256), 0, st, G);
      else hipLaunchKernelGGL((gemmx_tn_kernel<2>), grid, dim3(256), 0, st, G);
      eqf_prof_end(pid, st);
      EQF_CHECK_LAUNCH();
    }
  }
  return 0;
}

}  // extern "C"
