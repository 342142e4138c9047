// Fused SeparableFCTP kernels: depth-wise tensor product (DTP) + per-degree linear(s), forward and backward.
//
//   mid[e,(p,u),m3] = w[e,p,u] * sum_i M_p[e][i,m3] * x[e,l1(p),i,u]        (DTP; M = per-edge coupling matrices)
//   out[e,l3,m3,n]  = sum_{(p,u) -> l3} mid[e,(p,u),m3] * W_l3[(p,u),n]       (LinearRS on the DTP output)
//
// [ref: SeparableFCTP.forward, nets/graph_attention_transformer.py:234-248; sep_alpha :492; EdgeDegreeEmbedding :725-733]
//
// `mid` (3136 floats per edge for L_max = 2, 9408 for L_max = 3) never exists in HBM, in either direction:
//   forward      : one launch, workgroup = (output degree, tile of 64 edges).  Rows of the GEMM are (m3, edge) in
//                  m3-major order, so tiles are exact multiples of 32 rows for every degree; the A operand is generated
//                  slab by slab (32 channels) from x, w and the LDS-resident coupling tile; all consumers of the DTP
//                  output of degree 0 (value linear + attention-logit linear) share one concatenated weight, i.e. the
//                  DTP is generated once.  Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//   data grad    : workgroup = (tile of 32 edges, 32-channel chunk of one input degree l1).  For every path that
//                  reads the chunk: d_mid tile = d_out tile (staged transposed in LDS) x W_slab^T on the matrix cores
//                  (the 4 waves split the reduction, partial tiles are summed with LDS atomics), then the DTP backward
//                  contraction consumes it from LDS: dw written once, dx accumulated in registers over the paths and
//                  written once -- no atomics on dx, no d_mid in HBM.
//   weight grad  : workgroup = (32-channel slab of one path, chunk of edges); every wave owns a private edge range,
//                  regenerates mid for two edges per step in MFMA-operand layout (lane = channel) and streams the
//                  d_out rows straight from memory as the B operand (coalesced); no LDS, no barriers; fp32 atomics
//                  of the per-wave accumulators at the end.
#include "common.h"
#include "prof.h"
#include "sfc_common.h"
#include <cstdio>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) float sfc_lds[];  // dynamic LDS of every kernel in this file

// Development switches (phases of a kernel switched off, per-phase cycle counters) cost scalar instructions inside the hot
// loops: they are compiled in only with -DEQF_DEV_SWITCHES=1 (EQF_EXTRA_FLAGS="-DEQF_DEV_SWITCHES=1" python -m
// equiformer_amd.build); in the product build the eqf_*_debug_exp bits that act inside kernels are no-ops.
#ifndef EQF_DEV_SWITCHES
#define EQF_DEV_SWITCHES 0
#endif
#if EQF_DEV_SWITCHES
#define SFC_OFF(g, bit) ((g).exp & (bit))
#define SFC_DBG(g) ((g).dbg)
#else
#define SFC_OFF(g, bit) false
#define SFC_DBG(g) ((unsigned long long*)nullptr)
#endif

namespace {

unsigned long long* g_sfc_dbg = nullptr;  // set by eqf_sfc_debug_buffer (development aid)
// workgroup ordering per kernel {fwd, bwd_data, bwd_weight}, see SfcOrder; measured (MI355X, E = 25 354): order 1 is
// 8-12 % faster for bwd_weight, 0-6 % for bwd_data, and 0-14 % SLOWER for fwd (its workgroups of one degree share the
// staged weight slabs, which the degree-major order keeps hot).  eqf_sfc_debug_order overrides all three for A/B runs.
int g_sfc_order[3] = {0, 1, 1};
// forward matrix step: exact-fp32 MFMA (false, the default) or split-precision bf16 x 6 on the matrix cores (true; f_mma6).
// History: the split-precision step shipped in round 1, was found to return run-to-run different results at the bench size
// (E = 25 k edges: two rows of an edge tile wrong by ~1e-2 in a few launches out of a hundred) and was switched off.
// Root cause (round 2, profiles/r02/x6_investigation/): not the matrix step itself -- the PACKED-FP32 VALU instructions
// (v_pk_fma_f32 / v_pk_mul_f32) hipcc emitted for the generation of the A tile return wrong values in lanes 48-63 now
// and then while the co-resident workgroup's wave on the same SIMD runs the bf16 MFMAs.  The kernel variant that issues
// bf16 MFMAs therefore contains no packed-FP32 instruction any more (x6_fmac / x6_mul / x6_sub): 0 wrong results in
// 2 400 launches, full-size parity tests green with it.  Scalar instead of packed VALU work costs most of what the matrix
// cores gained (interleaved A/B, tools/sfc_fwd_ab.py: sep_act 211.4 us vs 219.8 us with the fp32 step, sep_value 158.6 vs
// 166.2; it was 203 vs 228 with the packed instructions): 0.6 % of a train step.  Not enough to change the default at
// the end of a round whose first job was determinism: the exact-fp32 step stays the default, eqf_sfc_debug_exp(64)
// selects the split-precision one (tests/test_gpu_fullsize.py keeps it bit-reproducible).
bool g_sfc_x6_default = false;
int g_sfc_exp = 0;  // development aid (eqf_sfc_debug_exp): bit mask that switches phases of the kernels OFF to time the rest

using namespace sfc;

__host__ inline SfcOrder make_order(int kernel, int nx, int ny, int& nblocks) {
  SfcOrder o;
  o.mode = g_sfc_order[kernel], o.nx = nx, o.ny = ny;
  o.per_xcd = (nx * ny + 7) / 8;
  nblocks = (o.mode == 1) ? 8 * o.per_xcd : nx * ny;
  return o;
}

// ------------------------------------------------------------------------------------------------ forward
// These kernels are instruction-issue bound unless every small loop is unrolled with compile-time trip counts (the
// first, runtime-indexed version spent 23 VALU + 10 SALU instructions per MFMA): the bodies are therefore templated
// on the degrees (D1 = 2*l1+1 of the slab, D3 = 2*l3+1 of the output) and dispatched with wave-uniform switches, so
// that all LDS offsets are immediates and no per-MFMA branches remain.
constexpr int F_TE = 64;     // edges per tile
constexpr int F_NP = 8;      // edges per generating thread
constexpr int F_MAXT = 4;    // 32x32 accumulator tiles per wave (upper bound; FT<MAXD> below)
constexpr int F_MAXCT = 6;   // column tiles per workgroup
// column tiles a workgroup of output degree d3 can take with `ft` accumulator tiles per wave, and the row stride of
// its staged weight tile
__host__ __device__ constexpr int f_ctcap(int d3, int ft) {
  return (4 * ft) / (F_TE * d3 / 32) < F_MAXCT ? (4 * ft) / (F_TE * d3 / 32) : F_MAXCT;
}
__host__ __device__ constexpr int f_sb(int d3, int ft) { return f_ctcap(d3, ft) * 32 + 4; }

template <int N>
struct IC {
  static constexpr int value = N;
};

struct SfcFwdArgs {
  SfcCommon c;
  const float* bias;   // [N1 of degree 0] or null
  const float* bias2;  // [N2] or null
  unsigned long long* dbg;  // optional phase timers (development aid), may be null
  int grp_floats;           // LDS floats per 256-thread group (paired workgroups)
  int exp;                  // development aid: 1 no MFMA, 2 no generation, 4 no loads after the first slab, 8 no weight loads
  int nsplit[SFC_MAX_DEG], cps[SFC_MAX_DEG];
  SfcOrder ord;                                // nx = edge tiles, ny = (degree, column split) pairs
  signed char y_deg[16], y_split[16];
};

template <int D3, int NT, int FT>
__device__ __forceinline__ void f_mma(const int (&aidx)[FT], const int (&bidx)[FT], f32x16 (&acc)[FT]) {
  constexpr int SA = F_TE * D3 + 1, F_SB = f_sb(D3, FT);
  int ai[NT], bi[NT];
  float an[NT], bn[NT];  // operands of the next k pair, requested before the MFMAs of the current pair are issued
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    ai[i] = aidx[i], bi[i] = bidx[i];
    an[i] = sfc_lds[ai[i]], bn[i] = sfc_lds[bi[i]];
  }
#pragma unroll 1
  for (int kq = 0; kq < 4; ++kq) {  // 4 x (4 k-pairs): bounded unrolling keeps the code and the registers in check
#pragma unroll
    for (int kk = 0; kk < 8; kk += 2) {
      float av[NT], bw[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) av[i] = an[i], bw[i] = bn[i];
      const int nk = (kk + 2 < 8) ? kk + 2 : 6;  // last pair of the group: re-read (the next group reloads)
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        an[i] = sfc_lds[ai[i] + nk * SA];
        bn[i] = sfc_lds[bi[i] + nk * F_SB];
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bw[i], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      ai[i] += 8 * SA, bi[i] += 8 * F_SB;
      if (kq < 3) an[i] = sfc_lds[ai[i]], bn[i] = sfc_lds[bi[i]];
    }
  }
}

// ---- split-precision matrix step (X6): fp32 operands, bf16 matrix cores ------------------------------------------
// The fp32 MFMA runs on the VALU's FMA lanes, so it cannot overlap with the generation / addressing
// work of the same kernel.  Here every fp32 operand value is split exactly into three bf16 terms x = x1 + x2 + x3
// (v_cvt_pk_bf16_f32, two subtractions) and a . b is evaluated as the six products a1b1, a1b2, a2b1, a1b3, a3b1, a2b2
// on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation): error 3-4e-7 of the result scale, the same
// as the fp32 GEMM (tools/bf16_split_error.py), at 12 x 32 cycles per 32 x 32 x 32 tile instead of 16 x 64 -- and on a
// pipe of its own.  The A tile is then kept row-major in LDS ([row][k], stride X6_SA floats) so that a lane's eight
// consecutive k values are two 16-byte reads.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int X6_SA = 36;  // floats per A row: 32 k + 4 pad (16-byte aligned rows, staggered banks)

// Plain (non-packed) VALU instruction, whatever the optimiser would like to do with neighbouring lanes of a vector:
// packed-FP32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) must not run in a kernel that also issues the
// bf16 MFMAs -- see X6_NOTE below.
__device__ __forceinline__ float x6_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void x6_fmac(float& acc, float a, float b) { asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
__device__ __forceinline__ float x6_mul(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    const float r1 = x6_sub(v[j], (float)h);
    const __bf16 m = (__bf16)r1;
    const float r2 = x6_sub(r1, (float)m);
    p1[j] = h, p2[j] = m, p3[j] = (__bf16)r2;
  }
}

template <int D3, int NT, int FT>
__device__ __forceinline__ void f_mma6(const int (&arow)[FT], const int (&bcol)[FT], f32x16 (&acc)[FT]) {
  constexpr int F_SB = f_sb(D3, FT);
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {  // two groups of 16 k per 32-channel slab
    bf16x8 x1, x2, x3, y1, y2, y3;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      {
        float av[8];
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(&sfc_lds[arow[i] + 16 * kg]);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(&sfc_lds[arow[i] + 16 * kg + 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = a0[j], av[4 + j] = a1[j];
        split3(av, x1, x2, x3);
      }
      {
        float bw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bw[j] = sfc_lds[bcol[i] + (16 * kg + j) * F_SB];
        split3(bw, y1, y2, y3);
      }
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, y2, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, y3, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, y1, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, y2, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, y1, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, y1, acc[i], 0, 0, 0);
    }
  }
}

// PAIR: the workgroup has 512 threads = two independent 256-thread groups working on neighbouring edge tiles with
// their own LDS partitions, run in ANTI-PHASE through the shared barriers: while one group's waves issue the MFMAs of
// slab s, the other group's waves (the co-resident wave of every SIMD) wait for loads, generate the next A tile on the
// VALU and write LDS.  Two free-running 256-thread workgroups per CU do the same work with the same resources, but
// measured additively (MFMA 120 us + loads 45 + generation 30 + fixed 66 of 246 us: tools/sfc_exp.py): nothing forces
// their matrix-pipe and memory phases apart.
template <int D3, int MAXD, bool PAIR, bool X6>
__device__ __forceinline__ void f_block(const SfcFwdArgs& g, const int di, const int b) {
  constexpr int ROWS = F_TE * D3, RT = ROWS / 32, SA = ROWS + 1;
  constexpr int A_FLOATS = X6 ? ROWS * X6_SA : 32 * SA;  // A tile: [row][k] (X6) or [k][row]
  constexpr int FT = (MAXD <= 5) ? 3 : F_MAXT;  // accumulator tiles per wave (host: `ft`)
  constexpr int CTCAP = f_ctcap(D3, FT), F_SB = f_sb(D3, FT);
  const SfcDeg& D = g.c.deg[di];
  const int nsplit = g.nsplit[di];
  const int grp = PAIR ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  const int xt = b / nsplit, ns = b - xt * nsplit;
  const int tile = PAIR ? 2 * xt + grp : xt;
  int e0 = tile * F_TE;
  int ecnt = min(F_TE, g.c.E - e0);
  if (ecnt <= 0) ecnt = 0, e0 = 0;  // odd tile count: the idle group runs along (barriers) on row 0 and stores nothing
  const int ncol0 = ns * g.cps[di];
  const int ncols = min(g.cps[di], D.Ncat - ncol0);
  const int CT = ncols >> 5;
  // LDS partition of this group (float offsets into sfc_lds)
  const int AS0 = grp * g.grp_floats, BS0 = AS0 + A_FLOATS, MT0 = BS0 + 32 * F_SB;
  const int m_len = D.m_len;

  const int t = PAIR ? ((int)threadIdx.x & 255) : (int)threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, r = lane & 31, hi = lane >> 5;

  // tiles of this wave: tt = wave + 4 i  ->  (rt = tt % RT, ct = tt / RT); all of it wave-uniform.  Every wave runs
  // NT = ceil(ntile / 4) tiles per step (the barrier makes the slowest wave the pace anyway); surplus slots recompute
  // tile `wave` into an accumulator that is never stored.
  const int ntile = RT * CT;
  const int NT = (ntile + 3) >> 2;
  const int ntw = (ntile - wave + 3) >> 2;
  int aoff[FT], boff[FT], aidx[FT], bidx[FT];
  f32x16 acc[FT];
#pragma unroll
  for (int i = 0; i < FT; ++i) {
    const int tt = (i < ntw) ? wave + 4 * i : wave;
    const int ct = tt / RT, rt = tt - ct * RT;
    aoff[i] = rt * 32;
    boff[i] = ct * 32;
    if (X6) {  // lane = (row r of the tile, k half hi): eight consecutive k of its row / its column
      aidx[i] = AS0 + (aoff[i] + r) * X6_SA + 8 * hi;
      bidx[i] = BS0 + (8 * hi) * F_SB + boff[i] + r;
    } else {
      aidx[i] = AS0 + hi * SA + r + aoff[i];
      bidx[i] = BS0 + hi * F_SB + r + boff[i];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  }

  // Generation mapping: thread = (channel quad c4 = t & 7 -> channels 4 c4 .. 4 c4 + 3 of the slab, edges eg and eg + 32
  // with eg = t >> 3).  x and w are fetched as 16-byte vectors (8 lanes x 16 B = one 128-byte channel run per edge):
  // 4x fewer, 4x wider loads than a thread-per-channel mapping, and every coupling entry read from LDS feeds 4 FMAs.
  const int c4 = t & 7, eg = t >> 3;
  const unsigned x_ld = g.c.x_ld, w_ld = g.c.w_ld;
  unsigned erow[2];  // clamped global edge index (out-of-range edges read a valid row and are zeroed by the mask)
  float emask[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int el = eg + 32 * q;
    erow[q] = e0 + (el < ecnt ? el : 0);
    emask[q] = (el < ecnt) ? 1.0f : 0.f;
  }

  f32x4 xv[2][MAXD], wv[2];
  f32x4 bv[CTCAP];
  int s_d1 = 0, s_mo = 0;  // of the slab whose inputs are in xv / wv / bv
  auto load_x = [&](auto tag, const SfcSlab& S) __attribute__((always_inline)) {
    constexpr int D1 = decltype(tag)::value;
    const float* xs = g.c.x + S.x_off + 4 * c4;
#pragma unroll
    for (int i = 0; i < D1; ++i) {
      const float* xi = xs + i * S.x_mul;
#pragma unroll
      for (int q = 0; q < 2; ++q) xv[q][i] = *reinterpret_cast<const f32x4*>(xi + erow[q] * x_ld);
    }
  };
  auto issue = [&](int s) __attribute__((always_inline)) {
    const SfcSlab S = g.c.slab[D.slab0 + s];
    s_d1 = S.d1;
    s_mo = S.m_off - D.m_base;
    if (g.c.w) {
      const float* ws = g.c.w + S.w_off + 4 * c4;
#pragma unroll
      for (int q = 0; q < 2; ++q) wv[q] = *reinterpret_cast<const f32x4*>(ws + erow[q] * w_ld);
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) wv[q] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    switch (S.d1) {
      case 1: load_x(IC<1>(), S); break;
      case 3: load_x(IC<3>(), S); break;
      case 5: load_x(IC<(MAXD >= 5 ? 5 : 1)>(), S); break;
      default: load_x(IC<(MAXD >= 7 ? 7 : 1)>(), S); break;
    }
    // column blocks beyond CT re-read the last valid block (their LDS columns are never used): no guards, no
    // dynamic indexing of bv
    const int wk = s * 32 + (t >> 3);  // row of the weight matrices
    if (!SFC_OFF(g, 8))
#pragma unroll
    for (int j = 0; j < CTCAP; ++j) {
      const int c = ncol0 + 32 * (j < CT ? j : CT - 1);  // 32-column block: entirely main or entirely second consumer
      const float* wp = (c < D.N1) ? D.W + (long)wk * D.N1 + c : D.W2 + (long)wk * D.N2 + (c - D.N1);
      bv[j] = *reinterpret_cast<const f32x4*>(wp + 4 * (t & 7));
    }
  };
  const int awb = AS0 + (4 * c4) * SA + eg;
  const int mrow = MT0 + eg * m_len;
  auto gen = [&](auto tag) __attribute__((always_inline)) {
    constexpr int D1 = decltype(tag)::value;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int mp = mrow + (32 * q) * m_len + s_mo;
      f32x4 wm;
      if (X6) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wm[c] = x6_mul(wv[q][c], emask[q]);
      } else {
        wm = wv[q] * emask[q];
      }
#pragma unroll
      for (int m3 = 0; m3 < D3; ++m3) {
        if (X6) {
          // X6_NOTE: scalar VALU instructions only.  With the vector expression below hipcc emits v_pk_fma_f32 /
          // v_pk_mul_f32, and on MI355X a packed-FP32 instruction of this wave returns a wrong result in lanes 48-63
          // every now and then while the OTHER wave of its SIMD (the co-resident workgroup) runs the bf16 MFMA step:
          // measured with the same registers fed twice to the same expression (tools/sfc_race.py, profiles/r02/
          // x6_investigation/: 8 differing evaluations per wrong output row, none with one workgroup per CU, none with
          // the fp32 MFMA, none with these scalar instructions; LDS contents and prefetched registers verified intact).
          float ac[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < D1; ++i) {
            const float m = sfc_lds[mp + i * D3 + m3];
#pragma unroll
            for (int c = 0; c < 4; ++c) x6_fmac(ac[c], xv[q][i][c], m);
          }
          f32x4 a;
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = x6_mul(ac[c], wm[c]);
          *reinterpret_cast<f32x4*>(&sfc_lds[AS0 + (m3 * F_TE + eg + 32 * q) * X6_SA + 4 * c4]) = a;
        } else {
          f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < D1; ++i) a += xv[q][i] * sfc_lds[mp + i * D3 + m3];
          a *= wm;
#pragma unroll
          for (int c = 0; c < 4; ++c) sfc_lds[awb + c * SA + 32 * q + m3 * F_TE] = a[c];
        }
      }
    }
  };
  const int bwp = BS0 + (t >> 3) * F_SB + 4 * (t & 7);
  auto commit = [&]() __attribute__((always_inline)) {
    if (!SFC_OFF(g, 2)) switch (s_d1) {
      case 1: gen(IC<1>()); break;
      case 3: gen(IC<3>()); break;
      case 5: gen(IC<(MAXD >= 5 ? 5 : 1)>()); break;
      default: gen(IC<(MAXD >= 7 ? 7 : 1)>()); break;
    }
#pragma unroll
    for (int j = 0; j < CTCAP; ++j) *reinterpret_cast<f32x4*>(&sfc_lds[bwp + 32 * j]) = bv[j];
  };

  unsigned long long t_mark = SFC_DBG(g) ? __builtin_amdgcn_s_memtime() : 0;
  auto tick = [&](int slot) __attribute__((always_inline)) {
    if (SFC_DBG(g)) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if (t == 0) atomicAdd(SFC_DBG(g) + slot, now - t_mark);
      t_mark = now;
    }
  };
  issue(0);
  // coupling tile: Mt[el][j] = coupling[e0+el, m_base + j]; rows of edges beyond the graph are zero.  Loads are
  // unconditional (clamped addresses) and batched: a one-element-per-iteration loop compiles to load / s_waitcnt
  // vmcnt(0) / ds_write per element, i.e. a chain of ~28 full memory round trips per workgroup (measured: 27 % of
  // the kernel).
  {
    const int total = F_TE * m_len;
    constexpr int MU = 7;
    const float* cp = g.c.coupling + D.m_base;
    for (int i0 = t; i0 < total; i0 += 256 * MU) {
      float v[MU];
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int i = min(i0 + 256 * u, total - 1);
        const int el = i / m_len, j = i - el * m_len;
        const float val = cp[(long)(e0 + (el < ecnt ? el : 0)) * g.c.m_ld + j];
        v[u] = (el < ecnt) ? val : 0.f;
      }
#pragma unroll
      for (int u = 0; u < MU; ++u)
        if (i0 + 256 * u < total) sfc_lds[MT0 + i0 + 256 * u] = v[u];
    }
  }
  __syncthreads();
  tick(0);  // prologue
  if (PAIR && grp == 1) __syncthreads();  // half a slab step behind group 0 from here on
  const int nslab = D.nslab;
  for (int s = 0; s < nslab; ++s) {
    commit();
    tick(1);  // wait for the prefetched inputs + generation + LDS writes
    __syncthreads();
    tick(2);  // barrier
    if (s + 1 < nslab && !SFC_OFF(g, 4)) issue(s + 1);
    tick(3);  // issue of the next slab's loads
    if (!SFC_OFF(g, 1)) {
      if constexpr (X6) {
        switch (NT) {
          case 1: f_mma6<D3, 1, FT>(aidx, bidx, acc); break;
          case 2: f_mma6<D3, 2, FT>(aidx, bidx, acc); break;
          case 3: f_mma6<D3, 3, FT>(aidx, bidx, acc); break;
          default:
            if constexpr (FT >= 4) f_mma6<D3, 4, FT>(aidx, bidx, acc);
            break;
        }
      } else {
        switch (NT) {
          case 1: f_mma<D3, 1, FT>(aidx, bidx, acc); break;
          case 2: f_mma<D3, 2, FT>(aidx, bidx, acc); break;
          case 3: f_mma<D3, 3, FT>(aidx, bidx, acc); break;
          default:
            if constexpr (FT >= 4) f_mma<D3, 4, FT>(aidx, bidx, acc);
            break;
        }
      }
    }
    tick(4);  // MFMA loop
    __syncthreads();
    tick(5);  // barrier
  }

  // epilogue: row = m3 * 64 + el ; column c of the concatenated output
#pragma unroll
  for (int i = 0; i < FT; ++i) {
    if (i >= ntw) continue;
    const int c = ncol0 + boff[i] + r;
    float bvl = 0.f;
    if (D.l3 == 0) {
      if (c < D.N1) bvl = g.bias ? g.bias[c] : 0.f;
      else bvl = g.bias2 ? g.bias2[c - D.N1] : 0.f;
    }
    float* base;
    long ld;
    int coff, mstride;
    if (c < D.N1) {
      base = g.c.o1, ld = g.c.ld1, coff = D.out1_off + c, mstride = D.N1;
    } else {
      base = g.c.o2, ld = g.c.ld2, coff = c - D.N1, mstride = 0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = aoff[i] + (q & 3) + 8 * (q >> 2) + 4 * hi;
      const int m3 = row >> 6, el = row & 63;
      if (el < ecnt) base[(long)(e0 + el) * ld + coff + m3 * mstride] = acc[i][q] + bvl;
    }
  }
  if (PAIR && grp == 0) __syncthreads();  // the barrier that ends group 1's last MFMA phase
}

template <int MAXD, bool PAIR, bool X6 = false>
__global__ __launch_bounds__((PAIR ? 512 : 256), (PAIR || MAXD > 5 ? 1 : 2)) void sfc_fwd_kernel(const SfcFwdArgs g) {
  int tile, y;
  if (!order_xy(g.ord, blockIdx.x, tile, y)) return;
  const int di = g.y_deg[y];
  const int b = tile * g.nsplit[di] + g.y_split[y];
  switch (g.c.deg[di].d3) {
    case 1: f_block<1, MAXD, PAIR, X6>(g, di, b); break;
    case 3: f_block<3, MAXD, PAIR, X6>(g, di, b); break;
    case 5: f_block<5, MAXD, PAIR, X6>(g, di, b); break;
    default:
      if constexpr (MAXD >= 7) f_block<7, MAXD, PAIR, X6>(g, di, b);
      break;
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct SfcWgArgs {
  SfcCommon c;
  int nitem;
  int echunk;  // edges per workgroup (multiple of 8)
  SfcOrder ord;  // nx = edge chunks, ny = items
  short item_slab[2 * SFC_MAX_SLABS];  // global index of the group's first slab
  short item_ns[2 * SFC_MAX_SLABS];    // slabs in the group (1..4)
  short item_cls[2 * SFC_MAX_SLABS];   // compile-time tile count of the item's body: 1, 2, 4 or 8
  short item_col0[2 * SFC_MAX_SLABS];  // first column (multiple of 32) of the item's column range
  short item_ct[2 * SFC_MAX_SLABS];    // number of 32-column tiles
};

constexpr int W_SUB = 32;  // edges whose coupling matrices a wave keeps in LDS at a time

// One wave, one private edge range [ebeg, eend): acc[ct] += mid^T (32 channels x edges*m3) * d_out (edges*m3 x 32 cols)
template <int D1, int D3, int CTT>
__device__ __forceinline__ void wg_wave(const SfcWgArgs& g, const SfcSlab& S, const SfcDeg& D, const int col0,
                                        const int CT, float* __restrict__ Mw, const int ebeg, const int eend,
                                        f32x16 (&acc)[CTT]) {
  constexpr int LEN = D1 * D3;
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int N1 = D.N1;
  // Uniform (scalar) base pointer per 32-column tile: a tile lies entirely in the main consumer's columns or in the
  // second consumer's (N1 % 32 == 0 is checked on the host); per-lane offsets are 32-bit.
  const float* cb[CTT];
  unsigned cld[CTT], cm3[CTT];
#pragma unroll
  for (int ct = 0; ct < CTT; ++ct) {
    const int c0 = col0 + (ct < CT ? ct : 0) * 32;
    if (c0 < N1) {
      cb[ct] = g.c.o1 + D.out1_off + c0, cld[ct] = g.c.ld1, cm3[ct] = N1;
    } else {
      cb[ct] = g.c.o2 + (c0 - N1), cld[ct] = g.c.ld2, cm3[ct] = 0;
    }
  }
  const float* xs = g.c.x + S.x_off;
  const float* ws = g.c.w ? g.c.w + S.w_off : nullptr;
  const unsigned x_ld = g.c.x_ld, w_ld = g.c.w_ld, x_mul = S.x_mul;

  // Raw inputs (x, w, d_out values) of the next PF edge pairs live in a ring of PF statically named register slots
  // (the step loop is unrolled by PF, so there is no register rotation): a step's MFMAs are only D3 * CTT * 64 cycles
  // -- 320 for the degree-2 items -- against ~2 us of memory latency, so narrow items need a deep look-ahead.
  constexpr int PF = 4;
  static_assert(W_SUB % (2 * PF) == 0, "a group of PF steps must not straddle a coupling block");
  float xq[PF][D1], wq[PF], bq[PF][D3][CTT];
  auto fetch = [&](int e, int slot) __attribute__((always_inline)) {
    const int ee = e + hi;
    const bool valid = ee < eend;
    const unsigned er = valid ? ee : (e < eend ? e : ebeg);
    const unsigned xo = er * x_ld + r;
#pragma unroll
    for (int i = 0; i < D1; ++i) xq[slot][i] = xs[xo + i * x_mul];
    wq[slot] = ws ? ws[er * w_ld + r] : 1.0f;
    if (!valid) wq[slot] = 0.f;
#pragma unroll
    for (int m3 = 0; m3 < D3; ++m3)
#pragma unroll
      for (int ct = 0; ct < CTT; ++ct) bq[slot][m3][ct] = cb[ct][er * cld[ct] + r + m3 * cm3[ct]];
  };
  // Coupling matrices of W_SUB edges at a time in this wave's LDS block: the loads of a lane are issued in batches of
  // NB before the LDS writes.  The first version loaded one edge per iteration -- load / s_waitcnt vmcnt(0) /
  // ds_write, 32 memory round trips in a row for every 32 edges, more than the MFMA time of those edges.  (Fetching
  // the next block a whole block ahead costs 13 live registers and made hipcc spill.)
  constexpr int NL = (W_SUB * LEN + 63) / 64;
  constexpr int NB = NL < 5 ? NL : 5;  // loads in flight per lane (register budget)
  const float* cpl = g.c.coupling + S.m_off;
  const unsigned m_ld = g.c.m_ld;
  auto stage_m = [&](int s0) __attribute__((always_inline)) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k0 = 0; k0 < NL; k0 += NB) {
      float mreg[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int idx = min(lane + 64 * (k0 + k), W_SUB * LEN - 1);
        const int el = idx / LEN, j = idx - el * LEN;
        const int e = min(s0 + el, eend - 1);  // rows past the range: a finite copy of the last valid row (their w is 0)
        mreg[k] = cpl[(unsigned)e * m_ld + (unsigned)j];
      }
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (k0 + k < NL && lane + 64 * (k0 + k) < W_SUB * LEN) Mw[lane + 64 * (k0 + k)] = mreg[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  int sub0 = ebeg;
  stage_m(sub0);
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(ebeg + 2 * u, u);
  for (int e0 = ebeg; e0 < eend; e0 += 2 * PF) {  // steps past eend (at most PF - 1) run with w = 0
    if (e0 >= sub0 + W_SUB) {
      sub0 = e0;
      stage_m(sub0);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int e = e0 + 2 * u;
      // form the A values of this pair from slot u and the LDS-resident coupling matrices (rows of edges past the
      // range are clamped to the last valid edge: their w is 0, the coupling entry must merely be finite)
      const int ee = (e + hi < eend) ? e + hi : eend - 1;
      const float* mp = Mw + (ee - sub0) * LEN;
      float a[D3], bc[D3][CTT];
#pragma unroll
      for (int m3 = 0; m3 < D3; ++m3) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < D1; ++i) v = fmaf(mp[i * D3 + m3], xq[u][i], v);
        a[m3] = v * wq[u];
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) bc[m3][ct] = bq[u][m3][ct];
      }
      fetch(e + 2 * PF, u);  // consumed PF steps from now; out-of-range pairs re-read a valid row with w = 0
#pragma unroll
      for (int m3 = 0; m3 < D3; ++m3)
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct)  // no guards here: a conditional MFMA makes hipcc shuttle the accumulators
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m3], bc[m3][ct], acc[ct], 0, 0, 0);
    }
  }
}

// Workgroup = (group of up to 4 slabs of one degree sharing a column range, chunk of edges): wave w owns slab w of the
// group over the WHOLE chunk, so the four waves stream the same d_out rows at the same time (one HBM fetch, three L1 /
// L2 hits) and every wave finishes with its own [32 x CTT*32] block of the weight gradient -- no cross-wave reduction.
template <int CTT, int MAXD>
__device__ __forceinline__ void wg_item(const SfcWgArgs& g, const int item, const int chunk,
                                        float (&Msh)[4][W_SUB * MAXD * MAXD]) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= g.item_ns[item]) return;  // a group of fewer than 4 slabs
  const int slab = g.item_slab[item] + wave;
  const SfcSlab S = g.c.slab[slab];
  const SfcDeg& D = g.c.deg[S.deg];
  const int col0 = g.item_col0[item], CT = g.item_ct[item];
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int ebeg = chunk * g.echunk;
  const int eend = min(g.c.E, ebeg + g.echunk);
  if (ebeg >= eend) return;

  f32x16 acc[CTT];
#pragma unroll
  for (int i = 0; i < CTT; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

#define WG_CASE(A, B) wg_wave<A, B, CTT>(g, S, D, col0, CT, Msh[wave], ebeg, eend, acc)
#define WG_D3(A)                                                    \
  if constexpr (CTT > 4) { /* wide items: scalar degrees only */    \
    WG_CASE(A, 1);                                                  \
  } else {                                                          \
    switch (D.d3) {                                                 \
      case 1: WG_CASE(A, 1); break;                                 \
      case 3: WG_CASE(A, (MAXD >= 3 ? 3 : 1)); break;               \
      case 5: WG_CASE(A, (MAXD >= 5 ? 5 : 1)); break;               \
      default: WG_CASE(A, (MAXD >= 7 ? 7 : 1)); break;              \
    }                                                               \
  }
  if (ebeg < eend) {
    switch (S.d1) {
      case 1: WG_D3(1); break;
      case 3: WG_D3((MAXD >= 3 ? 3 : 1)); break;
      case 5: WG_D3((MAXD >= 5 ? 5 : 1)); break;
      default: WG_D3((MAXD >= 7 ? 7 : 1)); break;
    }
  }
#undef WG_D3
#undef WG_CASE

  // C[i = channel of the slab][j = column]
  const int slab_in_deg = slab - D.slab0;
#pragma unroll
  for (int ct = 0; ct < CTT; ++ct)
    if (ct < CT) {
      const int c0 = col0 + ct * 32;  // a 32-column tile lies entirely in one of the two weight matrices (uniform select)
      float* base = (c0 < D.N1) ? D.dW + c0 : D.dW2 + (c0 - D.N1);
      const int ldw = (c0 < D.N1) ? D.N1 : D.N2;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ch = slab_in_deg * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi;
        atomicAdd(base + ch * ldw + r, acc[ct][q]);
      }
    }
}

// ONE launch for all items: grid.y enumerates the items of every column-tile class (1, 2, 4, 8 tiles), sorted wide
// first; a single large grid packs the CUs better than one launch per class.
template <int MAXD>
__global__ __launch_bounds__(256, (MAXD <= 5 ? 2 : 1)) void sfc_wgrad_kernel(const SfcWgArgs g_byval) {
  // read the argument block in place (kernarg segment): with four instantiated bodies hipcc otherwise copies the
  // by-value struct to scratch and serves every table lookup from there
#if defined(__HIP_DEVICE_COMPILE__)
  const SfcWgArgs& g = *(const SfcWgArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)g_byval;
#else
  const SfcWgArgs& g = g_byval;
#endif
  __shared__ float Msh[4][W_SUB * MAXD * MAXD];  // per wave: [edge][i*d3 + m3] of the slab's path
  int chunk, item;
  if (!order_xy(g.ord, blockIdx.x, chunk, item)) return;
  switch (g.item_cls[item]) {
    case 8: wg_item<8, MAXD>(g, item, chunk, Msh); break;
    case 4: wg_item<4, MAXD>(g, item, chunk, Msh); break;
    case 2: wg_item<2, MAXD>(g, item, chunk, Msh); break;
    default: wg_item<1, MAXD>(g, item, chunk, Msh); break;
  }
}

// ------------------------------------------------------------------------------------------------ data gradient
constexpr int B_TE = 32;
constexpr int B_KC = 384;         // columns of d_out staged per chunk
constexpr int B_MAXGRP = 12;
constexpr int B_MAXPATH = 12;

struct SfcBPath {
  short deg;    // index into deg[]
  short krow;   // first row of the slab in W (channel index inside the degree)
  short w_off;  // offset of the slab's weights in the w row
  short m_off;  // offset of the path's coupling matrix in the coupling row
  short mt_off; // offset inside the per-edge LDS coupling block of the group
  short pad;
};
struct SfcBGroup {
  int x_off;   // offset of (segment l1, first chunk) in the x row
  short x_mul, d1, npath, mt_len;
  short nch, pad0;  // 32-channel chunks handled by the workgroup (1, 2 or 4; nch * d1 <= 6)
  SfcBPath p[B_MAXPATH];
};
struct SfcBwdArgs {
  SfcCommon c;
  float* dx;
  float* dw;  // may be null
  float* dM;  // may be null; ACCUMULATED with atomics
  int ngrp;
  int dt_floats;  // LDS partition
  int full_m;     // the LDS coupling block holds whole coupling rows (mt_len == m_ld, mt_off == m_off)
  SfcOrder ord;   // nx = edge tiles, ny = chunk groups
  int exp;        // development aid: 1 no MFMA loop, 2 no register epilogue, 4 no d_out staging after the first
  unsigned long long* dbg;  // optional phase timers (cycles of wave 0 / lane 0 of every workgroup), may be null
  SfcBGroup grp[B_MAXGRP];
};

// Workgroup = (32 edges, 32-channel chunk of one input segment); wave = (16-edge half eh, 16-channel half chh).
// d_mid tile of a path for (16 edges x 16 channels x all m3) = d_out tile (LDS, [k][m3*32+el]) x W_slab^T with
// v_mfma_f32_16x16x4_f32; its accumulator layout (lane = channel, 4 edges per lane) is exactly what the DTP backward
// contraction wants, so the epilogue runs in registers: no exchange of d_mid between waves at all.
template <int D1, int CG, int MAXD>
__device__ __forceinline__ void b_block(const SfcBwdArgs& g, const SfcBGroup& G, const int tile,
                                        float* __restrict__ smem) {
  float* __restrict__ Dt = smem;                // [k][32*d3 + 4]
  float* __restrict__ Mt = smem + g.dt_floats;  // [32][mt_len]
  const int e0 = tile * B_TE;
  const int ecnt = min(B_TE, g.c.E - e0);
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, j = lane & 15, kg = lane >> 4;
  const int eh = wave & 1, chh = wave >> 1;
  const int ch = 16 * chh + j;         // channel inside the chunk
  const int el0 = 16 * eh + 4 * kg;    // first of this lane's 4 edges
  const int mt_len = G.mt_len;

  unsigned long long t_mark = SFC_DBG(g) ? __builtin_amdgcn_s_memtime() : 0;
  auto tick = [&](int slot) __attribute__((always_inline)) {
    if (SFC_DBG(g)) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if (t == 0) atomicAdd(SFC_DBG(g) + slot, now - t_mark);
      t_mark = now;
    }
  };
  unsigned eo[4];   // clamped global edge index of this lane's edges
  bool ev[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ev[q] = el0 + q < ecnt;
    eo[q] = e0 + (ev[q] ? el0 + q : 0);
  }
  float xv[CG][4][D1], gx[CG][4][D1];
#pragma unroll
  for (int c = 0; c < CG; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < D1; ++i) {
        xv[c][q][i] = ev[q] ? g.c.x[(long)eo[q] * g.c.x_ld + G.x_off + 32 * c + i * G.x_mul + ch] : 0.f;
        gx[c][q][i] = 0.f;
      }
  // coupling matrices: either the whole (contiguous) coupling rows of the 32 edges in one coalesced pass, or -- when
  // those do not fit (L_max = 3) -- only the matrices of the group's paths
  // (batched, unconditional loads with clamped addresses -- see the forward kernel)
  if (g.full_m) {
    const float* src = g.c.coupling + (long)e0 * g.c.m_ld;
    const int total = B_TE * mt_len, nvalid = ecnt * mt_len;
    constexpr int MU = 9;
    for (int i0 = t; i0 < total; i0 += 256 * MU) {
      float v[MU];
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int i = i0 + 256 * u;
        const float val = src[min(i, nvalid - 1)];
        v[u] = (i < nvalid) ? val : 0.f;
      }
#pragma unroll
      for (int u = 0; u < MU; ++u)
        if (i0 + 256 * u < total) Mt[i0 + 256 * u] = v[u];
    }
  } else {
    for (int pi = 0; pi < G.npath; ++pi) {
      const SfcBPath P = G.p[pi];
      const int len = D1 * g.c.deg[P.deg].d3;
      const int total = B_TE * len;
      constexpr int MU = 7;
      const float* cp = g.c.coupling + P.m_off;
      for (int i0 = t; i0 < total; i0 += 256 * MU) {
        float v[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
          const int i = min(i0 + 256 * u, total - 1);
          const int el = i / len, jj = i - el * len;
          const float val = cp[(long)(e0 + (el < ecnt ? el : 0)) * g.c.m_ld + jj];
          v[u] = (el < ecnt) ? val : 0.f;
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
          const int i = i0 + 256 * u;
          if (i < total) {
            const int el = i / len, jj = i - el * len;
            Mt[el * mt_len + P.mt_off + jj] = v[u];
          }
        }
      }
    }
  }
  const float* const mrow = Mt + el0 * mt_len;
  tick(0);  // prologue: x loads issued, coupling tile staged

  int staged_deg = -1;
  auto path = [&](auto tag, const SfcBPath& P, const int pi) __attribute__((always_inline)) {
    constexpr int D3 = decltype(tag)::value;
    constexpr int ROWS = B_TE * D3, SD = ROWS + 4;
    constexpr int RPT = ROWS / 16;                 // rows per thread and 64-column block when staging
    constexpr int CBG = (RPT <= 2) ? 4 : 1;        // 64-column blocks loaded back to back (<= 14 float4 in flight)
    // weight fragments are prefetched PF k-blocks ahead (register rotation): with one row tile (D3 == 1) a k block
    // is only 4 MFMAs = 128 cycles, far less than an L2 round trip
    constexpr int PF = (D3 == 1) ? 4 : (D3 == 3 ? 2 : 1);
    const SfcDeg& D = g.c.deg[P.deg];
    const int Ncat = D.Ncat, N1 = D.N1;
    const int nchunk = (Ncat + B_KC - 1) / B_KC;
    auto chunk = [&](auto ctag) __attribute__((always_inline)) {  // one 32-channel chunk: MFMA loop + register epilogue
      constexpr int c = decltype(ctag)::value;
      // two accumulators per row tile only where a single dependent chain would stall the matrix pipe (D3 == 1);
      // with D3 >= 3 the row tiles themselves are independent chains and the second set only costs registers
      constexpr int NACC = (D3 == 1) ? 2 : 1;
      f32x4 acc[NACC][D3];
#pragma unroll
      for (int i = 0; i < D3; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int a_ = 0; a_ < NACC; ++a_) acc[a_][i][q] = 0.f;
      float wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        wv[q] = (ev[q] && g.c.w) ? g.c.w[(long)eo[q] * g.c.w_ld + P.w_off + 32 * c + ch] : 1.0f;  // used in the epilogue

      for (int ck = 0; ck < nchunk; ++ck) {
        const int kc0 = ck * B_KC, kcn = min(B_KC, Ncat - kc0);
        // MFMA k mapping: lane group kg supplies k = kb + 4 kg + jj for the jj-th instruction of a 16-wide k block
        // weight fragment of the 16-wide k block starting at column kk of the concatenated [main | second] weight
        // The columns [kc0, kc0+kcn) of the concatenated [main | second] weight split into at most two segments with
        // ONE row pointer each (a per-load pointer select costs ~18 % in this issue-bound loop).
        const long wr = P.krow + 32 * c + ch;
        const float* w1row = D.W + wr * N1 + 4 * kg;
        const float* w2row = D.W2 ? D.W2 + wr * D.N2 - N1 + 4 * kg : nullptr;
        const int ks = min(max(N1 - kc0, 0), kcn);  // columns [0, ks) of the chunk come from W, [ks, kcn) from W2
        const float* rowA = (ks > 0) ? w1row : w2row;
        const int endA = (ks > 0) ? ks : kcn;
        // fragments of the first two k blocks (segment lengths are multiples of 32): in flight across the staging
        f32x4 bA = *reinterpret_cast<const f32x4*>(rowA + kc0);
        f32x4 bB = *reinterpret_cast<const f32x4*>(rowA + kc0 + 16);
        if (!(nchunk == 1 && staged_deg == P.deg) && !(SFC_OFF(g, 4) && staged_deg >= 0)) {
          __syncthreads();  // readers of the previous Dt contents are done
          // stage Dt[k][row] = d_out[e0 + el, m3, kc0 + k],  row = m3*32 + el.  A wave step covers 16 float4 columns x
          // 4 rows; all loads of a group of column blocks are issued before the first LDS write.
          const int c4 = lane & 15, rr = lane >> 4;
          for (int cb0 = 0; cb0 < kcn; cb0 += 64 * CBG) {
            f32x4 v[CBG][RPT];
#pragma unroll
            for (int cg_ = 0; cg_ < CBG; ++cg_) {
              const int cc = cb0 + 64 * cg_ + 4 * c4;
              const int cg = kc0 + cc;
              const bool main = cg < N1;
#pragma unroll
              for (int i = 0; i < RPT; ++i) {
                const int row = 16 * i + wave * 4 + rr;
                const int m3 = row >> 5, el = row & 31;
                v[cg_][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cc < kcn && el < ecnt) {
                  const float* src = main ? g.c.o1 + (long)(e0 + el) * g.c.ld1 + D.out1_off + m3 * N1 + cg
                                          : g.c.o2 + (long)(e0 + el) * g.c.ld2 + (cg - N1);
                  v[cg_][i] = *reinterpret_cast<const f32x4*>(src);
                }
              }
            }
#pragma unroll
            for (int cg_ = 0; cg_ < CBG; ++cg_) {
              const int cc = cb0 + 64 * cg_ + 4 * c4;
              if (cc < kcn) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                  float* q = Dt + cc * SD + 16 * i + wave * 4 + rr;
                  q[0] = v[cg_][i][0], q[SD] = v[cg_][i][1], q[2 * SD] = v[cg_][i][2], q[3 * SD] = v[cg_][i][3];
                }
              }
            }
          }
          staged_deg = P.deg;
          __syncthreads();
          tick(1);  // staging of a d_out tile
        }
        const float* ap = Dt + (4 * kg) * SD + 16 * eh + j;
        // Two k blocks per iteration with statically named ping-pong registers (no register rotation, no copies): the
        // LDS operands (aA / aB) and the weight fragments (bA / bB) of a block are requested one block / two blocks
        // before its MFMAs are issued.
        auto seg = [&](const int kb0, const int kb1, const float* rowp) __attribute__((always_inline)) {
          // Weight fragments are requested FOUR k blocks (16 * D3 MFMAs) before their use: with one row tile (D3 == 1)
          // two blocks are only 256 cycles, less than an L2 round trip.  (bA, bB) hold blocks kb, kb+16 on entry;
          // (bC, bD) the following pair.  LDS operands (aA / aB) ping-pong one block ahead.
          const float* wp = rowp + kc0;
          f32x4 bC = *reinterpret_cast<const f32x4*>(wp + (kb0 + 32 < kb1 ? kb0 + 32 : kb0));
          f32x4 bD = *reinterpret_cast<const f32x4*>(wp + (kb0 + 48 < kb1 ? kb0 + 48 : kb0));
          float aA[4][D3], aB[4][D3];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int rt = 0; rt < D3; ++rt) aA[jj][rt] = ap[jj * SD + rt * 32];
          auto pair = [&](f32x4& p0, f32x4& p1, const int kb) __attribute__((always_inline)) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
              for (int rt = 0; rt < D3; ++rt) aB[jj][rt] = ap[(16 + jj) * SD + rt * 32];
            const f32x4 b0 = p0;
            p0 = *reinterpret_cast<const f32x4*>(wp + (kb + 64 < kb1 ? kb + 64 : kb0));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
              for (int rt = 0; rt < D3; ++rt)
                acc[jj & (NACC - 1)][rt] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(aA[jj][rt], b0[jj], acc[jj & (NACC - 1)][rt], 0, 0, 0);
            ap += 32 * SD;  // after the last block of the chunk this points past the tile: the reads below are then
                            // of in-bounds LDS garbage that is never used
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
              for (int rt = 0; rt < D3; ++rt) aA[jj][rt] = ap[jj * SD + rt * 32];
            const f32x4 b1 = p1;
            p1 = *reinterpret_cast<const f32x4*>(wp + (kb + 80 < kb1 ? kb + 80 : kb0));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
              for (int rt = 0; rt < D3; ++rt)
                acc[jj & (NACC - 1)][rt] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(aB[jj][rt], b1[jj], acc[jj & (NACC - 1)][rt], 0, 0, 0);
          };
          int kb = kb0;
#pragma unroll 1
          for (; kb + 64 <= kb1; kb += 64) {
            pair(bA, bB, kb);
            pair(bC, bD, kb + 32);
          }
          if (kb < kb1) pair(bA, bB, kb);  // segment lengths are multiples of 32: at most one pair is left
        };
        if (!SFC_OFF(g, 1)) seg(0, endA, rowA);
        if (endA < kcn && !SFC_OFF(g, 1)) {
          bA = *reinterpret_cast<const f32x4*>(w2row + kc0 + endA);
          bB = *reinterpret_cast<const f32x4*>(w2row + kc0 + endA + 16);
          seg(endA, kcn, w2row);
        }
      }
      tick(2);  // MFMA loop
      // DTP backward contraction in registers: this lane holds d_mid[m3][edge el0+q][channel 32c + ch]
      if (!SFC_OFF(g, 2))
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float gw = 0.f;
        const float* mp = mrow + q * mt_len + P.mt_off;
#pragma unroll
        for (int m3 = 0; m3 < D3; ++m3) {
          const float dm = (NACC == 2) ? acc[0][m3][q] + acc[NACC - 1][m3][q] : acc[0][m3][q];
          const float dmw = dm * wv[q];
          float tm = 0.f;
#pragma unroll
          for (int i = 0; i < D1; ++i) {
            const float m = mp[i * D3 + m3];
            tm = fmaf(m, xv[c][q][i], tm);
            gx[c][q][i] = fmaf(m, dmw, gx[c][q][i]);
            if (g.dM) {
              float v = dmw * xv[c][q][i];  // sum over the 16 channels of this wave (lanes j)
              v += __shfl_xor(v, 8);
              v += __shfl_xor(v, 4);
              v += __shfl_xor(v, 2);
              v += __shfl_xor(v, 1);
              if (j == 0 && ev[q]) atomicAdd(g.dM + (long)eo[q] * g.c.m_ld + P.m_off + i * D3 + m3, v);
            }
          }
          gw = fmaf(dm, tm, gw);
        }
        if (g.dw && ev[q]) g.dw[(long)eo[q] * g.c.w_ld + P.w_off + 32 * c + ch] = gw;
      }
      tick(3);  // register epilogue
    };
    // a real loop (not unrolled) around a static dispatch: keeps the chunks' live ranges apart
#pragma unroll 1
    for (int c = 0; c < CG; ++c) {
      switch (c) {
        case 0: chunk(IC<0>()); break;
        case 1: chunk(IC<(CG > 1 ? 1 : 0)>()); break;
        case 2: chunk(IC<(CG > 2 ? 2 : 0)>()); break;
        default: chunk(IC<(CG > 3 ? 3 : 0)>()); break;
      }
    }
  };

  for (int pi = 0; pi < G.npath; ++pi) {
    const SfcBPath P = G.p[pi];
    switch (g.c.deg[P.deg].d3) {
      case 1: path(IC<1>(), P, pi); break;
      case 3: path(IC<3>(), P, pi); break;
      case 5: path(IC<(MAXD >= 5 ? 5 : 1)>(), P, pi); break;
      default: path(IC<(MAXD >= 7 ? 7 : 1)>(), P, pi); break;
    }
  }
#pragma unroll
  for (int c = 0; c < CG; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ev[q]) {
#pragma unroll
        for (int i = 0; i < D1; ++i) g.dx[(long)eo[q] * g.c.x_ld + G.x_off + 32 * c + i * G.x_mul + ch] = gx[c][q][i];
      }
  tick(4);  // dx stores
}

// One launch covers every (input degree, chunk group) flavour: the flavours have different costs, and a single large
// grid packs the CUs much better than one launch per flavour (measured: 3 launches of 793 workgroups were 25 % slower).
template <int MAXD>
__global__ __launch_bounds__(256, (MAXD <= 5 ? 2 : 1)) void sfc_bwd_kernel(const SfcBwdArgs g) {
  int tile, gi;
  if (!order_xy(g.ord, blockIdx.x, tile, gi)) return;
  const SfcBGroup& G = g.grp[gi];
  switch (G.d1) {
    case 1:
      if (G.nch == 2) b_block<1, 2, MAXD>(g, G, tile, sfc_lds);
      else b_block<1, 1, MAXD>(g, G, tile, sfc_lds);
      break;
    case 3: b_block<3, 1, MAXD>(g, G, tile, sfc_lds); break;
    case 5: b_block<(MAXD >= 5 ? 5 : 1), 1, MAXD>(g, G, tile, sfc_lds); break;
    default: b_block<(MAXD >= 7 ? 7 : 1), 1, MAXD>(g, G, tile, sfc_lds); break;
  }
}

}  // namespace

extern "C" {

/* development aid (not declared in the public header): 8 x u64 device counters the data-gradient kernel adds its
 * per-phase cycle counts to; NULL disables */
int eqf_sfc_debug_order(int mode) {
  if (mode < -1 || mode > 2) return EQF_E_BADARG;
  g_sfc_order[0] = mode < 0 ? 0 : mode;  // -1: defaults
  g_sfc_order[1] = g_sfc_order[2] = mode < 0 ? 1 : mode;
  return 0;
}

int eqf_sfc_debug_x6_default(void) { return g_sfc_x6_default ? 1 : 0; }

int eqf_sfc_debug_exp(int mask) {
  g_sfc_exp = mask;
  return 0;
}

int eqf_sfc_debug_buffer(void* p) {
  g_sfc_dbg = (unsigned long long*)p;
  return 0;
}

int eqf_sfc_fwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                const float* const* Wl, const float* bias0, const float* W2, const float* bias2, float* out1,
                const eqf_irreps* out1_irreps, float* out2, int n2, int E, void* stream) {
  if (!Wl || (n2 > 0 && !W2)) return EQF_E_BADARG;
  SfcFwdArgs A;
  int rc = build_common(x, coupling, w, paths, Wl, W2, nullptr, nullptr, out1, out1_irreps, out2, n2, E, A.c);
  if (rc) return rc;
  if (E <= 0) return 0;
  A.bias = bias0;
  A.bias2 = bias2;
  A.dbg = g_sfc_dbg;
  A.exp = g_sfc_exp;
  int md = max_d1(A.c);
  for (int d = 0; d < A.c.ndeg; ++d) md = A.c.deg[d].d3 > md ? A.c.deg[d].d3 : md;
  const int ft = md <= 5 ? 3 : F_MAXT;
  const int ntile = eqf_cdiv(E, F_TE);
  // matrix step: 1 = split-precision bf16 x 6 on the matrix cores (f_mma6), 0 = exact-fp32 MFMA (development switch
  // 64 of eqf_sfc_debug_exp selects the other one for A/B runs)
  const bool x6 = md <= 5 && ((g_sfc_exp & 64) ? !g_sfc_x6_default : g_sfc_x6_default);
  size_t lds = 0;
  int ny = 0;
  for (int d = 0; d < A.c.ndeg; ++d) {
    const SfcDeg& D = A.c.deg[d];
    if (!D.W) return EQF_E_BADARG;
    const int RT = F_TE * D.d3 / 32;
    (void)RT;
    int maxct = f_ctcap(D.d3, ft);  // tiles per workgroup <= 4 waves x accumulator tiles per wave
    if (maxct < 1) return EQF_E_UNSUPPORTED;
    const int cttot = D.Ncat / 32;
    A.nsplit[d] = eqf_cdiv(cttot, maxct);
    A.cps[d] = eqf_cdiv(cttot, A.nsplit[d]) * 32;
    for (int k = 0; k < A.nsplit[d]; ++k) {
      if (ny >= 16) return EQF_E_UNSUPPORTED;
      A.y_deg[ny] = (signed char)d, A.y_split[ny] = (signed char)k;
      ++ny;
    }
    const size_t a_floats = x6 ? (size_t)F_TE * D.d3 * X6_SA : (size_t)32 * (F_TE * D.d3 + 1);
    const size_t need = sizeof(float) * (a_floats + 32 * f_sb(D.d3, ft) + (size_t)F_TE * D.m_len);
    if (need > lds) lds = need;
  }
  if (lds > SFC_LDS_LIMIT) return EQF_E_UNSUPPORTED;
  lds = (lds + 15) & ~(size_t)15;
  // paired workgroups measured SLOWER than free-running ones (sep_act 297 vs 230 us, tools/sfc_exp.py): kept behind the
  // development switch only
  const bool pair = md <= 5 && !x6 && 2 * lds <= SFC_LDS_LIMIT && (g_sfc_exp & 16);
  A.grp_floats = (int)(lds / sizeof(float));
  int blk = 0;
  A.ord = make_order(0, pair ? eqf_cdiv(ntile, 2) : ntile, ny, blk);

  hipStream_t st = (hipStream_t)stream;
  const int pid = eqf_prof_begin("sfc_fwd", st, sfc_flops(A.c), sfc_bytes(A.c));
  // the dynamic-LDS limit is a per-device function attribute: one flag per device (a process drives one GPU, but
  // nothing in the ABI forbids several)
  static bool attr_done[64] = {};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_set = attr_done[dev_id & 63];
  if (!attr_set) {
    hipFuncSetAttribute((const void*)sfc_fwd_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SFC_LDS_LIMIT);
    hipFuncSetAttribute((const void*)sfc_fwd_kernel<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SFC_LDS_LIMIT);
    hipFuncSetAttribute((const void*)sfc_fwd_kernel<7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SFC_LDS_LIMIT);
    hipFuncSetAttribute((const void*)sfc_fwd_kernel<5, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        SFC_LDS_LIMIT);
    attr_set = true;
  }
  if (x6 && !pair)
    hipLaunchKernelGGL((sfc_fwd_kernel<5, false, true>), dim3(blk), dim3(256), (g_sfc_exp & 32) ? (size_t)100 * 1024 : lds, st, A);
  else if (pair) hipLaunchKernelGGL((sfc_fwd_kernel<5, true>), dim3(blk), dim3(512), 2 * lds, st, A);
  else if (md <= 5)
    hipLaunchKernelGGL((sfc_fwd_kernel<5, false>), dim3(blk), dim3(256), (g_sfc_exp & 32) ? (size_t)100 * 1024 : lds, st, A);
  else hipLaunchKernelGGL((sfc_fwd_kernel<7, false>), dim3(blk), dim3(256), lds, st, A);
  eqf_prof_end(pid, st);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_sfc_bwd_weight(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                       const float* d_out1, const eqf_irreps* out1_irreps, const float* d_out2, int n2,
                       float* const* dWl, float* dW2, int E, void* stream) {
  if (!dWl || (n2 > 0 && !dW2)) return EQF_E_BADARG;
  SfcWgArgs A;
  int rc = build_common(x, coupling, w, paths, nullptr, nullptr, dWl, dW2, const_cast<float*>(d_out1), out1_irreps,
                        const_cast<float*>(d_out2), n2, E, A.c);
  if (rc) return rc;
  if (E <= 0) return 0;
  int md = max_d1(A.c);
  for (int d = 0; d < A.c.ndeg; ++d) md = A.c.deg[d].d3 > md ? A.c.deg[d].d3 : md;
  hipStream_t st = (hipStream_t)stream;
  // Items = (slab, column range of CTT tiles) with CTT in {1, 2, 4, 8} fixed at compile time (8 only for scalar
  // degrees); a degree with cttot column tiles is decomposed greedily, the last item may be padded (its surplus
  // tiles recompute tile 0 and are dropped).  One launch per class.
  for (int d = 0; d < A.c.ndeg; ++d)
    if (!A.c.deg[d].dW) return EQF_E_BADARG;
  const int pid = eqf_prof_begin("sfc_wgrad", st, sfc_flops(A.c), sfc_bytes(A.c));
  const int classes[4] = {8, 4, 2, 1};
  A.nitem = 0;
  for (int ci = 0; ci < 4; ++ci) {
    const int cls = classes[ci];
    for (int d = 0; d < A.c.ndeg; ++d) {
      const SfcDeg& D = A.c.deg[d];
      int ct0 = 0, rem = D.Ncat / 32;
      while (rem > 0) {
        int take, used;
        if (D.d3 == 1 && rem >= 7) take = 8;
        else if (rem >= 3) take = 4;
        else take = rem;  // 1 or 2
        used = rem < take ? rem : take;
        if (take == cls)
          for (int q = 0; q < D.nslab; q += 4) {
            if (A.nitem >= 2 * SFC_MAX_SLABS) return EQF_E_UNSUPPORTED;
            A.item_slab[A.nitem] = (short)(D.slab0 + q);
            A.item_ns[A.nitem] = (short)(D.nslab - q < 4 ? D.nslab - q : 4);
            A.item_col0[A.nitem] = (short)(ct0 * 32);
            A.item_ct[A.nitem] = (short)used;
            A.item_cls[A.nitem] = (short)cls;
            A.nitem++;
          }
        ct0 += used, rem -= used;
      }
    }
  }
  if (A.nitem > 0) {
    // few, long-running workgroups: every wave ends with 32 x 32 x CTT atomics
    int z = eqf_cdiv(1536, A.nitem);
    int echunk = eqf_cdiv(E, z);
    echunk = ((echunk + 7) / 8) * 8;
    if (echunk < 128) echunk = 128;
    A.echunk = echunk;
    z = eqf_cdiv(E, echunk);
    int nblk = 0;
    A.ord = make_order(2, z, A.nitem, nblk);
    dim3 grid(nblk);
    if (md <= 5)
      hipLaunchKernelGGL((sfc_wgrad_kernel<5>), grid, dim3(256), 0, st, A);
    else
      hipLaunchKernelGGL((sfc_wgrad_kernel<7>), grid, dim3(256), 0, st, A);
    EQF_CHECK_LAUNCH();
  }
  eqf_prof_end(pid, st);
  return 0;
}

int eqf_sfc_bwd_data(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                     const float* const* Wl, const float* W2, const float* d_out1, const eqf_irreps* out1_irreps,
                     const float* d_out2, int n2, float* dx, float* dw, float* d_coupling, int E, void* stream) {
  if (!Wl || !dx || (n2 > 0 && !W2)) return EQF_E_BADARG;
  SfcBwdArgs A;
  int rc = build_common(x, coupling, w, paths, Wl, W2, nullptr, nullptr, const_cast<float*>(d_out1), out1_irreps,
                        const_cast<float*>(d_out2), n2, E, A.c);
  if (rc) return rc;
  if (E <= 0) return 0;
  A.dx = dx, A.dw = (w ? dw : nullptr), A.dM = d_coupling;
  A.dbg = g_sfc_dbg;
  A.exp = g_sfc_exp;
  // groups = (input segment, 32-channel chunk); paths sorted by output degree so that a staged d_out tile is shared
  A.ngrp = 0;
  int d3max = 1, mtmax = 0;
  size_t dtmax = 0;
  const eqf_dtp_paths* P = paths;
  for (int d = 0; d < A.c.ndeg; ++d) {
    const SfcDeg& D = A.c.deg[d];
    if (!D.W) return EQF_E_BADARG;
    if (D.d3 > d3max) d3max = D.d3;
    const size_t dt = (size_t)(D.Ncat < B_KC ? D.Ncat : B_KC) * (B_TE * D.d3 + 4);
    if (dt > dtmax) dtmax = dt;
  }
  // distinct input segments
  int seg_off[EQF_MAX_SEG], seg_mul[EQF_MAX_SEG], seg_l[EQF_MAX_SEG], nseg = 0;
  for (int p = 0; p < P->npaths; ++p) {
    bool found = false;
    for (int s = 0; s < nseg; ++s) found |= seg_off[s] == P->in_off[p];
    if (!found) {
      if (nseg >= EQF_MAX_SEG) return EQF_E_UNSUPPORTED;
      seg_off[nseg] = P->in_off[p], seg_mul[nseg] = P->mul[p], seg_l[nseg] = P->l1[p];
      nseg++;
    }
  }
  for (int s = 0; s < nseg; ++s) {
    const int d1s = 2 * seg_l[s] + 1;
    const int nchunks = seg_mul[s] / 32;
    int cgsz = d1s == 1 ? 2 : 1;  // chunks per workgroup (register budget: more chunks per workgroup spill)
    while (nchunks % cgsz != 0) cgsz >>= 1;
    for (int c = 0; c < seg_mul[s]; c += 32 * cgsz) {
      if (A.ngrp >= B_MAXGRP) return EQF_E_UNSUPPORTED;
      SfcBGroup& G = A.grp[A.ngrp];
      G.nch = (short)cgsz, G.pad0 = 0;
      G.x_off = seg_off[s] + c;
      G.x_mul = (short)seg_mul[s];
      G.d1 = (short)(2 * seg_l[s] + 1);
      G.npath = 0;
      int mt = 0;
      for (int d = 0; d < A.c.ndeg; ++d)
        for (int p = 0; p < P->npaths; ++p) {
          if (P->in_off[p] != seg_off[s] || P->l3[p] != A.c.deg[d].l3) continue;
          if (G.npath >= B_MAXPATH) return EQF_E_UNSUPPORTED;
          SfcBPath& Q = G.p[G.npath++];
          Q.deg = (short)d;
          Q.krow = (short)(P->out_ch[p] + c);
          Q.w_off = (short)(P->w_off[p] + c);
          Q.m_off = (short)P->m_off[p];
          Q.mt_off = (short)mt;
          Q.pad = 0;
          if (P->w_off[p] + c > 32767) return EQF_E_UNSUPPORTED;
          mt += G.d1 * A.c.deg[d].d3;
        }
      G.mt_len = (short)mt;
      if (mt > mtmax) mtmax = mt;
      if (G.npath > 0) A.ngrp++;
    }
  }
  A.full_m = P->m_numel <= 192;  // 32 whole coupling rows <= 24 KB of LDS
  if (A.full_m) {
    mtmax = P->m_numel;
    for (int gi = 0; gi < A.ngrp; ++gi) {
      A.grp[gi].mt_len = (short)P->m_numel;
      for (int q = 0; q < A.grp[gi].npath; ++q) A.grp[gi].p[q].mt_off = A.grp[gi].p[q].m_off;
    }
  }
  if (A.ngrp == 0) return EQF_E_BADARG;
  A.dt_floats = (int)((dtmax + 3) & ~(size_t)3);
  const size_t lds = sizeof(float) * ((size_t)A.dt_floats + (size_t)B_TE * mtmax);
  if (lds > SFC_LDS_LIMIT) return EQF_E_UNSUPPORTED;
  const int md = max_d1(A.c) > d3max ? max_d1(A.c) : d3max;
  hipStream_t st = (hipStream_t)stream;
  int nblk = 0;
  A.ord = make_order(1, eqf_cdiv(E, B_TE), A.ngrp, nblk);
  dim3 grid(nblk);
  const int pid = eqf_prof_begin("sfc_bwd_data", st, sfc_flops(A.c), sfc_bytes(A.c));
  if (md <= 5) {
    static bool attr5_done[64] = {};
    int dev5 = 0;
    (void)hipGetDevice(&dev5);
    bool& attr5 = attr5_done[dev5 & 63];
    if (!attr5) {
      hipFuncSetAttribute((const void*)sfc_bwd_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, SFC_LDS_LIMIT);
      attr5 = true;
    }
    hipLaunchKernelGGL(sfc_bwd_kernel<5>, grid, dim3(256), (g_sfc_exp & 32) ? (size_t)100 * 1024 : lds, st, A);
  } else {
    static bool attr7_done[64] = {};
    int dev7 = 0;
    (void)hipGetDevice(&dev7);
    bool& attr7 = attr7_done[dev7 & 63];
    if (!attr7) {
      hipFuncSetAttribute((const void*)sfc_bwd_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, SFC_LDS_LIMIT);
      attr7 = true;
    }
    hipLaunchKernelGGL(sfc_bwd_kernel<7>, grid, dim3(256), lds, st, A);
  }
  eqf_prof_end(pid, st);
  EQF_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
