// Fused SeparableFCTP weight gradient, second generation (round 6): workgroups of up to four slabs that share their d_out tiles.
//
//   dW_l3[(p,u), n] += sum_e sum_m3 mid[e,(p,u),m3] * d_out[e,l3,m3,n],   mid = w * sum_i M_p[i,m3] x[l1(p),i,u]
//
// [ref: the weight gradients autograd derives for SeparableFCTP.forward, nets/graph_attention_transformer.py:234-248]
//
// Why a second kernel (sfcx.hip's one-wave weight gradient stays as the cross-check and serves small graphs / degree-3 models):
// there every item = (32-channel slab of a path, column group) is one wave that fetches, transposes and splits the d_out tiles
// of its column group itself -- and the 7 / 12 / 11 slabs of an output degree of the QM9 operator all do that for the SAME tiles:
// 45 % of the launch's vector instructions and most of its 60 KB of requests per edge (9 KB of them distinct rows).  Alone on its
// SIMD a wave is ~30 % busy and waits for one memory round trip per component; three per SIMD slow one another 2x through the memory
// path (profiles/r06/r06_aa_*: software pipelining the one-wave kernel and cost-balancing its launch both measured slower).
//
//   * workgroup = up to 4 waves = 4 slabs of ONE output degree on one column group and one chunk of edges, in step half-step
//     (16 edges = the K of one matrix instruction) by half-step;
//   * the d_out tiles of half-step h + 1 (2 l3 + 1 components x <= 3 / 2 / 1 column tiles) are dealt round-robin to the waves:
//     the owner requests the row-major tile BEFORE the arithmetic of half-step h, afterwards transposes it through its private
//     LDS tile, splits it into bf16 planes once and leaves the planes in the workgroup's LDS in B-fragment order (double
//     buffered); every wave then reads its B operands with ds_read_b128 -- no wave splits a tile another one has split;
//   * x / w tiles of the NEXT half-step and the coupling block of the next 32 edges are requested one half-step ahead as well
//     (256 registers per wave at two workgroups per CU leave room for them);
//   * one s_barrier per half-step is the only synchronisation;
//   * the products and their order inside a (slab, column tile) accumulator are those of the one-wave kernel; chunk boundaries
//     differ (800 instead of 480 edges at the bench size), so the atomically added partial sums group differently.
#include "sfcx_common.h"

extern __shared__ __attribute__((aligned(16))) float sw_lds[];

namespace {
using namespace sfc;

// development: -DEQF_W_TRACE=1 prints the cycles every wave of the first workgroup of every type spends per phase
#ifndef EQF_W_TRACE
#define EQF_W_TRACE 0
#endif
#if EQF_W_TRACE
#define WT_STAMP(k)                        \
  do {                                     \
    const long long tn = clock64();        \
    wt[k] += tn - wt_last, wt_last = tn;   \
  } while (0)
#else
#define WT_STAMP(k) \
  do {              \
  } while (0)
#endif
constexpr int W_WAVES = 4;
constexpr int W_MAXTYPE = 48;
constexpr int W_PLANE_FLOATS = 256;  // one bf16 plane of a 16 x 32 tile in fragment order: 64 lanes x 16 bytes

struct WSlab {
  int x_off, w_off, m_off;
  int g_off;  // gated input: offset of the slab's gate scalars in the raw row (-1 scalar segment, -2 plain input)
  short x_mul, d1, sid, pad;
};
struct WType {
  short deg, slab0, nsl, ct0, ct, pad;
};
struct WArgs {
  const float *x, *coupling, *w;
  int x_ld, m_ld, w_ld, E;
  const float *d1, *d2;
  int ld1, ld2;
  int echunk, ntype, only_type;
  int plane_floats, wave_floats;  // LDS: [2 buffers of planes][per wave: coupling block, four transposition tiles]
  float *db, *db2;
  XGate gate;
  SfcOrder ord;  // nx = edge chunks, ny = types
  struct Deg {
    float *dW, *dW2;
    int d3, N1, N2, out1_off;
  } deg[SFC_MAX_DEG];
  WSlab slab[SFC_MAX_SLABS];
  WType type[W_MAXTYPE];
};

__device__ __forceinline__ void w_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// coupling rows of a 32-edge block, LEN columns, flat: element f = lane + 64 k of the [32][LEN] block (row f / LEN, column
// f % LEN) -- ceil(LEN / 2) loads of consecutive floats per row instead of 16 two-row loads with 32 - LEN idle lanes
template <int LEN>
__device__ __forceinline__ void m_fetch(float (&v)[(LEN + 1) / 2], const float* __restrict__ cp, const unsigned m_ld, const int e0,
                                        const int elast, const int lane) {
#pragma unroll
  for (int k = 0; k < (LEN + 1) / 2; ++k) {
    const int f = lane + 64 * k, row = f / LEN, col = f - row * LEN;
    v[k] = cp[(unsigned)min(e0 + row, elast) * m_ld + (unsigned)col];  // (rows >= 32 of the last load: clamped re-reads, not stored)
  }
}
template <int LEN, int LENP>
__device__ __forceinline__ void m_store(float* __restrict__ Mt, const float (&v)[(LEN + 1) / 2], const int lane) {
#pragma unroll
  for (int k = 0; k < (LEN + 1) / 2; ++k) {
    const int f = lane + 64 * k, row = f / LEN, col = f - row * LEN;
    if (f < 32 * LEN) Mt[row * LENP + col] = v[k];
  }
}

template <int D1, int D3, int CT, int MODE>
__device__ __forceinline__ void w_wave(const WArgs& g, const WType& T, const WSlab& S, const int wave, const int nact,
                                       const int ebeg, const int eend) {
  constexpr int NPA = Planes<MODE>::A, LEN = D1 * D3, LENP = (LEN + 3) & ~3, NT = D3 * CT, KMAX = (NT + 2) / 3, MV = (LEN + 1) / 2;
  constexpr bool MPRE = LEN <= 15;  // coupling block requested a half-step ahead (13 more registers for LEN = 25: scratch)  // (KMAX: the planner forms groups of >= 3 slabs wherever a wave would own more)
  const WArgs::Deg& D = g.deg[T.deg];
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int ct0 = T.ct0;
  float* const planes = sw_lds;
  float* const Mw = sw_lds + g.plane_floats + wave * g.wave_floats;  // [32 edges][LENP]
  float* const Tw = Mw + 32 * LENP;                                  // four 16 x 32 transposition tiles

  f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[ct][q] = 0.f;

  // this wave's d_out tiles: t = wave + k nact (component m3 = t / CT, column tile t % CT of the group).  Slot 0 is worked on
  // unconditionally -- a wave without a tile (three tiles, four waves) repeats tile 0 into a plane slot nobody reads -- the
  // further slots under one uniform branch each: the step body stays in few, large basic blocks.
  const float* tb_base[KMAX];  // uniform
  unsigned tb_ld[KMAX];
  int tb_t[KMAX];
  bool tb_on[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int t = wave + k * nact;
    tb_on[k] = t < NT;
    tb_t[k] = t < NT ? t : NT;  // (NT: the spare plane slot)
    const int tt = t < NT ? t : 0;
    const int m3 = tt / CT, c0 = (ct0 + (tt - m3 * CT)) * 32;
    if (c0 < D.N1) tb_base[k] = g.d1 + D.out1_off + c0 + m3 * D.N1, tb_ld[k] = g.ld1;
    else tb_base[k] = g.d2 + (c0 - D.N1), tb_ld[k] = g.ld2;
  }
  const float* const xs = g.x + S.x_off;
  const bool has_w = g.w != nullptr;
  const float* const ws = has_w ? g.w + S.w_off : xs;
  const unsigned x_ld = g.x_ld, w_ld = has_w ? g.w_ld : g.x_ld, x_mul = S.x_mul;
  const bool gated_seg = g.gate.on && S.g_off >= 0;  // uniform
  // bias gradients (degree 0 only): the workgroups that hold the FIRST slab of the degree cover every column once per edge
  // chunk; the owner of a tile sums the values it splits anyway
  const bool do_bias = D3 == 1 && g.slab[T.slab0].sid == 0 && (g.db != nullptr || g.db2 != nullptr);
  float bsum[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) bsum[k] = 0.f;

  Tile16 nx[D1], nw, ng, nb[KMAX];
  float mv[MV];
  nw.t0 = f32x4{0.f, 0.f, 0.f, 0.f}, nw.t1 = nw.t0;
  auto fetch_xw = [&](const int e_lo) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < D1; ++i) tile16_fetch(nx[i], xs + i * x_mul, x_ld, e_lo, eend - 1, lane);
    if (has_w) tile16_fetch(nw, ws, w_ld, e_lo, eend - 1, lane);  // (without per-edge weights the tile's values are replaced by ones)
    if (gated_seg) tile16_fetch(ng, g.x + S.g_off, x_ld, e_lo, eend - 1, lane);
  };
  auto fetch_b = [&](const int e_lo) __attribute__((always_inline)) {
    tile16_fetch(nb[0], tb_base[0], tb_ld[0], e_lo, eend - 1, lane);
#pragma unroll
    for (int k = 1; k < KMAX; ++k)
      if (tb_on[k]) tile16_fetch(nb[k], tb_base[k], tb_ld[k], e_lo, eend - 1, lane);
  };
  // the tiles in nb (rows e_lo .. e_lo + 15) -> planes of buffer `buf`
  auto produce_slot = [&](auto tag, const int buf, const int e_lo) __attribute__((always_inline)) {
    constexpr int k = decltype(tag)::value;
    wave_lds_order();  // earlier reads of the transposition tile are issued
    tile16_put(Tw + k * XT16_FLOATS, nb[k], lane);
    wave_lds_order();
    float b[8];
    tile16_get(Tw + k * XT16_FLOATS, lane, b);
    if constexpr (D3 == 1) {
      if (do_bias) {  // uniform
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += (e_lo + 8 * hi + j < eend) ? b[j] : 0.f;  // (rows past the chunk are clamped re-reads)
        bsum[k] += t;
      }
    }
    bf16x8 pb[NPA];
    split_planes<NPA>(b, pb);
    float* const dst = planes + ((buf * (NT + 1) + tb_t[k]) * NPA) * W_PLANE_FLOATS + lane * 4;
#pragma unroll
    for (int pl = 0; pl < NPA; ++pl) *reinterpret_cast<bf16x8*>(dst + pl * W_PLANE_FLOATS) = pb[pl];
  };
  auto produce = [&](const int buf, const int e_lo) __attribute__((always_inline)) {
    produce_slot(IC<0>(), buf, e_lo);
    if constexpr (KMAX > 1) {
      if (tb_on[1]) produce_slot(IC<1>(), buf, e_lo);
    }
    if constexpr (KMAX > 2) {
      if (tb_on[2]) produce_slot(IC<2>(), buf, e_lo);
    }
  };

  const int nhs = (eend - ebeg + 15) >> 4;
#if EQF_W_TRACE
  long long wt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long wt_last = clock64();
  const long long wt_begin = wt_last;
#endif
  fetch_b(ebeg);
  fetch_xw(ebeg);
  if constexpr (MPRE) m_fetch<LEN>(mv, g.coupling + S.m_off, g.m_ld, ebeg, eend - 1, lane);
  produce(0, ebeg);
  w_barrier();
  WT_STAMP(0);  // prologue
#pragma unroll 1
  for (int hs = 0; hs < nhs; ++hs) {
    const int half = hs & 1, buf = hs & 1;
    const int e_lo = ebeg + 16 * hs;
    const int ec = e_lo + 8 * hi;  // this lane's 8 edges: ec + j
    fetch_b(e_lo + 16);
    WT_STAMP(6);  // d_out tile requests issued
    if (!half) {  // the coupling block requested one half-step ago (before the loop for the first)
      if constexpr (!MPRE) m_fetch<LEN>(mv, g.coupling + S.m_off, g.m_ld, e_lo, eend - 1, lane);
      wave_lds_order();
      m_store<LEN, LENP>(Mw, mv, lane);
    }
    float xq[8][D1], wq[8];
#pragma unroll
    for (int b0 = 0; b0 < D1 + 1; b0 += 4) {
      wave_lds_order();  // the reads of the tiles' previous use are issued
#pragma unroll
      for (int k = b0; k < b0 + 4 && k < D1 + 1; ++k) tile16_put(Tw + (k & 3) * XT16_FLOATS, k < D1 ? nx[k < D1 ? k : 0] : nw, lane);
      wave_lds_order();
      if (b0 == 0) WT_STAMP(7);  // x / w tiles arrived and written to LDS
#pragma unroll
      for (int k = b0; k < b0 + 4 && k < D1 + 1; ++k) {
        float t[8];
        tile16_get(Tw + (k & 3) * XT16_FLOATS, lane, t);
        if (k < D1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xq[j][k < D1 ? k : 0] = t[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) wq[j] = has_w ? t[j] : 1.0f;
        }
      }
    }
    if (g.gate.on) {  // uniform: xq holds the gate's INPUT rows
      if (S.g_off == -1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xq[j][0] = g.gate.c_silu * xq[j][0] * xg_sigmoid(xq[j][0]);
      } else {
        float gq[8];
        wave_lds_order();
        tile16_put(Tw, ng, lane);
        wave_lds_order();
        tile16_get(Tw, lane, gq);
#pragma unroll
        for (int j = 0; j < 8; ++j) wq[j] *= g.gate.c_sig * xg_sigmoid(gq[j]);  // (the gate of (edge, channel) scales all i alike)
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (ec + j >= eend) wq[j] = 0.f;
    WT_STAMP(1);  // x / w tiles out of the registers, through LDS
    fetch_xw(e_lo + 16);                                                                    // next half-step's x / w
    if constexpr (MPRE)
      if (half) m_fetch<LEN>(mv, g.coupling + S.m_off, g.m_ld, e_lo + 16, eend - 1, lane);  // next block's coupling rows
    // A operands of all components, edge by edge: the entries of an edge's coupling row are read once, (i, .) group by group
    // (adjacent floats: the compiler merges them into 8 / 16-byte LDS reads where the offsets are aligned)
    float a_all[D3][8];
    const float* const mp = Mw + (16 * half + 8 * hi) * LENP;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int m3 = 0; m3 < D3; ++m3) a_all[m3][j] = 0.f;
#pragma unroll
      for (int i = 0; i < D1; ++i) {
        float mi[D3];
#pragma unroll
        for (int m3 = 0; m3 < D3; ++m3) mi[m3] = mp[j * LENP + i * D3 + m3];
#pragma unroll
        for (int m3 = 0; m3 < D3; ++m3) a_all[m3][j] = fmaf(mi[m3], xq[j][i], a_all[m3][j]);
      }
#pragma unroll
      for (int m3 = 0; m3 < D3; ++m3) a_all[m3][j] *= wq[j];
    }
    const float* const pbuf = planes + (buf * (NT + 1) * NPA) * W_PLANE_FLOATS + lane * 4;
#pragma unroll
    for (int m3 = 0; m3 < D3; ++m3) {
      bf16x8 pa[NPA];
      split_planes<NPA>(a_all[m3], pa);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        bf16x8 pb[NPA];
        const float* const src = pbuf + ((m3 * CT + ct) * NPA) * W_PLANE_FLOATS;
#pragma unroll
        for (int pl = 0; pl < NPA; ++pl) pb[pl] = *reinterpret_cast<const bf16x8*>(src + pl * W_PLANE_FLOATS);
        mma_terms<NPA, NPA>(pa, pb, acc[ct]);
      }
    }
    WT_STAMP(2);  // generation + matrix instructions (issue)
    produce(buf ^ 1, e_lo + 16);
    WT_STAMP(3);  // own d_out tiles -> planes
    w_barrier();
    WT_STAMP(4);  // barrier
  }
  // C[row = channel of the slab][column]: register q of lane (r, hi) = row (q & 3) + 8 (q >> 2) + 4 hi, column r
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c0 = (ct0 + ct) * 32;
    float* base = (c0 < D.N1) ? D.dW + c0 : D.dW2 + (c0 - D.N1);
    const int ldw = (c0 < D.N1) ? D.N1 : D.N2;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ch = S.sid * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi;
      atomicAdd(base + (size_t)ch * ldw + r, acc[ct][q]);
    }
  }
#if EQF_W_TRACE
  WT_STAMP(5);
  if (blockIdx.x < 8 * g.ntype && (blockIdx.x & 7) == 0 && lane == 0)
    printf("wtrace d1 %d d3 %d ct %d wave %d/%d half-steps %d: total %lld prologue %lld issue_b %lld xw_arrive %lld xw_get %lld gen+mma %lld produce %lld barrier %lld epilogue %lld\n",
           D1, D3, CT, wave, nact, nhs, clock64() - wt_begin, wt[0], wt[6], wt[7], wt[1], wt[2], wt[3], wt[4], wt[5]);
#endif
  if constexpr (D3 == 1) {
    if (do_bias) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (!tb_on[k]) continue;
        const int c0 = (ct0 + tb_t[k]) * 32;  // (D3 == 1: tile index = column tile of the group)
        float* const bb = (c0 < D.N1) ? g.db : g.db2;  // uniform
        const float v = bsum[k] + __shfl_xor(bsum[k], 32);  // the two 8-edge halves of the 16-edge tiles
        if (bb != nullptr && hi == 0) atomicAdd(bb + ((c0 < D.N1) ? c0 : c0 - D.N1) + r, v);
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(64 * W_WAVES, 2) void sfcw_wgrad_kernel(const WArgs g_byval) {
  KERNARG_IN_PLACE(WArgs);
  int chunk, y;
  if (!order_xy(g.ord, blockIdx.x, chunk, y)) return;
  if (g.only_type >= 0 && y != g.only_type) return;
  const WType T = g.type[y];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave >= T.nsl) return;  // (a terminated wave no longer counts at the workgroup's barriers)
  const int ebeg = chunk * g.echunk, eend = min(g.E, ebeg + g.echunk);
  if (ebeg >= eend) return;
  const WSlab& S = g.slab[T.slab0 + wave];
  const int d3 = g.deg[T.deg].d3;
#define W_CASE(A, B, C) w_wave<A, B, C, MODE>(g, T, S, wave, T.nsl, ebeg, eend)
#define W_D3(A)                                  \
  switch (d3) {                                  \
    case 1:                                      \
      if (T.ct == 3) W_CASE(A, 1, 3);            \
      else if (T.ct == 2) W_CASE(A, 1, 2);       \
      else W_CASE(A, 1, 1);                      \
      break;                                     \
    case 3:                                      \
      if (T.ct == 2) W_CASE(A, 3, 2);            \
      else W_CASE(A, 3, 1);                      \
      break;                                     \
    default: W_CASE(A, 5, 1); break;             \
  }
  switch (S.d1) {
    case 1: W_D3(1); break;
    case 3: W_D3(3); break;
    default: W_D3(5); break;
  }
#undef W_D3
#undef W_CASE
}

#ifndef EQF_W_ROUNDS
#define EQF_W_ROUNDS 3
#endif
static int g_w_rounds = 0;  // development (sfcw_dev_set 0): rounds of resident workgroups the chunk length is sized for
static int g_w_only = -1;  // development (sfcw_dev_set 2): only the workgroups of this type (index after the cost sort) run
static int g_w_order = 2;   // development (sfcw_dev_set 1): 0 heaviest-first in batches of 8 chunks per XCD, 1 chunk-major (XCD-aware), 2 type-major, heaviest first

int plan_wgrad2(const SfcCommon& C, const eqf_dtp_paths* P, const XGate* gate, WArgs& A, int& nblk, size_t& lds, int npa) {
  if (!fits32(C) || max_deg(C) > 5) return EQF_E_UNSUPPORTED;
  if ((C.x_ld | C.w_ld | C.ld1 | C.ld2) & 3) return EQF_E_UNSUPPORTED;
  if (gate && gate->on && !P) return EQF_E_UNSUPPORTED;
  memset(&A, 0, sizeof A);
  A.x = C.x, A.coupling = C.coupling, A.w = C.w;
  A.x_ld = C.x_ld, A.m_ld = C.m_ld, A.w_ld = C.w_ld, A.E = C.E;
  A.d1 = C.o1, A.d2 = C.o2, A.ld1 = C.ld1, A.ld2 = C.ld2;
  if (gate) A.gate = *gate;
  int ntype = 0, lenmax = 1, ntmax = 1;
  for (int d = 0; d < C.ndeg; ++d) {
    const SfcDeg& D = C.deg[d];
    if (!D.dW) return EQF_E_BADARG;
    if (D.d3 != 1 && D.d3 != 3 && D.d3 != 5) return EQF_E_UNSUPPORTED;
    if (D.nslab < 2) return EQF_E_UNSUPPORTED;  // (a lone slab would own all d_out tiles of its half-step: the one-wave kernel)
    A.deg[d].dW = D.dW, A.deg[d].dW2 = D.dW2, A.deg[d].d3 = D.d3, A.deg[d].N1 = D.N1, A.deg[d].N2 = D.N2;
    A.deg[d].out1_off = D.out1_off;
    const int ctm = x_ctmax(D.d3), cttot = D.Ncat / 32;
    const int ngc = eqf_cdiv(cttot, ctm), cps = eqf_cdiv(cttot, ngc);
    if (D.d3 * cps > ntmax) ntmax = D.d3 * cps;
    for (int q = 0; q < D.nslab; ++q) {
      const SfcSlab& S = C.slab[D.slab0 + q];
      WSlab& T = A.slab[D.slab0 + q];
      if (S.d1 != 1 && S.d1 != 3 && S.d1 != 5) return EQF_E_UNSUPPORTED;
      T.x_off = S.x_off, T.w_off = S.w_off, T.m_off = S.m_off, T.x_mul = S.x_mul, T.d1 = S.d1, T.sid = (short)q;
      T.g_off = -2;
      if (A.gate.on) {  // S.x_off = segment offset + 32 c in the gate's OUTPUT row: find the segment, map it, keep the chunk
        int seg_off = -1;
        for (int pp = 0; pp < P->npaths; ++pp)
          if (P->in_off[pp] <= S.x_off && S.x_off < P->in_off[pp] + P->mul[pp] && P->mul[pp] == S.x_mul && 2 * P->l1[pp] + 1 == S.d1)
            seg_off = P->in_off[pp];
        if (seg_off < 0) return EQF_E_BADARG;
        int raw_off = 0, g_off = -2;
        const int grc = gate_map(A.gate, P, seg_off, S.x_mul, S.d1, raw_off, g_off);
        if (grc) return grc;
        const int cch = S.x_off - seg_off;
        T.x_off = raw_off + cch, T.g_off = g_off >= 0 ? g_off + cch : g_off;
      }
      if (S.d1 * D.d3 > lenmax) lenmax = S.d1 * D.d3;
    }
    // slab groups of as even a size as possible, <= W_WAVES each
    const int ngs = eqf_cdiv(D.nslab, W_WAVES), gsz = D.nslab / ngs, grem = D.nslab % ngs;
    int s0 = 0;
    for (int gi = 0; gi < ngs; ++gi) {
      const int nsl = gsz + (gi < grem ? 1 : 0);
      for (int k = 0; k < ngc; ++k) {
        const int c0 = k * cps, cn = (cttot - c0 < cps) ? cttot - c0 : cps;
        if (cn <= 0) continue;
        if (ntype >= W_MAXTYPE) return EQF_E_UNSUPPORTED;
        if (eqf_cdiv(D.d3 * cn, nsl) > (D.d3 * cn + 2) / 3) return EQF_E_UNSUPPORTED;  // (more d_out tiles per wave than w_wave has slots)
        WType& Y = A.type[ntype++];
        Y.deg = (short)d, Y.slab0 = (short)(D.slab0 + s0), Y.nsl = (short)nsl, Y.ct0 = (short)c0, Y.ct = (short)cn;
      }
      s0 += nsl;
    }
  }
  if (ntype == 0) return EQF_E_BADARG;
  A.ntype = ntype;
  A.plane_floats = 2 * (ntmax + 1) * npa * W_PLANE_FLOATS;  // (+ 1: the spare slot of waves without a tile)
  A.wave_floats = 32 * ((lenmax + 3) & ~3) + 4 * XT16_FLOATS;
  lds = (size_t)(A.plane_floats + W_WAVES * A.wave_floats) * sizeof(float);
  if (lds > 80 * 1024) return EQF_E_UNSUPPORTED;  // two workgroups per CU
  // Workgroup types differ ~8x in cost per edge (slabs of input degree 2 into output degree 1 against scalar slabs): with one
  // round of equal chunks the launch lasted as long as its heaviest type and the average wave lived 43 % of it (rocprofv3 --pmc
  // SQ_WAVE_CYCLES).  Several rounds of shorter chunks, heaviest type first inside batches of 8 chunks per XCD (order_xy mode 4):
  // the light types fill the slots the heavy ones free.
  {
    long cost[W_MAXTYPE];
    for (int y = 0; y < ntype; ++y) {
      const WType& Y = A.type[y];
      const int d3 = A.deg[Y.deg].d3;
      int d1m = 1;
      for (int q = 0; q < Y.nsl; ++q) d1m = A.slab[Y.slab0 + q].d1 > d1m ? A.slab[Y.slab0 + q].d1 : d1m;
      cost[y] = (long)d3 * (8 * d1m + 30 + 35 * Y.ct) + 14 * (d1m + 1) + 50 * eqf_cdiv(d3 * Y.ct, Y.nsl);
    }
    for (int a = 1; a < ntype; ++a)  // stable insertion sort, descending
      for (int b = a; b > 0 && cost[b] > cost[b - 1]; --b) {
        const WType ty = A.type[b];
        A.type[b] = A.type[b - 1], A.type[b - 1] = ty;
        const long tc = cost[b];
        cost[b] = cost[b - 1], cost[b - 1] = tc;
      }
  }
  // (three rounds at the bench size: 152 / 129 / 124 / 133 us for 1 / 2 / 3 / 4; two on graphs of 10-17 k edges: 66 / 60 / 79 us at
  // 10 000, 106 / 89 / 101 at 17 000 -- profiles/r06/r06_ag_*)
  const int rounds = g_w_rounds > 0 ? g_w_rounds : (C.E >= 22000 ? EQF_W_ROUNDS : 2);
  int z = eqf_cdiv(rounds * 512, ntype);
  int echunk = eqf_cdiv(C.E, z);
  echunk = ((echunk + 31) / 32) * 32;
  if (echunk < 64) echunk = 64;
  A.echunk = echunk;
  z = eqf_cdiv(C.E, echunk);
  if (g_w_order == 1) A.ord = xcd_order(z, ntype, nblk);
  else A.ord = lpt_order(z, ntype, nblk, g_w_order != 2);
  return 0;
}

}  // namespace

void sfcw_dev_set(int key, int value) {
  if (key == 0) g_w_rounds = value;
  if (key == 1) g_w_order = value;
  if (key == 2) g_w_only = value;
}

// Text dump of the launch plan for the CPU tests (eqf_sfcx_dev_plan kind 3): header, one line per workgroup type
int sfcw_dev_plan(const sfc::SfcCommon* Cp, const eqf_dtp_paths* paths, int mode, char* buf, int buflen) {
  static thread_local WArgs A;
  int nblk = 0, n = 0;
  size_t lds = 0;
  const int npa = mode == 0 ? 2 : (mode == 1 ? 1 : 3);
  const int rc = plan_wgrad2(*Cp, paths, nullptr, A, nblk, lds, npa);
  if (rc) return rc;
#define PUT(...)                                                          \
  do {                                                                    \
    n += snprintf(buf + n, n < buflen ? buflen - n : 0, __VA_ARGS__);     \
    if (n >= buflen) return EQF_E_UNSUPPORTED;                            \
  } while (0)
  PUT("wgrad2 nblk %d lds %d echunk %d nx %d ny %d per_xcd %d mode %d\n", nblk, (int)lds, A.echunk, A.ord.nx, A.ord.ny, A.ord.per_xcd,
      A.ord.mode);
  for (int y = 0; y < A.ntype; ++y) {
    const WType& Y = A.type[y];
    PUT("type %d %d %d %d %d %d", Y.deg, A.deg[Y.deg].d3, Y.slab0, Y.nsl, Y.ct0, Y.ct);
    for (int q = 0; q < Y.nsl; ++q) PUT(" %d:%d", A.slab[Y.slab0 + q].sid, A.slab[Y.slab0 + q].d1);
    PUT("\n");
  }
#undef PUT
  return n;
}

// Launch of the multi-wave weight gradient for the operator described by C (built by sfcx.hip's entry point); returns
// EQF_E_UNSUPPORTED when the shape is outside this kernel's tables (the caller then runs the one-wave kernel).
int sfcw_wgrad_launch(const sfc::SfcCommon* Cp, const eqf_dtp_paths* paths, int mode, int gate_on, int gS, int gG, float c_silu,
                      float c_sig, float* d_bias0, float* d_bias2, void* stream) {
  const SfcCommon& C = *Cp;
  XGate XG;
  memset(&XG, 0, sizeof XG);
  XG.on = gate_on, XG.S = gS, XG.G = gG, XG.c_silu = c_silu, XG.c_sig = c_sig;
  static thread_local WArgs A;
  int nblk = 0;
  size_t lds = 0;
  const int npa = mode == 0 ? 2 : (mode == 1 ? 1 : 3);
  int rc = plan_wgrad2(C, paths, &XG, A, nblk, lds, npa);
  if (rc) return rc;
  A.db = d_bias0, A.db2 = d_bias2;
  A.only_type = g_w_only;
  hipStream_t st = (hipStream_t)stream;
#define WG_LAUNCH(M)                                                                                                 \
  do {                                                                                                               \
    static bool big_lds = false; /* (dynamic LDS beyond 64 KB has to be allowed once per kernel) */                  \
    if (!big_lds) {                                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sfcw_wgrad_kernel<M>),                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)                  \
        return EQF_E_UNSUPPORTED;                                                                                    \
      big_lds = true;                                                                                                \
    }                                                                                                                \
    hipLaunchKernelGGL((sfcw_wgrad_kernel<M>), dim3(nblk), dim3(64 * W_WAVES), lds, st, A);                          \
  } while (0)
  if (mode == 0) WG_LAUNCH(0);
  else if (mode == 1) WG_LAUNCH(1);
  else WG_LAUNCH(2);
#undef WG_LAUNCH
  return 0;
}
