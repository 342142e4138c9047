// Data gradient of the fused SeparableFCTP on the bf16 matrix cores, second generation: ONE WORKGROUP OF FOUR WAVES PER 32-EDGE
// TILE with the tile's d_out rows split into bf16 planes ONCE, in MFMA B-fragment order, in LDS.
//
//   d_mid^T[channel, edge] (per path, m3) = sum_n W[krow + channel, n] * d_out[edge, l3, m3, n]          (matrix cores)
//   dx[e, l1, i, u] = sum_{paths of (l1, u)} sum_m3 M_p[e][i, m3] * w[e, p, u] * d_mid[e, (p, u), m3]        (registers)
//   dw[e, p, u]     = sum_m3 d_mid[e, (p, u), m3] * sum_i M_p[e][i, m3] * x[e, l1, i, u]
//
// [ref: backward of SeparableFCTP.forward, nets/graph_attention_transformer.py:234-248 (autograd through e3nn's 'uvu' tensor
//  product and the per-degree linear, tensor_product_rescale.py:125-136)]
//
// Why a second kernel (profiles/r03): the first one (sfcx.hip: one wave per (tile, 32-channel slab)) re-reads
// and re-splits the same d_out rows once per slab (7 x for the QM9 trunk: 630 KB of a tile's 1.7 MB through the address unit,
// a third of its VALU instructions), and every HBM-latency load inside an item (d_out tiles, one coupling block and one w tile
// per path) sits in the same in-order return queue as the weight-fragment stream, which it stalls for a full miss each time.
// Here, per tile:
//   prologue   the four waves fetch the whole coupling rows of the 32 edges (-> LDS, fp32) and the d_out rows (row-major
//              128-byte lines), split each value into its bf16 planes and store them in B-fragment order -- all loads of the
//              prologue are issued before the first is used (one miss latency for the tile);
//   items      the 32-channel input slabs are dealt to the waves through an LDS counter (heaviest first).  An item issues its
//              x tiles, its first w tile and one touching load per further w tile up front (one more miss latency; the w tiles
//              are then fetched one path ahead from L2), then runs path after path:
//              matrix loop = packed W fragments (A operand, L2 hits, register ring) x B fragments from LDS; the DTP backward
//              contraction in registers with the coupling matrix read from LDS; dw written per path, dx once.
// After the prologue nothing in an item's stream misses to HBM but its own x / w tiles.
// LDS: planes 2 KB per (m3, 16 columns) [QM9 sep_act: 88 KB] + coupling rows [17.5 KB] + one transposition tile per wave.
// Operators whose planes do not fit (L_max = 3 widths) stay on the first kernel (sfcx::bwd2_launch returns EQF_E_UNSUPPORTED).
// Compiled with -fno-slp-vectorize like sfcx.hip (no packed-FP32 VALU beside bf16 MFMAs).
#include "sfcx_common.h"

extern __shared__ __attribute__((aligned(16))) float sx2_lds[];

namespace {

constexpr int XB2_NW = 4;        // waves per workgroup
constexpr int XB2_MAXPT = 48;    // (degree, m3, pair of 16-column blocks) tiles of d_out per edge tile
constexpr int XB2_NS = 4;        // register stages of the W-fragment stream (one 16-column block each)
#ifndef XB2_MT
#define XB2_MT 8
#endif
#ifndef XB2_ONLY_MODE0
#define XB2_ONLY_MODE0 0         // development: instantiate the split mode only (compile time)
#endif

// Dev build (-DEQF_XTRACE=1, tools/sfcx_trace2.py): non-serialising clock marks (tag << 56 | cycles) per wave of the first
// workgroups of one XCD.  Not compiled into the product.
#if EQF_XTRACE
#define XB2_MARK(tag)                                                                                       \
  do {                                                                                                      \
    if (trace_p && lane == 0 && trace_n < 63) trace_p[trace_n++] = ((unsigned long long)(tag) << 56) | (xt_mark() & 0xffffffffffffffull); \
  } while (0)
#else
#define XB2_MARK(tag) \
  do {                \
  } while (0)
#endif

struct XB2PTile {
  int deg, m3, np, frag;  // frag: index of the first of the pair's two B fragments (ints: read with scalar loads)
};
// Argument tables of this kernel: every field an int or a pointer, so that all table reads are SCALAR loads (s_load has no
// 8- / 16-bit form: a short or char field is fetched with a vector load, its consumers become vector code, and each such
// fetch in the W stream drained the whole in-order load queue with s_waitcnt vmcnt(0) -- first version of this kernel)
constexpr int XB2_MAXPATH = 8;   // paths of one input slab (L_max = 2: at most 6)
struct XB2Path {
  int deg;     // index into deg[]
  int krow;    // first row of the slab in W_l3
  int w_off;   // offset of the slab's weights in the w row
  int m_off;   // offset of the path's matrix in the coupling row
};
struct XB2Group {
  int x_off, mul, d1, npath;
  XB2Path p[XB2_MAXPATH];
};
struct XB2Base {
  const float *x, *coupling, *w;
  int x_ld, m_ld, w_ld, E;
  const float *d1, *d2;
  int ld1, ld2;
  float *dx, *dw, *dM;
  const __bf16* packed;
  struct Deg {
    int d3, N1, Ncat, out1_off, nt, pad;
    long pb;
  } deg[SFC_MAX_DEG];
  XB2Group grp[XB_MAXGRP];
};
struct XB2Args {
  XB2Base b;
  XB2PTile pt[XB2_MAXPT];
  int npt;
  int mc_ld;                      // row stride of the staged coupling rows (odd)
  int frag0[SFC_MAX_DEG];         // first B fragment of the degree: fragment (d, m3, nt) = frag0[d] + m3 * nt_d + nt
  int off_mc, off_tt, off_ctr;    // LDS offsets (floats) of the coupling rows, the transposition tiles, the item counter
  int ngrp;
  int order[XB_MAXGRP];           // groups, heaviest first
#if EQF_XTRACE
  unsigned long long* trace2;
#endif
};

template <int NP>
__device__ __forceinline__ void split4(const f32x4 v, __bf16 (&p)[NP][4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float r = v[j];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const __bf16 h = (__bf16)r;
      p[q][j] = h;
      if (q + 1 < NP) r = r - (float)h;
    }
  }
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// row-major registers -> wave-private LDS tile -> fragment registers (multi-wave workgroup: compiler-only ordering, the
// LDS instructions of one wave execute in order)
__device__ __forceinline__ void tile_to_frag_w(float* __restrict__ T, const f32x4 (&t)[4], float (&v)[16], const int lane) {
  const int c = lane & 7, rr = lane >> 3, r = lane & 31, hi = lane >> 5;
  wave_lds_order();
#pragma unroll
  for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(T + ((rr + 8 * it) * XT_LD + 4 * c)) = t[it];
  wave_lds_order();
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(T + (r * XT_LD + 8 * g4 + 4 * hi));
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * g4 + j] = u[j];
  }
}
__device__ __forceinline__ void tile_store_w(float* __restrict__ T, const float (&v)[16], float* __restrict__ base,
                                             const unsigned ld, const int e0, const int E, const int lane) {
  const int c = lane & 7, rr = lane >> 3, r = lane & 31, hi = lane >> 5;
  wave_lds_order();
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
    *reinterpret_cast<f32x4*>(T + (r * XT_LD + 8 * g4 + 4 * hi)) = f32x4{v[4 * g4], v[4 * g4 + 1], v[4 * g4 + 2], v[4 * g4 + 3]};
  wave_lds_order();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = rr + 8 * it;
    const f32x4 u = *reinterpret_cast<const f32x4*>(T + (row * XT_LD + 4 * c));
    if (e0 + row < E) *reinterpret_cast<f32x4*>(base + ((unsigned)(e0 + row) * ld + 4 * c)) = u;
  }
}

// The stream of packed W fragments of an item: path after path, 16-column block after block, every path padded to a multiple
// of XB2_NS blocks (the padding re-reads the path's last block and is not multiplied), so that the ring slot of a block is a
// compile-time constant and the loads run ahead ACROSS path boundaries: the last XB2_NS blocks of a path refill the ring with
// the first blocks of the NEXT path.  All of the stream's bookkeeping is scalar arithmetic on the loop counters (a first
// version kept a stream object with its own position: the compiler put it in vector registers and drained the load queue
// with s_waitcnt vmcnt(0) at every path switch).
template <int NPW>
struct WStage {
  bf16x8 aw[NPW];
};
// packed W fragments of one 16-column block (all planes) of a slab: base = first fragment of the slab's rows (uniform)
template <int NPW>
__device__ __forceinline__ void wload(WStage<NPW>& s, const __bf16* __restrict__ pa, const int nt, const int lane8) {
#pragma unroll
  for (int pl = 0; pl < NPW; ++pl) s.aw[pl] = *reinterpret_cast<const bf16x8*>(pa + ((unsigned)(nt * NPW + pl) * 512 + lane8));
}

// one 32-lane load that touches every 128-byte line of a 32 rows x 32 floats tile (lane = row): brings the tile into L2 ahead of
// its real, row-major fetch
__device__ __forceinline__ float tile_touch(const float* __restrict__ base, const unsigned ld, const int e0, const int elast,
                                            const int lane) {
  const unsigned e = min(e0 + (lane & 31), elast);
  return base[e * ld + 4 * (lane >> 5)];
}

template <int D1, int MODE, bool DM>
__device__ __forceinline__ void xb2_item(const XB2Args& g2, const XB2Group& G, const int tile, const __bf16* __restrict__ Bp,
                                         const float* __restrict__ Mc, float* __restrict__ Tt, const int lane
#if EQF_XTRACE
                                         ,
                                         unsigned long long* trace_p, int& trace_n
#endif
) {
  constexpr int NPA = Planes<MODE>::A, NPW = Planes<MODE>::W;
  const XB2Base& g = g2.b;
  XB2_MARK(10 + D1);
  const int r = lane & 31, hi = lane >> 5;
  const int e0 = tile * 32;
  const int elast = g.E - 1;
  const bool valid = e0 + r < g.E;
  const unsigned er = valid ? e0 + r : elast;
  const int mul = G.mul;
  const int npath = G.npath;

  // Everything of the item that misses to HBM goes out NOW, in one window: the x tiles (real loads), the first path's w tile,
  // and one touching load per further w tile -- the per-path fetches that follow (one path ahead) then hit in L2 and no longer
  // stall the in-order return queue they share with the W-fragment stream.
  f32x4 xt[D1][4];
#pragma unroll
  for (int i = 0; i < D1; ++i) tile_fetch(xt[i], g.x + G.x_off + i * mul, g.x_ld, e0, elast, lane);
  f32x4 wt[4];
  float touch = 0.f;
  if (g.w) {
    tile_fetch(wt, g.w + G.p[0].w_off, g.w_ld, e0, elast, lane);
#pragma unroll 1
    for (int p = 1; p < npath; ++p) touch += tile_touch(g.w + G.p[p].w_off, g.w_ld, e0, elast, lane);
  }
  // the W-fragment stream starts behind them
  const int lane8 = lane * 8;
  WStage<NPW> ring[XB2_NS];
  {
    const XB2Path P0 = G.p[0];
    const XB2Base::Deg& D0 = g.deg[P0.deg];
    const __bf16* const pa0 = g.packed + D0.pb + ((size_t)(P0.krow >> 5) * D0.nt) * NPW * 512;
#pragma unroll
    for (int k = 0; k < XB2_NS; ++k) wload<NPW>(ring[k], pa0, k < D0.nt ? k : D0.nt - 1, lane8);
  }

  float xv[D1][16], gx[D1][16];
#pragma unroll
  for (int i = 0; i < D1; ++i) {
    tile_to_frag_w(Tt, xt[i], xv[i], lane);
#pragma unroll
    for (int q = 0; q < 16; ++q) gx[i][q] = 0.f;
  }
  XB2_MARK(20);
  if (touch == 12345.678f) gx[0][0] = touch;  // keeps the touching loads (never true for finite data of this magnitude; harmless if so)
  const float* const Mrow = Mc + r * g2.mc_ld;

  auto path = [&](auto tag, const XB2Path P, const int pi) __attribute__((always_inline)) {
    constexpr int D3 = decltype(tag)::value;
    const XB2Base::Deg& D = g.deg[P.deg];
    const int NT = D.nt, NTpad = (NT + XB2_NS - 1) / XB2_NS * XB2_NS;
    const __bf16* const bfrag = Bp + (size_t)g2.frag0[P.deg] * NPA * 512 + lane * 8;
    const __bf16* const pa_c = g.packed + D.pb + ((size_t)(P.krow >> 5) * D.nt) * NPW * 512;
    // the path after this one (the last path "continues" with itself: four surplus loads per item)
    const XB2Path Pn = G.p[pi + 1 < npath ? pi + 1 : pi];
    const XB2Base::Deg& Dn = g.deg[Pn.deg];
    const __bf16* const pa_n = g.packed + Dn.pb + ((size_t)(Pn.krow >> 5) * Dn.nt) * NPW * 512;
    const int NTn = Dn.nt;
    // this path's w tile arrived during the previous path: into fragment layout now, so that its registers can take the
    // next path's tile
    float wv[16], gw[16];
    if (g.w) {
      tile_to_frag_w(Tt, wt, wv, lane);
      const int pn = pi + 1 < npath ? pi + 1 : pi;
      tile_fetch(wt, g.w + G.p[pn].w_off, g.w_ld, e0, elast, lane);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) wv[q] = 1.f;
    }
    // degree-2 slabs (2 x 80 registers of x / dx) take a degree-2 output in two chunks of m3: the second chunk re-reads the
    // path's few W blocks (two for 32 output channels) outside the ring, which by then holds the next path's fragments
    constexpr int MCH = (D1 >= 5 && D3 >= 5) ? 3 : D3, NCH = (D3 + MCH - 1) / MCH;
    static_assert(NCH <= 2, "at most two chunks");
    auto chunk = [&](auto chtag) __attribute__((always_inline)) {
      constexpr int CH = decltype(chtag)::value, M0 = CH * MCH, MC = (D3 - M0 < MCH) ? D3 - M0 : MCH;
      f32x16 acc[MC];
#pragma unroll
      for (int m3 = 0; m3 < MC; ++m3)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[m3][q] = 0.f;
      if constexpr (CH == 0) {
#pragma unroll 1
        for (int blk = 0; blk < NTpad; blk += XB2_NS) {
          // refill target of this block's slots: XB2_NS blocks ahead in this path, or the head of the next one
          const bool last = blk + XB2_NS >= NTpad;
          const __bf16* const pa_r = last ? pa_n : pa_c;
          const int lim = (last ? NTn : NT) - 1, a0 = last ? 0 : blk + XB2_NS;
#pragma unroll
          for (int k = 0; k < XB2_NS; ++k) {
            const int nt = blk + k;
            if (nt < NT) {  // uniform
              bf16x8 pb[MC][NPA];
#pragma unroll
              for (int m3 = 0; m3 < MC; ++m3)
#pragma unroll
                for (int pl = 0; pl < NPA; ++pl)
                  pb[m3][pl] = *reinterpret_cast<const bf16x8*>(bfrag + ((unsigned)(((M0 + m3) * NT + nt) * NPA + pl) * 512));
#pragma unroll
              for (int m3 = 0; m3 < MC; ++m3) mma_terms<NPW, NPA>(ring[k].aw, pb[m3], acc[m3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            wload<NPW>(ring[k], pa_r, a0 + k < lim ? a0 + k : lim, lane8);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        const __bf16* const pa = g.packed + D.pb + ((size_t)(P.krow >> 5) * D.nt) * NPW * 512 + lane * 8;
#pragma unroll 1
        for (int nt = 0; nt < NT; ++nt) {
          bf16x8 aw[NPW];
#pragma unroll
          for (int pl = 0; pl < NPW; ++pl) aw[pl] = *reinterpret_cast<const bf16x8*>(pa + (unsigned)(nt * NPW + pl) * 512);
#pragma unroll
          for (int m3 = 0; m3 < MC; ++m3) {
            bf16x8 pb[NPA];
#pragma unroll
            for (int pl = 0; pl < NPA; ++pl)
              pb[pl] = *reinterpret_cast<const bf16x8*>(bfrag + ((unsigned)(((M0 + m3) * NT + nt) * NPA + pl) * 512));
            mma_terms<NPW, NPA>(aw, pb, acc[m3]);
          }
        }
      }
      XB2_MARK(31);
      if constexpr (CH == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) gw[q] = 0.f;
      }
      // DTP backward contraction in registers: acc[m3][q] = d_mid[edge r][channel ch(q)][M0 + m3]
      float dMa[DM ? D1 * MC : 1];
      if constexpr (DM) {
#pragma unroll
        for (int k = 0; k < D1 * MC; ++k) dMa[k] = 0.f;
      }
#pragma unroll
      for (int m3 = 0; m3 < MC; ++m3) {
        float mv[D1];
#pragma unroll
        for (int i = 0; i < D1; ++i) mv[i] = Mrow[P.m_off + i * D3 + M0 + m3];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float dm = acc[m3][q];
          const float dmw = dm * wv[q];
          float tm = 0.f;
#pragma unroll
          for (int i = 0; i < D1; ++i) {
            tm = fmaf(mv[i], xv[i][q], tm);
            gx[i][q] = fmaf(mv[i], dmw, gx[i][q]);
            if constexpr (DM) dMa[i * MC + m3] = fmaf(dmw, xv[i][q], dMa[i * MC + m3]);
          }
          gw[q] = fmaf(dm, tm, gw[q]);
        }
      }
      if constexpr (DM) {
#pragma unroll
        for (int i = 0; i < D1; ++i)
#pragma unroll
          for (int m3 = 0; m3 < MC; ++m3) {
            const float v = dMa[i * MC + m3] + __shfl_xor(dMa[i * MC + m3], 32);  // the two channel halves of the same edge
            if (hi == 0 && valid) atomicAdd(g.dM + ((size_t)er * g.m_ld + P.m_off + i * D3 + M0 + m3), v);
          }
      }
    };
    chunk(IC<0>());
    if constexpr (NCH > 1) chunk(IC<1>());
    if (g.dw && g.w) tile_store_w(Tt, gw, g.dw + P.w_off, g.w_ld, e0, g.E, lane);
    XB2_MARK(32);
  };

#pragma unroll 1
  for (int pi = 0; pi < npath; ++pi) {
    const XB2Path P = G.p[pi];
    switch (g.deg[P.deg].d3) {
      case 1: path(IC<1>(), P, pi); break;
      case 3: path(IC<3>(), P, pi); break;
      default: path(IC<5>(), P, pi); break;
    }
  }

#pragma unroll
  for (int i = 0; i < D1; ++i) tile_store_w(Tt, gx[i], g.dx + G.x_off + i * mul, g.x_ld, e0, g.E, lane);
  XB2_MARK(40);
}

template <int MODE, bool DM>
__global__ __launch_bounds__(64 * XB2_NW, 1) void sfcx_bwd2_kernel(const XB2Args g_byval) {
  KERNARG_IN_PLACE(XB2Args);
  constexpr int NPA = Planes<MODE>::A;
  const XB2Base& b = g.b;
  const int tile = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // (uniform for the compiler too)
  const int e0 = tile * 32, elast = b.E - 1;
  __bf16* const Bp = reinterpret_cast<__bf16*>(sx2_lds);
  float* const Mc = sx2_lds + g.off_mc;
  float* const Tt = sx2_lds + g.off_tt + wave * XT_FLOATS;
  int* const ctr = reinterpret_cast<int*>(sx2_lds + g.off_ctr);
#if EQF_XTRACE
  unsigned long long* trace_p = nullptr;
  int trace_n = 0;
  if (g.trace2 && (blockIdx.x & 7) == 3 && (blockIdx.x >> 3) < 32) trace_p = g.trace2 + ((size_t)(blockIdx.x >> 3) * XB2_NW + wave) * 64;
#define XB2_TRACE_ARGS , trace_p, trace_n
#else
#define XB2_TRACE_ARGS
#endif
  XB2_MARK(1);

  // ---- prologue: every load first (coupling rows of this wave's 8 edges, its share of the d_out tiles), conversions after
  {
    constexpr int MJ = 4;  // coupling rows up to 256 floats
    float mv[MJ][8];
    const int m_ld = b.m_ld;
#pragma unroll
    for (int jj = 0; jj < MJ; ++jj) {
      const int j = jj * 64 + lane;
      const int jc = j < m_ld ? j : 0;
      if (jj * 64 < m_ld) {  // uniform
#pragma unroll
        for (int q = 0; q < 8; ++q) mv[jj][q] = b.coupling[(size_t)min(e0 + wave * 8 + q, elast) * m_ld + jc];
      }
    }
#pragma unroll
    for (int jj = 0; jj < MJ; ++jj) {
      const int j = jj * 64 + lane;
      if (jj * 64 < m_ld && j < m_ld) {
#pragma unroll
        for (int q = 0; q < 8; ++q) Mc[(wave * 8 + q) * g.mc_ld + j] = mv[jj][q];
      }
    }
    XB2_MARK(3);
    constexpr int MT = XB2_MT;  // d_out tiles of a wave in flight at once (4 x f32x4 each); one batch for QM9 / OC20 widths
    const int c = lane & 7, rr = lane >> 3;
    const int h = c >> 2, kh = (c >> 1) & 1, j0 = 4 * (c & 1);
#pragma unroll 1
    for (int base = 0; base < g.npt; base += MT * XB2_NW) {
      f32x4 t[MT][4];
#pragma unroll
      for (int k = 0; k < MT; ++k) {
        const int ti = base + wave + XB2_NW * k;
        if (ti < g.npt) {  // uniform
          const XB2PTile T = g.pt[ti];
          const XB2Base::Deg& D = b.deg[T.deg];
          const int n0 = 32 * T.np;
          const bool main = n0 < D.N1;  // N1 % 32 == 0; the second consumer exists on degree 0 only
          const float* const sb = main ? b.d1 + D.out1_off + (n0 + T.m3 * D.N1) : b.d2 + (n0 - D.N1);
          tile_fetch(t[k], sb, main ? b.ld1 : b.ld2, e0, elast, lane);
        }
      }
#pragma unroll
      for (int k = 0; k < MT; ++k) {
        const int ti = base + wave + XB2_NW * k;
        if (ti < g.npt) {
          const int frag = g.pt[ti].frag + h;
          __bf16* const dst = Bp + (size_t)frag * NPA * 512 + (32 * kh) * 8 + j0;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            __bf16 p[NPA][4];
            split4<NPA>(t[k][it], p);
            const int row = rr + 8 * it;
#pragma unroll
            for (int pl = 0; pl < NPA; ++pl)
              *reinterpret_cast<bf16x4*>(dst + pl * 512 + row * 8) = bf16x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
          }
        }
      }
    }
    if (threadIdx.x == 0) *ctr = 0;
  }
  XB2_MARK(4);
  __syncthreads();
  XB2_MARK(5);

  // ---- items: the input slabs, heaviest first, dealt through the LDS counter
  for (;;) {
    int gi = 0;
    if (lane == 0) gi = atomicAdd(ctr, 1);
    gi = __builtin_amdgcn_readfirstlane(gi);
    if (gi >= g.ngrp) break;
    const XB2Group& G = b.grp[g.order[gi]];
#ifdef XB2_DBG_ONLY_D1
    xb2_item<XB2_DBG_ONLY_D1, MODE, DM>(g, G, tile, Bp, Mc, Tt, lane XB2_TRACE_ARGS);
#else
    switch (G.d1) {
      case 1: xb2_item<1, MODE, DM>(g, G, tile, Bp, Mc, Tt, lane XB2_TRACE_ARGS); break;
      case 3: xb2_item<3, MODE, DM>(g, G, tile, Bp, Mc, Tt, lane XB2_TRACE_ARGS); break;
      default: xb2_item<5, MODE, DM>(g, G, tile, Bp, Mc, Tt, lane XB2_TRACE_ARGS); break;
    }
#endif
  }
  XB2_MARK(50);
}

template <int MODE, bool DM>
int launch_one(const XB2Args& A, int ntile, size_t lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sfcx_bwd2_kernel<MODE, DM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((sfcx_bwd2_kernel<MODE, DM>), dim3(ntile), dim3(64 * XB2_NW), lds, st, A);
  return 0;
}

}  // namespace

#if EQF_XTRACE
static unsigned long long* g_xtrace2 = nullptr;
extern "C" int eqf_sfcx_dev_set_trace2(void* p) {  // dev build only: 32 workgroups x 4 waves x 64 u64 marks, or NULL
  g_xtrace2 = (unsigned long long*)p;
  return 0;
}
#endif

namespace sfcx {

// Plans and launches the multi-wave data gradient; EQF_E_UNSUPPORTED (nothing launched) when the operator does not fit it:
// degrees above 2, planes + coupling rows beyond the LDS budget.
int bwd2_launch(const sfc::SfcCommon& C, const eqf_dtp_paths* P, int mode, float* dx, float* dw, float* dM, const void* packed,
                bool plan_only, void* stream) {
  if (max_deg(C) > 5 || C.m_ld > 256) return EQF_E_UNSUPPORTED;
  static thread_local XB2Args A;
  static thread_local XBwdArgs V;  // the one-wave kernel's plan: same groups and paths, copied into the all-int tables
  int nblk = 0, ngrp = 0;
  size_t lds1 = 0;
  int rc = plan_bwd(C, P, mode, V, nblk, lds1, ngrp);
  if (rc) return rc;
  memset(&A, 0, sizeof A);
  A.b.x = V.x, A.b.coupling = V.coupling, A.b.w = V.w;
  A.b.x_ld = V.x_ld, A.b.m_ld = V.m_ld, A.b.w_ld = V.w_ld, A.b.E = V.E;
  A.b.d1 = V.d1, A.b.d2 = V.d2, A.b.ld1 = V.ld1, A.b.ld2 = V.ld2;
  for (int d = 0; d < C.ndeg; ++d) {
    A.b.deg[d].d3 = V.deg[d].d3, A.b.deg[d].N1 = V.deg[d].N1, A.b.deg[d].Ncat = V.deg[d].Ncat;
    A.b.deg[d].out1_off = V.deg[d].out1_off, A.b.deg[d].nt = V.deg[d].nt, A.b.deg[d].pb = V.deg[d].pb;
  }
  for (int gi = 0; gi < ngrp; ++gi) {
    const XBGroup& G = V.grp[gi];
    if (G.npath > XB2_MAXPATH) return EQF_E_UNSUPPORTED;
    XB2Group& H = A.b.grp[gi];
    H.x_off = G.x_off, H.mul = G.mul, H.d1 = G.d1, H.npath = G.npath;
    for (int q = 0; q < G.npath; ++q)
      H.p[q].deg = G.p[q].deg, H.p[q].krow = G.p[q].krow, H.p[q].w_off = G.p[q].w_off, H.p[q].m_off = G.p[q].m_off;
  }
  const int npa = mode == 1 ? 1 : (mode == 2 ? 3 : 2);
  int nfrag = 0, npt = 0;
  for (int d = 0; d < C.ndeg; ++d) {
    const XB2Base::Deg& D = A.b.deg[d];
    A.frag0[d] = nfrag;
    for (int m3 = 0; m3 < D.d3; ++m3)
      for (int np = 0; np < D.nt / 2; ++np) {
        if (npt >= XB2_MAXPT) return EQF_E_UNSUPPORTED;
        A.pt[npt].deg = d, A.pt[npt].m3 = m3, A.pt[npt].np = np;
        A.pt[npt].frag = nfrag + m3 * D.nt + 2 * np;
        ++npt;
      }
    nfrag += D.d3 * D.nt;
  }
  A.npt = npt;
  A.mc_ld = C.m_ld | 1;
  long cost[XB_MAXGRP];
  for (int gi = 0; gi < ngrp; ++gi) {
    const XB2Group& G = A.b.grp[gi];
    cost[gi] = 0;
    for (int q = 0; q < G.npath; ++q) cost[gi] += (long)A.b.deg[G.p[q].deg].d3 * A.b.deg[G.p[q].deg].nt;
    A.order[gi] = gi;
  }
  for (int i = 1; i < ngrp; ++i)  // insertion sort, heaviest first (stable)
    for (int j = i; j > 0 && cost[A.order[j]] > cost[A.order[j - 1]]; --j) {
      const int t = A.order[j];
      A.order[j] = A.order[j - 1], A.order[j - 1] = t;
    }
  A.ngrp = ngrp;
  size_t off = (size_t)nfrag * npa * 1024 / 4;  // floats
  A.off_mc = (int)off;
  off += ((size_t)32 * A.mc_ld + 3) & ~(size_t)3;
  A.off_tt = (int)off;
  off += (size_t)XB2_NW * XT_FLOATS;
  A.off_ctr = (int)off;
  off += 4;
  const size_t lds = off * sizeof(float);
  if (lds > 160 * 1024) return EQF_E_UNSUPPORTED;
  if (plan_only) return 0;
  A.b.dx = dx, A.b.dw = (C.w ? dw : nullptr), A.b.dM = dM;
  A.b.packed = (const __bf16*)packed;
#if EQF_XTRACE
  A.trace2 = g_xtrace2;
#endif
  const int ntile = eqf_cdiv(C.E, 32);
  hipStream_t st = (hipStream_t)stream;
  const bool dm = dM != nullptr;
#define XB2_LAUNCH(M)                                            \
  (dm ? launch_one<M, true>(A, ntile, lds, st) : launch_one<M, false>(A, ntile, lds, st))
#if XB2_ONLY_MODE0
  if (mode != 0) return EQF_E_UNSUPPORTED;
  rc = XB2_LAUNCH(0);
#else
  rc = mode == 0 ? XB2_LAUNCH(0) : (mode == 1 ? XB2_LAUNCH(1) : XB2_LAUNCH(2));
#endif
#undef XB2_LAUNCH
  return rc;
}

}  // namespace sfcx
