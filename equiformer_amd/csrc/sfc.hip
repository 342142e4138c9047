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
#include <cstdio>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

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
  const float* W;  // [K, Ncat] row-major
  float* dW;       // weight-gradient target (same shape), accumulated
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

// ------------------------------------------------------------------------------------------------ forward
constexpr int F_TE = 64;     // edges per tile
constexpr int F_NP = 8;      // edges per generating thread
constexpr int F_MAXT = 4;    // 32x32 accumulator tiles per wave
constexpr int F_MAXCT = 6;   // column tiles per workgroup

struct SfcFwdArgs {
  SfcCommon c;
  const float* bias;  // [Ncat of degree 0] or null
  int nsplit[SFC_MAX_DEG], cps[SFC_MAX_DEG], blk0[SFC_MAX_DEG + 1];
};

template <int MAXD>
__global__ __launch_bounds__(256, 2) void sfc_fwd_kernel(const SfcFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int b = blockIdx.x, di = 0;
  while (di + 1 < g.c.ndeg && b >= g.blk0[di + 1]) ++di;
  b -= g.blk0[di];
  const SfcDeg& D = g.c.deg[di];
  const int nsplit = g.nsplit[di];
  const int tile = b / nsplit, ns = b - tile * nsplit;
  const int e0 = tile * F_TE;
  const int ecnt = min(F_TE, g.c.E - e0);
  const int d3 = D.d3;
  const int rows = F_TE * d3, RT = rows >> 5;
  const int ncol0 = ns * g.cps[di];
  const int ncols = min(g.cps[di], D.Ncat - ncol0);
  const int CT = ncols >> 5;
  const int SA = rows + 1, SB = ncols + 4;
  float* __restrict__ As = smem;
  float* __restrict__ Bs = As + 32 * SA;  // 32*SA floats: a multiple of 128 bytes
  float* __restrict__ Mt = Bs + 32 * SB;
  const int m_len = D.m_len;

  const int t = threadIdx.x;
  const int u = t & 31, grp = t >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, r = lane & 31, hi = lane >> 5;

  // tiles of this wave: tt = wave + 4 i  ->  (rt = tt % RT, ct = tt / RT); all of this is wave-uniform
  const int ntile = RT * CT;
  const int ntw = (ntile - wave + 3) >> 2;  // number of tiles of this wave
  int aoff[F_MAXT], boff[F_MAXT];
  f32x16 acc[F_MAXT];
#pragma unroll
  for (int i = 0; i < F_MAXT; ++i) {
    const int tt = (i < ntw) ? wave + 4 * i : wave;
    const int ct = tt / RT, rt = tt - ct * RT;
    aoff[i] = rt * 32;
    boff[i] = ct * 32;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  }

  // coupling tile: Mt[el][j] = coupling[e0+el, m_base + j]
  for (int i = t; i < ecnt * m_len; i += 256) {
    const int el = i / m_len, j = i - el * m_len;
    Mt[i] = g.c.coupling[(long)(e0 + el) * g.c.m_ld + D.m_base + j];
  }

  float xv[F_NP][MAXD], wv[F_NP];
  float4 bv[F_MAXCT];
  int s_d1 = 0, s_mo = 0;  // of the slab whose inputs are in xv / wv / bv
  auto issue = [&](int s) {
    const SfcSlab S = g.c.slab[D.slab0 + s];
    s_d1 = S.d1;
    s_mo = S.m_off - D.m_base;
#pragma unroll
    for (int p = 0; p < F_NP; ++p) {
      const int el = grp + 8 * p;
      wv[p] = 0.f;
#pragma unroll
      for (int i = 0; i < MAXD; ++i) xv[p][i] = 0.f;
      if (el < ecnt) {
        const long e = e0 + el;
        wv[p] = g.c.w ? g.c.w[e * g.c.w_ld + S.w_off + u] : 1.0f;
        const float* xp = g.c.x + e * g.c.x_ld + S.x_off + u;
#pragma unroll
        for (int i = 0; i < MAXD; ++i)
          if (i < S.d1) xv[p][i] = xp[i * S.x_mul];
      }
    }
    const float* wp = D.W + (long)(s * 32 + (t >> 3)) * D.Ncat + ncol0 + 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < F_MAXCT; ++j) {
      bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < CT) bv[j] = *reinterpret_cast<const float4*>(wp + 32 * j);
    }
  };
  auto commit = [&]() {
    // rows of edges beyond the graph get zeros (xv = wv = 0), so As is always fully defined
#pragma unroll
    for (int p = 0; p < F_NP; ++p) {
      const int el = grp + 8 * p;
      const float* mp = Mt + (el < ecnt ? el : 0) * m_len + s_mo;
      float* q = As + u * SA + el;
      for (int m3 = 0; m3 < d3; ++m3) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < MAXD; ++i)
          if (i < s_d1) a = fmaf(mp[i * d3 + m3], xv[p][i], a);
        q[m3 * F_TE] = a * wv[p];
      }
    }
    float* bp = Bs + (t >> 3) * SB + 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < F_MAXCT; ++j)
      if (j < CT) *reinterpret_cast<float4*>(bp + 32 * j) = bv[j];
  };

  issue(0);
  __syncthreads();
  const int nslab = D.nslab;
  for (int s = 0; s < nslab; ++s) {
    commit();
    __syncthreads();
    if (s + 1 < nslab) issue(s + 1);
#pragma unroll 2
    for (int kk = 0; kk < 32; kk += 2) {
      const float* ap = As + (kk + hi) * SA + r;
      const float* bp = Bs + (kk + hi) * SB + r;
      float av[F_MAXT], bw[F_MAXT];
#pragma unroll
      for (int i = 0; i < F_MAXT; ++i) {
        av[i] = ap[aoff[i]];
        bw[i] = bp[boff[i]];
      }
#pragma unroll
      for (int i = 0; i < F_MAXT; ++i)
        if (i < ntw) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bw[i], acc[i], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: row = m3 * 64 + el ; column c of the concatenated output
#pragma unroll
  for (int i = 0; i < F_MAXT; ++i) {
    if (i >= ntw) continue;
    const int c = ncol0 + boff[i] + r;
    const float bvl = (g.bias && D.l3 == 0) ? g.bias[c] : 0.f;
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
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct SfcWgArgs {
  SfcCommon c;
  int nitem;
  int echunk;  // edges per workgroup (multiple of 8)
  short item_slab[2 * SFC_MAX_SLABS];  // global slab index
  short item_col0[2 * SFC_MAX_SLABS];  // first column (multiple of 32) of the item's column range
  short item_ct[2 * SFC_MAX_SLABS];    // number of 32-column tiles
};

constexpr int W_SUB = 32;  // edges whose coupling matrices a wave keeps in LDS at a time

template <int CTT, int MAXD>
__global__ __launch_bounds__(256) void sfc_wgrad_kernel(const SfcWgArgs g) {
  __shared__ float Msh[4][W_SUB * MAXD * MAXD];  // per wave: [edge][i*d3 + m3] of the slab's path
  const int item = blockIdx.y;
  const SfcSlab S = g.c.slab[g.item_slab[item]];
  const SfcDeg& D = g.c.deg[S.deg];
  const int col0 = g.item_col0[item], CT = g.item_ct[item];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int d1 = S.d1, d3 = D.d3, len = d1 * d3;
  float* __restrict__ Mw = Msh[wave];
  // edge range of this wave (even start)
  const int eb = blockIdx.x * g.echunk;
  const int per = g.echunk >> 2;
  const int ebeg = eb + wave * per;
  const int eend = min(g.c.E, ebeg + per);

  f32x16 acc[CTT];
#pragma unroll
  for (int i = 0; i < CTT; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  const int N1 = D.N1;
  if (ebeg >= eend) return;
  // Flattened loop over (edge pair, m3).  The raw inputs of step it+1 (d_out values and -- when a new edge pair
  // starts -- x and w) are requested BEFORE the MFMAs of step it are issued, so their latency hides behind the matrix
  // pipe; the A value of step it+1 is formed after those MFMAs.  Coupling matrices come from a wave-private LDS
  // block refreshed every W_SUB edges (lane-uniform reads instead of d1*d3 vector loads per step).
  float xv[MAXD], wv = 0.f;          // current edge pair
  float xn[MAXD], wn = 0.f;          // next edge pair (prefetched)
  float bn[CTT];                     // d_out values of the next step
  float a_cur = 0.f, b_cur[CTT];
  int sub0 = ebeg;                   // first edge of the LDS-resident coupling block
  auto stage_m = [&](int s0) {
    __builtin_amdgcn_wave_barrier();
    for (int el = 0; el < W_SUB; ++el) {
      const int e = s0 + el;
      if (e < eend && lane < len) Mw[el * len + lane] = g.c.coupling[(long)e * g.c.m_ld + S.m_off + lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto load_edge = [&](int e) {
    const int ee = e + hi;
    const bool valid = ee < eend;
#pragma unroll
    for (int i = 0; i < MAXD; ++i) xn[i] = 0.f;
    wn = 0.f;
    if (valid) {
      const float* xp = g.c.x + (long)ee * g.c.x_ld + S.x_off + r;
#pragma unroll
      for (int i = 0; i < MAXD; ++i)
        if (i < d1) xn[i] = xp[i * S.x_mul];
      wn = g.c.w ? g.c.w[(long)ee * g.c.w_ld + S.w_off + r] : 1.0f;
    }
  };
  auto take_edge = [&]() {
#pragma unroll
    for (int i = 0; i < MAXD; ++i) xv[i] = xn[i];
    wv = wn;
  };
  auto load_step = [&](int e, int m3) {
    const int ee = e + hi;
    const bool valid = ee < eend;
    const long er = valid ? ee : e;
    const float* o1p = g.c.o1 + er * g.c.ld1 + D.out1_off + m3 * N1;
    const float* o2p = g.c.o2 ? g.c.o2 + er * g.c.ld2 : nullptr;
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) {
      bn[ct] = 0.f;
      if (ct < CT && valid) {
        const int c = col0 + ct * 32 + r;
        bn[ct] = (c < N1) ? o1p[c] : o2p[c - N1];
      }
    }
  };
  auto form = [&](int e, int m3) {
    const int ee = e + hi;
    const float* mp = Mw + (ee < eend ? ee - sub0 : 0) * len + m3;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d1) a = fmaf(mp[i * d3], xv[i], a);
    a_cur = a * wv;  // wv = 0 for the padding lane half
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) b_cur[ct] = bn[ct];
  };
  int e = ebeg, m3 = 0;
  stage_m(sub0);
  load_edge(e);
  take_edge();
  load_step(e, 0);
  form(e, 0);
  while (true) {
    int en = e, mnext = m3 + 1;
    if (mnext == d3) mnext = 0, en = e + 2;
    const bool more = en < eend;
    if (more) {
      if (mnext == 0) load_edge(en);
      load_step(en, mnext);
    }
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct)
      if (ct < CT) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[ct], acc[ct], 0, 0, 0);
    if (!more) break;
    if (mnext == 0) {
      take_edge();
      if (en >= sub0 + W_SUB) {
        sub0 = en;
        stage_m(sub0);
      }
    }
    form(en, mnext);
    e = en, m3 = mnext;
  }
  // C[i = channel of the slab][j = column]
  const int slab_in_deg = g.item_slab[item] - D.slab0;
#pragma unroll
  for (int ct = 0; ct < CTT; ++ct)
    if (ct < CT) {
      const int c = col0 + ct * 32 + r;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ch = slab_in_deg * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi;
        atomicAdd(D.dW + (long)ch * D.Ncat + c, acc[ct][q]);
      }
    }
}

// ------------------------------------------------------------------------------------------------ data gradient
constexpr int B_TE = 32;
constexpr int B_KC = 256;         // columns of d_out staged per chunk
constexpr int B_MAXGRP = 12;
constexpr int B_MAXPATH = 12;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SfcBPath {
  short deg;    // index into deg[]
  short krow;   // first row of the slab in W (channel index inside the degree)
  short w_off;  // offset of the slab's weights in the w row
  short m_off;  // offset of the path's coupling matrix in the coupling row
  short mt_off; // offset inside the per-edge LDS coupling block of the group
  short pad;
};
struct SfcBGroup {
  int x_off;   // offset of (segment l1, chunk) in the x row
  short x_mul, d1, npath, mt_len;
  SfcBPath p[B_MAXPATH];
};
struct SfcBwdArgs {
  SfcCommon c;
  float* dx;
  float* dw;  // may be null
  float* dM;  // may be null; ACCUMULATED with atomics
  int ngrp;
  int dt_floats;  // LDS partition
  SfcBGroup grp[B_MAXGRP];
};

// Workgroup = (32 edges, 32-channel chunk of one input segment); wave = (16-edge half eh, 16-channel half chh).
// d_mid tile of a path for (16 edges x 16 channels x all m3) = d_out tile (LDS, [k][m3*32+el]) x W_slab^T with
// v_mfma_f32_16x16x4_f32; its accumulator layout (lane = channel, 4 edges per lane) is exactly what the DTP backward
// contraction wants, so the epilogue runs in registers: no exchange of d_mid between waves at all.
template <int MAXD>
__global__ __launch_bounds__(256, 2) void sfc_bwd_kernel(const SfcBwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* __restrict__ Dt = smem;                // [k][32*d3 + 4]
  float* __restrict__ Mt = smem + g.dt_floats;  // [32][mt_len]
  const SfcBGroup& G = g.grp[blockIdx.y];
  const int e0 = blockIdx.x * B_TE;
  const int ecnt = min(B_TE, g.c.E - e0);
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63, j = lane & 15, kg = lane >> 4;
  const int eh = wave & 1, chh = wave >> 1;
  const int ch = 16 * chh + j;         // channel inside the chunk
  const int el0 = 16 * eh + 4 * kg;    // first of this lane's 4 edges
  const int d1 = G.d1, mt_len = G.mt_len;

  float xv[4][MAXD], gx[4][MAXD];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int el = el0 + q;
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
      xv[q][i] = (el < ecnt && i < d1) ? g.c.x[(long)(e0 + el) * g.c.x_ld + G.x_off + i * G.x_mul + ch] : 0.f;
      gx[q][i] = 0.f;
    }
  }
  // coupling matrices of the group's paths
  for (int pi = 0; pi < G.npath; ++pi) {
    const SfcBPath P = G.p[pi];
    const int len = d1 * g.c.deg[P.deg].d3;
    for (int i = t; i < B_TE * len; i += 256) {
      const int el = i / len, jj = i - el * len;
      Mt[el * mt_len + P.mt_off + jj] = (el < ecnt) ? g.c.coupling[(long)(e0 + el) * g.c.m_ld + P.m_off + jj] : 0.f;
    }
  }

  int staged_deg = -1;
  for (int pi = 0; pi < G.npath; ++pi) {
    const SfcBPath P = G.p[pi];
    const SfcDeg& D = g.c.deg[P.deg];
    const int d3 = D.d3, Ncat = D.Ncat, N1 = D.N1;
    const int rows = B_TE * d3, SD = rows + 4;
    float wv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      wv[q] = (el0 + q < ecnt && g.c.w) ? g.c.w[(long)(e0 + el0 + q) * g.c.w_ld + P.w_off + ch] : 1.0f;
    f32x4 acc[2][MAXD];
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[0][i][q] = 0.f, acc[1][i][q] = 0.f;

    const int nchunk = (Ncat + B_KC - 1) / B_KC;
    for (int ck = 0; ck < nchunk; ++ck) {
      const int kc0 = ck * B_KC, kcn = min(B_KC, Ncat - kc0);
      if (!(nchunk == 1 && staged_deg == P.deg)) {
        __syncthreads();  // readers of the previous Dt contents are done
        // stage Dt[k][row] = d_out[e0 + el, m3, kc0 + k],  row = m3*32 + el   (16 float4 columns x 4 rows per wave step)
        const int c4 = lane & 15, rr = lane >> 4;
        for (int cb = 0; cb < kcn; cb += 64) {
          const int c = cb + 4 * c4;
          for (int rb = 0; rb < rows; rb += 16) {
            const int row = rb + wave * 4 + rr;
            if (c < kcn && row < rows) {
              const int m3 = row >> 5, el = row & 31;
              float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
              if (el < ecnt) {
                const int cg = kc0 + c;
                const float* src = (cg < N1) ? g.c.o1 + (long)(e0 + el) * g.c.ld1 + D.out1_off + m3 * N1 + cg
                                             : g.c.o2 + (long)(e0 + el) * g.c.ld2 + (cg - N1);
                v = *reinterpret_cast<const float4*>(src);
              }
              float* q = Dt + c * SD + row;
              q[0] = v.x, q[SD] = v.y, q[2 * SD] = v.z, q[3 * SD] = v.w;
            }
          }
        }
        staged_deg = P.deg;
        __syncthreads();
      }
      // MFMA k mapping: lane group kg supplies k = kb + 4 kg + jj for the jj-th instruction of a 16-wide k block
      const float* wrow = D.W + (long)(P.krow + ch) * Ncat + kc0 + 4 * kg;
      const float* abase = Dt + (4 * kg) * SD + 16 * eh + j;
      for (int kb = 0; kb < kcn; kb += 16) {
        const float4 b4 = *reinterpret_cast<const float4*>(wrow + kb);
        const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
        const float* ap = abase + kb * SD;
        float av[4][MAXD];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int rt = 0; rt < MAXD; ++rt) av[jj][rt] = (rt < d3) ? ap[jj * SD + rt * 32] : 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int rt = 0; rt < MAXD; ++rt)
            if (rt < d3)
              acc[jj & 1][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj][rt], bj[jj], acc[jj & 1][rt], 0, 0, 0);
      }
    }
    // DTP backward contraction in registers: this lane holds d_mid[m3][edge el0+q][channel ch]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int el = el0 + q;
      float gw = 0.f;
      const float* mp = Mt + el * mt_len + P.mt_off;
#pragma unroll
      for (int m3 = 0; m3 < MAXD; ++m3)
        if (m3 < d3) {
          const float dm = acc[0][m3][q] + acc[1][m3][q];
          float tm = 0.f;
#pragma unroll
          for (int i = 0; i < MAXD; ++i)
            if (i < d1) {
              const float m = mp[i * d3 + m3];
              tm = fmaf(m, xv[q][i], tm);
              gx[q][i] = fmaf(m * wv[q], dm, gx[q][i]);
              if (g.dM) {
                float v = wv[q] * xv[q][i] * dm;  // sum over the 16 channels of this wave (lanes j)
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                if (j == 0 && el < ecnt) atomicAdd(g.dM + (long)(e0 + el) * g.c.m_ld + P.m_off + i * d3 + m3, v);
              }
            }
          gw = fmaf(dm, tm, gw);
        }
      if (g.dw && el < ecnt) g.dw[(long)(e0 + el) * g.c.w_ld + P.w_off + ch] = gw;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int el = el0 + q;
    if (el < ecnt) {
#pragma unroll
      for (int i = 0; i < MAXD; ++i)
        if (i < d1) g.dx[(long)(e0 + el) * g.c.x_ld + G.x_off + i * G.x_mul + ch] = gx[q][i];
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
// Fill the degree / slab tables.  o1_irreps: one segment per output degree; n2 extra scalar columns on degree 0.
int build_common(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* P,
                 const float* const* Wl, float* const* dWl, float* o1, const eqf_irreps* o1_irreps, float* o2, int n2,
                 int E, SfcCommon& C) {
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
    if (D.l3 > 3 || D.Ncat % 32 != 0 || D.N1 % 4 != 0) return EQF_E_UNSUPPORTED;
    D.W = Wl ? Wl[D.l3] : nullptr;
    D.dW = dWl ? dWl[D.l3] : nullptr;
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

int max_d1(const SfcCommon& C) {
  int m = 1;
  for (int d = 0; d < C.ndeg; ++d)
    for (int q = 0; q < C.deg[d].nslab; ++q) m = C.slab[C.deg[d].slab0 + q].d1 > m ? C.slab[C.deg[d].slab0 + q].d1 : m;
  return m;
}

double sfc_flops(const SfcCommon& C) {
  double f = 0;
  for (int d = 0; d < C.ndeg; ++d) f += 2.0 * C.E * C.deg[d].d3 * (double)C.deg[d].K * C.deg[d].Ncat;
  return f;
}
double sfc_bytes(const SfcCommon& C) {  // x, w, coupling in; out rows out; weights
  double b = 4.0 * C.E * ((double)C.x_ld + (C.w ? C.w_ld : 0) + C.m_ld + C.ld1 + C.ld2);
  for (int d = 0; d < C.ndeg; ++d) b += 4.0 * C.deg[d].K * C.deg[d].Ncat;
  return b;
}

}  // namespace

extern "C" {

int eqf_sfc_fwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                const float* const* Wl, const float* bias0, float* out1, const eqf_irreps* out1_irreps, float* out2,
                int n2, int E, void* stream) {
  if (!Wl) return EQF_E_BADARG;
  SfcFwdArgs A;
  int rc = build_common(x, coupling, w, paths, Wl, nullptr, out1, out1_irreps, out2, n2, E, A.c);
  if (rc) return rc;
  if (E <= 0) return 0;
  A.bias = bias0;
  const int ntile = eqf_cdiv(E, F_TE);
  size_t lds = 0;
  int blk = 0;
  // heaviest degrees first (their workgroups take longest)
  for (int d = 0; d < A.c.ndeg; ++d) {
    const SfcDeg& D = A.c.deg[d];
    if (!D.W) return EQF_E_BADARG;
    const int RT = F_TE * D.d3 / 32;
    int maxct = (4 * F_MAXT) / RT;  // tiles per workgroup <= 4 waves x F_MAXT
    if (maxct > F_MAXCT) maxct = F_MAXCT;
    if (maxct < 1) return EQF_E_UNSUPPORTED;
    const int cttot = D.Ncat / 32;
    A.nsplit[d] = eqf_cdiv(cttot, maxct);
    A.cps[d] = eqf_cdiv(cttot, A.nsplit[d]) * 32;
    A.blk0[d] = blk;
    blk += ntile * A.nsplit[d];
    const size_t need = sizeof(float) * (32 * (F_TE * D.d3 + 1) + 4 + 32 * (A.cps[d] + 4) + (size_t)F_TE * D.m_len);
    if (need > lds) lds = need;
  }
  A.blk0[A.c.ndeg] = blk;
  if (lds > SFC_LDS_LIMIT) return EQF_E_UNSUPPORTED;
  const int md = max_d1(A.c);
  hipStream_t st = (hipStream_t)stream;
  const int pid = eqf_prof_begin("sfc_fwd", st, sfc_flops(A.c), sfc_bytes(A.c));
  if (md <= 5) {
    static bool attr5 = false;
    if (!attr5) {
      hipFuncSetAttribute((const void*)sfc_fwd_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, SFC_LDS_LIMIT);
      attr5 = true;
    }
    hipLaunchKernelGGL(sfc_fwd_kernel<5>, dim3(blk), dim3(256), lds, st, A);
  } else {
    static bool attr7 = false;
    if (!attr7) {
      hipFuncSetAttribute((const void*)sfc_fwd_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, SFC_LDS_LIMIT);
      attr7 = true;
    }
    hipLaunchKernelGGL(sfc_fwd_kernel<7>, dim3(blk), dim3(256), lds, st, A);
  }
  eqf_prof_end(pid, st);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_sfc_bwd_weight(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                       const float* d_out1, const eqf_irreps* out1_irreps, const float* d_out2, int n2,
                       float* const* dWl, int E, void* stream) {
  if (!dWl) return EQF_E_BADARG;
  SfcWgArgs A;
  int rc = build_common(x, coupling, w, paths, nullptr, dWl, const_cast<float*>(d_out1), out1_irreps,
                        const_cast<float*>(d_out2), n2, E, A.c);
  if (rc) return rc;
  if (E <= 0) return 0;
  const int md = max_d1(A.c);
  hipStream_t st = (hipStream_t)stream;
  // three launches at most, by accumulator width (column tiles per item): <= 2, <= 4, <= 12
  const int caps[3] = {2, 4, 12};
  for (int d = 0; d < A.c.ndeg; ++d)
    if (!A.c.deg[d].dW) return EQF_E_BADARG;
  const int pid = eqf_prof_begin("sfc_wgrad", st, sfc_flops(A.c), sfc_bytes(A.c));
  for (int cls = 0; cls < 3; ++cls) {
    A.nitem = 0;
    for (int d = 0; d < A.c.ndeg; ++d) {
      const SfcDeg& D = A.c.deg[d];
      const int cttot = D.Ncat / 32;
      const int c = cttot <= caps[0] ? 0 : (cttot <= caps[1] ? 1 : 2);  // wide degrees are split into <= 12-tile items
      if (c != cls) continue;
      const int nsp = eqf_cdiv(cttot, caps[cls]);
      const int per = eqf_cdiv(cttot, nsp);
      for (int q = 0; q < D.nslab; ++q)
        for (int s = 0; s < nsp; ++s) {
          if (A.nitem >= 2 * SFC_MAX_SLABS) return EQF_E_UNSUPPORTED;
          const int ct0 = s * per, ctn = (cttot - ct0 < per) ? cttot - ct0 : per;
          A.item_slab[A.nitem] = (short)(D.slab0 + q);
          A.item_col0[A.nitem] = (short)(ct0 * 32);
          A.item_ct[A.nitem] = (short)ctn;
          A.nitem++;
        }
    }
    if (A.nitem == 0) continue;
    int z = eqf_cdiv(1024, A.nitem);
    int echunk = eqf_cdiv(E, z);
    echunk = ((echunk + 7) / 8) * 8;
    if (echunk < 64) echunk = 64;
    A.echunk = echunk;
    z = eqf_cdiv(E, echunk);
    dim3 grid(z, A.nitem);
#define LAUNCH_WG(CTT)                                                                                     \
  do {                                                                                                     \
    if (md <= 5)                                                                                           \
      hipLaunchKernelGGL((sfc_wgrad_kernel<CTT, 5>), grid, dim3(256), 0, st, A);                           \
    else                                                                                                   \
      hipLaunchKernelGGL((sfc_wgrad_kernel<CTT, 7>), grid, dim3(256), 0, st, A);                           \
  } while (0)
    if (cls == 0)
      LAUNCH_WG(2);
    else if (cls == 1)
      LAUNCH_WG(4);
    else
      LAUNCH_WG(12);
#undef LAUNCH_WG
    EQF_CHECK_LAUNCH();
  }
  eqf_prof_end(pid, st);
  return 0;
}

int eqf_sfc_bwd_data(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                     const float* const* Wl, const float* d_out1, const eqf_irreps* out1_irreps, const float* d_out2,
                     int n2, float* dx, float* dw, float* d_coupling, int E, void* stream) {
  if (!Wl || !dx) return EQF_E_BADARG;
  SfcBwdArgs A;
  int rc = build_common(x, coupling, w, paths, Wl, nullptr, const_cast<float*>(d_out1), out1_irreps,
                        const_cast<float*>(d_out2), n2, E, A.c);
  if (rc) return rc;
  if (E <= 0) return 0;
  A.dx = dx, A.dw = (w ? dw : nullptr), A.dM = d_coupling;
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
  for (int s = 0; s < nseg; ++s)
    for (int c = 0; c < seg_mul[s]; c += 32) {
      if (A.ngrp >= B_MAXGRP) return EQF_E_UNSUPPORTED;
      SfcBGroup& G = A.grp[A.ngrp];
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
  if (A.ngrp == 0) return EQF_E_BADARG;
  A.dt_floats = (int)((dtmax + 3) & ~(size_t)3);
  const size_t lds = sizeof(float) * ((size_t)A.dt_floats + (size_t)B_TE * mtmax);
  if (lds > SFC_LDS_LIMIT) return EQF_E_UNSUPPORTED;
  const int md = max_d1(A.c) > d3max ? max_d1(A.c) : d3max;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(eqf_cdiv(E, B_TE), A.ngrp);
  const int pid = eqf_prof_begin("sfc_bwd_data", st, sfc_flops(A.c), sfc_bytes(A.c));
  if (md <= 5) {
    static bool attr5 = false;
    if (!attr5) {
      hipFuncSetAttribute((const void*)sfc_bwd_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, SFC_LDS_LIMIT);
      attr5 = true;
    }
    hipLaunchKernelGGL(sfc_bwd_kernel<5>, grid, dim3(256), lds, st, A);
  } else {
    static bool attr7 = false;
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
