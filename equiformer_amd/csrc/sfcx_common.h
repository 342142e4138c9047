// Pieces of the split-precision SeparableFCTP kernels (sfcx.hip): plane splitting, packed-weight layout, row-major tile I/O, the
// argument tables of the data gradient and their planner.  (A second translation unit, the multi-wave data gradient of round 4,
// shared this header; it only ever reached parity with the one-wave kernel and was removed in round 5: DESIGN.md 3.1c.)
#pragma once
#include "common.h"
#include "prof.h"
#include "sfc_common.h"
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


namespace {
using namespace sfc;

template <int MODE>
struct Planes;
template <>
struct Planes<0> {
  static constexpr int A = 2, W = 3;
};
template <>
struct Planes<1> {
  static constexpr int A = 1, W = 1;
};
template <>
struct Planes<2> {
  static constexpr int A = 3, W = 3;
};
inline int mode_npw(int mode) { return mode == 1 ? 1 : 3; }

// x = p[0] + p[1] + ... (+ residual below 2^-(8 NP) |x|): each plane is the bf16 rounding of what is left.  The conversions go
// in PAIRS (v_cvt_pk_bf16_f32 takes two values; written per element hipcc converted every value alone and packed the planes
// with further conversions): 3 instead of 4 vector instructions per value and two planes, the same roundings bit for bit
// (round 6; the activation splits are the largest VALU item of the two gradient kernels)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void split_planes(const float (&v)[8], bf16x8 (&p)[NP]) {
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    float r0 = v[2 * jp], r1 = v[2 * jp + 1];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const bf16x2 h = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
      p[q][2 * jp] = h[0], p[q][2 * jp + 1] = h[1];
      if (q + 1 < NP) r0 -= (float)h[0], r1 -= (float)h[1];
    }
  }
}

// acc += sum over plane pairs (i, j) with i + j <= max(NA, NB) - 1, highest order (smallest terms) first
template <int NA, int NB>
__device__ __forceinline__ void mma_terms(const bf16x8 (&a)[NA], const bf16x8 (&b)[NB], f32x16& acc) {
  constexpr int TOP = (NA > NB ? NA : NB) - 1;
#pragma unroll
  for (int s = TOP; s >= 0; --s)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int j = s - i;
      if (j >= 0 && j < NB) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------- packed weight planes
// per degree:  Pf [K/16][Ncat/32][NPW][64 lanes][8]   B fragments of the forward    (lane: column n = 32 ct + (lane & 31),
//                                                      k = 16 kt + 8 (lane >> 5) + j)
//              Pb [K/32][Ncat/16][NPW][64 lanes][8]   A fragments of the data grad   (lane: row = 32 s + (lane & 31),
//                                                      n = 16 nt + 8 (lane >> 5) + j)
// of the concatenated [main | second consumer] weight [K, Ncat].  Offsets in bf16 elements.
struct PkDeg {
  long pf, pb;
};
inline long pack_layout(const SfcCommon& C, int npw, PkDeg (&pk)[SFC_MAX_DEG]) {
  long off = 0;
  for (int d = 0; d < C.ndeg; ++d) {
    const long n = (long)C.deg[d].K * C.deg[d].Ncat * npw;
    pk[d].pf = off, off += n;
    pk[d].pb = off, off += n;
  }
  return off;
}

// coupling block of 32 edges -> wave-private LDS: Mt[row * MS + j] = cp[(e0 + row) * m_ld + j], j < len.  Lane = (column
// r + 32 k, row parity): every load instruction covers two rows of up to 128 contiguous bytes; 16 loads in flight.
__device__ __forceinline__ void stage_m(float* __restrict__ Mt, const int MS, const float* __restrict__ cp,
                                        const unsigned m_ld, const int e0, const int elast, const int len,
                                        const int r, const int hi) {
  for (int k0 = 0; k0 < len; k0 += 32) {
    const int j = k0 + r;
    const bool jv = j < len;
    const int jc = jv ? j : 0;
    float v[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int e = min(e0 + 2 * p + hi, elast);
      v[p] = cp[(unsigned)e * m_ld + (unsigned)jc];  // 32-bit element offset (fits32): one base pointer, no 64-bit address pairs
    }
    if (jv) {
#pragma unroll
      for (int p = 0; p < 16; ++p) Mt[(2 * p + hi) * MS + j] = v[p];
    }
  }
}

// one wave per workgroup: LDS instructions of a wave execute in order, so write -> read of the wave's own tile needs no
// s_barrier, only that the compiler keeps the order
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// WAVE = true (workgroups whose waves run DIFFERENT work side by side -- the path-split data gradient of round 6): the tile is
// wave-private, ordering the wave's own LDS instructions is all the synchronisation it needs; false: the workgroup barrier.
template <bool WAVE>
__device__ __forceinline__ void tile_sync() {
  if constexpr (WAVE) wave_lds_order();
  else __syncthreads();
}
// A 32 rows x 32 floats tile between global memory and the layout of a 32x32 MFMA C fragment (lane (r, hi) holds, of row r,
// the four 16-byte runs at columns 8 g4 + 4 hi), by way of a wave-private LDS tile: every global instruction then moves whole
// 128-byte lines (8 rows per instruction) instead of 64 pieces of 32 bytes (a quarter of the address-unit work per byte;
// profiles/r03: the piecewise loads and stores were ~50 % of the data-gradient kernel).  XT_LD = 36 floats keeps the
// fragment-wise 16-byte LDS reads conflict free and the row-wise writes at most 2-way (tests/test_sfcx_tiles.py).
constexpr int XT_LD = 36;
constexpr int XT_FLOATS = 32 * XT_LD;
#ifndef EQF_XB_TILE_IO
#define EQF_XB_TILE_IO 1
#endif
// issue the four row-major loads of a tile (rows past elast re-read row elast)
__device__ __forceinline__ void tile_fetch(f32x4 (&t)[4], const float* __restrict__ base, const unsigned ld, const int e0,
                                           const int elast, const int lane) {
  const int c = lane & 7, rr = lane >> 3;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const unsigned e = min(e0 + rr + 8 * it, elast);
    t[it] = *reinterpret_cast<const f32x4*>(base + (e * ld + 4 * c));
  }
}
// row-major registers -> LDS tile -> fragment registers
template <bool WAVE = false>
__device__ __forceinline__ void tile_to_frag(float* __restrict__ T, const f32x4 (&t)[4], float (&v)[16], const int lane) {
  const int c = lane & 7, rr = lane >> 3, r = lane & 31, hi = lane >> 5;
  tile_sync<WAVE>();
#pragma unroll
  for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(T + ((rr + 8 * it) * XT_LD + 4 * c)) = t[it];
  tile_sync<WAVE>();
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(T + (r * XT_LD + 8 * g4 + 4 * hi));
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * g4 + j] = u[j];
  }
}
// fragment registers -> LDS tile -> row-major stores of the rows e0 + row < E
template <bool WAVE = false>
__device__ __forceinline__ void tile_store(float* __restrict__ T, const float (&v)[16], float* __restrict__ base,
                                           const unsigned ld, const int e0, const int E, const int lane) {
  const int c = lane & 7, rr = lane >> 3, r = lane & 31, hi = lane >> 5;
  tile_sync<WAVE>();
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
    *reinterpret_cast<f32x4*>(T + (r * XT_LD + 8 * g4 + 4 * hi)) = f32x4{v[4 * g4], v[4 * g4 + 1], v[4 * g4 + 2], v[4 * g4 + 3]};
  tile_sync<WAVE>();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = rr + 8 * it;
    const f32x4 u = *reinterpret_cast<const f32x4*>(T + (row * XT_LD + 4 * c));
    if (e0 + row < E) *reinterpret_cast<f32x4*>(base + ((unsigned)(e0 + row) * ld + 4 * c)) = u;
  }
}

// 16 rows x 32 floats, row-major in memory (rows e_lo .. e_lo + 15, clamped to elast) -> lane (r, hi) gets column r of the rows
// 8 hi .. 8 hi + 7: two 16-byte-per-lane loads of whole 128-byte lines and a wave-private LDS tile instead of eight 4-byte
// loads that each touch two lines (weight gradient: both MFMA operands are "lane = column, 8 edges per lane")
constexpr int XT16_FLOATS = 16 * XT_LD;
#ifndef EQF_XW_TILE
#define EQF_XW_TILE 1
#endif
struct Tile16 {
  f32x4 t0, t1;
};
__device__ __forceinline__ void tile16_fetch(Tile16& t, const float* __restrict__ base, const unsigned ld, const int e_lo,
                                             const int elast, const int lane) {
  const int c = lane & 7, rr = lane >> 3;
  t.t0 = *reinterpret_cast<const f32x4*>(base + ((unsigned)min(e_lo + rr, elast) * ld + 4 * c));
  t.t1 = *reinterpret_cast<const f32x4*>(base + ((unsigned)min(e_lo + rr + 8, elast) * ld + 4 * c));
}
__device__ __forceinline__ void tile16_put(float* __restrict__ T, const Tile16& t, const int lane) {
  const int c = lane & 7, rr = lane >> 3;
  *reinterpret_cast<f32x4*>(T + (rr * XT_LD + 4 * c)) = t.t0;
  *reinterpret_cast<f32x4*>(T + ((rr + 8) * XT_LD + 4 * c)) = t.t1;
}
__device__ __forceinline__ void tile16_get(const float* __restrict__ T, const int lane, float (&v)[8]) {
  const int r = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = T[(8 * hi + j) * XT_LD + r];
}

// read a by-value argument block in place (kernarg segment): with several instantiated bodies hipcc otherwise copies the
// struct to scratch and serves every dynamically indexed table lookup from there
#if defined(__HIP_DEVICE_COMPILE__)
#define KERNARG_IN_PLACE(T)                                           \
  const T& g = *(const T*)__builtin_amdgcn_kernarg_segment_ptr();     \
  (void)g_byval
#else
#define KERNARG_IN_PLACE(T) const T& g = g_byval
#endif

// Dev build (-DEQF_XTRACE=1, tools/sfcx_trace.py): serialising clock samples around the two phases of a forward step -- all
// operands arrived / matrix instructions retired -- for the first workgroups of one XCD.  Not compiled into the product.
#ifndef EQF_XTRACE
#define EQF_XTRACE 0
#endif
#if EQF_XTRACE
__device__ __forceinline__ unsigned long long xt_clock() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define XT_RETIRE(v) asm volatile("v_mov_b32 %0, %0" : "+v"(v))
// non-serialising sample (only the scalar result is waited for): phase marks of the data-gradient item
__device__ __forceinline__ unsigned long long xt_mark() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define XT_MARK()                                                    \
  do {                                                               \
    if (trace_on && lane == 0 && trace_n < 63) trace_p[++trace_n] = xt_mark(); \
  } while (0)
#else
#define XT_MARK() \
  do {            \
  } while (0)
#endif

template <int N>
struct IC {
  static constexpr int value = N;
};

inline SfcOrder xcd_order(int nx, int ny, int& nblocks) {
  SfcOrder o;
  o.mode = 1, o.nx = nx, o.ny = ny;
  o.per_xcd = (nx * ny + 7) / 8;
  nblocks = 8 * o.per_xcd;
  return o;
}

// Gated input rows (the consumer side of Gate, nets/fast_activation.py:132-148): the operator's input x is NOT materialised;
// the kernels read the gate's INPUT rows [scalars (S) | gates (G) | gated segments] and apply c_silu * silu to the scalar
// segment and c_sig * sigmoid(gate of the channel) to the l > 0 segments where they load x, and the data gradient writes the
// gradient of those raw rows (the gate's backward, complete inside a 32-channel slab: every gate scalar belongs to one slab).
struct XGate {
  int on, S, G;
  float c_silu, c_sig;
};
__device__ __forceinline__ float xg_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
// offsets of an input segment (offset in_off, multiplicity mul in the operator's input row = the gate's OUTPUT row) inside the
// raw row: raw_off = its data, g_off = its gate scalars (-1: the scalar segment, activated; -2: no gating)
inline int gate_map(const XGate& gt, const eqf_dtp_paths* P, int in_off, int mul, int d1, int& raw_off, int& g_off) {
  raw_off = in_off, g_off = -2;
  if (!gt.on) return 0;
  if (in_off == 0) {
    if (d1 != 1 || mul != gt.S) return EQF_E_UNSUPPORTED;
    g_off = -1;
    return 0;
  }
  if (d1 == 1 || in_off < gt.S) return EQF_E_UNSUPPORTED;  // a second scalar segment / E(3) rows: not gated layouts
  int before = 0, seen[EQF_MAX_SEG], ns = 0;
  for (int p = 0; p < P->npaths; ++p) {
    const int o = P->in_off[p];
    if (o <= 0 || o >= in_off) continue;
    bool dup = false;
    for (int k = 0; k < ns; ++k) dup |= seen[k] == o;
    if (dup) continue;
    if (ns >= EQF_MAX_SEG) return EQF_E_UNSUPPORTED;
    seen[ns++] = o;
    before += P->mul[p];
  }
  raw_off = in_off + gt.G;
  g_off = gt.S + before;
  return 0;
}

// items sorted by cost (heaviest first) by the planner; see order_xy mode 3
#ifndef EQF_X_LPT
#define EQF_X_LPT 1
#endif
inline SfcOrder lpt_order(int nx, int ny, int& nblocks, bool batched = false) {
  SfcOrder o;
  o.mode = batched ? 4 : 3, o.nx = nx, o.ny = ny;
  o.per_xcd = (nx + 7) / 8;
  if (batched) o.per_xcd = ((o.per_xcd + X_LPT_BATCH - 1) / X_LPT_BATCH) * X_LPT_BATCH;  // whole batches
  nblocks = 8 * o.per_xcd * ny;
  return o;
}

constexpr int XB_MAXGRP = 12;
constexpr int XB_MAXPATH = 12;
struct XBPath {
  short deg, mlen;  // index into deg[]; d1 * d3
  int krow;         // first row of the slab in W_l3
  int w_off;        // offset of the slab's weights in the w row
  int m_off;        // offset of the path's matrix in the coupling row
};
struct XBGroup {
  int x_off;  // offset of the slab (segment + 32 c) in the x row (the raw row when the input is gated)
  int g_off;  // gated input: offset of the slab's 32 gate scalars in the raw row (-1 scalar segment, -2 plain input)
  short mul, d1, npath, pad;
  XBPath p[XB_MAXPATH];
};
struct XBwdArgs {
  const float *x, *coupling, *w;
  int x_ld, m_ld, w_ld, E;
  const float *d1, *d2;
  int ld1, ld2;
  float *dx, *dw, *dM;
  const __bf16* packed;
  int ms;
  int only_d1;  // development switch: 0, or the only input degree (2 l + 1) whose items run
  int psplit;   // small graphs (L_max <= 2): two waves per item, each runs every other path (sfcx_bwd_kernel)
  XGate gate;
  SfcOrder ord;  // nx = edge tiles, ny = groups of this launch
  struct Deg {
    int d3, N1, Ncat, out1_off, nt;
    long pb;
  } deg[SFC_MAX_DEG];
  XBGroup grp[XB_MAXGRP];
#if EQF_XTRACE
  unsigned long long* trace;
#endif
};

float* const kDummyF = reinterpret_cast<float*>(16);  // non-null placeholder for tables built without data pointers

// lane offsets are 32-bit element indices: every per-edge tensor must stay below 2^31 elements
inline bool fits32(const SfcCommon& C) {
  const long E = C.E;
  const long big = (long)1 << 31;
  return E * C.x_ld < big && E * C.w_ld < big && E * C.m_ld < big && E * C.ld1 < big && E * (long)C.ld2 < big;
}

inline int max_deg(const SfcCommon& C) {
  int md = max_d1(C);
  for (int d = 0; d < C.ndeg; ++d) md = C.deg[d].d3 > md ? C.deg[d].d3 : md;
  return md;
}

// ------------------------------------------------------------------------------------------ forward: argument tables
constexpr int X_MAXSEG = 4;   // input segments (l1 <= 3)
constexpr int X_MAXP = 4;     // paths of one input segment into one output degree (the l2 values)

struct XPath {
  int w_off;    // first weight of the path in the w row
  int kbase;    // first row of the path in W_l3 (channel index inside the degree's DTP output)
  int m_rel;    // offset of the path's matrix inside the staged coupling block
};
struct XSeg {
  int x_off;  // offset of the segment in the x row (the raw row when the input is gated)
  int g_off;  // gated input: offset of the segment's gate scalars in the raw row (-1: scalar segment, -2: plain input)
  short mul, d1, npath, m_len;
  int m_off;  // offset of the block of the segment's matrices (all paths into this degree) in the coupling row
  XPath p[X_MAXP];
};
struct XFwdArgs {
  const float *x, *coupling, *w;
  int x_ld, m_ld, w_ld, E;
  float *o1, *o2;
  int ld1, ld2;
  const float *bias, *bias2;
  const __bf16* packed;
  int ms;  // row stride of the staged coupling block (odd)
  int wave_lds;  // floats of LDS per wave: its coupling block (up to four waves per workgroup on small graphs)
  int red_off;   // float offset of the partial-sum hand-over area behind the waves' regions (small graphs)
  XGate gate;
  SfcOrder ord;  // nx = edge tiles, ny = (degree, column group) items
  struct Deg {
    int d3, N1, Ncat, out1_off, cttot, nseg;
    long pf;
    XSeg seg[X_MAXSEG];
  } deg[SFC_MAX_DEG];
  signed char y_deg[16], y_ct0[16], y_ct[16];
#if EQF_XTRACE
  unsigned long long* trace;  // dev build: per-step clock samples of the first workgroups (tools/sfcx_trace.py)
#endif
};

// column tiles per wave item: 3 / 2 / 1 accumulator tiles per row tile (48 / 96 / 80+ accumulator registers); with 4 tiles
// for d3 == 1 the forward no longer fits 256 registers (2 waves per SIMD)
__host__ __device__ constexpr int x_ctmax(int d3) { return d3 == 1 ? 3 : (d3 == 3 ? 2 : 1); }

inline int plan_fwd(const SfcCommon& C, const eqf_dtp_paths* P, int mode, XFwdArgs& A, int& nblk, size_t& lds,
             const XGate* gate = nullptr) {
  if (!fits32(C)) return EQF_E_UNSUPPORTED;
  PkDeg pk[SFC_MAX_DEG];
  pack_layout(C, mode_npw(mode), pk);
  memset(&A, 0, sizeof A);
  A.x = C.x, A.coupling = C.coupling, A.w = C.w;
  A.x_ld = C.x_ld, A.m_ld = C.m_ld, A.w_ld = C.w_ld, A.E = C.E;
  A.o1 = C.o1, A.o2 = C.o2, A.ld1 = C.ld1, A.ld2 = C.ld2;
  if (gate) A.gate = *gate;
  int ny = 0, msmax = 1;
  for (int d = 0; d < C.ndeg; ++d) {
    const SfcDeg& D = C.deg[d];
    XFwdArgs::Deg& X = A.deg[d];
    X.d3 = D.d3, X.N1 = D.N1, X.Ncat = D.Ncat, X.out1_off = D.out1_off, X.cttot = D.Ncat / 32, X.pf = pk[d].pf;
    X.nseg = 0;
    // input segments in path (creation) order; the matrices of one segment's paths into this degree are contiguous
    for (int p = 0; p < P->npaths; ++p) {
      if (P->l3[p] != D.l3) continue;
      int si = -1;
      for (int s = 0; s < X.nseg; ++s)
        if (X.seg[s].x_off == P->in_off[p]) si = s;
      if (si < 0) {
        if (X.nseg >= X_MAXSEG) return EQF_E_UNSUPPORTED;
        si = X.nseg++;
        XSeg& S = X.seg[si];
        S.x_off = P->in_off[p], S.mul = (short)P->mul[p], S.d1 = (short)(2 * P->l1[p] + 1);
        S.npath = 0, S.m_off = P->m_off[p], S.m_len = 0;
        S.g_off = -2;
      }
      XSeg& S = X.seg[si];
      if (S.npath >= X_MAXP || P->mul[p] % 16 != 0) return EQF_E_UNSUPPORTED;
      if (P->m_off[p] != S.m_off + S.m_len) return EQF_E_UNSUPPORTED;  // not contiguous (never with layout.DtpTable)
      XPath& Q = S.p[S.npath++];
      Q.w_off = P->w_off[p], Q.kbase = P->out_ch[p], Q.m_rel = S.m_len;
      S.m_len = (short)(S.m_len + S.d1 * D.d3);
      if ((S.m_len | 1) > msmax) msmax = S.m_len | 1;
    }
    const int ctm = x_ctmax(D.d3);
    const int ng = eqf_cdiv(X.cttot, ctm), cps = eqf_cdiv(X.cttot, ng);
    for (int k = 0; k < ng; ++k) {
      if (ny >= 16) return EQF_E_UNSUPPORTED;
      const int c0 = k * cps, cn = (X.cttot - c0 < cps) ? X.cttot - c0 : cps;
      if (cn <= 0) continue;
      A.y_deg[ny] = (signed char)d, A.y_ct0[ny] = (signed char)c0, A.y_ct[ny] = (signed char)cn;
      ++ny;
    }
  }
  if (A.gate.on)  // input rows = the gate's input: data and gate-scalar offsets of every segment in the raw row
    for (int d = 0; d < C.ndeg; ++d)
      for (int si = 0; si < A.deg[d].nseg; ++si) {
        XSeg& S = A.deg[d].seg[si];
        int raw_off = 0, g_off = -2;
        const int grc = gate_map(A.gate, P, S.x_off, S.mul, S.d1, raw_off, g_off);
        if (grc) return grc;
        S.x_off = raw_off, S.g_off = g_off;
      }
  A.ms = msmax;
  lds = ((size_t)32 * msmax * sizeof(float) + 15) & ~(size_t)15;
  if (lds > 64 * 1024) return EQF_E_UNSUPPORTED;
  A.wave_lds = (int)(lds / sizeof(float));
  // Launch order (round 6, profiles/r06/r06_b_lpt_touch_ab.txt): the operator without per-edge weights (sep_value: 4 items per
  // tile, operands from L2) gains 25 % from heaviest-items-first (124 -> 93 us at E = 25 354); the one that streams w [E, 960]
  // (sep_act: 6 items per tile) loses 3 % -- its items no longer meet their tile's x / coupling rows in L2 -- and keeps the
  // tile-major order.
  if (EQF_X_LPT && C.w == nullptr) {
    // cost of an item = steps x (operand wait + generation + matrix instructions of a step), fitted to the traced steps of
    // profiles/r03/r03_s_what_bounds_the_forward.md
    long cost[16];
    for (int y = 0; y < ny; ++y) {
      const XFwdArgs::Deg& X = A.deg[A.y_deg[y]];
      long steps = 0;
      for (int si = 0; si < X.nseg; ++si) steps += (long)X.seg[si].npath * (X.seg[si].mul / 16);
      cost[y] = steps * (2500 + 500 * X.d3 + 170 * X.d3 * A.y_ct[y]);
    }
    for (int a = 1; a < ny; ++a)
      for (int b = a; b > 0 && cost[b] > cost[b - 1]; --b) {
        const signed char t0 = A.y_deg[b], t1 = A.y_ct0[b], t2 = A.y_ct[b];
        A.y_deg[b] = A.y_deg[b - 1], A.y_ct0[b] = A.y_ct0[b - 1], A.y_ct[b] = A.y_ct[b - 1];
        A.y_deg[b - 1] = t0, A.y_ct0[b - 1] = t1, A.y_ct[b - 1] = t2;
        const long tc = cost[b];
        cost[b] = cost[b - 1], cost[b - 1] = tc;
      }
    A.ord = lpt_order(eqf_cdiv(C.E, 32), ny, nblk);
  } else {
    A.ord = xcd_order(eqf_cdiv(C.E, 32), ny, nblk);
  }
  return 0;
}


inline int plan_bwd(const SfcCommon& C, const eqf_dtp_paths* P, int mode, XBwdArgs& A, int& nblk, size_t& lds, int& ngrp_out,
                    const XGate* gate = nullptr) {
  if (!fits32(C)) return EQF_E_UNSUPPORTED;
  if (max_deg(C) > 7) return EQF_E_UNSUPPORTED;
  if (gate && gate->on && max_deg(C) > 5) return EQF_E_UNSUPPORTED;  // (the half-pass items of degree-3 models are not gated)
  PkDeg pk[SFC_MAX_DEG];
  pack_layout(C, mode_npw(mode), pk);
  memset(&A, 0, sizeof A);
  A.x = C.x, A.coupling = C.coupling, A.w = C.w;
  A.x_ld = C.x_ld, A.m_ld = C.m_ld, A.w_ld = C.w_ld, A.E = C.E;
  A.d1 = C.o1, A.d2 = C.o2, A.ld1 = C.ld1, A.ld2 = C.ld2;
  if (gate) A.gate = *gate;
  for (int d = 0; d < C.ndeg; ++d) {
    const SfcDeg& D = C.deg[d];
    A.deg[d].d3 = D.d3, A.deg[d].N1 = D.N1, A.deg[d].Ncat = D.Ncat, A.deg[d].out1_off = D.out1_off;
    A.deg[d].nt = D.Ncat / 16, A.deg[d].pb = pk[d].pb;
  }
  int ngrp = 0, msmax = 1;
  // distinct input segments; one group per 32-channel slab of a segment; paths sorted by output degree
  int seg_off[EQF_MAX_SEG], nseg = 0;
  for (int p = 0; p < P->npaths; ++p) {
    bool found = false;
    for (int s = 0; s < nseg; ++s) found |= seg_off[s] == P->in_off[p];
    if (found) continue;
    if (nseg >= EQF_MAX_SEG) return EQF_E_UNSUPPORTED;
    seg_off[nseg++] = P->in_off[p];
    const int mul = P->mul[p], d1 = 2 * P->l1[p] + 1;
    if (mul % 32 != 0) return EQF_E_UNSUPPORTED;
    for (int c = 0; c < mul; c += 32) {
      if (ngrp >= XB_MAXGRP) return EQF_E_UNSUPPORTED;
      XBGroup& G = A.grp[ngrp];
      int raw_off = 0, g_off = -2;
      const int grc = gate_map(A.gate, P, P->in_off[p], mul, d1, raw_off, g_off);
      if (grc) return grc;
      G.x_off = raw_off + c, G.g_off = g_off >= 0 ? g_off + c : g_off, G.mul = (short)mul, G.d1 = (short)d1, G.npath = 0;
      for (int d = 0; d < C.ndeg; ++d)
        for (int q = 0; q < P->npaths; ++q) {
          if (P->in_off[q] != P->in_off[p] || P->l3[q] != C.deg[d].l3) continue;
          if (G.npath >= XB_MAXPATH) return EQF_E_UNSUPPORTED;
          XBPath& Q = G.p[G.npath++];
          Q.deg = (short)d, Q.mlen = (short)(d1 * C.deg[d].d3);
          Q.krow = P->out_ch[q] + c, Q.w_off = P->w_off[q] + c, Q.m_off = P->m_off[q];
          if ((Q.mlen | 1) > msmax) msmax = Q.mlen | 1;
        }
      if (G.npath > 0) ++ngrp;
    }
  }
  if (ngrp == 0) return EQF_E_BADARG;
  ngrp_out = ngrp;
#if EQF_X_LPT
  {  // heaviest groups first (order_xy mode 3): matrix instructions of a group's paths + its register contraction
    long cost[XB_MAXGRP];
    for (int k = 0; k < ngrp; ++k) {
      long c = 0;
      for (int q = 0; q < A.grp[k].npath; ++q) {
        const XBPath& Q = A.grp[k].p[q];
        const int d3 = A.deg[Q.deg].d3;
        c += (long)A.deg[Q.deg].nt * d3 * 5 * 32 + 64L * A.grp[k].d1 * d3 * 16 / 4 + 3000;
      }
      cost[k] = c;
    }
    for (int a = 1; a < ngrp; ++a)  // stable insertion sort, descending
      for (int b = a; b > 0 && cost[b] > cost[b - 1]; --b) {
        const XBGroup tg = A.grp[b];
        A.grp[b] = A.grp[b - 1], A.grp[b - 1] = tg;
        const long tc = cost[b];
        cost[b] = cost[b - 1], cost[b - 1] = tc;
      }
  }
#endif
  if ((C.x_ld | C.w_ld | C.ld1 | C.ld2) & 3) return EQF_E_UNSUPPORTED;  // the row-major tiles are read with 16-byte loads
  A.ms = (msmax + 3) & ~3;  // the transposition tile behind the coupling block stays 16-byte aligned
  lds = (size_t)(32 * A.ms + (1 + 5) * XT_FLOATS) * sizeof(float);  // coupling block + x / w / dx / dw tile + one d_out tile per m3
#if EQF_X_LPT
  A.ord = lpt_order(eqf_cdiv(C.E, 32), ngrp, nblk, true);
#else
  A.ord = xcd_order(eqf_cdiv(C.E, 32), ngrp, nblk);
#endif
  return 0;
}

}  // namespace

// csrc/sfcy.hip: the multi-wave forward (round 6).  EQF_E_UNSUPPORTED: shape outside its tables, nothing launched.
int sfcy_fwd_launch(const sfc::SfcCommon* C, const eqf_dtp_paths* paths, int mode, int gate_on, int gS, int gG, float c_silu,
                    float c_sig, const float* bias0, const float* bias2, const void* packed, void* stream);
// csrc/sfcw.hip: the multi-wave weight gradient (round 6).  EQF_E_UNSUPPORTED: shape outside its tables, nothing launched.
int sfcw_wgrad_launch(const sfc::SfcCommon* C, const eqf_dtp_paths* paths, int mode, int gate_on, int gS, int gG, float c_silu,
                      float c_sig, float* d_bias0, float* d_bias2, void* stream);
void sfcw_dev_set(int key, int value);
int sfcw_dev_plan(const sfc::SfcCommon* C, const eqf_dtp_paths* paths, int mode, char* buf, int buflen);
