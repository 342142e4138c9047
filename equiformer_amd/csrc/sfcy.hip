// Fused SeparableFCTP forward, second generation (round 6): multi-wave workgroups, every operand through LDS by DMA.
//
//   mid[e,(p,u),m3] = w[e,p,u] * sum_i M_p[e][i,m3] * x[e,l1(p),i,u]
//   out[e,l3,m3,n]  = sum_{(p,u) -> l3} mid[e,(p,u),m3] * W_l3[(p,u),n]
//
// [ref: SeparableFCTP.forward, nets/graph_attention_transformer.py:234-248]
//
// Why a second kernel (sfcx.hip's forward stays as the bit-level cross-check and serves small graphs / degree-3 models): the
// one-wave items of sfcx_fwd each wait 1.0-1.4 us per step for operands they requested when they needed them, two resident
// waves per SIMD cover neither, and 9 of the 13 wave-loads of a step are weight fragments that every 32-edge tile re-reads
// through the CU's one address unit (profiles/r03/r03_s_what_bounds_the_forward.md).  Here (E = 25 354, split mode: 126 us
// against 155 us for sep_act, 84 / 96 for sep_value, 100 / 115 with the gate folded in; results bit-identical; how it got there,
// step by step with cycle traces: profiles/r06/r06_c ... r06_m):
//
//   * workgroup = 8 waves on 4 consecutive 32-edge tiles, all running the same item (output degree, column group) in step:
//     COMPUTE wave t (0..3) owns tile t, LOADER wave 4 + t sits beside it on the same SIMD and issues every memory request
//     of the tile -- the compute waves have no vector-memory instruction in their loop;
//   * every operand arrives in LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per instruction, no registers, scalar base +
//     per-lane offset): the bf16 weight planes of a step (CT x NPW KB, one contiguous block of the packed buffer) ONCE per
//     workgroup into a ring of three slots (two where LDS is short), read by all four compute waves with ds_read_b128 -- a
//     quarter of the weight traffic through the address unit; x and w of a tile as 16 rows x 64 bytes per instruction into
//     the wave's private slots (x 2 chunks, w 3 pieces), row-major with an XOR swizzle that makes the fragment reads
//     conflict free;
//   * a step = (input segment, 16-channel chunk, path); its parameters come from a table the workgroup builds in LDS (one
//     broadcast ds_read per step instead of ~30 dependent scalar loads from the kernarg tables);
//   * one raw s_barrier per step is the only synchronisation: a loader waits (counted s_waitcnt vmcnt(N): the DMAs are
//     inline asm, hipcc neither counts nor drains them) until what the NEXT step needs has landed, everybody meets, the
//     loader requests what the step after next needs -- weight planes and w two steps ahead, x one or two;
//   * the steps of an item run as one loop per input degree with a branch-free body (generation, split, matrix instructions of
//     all 2 l3 + 1 components in one basic block): hipcc interleaves the vector work of component m3 + 1 with the matrix
//     instructions of m3;
//   * column groups of 6 tiles on the scalar degree: the degree-0 DTP output is generated twice per tile instead of four times;
//   * output tiles leave through a wave-private LDS image as 16-byte stores of whole lines.
// What bounds it now (r06_k / r06_l traces): a loader spends ~950 cycles per step issuing ~5 DMAs (the CU's address unit takes
// ~43 cycles per 1 KB DMA and serves four loaders), then waits as long again for the previous step's to land, against 1 400
// cycles of arithmetic per step -- deeper rings need LDS that the x slots (2 x 10 KB per tile) do not leave.
#include "sfcx_common.h"

extern __shared__ __attribute__((aligned(16))) float sy_lds[];

namespace {
using namespace sfc;

// development: -DEQF_Y_TRACE=1 prints the cycles one wave of the first workgroup of every item type spends per phase
#ifndef EQF_Y_TRACE
#define EQF_Y_TRACE 0
#endif
#if EQF_Y_TRACE
#define YT_STAMP(k)                        \
  do {                                     \
    const long long tn = clock64();        \
    yt[k] += tn - yt_last, yt_last = tn;   \
  } while (0)
#else
#define YT_STAMP(k) \
  do {              \
  } while (0)
#endif
constexpr int Y_WAVES = 4;
constexpr int Y_MAXTYPE = 8;
__host__ __device__ constexpr int y_ctmax(int d3) { return d3 == 1 ? 6 : (d3 == 3 ? 2 : 1); }

struct YType {
  int deg, ct0, ct;
  int xs_bytes;    // one x slot: (max d1 of the degree's segments [+ 1 gate piece]) x 2 KB
  int ms;          // row stride (floats, odd) of the staged coupling block
  int wave_bytes;  // LDS per wave: 2 x slots + 3 w slots + coupling block
  int nsteps;      // (segment, chunk, path) steps of the degree
  int nst[3];      // ... of the segment of input degree 2 l1 + 1 = 1, 3, 5 (the segments run in this order)
  int nsb;         // slots of the weight-plane ring: 3 (planes requested two steps ahead) where LDS allows, else 2 (one step)
  int tab_bytes;   // step table, rounded to 1 KB
};
struct YFwdArgs {
  XFwdArgs f;  // tensors, per-degree segment / path tables, gate (plan_fwd)
  int ngrp, ntype, has_w, batch;
  YType type[Y_MAXTYPE];
};

typedef __attribute__((address_space(3))) float lds_float;

// LDS-DMA: 64 lanes x 16 bytes from sbase + voff (scalar 64-bit base, per-lane 32-bit byte offset) to LDS byte address lds_dst
// + 16 lane.  M0 carries the LDS address; hipcc reserves M0 but holds nothing in it across statements of this kernel (no
// movrel / sendmsg / GWS), so it is written and left.  Five issue slots per DMA including the two scalar adds of the caller
// (the first version -- per-lane 64-bit pointers, M0 saved and restored, destination through v_readfirstlane -- took ~100
// cycles of a loader wave per DMA: profiles/r06/r06_g_*).
__device__ __forceinline__ void glds16(const void* sbase, const unsigned voff, const unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_dst)
               : "memory");
}
// a wave-uniform value that hipcc chose to compute with vector instructions (selects of loop counters), back in an SGPR: the
// opaque copy keeps the readfirstlane from being folded away
__device__ __forceinline__ unsigned force_sgpr(unsigned v) {
  asm volatile("" : "+v"(v));
  return __builtin_amdgcn_readfirstlane(v);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// at most n DMA instructions of a loader wave still in flight (two w pieces + up to 2 (5 + 1) x pieces + up to 5 weight pieces)
__device__ __forceinline__ void wait_vmcnt(const int n) {
#define Y_W(k) case k: wait_vm<k>(); break;
  switch (n) {
    Y_W(0) Y_W(1) Y_W(2) Y_W(3) Y_W(4) Y_W(5) Y_W(6) Y_W(7) Y_W(8) Y_W(9) Y_W(10) Y_W(11) Y_W(12) Y_W(13) Y_W(14) Y_W(15)
    Y_W(16) Y_W(17) Y_W(18) Y_W(19) Y_W(20) Y_W(21) Y_W(22) Y_W(23) Y_W(24)
    default: wait_vm<0>(); break;
  }
#undef Y_W
}

// One step of an item = (input segment, 16-channel chunk, path).  The table of an item's steps is built once per workgroup in
// LDS (thread t: step t) and read back one entry per iteration with a broadcast ds_read + v_readfirstlane: walking the
// segment / path tables of the kernarg segment instead cost ~30 dependent scalar loads per step, each behind an
// s_waitcnt lgkmcnt(0) that also drains the LDS reads in flight (profiles/r06/r06_d_*).
struct YStep {
  int x_off;   // offset of the chunk's first component in the x row (segment offset + c)
  int w_off;   // offset of the step's 16 weights in the w row
  int kt;      // 16-row block of W_l3: (kbase + c) / 16
  int m_rel;   // offset of the path's matrix in the staged coupling block
  int flags;   // bit 0: first step of a chunk, bit 1: first step of a segment, bit 2: x slot of the chunk, bit 3: the chunk's x is
               // requested ONE iteration ahead instead of two (its slot is read until then); bits 4-7: d1; bits 8-23: multiplicity
  int g_off;   // gated input: offset of the chunk's gate scalars in the raw row (-1 scalar segment, -2 plain)
  int m_blk;   // coupling block of the segment: m_off | m_len << 16
  int m_next;  // the same of the NEXT segment (0: none): fetched into registers while this segment runs
};
constexpr int Y_MAXSTEP = 64;
constexpr int Y_TAB_BYTES = Y_MAXSTEP * (int)sizeof(YStep);
constexpr int Y_MAXMLEN = 96;  // coupling floats per edge of one (segment, output degree) block: three 32-column pieces

__device__ __forceinline__ YStep y_unpack(const int4 a, const int4 b) {
  YStep e;
  e.x_off = __builtin_amdgcn_readfirstlane(a.x), e.w_off = __builtin_amdgcn_readfirstlane(a.y);
  e.kt = __builtin_amdgcn_readfirstlane(a.z), e.m_rel = __builtin_amdgcn_readfirstlane(a.w);
  e.flags = __builtin_amdgcn_readfirstlane(b.x), e.g_off = __builtin_amdgcn_readfirstlane(b.y);
  e.m_blk = __builtin_amdgcn_readfirstlane(b.z), e.m_next = __builtin_amdgcn_readfirstlane(b.w);
  return e;
}
__device__ __forceinline__ YStep y_entry(const float* tab, const int s) {
  const int4* const p = reinterpret_cast<const int4*>(tab) + 2 * s;  // uniform address: broadcast read
  return y_unpack(p[0], p[1]);
}

// LDS map of a workgroup, bytes from the start of the dynamic segment:
//   [step table][B slots: 2 or 3] then per compute wave [x 0][x 1][w 0][w 1][w 2][M]
struct YMap {
  unsigned lds0, bs_off, bs_bytes, wv_off, wave_bytes, xs_bytes, ws_rel, mb_rel;
};
template <int NPW>
__device__ __forceinline__ YMap y_map(const YFwdArgs& g, const YType& T) {
  YMap m;
  m.lds0 = (unsigned)(size_t)(lds_float*)sy_lds;
  m.bs_off = T.tab_bytes, m.bs_bytes = (unsigned)T.ct * NPW * 1024;
  m.wv_off = m.bs_off + T.nsb * m.bs_bytes, m.wave_bytes = T.wave_bytes, m.xs_bytes = T.xs_bytes;
  m.ws_rel = 2 * T.xs_bytes, m.mb_rel = m.ws_rel + (g.has_w ? 3 * 2048 : 0);
  return m;
}

__device__ __forceinline__ void y_build_table(const XFwdArgs::Deg& D, const int nsteps) {
  const int t = threadIdx.x;
  if (t < nsteps) {
    int t2 = t, si = 0, chunk0 = 0, first = 0;  // first: first step of the segment
    for (; si < D.nseg - 1; ++si) {
      const int n = D.seg[si].npath * (D.seg[si].mul >> 4);
      if (t2 < n) break;
      t2 -= n, first += n, chunk0 += D.seg[si].mul >> 4;
    }
    const XSeg& S = D.seg[si];
    const int ch = t2 / S.npath, pi = t2 - ch * S.npath, c = 16 * ch;
    // first step of the chunk two chunks back (same x slot): t - pi is this chunk's first step
    int late = 0;
    if (pi == 0 && chunk0 + ch >= 2) {
      int back = 0;  // steps of the two chunks before this one
      int sj = si, cj = ch;
      for (int k = 0; k < 2; ++k) {
        if (cj == 0) --sj, cj = D.seg[sj].mul >> 4;
        --cj;
        back += D.seg[sj].npath;
      }
      late = back < 3;
    }
    int4* const q = reinterpret_cast<int4*>(sy_lds) + 2 * t;
    const int flags = (pi == 0 ? 1 : 0) | ((pi == 0 && ch == 0) ? 2 : 0) | (((chunk0 + ch) & 1) << 2) | (late << 3) | (S.d1 << 4) | (S.mul << 8);
    const int m_next = si + 1 < D.nseg ? (D.seg[si + 1].m_off | (D.seg[si + 1].m_len << 16)) : 0;
    q[0] = int4{S.x_off + c, S.p[pi].w_off + c, (S.p[pi].kbase + c) >> 4, S.p[pi].m_rel};
    q[1] = int4{flags, S.g_off >= 0 ? S.g_off + c : S.g_off, S.m_off | (S.m_len << 16), m_next};
  }
  __syncthreads();
}

// --------------------------------------------------------------------------------------------------------- loader waves
// Waves 4..7 of the workgroup: loader k issues the DMAs of tile k (x, w) and every fourth weight piece, and waits for them;
// the compute waves only meet the loaders at the barrier that opens a step.  One loader beside one compute wave on every SIMD:
// issuing a DMA costs its wave tens of cycles (the address unit takes the next request when it is done with the last), and in
// the first version, where each wave loaded for itself, that was 45 k of an item's 130 k cycles in front of the step's
// arithmetic (profiles/r06/r06_f_*); a single loader for four tiles could not keep up (r06_g_*: the compute waves of the
// degree-0 items waited 2 500 cycles per step at the barrier).  Queue order per iteration s, after the barrier:
//   B(s+1) | x of step s+1 if flagged late | x of step s+2 unless late | w(s+2)
// and the wait that opens iteration s+1 leaves the last two groups in flight.
template <int NPW>
__device__ __forceinline__ void yf_loader(const YFwdArgs& g, const YType& T, const int grp, const int k) {
  const XFwdArgs& f = g.f;
  const XFwdArgs::Deg& D = f.deg[T.deg];
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int nsteps = T.nsteps, E = f.E;
  const YMap m = y_map<NPW>(g, T);
  const bool has_w = g.has_w != 0, gate_on = f.gate.on != 0;
  // x / w rows of a chunk (32 edges x 64 bytes per component) move as two DMAs of 16 rows x 64 bytes -- lane = (row, 16-byte
  // piece), 16 half lines per instruction; the first version fetched the A-fragment shape directly (lane = (edge, k half): 32
  // rows x 32 bytes, 32 lines per instruction) and the four loaders of a CU spent 1 250 - 2 000 cycles per step ISSUING them
  // behind one another (profiles/r06/r06_k_*).  The LDS image is row-major [row][4 pieces] with the piece index XOR (row / 4)
  // % 4, applied here on the source side (the DMA writes lane-linear) and again where the compute wave reads its fragment.
  const int prow = lane >> 2, ppiece = (lane & 3) ^ ((lane >> 4) & 3);
  const int e_lo = min((grp * Y_WAVES + k) * 32 + prow, E - 1), e_hi = min((grp * Y_WAVES + k) * 32 + prow + 16, E - 1);
  const unsigned xoff0 = ((unsigned)e_lo * f.x_ld + 4 * ppiece) * 4, xoff1 = ((unsigned)e_hi * f.x_ld + 4 * ppiece) * 4;  // bytes
  const unsigned woff0 = ((unsigned)e_lo * f.w_ld + 4 * ppiece) * 4, woff1 = ((unsigned)e_hi * f.w_ld + 4 * ppiece) * 4;
  const unsigned boff = lane * 16;
  (void)r, (void)hi;
  const char* const xg = reinterpret_cast<const char*>(f.x);
  const char* const wg = reinterpret_cast<const char*>(f.w);
  const char* const pf = reinterpret_cast<const char*>(f.packed + D.pf + (size_t)T.ct0 * NPW * 512);
  const size_t kt_bytes = (size_t)D.cttot * NPW * 1024;
  const int npiece = T.ct * NPW;
  const unsigned my = m.lds0 + m.wv_off + (unsigned)k * m.wave_bytes;

  auto issue_b = [&](const YStep& st, const int slot) __attribute__((always_inline)) {
    const char* const src = pf + (size_t)st.kt * kt_bytes;
    const unsigned dst = force_sgpr(m.lds0 + m.bs_off + slot * m.bs_bytes);
    for (int j = k; j < npiece; j += Y_WAVES) glds16(src + j * 1024, boff, dst + j * 1024);
  };
  auto issue_x = [&](const YStep& st) __attribute__((always_inline)) -> int {
    const int d1 = (st.flags >> 4) & 15, mul = (st.flags >> 8) & 0xffff;
    const bool gp = gate_on && st.g_off >= 0;
    const char* const xb = xg + (size_t)st.x_off * 4;
    const unsigned dst = my + ((st.flags >> 2) & 1) * m.xs_bytes;
    auto go = [&](auto tag) __attribute__((always_inline)) {
      constexpr int D1 = decltype(tag)::value;
#pragma unroll
      for (int i = 0; i < D1; ++i) {
        glds16(xb + (size_t)(i * mul) * 4, xoff0, dst + i * 2048);
        glds16(xb + (size_t)(i * mul) * 4, xoff1, dst + i * 2048 + 1024);
      }
      if (gp) {
        const char* const gb = xg + (size_t)st.g_off * 4;
        glds16(gb, xoff0, dst + D1 * 2048);
        glds16(gb, xoff1, dst + D1 * 2048 + 1024);
      }
    };
    switch (d1) {
      case 1: go(IC<1>()); break;
      case 3: go(IC<3>()); break;
      default: go(IC<5>()); break;
    }
    return 2 * (d1 + (gp ? 1 : 0));
  };
  auto issue_w = [&](const YStep& st, const int slot) __attribute__((always_inline)) {
    const char* const wb = wg + (size_t)st.w_off * 4;
    const unsigned dst = force_sgpr(my + m.ws_rel + slot * 2048);
    glds16(wb, woff0, dst);
    glds16(wb, woff1, dst + 1024);
  };

  YStep E1 = y_entry(sy_lds, min(1, nsteps - 1)), E2 = y_entry(sy_lds, min(2, nsteps - 1));
  const int nsb = T.nsb;
  const bool lead2 = nsb == 3;  // weight planes two steps ahead
  const int nb = (npiece - k + Y_WAVES - 1) / Y_WAVES;  // this loader's pieces of a weight stage
  int n_tail = 0;  // DMA instructions issued after the ones the next step needs
  {
    const YStep E0 = y_entry(sy_lds, 0);
    issue_x(E0);
    if (has_w) issue_w(E0, 0);
    issue_b(E0, 0);
    if (nsteps > 1) {
      if (lead2) issue_b(E1, 1), n_tail += nb;
      if (E1.flags & 1) n_tail += issue_x(E1);
      if (has_w) issue_w(E1, 1), n_tail += 2;
    }
  }
  int s3 = 0;
#if EQF_Y_TRACE
  long long yt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long yt_last = clock64();
#endif
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt(n_tail);
    YT_STAMP(1);  // operands landed
    __builtin_amdgcn_s_barrier();
    YT_STAMP(2);  // barrier
    const int4* const tq = reinterpret_cast<const int4*>(sy_lds) + 2 * min(s + 3, nsteps - 1);
    const int4 ta = tq[0], tb = tq[1];
    n_tail = 0;
    if (s + 1 < nsteps && (E1.flags & 9) == 9 && s > 0) issue_x(E1);  // late x of step s + 1 (step 1's went out with the prologue)
    if (lead2) {
      if (s + 2 < nsteps) issue_b(E2, s3 == 0 ? 2 : s3 - 1), n_tail += nb;  // slot (s + 2) % 3
    } else {
      if (s + 1 < nsteps) issue_b(E1, (s + 1) & 1);
    }
    if (s + 2 < nsteps) {
      if ((E2.flags & 9) == 1) n_tail += issue_x(E2);
      if (has_w) issue_w(E2, s3 == 0 ? 2 : s3 - 1), n_tail += 2;
    }
    s3 = s3 == 2 ? 0 : s3 + 1;
    E1 = E2, E2 = y_unpack(ta, tb);
    YT_STAMP(3);  // DMAs issued
  }
#if EQF_Y_TRACE
  if (grp == 3 && k == 0 && lane == 0)
    printf("ytrace loader d3 %d ct %d steps %d: wait %lld barrier %lld issue %lld\n", D.d3, T.ct, nsteps, yt[1], yt[2], yt[3]);
#endif
}

// -------------------------------------------------------------------------------------------------------- compute waves
// The steps of an item run segment by segment, and the input degree of a segment only shapes the generation of the A values:
// one loop per input degree (2 l1 + 1 = 1, 3, 5, in the order the planner guarantees), each with a branch-free step body --
// generation, split and matrix instructions of all 2 l3 + 1 components in ONE basic block, so that hipcc's scheduler can put
// the vector instructions of component m3 + 1 between the matrix instructions of m3.  (First version: a switch on the input
// degree inside the m3 loop; a wave issues one instruction per ~4 cycles whatever its kind, and with the blocks cut at every
// switch a step of the degree-2 item took 3 200 cycles for 320 vector + 25 matrix instructions: profiles/r06/r06_h_*.)
template <int D3, int MODE>
__device__ __forceinline__ void yf_compute(const YFwdArgs& g, const YType& T, const int grp, const int wave) {
  constexpr int CTM = y_ctmax(D3), NPA = Planes<MODE>::A, NPW = Planes<MODE>::W;
  const XFwdArgs& f = g.f;
  const XFwdArgs::Deg& D = f.deg[T.deg];
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int e0 = (grp * Y_WAVES + wave) * 32;  // (tiles past the end run on clamped rows and store nothing)
  const int CT = T.ct, ct0 = T.ct0, nsteps = T.nsteps, E = f.E;
  const bool valid = e0 + r < E;
#if EQF_Y_TRACE
  long long yt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long yt_last = clock64();
  const long long yt_begin = yt_last;
#endif
  const YMap m = y_map<NPW>(g, T);
  const unsigned wv = m.wv_off + (unsigned)wave * m.wave_bytes;
  float* const Mt = sy_lds + ((wv + m.mb_rel) >> 2);
  const int MS = T.ms;
  const bool has_w = g.has_w != 0, gate_on = f.gate.on != 0;
  const float c_silu = f.gate.c_silu, c_sig = f.gate.c_sig;
  const float* const cpl = f.coupling;
  const unsigned m_ld = f.m_ld;

  f32x16 acc[D3][CTM];
#pragma unroll
  for (int m3 = 0; m3 < D3; ++m3)
#pragma unroll
    for (int ct = 0; ct < CTM; ++ct)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m3][ct][q] = 0.f;

  // this lane's A-fragment pieces (channels 8 hi .. 8 hi + 3 and + 4 .. + 7 of edge r) in the row-major, XOR-swizzled image of a
  // chunk component (see yf_loader): float offsets of the first piece and from the first to the second
  const int swz = (r >> 2) & 3;
  const int frag0 = (r >> 4) * 256 + ((r & 15) * 4 + ((2 * hi) ^ swz)) * 4;
  const int fragd = (((2 * hi + 1) ^ swz) - ((2 * hi) ^ swz)) * 4;
  YStep E0 = y_entry(sy_lds, 0);
  YStep E1 = y_entry(sy_lds, min(1, nsteps - 1));
  const bool lead2 = T.nsb == 3;
  int s = 0, s3 = 0;  // s3 = s % 3
  YT_STAMP(0);

  auto run = [&](auto tag, const int nst) __attribute__((always_inline)) {
    constexpr int D1 = decltype(tag)::value;
    if (nst <= 0) return;
    float xf[D1][8];
    const int s_end = s + nst;
    {  // the segment's coupling block (ordinary loads -- the compute waves have no DMA in flight -- whose latency is exposed
       // up to three times per item: the next thing to move to the loaders)
      wave_lds_order();
      stage_m(Mt, MS, cpl + (E0.m_blk & 0xffff), m_ld, e0, E - 1, E0.m_blk >> 16, r, hi);
      wave_lds_order();
    }
#pragma unroll 1
    for (; s < s_end; ++s) {
      __builtin_amdgcn_s_barrier();
      YT_STAMP(2);  // barrier
      const int4* const tq = reinterpret_cast<const int4*>(sy_lds) + 2 * min(s + 2, nsteps - 1);
      const int4 ta = tq[0], tb = tq[1];
      if (E0.flags & 1) {  // first step of a chunk: its x rows out of their slot
        const float* const xs = sy_lds + ((wv + ((E0.flags >> 2) & 1) * m.xs_bytes) >> 2) + frag0;
#pragma unroll
        for (int i = 0; i < D1; ++i) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(xs + i * 512);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(xs + i * 512 + fragd);
#pragma unroll
          for (int j = 0; j < 4; ++j) xf[i][j] = a0[j], xf[i][4 + j] = a1[j];
        }
        if (gate_on) {  // uniform: the rows are the gate's INPUT, activate / gate them here
          if (E0.g_off == -1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[0][j] = c_silu * xf[0][j] * xg_sigmoid(xf[0][j]);
          } else if (E0.g_off >= 0) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(xs + D1 * 512);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(xs + D1 * 512 + fragd);
            float sg[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) sg[j] = c_sig * xg_sigmoid(g0[j]), sg[4 + j] = c_sig * xg_sigmoid(g1[j]);
#pragma unroll
            for (int i = 0; i < D1; ++i)
#pragma unroll
              for (int j = 0; j < 8; ++j) xf[i][j] *= sg[j];
          }
        }
      }
      YT_STAMP(3);  // x rows out of their slot
      float wf[8];
      if (has_w) {
        const float* const ws = sy_lds + ((wv + m.ws_rel + s3 * 2048) >> 2) + frag0;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ws);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(ws + fragd);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = valid ? a0[j] : 0.f, wf[4 + j] = valid ? a1[j] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = valid ? 1.f : 0.f;
      }
      // this edge's matrix of the step's path, all of it up front: one LDS round trip per step instead of one per m3
      const float* const mp = Mt + r * MS + E0.m_rel;
      float mm[D1 * D3];
#pragma unroll
      for (int k = 0; k < D1 * D3; ++k) mm[k] = mp[k];
      const __bf16* const bs =
          reinterpret_cast<const __bf16*>(sy_lds) + ((m.bs_off + (lead2 ? s3 : (s & 1)) * m.bs_bytes) >> 1) + lane * 8;
#pragma unroll
      for (int m3 = 0; m3 < D3; ++m3) {
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0.f;
#pragma unroll
        for (int i = 0; i < D1; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = fmaf(mm[i * D3 + m3], xf[i][j], a[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] *= wf[j];
        bf16x8 pa[NPA];
        split_planes<NPA>(a, pa);
#pragma unroll
        for (int ct = 0; ct < CTM; ++ct) {
          const int cc = ct < CT ? ct : CT - 1;  // (tiles past CT re-read the last one; their accumulators are never stored)
          bf16x8 bw[NPW];
#pragma unroll
          for (int pl = 0; pl < NPW; ++pl) bw[pl] = *reinterpret_cast<const bf16x8*>(bs + (cc * NPW + pl) * 512);
          mma_terms<NPA, NPW>(pa, bw, acc[m3][ct]);
        }
      }
      YT_STAMP(4);  // generation + matrix instructions (issue)
      s3 = s3 == 2 ? 0 : s3 + 1;
      E0 = E1, E1 = y_unpack(ta, tb);
    }
  };
  run(IC<1>(), T.nst[0]);
  run(IC<3>(), T.nst[1]);
  run(IC<5>(), T.nst[2]);

  // (two waves per SIMD -- a loader beside a compute wave -- so the register budget is 256 and hipcc keeps the accumulators in
  // VGPRs: no accumulator-file copies.  With 512 registers it copied all of them to VGPRs at the end of every iteration.)
  YT_STAMP(5);  // accumulators retired
  // epilogue: accumulator register q of lane (r, hi) = row (edge) (q & 3) + 8 (q >> 2) + 4 hi, column r of the tile.  The tile goes
  // through a wave-private LDS image (the wave's x slots are free: the loader's last DMA landed before the last barrier) and
  // leaves as 16-byte stores of whole 128-byte lines, 8 rows per instruction -- 4 store instructions per 32 x 32 tile instead of 16
  // (the 4-byte stores were 11-15 k of the 60 k cycles of a degree-0 item: profiles/r06/r06_i_*)
  {
    float* const Tt = sy_lds + (wv >> 2);  // two tile images of 32 x XT_LD floats, alternating
    const int c4 = lane & 7, rr = lane >> 3;
    int par = 0;
#pragma unroll
    for (int ct = 0; ct < CTM; ++ct) {
      if (ct >= CT) continue;
      const int c0 = (ct0 + ct) * 32;  // first column of the tile in the concatenated [main | second] output
      const bool main = c0 < D.N1;     // scalar: N1 % 32 == 0
      float bv = 0.f;
      if (D3 == 1) bv = main ? (f.bias ? f.bias[c0 + r] : 0.f) : (f.bias2 ? f.bias2[c0 + r - D.N1] : 0.f);
      asm volatile("" : "+v"(bv));  // (the bias has arrived: no wait inside the masked stores below)
      float* const base = main ? f.o1 + D.out1_off + c0 : f.o2 + (c0 - D.N1);  // uniform
      const unsigned ld = main ? f.ld1 : f.ld2;
#pragma unroll
      for (int m3 = 0; m3 < D3; ++m3) {
        float* const Tq = Tt + par * XT_FLOATS;
        par ^= 1;
        wave_lds_order();  // (the reads of this image's previous use are issued)
#pragma unroll
        for (int q = 0; q < 16; ++q) Tq[((q & 3) + 8 * (q >> 2) + 4 * hi) * XT_LD + r] = acc[m3][ct][q] + bv;
        wave_lds_order();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = rr + 8 * it;
          const f32x4 u = *reinterpret_cast<const f32x4*>(Tq + (row * XT_LD + 4 * c4));
          if (e0 + row < E)
            *reinterpret_cast<f32x4*>(base + ((unsigned)(e0 + row) * ld + (unsigned)(m3 * D.N1 + 4 * c4))) = u;
        }
      }
    }
  }
#if EQF_Y_TRACE
  YT_STAMP(6);  // stores issued
  if (grp == 3 && wave == 0 && lane == 0)
    printf("ytrace d3 %d ct %d steps %d: total %lld | prologue %lld barrier %lld stage %lld compute %lld retire %lld stores %lld\n",
           D3, CT, nsteps, clock64() - yt_begin, yt[0], yt[2], yt[3], yt[4], yt[5], yt[6]);
#endif
}

template <int MODE>
__global__ __launch_bounds__(128 * Y_WAVES, 2) void sfcy_fwd_kernel(const YFwdArgs g_byval) {
  KERNARG_IN_PLACE(YFwdArgs);
  // item-major launch order, heaviest item type first (the host sorts the types): grp fastest.  (Grouping the types of a tile
  // group on one XCD, so that they meet their x rows in its L2, measured 3 % slower: the short items no longer fill the tail.)
  // Launch order.  batch == 0: item-type-major over all tile groups, heaviest type first.  batch > 0: XCD k = blockIdx % 8 owns the
  // tile groups 8 q + k and runs them in batches of `batch` groups, heaviest type first inside a batch -- the four item types of a
  // group then meet its x / coupling rows in the XCD's L2.  Measured (profiles/r06/r06_x): with per-edge weights (sep_act) batches
  // of 8 are 3-4 % faster and cut the fetched bytes from 424 to 201 MB per launch; without (sep_value) they are 4 % SLOWER although
  // they fetch 114 instead of 199 MB -- the host picks per operator.
  int y, grp;
  if (g.batch > 0) {
    const int k8 = blockIdx.x & 7, s8 = blockIdx.x >> 3, per_batch = g.batch * g.ntype;
    const int bt = s8 / per_batch, r8 = s8 - bt * per_batch;
    y = r8 / g.batch;
    grp = 8 * (bt * g.batch + (r8 - y * g.batch)) + k8;
    if (grp >= g.ngrp) return;  // (whole workgroup, before any barrier)
  } else {
    y = blockIdx.x / g.ngrp, grp = blockIdx.x - y * g.ngrp;
  }
  const YType& T = g.type[y];
  y_build_table(g.f.deg[T.deg], T.nsteps);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave >= Y_WAVES) {
    yf_loader<Planes<MODE>::W>(g, T, grp, wave - Y_WAVES);
    return;
  }
  switch (g.f.deg[T.deg].d3) {
    case 1: yf_compute<1, MODE>(g, T, grp, wave); break;
    case 3: yf_compute<3, MODE>(g, T, grp, wave); break;
    default: yf_compute<5, MODE>(g, T, grp, wave); break;
  }
}

int plan_yfwd(const SfcCommon& C, const eqf_dtp_paths* P, int mode, const XGate* gate, YFwdArgs& A, int& nblk, size_t& lds) {
  if (max_deg(C) > 5) return EQF_E_UNSUPPORTED;
  size_t lds1 = 0;
  int nb1 = 0;
  int rc = plan_fwd(C, P, mode, A.f, nb1, lds1, gate);
  if (rc) return rc;
  if ((C.x_ld | C.w_ld) & 3) return EQF_E_UNSUPPORTED;  // 16-byte DMA pieces
  const int npw = mode_npw(mode);
  A.has_w = C.w != nullptr, A.batch = 0;
  A.ngrp = eqf_cdiv(eqf_cdiv(C.E, 32), Y_WAVES);
  int nt = 0;
  long cost[Y_MAXTYPE];
  size_t ldsmax = 0;
  for (int d = 0; d < C.ndeg; ++d) {
    const XFwdArgs::Deg& X = A.f.deg[d];
    if (X.d3 != 1 && X.d3 != 3 && X.d3 != 5) return EQF_E_UNSUPPORTED;
    int d1max = 1, mlen = 1, gpiece = 0;
    long steps = 0;
    int nst[3] = {0, 0, 0};
    for (int si = 0; si < X.nseg; ++si) {
      const XSeg& S = X.seg[si];
      if (S.d1 != 1 && S.d1 != 3 && S.d1 != 5) return EQF_E_UNSUPPORTED;
      if (si > 0 && S.d1 <= X.seg[si - 1].d1) return EQF_E_UNSUPPORTED;  // one segment per input degree, in increasing order
      nst[S.d1 >> 1] = S.npath * (S.mul / 16);
      if (S.mul % 16 != 0 || (S.x_off & 3)) return EQF_E_UNSUPPORTED;
      if (S.g_off >= 0 && (S.g_off & 3)) return EQF_E_UNSUPPORTED;
      d1max = S.d1 > d1max ? S.d1 : d1max;
      mlen = S.m_len > mlen ? S.m_len : mlen;
      gpiece |= (A.f.gate.on && S.g_off >= 0) ? 1 : 0;
      steps += (long)S.npath * (S.mul / 16);
      for (int q = 0; q < S.npath; ++q)
        if ((S.p[q].w_off & 3) || (S.p[q].kbase & 15)) return EQF_E_UNSUPPORTED;
      if (S.m_off > 0xffff || S.m_len > Y_MAXMLEN || S.mul > 0xffff) return EQF_E_UNSUPPORTED;
    }
    const int ctm = y_ctmax(X.d3);
    const int ng = eqf_cdiv(X.cttot, ctm), cps = eqf_cdiv(X.cttot, ng);
    for (int k = 0; k < ng; ++k) {
      const int c0 = k * cps, cn = (X.cttot - c0 < cps) ? X.cttot - c0 : cps;
      if (cn <= 0) continue;
      if (nt >= Y_MAXTYPE) return EQF_E_UNSUPPORTED;
      YType& T = A.type[nt];
      T.deg = d, T.ct0 = c0, T.ct = cn;
      T.xs_bytes = (d1max + gpiece) * 2048;
      T.ms = mlen | 1;
      T.wave_bytes = 2 * T.xs_bytes + (A.has_w ? 3 * 2048 : 0) + ((32 * T.ms * 4 + 15) & ~15);
      T.nsteps = (int)steps;
      T.nst[0] = nst[0], T.nst[1] = nst[1], T.nst[2] = nst[2];
      if (steps > Y_MAXSTEP) return EQF_E_UNSUPPORTED;
      if (2 * T.xs_bytes < 2 * XT_FLOATS * 4) return EQF_E_UNSUPPORTED;  // (the epilogue's two tile images live in the x slots)
      if ((C.ld1 | C.ld2) & 3) return EQF_E_UNSUPPORTED;                 // 16-byte stores
      T.tab_bytes = (int)((steps * sizeof(YStep) + 1023) & ~(size_t)1023);
      T.nsb = 3;
      size_t need = (size_t)T.tab_bytes + (size_t)T.nsb * cn * npw * 1024 + (size_t)Y_WAVES * T.wave_bytes;
      if (need > 160 * 1024) T.nsb = 2, need -= (size_t)cn * npw * 1024;
      if (need > 160 * 1024) return EQF_E_UNSUPPORTED;
      ldsmax = need > ldsmax ? need : ldsmax;
      cost[nt] = steps * (600 + 200 * X.d3 + 160 * X.d3 * cn);
      ++nt;
    }
  }
  for (int a = 1; a < nt; ++a)  // heaviest type first
    for (int b = a; b > 0 && cost[b] > cost[b - 1]; --b) {
      const YType t = A.type[b];
      A.type[b] = A.type[b - 1], A.type[b - 1] = t;
      const long tc = cost[b];
      cost[b] = cost[b - 1], cost[b - 1] = tc;
    }
  A.ntype = nt;
  A.batch = A.has_w ? 8 : 0;
  nblk = A.batch ? 8 * (eqf_cdiv(eqf_cdiv(A.ngrp, 8), A.batch) * A.batch) * nt : A.ngrp * nt;
  lds = ldsmax;
  return 0;
}

}  // namespace

// Launch of the multi-wave forward for the operator described by C (built by sfcx.hip's entry point); returns
// EQF_E_UNSUPPORTED when the shape is outside this kernel's tables (the caller then runs the one-wave kernel).
int sfcy_fwd_launch(const sfc::SfcCommon* Cp, const eqf_dtp_paths* paths, int mode, int gate_on, int gS, int gG, float c_silu,
                    float c_sig, const float* bias0, const float* bias2, const void* packed, void* stream) {
  const SfcCommon& C = *Cp;
  XGate XG;
  memset(&XG, 0, sizeof XG);
  XG.on = gate_on, XG.S = gS, XG.G = gG, XG.c_silu = c_silu, XG.c_sig = c_sig;
  static thread_local YFwdArgs A;
  int nblk = 0;
  size_t lds = 0;
  int rc = plan_yfwd(C, paths, mode, &XG, A, nblk, lds);
  if (rc) return rc;
  A.f.bias = bias0, A.f.bias2 = bias2;
  A.f.packed = (const __bf16*)packed;
  hipStream_t st = (hipStream_t)stream;
#define YF_LAUNCH(M)                                                                                                 \
  do {                                                                                                               \
    static bool big_lds = false; /* (dynamic LDS beyond 64 KB has to be allowed once per kernel) */                  \
    if (!big_lds) {                                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sfcy_fwd_kernel<M>),                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                 \
        return EQF_E_UNSUPPORTED;                                                                                    \
      big_lds = true;                                                                                                \
    }                                                                                                                \
    hipLaunchKernelGGL((sfcy_fwd_kernel<M>), dim3(nblk), dim3(128 * Y_WAVES), lds, st, A);                            \
  } while (0)
  if (mode == 0) YF_LAUNCH(0);
  else if (mode == 1) YF_LAUNCH(1);
  else YF_LAUNCH(2);
#undef YF_LAUNCH
  return 0;
}
