// Fused SeparableFCTP forward, second generation (round 6): multi-wave workgroups, every operand through LDS by DMA.
//
//   mid[e,(p,u),m3] = w[e,p,u] * sum_i M_p[e][i,m3] * x[e,l1(p),i,u]
//   out[e,l3,m3,n]  = sum_{(p,u) -> l3} mid[e,(p,u),m3] * W_l3[(p,u),n]
//
// [ref: SeparableFCTP.forward, nets/graph_attention_transformer.py:234-248]
//
// Why a second kernel (sfcx.hip's forward stays as the bit-level cross-check and serves small graphs / degree-3 models):
// the one-wave items of sfcx_fwd each wait 1.0-1.4 us per step for operands they requested when they needed them, two
// resident waves per SIMD cover neither, and 9 of the 13 wave-loads of a step are weight fragments that every 32-edge tile
// re-reads through the CU's one address unit (profiles/r03/r03_s_what_bounds_the_forward.md; round 6: a better launch order
// alone gave the operator without per-edge weights 25 %, a touch-prefetch of w made the other one slower -- the address unit,
// not HBM latency, is what the steps queue on).  Here:
//
//   * workgroup = 4 waves = 4 consecutive 32-edge tiles, all running the same item (output degree, column group) in step;
//     the bf16 weight planes of a step (CT x NPW KB, one contiguous block of the packed buffer) are fetched ONCE per
//     workgroup into a two-slot LDS ring by LDS-DMA (global_load_lds_dwordx4, 1 KB per instruction, no registers) and read by
//     all four waves with ds_read_b128: a quarter of the weight traffic through the address unit, none of it in registers;
//   * x and w of a wave's tile arrive by LDS-DMA as well, in the wave's private ring slots, one chunk / two steps ahead: a
//     step never waits for an operand it has not requested at least a step earlier.  The only waits in the loop are counted
//     s_waitcnt vmcnt(N) (the DMAs are inline asm: hipcc neither counts nor drains them) and one raw s_barrier per step;
//   * column groups of 6 tiles on the scalar degree (registers: one wave per SIMD, 512 per lane): the degree-0 DTP output is
//     generated twice per tile instead of four times.
//
// A step = (input segment, 16-channel chunk, path).  Order of the DMA queue of a wave, per iteration s (after the barrier):
//   [its share of B(s+1)] [x of the next chunk, if step s opens a chunk] [w(s+2)]
// and the wait that opens iteration s+1 leaves outstanding exactly what is not needed yet: w(s+2), and the x chunk unless
// step s+1 opens it.  Ring slots: B 2 (slot of step s-1 is free once every wave passed barrier s), x 2, w 3, all wave-
// private except B.
#include "sfcx_common.h"

extern __shared__ __attribute__((aligned(16))) float sy_lds[];

namespace {
using namespace sfc;

// development ablations (variant builds only: tools/bench_sfcy.py, profiles/r06): bit 0 no matrix instructions, bit 1 no
// generation / split (constant A planes), bit 2 coupling block staged once per item only, bit 3 no stores, bit 4 no x / w DMA
// after the prologue, bit 5 no B DMA after the prologue
#ifndef EQF_Y_ABLATE
#define EQF_Y_ABLATE 0
#endif
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
};
struct YFwdArgs {
  XFwdArgs f;  // tensors, per-degree segment / path tables, gate (plan_fwd)
  int ngrp, ntype, has_w, pad;
  YType type[Y_MAXTYPE];
};

typedef __attribute__((address_space(3))) float lds_float;

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + 16 lane (wave-uniform byte address in M0)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
  // (the destination is wave-uniform by construction, but hipcc keeps loop-carried copies of it in VGPRs and then cannot
  // satisfy an "s" constraint: it is read back with v_readfirstlane inside the statement)
  unsigned keep, dst;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_nop 0\n\tv_readfirstlane_b32 %1, %3\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(dst)
      : "v"(gsrc), "v"(lds_dst_uniform)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// at most n DMA instructions of this wave still in flight (n even, <= 14: two w pieces + up to 2 (5 + 1) x pieces)
__device__ __forceinline__ void wait_vmcnt(const int n) {
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 2: wait_vm<2>(); break;
    case 4: wait_vm<4>(); break;
    case 6: wait_vm<6>(); break;
    case 8: wait_vm<8>(); break;
    case 10: wait_vm<10>(); break;
    case 12: wait_vm<12>(); break;
    case 14: wait_vm<14>(); break;
    default: wait_vm<0>(); break;
  }
}

// One step of an item = (input segment, 16-channel chunk, path).  The table of an item's steps is built once per workgroup in
// LDS (thread t: step t) and read back one entry per iteration with a broadcast ds_read + v_readfirstlane: walking the
// segment / path tables of the kernarg segment instead cost ~30 dependent scalar loads per step, each behind an
// s_waitcnt lgkmcnt(0) that also drains the LDS reads in flight (profiles/r06/r06_d_*: 73 of 183 us with everything else off).
struct YStep {
  int x_off;   // offset of the chunk's first component in the x row (segment offset + c)
  int w_off;   // offset of the step's 16 weights in the w row
  int kt;      // 16-row block of W_l3: (kbase + c) / 16
  int m_rel;   // offset of the path's matrix in the staged coupling block
  int flags;   // bit 0: first step of a chunk, bit 1: first step of a segment, bit 2: x slot of the chunk; bits 4-7: d1
  int mul;     // multiplicity of the segment (stride between x components)
  int g_off;   // gated input: offset of the chunk's gate scalars in the raw row (-1 scalar segment, -2 plain)
  int m_blk;   // coupling block of the segment: m_off | m_len << 16
};
constexpr int Y_MAXSTEP = 64;
constexpr int Y_TAB_BYTES = Y_MAXSTEP * (int)sizeof(YStep);

__device__ __forceinline__ YStep y_entry(const float* tab, const int s) {
  const int4* const p = reinterpret_cast<const int4*>(tab) + 2 * s;  // uniform address: broadcast read
  const int4 a = p[0], b = p[1];
  YStep e;
  e.x_off = __builtin_amdgcn_readfirstlane(a.x), e.w_off = __builtin_amdgcn_readfirstlane(a.y);
  e.kt = __builtin_amdgcn_readfirstlane(a.z), e.m_rel = __builtin_amdgcn_readfirstlane(a.w);
  e.flags = __builtin_amdgcn_readfirstlane(b.x), e.mul = __builtin_amdgcn_readfirstlane(b.y);
  e.g_off = __builtin_amdgcn_readfirstlane(b.z), e.m_blk = __builtin_amdgcn_readfirstlane(b.w);
  return e;
}

template <int D3, int MODE>
__device__ __forceinline__ void yf_item(const YFwdArgs& g, const YType& T, const int grp) {
  constexpr int CTM = y_ctmax(D3), NPA = Planes<MODE>::A, NPW = Planes<MODE>::W, XD = 5;
  const XFwdArgs& f = g.f;
  const XFwdArgs::Deg& D = f.deg[T.deg];
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int e0 = (grp * Y_WAVES + wave) * 32;  // (tiles past the end run on clamped rows and store nothing)
  const bool valid = e0 + r < f.E;
  const unsigned er = valid ? e0 + r : f.E - 1;
  const int CT = T.ct, ct0 = T.ct0, nsteps = T.nsteps;
#if EQF_Y_TRACE
  long long yt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long yt_last = clock64();
  const long long yt_begin = yt_last;
#endif

  // LDS map, bytes from the start of the dynamic segment: [step table][B slot 0][B slot 1] then per wave
  // [x 0][x 1][w 0][w 1][w 2][M]
  const unsigned lds0 = (unsigned)(size_t)(lds_float*)sy_lds;
  const unsigned bs_bytes = (unsigned)CT * NPW * 1024;
  const unsigned bs_off = Y_TAB_BYTES;
  const unsigned wv_off = bs_off + 2 * bs_bytes + (unsigned)wave * T.wave_bytes;
  const unsigned xs_off = wv_off, ws_off = xs_off + 2 * T.xs_bytes, mb_off = ws_off + (g.has_w ? 3 * 2048 : 0);
  float* const Mt = sy_lds + (mb_off >> 2);
  const int MS = T.ms;

  {  // step table: thread t builds step t
    const int t = threadIdx.x;
    if (t < nsteps) {
      int t2 = t, si = 0, chunk0 = 0;
      for (; si < D.nseg - 1; ++si) {
        const int n = D.seg[si].npath * (D.seg[si].mul >> 4);
        if (t2 < n) break;
        t2 -= n, chunk0 += D.seg[si].mul >> 4;
      }
      const XSeg& S = D.seg[si];
      const int ch = t2 / S.npath, pi = t2 - ch * S.npath, c = 16 * ch;
      YStep e;
      e.x_off = S.x_off + c, e.w_off = S.p[pi].w_off + c, e.kt = (S.p[pi].kbase + c) >> 4, e.m_rel = S.p[pi].m_rel;
      e.flags = (pi == 0 ? 1 : 0) | ((pi == 0 && ch == 0) ? 2 : 0) | (((chunk0 + ch) & 1) << 2) | (S.d1 << 4);
      e.mul = S.mul, e.g_off = S.g_off >= 0 ? S.g_off + c : S.g_off, e.m_blk = S.m_off | (S.m_len << 16);
      int4* const q = reinterpret_cast<int4*>(sy_lds) + 2 * t;
      q[0] = int4{e.x_off, e.w_off, e.kt, e.m_rel};
      q[1] = int4{e.flags, e.mul, e.g_off, e.m_blk};
    }
    __syncthreads();
  }

  f32x16 acc[D3][CTM];
#pragma unroll
  for (int m3 = 0; m3 < D3; ++m3)
#pragma unroll
    for (int ct = 0; ct < CTM; ++ct)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m3][ct][q] = 0.f;

  const __bf16* const pf = f.packed + D.pf + (size_t)ct0 * NPW * 512 + lane * 8;
  const unsigned kt_stride = (unsigned)D.cttot * NPW * 512;
  const int npiece = CT * NPW;
  const unsigned xrow = er * f.x_ld + 8 * hi, wrow = er * f.w_ld + 8 * hi;
  const bool has_w = g.has_w != 0, gate_on = f.gate.on != 0;
  // loop-invariant arguments by value (read through the kernarg pointer they would be re-loaded after every asm statement)
  const float c_silu = f.gate.c_silu, c_sig = f.gate.c_sig;
  const float* const cpl = f.coupling;
  const float* const xg = f.x;
  const float* const wg = f.w;
  const unsigned m_ld = f.m_ld;
  const int E = f.E, xs_bytes = T.xs_bytes;

  // ---- DMA issue (all addresses: uniform base + 32-bit lane offset; the host rejects tensors of >= 2^31 elements)
  auto issue_b = [&](const YStep& e, const int slot) __attribute__((always_inline)) {
    const __bf16* const src = pf + (size_t)e.kt * kt_stride;
    const unsigned dst = lds0 + bs_off + slot * bs_bytes;
    for (int j = wave; j < npiece; j += Y_WAVES) glds16(src + j * 512, dst + j * 1024);
  };
  auto issue_x = [&](const YStep& e) __attribute__((always_inline)) -> int {
    const float* const xb = xg + e.x_off + xrow;
    unsigned dst = lds0 + xs_off + ((e.flags >> 2) & 1) * xs_bytes;
    const int d1 = (e.flags >> 4) & 15;
    int n = 0;
    for (int i = 0; i < d1; ++i, dst += 2048, n += 2) {
      glds16(xb + i * e.mul, dst);
      glds16(xb + i * e.mul + 4, dst + 1024);
    }
    if (gate_on && e.g_off >= 0) {
      const float* const gb = xg + e.g_off + xrow;
      glds16(gb, dst);
      glds16(gb + 4, dst + 1024);
      n += 2;
    }
    return n;
  };
  auto issue_w = [&](const YStep& e, const int slot) __attribute__((always_inline)) {
    const float* const wb = wg + e.w_off + wrow;
    const unsigned dst = lds0 + ws_off + slot * 2048;
    glds16(wb, dst);
    glds16(wb + 4, dst + 1024);
  };

  // E0 / E1 / E2: the entries of steps s, s + 1, s + 2 (clamped to the last step: never used past it)
  YStep E0 = y_entry(sy_lds, 0), E1 = y_entry(sy_lds, min(1, nsteps - 1)), E2 = y_entry(sy_lds, min(2, nsteps - 1));
  // prologue: queue = x(chunk of step 0), w(0), B(0), x(chunk of step 1)?, w(1)
  int nw_last = 0, nx_last = 0;
  issue_x(E0);
  if (has_w) issue_w(E0, 0);
  issue_b(E0, 0);
  if (nsteps > 1) {
    if (E1.flags & 1) nx_last = issue_x(E1);
    if (has_w) issue_w(E1, 1), nw_last = 2;
  }

  float xf[XD][8];
  int s3 = 0;  // s % 3
  YT_STAMP(0);  // table + prologue issue
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    const bool chunk_first = (E0.flags & 1) != 0;
    const int d1 = (E0.flags >> 4) & 15;
    wait_vmcnt(nw_last + nx_last);
    YT_STAMP(1);  // operand wait
    __builtin_amdgcn_s_barrier();
    YT_STAMP(2);  // barrier
    // the entry of step s + 3 (a broadcast LDS read, consumed at the end of the iteration)
    const int4* const tq = reinterpret_cast<const int4*>(sy_lds) + 2 * min(s + 3, nsteps - 1);
    const int4 ta = tq[0], tb = tq[1];
    if ((E0.flags & 2) && (!(EQF_Y_ABLATE & 4) || s == 0)) {  // a new input segment: its coupling block (ordinary loads; three times per item)
      wave_lds_order();
      stage_m(Mt, MS, cpl + (E0.m_blk & 0xffff), m_ld, e0, E - 1, E0.m_blk >> 16, r, hi);
      wave_lds_order();
    }
    if (chunk_first) {  // this chunk's x rows leave their slot before the DMA of the chunk after next may land in it
      const float* const xs = sy_lds + ((xs_off + ((E0.flags >> 2) & 1) * xs_bytes) >> 2) + lane * 4;
      auto ldx = [&](auto tag) __attribute__((always_inline)) {
        constexpr int D1 = decltype(tag)::value;
#pragma unroll
        for (int i = 0; i < D1; ++i) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(xs + i * 512);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(xs + i * 512 + 256);
#pragma unroll
          for (int j = 0; j < 4; ++j) xf[i][j] = a0[j], xf[i][4 + j] = a1[j];
        }
#pragma unroll
        for (int i = D1; i < XD; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[i][j] = 0.f;
        if (gate_on) {  // uniform: the rows are the gate's INPUT, activate / gate them here
          if (E0.g_off == -1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[0][j] = c_silu * xf[0][j] * xg_sigmoid(xf[0][j]);
          } else if (E0.g_off >= 0) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(xs + D1 * 512);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(xs + D1 * 512 + 256);
            float sg[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) sg[j] = c_sig * xg_sigmoid(g0[j]), sg[4 + j] = c_sig * xg_sigmoid(g1[j]);
#pragma unroll
            for (int i = 0; i < D1; ++i)
#pragma unroll
              for (int j = 0; j < 8; ++j) xf[i][j] *= sg[j];
          }
        }
      };
      switch (d1) {
        case 1: ldx(IC<1>()); break;
        case 3: ldx(IC<3>()); break;
        default: ldx(IC<5>()); break;
      }
#pragma unroll
      for (int i = 0; i < XD; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(xf[i][j]));  // (the reads have returned: lgkmcnt(0) before the DMAs below)
    }
    nx_last = 0, nw_last = 0;
    if (s + 1 < nsteps && !(EQF_Y_ABLATE & 32)) issue_b(E1, (s + 1) & 1);
    if (s + 2 < nsteps && !(EQF_Y_ABLATE & 16)) {
      if (E2.flags & 1) nx_last = issue_x(E2);
      if (has_w) {
        issue_w(E2, s3 == 0 ? 2 : s3 - 1);  // (s + 2) % 3
        nw_last = 2;
      }
    }

    YT_STAMP(3);  // coupling block, x rows out of their slot, DMA issue
    float wf[8];
    if (has_w) {
      const float* const ws = sy_lds + ((ws_off + s3 * 2048) >> 2) + lane * 4;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ws);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(ws + 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = valid ? a0[j] : 0.f, wf[4 + j] = valid ? a1[j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[j] = valid ? 1.f : 0.f;
    }
    const float* const mp = Mt + r * MS + E0.m_rel;
    const __bf16* const bs = reinterpret_cast<const __bf16*>(sy_lds) + ((bs_off + (s & 1) * bs_bytes) >> 1) + lane * 8;
#pragma unroll
    for (int m3 = 0; m3 < D3; ++m3) {
      float a[8];
      auto gen = [&](auto tag) __attribute__((always_inline)) {
        constexpr int D1 = decltype(tag)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0.f;
#pragma unroll
        for (int i = 0; i < D1; ++i) {
          const float m = mp[i * D3 + m3];
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = fmaf(m, xf[i][j], a[j]);
        }
      };
#if EQF_Y_ABLATE & 2
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = xf[0][j] + mp[m3];
#else
      switch (d1) {
        case 1: gen(IC<1>()); break;
        case 3: gen(IC<3>()); break;
        default: gen(IC<5>()); break;
      }
#endif
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
#if EQF_Y_ABLATE & 1
#pragma unroll
        for (int pl = 0; pl < NPW; ++pl) asm volatile("" ::"v"(bw[pl]));
#pragma unroll
        for (int pl = 0; pl < NPA; ++pl) asm volatile("" ::"v"(pa[pl]));
#else
        mma_terms<NPA, NPW>(pa, bw, acc[m3][ct]);
#endif
      }
    }

    YT_STAMP(4);  // generation + matrix instructions (issue)
    s3 = s3 == 2 ? 0 : s3 + 1;
    E0 = E1, E1 = E2;
    E2.x_off = __builtin_amdgcn_readfirstlane(ta.x), E2.w_off = __builtin_amdgcn_readfirstlane(ta.y);
    E2.kt = __builtin_amdgcn_readfirstlane(ta.z), E2.m_rel = __builtin_amdgcn_readfirstlane(ta.w);
    E2.flags = __builtin_amdgcn_readfirstlane(tb.x), E2.mul = __builtin_amdgcn_readfirstlane(tb.y);
    E2.g_off = __builtin_amdgcn_readfirstlane(tb.z), E2.m_blk = __builtin_amdgcn_readfirstlane(tb.w);
  }

  // (the accumulators leave the loop IN the accumulator file: without this hipcc copies all of them to VGPRs at the end of every
  // iteration -- 96 v_accvgpr_read per step, each waiting for the matrix instruction before it -- for the benefit of the stores)
#pragma unroll
  for (int m3 = 0; m3 < D3; ++m3)
#pragma unroll
    for (int ct = 0; ct < CTM; ++ct) asm volatile("" : "+a"(acc[m3][ct]));
  YT_STAMP(5);  // accumulators retired
  // epilogue: accumulator register q of lane (r, hi) = row (edge) (q & 3) + 8 (q >> 2) + 4 hi, column r of the tile
#pragma unroll
  for (int ct = 0; ct < CTM; ++ct) {
    if (ct >= CT) continue;
    const int c0 = (ct0 + ct) * 32;  // first column of the tile in the concatenated [main | second] output
    const bool main = c0 < D.N1;     // scalar: N1 % 32 == 0 (a lane-dependent test turns ld into a per-lane LOAD from the kernarg
                                     // segment and every store of the tile into store -> s_waitcnt vmcnt(0) -> store)
    float bv = 0.f;
    if (D3 == 1) bv = main ? (f.bias ? f.bias[c0 + r] : 0.f) : (f.bias2 ? f.bias2[c0 + r - D.N1] : 0.f);
    asm volatile("" : "+v"(bv));  // the bias has ARRIVED here: otherwise every masked store below carries its own s_waitcnt vmcnt(0),
                                  // which also waits for the store before it
    float* const base = main ? f.o1 + D.out1_off + c0 : f.o2 + (c0 - D.N1);  // uniform
    const unsigned ld = main ? f.ld1 : f.ld2;
#pragma unroll
    for (int m3 = 0; m3 < D3; ++m3)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hi;
        if (e0 + row < f.E && !(EQF_Y_ABLATE & 8)) base[(unsigned)(e0 + row) * ld + (unsigned)(m3 * D.N1 + r)] = acc[m3][ct][q] + bv;
      }
  }
#if EQF_Y_TRACE
  YT_STAMP(6);  // stores issued
  if (grp == 3 && threadIdx.x == 0)
    printf("ytrace d3 %d ct %d steps %d: total %lld | prologue %lld wait %lld barrier %lld stage+issue %lld compute %lld retire %lld stores %lld\n",
           D3, CT, nsteps, clock64() - yt_begin, yt[0], yt[1], yt[2], yt[3], yt[4], yt[5], yt[6]);
#endif
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void sfcy_fwd_kernel(const YFwdArgs g_byval) {
  KERNARG_IN_PLACE(YFwdArgs);
  // item-major launch order, heaviest item type first (the host sorts the types): grp fastest
  const int y = blockIdx.x / g.ngrp, grp = blockIdx.x - y * g.ngrp;
  const YType& T = g.type[y];
  switch (g.f.deg[T.deg].d3) {
    case 1: yf_item<1, MODE>(g, T, grp); break;
    case 3: yf_item<3, MODE>(g, T, grp); break;
    default: yf_item<5, MODE>(g, T, grp); break;
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
  A.has_w = C.w != nullptr, A.pad = 0;
  A.ngrp = eqf_cdiv(eqf_cdiv(C.E, 32), Y_WAVES);
  int nt = 0;
  long cost[Y_MAXTYPE];
  size_t ldsmax = 0;
  for (int d = 0; d < C.ndeg; ++d) {
    const XFwdArgs::Deg& X = A.f.deg[d];
    if (X.d3 != 1 && X.d3 != 3 && X.d3 != 5) return EQF_E_UNSUPPORTED;
    int d1max = 1, mlen = 1, gpiece = 0;
    long steps = 0;
    for (int si = 0; si < X.nseg; ++si) {
      const XSeg& S = X.seg[si];
      if (S.d1 != 1 && S.d1 != 3 && S.d1 != 5) return EQF_E_UNSUPPORTED;
      if (S.mul % 16 != 0 || (S.x_off & 3)) return EQF_E_UNSUPPORTED;
      if (S.g_off >= 0 && (S.g_off & 3)) return EQF_E_UNSUPPORTED;
      d1max = S.d1 > d1max ? S.d1 : d1max;
      mlen = S.m_len > mlen ? S.m_len : mlen;
      gpiece |= (A.f.gate.on && S.g_off >= 0) ? 1 : 0;
      steps += (long)S.npath * (S.mul / 16);
      for (int q = 0; q < S.npath; ++q)
        if ((S.p[q].w_off & 3) || (S.p[q].kbase & 15)) return EQF_E_UNSUPPORTED;
      if (S.m_off > 0xffff || S.m_len > 0x7fff) return EQF_E_UNSUPPORTED;
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
      if (steps > Y_MAXSTEP) return EQF_E_UNSUPPORTED;
      const size_t need = (size_t)Y_TAB_BYTES + (size_t)2 * cn * npw * 1024 + (size_t)Y_WAVES * T.wave_bytes;
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
  nblk = A.ngrp * nt;
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
    hipLaunchKernelGGL((sfcy_fwd_kernel<M>), dim3(nblk), dim3(64 * Y_WAVES), lds, st, A);                            \
  } while (0)
  if (mode == 0) YF_LAUNCH(0);
  else if (mode == 1) YF_LAUNCH(1);
  else YF_LAUNCH(2);
#undef YF_LAUNCH
  return 0;
}
