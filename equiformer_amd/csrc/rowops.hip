// Row-local feature ops of the Equiformer hot path (HBM-bound): equivariant layer norm, gate / SiLU
// activations, the radial MLP's LayerNorm+SiLU, atom-type embedding and column sums (bias gradients).
// One wavefront (64 lanes) owns one row; rows are 0.25-3.5 KB, so a row is read once into L1/registers
// and written once -- algorithmic traffic = 2 x rows x D x 4 bytes.
#include "common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;

struct SegTab {
  int nseg;
  int off[EQF_MAX_SEG];   // row offset
  int len[EQF_MAX_SEG];   // mul * (2l+1)
  int mul[EQF_MAX_SEG];
  int l[EQF_MAX_SEG];
  int woff[EQF_MAX_SEG];  // offset into affine_weight
  int boff[EQF_MAX_SEG];  // offset into affine_bias (l == 0 only, else -1)
  int D;
};

SegTab make_segtab(const eqf_irreps& ir) {
  SegTab t{};
  t.nseg = ir.nseg;
  int off = 0, w = 0, b = 0;
  for (int s = 0; s < ir.nseg; ++s) {
    t.off[s] = off;
    t.mul[s] = ir.mul[s];
    // "l == 0" below means an invariant scalar (0e): mean subtraction and bias.  A pseudo-scalar segment (0o, E(3) models)
    // is normalised like any l > 0 segment [ref: nets/layer_norm.py EquivariantLayerNormV2: `ir.l == 0 and ir.p == 1`]
    const bool scalar = ir.l[s] == 0 && !ir.odd[s];
    t.l[s] = (ir.l[s] == 0 && ir.odd[s]) ? -1 : ir.l[s];
    t.len[s] = ir.mul[s] * (2 * ir.l[s] + 1);
    t.woff[s] = w;
    t.boff[s] = scalar ? b : -1;
    w += ir.mul[s];
    if (scalar) b += ir.mul[s];
    off += t.len[s];
  }
  t.D = off;
  return t;
}

// ---------------------------------------------------------------------------------------------- layer norm
// x2 / xsum optional: the residual add in front of the norm (x + x2 is normalised and also written to xsum, which the
// next residual connection consumes) -- one pass instead of an element-wise add launch plus a re-read.  Every lane
// re-forms the sums it needs (no read-back of values another lane stored).
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                            float* __restrict__ xsum, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ rstd, float* __restrict__ mean0,
                                                            int rows, SegTab T, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * T.D;
  const float* x2r = x2 ? x2 + (long)row * T.D : nullptr;
  float* sr = x2 ? xsum + (long)row * T.D : nullptr;
  float* yr = y + (long)row * T.D;
  bool first0 = true;
  for (int s = 0; s < T.nseg; ++s) {
    const int off = T.off[s];
    const int n = T.len[s], mul = T.mul[s];
    float mean = 0.f;
    if (T.l[s] == 0) {
      float sum = 0.f;
      for (int i = lane; i < n; i += 64) sum += xr[off + i] + (x2r ? x2r[off + i] : 0.f);
      mean = wave_sum(sum) / n;
    }
    float sq = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float v = xr[off + i] + (x2r ? x2r[off + i] : 0.f) - mean;
      sq += v * v;
    }
    const float rs = rsqrtf(wave_sum(sq) / n + eps);
    if (lane == 0) {
      rstd[(long)row * T.nseg + s] = rs;
      if (T.l[s] == 0 && first0) mean0[row] = mean;
    }
    if (T.l[s] == 0) first0 = false;
    const float* ws = w + T.woff[s];
    for (int i = lane; i < n; i += 64) {
      const int u = i % mul;
      const float xv = xr[off + i] + (x2r ? x2r[off + i] : 0.f);
      if (sr) sr[off + i] = xv;
      float v = (xv - mean) * rs * ws[u];
      if (T.boff[s] >= 0) v += b[T.boff[s] + u];
      yr[off + i] = v;
    }
  }
}

// (Round 5 measured register-resident variants of the two kernels around this comment -- the row read once into 16 registers
// per lane, per-segment sums formed from the registers: 23.5 us against 13.8 / 16.4 us at 2 304 rows.  The run-time loops
// below are not the problem; the selects and index arithmetic of a segment-agnostic register layout cost more than the
// re-reads from L1 they save.  Patch and A/B: profiles/r05/r05_k_*.)
// Backward: dx = LN'(dy) (+ dres, the gradient arriving at the normalised sum from the residual branch); one wave per row.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ dy, const float* __restrict__ dres,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            int rows, SegTab T) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * T.D;
  const float* gr = dy + (long)row * T.D;
  const float* rr = dres ? dres + (long)row * T.D : nullptr;
  float* dr = dx + (long)row * T.D;
  for (int s = 0; s < T.nseg; ++s) {
    const float* xs = xr + T.off[s];
    const float* gs = gr + T.off[s];
    const float* ws = w + T.woff[s];
    const int n = T.len[s], mul = T.mul[s];
    const float rs = rstd[(long)row * T.nseg + s];
    float mean = 0.f;
    if (T.l[s] == 0) {
      float sum = 0.f;
      for (int i = lane; i < n; i += 64) sum += xs[i];
      mean = wave_sum(sum) / n;
    }
    float sg = 0.f, sgx = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float g = gs[i] * ws[i % mul];
      const float xh = (xs[i] - mean) * rs;
      sg += g;
      sgx += g * xh;
    }
    sgx = wave_sum(sgx) / n;
    sg = (T.l[s] == 0) ? wave_sum(sg) / n : 0.f;
    for (int i = lane; i < n; i += 64) {
      const float g = gs[i] * ws[i % mul];
      const float xh = (xs[i] - mean) * rs;
      float v = rs * (g - sg - xh * sgx);
      if (rr) v += rr[T.off[s] + i];
      dr[T.off[s] + i] = v;
    }
  }
}

// (Two restructurings of this reduction were measured in round 2 and dropped: folding it into the dx kernel with one
// atomic per (workgroup, column) -- 61 us at 2 304 rows, because same-address fp32 atomics retire at ~0.35 ns each
// (175 k of them) -- and a wave-per-row-block version with few fat workgroups -- 127 us, 64 waves cannot hide the load
// latency.  The column-per-thread kernel below stays at 25 us.)
// d_weight[woff+u] += sum_rows sum_m dy*xhat ; d_bias[boff+u] += sum_rows dy.   One thread per row column,
// each block reduces CH rows, atomics at the end (columns of the same channel collide only (2l+1) ways).
constexpr int LN_WGRAD_ROWS = 64, LN_WGRAD_CHUNK = 16;
__global__ __launch_bounds__(256) void layernorm_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ mean0, float* __restrict__ dw,
                                                              float* __restrict__ db, int rows, SegTab T) {
  constexpr int CH = LN_WGRAD_CHUNK;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= T.D) return;
  int s = 0;
  while (s + 1 < T.nseg && c >= T.off[s + 1]) ++s;
  const int u = (c - T.off[s]) % T.mul[s];
  const bool is0 = T.l[s] == 0;
  // LN_WGRAD_ROWS rows per thread in chunks of CH: the CH rows of a chunk are requested before the first is used
  // (compile-time trip count, rows past the end clamped and masked).  The cost of this kernel is its atomics (same-address
  // fp32 atomics retire at ~0.35 ns each): 64 rows per thread = 37 k of them at 2 304 rows instead of 147 k with 16.
  float aw = 0.f, ab = 0.f;
  for (int r0 = blockIdx.y * LN_WGRAD_ROWS; r0 < min(rows, (int)(blockIdx.y + 1) * LN_WGRAD_ROWS); r0 += CH) {
    float gv[CH], xs[CH], rs[CH], m0[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int r = min(r0 + k, rows - 1);
      gv[k] = dy[(long)r * T.D + c];
      xs[k] = x[(long)r * T.D + c];
      rs[k] = rstd[(long)r * T.nseg + s];
      m0[k] = is0 ? mean0[r] : 0.f;  // mean0 holds the mean of the first 0e segment; other 0e segments (none in practice) recompute
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const float g = (r0 + k < rows) ? gv[k] : 0.f;
      aw += g * (xs[k] - m0[k]) * rs[k];
      ab += g;
    }
  }
  atomicAdd(dw + T.woff[s] + u, aw);
  if (T.boff[s] >= 0) atomicAdd(db + T.boff[s] + u, ab);
}

// ---------------------------------------------------------------------------------------------- gate
struct GateTab {
  int S, G, nseg;
  int in_off[EQF_MAX_SEG], out_off[EQF_MAX_SEG], goff[EQF_MAX_SEG], mul[EQF_MAX_SEG], d[EQF_MAX_SEG];
  int Din, Dout;
};

GateTab make_gatetab(int S, const eqf_irreps& gated) {
  GateTab t{};
  t.S = S;
  t.nseg = gated.nseg;
  int G = 0;
  for (int s = 0; s < gated.nseg; ++s) G += gated.mul[s];
  t.G = G;
  int in_off = S + G, out_off = S, g = 0;
  for (int s = 0; s < gated.nseg; ++s) {
    t.in_off[s] = in_off, t.out_off[s] = out_off, t.goff[s] = S + g;
    t.mul[s] = gated.mul[s], t.d[s] = 2 * gated.l[s] + 1;
    in_off += t.mul[s] * t.d[s], out_off += t.mul[s] * t.d[s], g += t.mul[s];
  }
  t.Din = in_off, t.Dout = out_off;
  return t;
}

__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, long total,
                                                       GateTab T, float c_silu, float c_sig) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long row = idx / T.Dout;
  const int c = (int)(idx - row * T.Dout);
  const float* ir = in + row * T.Din;
  float v;
  if (c < T.S) {
    const float s = ir[c];
    v = c_silu * s * sigmoidf_(s);
  } else {
    int sg = 0;
    while (sg + 1 < T.nseg && c >= T.out_off[sg + 1]) ++sg;
    const int j = c - T.out_off[sg];
    const int u = j % T.mul[sg];
    v = ir[T.in_off[sg] + j] * c_sig * sigmoidf_(ir[T.goff[sg] + u]);
  }
  out[idx] = v;
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ in, const float* __restrict__ d_out,
                                                       float* __restrict__ d_in, long total, GateTab T, float c_silu,
                                                       float c_sig) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per INPUT element
  if (idx >= total) return;
  const long row = idx / T.Din;
  const int c = (int)(idx - row * T.Din);
  const float* ir = in + row * T.Din;
  const float* gr = d_out + row * T.Dout;
  float v;
  if (c < T.S) {
    const float s = ir[c], sg = sigmoidf_(s);
    v = gr[c] * c_silu * (sg + s * sg * (1.f - sg));
  } else if (c < T.S + T.G) {
    int sg = 0;
    while (sg + 1 < T.nseg && c >= T.goff[sg + 1]) ++sg;
    const int u = c - T.goff[sg];
    const float gsig = sigmoidf_(ir[c]);
    float acc = 0.f;
    for (int m = 0; m < T.d[sg]; ++m) acc += gr[T.out_off[sg] + m * T.mul[sg] + u] * ir[T.in_off[sg] + m * T.mul[sg] + u];
    v = acc * c_sig * gsig * (1.f - gsig);
  } else {
    int sg = 0;
    while (sg + 1 < T.nseg && c >= T.in_off[sg + 1]) ++sg;
    const int j = c - T.in_off[sg];
    const int u = j % T.mul[sg];
    v = gr[T.out_off[sg] + j] * c_sig * sigmoidf_(ir[T.goff[sg] + u]);
  }
  d_in[idx] = v;
}

// Column-fixed variants: a thread owns ONE column and walks RB rows, so the 64-bit division idx / D, the segment search
// and the `% mul` of the element-per-thread kernels above (~150 VALU instructions per element, paid on the same lanes
// the fp32 MFMAs of concurrently running kernels would use) happen once per thread instead of once per element.
__global__ __launch_bounds__(256) void gate_fwd_cols_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                            GateTab T, float c_silu, float c_sig, int RB) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;  // row blocks run along grid.x (no 65 535 limit)
  if (c >= T.Dout) return;
  int ia = c, ib = -1;
  if (c >= T.S) {
    int sg = 0;
    while (sg + 1 < T.nseg && c >= T.out_off[sg + 1]) ++sg;
    const int j = c - T.out_off[sg];
    ia = T.in_off[sg] + j;
    ib = T.goff[sg] + j % T.mul[sg];
  }
  const int r0 = blockIdx.x * RB, r1 = min(rows, r0 + RB);
  const float* ir = in + (long)r0 * T.Din;
  float* orow = out + (long)r0 * T.Dout + c;
#pragma unroll 4
  for (int r = r0; r < r1; ++r, ir += T.Din, orow += T.Dout) {
    const float s = ir[ia];
    *orow = (ib < 0) ? c_silu * s * sigmoidf_(s) : s * c_sig * sigmoidf_(ir[ib]);
  }
}

__global__ __launch_bounds__(256) void gate_bwd_cols_kernel(const float* __restrict__ in, const float* __restrict__ d_out,
                                                            float* __restrict__ d_in, int rows, GateTab T, float c_silu,
                                                            float c_sig, int RB) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;  // one thread per INPUT column; row blocks along grid.x
  if (c >= T.Din) return;
  int kind = 0, ia = c, ib = 0, mul = 0, d = 0;  // 0 scalar, 1 gate, 2 gated
  if (c >= T.S && c < T.S + T.G) {
    kind = 1;
    int sg = 0;
    while (sg + 1 < T.nseg && c >= T.goff[sg + 1]) ++sg;
    const int u = c - T.goff[sg];
    ia = T.out_off[sg] + u, ib = T.in_off[sg] + u, mul = T.mul[sg], d = T.d[sg];
  } else if (c >= T.S + T.G) {
    kind = 2;
    int sg = 0;
    while (sg + 1 < T.nseg && c >= T.in_off[sg + 1]) ++sg;
    const int j = c - T.in_off[sg];
    ia = T.out_off[sg] + j, ib = T.goff[sg] + j % T.mul[sg];
  }
  const int r0 = blockIdx.x * RB, r1 = min(rows, r0 + RB);
  const float* ir = in + (long)r0 * T.Din;
  const float* gr = d_out + (long)r0 * T.Dout;
  float* drow = d_in + (long)r0 * T.Din + c;
#pragma unroll 2
  for (int r = r0; r < r1; ++r, ir += T.Din, gr += T.Dout, drow += T.Din) {
    float v;
    if (kind == 0) {
      const float s = ir[c], sg = sigmoidf_(s);
      v = gr[c] * c_silu * (sg + s * sg * (1.f - sg));
    } else if (kind == 1) {
      // (all 2 x (2l+1) loads of the gate column are requested before the first is used: with a run-time trip count every
      // iteration was a dependent memory round trip -- 10 per row for a degree-2 gate, 80 per thread -- and the kernel took
      // 27 us whatever the row count; indices past the degree are clamped and their products masked)
      const float gsig = sigmoidf_(ir[c]);
      float ga[7], xa[7];
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        const int mm = min(m, d - 1);
        ga[m] = gr[ia + mm * mul];
        xa[m] = ir[ib + mm * mul];
      }
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < 7; ++m) acc += (m < d) ? ga[m] * xa[m] : 0.f;
      v = acc * c_sig * gsig * (1.f - gsig);
    } else {
      v = gr[ia] * c_sig * sigmoidf_(ir[ib]);
    }
    *drow = v;
  }
}

// ---------------------------------------------------------------------------------------------- silu
__global__ __launch_bounds__(256) void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4,
                                                       long n, float c) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = c * v.x * sigmoidf_(v.x), v.y = c * v.y * sigmoidf_(v.y);
    v.z = c * v.z * sigmoidf_(v.z), v.w = c * v.w * sigmoidf_(v.w);
    reinterpret_cast<float4*>(y)[i] = v;
  }
  if (i == 0)
    for (long j = n4 * 4; j < n; ++j) y[j] = c * x[j] * sigmoidf_(x[j]);
}

__device__ __forceinline__ float dsilu(float s) {
  const float sg = sigmoidf_(s);
  return sg + s * sg * (1.f - sg);
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n4, long n, float c) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    float4 o;
    o.x = c * g.x * dsilu(v.x), o.y = c * g.y * dsilu(v.y), o.z = c * g.z * dsilu(v.z), o.w = c * g.w * dsilu(v.w);
    reinterpret_cast<float4*>(dx)[i] = o;
  }
  if (i == 0)
    for (long j = n4 * 4; j < n; ++j) dx[j] = c * dy[j] * dsilu(x[j]);
}

// ---------------------------------------------------------------------------------------------- LN(C<=64) + SiLU
__global__ __launch_bounds__(256) void lnsilu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, int rows,
                                                         int C, float eps, int G) {
  // group g = blockIdx.y of a [rows][G][C] tensor with per-group affine parameters (G == 1: plain rows)
  const long ld = (long)G * C;
  {
    const int g_ = blockIdx.y * C;
    x += g_;
    gamma += g_;
    beta += g_;
    y += g_;
  }
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bool act = lane < C;
  const float v = act ? x[(long)row * ld + lane] : 0.f;
  const float mean = wave_sum(v) / C;
  const float d = act ? v - mean : 0.f;
  const float rs = rsqrtf(wave_sum(d * d) / C + eps);
  if (act) {
    const float z = d * rs * gamma[lane] + beta[lane];
    y[(long)row * ld + lane] = z * sigmoidf_(z);
  }
}

__global__ __launch_bounds__(256) void lnsilu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ dy,
                                                         float* __restrict__ dx, float* __restrict__ d_gamma,
                                                         float* __restrict__ d_beta, int rows, int C, float eps, int G) {
  // group g = blockIdx.y of a [rows][G][C] tensor with per-group affine parameters (G == 1: plain rows)
  const long ld = (long)G * C;
  {
    const int g_ = blockIdx.y * C;
    x += g_;
    gamma += g_;
    beta += g_;
    dy += g_;
    dx += g_;
    d_gamma += g_;
    d_beta += g_;
  }
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * WAVES_PER_BLOCK;
  const bool act = lane < C;
  const float gm = act ? gamma[lane] : 0.f, bt = act ? beta[lane] : 0.f;
  float acc_g = 0.f, acc_b = 0.f;
  for (int row = wave_global; row < rows; row += nwaves) {
    const float v = act ? x[(long)row * ld + lane] : 0.f;
    const float mean = wave_sum(v) / C;
    const float d = act ? v - mean : 0.f;
    const float rs = rsqrtf(wave_sum(d * d) / C + eps);
    const float xh = d * rs;
    const float z = xh * gm + bt;
    const float dz = act ? dy[(long)row * ld + lane] * dsilu(z) : 0.f;
    acc_g += dz * xh;
    acc_b += dz;
    const float g = dz * gm;
    const float sg = wave_sum(g) / C;
    const float sgx = wave_sum(g * xh) / C;
    if (act) dx[(long)row * ld + lane] = rs * (g - sg - xh * sgx);
  }
  __shared__ float red_g[WAVES_PER_BLOCK][64], red_b[WAVES_PER_BLOCK][64];
  red_g[threadIdx.x >> 6][lane] = acc_g;
  red_b[threadIdx.x >> 6][lane] = acc_b;
  __syncthreads();
  if (threadIdx.x < 64 && threadIdx.x < C) {
    float a = 0.f, b = 0.f;
    for (int wv = 0; wv < WAVES_PER_BLOCK; ++wv) a += red_g[wv][threadIdx.x], b += red_b[wv][threadIdx.x];
    atomicAdd(d_gamma + threadIdx.x, a);
    atomicAdd(d_beta + threadIdx.x, b);
  }
}

// C % 4 == 0: 16 lanes x float4 per row -> four rows per wave instruction, U independent row groups in flight (the
// one-row-per-wave kernel above is a chain of four 64-lane reductions per row), and 16x fewer atomics on d_gamma/d_beta.
__device__ __forceinline__ float sum16(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int U>
__global__ __launch_bounds__(256) void lnsilu_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ dy,
                                                          float* __restrict__ dx, float* __restrict__ d_gamma,
                                                          float* __restrict__ d_beta, int rows, int C, float eps, int G) {
  // group g = blockIdx.y of a [rows][G][C] tensor with per-group affine parameters (G == 1: plain rows)
  const long ld = (long)G * C;
  {
    const int g_ = blockIdx.y * C;
    x += g_;
    gamma += g_;
    beta += g_;
    dy += g_;
    dx += g_;
    d_gamma += g_;
    d_beta += g_;
  }
  const int lane = threadIdx.x & 63, sub = lane & 15, rw = lane >> 4;
  const int wave_global = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * WAVES_PER_BLOCK;
  const int c0 = sub * 4;
  const bool act = c0 < C;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 gm = act ? *reinterpret_cast<const float4*>(gamma + c0) : zero4;
  const float4 bt = act ? *reinterpret_cast<const float4*>(beta + c0) : zero4;
  const float invC = 1.f / (float)C;
  float4 ag = zero4, ab = zero4;
  for (int base = wave_global * 4 * U; base < rows; base += nwaves * 4 * U) {
    float4 v[U], g[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + u * 4 + rw;
      ok[u] = act && row < rows;
      v[u] = ok[u] ? *reinterpret_cast<const float4*>(x + (long)row * ld + c0) : zero4;
      g[u] = ok[u] ? *reinterpret_cast<const float4*>(dy + (long)row * ld + c0) : zero4;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float mean = sum16(v[u].x + v[u].y + v[u].z + v[u].w) * invC;
      float4 d;
      d.x = act ? v[u].x - mean : 0.f, d.y = act ? v[u].y - mean : 0.f;
      d.z = act ? v[u].z - mean : 0.f, d.w = act ? v[u].w - mean : 0.f;
      const float rs = rsqrtf(sum16(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * invC + eps);
      float4 xh, dz, gg;
      xh.x = d.x * rs, xh.y = d.y * rs, xh.z = d.z * rs, xh.w = d.w * rs;
      dz.x = g[u].x * dsilu(xh.x * gm.x + bt.x), dz.y = g[u].y * dsilu(xh.y * gm.y + bt.y);
      dz.z = g[u].z * dsilu(xh.z * gm.z + bt.z), dz.w = g[u].w * dsilu(xh.w * gm.w + bt.w);
      ag.x += dz.x * xh.x, ag.y += dz.y * xh.y, ag.z += dz.z * xh.z, ag.w += dz.w * xh.w;
      ab.x += dz.x, ab.y += dz.y, ab.z += dz.z, ab.w += dz.w;
      gg.x = dz.x * gm.x, gg.y = dz.y * gm.y, gg.z = dz.z * gm.z, gg.w = dz.w * gm.w;
      const float sg = sum16(gg.x + gg.y + gg.z + gg.w) * invC;
      const float sgx = sum16(gg.x * xh.x + gg.y * xh.y + gg.z * xh.z + gg.w * xh.w) * invC;
      if (ok[u]) {
        float4 o;
        o.x = rs * (gg.x - sg - xh.x * sgx), o.y = rs * (gg.y - sg - xh.y * sgx);
        o.z = rs * (gg.z - sg - xh.z * sgx), o.w = rs * (gg.w - sg - xh.w * sgx);
        *reinterpret_cast<float4*>(dx + (long)(base + u * 4 + rw) * ld + c0) = o;
      }
    }
  }
  // fold the four row groups of the wave, then the waves of the block
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    ag.x += __shfl_xor(ag.x, o), ag.y += __shfl_xor(ag.y, o), ag.z += __shfl_xor(ag.z, o), ag.w += __shfl_xor(ag.w, o);
    ab.x += __shfl_xor(ab.x, o), ab.y += __shfl_xor(ab.y, o), ab.z += __shfl_xor(ab.z, o), ab.w += __shfl_xor(ab.w, o);
  }
  __shared__ float red[WAVES_PER_BLOCK][2][64];
  if (lane < 16) {
    float* rg = red[threadIdx.x >> 6][0] + c0;
    float* rb = red[threadIdx.x >> 6][1] + c0;
    rg[0] = ag.x, rg[1] = ag.y, rg[2] = ag.z, rg[3] = ag.w;
    rb[0] = ab.x, rb[1] = ab.y, rb[2] = ab.z, rb[3] = ab.w;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
    if (c < C) {
      float a = 0.f;
      for (int wv = 0; wv < WAVES_PER_BLOCK; ++wv) a += red[wv][which][c];
      atomicAdd((which ? d_beta : d_gamma) + c, a);
    }
  }
}

// ---------------------------------------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int* __restrict__ type, const float* __restrict__ W,
                                                        const float* __restrict__ b, float* __restrict__ y, long total,
                                                        int C, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long row = idx / D;
  const int c = (int)(idx - row * D);
  y[idx] = (c < C) ? W[(long)type[row] * C + c] + (b ? b[c] : 0.f) : 0.f;
}

// dW[type[r], c] += dy[r, c], db[c] += dy[r, c].  A thread owns one column of a chunk of CH rows; the atom type of a row is the
// same for every thread of the workgroup, so the running sums per type live in a small LDS table indexed by a wave-uniform slot
// (EMB_SLOTS distinct types per chunk; the table is flushed when a chunk shows more) and every (chunk, type, column) costs ONE
// atomic (until round 5: one atomic per (row, column) -- 2 304 x 128 of them onto the 5 rows of a QM9 embedding).
constexpr int EMB_SLOTS = 16;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ type, const float* __restrict__ dy,
                                                        float* __restrict__ dW, float* __restrict__ db, int rows, int C,
                                                        int D, int CH) {
  __shared__ float acc[EMB_SLOTS][256];
  __shared__ int key[EMB_SLOTS];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * blockDim.x + tid;
  const bool act = c < C;
  const int r0 = blockIdx.y * CH, r1 = min(rows, r0 + CH);
  int nkeys = 0;  // uniform
  float ab = 0.f;
  auto flush = [&]() {
    for (int k = 0; k < nkeys; ++k) {
      if (act) atomicAdd(dW + (long)key[k] * C + c, acc[k][tid]);
    }
    nkeys = 0;
  };
  constexpr int RB = 8;  // rows whose loads are in flight together (one load per iteration of a run-time loop = one dependent
                         // memory round trip per row: 64 of them were the 43 us of this kernel, not its atomics)
  for (int rb = r0; rb < r1; rb += RB) {
    float g[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int r = min(rb + j, r1 - 1);
      g[j] = dy[(long)r * D + (act ? c : 0)];
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if (rb + j >= r1) break;  // uniform
      const int t = __builtin_amdgcn_readfirstlane(type[rb + j]);
      int slot = -1;
      for (int k = 0; k < nkeys; ++k) slot = (key[k] == t) ? k : slot;
      if (slot < 0) {
        if (nkeys == EMB_SLOTS) flush();
        slot = nkeys++;
        __syncthreads();  // (uniform branch: every thread of the workgroup sees the same types)
        if (tid == 0) key[slot] = t;
        acc[slot][tid] = 0.f;
        __syncthreads();
      }
      const float gv = act ? g[j] : 0.f;
      ab += gv;
      acc[slot][tid] += gv;
    }
  }
  flush();
  if (db && act) atomicAdd(db + c, ab);
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int d, int ld, int inner, int R, int N,
                                                     float* __restrict__ out, int RCH) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * RCH, r1 = min(R, r0 + RCH);
  float acc = 0.f;
  if (c < N) {
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 4) acc += X[row_off2(r, d, ld, inner) + c];
  }
  __shared__ float red[4][64];
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < N) atomicAdd(out + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}


// ---------------------------------------------------------------------------------------------- weight folding
// SeparableFCTP with shared (internal) depth-wise weights [ref: nets/graph_attention_transformer.py:449-451 sep_value]:
// Linear(DTP_w(x)) == Linear'(DTP_1(x)) with W'[row, :] = w[w_of_row[row]] * W[row, :] -- the fold and its two gradients.
// W is the flat per-degree [K(l), N(l)] weight; row_start[r] = first element of row r.
__global__ __launch_bounds__(256) void fold_fwd_kernel(const float* __restrict__ W, const float* __restrict__ w,
                                                       const int* __restrict__ row_start, const int* __restrict__ w_of_row,
                                                       float* __restrict__ out, int rows) {
  const int r = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float s = w[w_of_row[r]];
  for (int i = row_start[r] + lane; i < row_start[r + 1]; i += 64) out[i] = W[i] * s;
}

// dW[i] = g[i] * w_row ; dw[w_of_row[r]] = <g[row], W[row]>  (rows and shared weights are in one-to-one correspondence)
__global__ __launch_bounds__(256) void fold_bwd_kernel(const float* __restrict__ W, const float* __restrict__ w,
                                                       const int* __restrict__ row_start, const int* __restrict__ w_of_row,
                                                       const float* __restrict__ g, float* __restrict__ dW,
                                                       float* __restrict__ dw, int rows) {
  const int r = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const int j = w_of_row[r];
  const float s = w[j];
  float acc = 0.f;
  for (int i = row_start[r] + lane; i < row_start[r + 1]; i += 64) {
    const float gv = g[i];
    dW[i] = gv * s;
    acc = fmaf(gv, W[i], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) dw[j] = acc;
}

}  // namespace

extern "C" {

// EQF_SOURCE_HASH = equiformer_amd.build.source_hash() of the sources this object was compiled from (build.py passes it):
// lib.load() compares it with the hash of the sources beside the library and refuses a stale binary.
#ifndef EQF_SOURCE_HASH
#define EQF_SOURCE_HASH "unknown"
#endif
const char* eqf_version(void) { return "equiformer_hip 0.1 gfx950 src=" EQF_SOURCE_HASH; }

int eqf_add_layernorm_fwd(const float* x, const float* x2, float* xsum, const float* weight, const float* bias, float* y,
                          float* rstd, float* mean0, int rows, const eqf_irreps* irreps, float eps, void* stream) {
  if (!x || !weight || !bias || !y || !rstd || !mean0 || !irreps || irreps->nseg < 1 || irreps->nseg > EQF_MAX_SEG)
    return EQF_E_BADARG;
  if ((x2 != nullptr) != (xsum != nullptr)) return EQF_E_BADARG;
  if (rows <= 0) return 0;
  const SegTab T = make_segtab(*irreps);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(eqf_cdiv(rows, WAVES_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, x, x2,
                     xsum, weight, bias, y, rstd, mean0, rows, T, eps);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd, float* mean0,
                      int rows, const eqf_irreps* irreps, float eps, void* stream) {
  return eqf_add_layernorm_fwd(x, nullptr, nullptr, weight, bias, y, rstd, mean0, rows, irreps, eps, stream);
}

int eqf_add_layernorm_bwd(const float* x, const float* weight, const float* dy, const float* dres, const float* rstd,
                          const float* mean0, float* dx, float* d_weight, float* d_bias, int rows,
                          const eqf_irreps* irreps, void* stream) {
  if (!x || !weight || !dy || !rstd || !mean0 || !dx || !irreps || irreps->nseg < 1 || irreps->nseg > EQF_MAX_SEG)
    return EQF_E_BADARG;
  if (rows <= 0) return 0;
  const SegTab T = make_segtab(*irreps);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(eqf_cdiv(rows, WAVES_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, x,
                     weight, dy, dres, rstd, dx, rows, T);
  EQF_CHECK_LAUNCH();
  if (d_weight && d_bias) {
    hipLaunchKernelGGL(layernorm_wgrad_kernel, dim3(eqf_cdiv(T.D, 256), eqf_cdiv(rows, LN_WGRAD_ROWS)), dim3(256), 0,
                       (hipStream_t)stream, x, dy, rstd, mean0, d_weight, d_bias, rows, T);
    EQF_CHECK_LAUNCH();
  }
  return 0;
}

int eqf_layernorm_bwd(const float* x, const float* weight, const float* dy, const float* rstd, const float* mean0,
                      float* dx, float* d_weight, float* d_bias, int rows, const eqf_irreps* irreps, void* stream) {
  return eqf_add_layernorm_bwd(x, weight, dy, nullptr, rstd, mean0, dx, d_weight, d_bias, rows, irreps, stream);
}

int eqf_gate_fwd(const float* in, float* out, int rows, int S, const eqf_irreps* gated, float c_silu, float c_sig,
                 void* stream) {
  if (!in || !out || !gated) return EQF_E_BADARG;
  if (rows <= 0) return 0;
  const GateTab T = make_gatetab(S, *gated);
  const int RB = 8;
  hipLaunchKernelGGL(gate_fwd_cols_kernel, dim3(eqf_cdiv(rows, RB), eqf_cdiv(T.Dout, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, rows, T, c_silu, c_sig, RB);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_gate_bwd(const float* in, const float* d_out, float* d_in, int rows, int S, const eqf_irreps* gated,
                 float c_silu, float c_sig, void* stream) {
  if (!in || !d_out || !d_in || !gated) return EQF_E_BADARG;
  for (int s = 0; s < gated->nseg; ++s)
    if (gated->l[s] > 3) return EQF_E_UNSUPPORTED;  // the gate-scalar gradient of gate_bwd_cols_kernel holds 2 l + 1 <= 7 components
  if (rows <= 0) return 0;
  const GateTab T = make_gatetab(S, *gated);
  const int RB = 8;
  hipLaunchKernelGGL(gate_bwd_cols_kernel, dim3(eqf_cdiv(rows, RB), eqf_cdiv(T.Din, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, d_out, d_in, rows, T, c_silu, c_sig, RB);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_silu_fwd(const float* x, float* y, long n, float c, void* stream) {
  if (!x || !y) return EQF_E_BADARG;
  if (n <= 0) return 0;
  const long n4 = n / 4;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(eqf_cdiv(n4 > 0 ? n4 : 1, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n4,
                     n, c);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_silu_bwd(const float* x, const float* dy, float* dx, long n, float c, void* stream) {
  if (!x || !dy || !dx) return EQF_E_BADARG;
  if (n <= 0) return 0;
  const long n4 = n / 4;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(eqf_cdiv(n4 > 0 ? n4 : 1, 256)), dim3(256), 0, (hipStream_t)stream, x, dy,
                     dx, n4, n, c);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_lnsilu_group_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, int groups,
                         float eps, void* stream) {
  if (!x || !gamma || !beta || !y || C < 1 || groups < 1 || groups > 65535) return EQF_E_BADARG;
  if (C > 64) return EQF_E_UNSUPPORTED;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(lnsilu_fwd_kernel, dim3(eqf_cdiv(rows, WAVES_PER_BLOCK), groups), dim3(256), 0, (hipStream_t)stream,
                     x, gamma, beta, y, rows, C, eps, groups);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_lnsilu_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, float eps,
                   void* stream) {
  return eqf_lnsilu_group_fwd(x, gamma, beta, y, rows, C, 1, eps, stream);
}

int eqf_lnsilu_bwd(const float* x, const float* gamma, const float* beta, const float* dy, float* dx, float* d_gamma,
                   float* d_beta, int rows, int C, float eps, void* stream) {
  return eqf_lnsilu_group_bwd(x, gamma, beta, dy, dx, d_gamma, d_beta, rows, C, 1, eps, stream);
}

int eqf_lnsilu_group_bwd(const float* x, const float* gamma, const float* beta, const float* dy, float* dx,
                         float* d_gamma, float* d_beta, int rows, int C, int groups, float eps, void* stream) {
  if (!x || !gamma || !beta || !dy || !dx || !d_gamma || !d_beta || C < 1 || groups < 1 || groups > 65535)
    return EQF_E_BADARG;
  if (C > 64) return EQF_E_UNSUPPORTED;
  if (rows <= 0) return 0;
  if (C % 4 == 0) {
    constexpr int U = 2;  // 8 rows per wave step, 2 steps per wave
    int blocks = eqf_cdiv(rows, WAVES_PER_BLOCK * 4 * U * 2);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lnsilu_bwd4_kernel<U>, dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, dy,
                       dx, d_gamma, d_beta, rows, C, eps, groups);
    EQF_CHECK_LAUNCH();
    return 0;
  }
  int blocks = eqf_cdiv(rows, WAVES_PER_BLOCK * 2);  // two rows per wave: the loop is a chain of dependent reductions
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(lnsilu_bwd_kernel, dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, dy, dx,
                     d_gamma, d_beta, rows, C, eps, groups);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_embed_fwd(const int* type, const float* W, const float* b, float* y, int rows, int C, int D, void* stream) {
  if (!type || !W || !y || C > D) return EQF_E_BADARG;
  if (rows <= 0) return 0;
  const long total = (long)rows * D;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, type, W, b, y,
                     total, C, D);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_embed_bwd(const int* type, const float* dy, float* dW, float* db, int rows, int C, int D, void* stream) {
  if (!type || !dy || !dW) return EQF_E_BADARG;
  if (rows <= 0) return 0;
  const int CH = 64;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(eqf_cdiv(C, 256), eqf_cdiv(rows, CH)), dim3(256), 0, (hipStream_t)stream,
                     type, dy, dW, db, rows, C, D, CH);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_colsum(const float* X, eqf_rows rx, int R, int N, float* out, void* stream) {
  if (!X || !out || rx.d < 1) return EQF_E_BADARG;
  if (R <= 0 || N <= 0) return 0;
  const int RCH = 128;
  hipLaunchKernelGGL(colsum_kernel, dim3(eqf_cdiv(N, 64), eqf_cdiv(R, RCH)), dim3(256), 0, (hipStream_t)stream, X, rx.d,
                     rx.ld, rx.inner, R, N, out, RCH);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_fold_weight_fwd(const float* W, const float* w, const int* row_start, const int* w_of_row, float* out, int rows,
                        void* stream) {
  if (!W || !w || !row_start || !w_of_row || !out) return EQF_E_BADARG;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(fold_fwd_kernel, dim3(eqf_cdiv(rows, WAVES_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, W, w,
                     row_start, w_of_row, out, rows);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_fold_weight_bwd(const float* W, const float* w, const int* row_start, const int* w_of_row, const float* g,
                        float* dW, float* dw, int rows, void* stream) {
  if (!W || !w || !row_start || !w_of_row || !g || !dW || !dw) return EQF_E_BADARG;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(fold_bwd_kernel, dim3(eqf_cdiv(rows, WAVES_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, W, w,
                     row_start, w_of_row, g, dW, dw, rows);
  EQF_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
