// Second-derivative kernels of the eight non-linear operators of the hot path.
//
// MD17 training takes forces with create_graph=True and back-propagates a force loss
// [ref: nets/graph_attention_transformer_md17.py:318-325, main_md17.py:384-390], i.e. autograd differentiates the
// first backward pass.  For an operator y = f(x; theta) whose first-order backward produced dx = J_x^T dy, the second
// pass hands us c = d loss / d dx and needs the gradient of  Phi = <c, dx(x, theta, dy)>  wrt x, theta and dy:
//
//     g_dy = J_x c          g_x = d Phi / d x          g_theta = d Phi / d theta
//
// (the parameter gradients of the FIRST pass are not part of the force graph, so they carry no cotangent).  The
// multilinear operators reuse their first-order kernels for this (ops.py); the kernels below are the closed forms for
// the rest.  Each cites the reference code whose first-order kernel it differentiates.
#include "common.h"
#include "geom.h"

namespace {

constexpr int WPB = 4;  // waves per 256-thread block

__device__ __forceinline__ float silu_d1(float s, float sg) { return sg + s * sg * (1.f - sg); }
__device__ __forceinline__ float silu_d2(float s, float sg) { return sg * (1.f - sg) * (2.f + s * (1.f - 2.f * sg)); }

// ---------------------------------------------------------------------------------------------- scaled SiLU
// y = c0 silu(x)  [ref: nets/fast_activation.py:68-87 with normalize2mom]
__global__ __launch_bounds__(256) void silu_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ c, float* __restrict__ g_x,
                                                        float* __restrict__ g_dy, long n, float c0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = x[i], sg = sigmoidf_(s), ci = c[i] * c0;
  g_dy[i] = ci * silu_d1(s, sg);
  g_x[i] = ci * dy[i] * silu_d2(s, sg);
}

// ---------------------------------------------------------------------------------------------- gate
// [scalars | gates | gated] -> [c1 silu(scalars) | gated * c2 sigmoid(gates)]  [ref: nets/fast_activation.py:132-148]
struct GateTab2 {
  int S, G, nseg;
  int in_off[EQF_MAX_SEG], out_off[EQF_MAX_SEG], goff[EQF_MAX_SEG], mul[EQF_MAX_SEG], d[EQF_MAX_SEG];
  int Din, Dout;
};

GateTab2 make_gatetab2(int S, const eqf_irreps& gated) {
  GateTab2 t{};
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

// one thread per INPUT column, RB rows per thread.  c has the shape of the input (cotangent of d_in).
__global__ __launch_bounds__(256) void gate_bwd2_kernel(const float* __restrict__ in, const float* __restrict__ d_out,
                                                        const float* __restrict__ c, float* __restrict__ g_in,
                                                        float* __restrict__ g_dout, int rows, GateTab2 T, float c1,
                                                        float c2, int RB) {
  const int col = blockIdx.y * blockDim.x + threadIdx.x;
  if (col >= T.Din) return;
  int kind = 0, io = col, ix = 0, ig = 0, mul = 0, d = 0;  // 0 scalar, 1 gate, 2 gated
  if (col >= T.S && col < T.S + T.G) {
    kind = 1;
    int sg = 0;
    while (sg + 1 < T.nseg && col >= T.goff[sg + 1]) ++sg;
    const int u = col - T.goff[sg];
    io = T.out_off[sg] + u, ix = T.in_off[sg] + u, mul = T.mul[sg], d = T.d[sg];
  } else if (col >= T.S + T.G) {
    kind = 2;
    int sg = 0;
    while (sg + 1 < T.nseg && col >= T.in_off[sg + 1]) ++sg;
    const int j = col - T.in_off[sg];
    io = T.out_off[sg] + j, ig = T.goff[sg] + j % T.mul[sg];
  }
  const int r0 = blockIdx.x * RB, r1 = min(rows, r0 + RB);
  for (int r = r0; r < r1; ++r) {
    const float* ir = in + (long)r * T.Din;
    const float* cr = c + (long)r * T.Din;
    const float* gr = d_out + (long)r * T.Dout;
    float* gi = g_in + (long)r * T.Din;
    float* go = g_dout + (long)r * T.Dout;
    if (kind == 0) {
      const float s = ir[col], sg = sigmoidf_(s), cc = cr[col] * c1;
      go[col] = cc * silu_d1(s, sg);
      gi[col] = cc * gr[col] * silu_d2(s, sg);
    } else if (kind == 1) {
      // d_gate[u] = c2 sig'(g) sum_m x[m,u] dy[m,u]
      const float sg = sigmoidf_(ir[col]);
      const float s1 = sg * (1.f - sg), s2 = s1 * (1.f - 2.f * sg);
      float xdy = 0.f, cdy = 0.f;
      for (int m = 0; m < d; ++m) {
        const float dyv = gr[io + m * mul];
        xdy += ir[ix + m * mul] * dyv;
        cdy += cr[ix + m * mul] * dyv;
      }
      gi[col] = c2 * (s2 * cr[col] * xdy + s1 * cdy);
    } else {
      const float sg = sigmoidf_(ir[ig]);
      const float s1 = sg * (1.f - sg);
      const float cg = cr[ig];
      gi[col] = c2 * s1 * cg * gr[io];
      go[io] = c2 * (s1 * cg * ir[col] + sg * cr[col]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- LayerNorm(C <= 64) + SiLU
// y = silu(gamma xhat + beta)  [ref: nets/radial_func.py:13-36].  One wave per row.
__global__ __launch_bounds__(256) void lnsilu_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ dy,
                                                          const float* __restrict__ c, float* __restrict__ g_x,
                                                          float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                          float* __restrict__ g_dy, int rows, int C, float eps) {
  // blockIdx.y = group: rows are [groups][C] wide, every group has its own gamma / beta [C] (the radial bank, round 5)
  const int grp = blockIdx.y, ld = C * gridDim.y;
  gamma += grp * C, beta += grp * C, g_gamma += grp * C, g_beta += grp * C;
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * WPB;
  const bool act = lane < C;
  const float gm = act ? gamma[lane] : 0.f, bt = act ? beta[lane] : 0.f;
  const float invC = 1.f / (float)C;
  float acc_g = 0.f, acc_b = 0.f;
  for (int row = wave_global; row < rows; row += nwaves) {
    const long o = (long)row * ld + grp * C + lane;
    const float v = act ? x[o] : 0.f;
    const float mean = wave_sum(v) * invC;
    const float dv = act ? v - mean : 0.f;
    const float r = rsqrtf(wave_sum(dv * dv) * invC + eps);
    const float xh = dv * r;
    const float z = xh * gm + bt;
    const float sg = sigmoidf_(z);
    const float d1 = silu_d1(z, sg), d2 = silu_d2(z, sg);
    const float dyv = act ? dy[o] : 0.f, cv = act ? c[o] : 0.f;
    const float dz = dyv * d1;
    const float g = gm * dz;
    const float cbar = wave_sum(cv) * invC, gbar = wave_sum(g) * invC;
    const float p = wave_sum(cv * xh) * invC, q = wave_sum(g * xh) * invC;
    const float A = wave_sum(cv * g) * invC - cbar * gbar - p * q;
    const float chat = act ? cv - cbar - xh * p : 0.f;     // d Phi / d g = r chat
    const float e = gm * r * chat * d2 * dyv;               // d Phi / d z
    acc_g += dz * r * chat + e * xh;
    acc_b += e;
    const float t = act ? e * gm - r * (cv * q + p * g) : 0.f;  // d Phi / d xhat
    const float tbar = wave_sum(t) * invC, tx = wave_sum(t * xh) * invC;
    if (act) {
      g_dy[o] = d1 * gm * r * chat;
      g_x[o] = r * (t - tbar - xh * tx) - r * r * xh * A;
    }
  }
  __shared__ float red_g[WPB][64], red_b[WPB][64];
  red_g[threadIdx.x >> 6][lane] = acc_g;
  red_b[threadIdx.x >> 6][lane] = acc_b;
  __syncthreads();
  if ((int)threadIdx.x < C && threadIdx.x < 64) {
    float a = 0.f, b = 0.f;
    for (int wv = 0; wv < WPB; ++wv) a += red_g[wv][threadIdx.x], b += red_b[wv][threadIdx.x];
    atomicAdd(g_gamma + threadIdx.x, a);
    atomicAdd(g_beta + threadIdx.x, b);
  }
}

// ---------------------------------------------------------------------------------------------- equivariant layer norm
// [ref: nets/layer_norm.py:89-152]: per segment, 0e: x - mean over channels; y = xc * (mean(xc^2) + eps)^-1/2 * w_u (+ b).
struct SegTab2 {
  int nseg;
  int off[EQF_MAX_SEG], len[EQF_MAX_SEG], mul[EQF_MAX_SEG], l[EQF_MAX_SEG], woff[EQF_MAX_SEG];
  int D;
};

SegTab2 make_segtab2(const eqf_irreps& ir) {
  SegTab2 t{};
  t.nseg = ir.nseg;
  int off = 0, w = 0;
  for (int s = 0; s < ir.nseg; ++s) {
    t.off[s] = off, t.mul[s] = ir.mul[s], t.l[s] = (ir.l[s] == 0 && ir.odd[s]) ? -1 : ir.l[s];  // 0 = invariant scalar
    t.len[s] = ir.mul[s] * (2 * ir.l[s] + 1);
    t.woff[s] = w;
    w += ir.mul[s];
    off += t.len[s];
  }
  t.D = off;
  return t;
}

__global__ __launch_bounds__(256) void layernorm_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ dy, const float* __restrict__ c,
                                                             float* __restrict__ g_x, float* __restrict__ g_w,
                                                             float* __restrict__ g_dy, int rows, SegTab2 T, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
  // g_w: the rows of a workgroup first add into LDS (the 2 l + 1 components of a channel and the WPB rows meet there), then
  // ONE global atomic per channel and workgroup -- round 6: every element used to add straight into g_w, 80 k atomics on 224
  // addresses per launch of the MD17 step (168 rows): 38.8 us per launch, 13 launches per step
  constexpr int GW_S = 1024;
  __shared__ float gw_s[GW_S];
  const int nw = T.woff[T.nseg - 1] + T.mul[T.nseg - 1];
  const bool via_lds = nw <= GW_S;  // uniform
  if (via_lds) {
    for (int i = threadIdx.x; i < nw; i += blockDim.x) gw_s[i] = 0.f;
    __syncthreads();
  }
  const bool live = row < rows;
  const long base = (long)(live ? row : 0) * T.D;
  for (int s = 0; s < T.nseg && live; ++s) {
    const int n = T.len[s], mul = T.mul[s], off = T.off[s];
    const float invn = 1.f / (float)n;
    const float* xs = x + base + off;
    const float* gs = dy + base + off;
    const float* cs = c + base + off;
    const float* ws = w + T.woff[s];
    const bool is0 = T.l[s] == 0;
    float mean = 0.f;
    if (is0) {
      float sum = 0.f;
      for (int i = lane; i < n; i += 64) sum += xs[i];
      mean = wave_sum(sum) * invn;
    }
    float sq = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float v = xs[i] - mean;
      sq += v * v;
    }
    const float r = rsqrtf(wave_sum(sq) * invn + eps);
    float sc = 0.f, sgm = 0.f, sp = 0.f, sqq = 0.f, scg = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float xh = (xs[i] - mean) * r;
      const float g = gs[i] * ws[i % mul], cv = cs[i];
      sc += cv, sgm += g, sp += cv * xh, sqq += g * xh, scg += cv * g;
    }
    const float cbar = is0 ? wave_sum(sc) * invn : 0.f, gbar = is0 ? wave_sum(sgm) * invn : 0.f;
    const float p = wave_sum(sp) * invn, q = wave_sum(sqq) * invn;
    const float A = wave_sum(scg) * invn - cbar * gbar - p * q;
    float st = 0.f, stx = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float xh = (xs[i] - mean) * r;
      const float g = gs[i] * ws[i % mul], cv = cs[i];
      const float t = -r * (cv * q + p * g);
      st += t, stx += t * xh;
    }
    const float tbar = is0 ? wave_sum(st) * invn : 0.f, tx = wave_sum(stx) * invn;
    for (int i = lane; i < n; i += 64) {
      const int u = i % mul;
      const float xh = (xs[i] - mean) * r;
      const float wv = ws[u], dyv = gs[i], cv = cs[i];
      const float g = dyv * wv;
      const float chat = cv - cbar - xh * p;
      const float t = -r * (cv * q + p * g);
      g_dy[base + off + i] = wv * r * chat;
      g_x[base + off + i] = r * (t - tbar - xh * tx) - r * r * xh * A;
      if (via_lds) atomicAdd(gw_s + T.woff[s] + u, dyv * r * chat);
      else atomicAdd(g_w + T.woff[s] + u, dyv * r * chat);
    }
  }
  if (via_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nw; i += blockDim.x) atomicAdd(g_w + i, gw_s[i]);
  }
}

// ---------------------------------------------------------------------------------------------- attention logits
// logit[e,h] = sum_k c act(a[e,h,k]) alpha_dot[h,k], act = SmoothLeakyReLU(0.2)
// [ref: nets/graph_attention_transformer.py:54-63,506-507]
__device__ __forceinline__ float slrelu_d1(float x, float s) { return 0.6f + 0.4f * (2.f * s - 1.f) + 0.8f * x * s * (1.f - s); }
__device__ __forceinline__ float slrelu_d2(float x, float s) { return 0.8f * s * (1.f - s) * (2.f + x * (1.f - 2.f * s)); }

// column-per-thread, CH edges per thread; Kh consecutive lanes share a head (Kh in {8,16,32,64})
__global__ __launch_bounds__(256) void alpha_bwd2_kernel(const float* __restrict__ a, const float* __restrict__ adot,
                                                         const float* __restrict__ d_logit, const float* __restrict__ ca,
                                                         float* __restrict__ g_a, float* __restrict__ g_adot,
                                                         float* __restrict__ g_dlogit, int E, int HK, int Kh, float c,
                                                         int CH) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = col < HK;
  const int cc = ok ? col : HK - 1;
  const int h = cc / Kh, H = HK / Kh;
  const int e0 = blockIdx.y * CH, e1 = min(E, e0 + CH);
  const float ad = adot[cc];
  float acc = 0.f;
  for (int e = e0; e < e1; ++e) {
    const long o = (long)e * HK + cc;
    const float av = a[o], s = sigmoidf_(av);
    const float d1 = slrelu_d1(av, s), d2 = slrelu_d2(av, s);
    const float cv = ok ? ca[o] * c : 0.f;
    const float dl = d_logit[(long)e * H + h];
    if (ok) g_a[o] = cv * d2 * ad * dl;
    acc += cv * d1 * dl;
    float v = cv * d1 * ad;
    for (int w = Kh >> 1; w > 0; w >>= 1) v += __shfl_xor(v, w);
    if (ok && (cc % Kh) == 0) g_dlogit[(long)e * H + h] = v;
  }
  if (ok) atomicAdd(g_adot + col, acc);
}

// ---------------------------------------------------------------------------------------------- softmax + aggregate
// out[n,col] = sum_{e -> n} alpha[e,h(col)] keep[e,h] value[e,col],  alpha = segment softmax of the logits
// [ref: torch_geometric.utils.softmax + dropout + scatter, nets/graph_attention_transformer.py:508-514]
struct HeadTab2 {
  int nseg, H, D, G;  // G = float4 groups per head
  int off[EQF_MAX_SEG], mul[EQF_MAX_SEG], d[EQF_MAX_SEG], gcum[EQF_MAX_SEG + 1];
};

__device__ __forceinline__ int head_col2(const HeadTab2& T, int h, int g) {
  int s = 0;
  while (s + 1 < T.nseg && g >= T.gcum[s + 1]) ++s;
  const int mh = T.mul[s] / T.H;
  const int q4 = mh >> 2;
  const int j = g - T.gcum[s];
  const int m = j / q4, q = j - m * q4;
  return T.off[s] + m * T.mul[s] + h * mh + 4 * q;
}

__device__ __forceinline__ float keep_scale2(unsigned long long seed, unsigned long long idx, float p) {
  if (p <= 0.f) return 1.f;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  return (u >= p) ? 1.f / (1.f - p) : 0.f;
}

constexpr int MAX_SLOTS2 = 4;

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// one workgroup per destination node, one wave per head.  First-order backward (edge.hip attn_bwd_*):
//   d_value[e,col] = alpha_e keep_e d_out[col],  u_e = keep_e <value_e, d_out>,  d_logit_e = alpha_e (u_e - S), S = sum alpha u.
// With cotangents cv (of d_value) and cl (of d_logit), w_e = keep_e <cv_e, d_out>, T = sum cl alpha:
//   Phi = sum_e alpha_e w_e + sum_e cl_e alpha_e u_e - T S
//   b_e = d Phi / d alpha_e = w_e + cl_e u_e - cl_e S - T u_e,  B = sum alpha b  ->  g_logit_e = alpha_e (b_e - B)
//   k_e = alpha_e keep_e (cl_e - T)  ->  g_value[e,col] = k_e d_out[col],  g_dout[col] = sum_e alpha_e keep_e cv[e,col] + k_e value[e,col]
__global__ void attn_bwd2_kernel(const float* __restrict__ alpha, const float* __restrict__ value,
                                 const int* __restrict__ row_ptr, const float* __restrict__ d_out,
                                 const float* __restrict__ cv, const float* __restrict__ cl, float* __restrict__ g_logit,
                                 float* __restrict__ g_value, float* __restrict__ g_dout, const HeadTab2 T, float drop_p,
                                 unsigned long long seed) {
  const int n = blockIdx.x;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int beg = row_ptr[n], end = row_ptr[n + 1];
  const int H = T.H, D4 = T.D >> 2;
  int col4[MAX_SLOTS2];
  float4 go[MAX_SLOTS2], gacc[MAX_SLOTS2];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < MAX_SLOTS2; ++s) {
    const int g = lane + 64 * s;
    col4[s] = (g < T.G) ? head_col2(T, h, g) >> 2 : -1;
    go[s] = (col4[s] >= 0) ? reinterpret_cast<const float4*>(d_out)[(long)n * D4 + col4[s]] : z4;
    gacc[s] = z4;
  }
  // pass 1: S = sum alpha u, T = sum alpha cl, sum alpha w, sum alpha cl u
  float S = 0.f, Tt = 0.f, Saw = 0.f, Sacu = 0.f;
  for (int e = beg; e < end; ++e) {
    const float a = alpha[(long)e * H + h], clv = cl ? cl[(long)e * H + h] : 0.f;
    const float keep = keep_scale2(seed, (unsigned long long)e * H + h, drop_p);
    float pu = 0.f, pw = 0.f;
#pragma unroll
    for (int s = 0; s < MAX_SLOTS2; ++s)
      if (col4[s] >= 0) {
        pu += dot4(reinterpret_cast<const float4*>(value)[(long)e * D4 + col4[s]], go[s]);
        if (cv) pw += dot4(reinterpret_cast<const float4*>(cv)[(long)e * D4 + col4[s]], go[s]);
      }
    const float u = wave_sum(pu) * keep, w = wave_sum(pw) * keep;
    S = fmaf(a, u, S), Tt = fmaf(a, clv, Tt), Saw = fmaf(a, w, Saw), Sacu = fmaf(a * clv, u, Sacu);
  }
  const float B = Saw + Sacu - 2.f * S * Tt;
  // pass 2
  for (int e = beg; e < end; ++e) {
    const float a = alpha[(long)e * H + h], clv = cl ? cl[(long)e * H + h] : 0.f;
    const float keep = keep_scale2(seed, (unsigned long long)e * H + h, drop_p);
    float4 vv[MAX_SLOTS2], cc[MAX_SLOTS2];
    float pu = 0.f, pw = 0.f;
#pragma unroll
    for (int s = 0; s < MAX_SLOTS2; ++s) {
      vv[s] = z4, cc[s] = z4;
      if (col4[s] >= 0) {
        vv[s] = reinterpret_cast<const float4*>(value)[(long)e * D4 + col4[s]];
        if (cv) cc[s] = reinterpret_cast<const float4*>(cv)[(long)e * D4 + col4[s]];
        pu += dot4(vv[s], go[s]);
        pw += dot4(cc[s], go[s]);
      }
    }
    const float u = wave_sum(pu) * keep, w = wave_sum(pw) * keep;
    const float b = w + clv * u - clv * S - Tt * u;
    if (lane == 0) g_logit[(long)e * H + h] = a * (b - B);
    const float k = a * keep * (clv - Tt), ak = a * keep;
#pragma unroll
    for (int s = 0; s < MAX_SLOTS2; ++s)
      if (col4[s] >= 0) {
        reinterpret_cast<float4*>(g_value)[(long)e * D4 + col4[s]] =
            make_float4(k * go[s].x, k * go[s].y, k * go[s].z, k * go[s].w);
        gacc[s].x += ak * cc[s].x + k * vv[s].x, gacc[s].y += ak * cc[s].y + k * vv[s].y;
        gacc[s].z += ak * cc[s].z + k * vv[s].z, gacc[s].w += ak * cc[s].w + k * vv[s].w;
      }
  }
#pragma unroll
  for (int s = 0; s < MAX_SLOTS2; ++s)
    if (col4[s] >= 0) reinterpret_cast<float4*>(g_dout)[(long)n * D4 + col4[s]] = gacc[s];
}

HeadTab2 make_headtab2(const eqf_irreps& ir, int H, int* err) {
  HeadTab2 T{};
  *err = 0;
  T.nseg = ir.nseg;
  T.H = H;
  int off = 0, g = 0;
  for (int s = 0; s < ir.nseg; ++s) {
    T.off[s] = off, T.mul[s] = ir.mul[s], T.d[s] = 2 * ir.l[s] + 1;
    T.gcum[s] = g;
    if (ir.mul[s] % (4 * H) != 0) *err = EQF_E_UNSUPPORTED;
    g += T.d[s] * (ir.mul[s] / H) / 4;
    off += ir.mul[s] * T.d[s];
  }
  T.gcum[ir.nseg] = g;
  T.G = g;
  T.D = off;
  if (g > 64 * MAX_SLOTS2 || H > 16) *err = EQF_E_UNSUPPORTED;
  return T;
}

// ---------------------------------------------------------------------------------------------- exp-normal radial basis
// f_r(d) = cut(d) exp(-beta_r (exp(-alpha d) - mu_r)^2), cut = (cos(pi d / rc) + 1) / 2 for d < rc
// [ref: nets/graph_attention_transformer_md17.py:51-81,119-124].  One wave per edge.
__global__ __launch_bounds__(256) void rbf_expnorm_bwd2_kernel(const float* __restrict__ len, const float* __restrict__ g,
                                                               const float* __restrict__ cd, int E, int R,
                                                               const float* __restrict__ means,
                                                               const float* __restrict__ betas, float alpha, float rc,
                                                               float* __restrict__ g_len, float* __restrict__ g_g) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= E) return;
  const float d = len[e], cdv = cd[e];
  const float pi = 3.14159265358979323846f, k = pi / rc;
  const bool in = d < rc;
  const float cut = in ? 0.5f * (cosf(d * k) + 1.f) : 0.f;
  const float cut1 = in ? -0.5f * sinf(d * k) * k : 0.f;
  const float cut2 = in ? -0.5f * cosf(d * k) * k * k : 0.f;
  const float ex = expf(-alpha * d), ex1 = -alpha * ex, ex2 = alpha * alpha * ex;
  float acc = 0.f;
  for (int r = lane; r < R; r += 64) {
    const float b = betas[r], q = ex - means[r];
    const float G = expf(-b * q * q);
    const float lg1 = -2.f * b * q * ex1;                      // (log G)'
    const float G1 = G * lg1;
    const float G2 = G * (lg1 * lg1 - 2.f * b * (ex1 * ex1 + q * ex2));
    const float f1 = cut1 * G + cut * G1;
    const float f2 = cut2 * G + 2.f * cut1 * G1 + cut * G2;
    g_g[(long)e * R + r] = cdv * f1;
    acc += g[(long)e * R + r] * f2;
  }
  acc = wave_sum(acc);
  if (lane == 0) g_len[e] = cdv * acc;
}

// ---------------------------------------------------------------------------------------------- Gaussian radial basis
// out[e,r] = exp(-t^2 / 2) / (A sd_r),  t = (w len / rc + b - mu_r) / sd_r,  sd = |std| + 1e-5   [ref: nets/gaussian_rbf.py:6-40]
// first backward: d_len[e] = w ic sum_r g[e,r] h(t, sd), h = -t exp(-t^2/2) / (A sd^2).  cd = cotangent of d_len.
constexpr float kGaussA2 = 2.5066272160016134f;  // sqrt(2 * 3.14159), the reference's truncated pi

// per-edge part: g_len[e], g_g[e, :]; one wave per edge
__global__ __launch_bounds__(256) void rbf_gauss_bwd2_edge_kernel(const float* __restrict__ len, const float* __restrict__ g,
                                                                  const float* __restrict__ cd, int E, int R,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ stdp,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias, float ic,
                                                                  float* __restrict__ g_len, float* __restrict__ g_g) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= E) return;
  const float w = weight[0], x = w * (len[e] * ic) + bias[0], cdv = cd[e] * w * ic;
  float acc = 0.f;
  for (int r = lane; r < R; r += 64) {
    const float sd = fabsf(stdp[r]) + 1e-5f;
    const float t = (x - mean[r]) / sd;
    const float ex = __expf(-0.5f * t * t) / (kGaussA2 * sd * sd);
    const float h = -t * ex, ht = -(1.f - t * t) * ex;  // h and dh/dt
    g_g[(long)e * R + r] = cdv * h;
    acc += g[(long)e * R + r] * ht / sd;                 // dh/dx
  }
  acc = wave_sum(acc);
  if (lane == 0) g_len[e] = cdv * acc * w * ic;
}

// parameter part: thread r reduces CH edges for g_mean[r], g_std[r]; g_weight / g_bias reduced over the block
__global__ __launch_bounds__(256) void rbf_gauss_bwd2_param_kernel(const float* __restrict__ len, const float* __restrict__ g,
                                                                   const float* __restrict__ cd, int E, int R,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ stdp,
                                                                   const float* __restrict__ weight,
                                                                   const float* __restrict__ bias, float ic,
                                                                   float* __restrict__ g_mean, float* __restrict__ g_std,
                                                                   float* __restrict__ g_weight,
                                                                   float* __restrict__ g_bias, int CH) {
  const int r = threadIdx.x;
  const int e0 = blockIdx.x * CH, e1 = min(E, e0 + CH);
  float am = 0.f, as = 0.f, aw = 0.f, ab = 0.f;
  if (r < R) {
    const float w = weight[0], b = bias[0], mu = mean[r], sp = stdp[r];
    const float sd = fabsf(sp) + 1e-5f, sgn = (sp >= 0.f) ? 1.f : -1.f;
    for (int e = e0; e < e1; ++e) {
      const float xs = len[e] * ic;
      const float t = (w * xs + b - mu) / sd;
      const float ex = __expf(-0.5f * t * t) / (kGaussA2 * sd * sd);
      const float h = -t * ex, ht = -(1.f - t * t) * ex;
      const float cg = cd[e] * ic * g[(long)e * R + r];   // Phi = sum cg w h
      const float dphi_dx = cg * w * ht / sd;
      am -= dphi_dx;
      as += cg * w * (-(t / sd) * ht - 2.f * h / sd) * sgn;
      aw += dphi_dx * xs + cg * h;
      ab += dphi_dx;
    }
    atomicAdd(g_mean + r, am);
    atomicAdd(g_std + r, as);
  }
  __shared__ float rw[4], rb[4];
  aw = wave_sum(aw), ab = wave_sum(ab);
  if ((threadIdx.x & 63) == 0) rw[threadIdx.x >> 6] = aw, rb[threadIdx.x >> 6] = ab;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, cc = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) a += rw[i], cc += rb[i];
    atomicAdd(g_weight, a);
    atomicAdd(g_bias, cc);
  }
}

// ---------------------------------------------------------------------------------------------- spherical Bessel basis
// h_k(x) = nc env(x) sin(f_k x) / x, x = len / rc (graph.hip rbf_bessel_*; GemNet RadialBasis of ocpmodels).
// first backward: d_len[e] = (1/rc) sum_k g[e,k] h_k'(x).  One wave per edge; g_freq accumulated through LDS.
__global__ __launch_bounds__(256) void rbf_bessel_bwd2_kernel(const float* __restrict__ len, const float* __restrict__ g,
                                                              const float* __restrict__ cd, int E, int R,
                                                              const float* __restrict__ freq, float inv_rc, float nc,
                                                              float* __restrict__ g_len, float* __restrict__ g_g,
                                                              float* __restrict__ g_freq, int EPB) {
  extern __shared__ float red2[];  // [R]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = threadIdx.x; r < R; r += blockDim.x) red2[r] = 0.f;
  __syncthreads();
  const int e0 = blockIdx.x * EPB, e1 = min(E, e0 + EPB);
  for (int e = e0 + wave; e < e1; e += 4) {
    const float x = len[e] * inv_rc, cdv = cd[e] * inv_rc;
    float en0 = 0.f, en1 = 0.f, en2 = 0.f;
    if (x < 1.f) {
      const float x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
      en0 = 1.f + x5 * (-21.f + x * (35.f - 15.f * x));
      en1 = x4 * (-105.f + x * (210.f - 105.f * x));
      en2 = x3 * (-420.f + x * (1050.f - 630.f * x));
    }
    const float ix = 1.f / x;
    float acc = 0.f;
    for (int r = lane; r < R; r += 64) {
      const float f = freq[r], sn = sinf(f * x), cs = cosf(f * x);
      const float s0 = sn * ix, s1 = f * cs * ix - sn * ix * ix;
      const float s2 = -f * f * sn * ix - 2.f * f * cs * ix * ix + 2.f * sn * ix * ix * ix;
      const float h1 = nc * (en1 * s0 + en0 * s1);
      const float h2 = nc * (en2 * s0 + 2.f * en1 * s1 + en0 * s2);
      const float gv = g[(long)e * R + r];
      g_g[(long)e * R + r] = cdv * h1;
      acc += gv * h2;
      atomicAdd(&red2[r], cdv * gv * nc * (en1 * cs - en0 * f * sn));  // d h_k' / d f_k
    }
    acc = wave_sum(acc);
    if (lane == 0) g_len[e] = cdv * acc * inv_rc;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += blockDim.x) atomicAdd(g_freq + r, red2[r]);
}

// ---------------------------------------------------------------------------------------------- edge geometry
// first backward: d_vec = grad_vec ( <d_sh, sh(vec)> + d_len |vec| ) =: grad psi.  With c = cotangent of d_vec:
//   g_vec = Hessian(psi) c,  g_dsh = J_sh c,  g_dlen = <c, vec> / |vec|  -- all three are the eps parts of the first-order
//   code evaluated at vec + eps c (geom.h).
__global__ __launch_bounds__(256) void edge_geom_bwd2_kernel(const float* __restrict__ vec, const float* __restrict__ d_sh,
                                                             const float* __restrict__ d_len, const float* __restrict__ c,
                                                             int E, int lmax, float* __restrict__ g_vec,
                                                             float* __restrict__ g_dsh, float* __restrict__ g_dlen) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int S = (lmax + 1) * (lmax + 1);
  const Dual vx{vec[3 * e], c[3 * e]}, vy{vec[3 * e + 1], c[3 * e + 1]}, vz{vec[3 * e + 2], c[3 * e + 2]};
  Dual ox, oy, oz;
  geom_grad<Dual>(vx, vy, vz, lmax, d_sh ? d_sh + (long)e * S : nullptr, d_len != nullptr, d_len ? d_len[e] : 0.f, ox, oy,
                  oz);
  g_vec[3 * e] = ox.e, g_vec[3 * e + 1] = oy.e, g_vec[3 * e + 2] = oz.e;
  Dual o[16];
  const Dual L = geom_sh<Dual>(vx, vy, vz, lmax, o);
  if (g_dlen) g_dlen[e] = L.e;
  if (g_dsh) {
    float* out = g_dsh + (long)e * S;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < S) out[i] = o[i].e;
  }
}

}  // namespace

extern "C" {

int eqf_silu_bwd2(const float* x, const float* dy, const float* c, float* g_x, float* g_dy, long n, float c0,
                  void* stream) {
  if (!x || !dy || !c || !g_x || !g_dy) return EQF_E_BADARG;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(silu_bwd2_kernel, dim3(eqf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, c, g_x, g_dy, n,
                     c0);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_gate_bwd2(const float* in, const float* d_out, const float* c, float* g_in, float* g_dout, int rows, int S,
                  const eqf_irreps* gated, float c_silu, float c_sig, void* stream) {
  if (!in || !d_out || !c || !g_in || !g_dout || !gated) return EQF_E_BADARG;
  if (rows <= 0) return 0;
  const GateTab2 T = make_gatetab2(S, *gated);
  const int RB = 4;
  hipLaunchKernelGGL(gate_bwd2_kernel, dim3(eqf_cdiv(rows, RB), eqf_cdiv(T.Din, 256)), dim3(256), 0, (hipStream_t)stream,
                     in, d_out, c, g_in, g_dout, rows, T, c_silu, c_sig, RB);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_lnsilu_group_bwd2(const float* x, const float* gamma, const float* beta, const float* dy, const float* c,
                          float* g_x, float* g_gamma, float* g_beta, float* g_dy, int rows, int C, int groups, float eps,
                          void* stream) {
  if (!x || !gamma || !beta || !dy || !c || !g_x || !g_gamma || !g_beta || !g_dy || C < 1 || groups < 1 || groups > 65535)
    return EQF_E_BADARG;
  if (C > 64) return EQF_E_UNSUPPORTED;
  if (rows <= 0) return 0;
  int blocks = eqf_cdiv(rows, WPB * 2);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lnsilu_bwd2_kernel, dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, dy, c, g_x,
                     g_gamma, g_beta, g_dy, rows, C, eps);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_lnsilu_bwd2(const float* x, const float* gamma, const float* beta, const float* dy, const float* c, float* g_x,
                    float* g_gamma, float* g_beta, float* g_dy, int rows, int C, float eps, void* stream) {
  return eqf_lnsilu_group_bwd2(x, gamma, beta, dy, c, g_x, g_gamma, g_beta, g_dy, rows, C, 1, eps, stream);
}

int eqf_layernorm_bwd2(const float* x, const float* weight, const float* dy, const float* c, float* g_x, float* g_weight,
                       float* g_dy, int rows, const eqf_irreps* irreps, float eps, void* stream) {
  if (!x || !weight || !dy || !c || !g_x || !g_weight || !g_dy || !irreps || irreps->nseg < 1 ||
      irreps->nseg > EQF_MAX_SEG)
    return EQF_E_BADARG;
  if (rows <= 0) return 0;
  const SegTab2 T = make_segtab2(*irreps);
  hipLaunchKernelGGL(layernorm_bwd2_kernel, dim3(eqf_cdiv(rows, WPB)), dim3(256), 0, (hipStream_t)stream, x, weight, dy,
                     c, g_x, g_weight, g_dy, rows, T, eps);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_alpha_bwd2(const float* a, const float* alpha_dot, const float* d_logit, const float* ca, float* g_a,
                   float* g_alpha_dot, float* g_dlogit, int E, int H, int Kh, float c, void* stream) {
  if (!a || !alpha_dot || !d_logit || !ca || !g_a || !g_alpha_dot || !g_dlogit) return EQF_E_BADARG;
  if (!(Kh == 8 || Kh == 16 || Kh == 32 || Kh == 64)) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  const int CH = 8;
  hipLaunchKernelGGL(alpha_bwd2_kernel, dim3(eqf_cdiv(H * Kh, 256), eqf_cdiv(E, CH)), dim3(256), 0, (hipStream_t)stream,
                     a, alpha_dot, d_logit, ca, g_a, g_alpha_dot, g_dlogit, E, H * Kh, Kh, c, CH);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_attn_aggregate_bwd2(const float* alpha, const float* value, const int* row_ptr, const float* d_out,
                            const float* c_value, const float* c_logit, float* g_logit, float* g_value, float* g_dout,
                            int N, int H, const eqf_irreps* irreps, float drop_p, unsigned long long seed, void* stream) {
  if (!alpha || !value || !row_ptr || !d_out || !g_logit || !g_value || !g_dout || !irreps || H < 1) return EQF_E_BADARG;
  int err;
  const HeadTab2 T = make_headtab2(*irreps, H, &err);
  if (err) return err;
  if (N <= 0) return 0;
  hipLaunchKernelGGL(attn_bwd2_kernel, dim3(N), dim3(64 * H), 0, (hipStream_t)stream, alpha, value, row_ptr, d_out,
                     c_value, c_logit, g_logit, g_value, g_dout, T, drop_p, seed);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_rbf_expnorm_bwd2(const float* len, const float* d_out, const float* c_len, int E, int R, const float* means,
                         const float* betas, float alpha, float cutoff, float* g_len, float* g_dout, void* stream) {
  if (!len || !d_out || !c_len || !means || !betas || !g_len || !g_dout) return EQF_E_BADARG;
  if (E <= 0) return 0;
  hipLaunchKernelGGL(rbf_expnorm_bwd2_kernel, dim3(eqf_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, len, d_out, c_len,
                     E, R, means, betas, alpha, cutoff, g_len, g_dout);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_rbf_gaussian_bwd2(const float* len, const float* d_out, const float* c_len, int E, int R, const float* mean,
                          const float* std, const float* weight, const float* bias, float cutoff, float* g_len,
                          float* g_dout, float* g_mean, float* g_std, float* g_weight, float* g_bias, void* stream) {
  if (!len || !d_out || !c_len || !mean || !std || !weight || !bias || !g_len || !g_dout || !g_mean || !g_std ||
      !g_weight || !g_bias)
    return EQF_E_BADARG;
  if (R > 256) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  const float ic = 1.f / cutoff;
  hipLaunchKernelGGL(rbf_gauss_bwd2_edge_kernel, dim3(eqf_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, len, d_out,
                     c_len, E, R, mean, std, weight, bias, ic, g_len, g_dout);
  EQF_CHECK_LAUNCH();
  const int CH = 64;
  const int threads = ((R + 63) / 64) * 64;
  hipLaunchKernelGGL(rbf_gauss_bwd2_param_kernel, dim3(eqf_cdiv(E, CH)), dim3(threads), 0, (hipStream_t)stream, len,
                     d_out, c_len, E, R, mean, std, weight, bias, ic, g_mean, g_std, g_weight, g_bias, CH);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_edge_geom_bwd2(const float* vec, const float* d_sh, const float* d_len, const float* c_vec, int E, int lmax,
                       float* g_vec, float* g_dsh, float* g_dlen, void* stream) {
  if (!vec || !c_vec || !g_vec || lmax < 0 || lmax > 3) return EQF_E_BADARG;
  if (E <= 0) return 0;
  hipLaunchKernelGGL(edge_geom_bwd2_kernel, dim3(eqf_cdiv(E, 256)), dim3(256), 0, (hipStream_t)stream, vec, d_sh, d_len,
                     c_vec, E, lmax, g_vec, g_dsh, g_dlen);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_rbf_bessel_bwd2(const float* len, const float* d_out, const float* c_len, int E, int R, const float* freq,
                        float cutoff, float* g_len, float* g_dout, float* g_freq, void* stream) {
  if (!len || !d_out || !c_len || !freq || !g_len || !g_dout || !g_freq || cutoff <= 0.f) return EQF_E_BADARG;
  if (E <= 0) return 0;
  const int EPB = 64;
  hipLaunchKernelGGL(rbf_bessel_bwd2_kernel, dim3(eqf_cdiv(E, EPB)), dim3(256), sizeof(float) * R, (hipStream_t)stream, len,
                     d_out, c_len, E, R, freq, 1.f / cutoff, sqrtf(2.f / (cutoff * cutoff * cutoff)), g_len, g_dout, g_freq,
                     EPB);
  EQF_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
