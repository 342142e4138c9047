// Edge-wise gather / scatter and attention kernels over the dst-sorted radius graph (HBM-bound).
//
// Edges are stored sorted by destination node (CSR row_ptr), so every "scatter" of the reference
// (torch_scatter atomics) becomes a segmented reduction: one workgroup / wavefront owns one destination
// row, reads its incoming edge rows coalesced (1920-byte rows for 480 channels) and writes the node row
// once -- no atomics, deterministic.  The softmax over incoming edges is reduced with wavefront shuffles.
#include "common.h"
#include "prof.h"

namespace {

// ---------------------------------------------------------------------------------------------- gather
__global__ __launch_bounds__(256) void gather_add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const int* __restrict__ src, const int* __restrict__ dst,
                                                         float* __restrict__ msg, long total4, int D4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const long e = idx / D4;
  const int c = (int)(idx - e * D4);
  float4 v = reinterpret_cast<const float4*>(a)[(long)src[e] * D4 + c];
  if (b) {
    const float4 w = reinterpret_cast<const float4*>(b)[(long)dst[e] * D4 + c];
    v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
  }
  reinterpret_cast<float4*>(msg)[idx] = v;
}

__global__ __launch_bounds__(128) void segment_sum_kernel(const float* __restrict__ x, const int* __restrict__ ptr,
                                                          const int* __restrict__ perm, float* __restrict__ out, int D,
                                                          float scale, int accumulate) {
  const int n = blockIdx.x;
  const int beg = ptr[n], end = ptr[n + 1];
  if ((D & 3) == 0) {
    const int D4 = D >> 2;
    for (int c = threadIdx.x; c < D4; c += blockDim.x) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      // four rows per step, their index and value loads in flight together (a row is ~11 entries: one load per iteration of a
      // run-time loop made the kernel a chain of 2 x 11 dependent memory round trips); same summation order as before
      for (int q = beg; q < end; q += 4) {
        long row[4];
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int qq = min(q + j, end - 1);
          row[j] = perm ? perm[qq] : qq;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const float4*>(x)[row[j] * D4 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (q + j < end) acc.x += v[j].x, acc.y += v[j].y, acc.z += v[j].z, acc.w += v[j].w;
      }
      acc.x *= scale, acc.y *= scale, acc.z *= scale, acc.w *= scale;
      float4* o = reinterpret_cast<float4*>(out) + (long)n * D4 + c;
      if (accumulate) {
        const float4 p = *o;
        acc.x += p.x, acc.y += p.y, acc.z += p.z, acc.w += p.w;
      }
      *o = acc;
    }
  } else {
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
      float acc = 0.f;
      for (int q = beg; q < end; ++q) {
        const long row = perm ? perm[q] : q;
        acc += x[row * D + c];
      }
      acc *= scale;
      float* o = out + (long)n * D + c;
      *o = accumulate ? *o + acc : acc;
    }
  }
}

__global__ __launch_bounds__(256) void segment_bcast_kernel(const float* __restrict__ x, const int* __restrict__ seg_of,
                                                            float* __restrict__ out, long total, int D, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long row = idx / D;
  const int c = (int)(idx - row * D);
  out[idx] = scale * x[(long)seg_of[row] * D + c];
}

// out[q,c] = s[seg_of[q]] * x[q,c]: per-graph stochastic depth (GraphDropPath).  Linear in x, so the same kernel is its
// own backward (on dy) and second-order term.
__global__ __launch_bounds__(256) void segment_scale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                            const int* __restrict__ seg_of, float* __restrict__ out,
                                                            long total4, int D4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const long row = idx / D4;
  const float f = s[seg_of[row]];
  float4 v = reinterpret_cast<const float4*>(x)[idx];
  v.x *= f; v.y *= f; v.z *= f; v.w *= f;
  reinterpret_cast<float4*>(out)[idx] = v;
}

// ---------------------------------------------------------------------------------------------- DTP coupling
__global__ __launch_bounds__(256) void coupling_fwd_kernel(const float* __restrict__ sh, const float* __restrict__ cg,
                                                           const eqf_dtp_paths P, float* __restrict__ M, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long e = idx / P.npaths;
  const int p = (int)(idx - e * P.npaths);
  const int d1 = 2 * P.l1[p] + 1, d2 = 2 * P.l2[p] + 1, d3 = 2 * P.l3[p] + 1;
  const float* y = sh + e * P.sh_dim + P.l2[p] * P.l2[p];
  const float* c = cg + P.cg_off[p];
  float* m = M + e * P.m_numel + P.m_off[p];
  for (int i = 0; i < d1; ++i)
    for (int k = 0; k < d3; ++k) {
      float acc = 0.f;
      for (int j = 0; j < d2; ++j) acc = fmaf(c[(i * d2 + j) * d3 + k], y[j], acc);
      m[i * d3 + k] = acc;
    }
}

__global__ __launch_bounds__(256) void coupling_bwd_kernel(const float* __restrict__ dM, const float* __restrict__ cg,
                                                           const eqf_dtp_paths P, float* __restrict__ d_sh, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long e = idx / P.sh_dim;
  const int s = (int)(idx - e * P.sh_dim);
  int l2 = 0;
  while ((l2 + 1) * (l2 + 1) <= s) ++l2;
  const int j = s - l2 * l2;
  float acc = 0.f;
  for (int p = 0; p < P.npaths; ++p) {
    if (P.l2[p] != l2) continue;
    const int d1 = 2 * P.l1[p] + 1, d2 = 2 * l2 + 1, d3 = 2 * P.l3[p] + 1;
    const float* c = cg + P.cg_off[p];
    const float* g = dM + e * P.m_numel + P.m_off[p];
    for (int i = 0; i < d1; ++i)
      for (int k = 0; k < d3; ++k) acc = fmaf(c[(i * d2 + j) * d3 + k], g[i * d3 + k], acc);
  }
  d_sh[idx] = acc;
}

// ---------------------------------------------------------------------------------------------- DTP (un-fused)
__global__ __launch_bounds__(256) void dtp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ M,
                                                      const float* __restrict__ w, const eqf_dtp_paths P,
                                                      float* __restrict__ out) {
  const long e = blockIdx.x;
  const int item = blockIdx.y * blockDim.x + threadIdx.x;  // (path, channel) == index into the weight row
  if (item >= P.w_numel) return;
  int p = 0;
  while (p + 1 < P.npaths && item >= P.w_off[p + 1]) ++p;
  const int u = item - P.w_off[p];
  const int d1 = 2 * P.l1[p] + 1, d3 = 2 * P.l3[p] + 1;
  const float wv = w ? w[e * P.w_numel + item] : 1.f;
  const float* xp = x + e * P.in_dim + P.in_off[p] + u;
  const float* mp = M + e * P.m_numel + P.m_off[p];
  float* op = out + e * P.out_dim + P.out_off[p] + P.out_ch[p] + u;
  float xv[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) xv[i] = (i < d1) ? xp[i * P.mul[p]] : 0.f;
  for (int k = 0; k < d3; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (i < d1) acc = fmaf(mp[i * d3 + k], xv[i], acc);
    op[k * P.out_k[p]] = acc * wv;
  }
}

struct InSegs {
  int nseg;
  int off[EQF_MAX_SEG], mul[EQF_MAX_SEG], cum[EQF_MAX_SEG + 1];
};

// one thread per (edge, input channel): owns dx[e, seg, :, u]; loops over the paths fed by that segment
__global__ __launch_bounds__(256) void dtp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ M,
                                                      const float* __restrict__ w, const eqf_dtp_paths P,
                                                      const InSegs S, const float* __restrict__ d_out,
                                                      float* __restrict__ dx, float* __restrict__ dw,
                                                      float* __restrict__ dM) {
  extern __shared__ float dM_s[];  // m_numel floats when dM != null
  const long e = blockIdx.x;
  if (dM) {
    for (int i = threadIdx.x; i < P.m_numel; i += blockDim.x) dM_s[i] = 0.f;
    __syncthreads();
  }
  for (int ch = threadIdx.x; ch < S.cum[S.nseg]; ch += blockDim.x) {
    int s = 0;
    while (s + 1 < S.nseg && ch >= S.cum[s + 1]) ++s;
    const int u = ch - S.cum[s];
    const int in_off = S.off[s], mul = S.mul[s];
    float xv[7], gx[7];
    int d1 = 1;
    for (int p = 0; p < P.npaths; ++p)
      if (P.in_off[p] == in_off) d1 = 2 * P.l1[p] + 1;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      xv[i] = (i < d1) ? x[e * P.in_dim + in_off + i * mul + u] : 0.f;
      gx[i] = 0.f;
    }
    for (int p = 0; p < P.npaths; ++p) {
      if (P.in_off[p] != in_off) continue;
      const int d3 = 2 * P.l3[p] + 1;
      const float wv = w ? w[e * P.w_numel + P.w_off[p] + u] : 1.f;
      const float* mp = M + e * P.m_numel + P.m_off[p];
      const float* gp = d_out + e * P.out_dim + P.out_off[p] + P.out_ch[p] + u;
      float gw = 0.f;
      for (int k = 0; k < d3; ++k) {
        const float g = gp[k * P.out_k[p]];
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (i < d1) {
            const float m = mp[i * d3 + k];
            t = fmaf(m, xv[i], t);
            gx[i] = fmaf(m * wv, g, gx[i]);
            if (dM) atomicAdd(&dM_s[P.m_off[p] + i * d3 + k], wv * xv[i] * g);
          }
        gw = fmaf(g, t, gw);
      }
      if (dw) dw[e * P.w_numel + P.w_off[p] + u] = gw;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (i < d1) dx[e * P.in_dim + in_off + i * mul + u] = gx[i];
  }
  if (dM) {
    __syncthreads();
    for (int i = threadIdx.x; i < P.m_numel; i += blockDim.x) dM[e * P.m_numel + i] = dM_s[i];
  }
}

// ---------------------------------------------------------------------------------------------- attention logits
__device__ __forceinline__ float slrelu(float x) { return 0.6f * x + 0.4f * x * (2.f * sigmoidf_(x) - 1.f); }
__device__ __forceinline__ float dslrelu(float x) {
  const float s = sigmoidf_(x);
  return 0.6f + 0.4f * (2.f * s - 1.f) + 0.8f * x * s * (1.f - s);
}

// Kh in {8,16,32,64}: Kh consecutive lanes share (edge, head)
__global__ __launch_bounds__(256) void alpha_fwd_kernel(const float* __restrict__ a, const float* __restrict__ adot,
                                                        float* __restrict__ logit, long total, int HK, int Kh, float c) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  long e = 0;
  int col = 0;
  if (idx < total) {
    e = idx / HK;
    col = (int)(idx - e * HK);
    v = c * slrelu(a[idx]) * adot[col];
  }
  for (int o = Kh >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (idx < total && (col % Kh) == 0) logit[e * (HK / Kh) + col / Kh] = v;
}

__global__ __launch_bounds__(256) void alpha_bwd_kernel(const float* __restrict__ a, const float* __restrict__ adot,
                                                        const float* __restrict__ d_logit, float* __restrict__ da,
                                                        float* __restrict__ d_adot, int E, int HK, int Kh, float c,
                                                        int CH) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= HK) return;
  const int h = col / Kh, H = HK / Kh;
  const int e0 = blockIdx.y * CH, e1 = min(E, e0 + CH);
  const float ad = adot[col];
  float acc = 0.f;
  for (int e = e0; e < e1; ++e) {
    const float g = d_logit[(long)e * H + h] * c;
    const float av = a[(long)e * HK + col];
    da[(long)e * HK + col] = g * ad * dslrelu(av);
    acc += g * slrelu(av);
  }
  atomicAdd(d_adot + col, acc);
}

// float4 columns, RP edge rows per pass, four passes in flight: the column-per-thread kernel above walks 16 edges in a
// serial loop with 3 waves per SIMD on the chip -- 47 us for 26 MB.  HK % 4 == 0, HK / 4 <= 256, Kh % 4 == 0.
__global__ __launch_bounds__(256) void alpha_bwd4_kernel(const float* __restrict__ a, const float* __restrict__ adot,
                                                         const float* __restrict__ d_logit, float* __restrict__ da,
                                                         float* __restrict__ d_adot, int E, int HK, int Kh, float c,
                                                         int CH) {
  const int Q = HK >> 2;               // float4 columns per row
  const int RP = 256 / Q;              // rows per pass
  const int tx = threadIdx.x % Q, ty = threadIdx.x / Q;
  const int H = HK / Kh, h = (4 * tx) / Kh;
  const int e0 = blockIdx.x * CH, e1 = min(E, e0 + CH);
  __shared__ float red[4 * 256];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ty < RP) {
    const float4 ad = reinterpret_cast<const float4*>(adot)[tx];
    for (int eb = e0 + ty; eb < e1; eb += 4 * RP) {
      float4 av[4];
      float g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = eb + u * RP;
        const bool ok = e < e1;
        av[u] = ok ? reinterpret_cast<const float4*>(a)[(long)e * Q + tx] : make_float4(0.f, 0.f, 0.f, 0.f);
        g[u] = ok ? d_logit[(long)e * H + h] * c : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = eb + u * RP;
        if (e < e1) {
          float4 o;
          o.x = g[u] * ad.x * dslrelu(av[u].x), o.y = g[u] * ad.y * dslrelu(av[u].y);
          o.z = g[u] * ad.z * dslrelu(av[u].z), o.w = g[u] * ad.w * dslrelu(av[u].w);
          reinterpret_cast<float4*>(da)[(long)e * Q + tx] = o;
          acc.x += g[u] * slrelu(av[u].x), acc.y += g[u] * slrelu(av[u].y);
          acc.z += g[u] * slrelu(av[u].z), acc.w += g[u] * slrelu(av[u].w);
        }
      }
    }
  }
  float* my = red + 4 * threadIdx.x;
  my[0] = acc.x, my[1] = acc.y, my[2] = acc.z, my[3] = acc.w;
  __syncthreads();
  if ((int)threadIdx.x < HK) {
    const int col = threadIdx.x, q = col >> 2, k = col & 3;
    float sum = 0.f;
    for (int r = 0; r < RP; ++r) sum += red[4 * (r * Q + q) + k];
    atomicAdd(d_adot + col, sum);
  }
}

// ---------------------------------------------------------------------------------------------- softmax + aggregate
struct HeadTab {
  int nseg, H, D, G;  // G = float4 groups per head
  int off[EQF_MAX_SEG], mul[EQF_MAX_SEG], d[EQF_MAX_SEG], gcum[EQF_MAX_SEG + 1];
};

__device__ __forceinline__ int head_col(const HeadTab& T, int h, int g) {
  int s = 0;
  while (s + 1 < T.nseg && g >= T.gcum[s + 1]) ++s;
  const int mh = T.mul[s] / T.H;  // channels of this head in segment s
  const int q4 = mh >> 2;
  const int j = g - T.gcum[s];
  const int m = j / q4, q = j - m * q4;
  return T.off[s] + m * T.mul[s] + h * mh + 4 * q;
}

__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned long long idx, float p) {
  if (p <= 0.f) return 1.f;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  return (u >= p) ? 1.f / (1.f - p) : 0.f;
}

constexpr int MAX_SLOTS = 4;  // up to 256 float4 groups (1024 channels) per head

__global__ void attn_fwd_kernel(const float* __restrict__ logit, const float* __restrict__ value,
                                const int* __restrict__ row_ptr, float* __restrict__ alpha, float* __restrict__ out,
                                const HeadTab T, float drop_p, unsigned long long seed0,
    const unsigned long long* __restrict__ seed_off) {
  // (the mask seed of a launch captured in a HIP graph: host part + a device word the caller advances between replays)
  const unsigned long long seed = seed0 + (seed_off ? *seed_off : 0ull);
  const int n = blockIdx.x;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int beg = row_ptr[n], end = row_ptr[n + 1];
  const int H = T.H, D4 = T.D >> 2;
  int col4[MAX_SLOTS];
  float4 acc[MAX_SLOTS];
#pragma unroll
  for (int s = 0; s < MAX_SLOTS; ++s) {
    const int g = lane + 64 * s;
    col4[s] = (g < T.G) ? head_col(T, h, g) >> 2 : -1;
    acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float mx = -INFINITY;
  for (int e = beg + lane; e < end; e += 64) mx = fmaxf(mx, logit[(long)e * H + h]);
  mx = wave_max(mx);
  float sm = 0.f;
  for (int e = beg + lane; e < end; e += 64) sm += __expf(logit[(long)e * H + h] - mx);
  sm = wave_sum(sm);
  const float inv = 1.f / (sm + 1e-16f);
  for (int e = beg; e < end; ++e) {
    const float a = __expf(logit[(long)e * H + h] - mx) * inv;
    if (lane == 0) alpha[(long)e * H + h] = a;
    const float ak = a * keep_scale(seed, (unsigned long long)e * H + h, drop_p);
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s)
      if (col4[s] >= 0) {
        const float4 v = reinterpret_cast<const float4*>(value)[(long)e * D4 + col4[s]];
        acc[s].x = fmaf(ak, v.x, acc[s].x), acc[s].y = fmaf(ak, v.y, acc[s].y);
        acc[s].z = fmaf(ak, v.z, acc[s].z), acc[s].w = fmaf(ak, v.w, acc[s].w);
      }
  }
#pragma unroll
  for (int s = 0; s < MAX_SLOTS; ++s)
    if (col4[s] >= 0) reinterpret_cast<float4*>(out)[(long)n * D4 + col4[s]] = acc[s];
}

// Heads of <= 32 float4 groups (QM9 / MD17: 30): a wave covers TWO edges per step (lanes 0-31 edge e, lanes 32-63 edge
// e+1) and two such steps are issued before the first FMA -- four independent 16-byte value loads per lane in flight
// instead of one with 30 of 64 lanes working; the halves are folded with one shuffle at the end.
__global__ void attn_fwd_half_kernel(const float* __restrict__ logit, const float* __restrict__ value,
                                     const int* __restrict__ row_ptr, float* __restrict__ alpha,
                                     float* __restrict__ out, const HeadTab T, float drop_p, unsigned long long seed0,
    const unsigned long long* __restrict__ seed_off) {
  // (the mask seed of a launch captured in a HIP graph: host part + a device word the caller advances between replays)
  const unsigned long long seed = seed0 + (seed_off ? *seed_off : 0ull);
  const int n = blockIdx.x;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, gl = lane & 31;
  const int beg = row_ptr[n], end = row_ptr[n + 1];
  const int H = T.H, D4 = T.D >> 2;
  const int deg = end - beg;
  const int col4 = (gl < T.G) ? head_col(T, h, gl) >> 2 : -1;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* const V = reinterpret_cast<const float4*>(value);
  // A row is ~11 edges: the kernel is a chain of dependent memory round trips, not a stream.  Two of the three are
  // taken out of the chain: (1) the value rows of the first four edges depend on row_ptr only and are requested BEFORE
  // the softmax statistics; inside the loop the next four are requested before the current ones are used; (2) the
  // head's logits are read ONCE, lane j holding edge beg + j, and handed to the edge loop by lane shuffles (rows longer
  // than 64 edges re-read memory for the rest).
  float4 xa = (beg + half < end && col4 >= 0) ? V[(long)(beg + half) * D4 + col4] : z4;
  float4 xb = (beg + 2 + half < end && col4 >= 0) ? V[(long)(beg + 2 + half) * D4 + col4] : z4;
  const float lg = lane < deg ? logit[(long)(beg + lane) * H + h] : -INFINITY;
  float mx = lg;
  for (int e = beg + 64 + lane; e < end; e += 64) mx = fmaxf(mx, logit[(long)e * H + h]);
  mx = wave_max(mx);
  float sm = lane < deg ? __expf(lg - mx) : 0.f;
  for (int e = beg + 64 + lane; e < end; e += 64) sm += __expf(logit[(long)e * H + h] - mx);
  sm = wave_sum(sm);
  const float inv = 1.f / (sm + 1e-16f);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e0 = beg; e0 < end; e0 += 4) {
    const int ea = e0 + half, eb = e0 + 2 + half;
    const bool va = ea < end, vb = eb < end;
    const float4 xan = (ea + 4 < end && col4 >= 0) ? V[(long)(ea + 4) * D4 + col4] : z4;
    const float4 xbn = (eb + 4 < end && col4 >= 0) ? V[(long)(eb + 4) * D4 + col4] : z4;
    const int ia = ea - beg, ib = eb - beg;
    float la = __shfl(lg, ia & 63), lb = __shfl(lg, ib & 63);
    if (ia >= 64 && va) la = logit[(long)ea * H + h];
    if (ib >= 64 && vb) lb = logit[(long)eb * H + h];
    const float aa = va ? __expf(la - mx) * inv : 0.f, ab = vb ? __expf(lb - mx) * inv : 0.f;
    if (gl == 0) {
      if (va) alpha[(long)ea * H + h] = aa;
      if (vb) alpha[(long)eb * H + h] = ab;
    }
    const float ka = aa * keep_scale(seed, (unsigned long long)ea * H + h, drop_p);
    const float kb = ab * keep_scale(seed, (unsigned long long)eb * H + h, drop_p);
    acc.x = fmaf(ka, xa.x, acc.x), acc.y = fmaf(ka, xa.y, acc.y), acc.z = fmaf(ka, xa.z, acc.z), acc.w = fmaf(ka, xa.w, acc.w);
    acc.x = fmaf(kb, xb.x, acc.x), acc.y = fmaf(kb, xb.y, acc.y), acc.z = fmaf(kb, xb.z, acc.z), acc.w = fmaf(kb, xb.w, acc.w);
    xa = xan, xb = xbn;
  }
  acc.x += __shfl_xor(acc.x, 32), acc.y += __shfl_xor(acc.y, 32);
  acc.z += __shfl_xor(acc.z, 32), acc.w += __shfl_xor(acc.w, 32);
  if (half == 0 && col4 >= 0) reinterpret_cast<float4*>(out)[(long)n * D4 + col4] = acc;
}

__global__ void attn_bwd_half_kernel(const float* __restrict__ alpha, const float* __restrict__ value,
                                     const int* __restrict__ row_ptr, const float* __restrict__ d_out,
                                     float* __restrict__ d_value, float* __restrict__ d_logit, const HeadTab T,
                                     float drop_p, unsigned long long seed0,
    const unsigned long long* __restrict__ seed_off) {
  // (the mask seed of a launch captured in a HIP graph: host part + a device word the caller advances between replays)
  const unsigned long long seed = seed0 + (seed_off ? *seed_off : 0ull);
  const int n = blockIdx.x;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, gl = lane & 31;
  const int beg = row_ptr[n], end = row_ptr[n + 1];
  const int H = T.H, D4 = T.D >> 2;
  const int col4 = (gl < T.G) ? head_col(T, h, gl) >> 2 : -1;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 go = (col4 >= 0) ? reinterpret_cast<const float4*>(d_out)[(long)n * D4 + col4] : z4;
  float s_acc = 0.f;
  for (int e0 = beg; e0 < end; e0 += 4) {
    const int ea = e0 + half, eb = e0 + 2 + half;
    const bool va = ea < end, vb = eb < end;
    const float aa = va ? alpha[(long)ea * H + h] : 0.f, ab = vb ? alpha[(long)eb * H + h] : 0.f;
    const float4 xa = (va && col4 >= 0) ? reinterpret_cast<const float4*>(value)[(long)ea * D4 + col4] : z4;
    const float4 xb = (vb && col4 >= 0) ? reinterpret_cast<const float4*>(value)[(long)eb * D4 + col4] : z4;
    const float keepa = keep_scale(seed, (unsigned long long)ea * H + h, drop_p);
    const float keepb = keep_scale(seed, (unsigned long long)eb * H + h, drop_p);
    float pa = xa.x * go.x + xa.y * go.y + xa.z * go.z + xa.w * go.w;
    float pb = xb.x * go.x + xb.y * go.y + xb.z * go.z + xb.w * go.w;
    if (va && col4 >= 0) {
      const float k = aa * keepa;
      reinterpret_cast<float4*>(d_value)[(long)ea * D4 + col4] = make_float4(k * go.x, k * go.y, k * go.z, k * go.w);
    }
    if (vb && col4 >= 0) {
      const float k = ab * keepb;
      reinterpret_cast<float4*>(d_value)[(long)eb * D4 + col4] = make_float4(k * go.x, k * go.y, k * go.z, k * go.w);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) pa += __shfl_xor(pa, o), pb += __shfl_xor(pb, o);  // sums inside each half
    const float da = pa * keepa, db = pb * keepb;
    float contrib = aa * da + ab * db;           // this half's two edges
    contrib += __shfl_xor(contrib, 32);          // + the other half's
    s_acc += contrib;
    if (gl == 0) {  // stash d(alpha); fixed up below
      if (va) d_logit[(long)ea * H + h] = da;
      if (vb) d_logit[(long)eb * H + h] = db;
    }
  }
  // the stash above was written by lanes 0 / 32 of THIS wave and is re-read below by its other lanes: same wave, same CU,
  // so both accesses go through this CU's write-through L1; the workgroup fence (s_waitcnt vmcnt(0)) retires the stores
  // before the loads are issued.  No other wave touches these addresses.
  __threadfence_block();
  for (int e = beg + lane; e < end; e += 64) {
    const float da = d_logit[(long)e * H + h];
    d_logit[(long)e * H + h] = alpha[(long)e * H + h] * (da - s_acc);
  }
}

__global__ void attn_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ value,
                                const int* __restrict__ row_ptr, const float* __restrict__ d_out,
                                float* __restrict__ d_value, float* __restrict__ d_logit, const HeadTab T, float drop_p,
                                unsigned long long seed0,
    const unsigned long long* __restrict__ seed_off) {
  // (the mask seed of a launch captured in a HIP graph: host part + a device word the caller advances between replays)
  const unsigned long long seed = seed0 + (seed_off ? *seed_off : 0ull);
  const int n = blockIdx.x;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int beg = row_ptr[n], end = row_ptr[n + 1];
  const int H = T.H, D4 = T.D >> 2;
  int col4[MAX_SLOTS];
  float4 go[MAX_SLOTS];
#pragma unroll
  for (int s = 0; s < MAX_SLOTS; ++s) {
    const int g = lane + 64 * s;
    col4[s] = (g < T.G) ? head_col(T, h, g) >> 2 : -1;
    go[s] = (col4[s] >= 0) ? reinterpret_cast<const float4*>(d_out)[(long)n * D4 + col4[s]]
                           : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float s_acc = 0.f;
  for (int e = beg; e < end; ++e) {
    const float a = alpha[(long)e * H + h];
    const float keep = keep_scale(seed, (unsigned long long)e * H + h, drop_p);
    const float ak = a * keep;
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s)
      if (col4[s] >= 0) {
        const float4 v = reinterpret_cast<const float4*>(value)[(long)e * D4 + col4[s]];
        part += v.x * go[s].x + v.y * go[s].y + v.z * go[s].z + v.w * go[s].w;
        float4 dv;
        dv.x = ak * go[s].x, dv.y = ak * go[s].y, dv.z = ak * go[s].z, dv.w = ak * go[s].w;
        reinterpret_cast<float4*>(d_value)[(long)e * D4 + col4[s]] = dv;
      }
    const float da = wave_sum(part) * keep;
    s_acc = fmaf(a, da, s_acc);
    if (lane == 0) d_logit[(long)e * H + h] = da;  // stash d(alpha); fixed up below
  }
  __threadfence_block();
  for (int e = beg + lane; e < end; e += 64) {
    const float da = d_logit[(long)e * H + h];
    d_logit[(long)e * H + h] = alpha[(long)e * H + h] * (da - s_acc);
  }
}

HeadTab make_headtab(const eqf_irreps& ir, int H, int* err) {
  HeadTab T{};
  *err = 0;
  T.nseg = ir.nseg;
  T.H = H;
  int off = 0, g = 0;
  for (int s = 0; s < ir.nseg; ++s) {
    T.off[s] = off;
    T.mul[s] = ir.mul[s];
    T.d[s] = 2 * ir.l[s] + 1;
    T.gcum[s] = g;
    if (ir.mul[s] % (4 * H) != 0) *err = EQF_E_UNSUPPORTED;
    g += T.d[s] * (ir.mul[s] / H) / 4;
    off += ir.mul[s] * T.d[s];
  }
  T.gcum[ir.nseg] = g;
  T.G = g;
  T.D = off;
  if (g > 64 * MAX_SLOTS || H > 16) *err = EQF_E_UNSUPPORTED;
  return T;
}

}  // namespace

extern "C" {

int eqf_gather_add_fwd(const float* a, const float* b, const int* src, const int* dst, float* msg, int E, int D,
                       void* stream) {
  if (!a || !src || !msg || (b && !dst)) return EQF_E_BADARG;
  if (D % 4 != 0) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  const long total4 = (long)E * (D / 4);
  hipLaunchKernelGGL(gather_add_kernel, dim3(eqf_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, a, b, src, dst,
                     msg, total4, D / 4);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_segment_sum(const float* x, const int* ptr, const int* perm, float* out, int nseg, int D, float scale,
                    int accumulate, void* stream) {
  if (!x || !ptr || !out) return EQF_E_BADARG;
  if (nseg <= 0 || D <= 0) return 0;
  hipLaunchKernelGGL(segment_sum_kernel, dim3(nseg), dim3(128), 0, (hipStream_t)stream, x, ptr, perm, out, D, scale,
                     accumulate);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_segment_scale(const float* x, const float* s, const int* seg_of, float* out, int rows, int D, void* stream) {
  if (!x || !s || !seg_of || !out) return EQF_E_BADARG;
  if (D % 4) return EQF_E_UNSUPPORTED;
  if (rows <= 0 || D <= 0) return 0;
  const long total4 = (long)rows * (D / 4);
  hipLaunchKernelGGL(segment_scale_kernel, dim3(eqf_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, s, seg_of,
                     out, total4, D / 4);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_segment_bcast(const float* x, const int* seg_of, float* out, int rows, int D, float scale, void* stream) {
  if (!x || !seg_of || !out) return EQF_E_BADARG;
  if (rows <= 0 || D <= 0) return 0;
  const long total = (long)rows * D;
  hipLaunchKernelGGL(segment_bcast_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, seg_of,
                     out, total, D, scale);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_dtp_coupling_fwd(const float* sh, const float* cg, const eqf_dtp_paths* paths, float* coupling, int E,
                         void* stream) {
  if (!sh || !cg || !paths || !coupling || paths->npaths < 1 || paths->npaths > EQF_MAX_PATHS) return EQF_E_BADARG;
  if (E <= 0) return 0;
  const long total = (long)E * paths->npaths;
  hipLaunchKernelGGL(coupling_fwd_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, sh, cg, *paths,
                     coupling, total);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_dtp_coupling_bwd(const float* d_coupling, const float* cg, const eqf_dtp_paths* paths, float* d_sh, int E,
                         void* stream) {
  if (!d_coupling || !cg || !paths || !d_sh) return EQF_E_BADARG;
  if (E <= 0) return 0;
  const long total = (long)E * paths->sh_dim;
  hipLaunchKernelGGL(coupling_bwd_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, d_coupling, cg,
                     *paths, d_sh, total);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_dtp_fwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths, float* out, int E,
                void* stream) {
  if (!x || !coupling || !paths || !out) return EQF_E_BADARG;
  for (int p = 0; p < paths->npaths; ++p)
    if (paths->l1[p] > 3) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  hipLaunchKernelGGL(dtp_fwd_kernel, dim3(E, eqf_cdiv(paths->w_numel, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     coupling, w, *paths, out);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_dtp_bwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths, const float* d_out,
                float* dx, float* dw, float* d_coupling, int E, void* stream) {
  if (!x || !coupling || !paths || !d_out || !dx) return EQF_E_BADARG;
  InSegs S{};
  for (int p = 0; p < paths->npaths; ++p) {
    if (paths->l1[p] > 3) return EQF_E_UNSUPPORTED;
    bool found = false;
    for (int s = 0; s < S.nseg; ++s) found |= (S.off[s] == paths->in_off[p]);
    if (!found) {
      if (S.nseg >= EQF_MAX_SEG) return EQF_E_UNSUPPORTED;
      S.off[S.nseg] = paths->in_off[p];
      S.mul[S.nseg] = paths->mul[p];
      S.nseg++;
    }
  }
  // input channels no path reads still need dx = 0: the caller zero-fills dx (see ops.py); here we cover fed segments
  S.cum[0] = 0;
  for (int s = 0; s < S.nseg; ++s) S.cum[s + 1] = S.cum[s] + S.mul[s];
  if (E <= 0) return 0;
  const size_t lds = d_coupling ? sizeof(float) * paths->m_numel : 0;
  hipLaunchKernelGGL(dtp_bwd_kernel, dim3(E), dim3(256), lds, (hipStream_t)stream, x, coupling, w, *paths, S, d_out, dx,
                     dw, d_coupling);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_alpha_fwd(const float* a, const float* alpha_dot, float* logit, int E, int H, int Kh, float c, void* stream) {
  if (!a || !alpha_dot || !logit) return EQF_E_BADARG;
  if (!(Kh == 8 || Kh == 16 || Kh == 32 || Kh == 64)) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  const long total = (long)E * H * Kh;
  hipLaunchKernelGGL(alpha_fwd_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a, alpha_dot,
                     logit, total, H * Kh, Kh, c);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_alpha_bwd(const float* a, const float* alpha_dot, const float* d_logit, float* da, float* d_alpha_dot, int E,
                  int H, int Kh, float c, void* stream) {
  if (!a || !alpha_dot || !d_logit || !da || !d_alpha_dot) return EQF_E_BADARG;
  if (E <= 0) return 0;
  if ((H * Kh) % 4 == 0 && Kh % 4 == 0 && H * Kh <= 256 && 256 % (H * Kh / 4) == 0) {
    const int CH4 = 64;
    hipLaunchKernelGGL(alpha_bwd4_kernel, dim3(eqf_cdiv(E, CH4)), dim3(256), 0, (hipStream_t)stream, a, alpha_dot, d_logit,
                       da, d_alpha_dot, E, H * Kh, Kh, c, CH4);
    EQF_CHECK_LAUNCH();
    return 0;
  }
  const int CH = 16;  // few edges per thread (serial loop), one atomic per column and workgroup
  hipLaunchKernelGGL(alpha_bwd_kernel, dim3(eqf_cdiv(H * Kh, 128), eqf_cdiv(E, CH)), dim3(128), 0, (hipStream_t)stream,
                     a, alpha_dot, d_logit, da, d_alpha_dot, E, H * Kh, Kh, c, CH);
  EQF_CHECK_LAUNCH();
  return 0;
}

static int attn_aggregate_fwd_impl(const float* logit, const float* value, const int* row_ptr, float* alpha, float* out, int N,
                                   int H, const eqf_irreps* irreps, float drop_p, unsigned long long seed,
                                   const unsigned long long* seed_off, void* stream) {
  if (!logit || !value || !row_ptr || !alpha || !out || !irreps || H < 1) return EQF_E_BADARG;
  int err;
  const HeadTab T = make_headtab(*irreps, H, &err);
  if (err) return err;
  if (N <= 0) return 0;
  // HBM-bound kernel: timed for the "HBM GB/s on the scatter" figure of bench.py (bytes are filled in there, the edge
  // count lives on the device)
  const int pid = eqf_prof_begin("attn_fwd", (hipStream_t)stream, 0.0, 0.0);
  if (T.G <= 32)
    hipLaunchKernelGGL(attn_fwd_half_kernel, dim3(N), dim3(64 * H), 0, (hipStream_t)stream, logit, value, row_ptr, alpha,
                       out, T, drop_p, seed, seed_off);
  else
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(N), dim3(64 * H), 0, (hipStream_t)stream, logit, value, row_ptr, alpha, out,
                       T, drop_p, seed, seed_off);
  eqf_prof_end(pid, (hipStream_t)stream);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_attn_aggregate_fwd(const float* logit, const float* value, const int* row_ptr, float* alpha, float* out, int N,
                           int H, const eqf_irreps* irreps, float drop_p, unsigned long long seed, void* stream) {
  return attn_aggregate_fwd_impl(logit, value, row_ptr, alpha, out, N, H, irreps, drop_p, seed, nullptr, stream);
}
int eqf_attn_aggregate_fwd_dseed(const float* logit, const float* value, const int* row_ptr, float* alpha, float* out, int N,
                                 int H, const eqf_irreps* irreps, float drop_p, unsigned long long seed,
                                 const unsigned long long* seed_offset, void* stream) {
  return attn_aggregate_fwd_impl(logit, value, row_ptr, alpha, out, N, H, irreps, drop_p, seed, seed_offset, stream);
}

static int attn_aggregate_bwd_impl(const float* alpha, const float* value, const int* row_ptr, const float* d_out,
                                   float* d_value, float* d_logit, int N, int H, const eqf_irreps* irreps, float drop_p,
                                   unsigned long long seed, const unsigned long long* seed_off, void* stream) {
  if (!alpha || !value || !row_ptr || !d_out || !d_value || !d_logit || !irreps || H < 1) return EQF_E_BADARG;
  int err;
  const HeadTab T = make_headtab(*irreps, H, &err);
  if (err) return err;
  if (N <= 0) return 0;
  const int pid = eqf_prof_begin("attn_bwd", (hipStream_t)stream, 0.0, 0.0);
  if (T.G <= 32)
    hipLaunchKernelGGL(attn_bwd_half_kernel, dim3(N), dim3(64 * H), 0, (hipStream_t)stream, alpha, value, row_ptr, d_out,
                       d_value, d_logit, T, drop_p, seed, seed_off);
  else
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(N), dim3(64 * H), 0, (hipStream_t)stream, alpha, value, row_ptr, d_out,
                       d_value, d_logit, T, drop_p, seed, seed_off);
  eqf_prof_end(pid, (hipStream_t)stream);
  EQF_CHECK_LAUNCH();
  return 0;
}
int eqf_attn_aggregate_bwd(const float* alpha, const float* value, const int* row_ptr, const float* d_out,
                           float* d_value, float* d_logit, int N, int H, const eqf_irreps* irreps, float drop_p,
                           unsigned long long seed, void* stream) {
  return attn_aggregate_bwd_impl(alpha, value, row_ptr, d_out, d_value, d_logit, N, H, irreps, drop_p, seed, nullptr, stream);
}
int eqf_attn_aggregate_bwd_dseed(const float* alpha, const float* value, const int* row_ptr, const float* d_out,
                                 float* d_value, float* d_logit, int N, int H, const eqf_irreps* irreps, float drop_p,
                                 unsigned long long seed, const unsigned long long* seed_offset, void* stream) {
  return attn_aggregate_bwd_impl(alpha, value, row_ptr, d_out, d_value, d_logit, N, H, irreps, drop_p, seed, seed_offset, stream);
}

}  // extern "C"
