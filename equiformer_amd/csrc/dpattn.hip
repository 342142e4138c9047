// Dot-product attention over the dst-sorted radius graph (the dp_attention_transformer ablation family)
// [ref: nets/dp_attention_transformer.py:45-66 ScaleFactor, :131-152 DotProductAttention.forward].
//
// The key/value tensor product emits ONE row per edge holding 2H heads: in every irreps segment of degree l the
// channel index is (head, channel-of-head) with heads 0..H-1 = keys and H..2H-1 = values (Vec2AttnHeads on
// irreps_head * 2H, then narrow).  In the channel-fastest row layout used here a segment is [2l+1][2*H*mh], so keys are
// the first H*mh channels of each m-row and values the last H*mh.
//   eqf_kv_split / eqf_kv_merge   : kv [E, 2D] <-> k [E, D], v [E, D] (both follow the H-head irreps; pure copies)
//   eqf_dp_logits_fwd             : logit[e,h] = sum_{s,m,c} scale_s * q[dst[e], s,m,h,c] * k[e, s,m,h,c]
//                                   scale_s = 1/sqrt(num_irreps(head)) / sqrt(2 l_s + 1)   (ScaleFactor folded in)
//   eqf_dp_logits_bwd             : dk[e,..] = scale * dlogit[e,h] * q[dst[e],..];  dq[n,..] = scale * sum_{e in seg(n)}
//                                   dlogit[e,h] * k[e,..]  (one workgroup per destination row: no atomics)
// All HBM-bound row streams (1920-byte rows at the QM9 width); the softmax + weighted aggregation that follows is the
// shared eqf_attn_aggregate_* kernel.  The logits are bilinear in (q, k), so their second-order terms reuse these entry
// points (ops._DpLogitsBwd).
#include "common.h"

namespace {

struct DpTab {
  int nseg, H, D, G;  // D = floats per q/k/v row, G = float4 groups per row
  int off[EQF_MAX_SEG], mul[EQF_MAX_SEG], d[EQF_MAX_SEG], gcum[EQF_MAX_SEG + 1];
  float scale[EQF_MAX_SEG];
};

DpTab make_dptab(const eqf_irreps& ir, int H, int* err) {
  DpTab T;
  *err = 0;
  T.nseg = ir.nseg, T.H = H;
  int off = 0, g = 0, num_irreps = 0;
  if (ir.nseg < 1 || ir.nseg > EQF_MAX_SEG || H < 1 || H > 8) *err = EQF_E_UNSUPPORTED;
  for (int s = 0; s < ir.nseg && !*err; ++s) {
    const int mul = ir.mul[s], d = 2 * ir.l[s] + 1;
    if (mul % H || (mul / H) % 4) { *err = EQF_E_UNSUPPORTED; break; }
    T.off[s] = off, T.mul[s] = mul, T.d[s] = d, T.gcum[s] = g;
    off += mul * d, g += mul * d / 4;
    num_irreps += mul / H;
  }
  T.gcum[ir.nseg] = g;
  T.D = off, T.G = g;
  for (int s = 0; s < ir.nseg && !*err; ++s)
    T.scale[s] = (float)(1.0 / (sqrt((double)num_irreps) * sqrt((double)T.d[s])));
  return T;
}

// float4 group g of a q/k/v row -> (segment, float offset in the row, head)
__device__ __forceinline__ void locate(const DpTab& T, int g, int& s, int& col, int& h) {
  s = 0;
  while (s + 1 < T.nseg && g >= T.gcum[s + 1]) ++s;
  const int j4 = 4 * (g - T.gcum[s]);  // float index inside the segment = m * mul + c
  col = T.off[s] + j4;
  h = (j4 % T.mul[s]) / (T.mul[s] / T.H);
}

// float offset of the same (m, c) inside the 2D-wide kv row; values sit `mul` further
__device__ __forceinline__ int kv_col(const DpTab& T, int s, int col) {
  const int j = col - T.off[s];
  const int m = j / T.mul[s], c = j - m * T.mul[s];
  return 2 * T.off[s] + m * 2 * T.mul[s] + c;
}

__global__ __launch_bounds__(256) void kv_split_kernel(const float* __restrict__ kv, float* __restrict__ k,
                                                       float* __restrict__ v, const DpTab T, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long e = idx / T.G;
  const int g = (int)(idx - e * T.G);
  int s, col, h;
  locate(T, g, s, col, h);
  const int kc = kv_col(T, s, col);
  const float* row = kv + e * 2L * T.D;
  *reinterpret_cast<float4*>(k + e * (long)T.D + col) = *reinterpret_cast<const float4*>(row + kc);
  *reinterpret_cast<float4*>(v + e * (long)T.D + col) = *reinterpret_cast<const float4*>(row + kc + T.mul[s]);
}

__global__ __launch_bounds__(256) void kv_merge_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                       float* __restrict__ kv, const DpTab T, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long e = idx / T.G;
  const int g = (int)(idx - e * T.G);
  int s, col, h;
  locate(T, g, s, col, h);
  const int kc = kv_col(T, s, col);
  float* row = kv + e * 2L * T.D;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(row + kc) = k ? *reinterpret_cast<const float4*>(k + e * (long)T.D + col) : z;
  *reinterpret_cast<float4*>(row + kc + T.mul[s]) = v ? *reinterpret_cast<const float4*>(v + e * (long)T.D + col) : z;
}

constexpr int DP_SLOTS = 4;  // a wave covers rows of up to 256 float4 groups (1024 floats)

// one wavefront per edge: every lane owns up to DP_SLOTS float4 groups, each inside one head
__global__ __launch_bounds__(256) void dp_logits_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const int* __restrict__ dst, float* __restrict__ logit,
                                                            const DpTab T, int E) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= E) return;
  const float* qr = q + (long)dst[e] * T.D;
  const float* kr = k + (long)e * T.D;
  float acc[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) acc[h] = 0.f;
#pragma unroll
  for (int t = 0; t < DP_SLOTS; ++t) {
    const int g = lane + 64 * t;
    if (g < T.G) {
      int s, col, h;
      locate(T, g, s, col, h);
      const float4 a = *reinterpret_cast<const float4*>(qr + col);
      const float4 b = *reinterpret_cast<const float4*>(kr + col);
      const float p = T.scale[s] * (a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w);
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) acc[hh] += (hh == h) ? p : 0.f;
    }
  }
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    if (h < T.H) {
      const float r = wave_sum(acc[h]);
      if (lane == 0) logit[(long)e * T.H + h] = r;
    }
  }
}

// one workgroup per destination node, one thread per float4 group of the row
__global__ void dp_logits_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                     const float* __restrict__ dlogit, const int* __restrict__ row_ptr,
                                     float* __restrict__ dq, float* __restrict__ dk, const DpTab T) {
  const int n = blockIdx.x, g = threadIdx.x;
  if (g >= T.G) return;
  int s, col, h;
  locate(T, g, s, col, h);
  const float sc = T.scale[s];
  const int beg = row_ptr[n], end = row_ptr[n + 1];
  float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), acc = qv;
  if (dk) qv = *reinterpret_cast<const float4*>(q + (long)n * T.D + col);
  for (int e = beg; e < end; ++e) {
    const float w = sc * dlogit[(long)e * T.H + h];
    if (dk) *reinterpret_cast<float4*>(dk + (long)e * T.D + col) = make_float4(w * qv.x, w * qv.y, w * qv.z, w * qv.w);
    if (dq) {
      const float4 kv = *reinterpret_cast<const float4*>(k + (long)e * T.D + col);
      acc.x = fmaf(w, kv.x, acc.x), acc.y = fmaf(w, kv.y, acc.y);
      acc.z = fmaf(w, kv.z, acc.z), acc.w = fmaf(w, kv.w, acc.w);
    }
  }
  if (dq) *reinterpret_cast<float4*>(dq + (long)n * T.D + col) = acc;
}

}  // namespace

int eqf_kv_split(const float* kv, float* k, float* v, int E, int H, const eqf_irreps* irreps, void* stream) {
  if (!kv || !k || !v || !irreps) return EQF_E_BADARG;
  int err;
  const DpTab T = make_dptab(*irreps, H, &err);
  if (err) return err;
  if (E <= 0) return 0;
  const long total = (long)E * T.G;
  hipLaunchKernelGGL(kv_split_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, kv, k, v, T, total);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_kv_merge(const float* k, const float* v, float* kv, int E, int H, const eqf_irreps* irreps, void* stream) {
  if (!kv || !irreps) return EQF_E_BADARG;
  int err;
  const DpTab T = make_dptab(*irreps, H, &err);
  if (err) return err;
  if (E <= 0) return 0;
  const long total = (long)E * T.G;
  hipLaunchKernelGGL(kv_merge_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, k, v, kv, T, total);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_dp_logits_fwd(const float* q, const float* k, const int* dst, float* logit, int E, int H,
                      const eqf_irreps* irreps, void* stream) {
  if (!q || !k || !dst || !logit || !irreps) return EQF_E_BADARG;
  int err;
  const DpTab T = make_dptab(*irreps, H, &err);
  if (err) return err;
  if (T.G > 64 * DP_SLOTS) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  hipLaunchKernelGGL(dp_logits_fwd_kernel, dim3(eqf_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, q, k, dst, logit, T,
                     E);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_dp_logits_bwd(const float* q, const float* k, const float* d_logit, const int* row_ptr, float* dq, float* dk,
                      int N, int H, const eqf_irreps* irreps, void* stream) {
  if (!d_logit || !row_ptr || !irreps || (dq && !k) || (dk && !q)) return EQF_E_BADARG;
  int err;
  const DpTab T = make_dptab(*irreps, H, &err);
  if (err) return err;
  if (T.G > 1024) return EQF_E_UNSUPPORTED;
  if (N <= 0 || (!dq && !dk)) return 0;
  const int threads = 64 * eqf_cdiv(T.G, 64);
  hipLaunchKernelGGL(dp_logits_bwd_kernel, dim3(N), dim3(threads), 0, (hipStream_t)stream, q, k, d_logit, row_ptr, dq, dk,
                     T);
  EQF_CHECK_LAUNCH();
  return 0;
}
