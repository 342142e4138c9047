// Optimizer step of the train loop on ONE flat fp32 buffer: global gradient norm (clip_grad_norm_), AdamW with a
// per-element weight-decay value, and the EMA of the weights, fused into two launches.
// [ref: optim_factory.py:27-42,126-127 (AdamW, name-based no-weight-decay groups), engine.py:73-90 (clip, step,
//  model_ema.update); torch.optim.AdamW / timm.utils.ModelEmaV2 semantics restated in oracle/optim.py]
#include <hip/hip_runtime.h>
#include <cmath>
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  const long n4 = n >> 2;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long j = n4 << 2; j < n; ++j) acc += g[j] * g[j];
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  const float* wd;     // per-element weight decay (0 for the no-decay group); NEGATIVE = the element's parameter has no
                       // gradient this step: torch.optim.AdamW skips it entirely (only the EMA follows the weight)
  float* ema;          // may be null
  const float* sumsq;  // may be null (no clipping)
  long n;
  float lr, beta1, beta2, eps, bc1, bc2_sqrt, max_norm, ema_decay;
  const float* hyper;  // may be null; else {lr, bc1, bc2_sqrt} are read from this device array (a launch captured in a HIP graph:
                       // the by-value ones are frozen at capture)
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float wd, float* ema, const AdamArgs& a,
                                         float clip) {
  if (wd < 0.f) {
    if (ema) *ema = a.ema_decay * *ema + (1.f - a.ema_decay) * p;
    return;
  }
  g *= clip;
  p *= 1.f - a.lr * wd;                       // decoupled weight decay
  m = a.beta1 * m + (1.f - a.beta1) * g;      // torch: exp_avg.lerp_(grad, 1 - beta1)
  v = a.beta2 * v + (1.f - a.beta2) * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= (a.lr / a.bc1) * (m / denom);
  if (ema) *ema = a.ema_decay * *ema + (1.f - a.ema_decay) * p;
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs a_in) {
  AdamArgs a = a_in;
  if (a.hyper) a.lr = a.hyper[0], a.bc1 = a.hyper[1], a.bc2_sqrt = a.hyper[2];
  float clip = 1.f;
  if (a.sumsq) {  // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float c = a.max_norm / (sqrtf(*a.sumsq) + 1e-6f);
    clip = c < 1.f ? c : 1.f;
  }
  const long n4 = a.n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 p = reinterpret_cast<float4*>(a.p)[i], m = reinterpret_cast<float4*>(a.m)[i],
           v = reinterpret_cast<float4*>(a.v)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i], wd = reinterpret_cast<const float4*>(a.wd)[i];
    float4 e = a.ema ? reinterpret_cast<float4*>(a.ema)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    adam_one(p.x, g.x, m.x, v.x, wd.x, a.ema ? &e.x : nullptr, a, clip);
    adam_one(p.y, g.y, m.y, v.y, wd.y, a.ema ? &e.y : nullptr, a, clip);
    adam_one(p.z, g.z, m.z, v.z, wd.z, a.ema ? &e.z : nullptr, a, clip);
    adam_one(p.w, g.w, m.w, v.w, wd.w, a.ema ? &e.w : nullptr, a, clip);
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
    if (a.ema) reinterpret_cast<float4*>(a.ema)[i] = e;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long j = n4 << 2; j < a.n; ++j) adam_one(a.p[j], a.g[j], a.m[j], a.v[j], a.wd[j], a.ema ? a.ema + j : nullptr, a, clip);
}

}  // namespace

extern "C" {

int eqf_sumsq(const float* g, long n, float* out, void* stream) {
  if (!g || !out || n < 0) return EQF_E_BADARG;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return 0;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, g, n, out);
  EQF_CHECK_LAUNCH();
  return 0;
}

static int adamw_launch(float* p, const float* g, float* m, float* v, const float* wd, float* ema, const float* sumsq, long n,
                        float lr, float beta1, float beta2, float eps, int step, const float* hyper, float max_norm,
                        float ema_decay, void* stream) {
  if (!p || !g || !m || !v || !wd || n < 0 || (!hyper && step < 1)) return EQF_E_BADARG;
  if (n == 0) return 0;
  AdamArgs a;
  a.p = p, a.g = g, a.m = m, a.v = v, a.wd = wd, a.ema = ema, a.sumsq = sumsq, a.n = n;
  a.lr = lr, a.beta1 = beta1, a.beta2 = beta2, a.eps = eps;
  a.hyper = hyper;
  a.bc1 = hyper ? 1.f : (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = hyper ? 1.f : (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.max_norm = max_norm, a.ema_decay = ema_decay;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_adamw_step(float* p, const float* g, float* m, float* v, const float* wd, float* ema, const float* sumsq, long n,
                   float lr, float beta1, float beta2, float eps, int step, float max_norm, float ema_decay,
                   void* stream) {
  return adamw_launch(p, g, m, v, wd, ema, sumsq, n, lr, beta1, beta2, eps, step, nullptr, max_norm, ema_decay, stream);
}

int eqf_adamw_step_dev(float* p, const float* g, float* m, float* v, const float* wd, float* ema, const float* sumsq, long n,
                       const float* hyper, float beta1, float beta2, float eps, float max_norm, float ema_decay,
                       void* stream) {
  if (!hyper) return EQF_E_BADARG;
  return adamw_launch(p, g, m, v, wd, ema, sumsq, n, 0.f, beta1, beta2, eps, 0, hyper, max_norm, ema_decay, stream);
}

}  // extern "C"
