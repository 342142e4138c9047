// Shared device helpers for libequiformer_hip.so (gfx950 / CDNA4 only: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/equiformer_hip.h"

#define EQF_CHECK_LAUNCH()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

static inline int eqf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// reduce over aligned groups of 32 lanes
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// two-level row addressing (see eqf_rows in the public header)
// (an integer division by a run-time d costs ~30 VALU instructions, and on gfx950 the fp32 MFMA shares the ALUs with
// the VALU: one such division per stored element made the epilogue of the radial-MLP GEMM as expensive as its MFMAs.
// Plain rows (d == 1) therefore take a wave-uniform fast path; hot loops with d > 1 use SmallDiv below.)
__device__ __forceinline__ long row_off2(int i, int d, int ld, int inner) {
  if (d == 1) return (long)i * ld;
  int q;
  // d = 2l+1 <= 16: (i + 0.5) / d is at least 1 / (2 d) away from every integer, the float product errs by
  // < i * 2^-22 / d  ->  exact for i < 2^21; larger indices take the real division
  if (i < (1 << 21)) q = (int)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)d));
  else q = i / d;
  int r = i - q * d;
  return (long)q * ld + (long)r * inner;
}

// exact t / d for 0 <= t < 1024, 1 <= d <= 16 in three VALU instructions: (t + 0.5) / d is at least 1 / (2 d) away from
// every integer, far more than the rounding error of the float product
struct SmallDiv {
  int d;
  float inv;
  __device__ __forceinline__ explicit SmallDiv(int d_) : d(d_), inv(1.0f / (float)d_) {}
  __device__ __forceinline__ int div(int t) const { return (int)(((float)t + 0.5f) * inv); }
};

// row offset of segment s and total row length of an irreps descriptor (CF layout)
__host__ __device__ inline int irreps_dim(const eqf_irreps& ir) {
  int D = 0;
  for (int s = 0; s < ir.nseg; ++s) D += ir.mul[s] * (2 * ir.l[s] + 1);
  return D;
}
