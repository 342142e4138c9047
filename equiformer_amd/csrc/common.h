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
__device__ __forceinline__ long row_off2(int i, int d, int ld, int inner) {
  int q = i / d;
  int r = i - q * d;
  return (long)q * ld + (long)r * inner;
}

// row offset of segment s and total row length of an irreps descriptor (CF layout)
__host__ __device__ inline int irreps_dim(const eqf_irreps& ir) {
  int D = 0;
  for (int s = 0; s < ir.nseg; ++s) D += ir.mul[s] * (2 * ir.l[s] + 1);
  return D;
}
