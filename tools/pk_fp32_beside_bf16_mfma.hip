// Stand-alone reproducer for the round-2 finding behind DESIGN.md 3.1 ("no packed-FP32 VALU instruction in a kernel that
// issues bf16 MFMAs"): does v_pk_fma_f32 of one wave return wrong lanes while the OTHER wave of its SIMD runs
// v_mfma_f32_32x32x16_bf16?   hipcc --offload-arch=gfx950 -O2 tools/pk_fp32_beside_bf16_mfma.hip -o /tmp/pk && /tmp/pk
// 512-thread workgroups put two waves on every SIMD: waves 0-3 run the partner loop (bf16 MFMA / fp32 MFMA / idle), waves
// 4-7 evaluate the same FMA twice on the same registers (packed or scalar) and count lanes whose two results differ, or
// differ from fmaf().  Self-checking: prints one line per (partner, VALU flavour) and exits 1 if any packed run
// mismatched while every scalar run was clean (= the finding reproduces), 0 if nothing mismatched (= retract the rule).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PARTNER, bool PACKED>  // PARTNER: 0 idle, 1 bf16 MFMA, 2 fp32 MFMA
__global__ __launch_bounds__(512) void k(int iters, unsigned* bad, unsigned* lanes, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    f32x16 acc = {};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)(0.01f * (lane + j)), b[j] = (__bf16)(0.02f * (lane - j));
    for (int it = 0; it < iters * 4; ++it) {
      if (PARTNER == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      if (PARTNER == 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(0.01f * lane, 0.5f, acc, 0, 0, 0);
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
    return;
  }
  unsigned nbad = 0;
  float s = 0.f;
  for (int it = 0; it < iters; ++it) {
    const f2 x = {1.0f + 1e-3f * ((it * 7 + lane) & 1023), 0.5f + 1e-3f * ((it * 13 + lane) & 511)};
    const f2 y = {0.75f + 1e-4f * ((it * 3 + lane) & 255), 1.25f - 1e-4f * ((it * 5 + lane) & 127)};
    const f2 z = {0.1f * (lane & 7), -0.2f * (lane & 3)};
    f2 r1, r2;
    if (PACKED) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(x), "v"(y), "v"(z));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r2) : "v"(x), "v"(y), "v"(z));
    } else {
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1.x) : "v"(x.x), "v"(y.x), "v"(z.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1.y) : "v"(x.y), "v"(y.y), "v"(z.y));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r2.x) : "v"(x.x), "v"(y.x), "v"(z.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r2.y) : "v"(x.y), "v"(y.y), "v"(z.y));
    }
    const float e0 = __builtin_fmaf(x.x, y.x, z.x), e1 = __builtin_fmaf(x.y, y.y, z.y);
    if (r1.x != r2.x || r1.y != r2.y || r1.x != e0 || r1.y != e1) ++nbad, atomicOr(&lanes[lane >> 5], 1u << (lane & 31));
    s += r1.x + r2.y;
  }
  if (nbad) atomicAdd(bad, nbad);
  if (s == 123.456f) sink[1] = s;
}

template <int PARTNER, bool PACKED>
unsigned run(const char* tag, unsigned* d, float* sink) {
  hipMemset(d, 0, 16);
  hipLaunchKernelGGL((k<PARTNER, PACKED>), dim3(512), dim3(512), 0, 0, 200000, d, d + 1, sink);  // 2 workgroups per CU
  hipDeviceSynchronize();
  unsigned h[3];
  hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("%-44s mismatching evaluations %u  lane mask %08x%08x\n", tag, h[0], h[2], h[1]);
  return h[0];
}

int main() {
  unsigned* d;
  float* sink;
  hipMalloc(&d, 16), hipMalloc(&sink, 8);
  unsigned pk = 0, sc = 0;
  pk += run<1, true>("v_pk_fma_f32 beside bf16 MFMA", d, sink);
  sc += run<1, false>("v_fma_f32    beside bf16 MFMA", d, sink);
  pk += run<2, true>("v_pk_fma_f32 beside fp32 MFMA", d, sink);
  pk += run<0, true>("v_pk_fma_f32 beside an idle partner", d, sink);
  sc += run<2, false>("v_fma_f32    beside fp32 MFMA", d, sink);
  printf("packed mismatches %u, scalar mismatches %u -> %s\n", pk, sc,
         pk && !sc ? "REPRODUCED" : (pk || sc ? "inconclusive" : "NOT reproduced"));
  return pk && !sc ? 1 : 0;
}
