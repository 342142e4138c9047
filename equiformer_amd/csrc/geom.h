// Edge geometry shared by the first-order kernels (graph.hip, T = float) and the second-derivative kernel
// (second.hip, T = Dual): component-normalised real spherical harmonics of the unit edge vector, l <= 3, (x, y, z)
// order with y the polar axis, and the gradient of <d_sh, sh(vec)> + d_len |vec| wrt vec.
// [ref: e3nn 0.4.4 o3.spherical_harmonics(normalize=True, normalization='component'), called at
//  nets/graph_attention_transformer.py:868-870; edge length :871]
//
// Evaluated on dual numbers v + eps c (eps^2 = 0) the same code yields directional derivatives: the eps part of
// sh(vec + eps c) is J_sh c, and the eps part of the gradient is the Hessian-vector product that
// `loss.backward()` needs when forces were taken with create_graph=True
// [ref: nets/graph_attention_transformer_md17.py:318-325].
#pragma once
#include "common.h"

struct Dual {
  float v, e;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.e + b.e}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.e - b.e}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.v * b.e + a.e * b.v}; }
__device__ __forceinline__ Dual operator+(Dual a, float b) { return {a.v + b, a.e}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return {a.v - b, a.e}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return {a.v * b, a.e * b}; }
__device__ __forceinline__ Dual operator*(float b, Dual a) { return {a.v * b, a.e * b}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.e}; }
__device__ __forceinline__ Dual& operator+=(Dual& a, Dual b) {
  a.v += b.v, a.e += b.e;
  return a;
}

template <typename T>
struct GeomOps;
template <>
struct GeomOps<float> {
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ float norm(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }
  static __device__ __forceinline__ float inv_clamped(float L) { return 1.f / fmaxf(L, 1e-12f); }
};
template <>
struct GeomOps<Dual> {
  static __device__ __forceinline__ Dual zero() { return {0.f, 0.f}; }
  static __device__ __forceinline__ Dual norm(Dual x, Dual y, Dual z) {
    const float L = sqrtf(x.v * x.v + y.v * y.v + z.v * z.v);
    const float d = (x.v * x.e + y.v * y.e + z.v * z.e) / fmaxf(L, 1e-12f);
    return {L, d};
  }
  static __device__ __forceinline__ Dual inv_clamped(Dual L) {
    if (L.v < 1e-12f) return {1e12f, 0.f};  // clamped branch: constant
    const float i = 1.f / L.v;
    return {i, -L.e * i * i};
  }
};

// raw (norm-normalised) harmonics of degree 2 and their gradients wrt the unit vector
template <typename T>
struct SH2T {
  T v[5];
  T g[5][3];
};
template <typename T>
__device__ __forceinline__ SH2T<T> sh2_of(T x, T y, T z) {
  const float s3 = 1.7320508075688772f;
  const T zero = GeomOps<T>::zero();
  SH2T<T> s;
  s.v[0] = s3 * (x * z), s.g[0][0] = s3 * z, s.g[0][1] = zero, s.g[0][2] = s3 * x;
  s.v[1] = s3 * (x * y), s.g[1][0] = s3 * y, s.g[1][1] = s3 * x, s.g[1][2] = zero;
  s.v[2] = y * y - 0.5f * (x * x + z * z), s.g[2][0] = -x, s.g[2][1] = 2.f * y, s.g[2][2] = -z;
  s.v[3] = s3 * (y * z), s.g[3][0] = zero, s.g[3][1] = s3 * z, s.g[3][2] = s3 * y;
  s.v[4] = (0.5f * s3) * (z * z - x * x), s.g[4][0] = -(s3 * x), s.g[4][1] = zero, s.g[4][2] = s3 * z;
  return s;
}

constexpr float kShC1 = 1.7320508075688772f, kShC2 = 2.23606797749979f, kShC3 = 2.6457513110645907f;
constexpr float kShA = 0.9128709291752769f /* sqrt(5/6) */, kShB5 = 2.23606797749979f, kShC38 = 0.6123724356957945f;

// o[0 .. (lmax+1)^2): harmonics of the direction of (vx, vy, vz); returns |vec|
template <typename T>
__device__ __forceinline__ T geom_sh(T vx, T vy, T vz, int lmax, T* o) {
  const T L = GeomOps<T>::norm(vx, vy, vz);
  const T inv = GeomOps<T>::inv_clamped(L);
  const T x = vx * inv, y = vy * inv, z = vz * inv;
  o[0] = GeomOps<T>::zero() + 1.f;
  if (lmax >= 1) o[1] = kShC1 * x, o[2] = kShC1 * y, o[3] = kShC1 * z;
  if (lmax >= 2) {
    const SH2T<T> s2 = sh2_of(x, y, z);
#pragma unroll
    for (int i = 0; i < 5; ++i) o[4 + i] = kShC2 * s2.v[i];
    if (lmax >= 3) {
      const T y2 = y * y, x2z2 = x * x + z * z;
      o[9] = (kShC3 * kShA) * (s2.v[0] * z + s2.v[4] * x);
      o[10] = (kShC3 * kShB5) * (s2.v[0] * y);
      o[11] = (kShC3 * kShC38) * ((4.f * y2 - x2z2) * x);
      o[12] = (kShC3 * 0.5f) * (y * (2.f * y2 - 3.f * x2z2));
      o[13] = (kShC3 * kShC38) * (z * (4.f * y2 - x2z2));
      o[14] = (kShC3 * kShB5) * (s2.v[4] * y);
      o[15] = (kShC3 * kShA) * (s2.v[4] * z - s2.v[0] * x);
    }
  }
  return L;
}

// gradient of  sum_j g[j] sh_j(vec) + gl |vec|  wrt vec  (g may be null: no harmonics term; has_len: gl valid)
template <typename T>
__device__ __forceinline__ void geom_grad(T vx, T vy, T vz, int lmax, const float* g, bool has_len, float gl, T& ox,
                                          T& oy, T& oz) {
  const T L = GeomOps<T>::norm(vx, vy, vz);
  const T inv = GeomOps<T>::inv_clamped(L);
  const T x = vx * inv, y = vy * inv, z = vz * inv;
  T gx = GeomOps<T>::zero(), gy = gx, gz = gx;  // gradient wrt the unit vector
  if (g) {
    if (lmax >= 1) {
      gx = gx + kShC1 * g[1], gy = gy + kShC1 * g[2], gz = gz + kShC1 * g[3];
    }
    if (lmax >= 2) {
      const SH2T<T> s2 = sh2_of(x, y, z);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float t = kShC2 * g[4 + i];
        gx += t * s2.g[i][0], gy += t * s2.g[i][1], gz += t * s2.g[i][2];
      }
      if (lmax >= 3) {
        const T y2 = y * y, x2z2 = x * x + z * z;
        const T q = 4.f * y2 - x2z2;
        float t;
        // t0 = a (s0 z + s4 x)
        t = kShC3 * kShA * g[9];
        gx += t * (s2.g[0][0] * z + s2.g[4][0] * x + s2.v[4]);
        gy += t * (s2.g[0][1] * z + s2.g[4][1] * x);
        gz += t * (s2.g[0][2] * z + s2.v[0] + s2.g[4][2] * x);
        // t1 = sqrt5 s0 y
        t = kShC3 * kShB5 * g[10];
        gx += t * (s2.g[0][0] * y), gy += t * (s2.g[0][1] * y + s2.v[0]), gz += t * (s2.g[0][2] * y);
        // t2 = c38 q x
        t = kShC3 * kShC38 * g[11];
        gx += t * (q - 2.f * (x * x)), gy += t * (8.f * (y * x)), gz += t * (-2.f * (z * x));
        // t3 = .5 y (2 y2 - 3 x2z2)
        t = kShC3 * 0.5f * g[12];
        gx += t * (-6.f * (x * y)), gy += t * (2.f * y2 - 3.f * x2z2 + 4.f * y2), gz += t * (-6.f * (z * y));
        // t4 = c38 z q
        t = kShC3 * kShC38 * g[13];
        gx += t * (-2.f * (x * z)), gy += t * (8.f * (y * z)), gz += t * (q - 2.f * (z * z));
        // t5 = sqrt5 s4 y
        t = kShC3 * kShB5 * g[14];
        gx += t * (s2.g[4][0] * y), gy += t * (s2.g[4][1] * y + s2.v[4]), gz += t * (s2.g[4][2] * y);
        // t6 = a (s4 z - s0 x)
        t = kShC3 * kShA * g[15];
        gx += t * (s2.g[4][0] * z - s2.g[0][0] * x - s2.v[0]);
        gy += t * (s2.g[4][1] * z - s2.g[0][1] * x);
        gz += t * (s2.g[4][2] * z + s2.v[4] - s2.g[0][2] * x);
      }
    }
  }
  // d unit / d vec = (I - u u^T) / L
  const T ug = x * gx + y * gy + z * gz;
  ox = (gx - x * ug) * inv, oy = (gy - y * ug) * inv, oz = (gz - z * ug) * inv;
  if (has_len) ox = ox + gl * x, oy = oy + gl * y, oz = oz + gl * z;
}
