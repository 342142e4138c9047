// Graph construction and edge geometry: radius graph -> dst-sorted CSR, edge vectors, spherical harmonics,
// radial basis.  Molecules are tiny (N ~ 18-80 atoms), so one workgroup owns one molecule and brute-forces its
// pairs out of L1; the output is already sorted by destination (the order the segmented kernels want).
#include "common.h"
#include "geom.h"

namespace {

__global__ __launch_bounds__(64) void radius_count_kernel(const float* __restrict__ pos, const int* __restrict__ mol_ptr,
                                                          float r2, int max_nbr, int* __restrict__ deg) {
  const int b = blockIdx.x;
  const int n0 = mol_ptr[b], n1 = mol_ptr[b + 1];
  for (int i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    int c = 0;
    for (int j = n0; j < n1; ++j) {
      const float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (j != i && d2 < r2 && c < max_nbr) ++c;
    }
    deg[i] = c;
  }
}

__global__ __launch_bounds__(64) void radius_fill_kernel(const float* __restrict__ pos, const int* __restrict__ mol_ptr,
                                                         float r2, int max_nbr, const int* __restrict__ row_ptr,
                                                         int* __restrict__ src, int* __restrict__ dst) {
  const int b = blockIdx.x;
  const int n0 = mol_ptr[b], n1 = mol_ptr[b + 1];
  for (int i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    int c = 0;
    const int base = row_ptr[i];
    for (int j = n0; j < n1; ++j) {
      const float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (j != i && d2 < r2 && c < max_nbr) {
        src[base + c] = j;
        dst[base + c] = i;
        ++c;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- periodic radius graph
// ocpmodels.common.utils.radius_graph_pbc semantics (un-vendored dependency of the reference; restated in
// oracle/pbc.py): candidates = (centre i, neighbour j of the same structure incl. j == i, image n with |n_k| <= rep_k),
// kept when 1e-4 < |pos_j + n.cell - pos_i|^2 <= r^2; per centre the max_nbr nearest survive (ties: candidate order).
struct PbcCell {
  float a[3][3];
  int rep[3];
};
__device__ __forceinline__ PbcCell pbc_cell(const float* __restrict__ cell, float r) {
  PbcCell c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) c.a[i][k] = cell[3 * i + k];
  const double a1[3] = {c.a[0][0], c.a[0][1], c.a[0][2]}, a2[3] = {c.a[1][0], c.a[1][1], c.a[1][2]},
               a3[3] = {c.a[2][0], c.a[2][1], c.a[2][2]};
  auto cross = [](const double* u, const double* v, double* o) {
    o[0] = u[1] * v[2] - u[2] * v[1], o[1] = u[2] * v[0] - u[0] * v[2], o[2] = u[0] * v[1] - u[1] * v[0];
  };
  double c23[3], c31[3], c12[3];
  cross(a2, a3, c23), cross(a3, a1, c31), cross(a1, a2, c12);
  const double vol = fabs(a1[0] * c23[0] + a1[1] * c23[1] + a1[2] * c23[2]);
  const double* cs[3] = {c23, c31, c12};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double n = sqrt(cs[k][0] * cs[k][0] + cs[k][1] * cs[k][1] + cs[k][2] * cs[k][2]) / vol;
    c.rep[k] = (int)ceil((double)r * n - 1e-12);
  }
  return c;
}

// visit the candidates of centre i in (j, n1, n2, n3) order: f(j, n1, n2, n3, d2, ox, oy, oz)
template <class F>
__device__ __forceinline__ void pbc_visit(const float* __restrict__ pos, const PbcCell& c, int n0, int n1_, int i, float r2,
                                          F&& f) {
  const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  for (int j = n0; j < n1_; ++j) {
    const float dx0 = pos[3 * j] - xi, dy0 = pos[3 * j + 1] - yi, dz0 = pos[3 * j + 2] - zi;
    for (int u = -c.rep[0]; u <= c.rep[0]; ++u)
      for (int v = -c.rep[1]; v <= c.rep[1]; ++v)
        for (int w = -c.rep[2]; w <= c.rep[2]; ++w) {
          const float ox = u * c.a[0][0] + v * c.a[1][0] + w * c.a[2][0];
          const float oy = u * c.a[0][1] + v * c.a[1][1] + w * c.a[2][1];
          const float oz = u * c.a[0][2] + v * c.a[1][2] + w * c.a[2][2];
          const float dx = dx0 + ox, dy = dy0 + oy, dz = dz0 + oz;
          const float d2 = dx * dx + dy * dy + dz * dz;
          if (d2 <= r2 && d2 > 1e-4f) f(j, u, v, w, d2, ox, oy, oz);
        }
  }
}

__global__ __launch_bounds__(64) void pbc_count_kernel(const float* __restrict__ pos, const float* __restrict__ cell,
                                                       const int* __restrict__ mol_ptr, float r, int max_nbr,
                                                       int* __restrict__ cand, int* __restrict__ deg) {
  const int b = blockIdx.x;
  const int n0 = mol_ptr[b], n1 = mol_ptr[b + 1];
  const PbcCell c = pbc_cell(cell + 9 * b, r);
  for (int i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    int n = 0;
    pbc_visit(pos, c, n0, n1, i, r * r, [&](int, int, int, int, float, float, float, float) { ++n; });
    cand[i] = n;
    deg[i] = n < max_nbr ? n : max_nbr;
  }
}

__global__ __launch_bounds__(64) void pbc_fill_kernel(const float* __restrict__ pos, const float* __restrict__ cell,
                                                      const int* __restrict__ mol_ptr, float r, int max_nbr,
                                                      const int* __restrict__ row_ptr, const int* __restrict__ cand_ptr,
                                                      float* __restrict__ scratch_d2, int* __restrict__ src,
                                                      int* __restrict__ dst, int* __restrict__ cell_offsets,
                                                      float* __restrict__ offsets) {
  const int b = blockIdx.x;
  const int n0 = mol_ptr[b], n1 = mol_ptr[b + 1];
  const PbcCell c = pbc_cell(cell + 9 * b, r);
  for (int i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    const int base = row_ptr[i], cnt = row_ptr[i + 1] - base;
    const int cb = cand_ptr[i], ncand = cand_ptr[i + 1] - cb;
    const bool trunc = ncand > cnt;
    if (trunc) {  // distances of all candidates of the row, for the nearest-max_nbr selection
      int k = 0;
      pbc_visit(pos, c, n0, n1, i, r * r,
                [&](int, int, int, int, float d2, float, float, float) { scratch_d2[cb + k++] = d2; });
    }
    int k = 0, o = 0;
    pbc_visit(pos, c, n0, n1, i, r * r, [&](int j, int u, int v, int w, float d2, float ox, float oy, float oz) {
      bool keep = true;
      if (trunc) {  // rank = candidates strictly nearer, or equally near and earlier
        int rank = 0;
        for (int q = 0; q < ncand; ++q) {
          const float dq = scratch_d2[cb + q];
          rank += (dq < d2 || (dq == d2 && q < k)) ? 1 : 0;
        }
        keep = rank < max_nbr;
      }
      ++k;
      if (keep && o < cnt) {
        const int e = base + o++;
        src[e] = j, dst[e] = i;
        if (cell_offsets) cell_offsets[3 * e] = u, cell_offsets[3 * e + 1] = v, cell_offsets[3 * e + 2] = w;
        offsets[3 * e] = ox, offsets[3 * e + 1] = oy, offsets[3 * e + 2] = oz;
      }
    });
  }
}

// ---------------------------------------------------------------------------------------------- CSR bookkeeping
// ptr[g] = first i with seg_of[i] >= g (seg_of ascending), ptr[n_seg] = n; optional max segment length.
__global__ __launch_bounds__(256) void segment_ptr_kernel(const int* __restrict__ seg_of, int n, int n_seg,
                                                          int* __restrict__ ptr, int* __restrict__ max_len) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > n_seg) return;
  auto lower = [&](int key) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (seg_of[mid] < key) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  const int a = lower(g);
  ptr[g] = a;
  if (max_len && g < n_seg) atomicMax(max_len, lower(g + 1) - a);
}

// ptr[0..n] = exclusive scan of counts[0..n-1]; one workgroup (n is a node count), serial chunks + one LDS scan.
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const int* __restrict__ counts, int n, int* __restrict__ ptr,
                                                              int* __restrict__ total) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int i0 = min(n, t * chunk), i1 = min(n, i0 + chunk);
  int s = 0;
  for (int i = i0; i < i1; ++i) s += counts[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = i0; i < i1; ++i) {
    ptr[i] = run;
    run += counts[i];
  }
  if (t == 1023) {
    ptr[n] = part[1023];
    if (total) *total = part[1023];
  }
}

// By-source view of a dst-sorted, molecule-blocked edge list: src_perm = stable argsort of src, src_ptr = CSR offsets
// over the source node.  Edges never cross molecules, so the by-source order is molecule-blocked too and occupies the
// same edge range: one workgroup per molecule, counters in LDS.  Inside one destination row every source occurs at
// most once (radius graphs have no multi-edges), so walking the rows in order and bumping a per-source cursor
// reproduces the stable order without any sort.
constexpr int CSR_MAX_NODES = 16384;  // nodes per molecule (64 KB of LDS cursors)
__global__ __launch_bounds__(256) void csr_by_source_kernel(const int* __restrict__ src, const int* __restrict__ row_ptr,
                                                            const int* __restrict__ mol_ptr, int n_mol,
                                                            int* __restrict__ src_perm, int* __restrict__ src_ptr) {
  extern __shared__ int cur[];  // [nm]
  __shared__ int part[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const int n0 = mol_ptr[b], n1 = mol_ptr[b + 1], nm = n1 - n0;
  const int e0 = row_ptr[n0], e1 = row_ptr[n1];
  for (int i = t; i < nm; i += 256) cur[i] = 0;
  __syncthreads();
  for (int e = e0 + t; e < e1; e += 256) atomicAdd(&cur[src[e] - n0], 1);
  __syncthreads();
  // exclusive scan of cur[0..nm) in place
  const int chunk = (nm + 255) / 256;
  const int i0 = min(nm, t * chunk), i1 = min(nm, i0 + chunk);
  int s = 0;
  for (int i = i0; i < i1; ++i) s += cur[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = e0 + part[t] - s;
  for (int i = i0; i < i1; ++i) {
    const int c = cur[i];
    cur[i] = run;
    src_ptr[n0 + i] = run;
    run += c;
  }
  if (b == n_mol - 1 && t == 0) src_ptr[n1] = e1;
  __syncthreads();
  for (int d = n0; d < n1; ++d) {
    const int r0 = row_ptr[d], r1 = row_ptr[d + 1];
    for (int e = r0 + t; e < r1; e += 256) {
      const int sn = src[e] - n0;
      const int pos = cur[sn];
      cur[sn] = pos + 1;
      src_perm[pos] = e;
    }
    __syncthreads();
  }
}

// spherical harmonics / edge length and their gradient: geom.h (shared with the second-derivative kernel)
__global__ __launch_bounds__(256) void edge_geom_fwd_kernel(const float* __restrict__ pos, const int* __restrict__ src,
                                                            const int* __restrict__ dst,
                                                            const float* __restrict__ offsets, int E, int lmax,
                                                            float* __restrict__ vec, float* __restrict__ len,
                                                            float* __restrict__ sh) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = src[e], d = dst[e];
  float vx = pos[3 * s] - pos[3 * d], vy = pos[3 * s + 1] - pos[3 * d + 1], vz = pos[3 * s + 2] - pos[3 * d + 2];
  if (offsets) vx += offsets[3 * e], vy += offsets[3 * e + 1], vz += offsets[3 * e + 2];
  vec[3 * e] = vx, vec[3 * e + 1] = vy, vec[3 * e + 2] = vz;
  float o[16];
  len[e] = geom_sh<float>(vx, vy, vz, lmax, o);
  const int S = (lmax + 1) * (lmax + 1);
  float* out = sh + (long)e * S;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < S) out[i] = o[i];
}

// Spherical harmonics of per-node vectors scaled by |v| * norm_scale, rows with keep[n] == 0 zeroed: the force encoding
// of the DeNS model [ref: nets/equiformer_md17_dens.py:276-289].  Input data, not differentiated.
__global__ __launch_bounds__(256) void vec_sh_kernel(const float* __restrict__ vec, const unsigned char* __restrict__ keep,
                                                     int N, int lmax, float norm_scale, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float o[16];
  const float len = geom_sh<float>(vec[3 * n], vec[3 * n + 1], vec[3 * n + 2], lmax, o);
  const float f = (keep && !keep[n]) ? 0.f : len * norm_scale;
  const int S = (lmax + 1) * (lmax + 1);
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < S) out[(long)n * S + i] = o[i] * f;
}

__global__ __launch_bounds__(256) void edge_geom_bwd_kernel(const float* __restrict__ vec, const float* __restrict__ d_sh,
                                                            const float* __restrict__ d_len, int E, int lmax,
                                                            float* __restrict__ d_vec) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int S = (lmax + 1) * (lmax + 1);
  float ox, oy, oz;
  geom_grad<float>(vec[3 * e], vec[3 * e + 1], vec[3 * e + 2], lmax, d_sh ? d_sh + (long)e * S : nullptr, d_len != nullptr,
                   d_len ? d_len[e] : 0.f, ox, oy, oz);
  d_vec[3 * e] = ox, d_vec[3 * e + 1] = oy, d_vec[3 * e + 2] = oz;
}

// ---------------------------------------------------------------------------------------------- radial basis
constexpr float kGaussA = 2.5066272160016134f;  // sqrt(2 * 3.14159), the reference's truncated pi

__global__ __launch_bounds__(256) void rbf_gauss_fwd_kernel(const float* __restrict__ len, long total, int R,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ stdp,
                                                            const float* __restrict__ weight,
                                                            const float* __restrict__ bias, float inv_cut,
                                                            float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long e = idx / R;
  const int r = (int)(idx - e * R);
  const float x = weight[0] * (len[e] * inv_cut) + bias[0];
  const float sd = fabsf(stdp[r]) + 1e-5f;
  const float t = (x - mean[r]) / sd;
  out[idx] = __expf(-0.5f * t * t) / (kGaussA * sd);
}

// thread r reduces CH edges for d_mean[r], d_std[r]; d_weight / d_bias reduced over the block
__global__ __launch_bounds__(256) void rbf_gauss_bwd_kernel(const float* __restrict__ len, const float* __restrict__ g,
                                                            int E, int R, const float* __restrict__ mean,
                                                            const float* __restrict__ stdp,
                                                            const float* __restrict__ weight,
                                                            const float* __restrict__ bias, float inv_cut,
                                                            float* __restrict__ d_mean, float* __restrict__ d_std,
                                                            float* __restrict__ d_weight, float* __restrict__ d_bias,
                                                            int CH) {
  const int r = threadIdx.x;
  const int e0 = blockIdx.x * CH, e1 = min(E, e0 + CH);
  float am = 0.f, as = 0.f, aw = 0.f, ab = 0.f;
  if (r < R) {
    const float w = weight[0], b = bias[0], mu = mean[r], sp = stdp[r];
    const float sd = fabsf(sp) + 1e-5f, sgn = (sp >= 0.f) ? 1.f : -1.f;
    for (int e = e0; e < e1; ++e) {
      const float xs = len[e] * inv_cut;
      const float t = (w * xs + b - mu) / sd;
      const float o = __expf(-0.5f * t * t) / (kGaussA * sd);
      const float go = g[(long)e * R + r] * o;
      const float dxv = -go * t / sd;
      am -= dxv;
      as += go * (t * t - 1.f) / sd * sgn;
      aw += dxv * xs;
      ab += dxv;
    }
    atomicAdd(d_mean + r, am);
    atomicAdd(d_std + r, as);
  }
  __shared__ float rw[4], rb[4];
  aw = wave_sum(aw), ab = wave_sum(ab);
  if ((threadIdx.x & 63) == 0) rw[threadIdx.x >> 6] = aw, rb[threadIdx.x >> 6] = ab;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, c = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) a += rw[i], c += rb[i];
    atomicAdd(d_weight, a);
    atomicAdd(d_bias, c);
  }
}

// one wavefront per edge: d_len[e] = sum_r g * d out / d len
__global__ __launch_bounds__(256) void rbf_gauss_dlen_kernel(const float* __restrict__ len, const float* __restrict__ g,
                                                             int E, int R, const float* __restrict__ mean,
                                                             const float* __restrict__ stdp,
                                                             const float* __restrict__ weight,
                                                             const float* __restrict__ bias, float inv_cut,
                                                             float* __restrict__ d_len) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= E) return;
  const float w = weight[0], x = w * (len[e] * inv_cut) + bias[0];
  float acc = 0.f;
  for (int r = lane; r < R; r += 64) {
    const float sd = fabsf(stdp[r]) + 1e-5f;
    const float t = (x - mean[r]) / sd;
    const float o = __expf(-0.5f * t * t) / (kGaussA * sd);
    acc += g[(long)e * R + r] * (-o * t / sd);
  }
  acc = wave_sum(acc);
  if (lane == 0) d_len[e] = acc * w * inv_cut;
}

__global__ __launch_bounds__(256) void rbf_expnorm_fwd_kernel(const float* __restrict__ len, long total, int R,
                                                              const float* __restrict__ means,
                                                              const float* __restrict__ betas, float alpha, float rc,
                                                              float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long e = idx / R;
  const int r = (int)(idx - e * R);
  const float d = len[e];
  const float cut = (d < rc) ? 0.5f * (cosf(d * 3.14159265358979323846f / rc) + 1.f) : 0.f;
  const float q = expf(-alpha * d) - means[r];
  out[idx] = cut * expf(-betas[r] * q * q);
}

__global__ __launch_bounds__(256) void rbf_expnorm_dlen_kernel(const float* __restrict__ len, const float* __restrict__ g,
                                                               int E, int R, const float* __restrict__ means,
                                                               const float* __restrict__ betas, float alpha, float rc,
                                                               float* __restrict__ d_len) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= E) return;
  const float d = len[e];
  const float pi = 3.14159265358979323846f;
  const bool in = d < rc;
  const float cut = in ? 0.5f * (cosf(d * pi / rc) + 1.f) : 0.f;
  const float dcut = in ? -0.5f * sinf(d * pi / rc) * pi / rc : 0.f;
  const float ex = expf(-alpha * d);
  float acc = 0.f;
  for (int r = lane; r < R; r += 64) {
    const float q = ex - means[r];
    const float En = expf(-betas[r] * q * q);
    acc += g[(long)e * R + r] * (dcut * En + cut * En * (-2.f * betas[r] * q) * (-alpha * ex));
  }
  acc = wave_sum(acc);
  if (lane == 0) d_len[e] = acc;
}


// ---- spherical Bessel basis with polynomial envelope (GemNet RadialBasis of ocpmodels 0.0.3, un-vendored dependency of
// the reference: ocpmodels/models/gemnet/layers/radial_basis.py; call sites nets/graph_attention_transformer.py:26,
// 786-788).  x = len / rc;  out[e,k] = env(x) sqrt(2 / rc^3) sin(f_k x) / x,  env = 1 + a x^5 + b x^6 + c x^7 for x < 1
// (p = 5: a = -21, b = 35, c = -15), frequencies f_k trainable (initialised k pi).
struct BesselEnv {
  float e0, e1, e2;  // envelope value and its first two derivatives wrt x
};
__device__ __forceinline__ BesselEnv bessel_env(float x) {
  BesselEnv r{0.f, 0.f, 0.f};
  if (x < 1.f) {
    const float x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
    r.e0 = 1.f + x5 * (-21.f + x * (35.f - 15.f * x));
    r.e1 = x4 * (-105.f + x * (210.f - 105.f * x));
    r.e2 = x3 * (-420.f + x * (1050.f - 630.f * x));
  }
  return r;
}

__global__ __launch_bounds__(256) void rbf_bessel_fwd_kernel(const float* __restrict__ len, long total, int R,
                                                             const float* __restrict__ freq, float inv_rc, float nc,
                                                             float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long e = idx / R;
  const int r = (int)(idx - e * R);
  const float x = len[e] * inv_rc;
  const BesselEnv ev = bessel_env(x);
  out[idx] = ev.e0 * nc * sinf(freq[r] * x) / x;
}

// one wave per edge: d_len[e]; d_freq accumulated per block through LDS
__global__ __launch_bounds__(256) void rbf_bessel_bwd_kernel(const float* __restrict__ len, const float* __restrict__ g,
                                                             int E, int R, const float* __restrict__ freq, float inv_rc,
                                                             float nc, float* __restrict__ d_len,
                                                             float* __restrict__ d_freq, int EPB) {
  extern __shared__ float red[];  // [R]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = threadIdx.x; r < R; r += blockDim.x) red[r] = 0.f;
  __syncthreads();
  const int e0 = blockIdx.x * EPB, e1 = min(E, e0 + EPB);
  for (int e = e0 + wave; e < e1; e += 4) {
    const float x = len[e] * inv_rc;
    const BesselEnv ev = bessel_env(x);
    const float ix = 1.f / x;
    float acc = 0.f;
    for (int r = lane; r < R; r += 64) {
      const float f = freq[r], sn = sinf(f * x), cs = cosf(f * x);
      const float s0 = sn * ix, s1 = f * cs * ix - sn * ix * ix;
      const float gv = g[(long)e * R + r];
      acc += gv * nc * (ev.e1 * s0 + ev.e0 * s1);
      if (d_freq) atomicAdd(&red[r], gv * nc * ev.e0 * cs);  // d/df [sin(f x) / x] = cos(f x)
    }
    acc = wave_sum(acc);
    if (lane == 0 && d_len) d_len[e] = acc * inv_rc;
  }
  __syncthreads();
  if (d_freq)
    for (int r = threadIdx.x; r < R; r += blockDim.x) atomicAdd(d_freq + r, red[r]);
}

}  // namespace

extern "C" {

int eqf_radius_graph_count(const float* pos, const int* mol_ptr, int n_mol, float r, int max_nbr, int* deg,
                           void* stream) {
  if (!pos || !mol_ptr || !deg || max_nbr < 1) return EQF_E_BADARG;
  if (n_mol <= 0) return 0;
  hipLaunchKernelGGL(radius_count_kernel, dim3(n_mol), dim3(64), 0, (hipStream_t)stream, pos, mol_ptr, r * r, max_nbr,
                     deg);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_radius_graph_fill(const float* pos, const int* mol_ptr, int n_mol, float r, int max_nbr, const int* row_ptr,
                          int* src, int* dst, void* stream) {
  if (!pos || !mol_ptr || !row_ptr || !src || !dst || max_nbr < 1) return EQF_E_BADARG;
  if (n_mol <= 0) return 0;
  hipLaunchKernelGGL(radius_fill_kernel, dim3(n_mol), dim3(64), 0, (hipStream_t)stream, pos, mol_ptr, r * r, max_nbr,
                     row_ptr, src, dst);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_radius_graph_pbc_count(const float* pos, const float* cell, const int* mol_ptr, int n_mol, float r, int max_nbr,
                               int* cand, int* deg, void* stream) {
  if (!pos || !cell || !mol_ptr || !cand || !deg || max_nbr < 1 || !(r > 0.f)) return EQF_E_BADARG;
  if (n_mol <= 0) return 0;
  hipLaunchKernelGGL(pbc_count_kernel, dim3(n_mol), dim3(64), 0, (hipStream_t)stream, pos, cell, mol_ptr, r, max_nbr, cand,
                     deg);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_radius_graph_pbc_fill(const float* pos, const float* cell, const int* mol_ptr, int n_mol, float r, int max_nbr,
                              const int* row_ptr, const int* cand_ptr, float* scratch_d2, int* src, int* dst,
                              int* cell_offsets, float* offsets, void* stream) {
  if (!pos || !cell || !mol_ptr || !row_ptr || !cand_ptr || !scratch_d2 || !src || !dst || !offsets || max_nbr < 1)
    return EQF_E_BADARG;
  if (n_mol <= 0) return 0;
  hipLaunchKernelGGL(pbc_fill_kernel, dim3(n_mol), dim3(64), 0, (hipStream_t)stream, pos, cell, mol_ptr, r, max_nbr,
                     row_ptr, cand_ptr, scratch_d2, src, dst, cell_offsets, offsets);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_segment_ptr(const int* seg_of, int n, int n_seg, int* ptr, int* max_len, void* stream) {
  if (!ptr || (n > 0 && !seg_of) || n < 0 || n_seg < 0) return EQF_E_BADARG;
  if (max_len) {
    hipError_t e = hipMemsetAsync(max_len, 0, sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(segment_ptr_kernel, dim3(eqf_cdiv(n_seg + 1, 256)), dim3(256), 0, (hipStream_t)stream, seg_of, n,
                     n_seg, ptr, max_len);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_exclusive_scan_i32(const int* counts, int n, int* ptr, int* total, void* stream) {
  if (!ptr || (n > 0 && !counts) || n < 0) return EQF_E_BADARG;
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, n, ptr, total);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_csr_by_source(const int* src, const int* row_ptr, const int* mol_ptr, int n_mol, int max_mol_nodes,
                      int* src_perm, int* src_ptr, void* stream) {
  if (!row_ptr || !mol_ptr || !src_perm || !src_ptr || !src) return EQF_E_BADARG;
  if (max_mol_nodes > CSR_MAX_NODES) return EQF_E_UNSUPPORTED;
  if (n_mol <= 0) return 0;
  static bool attr_done[64] = {};  // per device (function attribute)
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr = attr_done[dev_id & 63];
  if (!attr) {
    hipFuncSetAttribute((const void*)csr_by_source_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                        CSR_MAX_NODES * (int)sizeof(int));
    attr = true;
  }
  const size_t lds = sizeof(int) * (size_t)(max_mol_nodes > 0 ? max_mol_nodes : 1);
  hipLaunchKernelGGL(csr_by_source_kernel, dim3(n_mol), dim3(256), lds, (hipStream_t)stream, src, row_ptr, mol_ptr, n_mol,
                     src_perm, src_ptr);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_edge_geom_fwd(const float* pos, const int* src, const int* dst, const float* offsets, int E, int lmax,
                      float* vec, float* len, float* sh, void* stream) {
  if (!pos || !src || !dst || !vec || !len || !sh) return EQF_E_BADARG;
  if (lmax < 0 || lmax > 3) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  hipLaunchKernelGGL(edge_geom_fwd_kernel, dim3(eqf_cdiv(E, 256)), dim3(256), 0, (hipStream_t)stream, pos, src, dst,
                     offsets, E, lmax, vec, len, sh);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_vec_sh(const float* vec, const unsigned char* keep, int N, int lmax, float norm_scale, float* out, void* stream) {
  if (!vec || !out) return EQF_E_BADARG;
  if (lmax < 0 || lmax > 3) return EQF_E_UNSUPPORTED;
  if (N <= 0) return 0;
  hipLaunchKernelGGL(vec_sh_kernel, dim3(eqf_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, vec, keep, N, lmax,
                     norm_scale, out);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_edge_geom_bwd(const float* vec, const float* d_sh, const float* d_len, int E, int lmax, float* d_vec,
                      void* stream) {
  if (!vec || !d_vec) return EQF_E_BADARG;
  if (lmax < 0 || lmax > 3) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  hipLaunchKernelGGL(edge_geom_bwd_kernel, dim3(eqf_cdiv(E, 256)), dim3(256), 0, (hipStream_t)stream, vec, d_sh, d_len,
                     E, lmax, d_vec);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_rbf_gaussian_fwd(const float* len, int E, int R, const float* mean, const float* std, const float* weight,
                         const float* bias, float cutoff, float* out, void* stream) {
  if (!len || !mean || !std || !weight || !bias || !out || R < 1) return EQF_E_BADARG;
  if (E <= 0) return 0;
  const long total = (long)E * R;
  hipLaunchKernelGGL(rbf_gauss_fwd_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, len, total, R,
                     mean, std, weight, bias, 1.f / cutoff, out);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_rbf_gaussian_bwd(const float* len, const float* d_out, int E, int R, const float* mean, const float* std,
                         const float* weight, const float* bias, float cutoff, float* d_mean, float* d_std,
                         float* d_weight, float* d_bias, float* d_len, void* stream) {
  if (!len || !d_out || !mean || !std || !weight || !bias || !d_mean || !d_std || !d_weight || !d_bias)
    return EQF_E_BADARG;
  if (R > 256) return EQF_E_UNSUPPORTED;
  if (E <= 0) return 0;
  const int CH = 64;
  const int threads = ((R + 63) / 64) * 64;
  hipLaunchKernelGGL(rbf_gauss_bwd_kernel, dim3(eqf_cdiv(E, CH)), dim3(threads), 0, (hipStream_t)stream, len, d_out, E,
                     R, mean, std, weight, bias, 1.f / cutoff, d_mean, d_std, d_weight, d_bias, CH);
  EQF_CHECK_LAUNCH();
  if (d_len) {
    hipLaunchKernelGGL(rbf_gauss_dlen_kernel, dim3(eqf_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, len, d_out, E, R,
                       mean, std, weight, bias, 1.f / cutoff, d_len);
    EQF_CHECK_LAUNCH();
  }
  return 0;
}

int eqf_rbf_expnorm_fwd(const float* len, int E, int R, const float* means, const float* betas, float alpha,
                        float cutoff, float* out, void* stream) {
  if (!len || !means || !betas || !out || R < 1) return EQF_E_BADARG;
  if (E <= 0) return 0;
  const long total = (long)E * R;
  hipLaunchKernelGGL(rbf_expnorm_fwd_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, len, total,
                     R, means, betas, alpha, cutoff, out);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_rbf_expnorm_bwd(const float* len, const float* d_out, int E, int R, const float* means, const float* betas,
                        float alpha, float cutoff, float* d_len, void* stream) {
  if (!len || !d_out || !means || !betas || !d_len) return EQF_E_BADARG;
  if (E <= 0) return 0;
  hipLaunchKernelGGL(rbf_expnorm_dlen_kernel, dim3(eqf_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, len, d_out, E, R,
                     means, betas, alpha, cutoff, d_len);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_rbf_bessel_fwd(const float* len, int E, int R, const float* freq, float cutoff, float* out, void* stream) {
  if (!len || !freq || !out || cutoff <= 0.f) return EQF_E_BADARG;
  if (E <= 0) return 0;
  const long total = (long)E * R;
  hipLaunchKernelGGL(rbf_bessel_fwd_kernel, dim3(eqf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, len, total, R,
                     freq, 1.f / cutoff, sqrtf(2.f / (cutoff * cutoff * cutoff)), out);
  EQF_CHECK_LAUNCH();
  return 0;
}

int eqf_rbf_bessel_bwd(const float* len, const float* d_out, int E, int R, const float* freq, float cutoff,
                       float* d_freq, float* d_len, void* stream) {
  if (!len || !d_out || !freq || cutoff <= 0.f) return EQF_E_BADARG;
  if (E <= 0) return 0;
  const int EPB = 64;
  hipLaunchKernelGGL(rbf_bessel_bwd_kernel, dim3(eqf_cdiv(E, EPB)), dim3(256), sizeof(float) * R, (hipStream_t)stream, len,
                     d_out, E, R, freq, 1.f / cutoff, sqrtf(2.f / (cutoff * cutoff * cutoff)), d_len, d_freq, EPB);
  EQF_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
