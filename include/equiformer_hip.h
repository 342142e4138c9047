/*
 * equiformer_hip.h -- C ABI of libequiformer_hip.so: the MI355X (gfx950) hot path of Equiformer
 * (atomicarchitects/equiformer), forward and backward.
 *
 * The reference has no native boundary of its own: its hot path is Python that dispatches into
 * e3nn / torch_scatter / torch_cluster / PyG kernels.  Each entry point below replaces one such
 * dispatch; the reference call site it stands in for is cited as  [ref: file:line]  (paths relative
 * to the reference repository root).  The Python host package `equiformer_amd.nets` (same registry,
 * module tree and parameter names as the reference's `nets`) binds these symbols with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (hipMalloc'ed / torch CUDA tensors' data_ptr()) unless marked
 *    "host".  fp32 data, int32 indices.  Nothing here allocates, synchronises or touches the host
 *    copy of any buffer; every launch goes to `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream).  All entry points are safe under HIP-graph capture.
 *  - Return value: 0 on success, otherwise the hipError_t of the failed launch (as int), or a
 *    negative EQF_E_* code for argument errors.  Nothing is written on argument errors.
 *  - Feature tensors are row-major [rows, D].  Inside a row the channels of an irreps list
 *    mul_0 x l_0 + mul_1 x l_1 + ... are stored segment after segment, and inside the segment of
 *    degree l as [2l+1][mul]  (component m slow, channel u fast).  This "channel-fastest" (CF)
 *    layout is the library's internal layout (e3nn uses [mul][2l+1]); for 0e-only tensors the two
 *    coincide.  `eqf_irreps` describes such a row.
 *  - Edges are sorted by destination node: `row_ptr[N+1]` (CSR over dst) and `src[E]`, `dst[E]`.
 */
#ifndef EQUIFORMER_HIP_H
#define EQUIFORMER_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define EQF_MAX_SEG 8
#define EQF_MAX_PATHS 72

#define EQF_E_BADARG (-1)
#define EQF_E_UNSUPPORTED (-2)

/* One row of a feature tensor: nseg segments, segment s has degree l[s] and multiplicity mul[s]. */
typedef struct eqf_irreps {
  int nseg;
  int l[EQF_MAX_SEG];
  int mul[EQF_MAX_SEG];
  int odd[EQF_MAX_SEG]; /* 1 = odd parity (E(3) models); 0 = even.  Only the 0e segments are "scalars" (mean, bias) */
} eqf_irreps;

/* Depth-wise tensor product ('uvu', mul2 == 1) path table.  Path p couples input segment of degree
 * l1[p] (row offset in_off[p], multiplicity mul[p]) with the spherical harmonic of degree l2[p] into
 * output degree l3[p]; its mul[p] output channels start at channel out_ch[p] of the output segment
 * (row offset out_off[p], total channels out_k[p]); its per-edge weights start at w_off[p].
 * cg_off[p] indexes the dense table  cg[cg_off + (i*(2*l2+1) + j)*(2*l3+1) + k]  which already
 * includes the sqrt(2*l3+1) path normalisation.                     [ref: e3nn o3.TensorProduct codegen] */
typedef struct eqf_dtp_paths {
  int npaths;
  int sh_dim;      /* (lmax_sh+1)^2: row length of the spherical-harmonics tensor */
  int in_dim;      /* D of the input rows */
  int out_dim;     /* D of the output rows */
  int w_numel;     /* weights per edge */
  int m_numel;     /* sum_p (2*l1+1)*(2*l3+1): size of the per-edge coupling matrices */
  int l1[EQF_MAX_PATHS], l2[EQF_MAX_PATHS], l3[EQF_MAX_PATHS], mul[EQF_MAX_PATHS];
  int in_off[EQF_MAX_PATHS], out_off[EQF_MAX_PATHS], out_ch[EQF_MAX_PATHS], out_k[EQF_MAX_PATHS];
  int w_off[EQF_MAX_PATHS], cg_off[EQF_MAX_PATHS], m_off[EQF_MAX_PATHS];
} eqf_dtp_paths;

/* Library identification: returns a static string "equiformer_hip <version> gfx950". */
const char* eqf_version(void);

/* ---------------------------------------------------------------------------------------------
 * Graph construction and edge geometry
 * ------------------------------------------------------------------------------------------- */

/* Degree of every node in a radius graph: deg[i] = min(max_nbr, #{j != i in the same molecule :
 * |pos_j - pos_i| < r}).  mol_ptr[n_mol+1] are node offsets of the molecules (nodes of a molecule
 * are contiguous, as in a PyG Batch).
 * [ref: torch_cluster.radius_graph call, nets/graph_attention_transformer.py:866-867] */
int eqf_radius_graph_count(const float* pos, const int* mol_ptr, int n_mol, float r, int max_nbr,
                           int* deg, void* stream);

/* Fill the dst-sorted edge list given row_ptr = exclusive scan of deg: for node i, sources in
 * ascending index order (the order torch_cluster emits).  src/dst are [E].                       */
int eqf_radius_graph_fill(const float* pos, const int* mol_ptr, int n_mol, float r, int max_nbr,
                          const int* row_ptr, int* src, int* dst, void* stream);

/* Periodic radius graph (OC20): for every centre i the neighbours (j, image n) of the same structure with
 * 1e-4 < |pos_j + n.cell - pos_i|^2 <= r^2, images |n_k| <= ceil(r * |a_l x a_m| / |det cell|); at most max_nbr
 * nearest per centre (ties in candidate order (j, n1, n2, n3)).  cell[n_mol,3,3] rows = lattice vectors.
 * count: cand[N] = candidates per centre, deg[N] = min(cand, max_nbr).  fill: row_ptr / cand_ptr = exclusive scans
 * of deg / cand, scratch_d2[sum cand] floats; writes the dst-sorted edge list src/dst[E], the integer images
 * cell_offsets[E,3] (may be NULL) and their Cartesian offsets[E,3] (edge_vec = pos[src] - pos[dst] + offsets).
 * [ref: ocpmodels radius_graph_pbc + get_pbc_distances as called at nets/graph_attention_transformer_oc20.py:267-293;
 *  un-vendored dependency, algorithm restated in oracle/pbc.py] */
int eqf_radius_graph_pbc_count(const float* pos, const float* cell, const int* mol_ptr, int n_mol, float r,
                               int max_nbr, int* cand, int* deg, void* stream);
int eqf_radius_graph_pbc_fill(const float* pos, const float* cell, const int* mol_ptr, int n_mol, float r,
                              int max_nbr, const int* row_ptr, const int* cand_ptr, float* scratch_d2, int* src,
                              int* dst, int* cell_offsets, float* offsets, void* stream);

/* CSR bookkeeping of the graph (replaces torch.bincount / cumsum / argsort around the radius graph; the reference
 * gets the same quantities from torch_cluster / torch_scatter internals and `degree`,
 * nets/graph_attention_transformer.py:866-867,517).
 * eqf_segment_ptr: seg_of[n] ascending (PyG `batch`) -> ptr[n_seg+1] with ptr[g] = first i with seg_of[i] >= g;
 *   max_len (may be NULL) receives the longest segment.
 * eqf_exclusive_scan_i32: ptr[0..n] = exclusive prefix sums of counts[n]; total (may be NULL) receives ptr[n].
 * eqf_csr_by_source: for a dst-sorted edge list whose edges never cross the molecules of mol_ptr and whose rows
 *   hold every source at most once: src_perm[E] = stable argsort of src, src_ptr[N+1] = offsets of the by-source
 *   groups.  max_mol_nodes = an upper bound of the nodes per molecule (<= 16384, else EQF_E_UNSUPPORTED).          */
int eqf_segment_ptr(const int* seg_of, int n, int n_seg, int* ptr, int* max_len, void* stream);
int eqf_exclusive_scan_i32(const int* counts, int n, int* ptr, int* total, void* stream);
int eqf_csr_by_source(const int* src, const int* row_ptr, const int* mol_ptr, int n_mol, int max_mol_nodes,
                      int* src_perm, int* src_ptr, void* stream);

/* edge_vec = pos[src] - pos[dst] (+ offsets, may be NULL), len = |edge_vec|,
 * sh = Y^0..Y^lmax(edge_vec/len) * sqrt(2l+1)  ("component" normalisation), lmax <= 3.
 * [ref: nets/graph_attention_transformer.py:868-870,874;  e3nn o3.spherical_harmonics] */
int eqf_edge_geom_fwd(const float* pos, const int* src, const int* dst, const float* offsets, int E,
                      int lmax, float* vec, float* len, float* sh, void* stream);

/* out[n, :] = keep[n] ? |v_n| * norm_scale * Y(v_n / |v_n|) : 0   (component-normalised real spherical harmonics
 * up to lmax <= 3, [N, (lmax+1)^2]; keep may be NULL = all kept).  The force encoding of the DeNS model (input data,
 * no gradient).  [ref: nets/equiformer_md17_dens.py:276-289] */
int eqf_vec_sh(const float* vec, const unsigned char* keep, int N, int lmax, float norm_scale, float* out,
               void* stream);
/* d_vec[E,3] from d_sh[E,(lmax+1)^2] (may be NULL) and d_len[E] (may be NULL). */
int eqf_edge_geom_bwd(const float* vec, const float* d_sh, const float* d_len, int E, int lmax,
                      float* d_vec, void* stream);

/* Graphormer Gaussian radial basis, out[e,r] = exp(-.5 ((w*len/cutoff+b-mean_r)/std_r)^2)/(sqrt(2*3.14159) std_r),
 * std_r = |std_r| + 1e-5.                                  [ref: nets/gaussian_rbf.py:5-9,32-40] */
int eqf_rbf_gaussian_fwd(const float* len, int E, int R, const float* mean, const float* std,
                         const float* weight, const float* bias, float cutoff, float* out, void* stream);
/* d_mean[R], d_std[R], d_weight[1], d_bias[1] are ACCUMULATED (+=); d_len[E] (may be NULL) is written. */
int eqf_rbf_gaussian_bwd(const float* len, const float* d_out, int E, int R, const float* mean,
                         const float* std, const float* weight, const float* bias, float cutoff,
                         float* d_mean, float* d_std, float* d_weight, float* d_bias, float* d_len,
                         void* stream);

/* Exp-normal smearing with cosine cutoff, out[e,r] = .5(cos(pi len/rc)+1)[len<rc] exp(-beta_r (exp(-alpha len)-mu_r)^2).
 * [ref: nets/graph_attention_transformer_md17.py:51-81,119-124] */
int eqf_rbf_expnorm_fwd(const float* len, int E, int R, const float* means, const float* betas,
                        float alpha, float cutoff, float* out, void* stream);
int eqf_rbf_expnorm_bwd(const float* len, const float* d_out, int E, int R, const float* means,
                        const float* betas, float alpha, float cutoff, float* d_len, void* stream);

/* Spherical Bessel basis with the polynomial envelope (exponent 5): x = len / rc,
 * out[e,k] = env(x) sqrt(2 / rc^3) sin(freq_k x) / x, env = 1 - 21 x^5 + 35 x^6 - 15 x^7 for x < 1, else 0; freq trainable.
 * [ref: RadialBasis(rbf={'name': 'spherical_bessel'}) of ocpmodels 0.0.3 (gemnet/layers/radial_basis.py; un-vendored
 * dependency, restated), constructed at nets/graph_attention_transformer.py:786-788 and ..._md17.py:178-180]
 * bwd: d_freq[R] ACCUMULATED (may be NULL), d_len[E] written (may be NULL). */
int eqf_rbf_bessel_fwd(const float* len, int E, int R, const float* freq, float cutoff, float* out, void* stream);
int eqf_rbf_bessel_bwd(const float* len, const float* d_out, int E, int R, const float* freq, float cutoff,
                       float* d_freq, float* d_len, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense contractions on the matrix cores (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32)
 * ------------------------------------------------------------------------------------------- */

/* "Two-level" row addressing used for the l-segments of CF rows: logical row i of a matrix lives at
 *    base + (i / d) * ld + (i % d) * inner ,   its elements are contiguous from there.
 * A plain row-major matrix is d = 1, inner = 0, ld = leading dimension.                          */
typedef struct eqf_rows {
  int d;
  int ld;
  int inner;
} eqf_rows;

/* C[i,n] (=|+=) sum_k A[i,k] * B[k,n]  (+ bias[n]).   A: M rows (two-level) x K, B: plain [K,N]
 * with leading dimension ldb, C: M rows (two-level) x N.  accumulate != 0 adds into C.
 * [ref: LinearRS / FullyConnectedTensorProductRescale forward, nets/tensor_product_rescale.py:125-136,171-174;
 *       torch.nn.Linear inside RadialProfile, nets/radial_func.py:46-49] */
int eqf_gemm_nn(const float* A, eqf_rows ra, const float* B, int ldb, float* C, eqf_rows rc,
                const float* bias, int M, int N, int K, int accumulate, void* stream);
/* C[i,n] (=|+=) sum_k A[i,k] * B[n,k] (+ bias[n]).   B: plain [N,K] with leading dimension ldb. */
int eqf_gemm_nt(const float* A, eqf_rows ra, const float* B, int ldb, float* C, eqf_rows rc,
                const float* bias, int M, int N, int K, int accumulate, void* stream);
/* C[m,n] += sum_i A[i,m] * B[i,n]   over R rows (two-level on both operands); C plain [M,N] with
 * leading dimension ldc.  C is always ACCUMULATED into (split over row chunks with fp32 atomics). */
int eqf_gemm_tn(const float* A, eqf_rows ra, const float* B, eqf_rows rb, float* C, int ldc, int M,
                int N, int R, void* stream);
/* eqf_gemm_tn that also accumulates the column sums of its operands while they are staged (the bias gradient of the
 * same linear layer, otherwise a separate eqf_colsum pass over dy): colsum_a[m] += sum_rows A[row, m] and
 * colsum_b[n] += sum_rows B[row, n]; either may be NULL.  In eqf_gemm_group a kind-2 descriptor's `bias` field is the
 * colsum_b accumulator.  [ref: the `.bias` gradients of nn.Linear (radial_func.py) and of LinearRS,
 * nets/tensor_product_rescale.py:93-110] */
int eqf_gemm_tn_colsum(const float* A, eqf_rows ra, const float* B, eqf_rows rb, float* C, int ldc, int M, int N,
                       int R, float* colsum_a, float* colsum_b, void* stream);

/* Several independent GEMMs in ONE launch per kind (the per-degree GEMMs of one irreps linear; the node-level linears
 * have only N ~ 2 k rows and are launch / latency bound; the radial MLPs of all blocks side by side).  n <= 8 problems.
 *   kind 0:  C[i,n] (=|+=) sum_k A[i,k] B[k,n] (+ bias)     A: M two-level rows (ra) x K, B plain [K,N] (ldb), C rows (rc)
 *   kind 1:  C[i,n] (=|+=) sum_k A[i,k] B[n,k] (+ bias)     B plain [N,K] (ldb)
 *   kind 2:  C[m,n] += sum_{i<K} A[i,m] B[i,n]              A: K two-level rows (ra) x M, B: K two-level rows (rc) x N,
 *                                                            C plain [M,N] with leading dimension ldb (atomics);
 *                                                            `bias` (optional) accumulates the column sums of B
 *   kind 3:  as kind 2, `bias` accumulates the column sums of A (nn.Linear orientation: dW[out,in] = dy^T x, db = colsum dy)
 * [ref: LinearRS / FullyConnectedTensorProductRescale, nets/tensor_product_rescale.py:125-136,171-174] */
typedef struct eqf_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  eqf_rows ra, rc;
  int ldb, M, N, K, accumulate, kind;
} eqf_gemm_desc;
int eqf_gemm_group(const eqf_gemm_desc* d, int n, void* stream);
/* The same grouped GEMMs with the matrix steps on the bf16 matrix cores (csrc/gemmx.hip): fp32 operands split into bf16
 * planes where they are staged -- mode 0 (split): activations 2 planes, weights 3, 5 products (weight gradients: 2 + 2 planes,
 * 3 products), fp32-class results; mode 1 (bf16): plain bf16 operands, fp32 accumulation (the arithmetic of the reference's
 * AMP linears, main_qm9.py:117-119, engine.py:58-66); mode 2: 3 + 3 planes.  Same descriptors, same results up to the
 * mode's rounding; up to 24 problems per call (eqf_gemm_group: 8).  [ref: as eqf_gemm_group] */
int eqf_gemmx_group(const eqf_gemm_desc* d, int n, int mode, void* stream);

/* out[n] += sum_rows x[row, n]  over a two-level-row matrix (bias gradients). */
int eqf_colsum(const float* X, eqf_rows rx, int R, int N, float* out, void* stream);

/* Per-edge coupling matrices of a DTP path table (depend on geometry only, shared by every DTP that uses
 * the same path table -- all blocks of a model):
 *   coupling[e, m_off[p] + i*(2*l3+1) + k] = sum_j cg_p[i,j,k] * sh[e, l2^2 + j]
 * so that  DTP(x, sh, w)[e, p, k, u] = w[e,p,u] * sum_i coupling[e,p,i,k] * x[e, l1, i, u].
 * [ref: o3.TensorProduct('uvu') einsum 'zuv,ijk,zuvij->zuk', nets/tensor_product_rescale.py:33-37] */
int eqf_dtp_coupling_fwd(const float* sh, const float* cg, const eqf_dtp_paths* paths, float* coupling,
                         int E, void* stream);
/* d_sh[E, sh_dim] written from d_coupling[E, m_numel]. */
int eqf_dtp_coupling_bwd(const float* d_coupling, const float* cg, const eqf_dtp_paths* paths,
                         float* d_sh, int E, void* stream);

/* Fused depth-wise tensor product -> per-degree linear:  the A operand of the GEMM is generated on the
 * fly from (x, coupling, w) by the DTP path table and never written to memory.
 *   out[e, seg(l3)] = DTP(x, sh, w)[e, seg(l3)] . Wl[l3]    (+ bias0 on l3 = 0)
 * for every output degree present in `paths`.  w may be NULL (all path weights 1).
 * Wl[l3] are plain [out_k(l3), N(l3)] row-major matrices given as device pointers in a HOST array
 * indexed by l3 (entries for absent degrees ignored); out_irreps describes the output rows (one
 * segment per degree).  Requires every path multiplicity to be a multiple of 32.
 * [ref: SeparableFCTP.forward, nets/graph_attention_transformer.py:234-248] */
int eqf_dtp_linear_fwd(const float* x, const float* coupling, const float* w,
                       const eqf_dtp_paths* paths, const float* const* Wl, const float* bias0,
                       float* out, const eqf_irreps* out_irreps, int E, void* stream);
/* dWl[l3][k, n] += sum_{e,m} DTP(x,sh,w)[e,l3,m,k] * d_out[e,l3,m,n]  (A operand regenerated). */
int eqf_dtp_linear_wgrad(const float* x, const float* coupling, const float* w,
                         const eqf_dtp_paths* paths, const float* d_out,
                         const eqf_irreps* out_irreps, float* const* dWl, int E, void* stream);

/* Fused SeparableFCTP (the edge hot loop): depth-wise tensor product -> per-degree linear for ALL output degrees and
 * ALL consumers of the DTP output in one launch; the DTP result is never written to memory in either direction.
 *   out1[e, seg(l3)]  = DTP(x, sh, w)[e, seg(l3)] . Wl[l3]          (+ bias0 on degree 0)  for every degree of out1_irreps
 *   out2[e, 0:n2]     = DTP(x, sh, w)[e, seg(0)]  . W2  (+ bias2)    optional second scalar consumer (attention logits)
 * Wl[l3]: device pointers (HOST array indexed by l3) to row-major [K(l3), N1(l3)] matrices, K(l3) = channels of the DTP
 * output of degree l3 -- e3nn's flat `tp.weight` of the LinearRS holds exactly these blocks in ascending degree, so
 * the pointers go straight into the parameter; W2 is [K(0), n2].  out2 == NULL <=> n2 == 0 <=> W2 == NULL.  w may be
 * NULL (unit path weights); bias0 [N1(0)] / bias2 [n2] may be NULL.  Requires every path multiplicity, every N1(l3) and
 * n2 to be multiples of 32.
 * [ref: SeparableFCTP.forward nets/graph_attention_transformer.py:234-248 (dtp + lin) together with sep_alpha :492;
 *       EdgeDegreeEmbeddingNetwork.forward :725-733 (dw + proj)] */
int eqf_sfc_fwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                const float* const* Wl, const float* bias0, const float* W2, const float* bias2, float* out1,
                const eqf_irreps* out1_irreps, float* out2, int n2, int E, void* stream);
/* Data gradient of eqf_sfc_fwd: dx[E,in_dim] written; dw[E,w_numel] written if non-NULL (ignored when w == NULL);
 * d_coupling[E,m_numel] (may be NULL; needed only when forces are differentiated) is ACCUMULATED with atomics. */
int eqf_sfc_bwd_data(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                     const float* const* Wl, const float* W2, const float* d_out1, const eqf_irreps* out1_irreps,
                     const float* d_out2, int n2, float* dx, float* dw, float* d_coupling, int E, void* stream);
/* Weight gradient of eqf_sfc_fwd: dWl[l3] / dW2 (same shapes as Wl[l3] / W2) ACCUMULATED (fp32 atomics). */
int eqf_sfc_bwd_weight(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                       const float* d_out1, const eqf_irreps* out1_irreps, const float* d_out2, int n2,
                       float* const* dWl, float* dW2, int E, void* stream);

/* The same fused SeparableFCTP with every matrix step on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32
 * accumulation).  fp32 operands are split into bf16 planes (value = plane1 + plane2 (+ plane3)); `mode` selects how many:
 *   0  activations 2 planes, weights 3 planes, 5 plane products (two activation operands: 3)  -- fp32-class results
 *      (model-level error measured by tools/split_model_error.py: 2e-6 on energies, 1e-5 on gradients vs fp64)
 *   1  one plane each: plain bf16 operands = the arithmetic of torch.autocast(bfloat16) that the reference's drivers
 *      enable by default (main_qm9.py:117-119,197-201); BASELINE config #2
 *   2  3 + 3 planes, 6 products (3e-7), for cross-checks
 * The weight planes are produced once per weight value by eqf_sfcx_pack into `packed` (eqf_sfcx_packed_numel bf16
 * elements, 16-byte aligned), in MFMA fragment order for the forward and for the data gradient; the other arguments
 * are those of eqf_sfc_fwd / _bwd_data / _bwd_weight.  Limits: per-edge tensors < 2^31 elements, degrees <= 3, row
 * strides that are multiples of four floats (EQF_E_UNSUPPORTED otherwise: use the eqf_sfc_* entry point of the same shape).
 * Which kernel serves a call is the library's choice by shape and edge count (csrc/sfcy.hip / sfcw.hip: multi-wave forward and
 * weight gradient of L_max <= 2 operators from 1 500-4 000 / 9 000-20 000 edges on; csrc/sfcx.hip: one-wave kernels otherwise, with items split over
 * several waves on graphs too small to fill the machine); results differ only in fp32 summation order between them.
 * [ref: as eqf_sfc_fwd; the dtype policy replaces torch.cuda.amp.autocast of engine.py:58-66] */
long eqf_sfcx_packed_numel(const eqf_dtp_paths* paths, const eqf_irreps* out1_irreps, int n2, int mode);
/* host only, no launch: bit mask of the split-precision launches that can serve this operator (1 forward, 2 data gradient,
 * 4 weight gradient; the planners' own verdict on their table limits) -- use the eqf_sfc_* entry point for a cleared bit */
int eqf_sfcx_supported(const eqf_dtp_paths* paths, const eqf_irreps* out1_irreps, int n2, int mode);
int eqf_sfcx_pack(const float* const* Wl, const float* W2, const eqf_dtp_paths* paths, const eqf_irreps* out1_irreps,
                  int n2, int mode, void* packed, void* stream);
int eqf_sfcx_fwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths, const void* packed,
                 const float* bias0, const float* bias2, float* out1, const eqf_irreps* out1_irreps, float* out2, int n2,
                 int E, int mode, void* stream);
int eqf_sfcx_bwd_data(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                      const void* packed, const float* d_out1, const eqf_irreps* out1_irreps, const float* d_out2, int n2,
                      float* dx, float* dw, float* d_coupling, int E, int mode, void* stream);
int eqf_sfcx_bwd_weight(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                        const float* d_out1, const eqf_irreps* out1_irreps, const float* d_out2, int n2,
                        float* const* dWl, float* dW2, int E, int mode, void* stream);

/* The same three kernels with the Gate in front of the operator folded into them (GraphAttention: value = sep_value(
 * sep_act.gate(sep_act.lin(...))), nets/graph_attention_transformer.py:494-496 with the Gate of nets/fast_activation.py:
 * 132-148): `x_raw` holds the GATE'S INPUT rows [scalars (S) | gates (G) | gated segments], S + G + dim(gated) floats each,
 * and the operator's input row (the gate's output [scalars | gated segments], which `paths` describes) is never written:
 * every kernel applies c_silu * silu to the scalar segment and c_sig * sigmoid(gate of the channel) to the l > 0 segments
 * where it loads x, and the data gradient returns d_x_raw -- the gradient of the raw rows, i.e. the gate's backward as well
 * (scalars, gates and gated parts; every element written once).  Operators without a second consumer (n2 = 0); degrees <= 2.
 * EQF_E_UNSUPPORTED otherwise: run eqf_gate_* and the plain entry points. */
typedef struct eqf_gate_in {
  int S, G;
  float c_silu, c_sig;
} eqf_gate_in;
int eqf_sfcx_fwd_gated(const float* x_raw, const eqf_gate_in* gate, const float* coupling, const float* w,
                       const eqf_dtp_paths* paths, const void* packed, const float* bias0, float* out1,
                       const eqf_irreps* out1_irreps, int E, int mode, void* stream);
int eqf_sfcx_bwd_data_gated(const float* x_raw, const eqf_gate_in* gate, const float* coupling, const float* w,
                            const eqf_dtp_paths* paths, const void* packed, const float* d_out1,
                            const eqf_irreps* out1_irreps, float* d_x_raw, float* dw, float* d_coupling, int E, int mode,
                            void* stream);
int eqf_sfcx_bwd_weight_gated(const float* x_raw, const eqf_gate_in* gate, const float* coupling, const float* w,
                              const eqf_dtp_paths* paths, const float* d_out1, const eqf_irreps* out1_irreps,
                              float* const* dWl, int E, int mode, void* stream);

/* The two weight-gradient launches with the BIAS gradients of the operator's linears taken along: d_bias0[N1(0)] += column sums
 * of the scalar block of d_out1, d_bias2[n2] += column sums of d_out2 (either may be NULL), accumulated by the work items
 * that stream those columns anyway -- the separate eqf_colsum launch per bias re-read them (19 launches per QM9 step).
 * [ref: the 0e bias of LinearRS / FullyConnectedTensorProductRescale, nets/tensor_product_rescale.py:125-136, whose gradient
 *  autograd forms as dy.sum(0) for sep_act.lin, sep_alpha and sep_value.lin, nets/graph_attention_transformer.py:445-451] */
int eqf_sfcx_bwd_weight_bias(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                             const float* d_out1, const eqf_irreps* out1_irreps, const float* d_out2, int n2,
                             float* const* dWl, float* dW2, float* d_bias0, float* d_bias2, int E, int mode, void* stream);
int eqf_sfcx_bwd_weight_gated_bias(const float* x_raw, const eqf_gate_in* gate, const float* coupling, const float* w,
                                   const eqf_dtp_paths* paths, const float* d_out1, const eqf_irreps* out1_irreps,
                                   float* const* dWl, float* d_bias0, int E, int mode, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row-local feature ops (nodes or edges)
 * ------------------------------------------------------------------------------------------- */

/* EquivariantLayerNormV2 ('component' normalisation, affine).  rstd is [rows, nseg], mean0 is [rows]
 * (mean of the first 0e segment); both are outputs of fwd and inputs of bwd.
 * [ref: nets/layer_norm.py:89-152] */
int eqf_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd,
                      float* mean0, int rows, const eqf_irreps* irreps, float eps, void* stream);
/* dx written; d_weight[num_irreps], d_bias[mul of 0e] ACCUMULATED. */
int eqf_layernorm_bwd(const float* x, const float* weight, const float* dy, const float* rstd,
                      const float* mean0, float* dx, float* d_weight, float* d_bias, int rows,
                      const eqf_irreps* irreps, void* stream);

/* The same with the residual add in front of the norm folded in [ref: TransBlock.forward, nets/graph_attention_transformer.py:
 * 639-667: node_output = node_input + ga(...); ffn(norm_2(node_output)) ...]: xsum = x + x2 is written and normalised in one
 * pass (x2 == xsum == NULL: plain layer norm).  bwd: dx = LN'(dy) + dres (dres = gradient arriving at xsum from the
 * residual branch, may be NULL); d_weight / d_bias ACCUMULATED (both may be NULL together); `x` is the normalised input
 * (xsum of the forward). */
int eqf_add_layernorm_fwd(const float* x, const float* x2, float* xsum, const float* weight, const float* bias,
                          float* y, float* rstd, float* mean0, int rows, const eqf_irreps* irreps, float eps,
                          void* stream);
int eqf_add_layernorm_bwd(const float* x, const float* weight, const float* dy, const float* dres, const float* rstd,
                          const float* mean0, float* dx, float* d_weight, float* d_bias, int rows,
                          const eqf_irreps* irreps, void* stream);

/* Gate: in = [scalars(S) | gates(G) | gated segments], out = [c_silu*silu(scalars) | gated * c_sig*sigmoid(gates)].
 * `gated` lists the l>0 segments (sum of mul = G).  in rows have S+G+dim(gated) floats, out rows S+dim(gated).
 * [ref: nets/fast_activation.py:132-148] */
int eqf_gate_fwd(const float* in, float* out, int rows, int S, const eqf_irreps* gated, float c_silu,
                 float c_sig, void* stream);
int eqf_gate_bwd(const float* in, const float* d_out, float* d_in, int rows, int S,
                 const eqf_irreps* gated, float c_silu, float c_sig, void* stream);

/* y = c * silu(x) elementwise over n floats, and its backward.  [ref: nets/fast_activation.py:68-71] */
int eqf_silu_fwd(const float* x, float* y, long n, float c, void* stream);
int eqf_silu_bwd(const float* x, const float* dy, float* dx, long n, float c, void* stream);

/* Radial-MLP inner step: y = silu(LayerNorm_C(x) * gamma + beta), rows of C <= 64 channels (C == 64 in
 * every reference config).  [ref: nets/radial_func.py:13-36 (nn.LayerNorm + nn.SiLU)] */
int eqf_lnsilu_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int C,
                   float eps, void* stream);
/* dx written; d_gamma[C], d_beta[C] ACCUMULATED. */
int eqf_lnsilu_bwd(const float* x, const float* gamma, const float* beta, const float* dy, float* dx,
                   float* d_gamma, float* d_beta, int rows, int C, float eps, void* stream);

/* The same on a [rows][groups][C] tensor with per-group parameters gamma / beta [groups][C] (d_gamma / d_beta alike):
 * the radial MLPs of all blocks of a model evaluated side by side on the shared radial basis
 * [ref: every TransBlock owns a RadialProfile over the same edge_length_embedding,
 *  nets/graph_attention_transformer.py:200-208,445-447,717 and :880-886]. */
int eqf_lnsilu_group_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, int groups,
                         float eps, void* stream);
int eqf_lnsilu_group_bwd(const float* x, const float* gamma, const float* beta, const float* dy, float* dx,
                         float* d_gamma, float* d_beta, int rows, int C, int groups, float eps, void* stream);

/* Shared (internal) depth-wise weights folded into the rows of the linear that follows the tensor product:
 * out[i] = W[i] * w[w_of_row[row(i)]] over the flat weight W (row r = elements row_start[r] .. row_start[r+1]); bwd: dW[i] =
 * g[i] * w[...] and dw[w_of_row[r]] = <g[row r], W[row r]> (written; rows <-> shared weights one to one).
 * [ref: SeparableFCTP(internal_weights=True) = sep_value, nets/graph_attention_transformer.py:449-451] */
int eqf_fold_weight_fwd(const float* W, const float* w, const int* row_start, const int* w_of_row, float* out, int rows,
                        void* stream);
int eqf_fold_weight_bwd(const float* W, const float* w, const int* row_start, const int* w_of_row, const float* g,
                        float* dW, float* dw, int rows, void* stream);

/* Atom-type embedding: y[n, 0:C] = W[type[n], 0:C] + b[0:C], remaining D-C floats of the row zeroed
 * (LinearRS applied to a one-hot vector).  [ref: nets/graph_attention_transformer.py:682-690] */
int eqf_embed_fwd(const int* type, const float* W, const float* b, float* y, int rows, int C, int D,
                  void* stream);
/* dW[type[n], :] += dy[n, 0:C]; db += dy[n, 0:C]  (ACCUMULATED). */
int eqf_embed_bwd(const int* type, const float* dy, float* dW, float* db, int rows, int C, int D,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * Edge-wise gather / scatter over the dst-sorted radius graph
 * ------------------------------------------------------------------------------------------- */

/* msg[e,:] = a[src[e],:] + b[dst[e],:]   (b may be NULL).
 * [ref: nets/graph_attention_transformer.py:487, :729] */
int eqf_gather_add_fwd(const float* a, const float* b, const int* src, const int* dst, float* msg,
                       int E, int D, void* stream);

/* out[n,:] (=|+=) scale * sum_{q in [ptr[n], ptr[n+1])} x[perm ? perm[q] : q, :]
 * Segmented reduction without atomics: the backward of the gather above (ptr/perm = CSR over dst or
 * over src), ScaledScatter over edges or over the nodes of a molecule.
 * [ref: torch_scatter.scatter call sites, nets/graph_attention_transformer.py:513,700] */
int eqf_segment_sum(const float* x, const int* ptr, const int* perm, float* out, int nseg, int D,
                    float scale, int accumulate, void* stream);
/* out[q,:] = scale * x[seg_of[q],:]  (row broadcast: backward of eqf_segment_sum with perm == NULL). */
int eqf_segment_bcast(const float* x, const int* seg_of, float* out, int rows, int D, float scale,
                      void* stream);

/* out[q,:] = s[seg_of[q]] * x[q,:]  (D % 4 == 0; out may alias x).  Per-graph stochastic depth: s holds 0 or
 * 1/keep_prob per graph.  Linear in x: its backward is the same call on dy.
 * [ref: nets/drop.py:45-61 GraphDropPath; nets/graph_attention_transformer.py:652-664] */
int eqf_segment_scale(const float* x, const float* s, const int* seg_of, float* out, int rows, int D,
                      void* stream);

/* Depth-wise tensor product, un-fused form (kept as the building block for shapes the fused GEMM
 * does not cover and as its on-device cross-check): out[e, :] per eqf_dtp_paths.  w may be NULL.
 * [ref: DepthwiseTensorProduct + o3.TensorProduct('uvu'), nets/graph_attention_transformer.py:157-183] */
int eqf_dtp_fwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                float* out, int E, void* stream);
/* dx[E,in_dim] written; dw[E,w_numel] written if non-NULL (w == NULL means unit weights);
 * d_coupling[E,m_numel] written if non-NULL (needed only when forces are differentiated). */
int eqf_dtp_bwd(const float* x, const float* coupling, const float* w, const eqf_dtp_paths* paths,
                const float* d_out, float* dx, float* dw, float* d_coupling, int E, void* stream);

/* Attention logits: logit[e,h] = sum_k c * SmoothLeakyReLU_0.2(a[e, h*Kh+k]) * alpha_dot[h,k].
 * [ref: nets/graph_attention_transformer.py:54-63,506-507] */
int eqf_alpha_fwd(const float* a, const float* alpha_dot, float* logit, int E, int H, int Kh, float c,
                  void* stream);
/* da written; d_alpha_dot[H*Kh] ACCUMULATED. */
int eqf_alpha_bwd(const float* a, const float* alpha_dot, const float* d_logit, float* da,
                  float* d_alpha_dot, int E, int H, int Kh, float c, void* stream);

/* Per-destination softmax + weighted aggregation (no atomics, deterministic):
 *   alpha[e,h] = exp(logit[e,h]-max_seg)/(sum_seg + 1e-16);   keep[e,h] = dropout mask/(1-p) (p = 0: 1)
 *   out[n, c]  = sum_{e in seg(n)} alpha[e,head(c)] * keep[e,head(c)] * value[e,c]
 * value/out rows follow `irreps` (the H-head irreps: channel u of a segment belongs to head u / (mul/H)).
 * alpha[E,H] (post-softmax, pre-dropout) is an output saved for the backward.  drop_p in [0,1);
 * the mask is a counter-based hash of (seed, e*H+h), recomputed identically in the backward.
 * [ref: torch_geometric.utils.softmax + Dropout + scatter, nets/graph_attention_transformer.py:508-514] */
int eqf_attn_aggregate_fwd(const float* logit, const float* value, const int* row_ptr, float* alpha,
                           float* out, int N, int H, const eqf_irreps* irreps, float drop_p,
                           unsigned long long seed, void* stream);
/* d_value[E,D], d_logit[E,H] written. */
int eqf_attn_aggregate_bwd(const float* alpha, const float* value, const int* row_ptr,
                           const float* d_out, float* d_value, float* d_logit, int N, int H,
                           const eqf_irreps* irreps, float drop_p, unsigned long long seed,
                           void* stream);

/* The same two launches with the dropout seed = seed + *seed_offset, read on the device: a launch captured in a HIP graph
 * (equiformer_amd/capture.py) draws a fresh mask at every replay when the caller advances the device word in between (the
 * by-value seed of a captured launch is frozen).  seed_offset may be NULL (= the entry points above).
 * [ref: the per-step Dropout draw of nets/graph_attention_transformer.py:510-511] */
int eqf_attn_aggregate_fwd_dseed(const float* logit, const float* value, const int* row_ptr, float* alpha,
                                 float* out, int N, int H, const eqf_irreps* irreps, float drop_p,
                                 unsigned long long seed, const unsigned long long* seed_offset, void* stream);
int eqf_attn_aggregate_bwd_dseed(const float* alpha, const float* value, const int* row_ptr,
                                 const float* d_out, float* d_value, float* d_logit, int N, int H,
                                 const eqf_irreps* irreps, float drop_p, unsigned long long seed,
                                 const unsigned long long* seed_offset, void* stream);

/* ---- dot-product attention (the dp_attention_transformer family) ---------------------------------------------------
 * `irreps` = the H-head irreps of q / k / v rows (segment mul = H * channels-of-head, channels-of-head % 4 == 0,
 * H <= 8).  The key/value row of an edge is 2D wide: in every segment [2l+1][2*mul] the first `mul` channels of an m-row
 * are the H key heads, the last `mul` the H value heads (Vec2AttnHeads(irreps_head, 2H) followed by narrow).
 * [ref: nets/dp_attention_transformer.py:131-152] */
/* k[E,D], v[E,D] <- kv[E,2D] */
int eqf_kv_split(const float* kv, float* k, float* v, int E, int H, const eqf_irreps* irreps, void* stream);
/* kv[E,2D] <- k, v (either may be NULL = zeros): the backward of eqf_kv_split */
int eqf_kv_merge(const float* k, const float* v, float* kv, int E, int H, const eqf_irreps* irreps, void* stream);
/* logit[e,h] = sum over the channels of head h of  scale_l * q[dst[e], .] * k[e, .],
 * scale_l = 1 / sqrt(num_irreps(head) * (2l+1))  (ScaleFactor, :45-66, applied to q upstream).  D <= 1024. */
int eqf_dp_logits_fwd(const float* q, const float* k, const int* dst, float* logit, int E, int H,
                      const eqf_irreps* irreps, void* stream);
/* dq[N,D] (needs k) and/or dk[E,D] (needs q) written; either output may be NULL.  Edges are dst-sorted (row_ptr), dq is
 * a segmented reduction without atomics.  The logits are bilinear, so these two entry points also serve the
 * second-order pass. */
int eqf_dp_logits_bwd(const float* q, const float* k, const float* d_logit, const int* row_ptr, float* dq, float* dk,
                      int N, int H, const eqf_irreps* irreps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Second-derivative entry points (MD17 force-loss training).  The reference takes forces with
 * create_graph=True and back-propagates a loss on them [ref: nets/graph_attention_transformer_md17.py:
 * 318-325, main_md17.py:384-390], so autograd differentiates the first backward pass.  For an
 * operator whose first-order backward wrote dx = J^T dy, each *_bwd2 takes the cotangent c of dx and
 * returns the gradient of Phi = <c, dx> wrt the operator's input(s), parameters (ACCUMULATED) and dy.
 * The multilinear operators (linears, fused SeparableFCTP, gathers, coupling) need no extra entry
 * points: their second-order terms are their first-order kernels with one argument substituted.
 * ------------------------------------------------------------------------------------------- */
int eqf_silu_bwd2(const float* x, const float* dy, const float* c, float* g_x, float* g_dy, long n, float c0,
                  void* stream);
/* c, g_in: [rows, S+G+dim(gated)]; d_out, g_dout: [rows, S+dim(gated)] */
int eqf_gate_bwd2(const float* in, const float* d_out, const float* c, float* g_in, float* g_dout, int rows, int S,
                  const eqf_irreps* gated, float c_silu, float c_sig, void* stream);
/* g_gamma[C], g_beta[C] ACCUMULATED */
int eqf_lnsilu_bwd2(const float* x, const float* gamma, const float* beta, const float* dy, const float* c,
                    float* g_x, float* g_gamma, float* g_beta, float* g_dy, int rows, int C, float eps, void* stream);
/* the same on [rows][groups][C] rows with per-group gamma / beta (g_gamma / g_beta [groups][C], ACCUMULATED): the second-order
 * term of the radial bank -- all RadialProfile MLPs of a model side by side -- when forces are taken with create_graph
 * [ref: nets/radial_func.py:13-36 under nets/graph_attention_transformer_md17.py:318-325] */
int eqf_lnsilu_group_bwd2(const float* x, const float* gamma, const float* beta, const float* dy, const float* c,
                          float* g_x, float* g_gamma, float* g_beta, float* g_dy, int rows, int C, int groups, float eps,
                          void* stream);
/* g_weight[num_irreps] ACCUMULATED (the bias does not enter dx) */
int eqf_layernorm_bwd2(const float* x, const float* weight, const float* dy, const float* c, float* g_x,
                       float* g_weight, float* g_dy, int rows, const eqf_irreps* irreps, float eps, void* stream);
/* ca: cotangent of da [E, H*Kh]; g_alpha_dot[H*Kh] ACCUMULATED */
int eqf_alpha_bwd2(const float* a, const float* alpha_dot, const float* d_logit, const float* ca, float* g_a,
                   float* g_alpha_dot, float* g_dlogit, int E, int H, int Kh, float c, void* stream);
/* c_value [E,D] / c_logit [E,H]: cotangents of d_value / d_logit (either may be NULL = zero); the dropout mask is
 * the one of the forward (same seed).  g_logit[E,H], g_value[E,D], g_dout[N,D] written. */
int eqf_attn_aggregate_bwd2(const float* alpha, const float* value, const int* row_ptr, const float* d_out,
                            const float* c_value, const float* c_logit, float* g_logit, float* g_value,
                            float* g_dout, int N, int H, const eqf_irreps* irreps, float drop_p,
                            unsigned long long seed, void* stream);
/* c_len [E]: cotangent of d_len; g_len[E], g_dout[E,R] written */
int eqf_rbf_expnorm_bwd2(const float* len, const float* d_out, const float* c_len, int E, int R, const float* means,
                         const float* betas, float alpha, float cutoff, float* g_len, float* g_dout, void* stream);
/* as above for the Gaussian basis; g_mean[R], g_std[R], g_weight[1], g_bias[1] ACCUMULATED */
int eqf_rbf_gaussian_bwd2(const float* len, const float* d_out, const float* c_len, int E, int R, const float* mean,
                          const float* std, const float* weight, const float* bias, float cutoff, float* g_len,
                          float* g_dout, float* g_mean, float* g_std, float* g_weight, float* g_bias, void* stream);
/* as above for the spherical Bessel basis; g_freq[R] ACCUMULATED */
int eqf_rbf_bessel_bwd2(const float* len, const float* d_out, const float* c_len, int E, int R, const float* freq,
                        float cutoff, float* g_len, float* g_dout, float* g_freq, void* stream);
/* c_vec [E,3]: cotangent of d_vec.  g_vec[E,3] written; g_dsh[E,(lmax+1)^2] and g_dlen[E] written if non-NULL;
 * d_sh / d_len may be NULL exactly as in eqf_edge_geom_bwd. */
int eqf_edge_geom_bwd2(const float* vec, const float* d_sh, const float* d_len, const float* c_vec, int E, int lmax,
                       float* g_vec, float* g_dsh, float* g_dlen, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart; used by bench.py for the roofline line)
 * ------------------------------------------------------------------------------------------- */

/* filter == NULL: stop recording.  Otherwise record a HIP-event pair (on the launch stream) around every
 * matrix-core kernel launch whose name contains `filter` ("" = all). */
int eqf_prof_enable(const char* filter);
/* Wait for the recorded events; write "name launches total_ms algorithmic_flops algorithmic_bytes\n" per kernel
 * into buf (host memory); clears the records.  Returns bytes written or < 0. */
int eqf_prof_report(char* buf, int buflen);

/* ---- optimizer step on one flat fp32 buffer (the step right after the hot path) -------------------------------
 * eqf_sumsq: out[0] = sum g[i]^2 (zeroed here first) -- the global gradient norm of clip_grad_norm_
 *   [ref: engine.py:76-78 dispatch_clip_grad(..., mode='norm')].
 * eqf_adamw_step: torch.optim.AdamW update of p[n] from g[n] with state m, v (step = 1-based step count):
 *   g *= min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) when sumsq != NULL; p *= 1 - lr*wd[i]; m = b1 m + (1-b1) g;
 *   v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps); wd[n] = per-element weight decay
 *   (0 for the reference's no-decay names, optim_factory.py:27-42; wd[i] < 0 marks an element whose parameter has
 *   no gradient this step: p, m, v stay as they are, as torch.optim.AdamW skips such parameters); when ema != NULL also
 *   ema = ema_decay*ema + (1-ema_decay)*p  [ref: timm ModelEmaV2.update called at engine.py:89-90].              */
int eqf_sumsq(const float* g, long n, float* out, void* stream);
int eqf_adamw_step(float* p, const float* g, float* m, float* v, const float* wd, float* ema, const float* sumsq,
                   long n, float lr, float beta1, float beta2, float eps, int step, float max_norm, float ema_decay,
                   void* stream);
/* The same update with {lr, 1 - b1^t, sqrt(1 - b2^t)} read from the device array hyper[3] (fp32, computed by the host in
 * double as above): the form a HIP-graph-captured train step launches -- by-value arguments are frozen at capture, the host
 * refreshes the three floats before every replay (equiformer_amd/capture.py).  [ref: the per-step lr of the drivers'
 * schedulers, engine.py:73-90] */
int eqf_adamw_step_dev(float* p, const float* g, float* m, float* v, const float* wd, float* ema, const float* sumsq,
                       long n, const float* hyper, float beta1, float beta2, float eps, float max_norm,
                       float ema_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EQUIFORMER_HIP_H */
