/*
 * optex.h — C ABI of liboptex_hip.so: the MI355X (gfx950) implementation of the sliced-optimal-transport
 * inner loop of JCBrouwer/OptimalTextures.
 *
 * The reference has no FFI: its hot path is five plain Python functions.  Each entry point
 * below names the reference code it replaces (file:line in the reference repository); the Python shim that
 * keeps the reference signatures is optimaltextures_amd/{optex,histmatch}.py, and INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  Every function returns 0 on success or a negative
 *    OPTEX_E_* code; optex_last_error() returns a thread-local message for the last failure.
 *  - All pointers are DEVICE pointers (fp32 unless stated) on the device current when the call is made.
 *    The library never allocates, frees or synchronizes: the caller owns every buffer including scratch
 *    (`ws`, sized by the *_ws_bytes helpers) and every call is asynchronous on `stream` (a hipStream_t;
 *    NULL = the legacy default stream).  Calls are therefore stream-ordered, re-entrant and capturable
 *    in a hipGraph.
 *  - Every `void* ws` is followed by `size_t ws_bytes`, the size of the buffer the caller really owns: a call
 *    whose scratch is smaller than its *_ws_bytes helper asks for returns OPTEX_E_ARG before anything is
 *    enqueued (ABI 3; ABI 2 trusted the pointer).
 *  - Feature tensors are "channel-major segments": segment s (one independent texture), channel c,
 *    pixel i lives at  base + s*seg_stride + c*ld + i  (fp32 elements).  NCHW-contiguous memory is
 *    (ld = H*W, seg_stride = C*H*W); the reference's pooled layout hist.view(c, -1) (histmatch.py:11,17)
 *    is (ld = B*H*W, seg_stride = H*W) or simply one segment of n = B*H*W.
 *  - "pixel-major" (layout = OPTEX_PIXEL_MAJOR) is the reference's NHWC-contiguous [n, C] layout:
 *    pixel i, channel c at  base + s*seg_stride + i*ld + c.
 *  - `unsigned flags` (ABI 10; the argument in front of `stream` of the five entry points that launch the hot kernels):
 *    per-CALL choices that ABI 8 / 9 kept in process-wide setters.  0 = the defaults.  They hold for this call only (and for
 *    whatever it runs inside), on the calling thread only: two threads — two devices, two OptimalTexture objects — with
 *    different choices do not see each other (re-entrant per call, no global mutable state).
 */
#ifndef OPTEX_H
#define OPTEX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPTEX_ABI_VERSION 10
#define OPTEX_BINS 256 /* histmatch.py:49 `bins: int = 256` (the only value any caller uses) */

enum { OPTEX_OK = 0, OPTEX_E_ARG = -1, OPTEX_E_LAUNCH = -2, OPTEX_E_UNSUPPORTED = -3 };
enum { OPTEX_CHANNEL_MAJOR = 0, OPTEX_PIXEL_MAJOR = 1 };

/* flags (ABI 10) */
#define OPTEX_F_DEFAULT 0u
/* bits 0-7: the persistent R-stationary rotation GEMM leaves `n` of the CUs out of its grid (n <= 254; see optex_gemm_spare_cus
 * below for why).  Not given: 1.  OPTEX_F_SPARE_CUS(0) = one workgroup on every CU. */
#define OPTEX_F_SPARE_CUS(n) ((((unsigned)(n)) + 1u) & 0xffu)
/* the cdf matcher as the two-kernel pipeline (cdf_hist_lut_kernel + cdf_apply_kernel) even where the one-kernel matcher with the
 * column in registers would run — the same bits either way (tests/test_gpu_parity.py compares them) */
#define OPTEX_F_CDF_TWO_KERNEL 0x100u
/* the sort matcher on rank_match4_kernel (8-slot windows, rounds 2-5) even where rank_match5w_kernel (round 6: 8-bit buckets,
 * keys alone in their bucket ranked without a window) would run — the same bits either way */
#define OPTEX_F_SORT_RANK4 0x200u

int optex_abi_version(void);
const char* optex_last_error(void);
/* multiProcessorCount / LDS bytes per block of the current device (diagnostics for bench.py) */
int optex_device_info(int* n_cu, int* lds_bytes, int* wavefront);
/* ABI 8.  The R-stationary rotation GEMM is a persistent kernel: one workgroup per compute unit, each needing a WHOLE CU (all of
 * its registers).  A small kernel of another stream that sits on one CU — the rotation generator's sequential MT19937 walk, an
 * RCCL broadcast — then leaves one workgroup of every GEMM launch waiting for a CU: its tiles start when another workgroup has
 * finished, and the launch takes twice as long (measured at 8 textures per step: 46.3 ms per step against 43.8 without the
 * generator).  `spare` CUs are left out of the GEMM's grid (default 1: 0.4 % more work per workgroup, nothing to wait for);
 * 0 restores one workgroup on every CU.  Returns the previous value.
 * DEPRECATED since ABI 10 (kept for one version): this moves the process-wide DEFAULT only — what a call without
 * OPTEX_F_SPARE_CUS(n) in its flags gets; pass the choice with the call instead. */
int optex_gemm_spare_cus(int spare);
/* ABI 9.  optex_cdf_match / the cdf iteration of optex_ot_loop run range + histograms + LUT + interpolation of a column as ONE
 * kernel that keeps the column in registers (columns of at most 16384 values, 16-byte aligned rows, one workgroup per column:
 * every batched call of the hot loop) — the target is read from HBM once instead of twice.  `on` = 0 forces the two-kernel
 * pipeline (cdf_hist_lut_kernel + cdf_apply_kernel: what longer or chunked columns take anyway), the same bits either way
 * (tests/test_gpu_parity.py compares them).  Returns the previous value; default 1.
 * DEPRECATED since ABI 10 (kept for one version): the process-wide default of calls without OPTEX_F_CDF_TWO_KERNEL. */
int optex_cdf_fused(int on);

/* ---------------------------------------------------------------------------------------------------
 * K1  rotation / apply GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   OUT[s][m][i] = sum_k At[s][k][m] * (B[s][k][i] - bsub[s][k]) + badd[s][m]          (k ascending, one
 *   fp32 fma per product: bit-identical to a sequential fmaf chain), then optionally the caller epilogue
 *   of optex.py:115-117:  OUT += strength * (content - OUT)  (content has OUT's layout and strides).
 * Replaces:  optex.py:170 `pastiche_feature @ rotation`   (At = R,   B = pastiche)
 *            optex.py:171 `style_feature @ rotation`      (At = R,   B = style)
 *            optex.py:175 `matched_pastiche @ rotation.T` (At = R^T, B = matched)
 *            histmatch.py:27/34/42,44 `T @ hist_t + mu_s`  (At = T^T, bsub = mu_t, badd = mu_s)
 * At is [K, M] row-major with leading dimension lda; at_seg_stride = 0 shares one matrix between segments.
 * bsub / badd / content may be NULL; their seg strides are in elements (0 = shared).
 * ------------------------------------------------------------------------------------------------- */
int optex_gemm_tn(const float* At, long lda, long at_seg_stride,
                  const float* B, long ldb, long b_seg_stride, int b_layout,
                  float* OUT, long ldo, long o_seg_stride, int o_layout,
                  int M, int K, long n, int n_seg,
                  const float* bsub, long bsub_seg_stride, const float* badd, long badd_seg_stride,
                  const float* content, float strength, unsigned flags, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K2/K3  cdf mode, histmatch.py:49-69 (cdf_match) + histmatch.py:72-92 (interp), 256 bins.
 * Stage entry points (each has its own known-answer test) and the whole pipeline.
 * ------------------------------------------------------------------------------------------------- */
/* histmatch.py:52-53: per (segment, channel) min and max.  mn/mx are [n_seg, C] fp32. */
int optex_col_minmax(const float* x, long ld, long seg_stride, long n, int C, int n_seg, float* mn, float* mx,
                     void* stream);
/* histmatch.py:57-58: torch.histc(x, 256, lo, hi) per (segment, channel).  lo/hi are [n_seg, C]; hist is
 * [n_seg, C, 256] uint32 (exact integer counts; the reference holds the same integers in fp32). */
int optex_col_histc(const float* x, long ld, long seg_stride, long n, int C, int n_seg, const float* lo,
                    const float* hi, uint32_t* hist, void* stream);
/* histmatch.py:72-92 on arbitrary 1-D arrays: out[i] = interp(x[i], xp[0..np), fp[0..np)) */
int optex_interp(const float* x, long nx, const float* xp, const float* fp, long np_, float* out, void* stream);

size_t optex_cdf_ws_bytes(int C, int n_seg);
/* Whole cdf_match for n_seg independent target segments.  The source has src_n_seg in {1, n_seg} segments
 * (1 = every target segment is matched to the same source distribution).  `out` may alias `target`.
 * dbg (NULL or [n_seg, C, 2 + 4*256] fp32) receives lo, hi, hist_t, hist_s, bin_edges, remapped_cdf per
 * column, the intermediates of histmatch.py:52-67. */
int optex_cdf_match(const float* target, long ldt, long t_seg_stride, long nt,
                    const float* source, long lds, long s_seg_stride, long ns, int src_n_seg,
                    int C, int n_seg, float* out, long ldo, long o_seg_stride,
                    void* ws, size_t ws_bytes, float* dbg, unsigned flags, void* stream);

/* histmatch.py:49 `cdf_match(target, source, bins)` with the bin count free (ABI 6).  Every caller inside the reference leaves
 * bins at 256 (optex_cdf_match above, the hot path); this entry serves a direct call with another value.  Same arguments and
 * layout as optex_cdf_match; one workgroup per column does the whole function.  bins >= 1. */
size_t optex_cdf_bins_ws_bytes(int C, int n_seg, int bins);
int optex_cdf_match_bins(const float* target, long ldt, long t_seg_stride, long nt,
                         const float* source, long lds, long s_seg_stride, long ns, int src_n_seg,
                         int C, int n_seg, int bins, float* out, long ldo, long o_seg_stride,
                         void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K6  sort mode — exact 1-D optimal transport per rotated column (north-star addition; there
 * is no reference counterpart, the specification is oracle/optex_oracle.c orc_sort_columns/orc_sort_match).
 * Key order = IEEE totalOrder of the fp32 bit pattern, ties keep pixel order (stable); indices are exact.
 * ------------------------------------------------------------------------------------------------- */
size_t optex_sort_ws_bytes(long n, int C, int n_seg);
/* keys -> out_keys [n_seg, C, n] and out_idx [n_seg, C, n] uint32 (either may be NULL) */
int optex_sort_columns(const float* keys, long ld, long seg_stride, long n, int C, int n_seg, float* out_keys,
                       uint32_t* out_idx, void* ws, size_t ws_bytes, void* stream);
size_t optex_sort_match_ws_bytes(long nt, long ns, int C, int n_seg, int src_n_seg);
/* out[rank_i] = sorted_source[floor((2i+1)*ns / (2*nt))] where rank_i is the pixel holding the i-th smallest
 * target value of the column. */
int optex_sort_match(const float* target, long ldt, long t_seg_stride, long nt,
                     const float* source, long lds, long s_seg_stride, long ns, int src_n_seg,
                     int C, int n_seg, float* out, long ldo, long o_seg_stride, void* ws, size_t ws_bytes, unsigned flags,
                     void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K4  linear modes, histmatch.py:16-22: per-(segment, channel) spatial mean and the centred covariance
 *   cov = hist @ hist.T / N + eps * I.   pool = 0: one covariance per segment (independent textures);
 *   pool = 1: the reference's batch semantics — means per segment, ONE covariance pooled over all segments.
 * mu is [n_seg, C]; cov is [n_seg, C, C] (pool = 0) or [C, C] (pool = 1), fp32.
 * The C x C factorizations of histmatch.py:24-42 are K5 below; the apply GEMM `T @ hist_t + mu_s` is optex_gemm_tn
 * with bsub/badd.
 * ------------------------------------------------------------------------------------------------- */
size_t optex_linear_stats_ws_bytes(long n, int C, int n_seg);
int optex_linear_stats(const float* x, long ld, long seg_stride, long n, int C, int n_seg, int pool, float eps,
                       float* mu, float* cov, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K5  the C x C algebra of the linear modes, histmatch.py:24-42, batched over independent segments on the device
 * (csrc/linalg.hip) — replaces torch.linalg.cholesky / torch.inverse / torch.linalg.eigh of the reference.
 *   optex_chol_inv:  A = L L^T (histmatch.py:25-26) and L^-1 (the `torch.inverse(chol_t)` of :27) in one pass.
 *     U [batch, ld, ld] = L^T (upper), Linv [batch, ld, ld] = L^-1 (lower), ld = optex_chol_ld(C) = C rounded up to 32
 *     (zero outside the triangle, identity in the padding).  C <= 512.
 *   optex_spd_sqrt:  Y = A^1/2, Z = A^-1/2 of symmetric positive definite matrices — the `eve @ sqrt(diag(eva)) @ eve.T`
 *     of histmatch.py:30-31,33,37,40 and its inverse, by the scaled coupled Newton-Schulz iteration (GEMMs only).
 *     lambda_min: a lower bound of the spectrum (the eps of `cov + eps * I`; <= 0 if unknown) — it sets the scaling;
 *     fp32 round-off is reached for |A|_F / lambda_min <= 1e7.  Y, Z [batch, C, C] contiguous, either may be NULL.
 *   optex_transfer_operator:  Tt[s] = T_s^T with  matched = T @ hist_t  for mode 2 = chol (L_s L_t^-1), 3 = pca
 *     (Q_s Q_t^-1), 4 = sym (Q_t^-1 (Q_t S_s Q_t)^1/2 Q_t^-1);  cov_t [n_seg, C, C], cov_s [src_n_seg in {1, n_seg}, C, C]
 *     (both with eps * I already added, as optex_linear_stats returns them; eps is passed again as the spectrum bound of
 *     optex_spd_sqrt), Tt [n_seg, C, C] — the `At` operand of optex_gemm_tn for the apply step.
 * ------------------------------------------------------------------------------------------------- */
int optex_chol_ld(int C);
int optex_chol_inv(const float* A, long a_seg_stride, int C, int batch, float* U, float* Linv, void* stream);
size_t optex_spd_sqrt_ws_bytes(int C, int batch);
int optex_spd_sqrt(const float* A, long a_seg_stride, int C, int batch, float lambda_min, float* Y, float* Z, void* ws,
                   size_t ws_bytes, void* stream);
size_t optex_transfer_operator_ws_bytes(int mode, int C, int n_seg, int src_n_seg);
int optex_transfer_operator(int mode, const float* cov_t, const float* cov_s, int C, int n_seg, int src_n_seg, float eps,
                            float* Tt, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * R0  rotation generator, optex.py:142-149 -> scipy.stats.special_ortho_group.rvs (Householder chain, fp64).
 * `normals` holds, per rotation, the N(N+1)/2 - 1 standard normals scipy would draw (host RNG stream kept
 * on the host so that np.random.seed reproduces the reference's matrices); the O(N^3) accumulation runs on
 * the device.  Outputs (any may be NULL): R64 [count,N,N] fp64, R32 [count,N,N] fp32 (= optex.py:168's cast),
 * Rt32 = transposes of R32.  ws: optex_rotation_ws_bytes(N, count).
 * ------------------------------------------------------------------------------------------------- */
long optex_rotation_normals(int N);
size_t optex_rotation_ws_bytes(int N, int count);
/* The stream itself on the device (ABI 7): numpy's RandomState.normal — MT19937 + the legacy polar method with its one-value
 * cache, what scipy's rvs draws from (optex.py:149) — advanced from / to a state of optex_mt19937_state_bytes() bytes per
 * stream: uint32 key[624], uint32 pos, uint32 has_gauss, double cached_gaussian = RandomState.get_state()[1:5].  `states`
 * holds n_streams of them back to back and is updated in place; stream s writes its next `count` values to
 * out + s * out_stride (out_stride >= count).  One workgroup per stream walks the words, all CUs do the arithmetic.  Words, accept / reject decisions and state are
 * exact; every floating-point operation is the host's IEEE operation except log(), which is correctly rounded here and
 * 0.52-ulp accurate in glibc: >= 99.8 % of the values equal numpy's bit for bit, the rest differ by a few ulp. */
size_t optex_mt19937_state_bytes(void);
/* states of n_streams streams as numpy's RandomState(seed) leaves them for the 32-bit integer seeds first_seed + s * seed_stride
 * (mod 2^32): init_genrand, position 624, cache empty — no host transfer. */
int optex_mt19937_seed(void* states, int n_streams, uint32_t first_seed, uint32_t seed_stride, void* stream);
size_t optex_legacy_normals_ws_bytes(int n_streams, long count);
int optex_legacy_normals(void* states, int n_streams, long count, double* out, long out_stride, void* ws, size_t ws_bytes,
                         void* stream);
int optex_rotations_from_normals(const double* normals, int N, int count, double* R64, float* R32, float* Rt32,
                                 void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused hot loop, optex.py:112-117, every iteration of a (pass, layer) enqueued by one call:
 *   for it in range(iters):  x = ((x @ R_it) matched-to (style @ R_it)) @ R_it^T ; optional content blend
 * x is [n_seg, C, n] channel-major segments (independent textures), updated in place; style is [src_n_seg, C, ns].
 * R32 / Rt32 are [iters, C, C] as produced by optex_rotations_from_normals, shared by all segments (r_seg_stride = 0: the
 *   reference shares one R across its batch, optex.py:168-170), or one set per segment, R32 + s * r_seg_stride (ABI 5:
 *   r_seg_stride >= iters * C * C elements; the reference run once per image draws its own rotations every time).  With
 *   per-segment rotations nothing on the style side is shared either: every segment is matched to its own rotated copy
 *   of the style (linear modes: to the style statistics rotated by its own matrices).  cdf / sort: fuse_rotations = 0 only.
 * mode: 0 = cdf, 1 = sort, 2 = chol, 3 = pca, 4 = sym (histmatch.py:5 `mode`; eps = 1 as every caller leaves it).
 *   The linear modes (2-4) take the style's mean and covariance ONCE per call and rotate them as C x C matrices
 *   (cov(S R) = R^T cov(S) R — the style-feature statistics the multi-GPU path broadcasts); the pastiche side is the
 *   reference's sequence: rotate, centre, covariance of the rotated map, transfer operator (K5), apply, rotate back.
 *   C <= 512.
 * fuse_rotations = 0, the default: cdf / sort run the literal sequence (two feature-map GEMMs per iteration); the linear
 *   modes evaluate their last two products `(T @ hist_t + mu_s) @ R^T` as ONE feature-map GEMM with the C x C matrix
 *   R T (two feature-map GEMMs per iteration instead of three; same map, fp32 round-off differences).
 * fuse_rotations = 2: linear modes with the three feature-map GEMMs kept apart (tests, comparisons); cdf / sort: as 0.
 * fuse_rotations = 1, optional fast paths, never the default, results agree to fp32 round-off per step:
 *   cdf / sort (content must be NULL): `(m @ R_i^T) @ R_{i+1}` is evaluated as `m @ (R_i^T R_{i+1})` — one feature-map
 *     GEMM per iteration instead of two;
 *   linear modes: the whole step as ONE affine map in un-rotated space, x' = M (x - mu_x) + mu_s with M = R T R^T and
 *     cov(x R) = R^T cov(x) R: one covariance + one feature-map GEMM per iteration instead of three.
 * fuse_rotations = 3, optional fast path, never the default; linear modes, content must be NULL: the whole chain of
 *   iterations in C x C algebra.  The covariance the next step needs follows analytically,
 *   cov(x') = M cov(x) M^T, so the feature map is read once for its statistics and once by the single GEMM that applies
 *   M_k ... M_1; agrees with the literal chain to accumulated fp32 round-off (tests/test_gpu_linalg.py).
 * Scratch: optex_ot_loop_ws_bytes is a pure function of its arguments (no dependence on the current device).  In cdf mode
 *   (and sort mode with ns <= 16384) with ONE rotation sequence for the batch and a shared style, the style side of all
 *   iterations is prepared before the loop and kept in the scratch: iters rotated copies of the style, up to 1 GiB more
 *   than a single copy; beyond that budget the style is rotated (and sorted) per iteration — same bits either way.
 * ------------------------------------------------------------------------------------------------- */
size_t optex_ot_loop_ws_bytes(int mode, long n, long ns, int C, int n_seg, int src_n_seg, int iters,
                              int fuse_rotations, long r_seg_stride);
int optex_ot_loop(int mode, float* x, long n, int n_seg, const float* style, long ns, int src_n_seg, int C,
                  const float* R32, const float* Rt32, long r_seg_stride, int iters, const float* content, float strength,
                  int fuse_rotations, void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* The same loop between the PCA projection and unprojection of optex.py:109-110,119-120 (ABI 7), the reference's
 * default (PCA on):   x = feat @ E;  [iterations];  feat = x @ E^T   with E = eigvecs [C_full, C].
 * x_full [n_seg, C_full, n] holds the un-projected encoder features on entry and the un-projected result on exit; style
 * [src_n_seg, C, ns] and content (NULL or [n_seg, C, n]) are projected already (they are per call, not per iteration); eig is
 * E row-major [C_full, C], eig_t its transpose [C, C_full]; R32 / Rt32 [iters, C, C], one sequence for the batch.
 * The projection is FOLDED into the first rotation,  (feat @ E) @ R_0 = feat @ (E R_0), and — cdf / sort without a content
 * blend — the unprojection into the last rotation back,  (m @ R_l^T) @ E^T = m @ (E R_l)^T: two of the 2 * iters + 2
 * feature-map GEMMs of a (pass, layer) disappear.  Same products in another association: results agree with
 * project -> optex_ot_loop -> unproject to fp32 round-off (tests/test_gpu_parity.py), not bit for bit.  fuse_rotations = 0.
 * Arguments are validated before anything is enqueued.  The k-space state of the loop lives in the scratch only: with the
 * unprojection folded (cdf / sort, content == NULL) nothing but x_full holds a result afterwards. */
size_t optex_ot_loop_pca_ws_bytes(int mode, long n, long ns, int C, int C_full, int n_seg, int src_n_seg, int iters);
int optex_ot_loop_pca(int mode, float* x_full, int C_full, const float* eig, const float* eig_t, long n, int n_seg,
                      const float* style, long ns, int src_n_seg, int C, const float* R32, const float* Rt32, int iters,
                      const float* content, float strength, void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * N3  element-wise glue between the VGG convolutions, vgg.py:14-135: conv bias add, nn.ReLU,
 * nn.MaxPool2d(2, 2, ceil_mode=True) (vgg.py:26 ...), nn.UpsamplingNearest2d(2) (vgg.py:82 ...) and the
 * nn.ReflectionPad2d(1) in front of every 3x3 convolution, applied in that order in ONE pass:
 *   out[N, C, Ho, Wo] = pad(up(pool(relu(x[N, C, H, W] + bias[C]))))     (each stage optional; pool and up exclusive)
 * Ho = (pool ? ceil(H/2) : up ? 2H : H) + 2*pad.  NCHW-contiguous fp32, bias may be NULL.  The convolutions themselves
 * stay on PyTorch-ROCm (MIOpen) and run bias-free.  Results are bit-identical to the PyTorch module sequence.
 * ------------------------------------------------------------------------------------------------- */
int optex_vgg_glue(const float* x, const float* bias, float* out, int N, int C, int H, int W, int relu, int pool, int up,
                   int pad, void* stream);
/* The same pass with the memory layout of each side chosen independently: 0 = NCHW (planar), 1 = NHWC (channels-last,
 * what MIOpen's fp32 implicit-GEMM convolutions prefer).  A layout change costs nothing extra: the glue reads one and
 * writes the other.  NHWC on both sides needs C % 4 == 0. */
int optex_vgg_glue_layout(const float* x, const float* bias, float* out, int N, int C, int H, int W, int relu, int pool,
                          int up, int pad, int in_nhwc, int out_nhwc, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Measurement (no reference counterpart; the reference only wall-clocks forward(), optex.py:285-289).
 * When enabled, every kernel launch above is bracketed by HIP events recorded on its own stream and tallied
 * per kernel class together with its ALGORITHMIC flops / bytes.  optex_prof_collect is the only
 * blocking call of the library: it waits for the recorded events, returns per-class totals and resets them.
 * ------------------------------------------------------------------------------------------------- */
int optex_prof_enable(int on);
int optex_prof_num_classes(void);
const char* optex_prof_class_name(int cls);
int optex_prof_collect(int n_classes, double* ms, long long* launches, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* OPTEX_H */
