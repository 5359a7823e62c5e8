// ot_loop.hip — the reference's hot loop (optex.py:112-117), all iterations of a (pass, layer) enqueued back-to-back on
// one stream from C++ (no Python, no host synchronisation between launches):
//   for it in range(iters):  x = hist_match(x @ R, style @ R, mode) @ R.T ;  x += strength * (content - x)
// Everything stays channel-major ([segment][channel][pixel] = NCHW memory), so no kernel in the loop transposes anything.
//
// modes 0 / 1 (cdf, sort): the style is rotated every iteration like the reference does (the matchers need its samples).
// modes 2 / 3 / 4 (chol, pca, sym — histmatch.py:16-44): the matcher only needs the style's mean and covariance, and
//   cov(S R) = R^T cov(S) R,  mean(S R) = mean(S) R   (the eps * I term is rotation-invariant),
// so the style statistics are taken ONCE per call and rotated as C x C matrices for all iterations up front — exactly the
// "style-feature statistics" the north star broadcasts between GPUs.  The pastiche side follows the reference step by
// step — rotate, centre, covariance of the rotated map, transfer operator — with the C x C factorizations on the device
// (linalg.hip); the last two products, `T @ hist_t` and `@ R^T`, are evaluated as one feature-map GEMM with the C x C
// matrix R T (fuse_rotations = 2 keeps them apart, 1 is the labelled single-affine fast path, 3 the labelled collapsed
// chain: the statistics follow every step analytically and the feature map is touched twice per call).
#include "gemm_args.h"

using namespace optex;

namespace optex {
int small_gemm(const float* At, long lda, long at_ss, const float* B, long ldb, long b_ss, float* O, long ldo, long o_ss, int C,
               int batch, bool epi, float alpha, const float* alpha_seg, float diag, hipStream_t st, bool sym);
int small_gemm_nn_ld(const float* A, long lda, long a_ss, const float* B, long b_ss, float* O, long o_ss, int C, int batch,
                     hipStream_t st);
int small_gemm_nn(const float* A, long a_ss, const float* B, long b_ss, float* O, int C, int batch, float alpha,
                  const float* alpha_seg, float diag, const int* live_until, int live_idx, hipStream_t st);
__global__ void rot_mean_kernel(const float* __restrict__ R, long r_ss, const float* __restrict__ mu, int mu_per_set, int C, int per,
                                float* __restrict__ out);
int chol_np(int C);
int launch_chol_inv(const float* A, long a_ss, int C, int batch, float* U, float* Linv, hipStream_t st);
size_t ns_ws_floats(int C, int batch);
int ns_sqrt(const float* A, long a_ss, int C, int batch, float lambda_min, float* buf, float** Yout, float** Zout, hipStream_t st);
}  // namespace optex

namespace {

enum { MODE_CDF = 0, MODE_SORT = 1, MODE_CHOL = 2, MODE_PCA = 3, MODE_SYM = 4 };
constexpr size_t kHoistBytes = (size_t)1 << 30;  // rotated style copies of all iterations of a call: at most 1 GiB
constexpr float kEps = 1.0f;  // histmatch.py:5 `eps: float = 1`; no caller overrides it (optex.py:173,200-201)

// Leading dimension of the rotated scratch map of the linear modes.  The Gram kernel stages eight rows of a pixel chunk per
// wave: with rows exactly 64 KiB apart (n = 16384, the 512^2 pass of relu3_1) every row of a staged chunk sits on the same
// memory channel and the kernel loses 17 % (727 us against 604 / 618 us at n = 16352 / 16416, profiles/r03_gram_kernels.md).
// The map is scratch of this file: its rows simply get 256 bytes further apart.
long padded_ld(long n) { return (n * (long)sizeof(float)) % 65536 == 0 ? n + 64 : n; }

// bump allocator over the caller's scratch; with base == nullptr it only measures
struct Bump {
    char* base;
    size_t off = 0;
    explicit Bump(void* b) : base(static_cast<char*>(b)) {}
    template <typename T>
    T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += align_up(count * sizeof(T), 256);
        return p;
    }
};

struct LoopWs {
    // cdf / sort
    float* y = nullptr;    // rotated pastiche [n_seg, C, n]
    float* ys = nullptr;   // rotated style    [src_n_seg, C, ns]
    float* y2 = nullptr;   // second rotated buffer (fused rotations: the re-rotation cannot run in place) / matched map (linear)
    float* P = nullptr;    // [iters - 1, C, C] re-rotation matrices R_i^T R_{i+1} (fused rotations, cdf / sort)
    void* mode_ws = nullptr;
    size_t mode_ws_bytes = 0;
    // linear modes
    void* stats_ws = nullptr;
    size_t stats_ws_bytes = 0;
    float *mu_t = nullptr, *cov_t = nullptr, *mu_s = nullptr, *cov_s = nullptr;   // [n_seg, C], [n_seg, C, C], [Ss, C], [Ss, C, C]
    float *tmp_s = nullptr, *cov_sr = nullptr, *mu_sr = nullptr;                  // per (iteration, style segment): [NS, C, C], [NS, C, C], [NS, C]
    float* At = nullptr;                                                           // transfer operators, transposed: [n_seg, C, C]
    float *Us = nullptr, *Ls = nullptr, *Ut = nullptr, *Lt = nullptr;              // chol: [NS | n_seg, NP, NP]
    float *Ys = nullptr, *Yt = nullptr, *Zt = nullptr, *G1 = nullptr, *G = nullptr; // pca / sym
    float* ns_buf = nullptr;
    // fused (single-affine) linear path
    float *M1 = nullptr, *Mt = nullptr, *mu_x = nullptr;
    // collapsed chain: covariance of the (virtual) current map and the accumulated operator, with their ping-pong partners
    float *cov_x2 = nullptr, *acc = nullptr, *acc2 = nullptr;
    // per-tile row statistics of the rotated pastiche, written by the forward rotation GEMM's epilogue (GemmArgs::rowstat)
    float *rs_a = nullptr, *rs_b = nullptr;
    int rs_parts = 0;
    float* bias = nullptr;   // [n_seg, C]: mu_s - (R T) mu_t, the folded centring of the apply GEMM (linear_loop)
    // cdf / sort with one rotation sequence for the whole batch: the style side of EVERY iteration is prepared before the
    // loop (one batched GEMM, one min / max or one sort launch for all of them) — `ys` then holds iters rotated copies
    bool hoist = false;
    float *smn_all = nullptr, *smx_all = nullptr;   // cdf: style min / max per (iteration, style segment, channel)
    unsigned* shist_all = nullptr;                  // cdf: style histograms over that range, [iters, Ss, C, 256]
    int* sort_flags = nullptr;                      // sort: scratch of the one style sort

    // Ss: segments of the ROTATED style the matcher sees (= n_seg when every segment has its own rotations: own_rot)
    void layout(Bump& b, int mode, long n, long ns, int C, int n_seg, int Ss, int iters, int fused, bool own_rot) {
        const size_t xs = (size_t)n_seg * C * n, cc = (size_t)C * C;
        rs_parts = gemm_rowstat_parts(n);
        const size_t rs_floats = (size_t)n_seg * rs_parts * C;
        if (mode == MODE_CDF || mode == MODE_SORT) {
            y = b.take<float>(xs);
            const size_t ys_one = (size_t)Ss * C * ns;
            // (budget: the hoisted copies stay a fraction of the feature maps' own footprint; 2048^2 relu1_1 is not
            // launch-bound and keeps rotating its style per iteration.  sort: the one-launch sort is the LDS-resident one)
            hoist = !fused && !own_rot && Ss == 1 && iters > 1 && ys_one * iters * sizeof(float) <= kHoistBytes &&
                    (mode == MODE_CDF || ns <= 16384);
            ys = b.take<float>(hoist ? ys_one * iters : ys_one);
            if (hoist && mode == MODE_CDF) {
                smn_all = b.take<float>((size_t)iters * Ss * C);
                smx_all = b.take<float>((size_t)iters * Ss * C);
                shist_all = b.take<unsigned>((size_t)iters * Ss * C * kBins);
            }
            if (hoist && mode == MODE_SORT) sort_flags = b.take<int>((size_t)iters * Ss * C);
            if (!fused && rs_parts) {  // cdf: the joint range of histmatch.py:52-53; sort: the rank kernel's bucket range
                rs_a = b.take<float>(rs_floats);
                rs_b = b.take<float>(rs_floats);
            }
            if (fused) {
                y2 = b.take<float>(xs);
                P = b.take<float>((size_t)(iters > 1 ? iters - 1 : 1) * cc);
            }
            // the sort scratch depends on BOTH column lengths: pastiche columns longer than one LDS take the global radix
            mode_ws_bytes = mode == MODE_CDF ? optex_cdf_ws_bytes(C, n_seg) : optex_sort_match_ws_bytes(n, ns, C, n_seg, Ss);
            mode_ws = b.take<char>(mode_ws_bytes);
            return;
        }
        const size_t NS = (size_t)(iters > 0 ? iters : 1) * Ss;
        const int NP = chol_np(C);
        const size_t pp = (size_t)NP * NP;
        const size_t nb = NS > (size_t)n_seg ? NS : (size_t)n_seg;
        if (fused == 0) {         // default: rotate, then apply + rotate back as one GEMM
            y = b.take<float>((size_t)n_seg * C * padded_ld(n));
            M1 = b.take<float>((size_t)n_seg * cc);
            if (rs_parts) rs_a = b.take<float>(rs_floats);
            bias = b.take<float>((size_t)n_seg * C);
        } else if (fused == 2) {  // literal three-GEMM sequence
            y = b.take<float>(xs);
            y2 = b.take<float>(xs);
        } else {
            y2 = b.take<float>(xs);  // the affine map cannot run in place either
            M1 = b.take<float>((size_t)n_seg * cc);
            Mt = b.take<float>((size_t)n_seg * cc);
            mu_x = b.take<float>((size_t)n_seg * C);
            if (fused == 3) {
                cov_x2 = b.take<float>((size_t)n_seg * cc);
                acc = b.take<float>((size_t)n_seg * cc);
                acc2 = b.take<float>((size_t)n_seg * cc);
            }
        }
        const int smax = n_seg > Ss ? n_seg : Ss;
        const long nmax = n > ns ? n : ns;
        stats_ws_bytes = optex_linear_stats_ws_bytes(nmax, C, smax);
        stats_ws = b.take<char>(stats_ws_bytes);
        mu_t = b.take<float>((size_t)n_seg * C);
        cov_t = b.take<float>((size_t)n_seg * cc);
        mu_s = b.take<float>((size_t)Ss * C);
        cov_s = b.take<float>((size_t)Ss * cc);
        tmp_s = b.take<float>(NS * cc);
        cov_sr = b.take<float>(NS * cc);
        mu_sr = b.take<float>(NS * C);
        At = b.take<float>((size_t)n_seg * cc);
        if (mode == MODE_CHOL) {
            Us = b.take<float>(NS * pp);
            Ls = b.take<float>(NS * pp);
            Ut = b.take<float>((size_t)n_seg * pp);
            Lt = b.take<float>((size_t)n_seg * pp);
        } else {
            ns_buf = b.take<float>(ns_ws_floats(C, (int)nb));
            if (mode == MODE_PCA) Ys = b.take<float>(NS * cc);
            Yt = b.take<float>((size_t)n_seg * cc);
            Zt = b.take<float>((size_t)n_seg * cc);
            G1 = b.take<float>((size_t)n_seg * cc);
            G = b.take<float>((size_t)n_seg * cc);
        }
    }
};

int copy_async(float* dst, const float* src, size_t count, hipStream_t st) { return device_copy(dst, src, count, st); }

// out[seg][m] = badd[m] - sum_k At[seg][k][m] * bsub[seg][k]: the apply GEMM's  At^T (y - bsub) + badd  is  At^T y + out  — the
// centring as a per-row bias (accumulated in double: one rounding of the folded term).  grid (n_seg, ceil(C / 64)), block 256:
// 64 columns x 4 interleaved k ranges, eight loads in flight per thread (one dependent L2 round trip per k made it 65 us)
__global__ __launch_bounds__(256) void affine_bias_kernel(const float* __restrict__ At, long at_ss, const float* __restrict__ bsub,
                                                          const float* __restrict__ badd, long badd_ss, int C,
                                                          float* __restrict__ out) {
    const int seg = blockIdx.x, ml = threadIdx.x & 63, kp = threadIdx.x >> 6, m = blockIdx.y * 64 + ml;
    const float* A = At + (size_t)seg * at_ss;
    const float* mu = bsub + (size_t)seg * C;
    double acc = 0.0;
    if (m < C) {
        int k = kp;
        for (; k + 28 < C; k += 32) {
            float av[8], mv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                av[u] = A[(size_t)(k + 4 * u) * C + m];
                mv[u] = mu[k + 4 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc += (double)av[u] * (double)mv[u];
        }
        for (; k < C; k += 4) acc += (double)A[(size_t)k * C + m] * (double)mu[k];
    }
    __shared__ double part[4][64];
    part[kp][ml] = acc;
    __syncthreads();
    if (kp == 0 && m < C)
        out[(size_t)seg * C + m] = (float)((double)badd[(size_t)seg * badd_ss + m] - ((part[0][ml] + part[1][ml]) + (part[2][ml] + part[3][ml])));
}

// feature-map GEMM with every option spelled out (the C ABI entry point with the loop's fixed layouts)
int fgemm(const float* At, long at_ss, const float* B, float* O, int C, long n, int n_seg, const float* bsub, const float* badd,
          long badd_ss, const float* content, float strength, void* stream, long ldb = 0) {
    const long xs = (long)C * n;
    if (ldb == 0) ldb = n;
    return optex_gemm_tn(At, C, at_ss, B, ldb, (long)C * ldb, OPTEX_CHANNEL_MAJOR, O, n, xs, OPTEX_CHANNEL_MAJOR, C, C, n, n_seg, bsub,
                         C, badd, badd_ss, content, strength, 0u, stream);
}


// The PCA projection folded into the first rotation and the unprojection into the last (optex.py:110,120):
//   (feat @ E) @ R_0 == feat @ (E R_0),   (m @ R_l^T) @ E^T == m @ (E R_l)^T
// — iteration 0 reads the un-projected features with the C_full x C matrix E R_0, the last iteration writes un-projected
// features with (E R_l)^T: two feature-map GEMMs of a (pass, layer) disappear.  Same products, another association.
struct Fold {
    const float* xin = nullptr;   // [n_seg, Cf, n] un-projected input of iteration 0
    const float* ER0 = nullptr;   // [Cf, C] = E R_0
    int Cf = 0;
    float* xout = nullptr;        // [n_seg, Cf, n] un-projected output of the last iteration, or NULL: the loop ends in k-space
    const float* G = nullptr;     // [C, Cf] = (E R_last)^T
};

// optex.py:170  rotated = feature @ rotation  on the loop's layouts, with the per-row statistics of the result taken in the
// GEMM's epilogue when the launch takes the hot-loop kernel (rowstat 1 = min / max, 2 = sums; *took says whether it did).
// K: channels of the input map (0 = C; the folded projection reads C_full channels)
int rotate_with_stats(const float* R, long r_ss, const float* x, float* y, int C, long n, int n_seg, int rowstat, float* rs_a,
                      float* rs_b, bool* took, hipStream_t st, long ldy = 0, int K = 0) {
    if (ldy == 0) ldy = n;
    if (K == 0) K = C;
    GemmArgs a;
    a.At = R; a.lda = C; a.at_ss = r_ss;
    a.B = x; a.ldb = n; a.b_ss = (long)K * n;
    a.O = y; a.ldo = ldy; a.o_ss = (long)C * ldy;
    a.M = C; a.K = K; a.n = n; a.n_seg = n_seg;
    a.bsub = nullptr; a.bsub_ss = 0; a.badd = nullptr; a.badd_ss = 0; a.content = nullptr; a.strength = 0.f;
    a.epi = 0; a.alpha = 1.f; a.alpha_seg = nullptr; a.diag = 0.f; a.sym = 0; a.prof_cls = KC_GEMM;
    a.rowstat = 0; a.rs_a = nullptr; a.rs_b = nullptr;
    *took = rowstat != 0 && rs_a != nullptr && gemm_rowstat_supported(a);
    if (*took) {
        a.rowstat = rowstat;
        a.rs_a = rs_a;
        a.rs_b = rs_b;
    }
    return gemm_tn_launch(a, OPTEX_CHANNEL_MAJOR, OPTEX_CHANNEL_MAJOR, st);
}

// Transfer operator of one iteration, transposed (At[k][m] = T[m][k], what the apply GEMM takes), for every pastiche
// segment: cov_t [n_seg, C, C] (eps included) against the rotated style statistics of iteration `it`.
// with_rt (chol / pca, prepare_style's hoisted products): the result is T^T R_it^T = (R_it T)^T, what the loop's apply GEMM takes,
// written to `out` by the same single launch — the style-side factor arrives multiplied by R_it^T already (w.tmp_s).
int transfer_operators(int mode, LoopWs& w, const float* cov_t, int C, int n_seg, int Ss, int it, hipStream_t st,
                       bool with_rt = false, float* out = nullptr) {
    const size_t cc = (size_t)C * C;
    const int NP = chol_np(C);
    const size_t pp = (size_t)NP * NP;
    int rc;
    if (!out) out = w.At;
    const float* pre = w.tmp_s + (size_t)it * Ss * cc;
    if (mode == MODE_CHOL) {
        // histmatch.py:24-27  T = L_s L_t^-1  ->  T^T = (L_t^-1)^T L_s^T = Linv_t^T @ U_s
        if ((rc = launch_chol_inv(cov_t, (long)cc, C, n_seg, w.Ut, w.Lt, st))) return rc;
        if (with_rt)
            return small_gemm(w.Lt, NP, (long)pp, pre, C, Ss > 1 ? (long)cc : 0, out, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f,
                              st, false);
        const float* Us = w.Us + (size_t)it * Ss * pp;
        return small_gemm(w.Lt, NP, (long)pp, Us, NP, Ss > 1 ? (long)pp : 0, out, C, (long)cc, C, n_seg, false, 1.f, nullptr,
                          0.f, st, false);
    }
    float *Y, *Z;
    if ((rc = ns_sqrt(cov_t, (long)cc, C, n_seg, kEps, w.ns_buf, &Y, &Z, st))) return rc;
    if (mode == MODE_PCA) {
        // histmatch.py:29-34  T = Q_s Q_t^-1  ->  T^T = Q_t^-1 Q_s   (both symmetric)
        const float* Ys = with_rt ? pre : w.Ys + (size_t)it * Ss * cc;
        return small_gemm(Z, C, (long)cc, Ys, C, Ss > 1 ? (long)cc : 0, out, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false);
    }
    // histmatch.py:36-42  T = Q_t^-1 (Q_t S_s Q_t)^1/2 Q_t^-1   (symmetric: T^T = T)
    if ((rc = copy_async(w.Yt, Y, (size_t)n_seg * cc, st))) return rc;
    if ((rc = copy_async(w.Zt, Z, (size_t)n_seg * cc, st))) return rc;
    const float* Cs = w.cov_sr + (size_t)it * Ss * cc;
    if ((rc = small_gemm(Cs, C, Ss > 1 ? (long)cc : 0, w.Yt, C, (long)cc, w.G1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
        return rc;                                                                    // S_s Q_t
    if ((rc = small_gemm(w.Yt, C, (long)cc, w.G1, C, (long)cc, w.G, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, true)))
        return rc;                                                                    // Q_t S_s Q_t
    if ((rc = ns_sqrt(w.G, (long)cc, C, n_seg, kEps * kEps, w.ns_buf, &Y, &Z, st))) return rc;     // its square root
    if ((rc = small_gemm(Y, C, (long)cc, w.Zt, C, (long)cc, w.G1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
        return rc;                                                                    // (.)^1/2 Q_t^-1
    return small_gemm(w.Zt, C, (long)cc, w.G1, C, (long)cc, w.At, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false);
}

// style statistics once, rotated for every iteration and every ROTATION SET g (G sets: one per style segment when the
// rotations are shared, one per pastiche segment when every segment has its own, r_ss != 0):
//   cov_sr[it][g] = R_g,it^T cov(S_s(g)) R_g,it + eps I,   mu_sr[it][g] = R_g,it^T mu_s(g),   s(g) = g or 0
// and the style-side factor of the mode (chol: U_s = L_s^T; pca: Q_s)
// Rt32 != NULL (the default association of the loop, chol / pca): additionally tmp_s[it][g] = U_s R_it^T (Q_s R_it^T) — the
// loop's apply GEMM takes (R T)^T = T^T R^T = Linv_t^T (U_s R^T): with the style-side product taken here for all iterations in
// one batched launch, an iteration has ONE C x C product between its factorization and its apply GEMM instead of two
// (493 launches of a single-texture call, 1.8 ms of a 64-texture chol step).  The same product in another association.
int prepare_style(int mode, LoopWs& w, const float* style, long ns, int Ss, int G, int C, const float* R32, long r_ss, int iters,
                  hipStream_t st, void* stream, const float* Rt32 = nullptr) {
    const size_t cc = (size_t)C * C;
    int rc;
    if ((rc = optex_linear_stats(style, ns, (long)C * ns, ns, C, Ss, 0, 0.f, w.mu_s, w.cov_s, w.stats_ws, w.stats_ws_bytes, stream)))
        return rc;
    for (int g = 0; g < G; g++) {
        const float* Rg = R32 + (size_t)g * r_ss;
        const float* cov = w.cov_s + (size_t)(Ss > 1 ? g : 0) * cc;
        // tmp[it][g] = cov_s @ R_it   (cov_s symmetric);   cov_sr[it][g] = R_it^T @ tmp + eps I
        if ((rc = small_gemm(cov, C, 0, Rg, C, (long)cc, w.tmp_s + (size_t)g * cc, C, (long)(cc * G), C, iters, false, 1.f, nullptr,
                             0.f, st, false)))
            return rc;
        if ((rc = small_gemm(Rg, C, (long)cc, w.tmp_s + (size_t)g * cc, C, (long)(cc * G), w.cov_sr + (size_t)g * cc, C,
                             (long)(cc * G), C, iters, true, 1.f, nullptr, kEps, st, true)))
            return rc;
    }
    hipLaunchKernelGGL(rot_mean_kernel, dim3(iters * G), dim3(256), 0, st, R32, r_ss, w.mu_s, Ss > 1 ? 1 : 0, C, G, w.mu_sr);
    if ((rc = check_launch("rot_mean_kernel"))) return rc;
    const int NP = chol_np(C);
    const size_t pp = (size_t)NP * NP;
    if (mode == MODE_CHOL) {
        if ((rc = launch_chol_inv(w.cov_sr, (long)cc, C, iters * G, w.Us, w.Ls, st))) return rc;
        if (Rt32)
            for (int g = 0; g < G; g++)
                if ((rc = small_gemm_nn_ld(w.Us + (size_t)g * pp, NP, (long)(pp * G), Rt32 + (size_t)g * r_ss, (long)cc,
                                           w.tmp_s + (size_t)g * cc, (long)(cc * G), C, iters, st)))
                    return rc;
        return OPTEX_OK;
    }
    if (mode == MODE_PCA) {
        float *Y, *Z;
        if ((rc = ns_sqrt(w.cov_sr, (long)cc, C, iters * G, kEps, w.ns_buf, &Y, &Z, st))) return rc;
        if ((rc = copy_async(w.Ys, Y, (size_t)iters * G * cc, st))) return rc;
        if (Rt32)
            for (int g = 0; g < G; g++)
                if ((rc = small_gemm_nn_ld(w.Ys + (size_t)g * cc, C, (long)(cc * G), Rt32 + (size_t)g * r_ss, (long)cc,
                                           w.tmp_s + (size_t)g * cc, (long)(cc * G), C, iters, st)))
                    return rc;
    }
    return OPTEX_OK;
}

// G: rotation sets per iteration (prepare_style); r_ss: elements between the rotation sets of two segments (0 = shared)
int linear_loop(int mode, float* x, long n, int n_seg, const float* style, long ns, int Ss, int G, int C, const float* R32,
                const float* Rt32, long r_ss, int iters, const float* content, float strength, int fused, LoopWs& w, void* stream,
                const Fold* fold = nullptr) {
    hipStream_t st = as_stream(stream);
    const size_t cc = (size_t)C * C;
    const long xs = (long)C * n;
    int rc;
    const bool with_rt = fused == 0 && (mode == MODE_CHOL || mode == MODE_PCA);
    if ((rc = prepare_style(mode, w, style, ns, Ss, G, C, R32, r_ss, iters, st, stream, with_rt ? Rt32 : nullptr))) return rc;
    if (fused == 3) {
        // The whole chain in C x C algebra (no content blend): every step is x' = M_i (x - mean) + mu_s with
        // M_i = R_i T_i R_i^T, and the statistics the next step needs follow analytically,
        //   cov(x') = M_i cov(x) M_i^T,   mean(x') = mu_s,
        // so the feature map is read once for its statistics and once by the single GEMM that applies M_k ... M_1.
        // small_gemm(L, B) = L^T @ B, small_gemm_nn(A, B) = A @ B.
        float *cov = w.cov_t, *cov2 = w.cov_x2, *acc = w.acc, *acc2 = w.acc2;
        if ((rc = optex_linear_stats(x, n, xs, n, C, n_seg, 0, 0.f, w.mu_x, cov, w.stats_ws, w.stats_ws_bytes, stream))) return rc;
        for (int it = 0; it < iters; it++) {
            const float* R = R32 + (size_t)it * cc;
            const float* Rt = Rt32 + (size_t)it * cc;
            if ((rc = small_gemm(cov, C, (long)cc, R, C, r_ss, w.M1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
                return rc;                                                                   // cov(x) R
            if ((rc = small_gemm(R, C, r_ss, w.M1, C, (long)cc, w.Mt, C, (long)cc, C, n_seg, true, 1.f, nullptr, kEps, st, true)))
                return rc;                                                                   // R^T cov(x) R + eps I
            if ((rc = transfer_operators(mode, w, w.Mt, C, n_seg, G, it, st))) return rc;   // At = T^T
            if ((rc = small_gemm(w.At, C, (long)cc, Rt, C, r_ss, w.M1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
                return rc;                                                                   // T R^T
            if ((rc = small_gemm(w.M1, C, (long)cc, Rt, C, r_ss, w.Mt, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
                return rc;                                                                   // R T^T R^T = M^T
            if (it + 1 < iters) {
                if ((rc = small_gemm(w.Mt, C, (long)cc, cov, C, (long)cc, w.M1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
                    return rc;                                                               // M cov(x)
                if ((rc = small_gemm_nn(w.M1, (long)cc, w.Mt, (long)cc, cov2, C, n_seg, 1.f, nullptr, 0.f, nullptr, 0, st)))
                    return rc;                                                               // M cov(x) M^T
                float* t = cov; cov = cov2; cov2 = t;
            }
            if (it == 0) {
                if ((rc = copy_async(acc, w.Mt, (size_t)n_seg * cc, st))) return rc;
            } else {
                // (M_i ... M_1)^T = (M_{i-1} ... M_1)^T M_i^T
                if ((rc = small_gemm_nn(acc, (long)cc, w.Mt, (long)cc, acc2, C, n_seg, 1.f, nullptr, 0.f, nullptr, 0, st))) return rc;
                float* t = acc; acc = acc2; acc2 = t;
            }
        }
        if ((rc = fgemm(acc, (long)cc, x, w.y2, C, n, n_seg, w.mu_x, w.mu_s, Ss > 1 ? C : 0, nullptr, 0.f, stream))) return rc;
        return copy_async(x, w.y2, (size_t)n_seg * xs, st);
    }
    float* cur = x;       // fused path: the affine map cannot run in place, x and y2 take turns
    float* nxt = w.y2;
    for (int it = 0; it < iters; it++) {
        const float* R = R32 + (size_t)it * cc;
        const float* Rt = Rt32 + (size_t)it * cc;
        const float* mu_sr = w.mu_sr + (size_t)it * G * C;
        if (fused == 0) {
            // optex.py:170  rotated_pastiche = pastiche_feature @ rotation   (+ the row sums for the means, in the epilogue)
            bool sums = false;
            const long ldy = padded_ld(n);
            if (fold && it == 0)   // the PCA projection rides in the first rotation: E R_0 on the un-projected map
                rc = rotate_with_stats(fold->ER0, 0, fold->xin, w.y, C, n, n_seg, 2, w.rs_a, nullptr, &sums, st, ldy, fold->Cf);
            else
                rc = rotate_with_stats(R, r_ss, x, w.y, C, n, n_seg, 2, w.rs_a, nullptr, &sums, st, ldy);
            if (rc) return rc;
            // histmatch.py:16-18  mu_t, cov_t = hist_t hist_t^T / N + eps I   (statistics of the ROTATED map, like the reference)
            if ((rc = linear_stats_parts(w.y, ldy, (long)C * ldy, n, C, n_seg, 0, kEps, w.mu_t, w.cov_t, w.stats_ws,
                                         w.stats_ws_bytes, sums ? w.rs_a : nullptr, w.rs_parts, stream)))
                return rc;
            // histmatch.py:27/34/42,44 + optex.py:175, 115-117:  (T hist_t + mu_sr) @ R^T  evaluated as ONE feature-map GEMM
            //   x = (R T)(y - mu_t) + R mu_sr,   R mu_sr = R R^T mu_s = mu_s,   (R T)^T = T^T R^T = At @ Rt
            // — the same product in another association (a C x C GEMM instead of a second C x n one); the content blend
            // rides in the epilogue as before.  chol / pca: T^T R^T comes out of transfer_operators' one product (prepare_style).
            if (with_rt) {
                if ((rc = transfer_operators(mode, w, w.cov_t, C, n_seg, G, it, st, true, w.M1))) return rc;
            } else {
                if ((rc = transfer_operators(mode, w, w.cov_t, C, n_seg, G, it, st))) return rc;   // At = T^T
                if ((rc = small_gemm_nn(w.At, (long)cc, Rt, r_ss, w.M1, C, n_seg, 1.f, nullptr, 0.f, nullptr, 0, st))) return rc;
            }
            // Large maps (round 6): the centring rides as a per-row bias, x = (R T) y + (mu_s - (R T) mu_t), so that the apply GEMM is
            // the R-stationary kernel's plain loop with a bias in its epilogue — centring inside the k-loop (a second operand
            // stream and a subtraction per k-step) made it 685 us against 521 at [64, 256, 16384], 8.5 ms of a 64-texture chol
            // step.  The same affine map, the subtraction taken once in double instead of per element.  Small maps (one texture's
            // deep layers) keep the centring in the GEMM: there a launch costs more than the subtractions.
            if ((double)n_seg * (double)n >= 262144.0) {
                hipLaunchKernelGGL(affine_bias_kernel, dim3(n_seg, (C + 63) / 64), dim3(256), 0, st, w.M1, (long)cc, w.mu_t, w.mu_s,
                                   (long)(Ss > 1 ? C : 0), C, w.bias);
                if ((rc = check_launch("affine_bias_kernel"))) return rc;
                if ((rc = fgemm(w.M1, (long)cc, w.y, x, C, n, n_seg, nullptr, w.bias, C, content, strength, stream, ldy))) return rc;
            } else if ((rc = fgemm(w.M1, (long)cc, w.y, x, C, n, n_seg, w.mu_t, w.mu_s, Ss > 1 ? C : 0, content, strength, stream, ldy)))
                return rc;
        } else if (fused == 2) {
            // the literal sequence, three feature-map GEMMs (kept for tests and comparisons)
            // optex.py:170  rotated_pastiche = pastiche_feature @ rotation
            if ((rc = fgemm(R, r_ss, x, w.y, C, n, n_seg, nullptr, nullptr, 0, nullptr, 0.f, stream))) return rc;
            // histmatch.py:16-18  mu_t, cov_t = hist_t hist_t^T / N + eps I
            if ((rc = optex_linear_stats(w.y, n, xs, n, C, n_seg, 0, kEps, w.mu_t, w.cov_t, w.stats_ws, w.stats_ws_bytes, stream)))
                return rc;
            if ((rc = transfer_operators(mode, w, w.cov_t, C, n_seg, G, it, st))) return rc;
            // histmatch.py:27/34/42,44  matched = T @ hist_t + mu_s
            if ((rc = fgemm(w.At, (long)cc, w.y, w.y2, C, n, n_seg, w.mu_t, mu_sr, G > 1 ? C : 0, nullptr, 0.f, stream))) return rc;
            // optex.py:175 + 115-117  pastiche = matched @ rotation.T ; content blend
            if ((rc = fgemm(Rt, r_ss, w.y2, x, C, n, n_seg, nullptr, nullptr, 0, content, strength, stream))) return rc;
        } else {
            // Single affine step in un-rotated space, the labelled fast path:
            //   cov(x R) = R^T cov(x) R,   x' = M (x - mu_x) + mu_s   with   M = R T R^T
            // one covariance and ONE feature-map GEMM per iteration instead of three.  small_gemm(L, B) = L^T @ B.
            if ((rc = optex_linear_stats(cur, n, xs, n, C, n_seg, 0, 0.f, w.mu_x, w.cov_t, w.stats_ws, w.stats_ws_bytes, stream)))
                return rc;
            if ((rc = small_gemm(w.cov_t, C, (long)cc, R, C, r_ss, w.M1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
                return rc;                                                                   // cov(x) R
            if ((rc = small_gemm(R, C, r_ss, w.M1, C, (long)cc, w.Mt, C, (long)cc, C, n_seg, true, 1.f, nullptr, kEps, st, true)))
                return rc;                                                                   // R^T cov(x) R + eps I
            if ((rc = transfer_operators(mode, w, w.Mt, C, n_seg, G, it, st))) return rc;   // At = T^T
            if ((rc = small_gemm(w.At, C, (long)cc, Rt, C, r_ss, w.M1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
                return rc;                                                                   // (T^T)^T R^T = T R^T
            if ((rc = small_gemm(w.M1, C, (long)cc, Rt, C, r_ss, w.Mt, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
                return rc;                                                                   // (T R^T)^T R^T = R T^T R^T = M^T
            // x' = M (x - mu_x) + mu_s  (R mu_sr = R R^T mu_s = the un-rotated style mean), content blend in the epilogue
            if ((rc = fgemm(w.Mt, (long)cc, cur, nxt, C, n, n_seg, w.mu_x, w.mu_s, Ss > 1 ? C : 0, content, strength, stream)))
                return rc;
            float* t = cur; cur = nxt; nxt = t;
        }
    }
    if (fused == 1 && cur != x) return copy_async(x, cur, (size_t)n_seg * xs, st);
    return OPTEX_OK;
}

}  // namespace

extern "C" size_t optex_ot_loop_ws_bytes(int mode, long n, long ns, int C, int n_seg, int src_n_seg, int iters,
                                         int fuse_rotations, long r_seg_stride) {
    LoopWs w;
    Bump b(nullptr);
    w.layout(b, mode, n, ns, C, n_seg, r_seg_stride != 0 ? n_seg : src_n_seg, iters,
             (mode < MODE_CHOL && fuse_rotations == 2) ? 0 : fuse_rotations, r_seg_stride != 0);
    return b.off;
}

static int ot_loop_impl(int mode, float* x, long n, int n_seg, const float* style, long ns, int src_n_seg, int C,
                        const float* R32, const float* Rt32, long r_seg_stride, int iters, const float* content,
                        float strength, int fuse_rotations, void* ws, size_t ws_bytes, void* stream, const Fold* fold) {
    if (!x || !style || !R32 || !Rt32 || !ws || n <= 0 || ns <= 0 || C < 2 || n_seg <= 0 || iters < 0) {
        set_error("optex_ot_loop: bad argument (n=%ld ns=%ld C=%d n_seg=%d iters=%d)", n, ns, C, n_seg, iters);
        return OPTEX_E_ARG;
    }
    if (mode < MODE_CDF || mode > MODE_SYM) {
        set_error("optex_ot_loop: mode %d (0 = cdf, 1 = sort, 2 = chol, 3 = pca, 4 = sym)", mode);
        return OPTEX_E_ARG;
    }
    if (src_n_seg != 1 && src_n_seg != n_seg) {
        set_error("optex_ot_loop: style has %d segments, expected 1 or %d", src_n_seg, n_seg);
        return OPTEX_E_ARG;
    }
    const bool linear = mode >= MODE_CHOL;
    if (fuse_rotations < 0 || fuse_rotations > 3) {
        set_error("optex_ot_loop: fuse_rotations = %d (0, 1, 2 or 3)", fuse_rotations);
        return OPTEX_E_ARG;
    }
    if (fuse_rotations == 3 && (!linear || content)) {
        set_error("optex_ot_loop: fuse_rotations = 3 (collapsed chain) is for the linear modes without a content blend");
        return OPTEX_E_ARG;
    }
    if (!linear && fuse_rotations == 2) fuse_rotations = 0;
    if (r_seg_stride != 0 && !linear && fuse_rotations) {
        set_error("optex_ot_loop: cdf / sort with per-segment rotations (r_seg_stride != 0) run with fuse_rotations = 0 only");
        return OPTEX_E_UNSUPPORTED;
    }
    if (r_seg_stride != 0 && r_seg_stride < (long)iters * C * C) {
        set_error("optex_ot_loop: r_seg_stride %ld is smaller than one segment's rotations (%ld)", r_seg_stride, (long)iters * C * C);
        return OPTEX_E_ARG;
    }
    if (fuse_rotations && content && !linear) {
        set_error("optex_ot_loop: fuse_rotations needs the un-rotated pastiche between iterations for the content blend");
        return OPTEX_E_ARG;
    }
    if (linear && C > 512) {
        set_error("optex_ot_loop: the linear modes support C <= 512 channels (got %d)", C);
        return OPTEX_E_UNSUPPORTED;
    }
    if (int rc = check_ws("optex_ot_loop", ws, ws_bytes,
                          optex_ot_loop_ws_bytes(mode, n, ns, C, n_seg, src_n_seg, iters, fuse_rotations, r_seg_stride)))
        return rc;
    if (iters == 0) return OPTEX_OK;
    LoopWs w;
    Bump bump(ws);
    // With its own rotations every segment sees its own rotated copy of the style: the matchers then run one source
    // segment per target segment (nothing on the style side is shared any more, optex.py:168-171 run per image).
    const int rs_seg = r_seg_stride != 0 ? n_seg : src_n_seg;
    w.layout(bump, mode, n, ns, C, n_seg, rs_seg, iters, fuse_rotations, r_seg_stride != 0);
    if (fold && (fuse_rotations != 0 || r_seg_stride != 0)) {
        set_error("optex_ot_loop: the folded PCA projection runs with fuse_rotations = 0 and one rotation sequence per batch");
        return OPTEX_E_UNSUPPORTED;
    }
    if (linear)
        return linear_loop(mode, x, n, n_seg, style, ns, src_n_seg, rs_seg, C, R32, Rt32, r_seg_stride, iters, content, strength,
                           fuse_rotations, w, stream, fold);

    hipStream_t st = as_stream(stream);
    const long xs = (long)C * n, ss = (long)C * ns;
    if (fuse_rotations) {
        // Re-association of optex.py:175 + :170 of the next iteration:  (m @ R_i^T) @ R_{i+1} == m @ (R_i^T R_{i+1}).
        // One feature-map GEMM per iteration instead of two; the C x C products P_i = R_i^T R_{i+1} cost nothing.
        // Same fp32 arithmetic contract (k-ordered fma chains), different association: results agree with the literal
        // loop to fp32 round-off per step (tests/test_gpu_parity.py), not bit for bit.
        int rc;
        if (iters > 1 &&
            (rc = optex_gemm_tn(R32, C, (long)C * C, R32 + (size_t)C * C, C, (long)C * C, OPTEX_CHANNEL_MAJOR, w.P, C,
                                (long)C * C, OPTEX_CHANNEL_MAJOR, C, C, C, iters - 1, nullptr, 0, nullptr, 0, nullptr, 0.f, 0u,
                                stream)))
            return rc;
        float* cur = w.y;
        float* nxt = w.y2;
        if ((rc = optex_gemm_tn(R32, C, 0, x, n, xs, OPTEX_CHANNEL_MAJOR, cur, n, xs, OPTEX_CHANNEL_MAJOR, C, C, n, n_seg,
                                nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream)))
            return rc;
        for (int it = 0; it < iters; it++) {
            const float* R = R32 + (size_t)it * C * C;
            if ((rc = optex_gemm_tn(R, C, 0, style, ns, ss, OPTEX_CHANNEL_MAJOR, w.ys, ns, ss, OPTEX_CHANNEL_MAJOR, C, C,
                                    ns, src_n_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream)))
                return rc;
            if (mode == MODE_CDF)
                rc = cdf_match_impl(cur, n, xs, n, w.ys, ns, ss, ns, src_n_seg, C, n_seg, cur, n, xs, w.mode_ws, nullptr, st);
            else
                rc = sort_match_impl(cur, n, xs, n, w.ys, ns, ss, ns, src_n_seg, C, n_seg, cur, n, xs, w.mode_ws, st);
            if (rc) return rc;
            if (it + 1 < iters) {  // straight into the next iteration's rotated frame
                if ((rc = optex_gemm_tn(w.P + (size_t)it * C * C, C, 0, cur, n, xs, OPTEX_CHANNEL_MAJOR, nxt, n, xs,
                                        OPTEX_CHANNEL_MAJOR, C, C, n, n_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream)))
                    return rc;
                float* t = cur; cur = nxt; nxt = t;
            } else {               // optex.py:175 of the last iteration
                if ((rc = optex_gemm_tn(Rt32 + (size_t)it * C * C, C, 0, cur, n, xs, OPTEX_CHANNEL_MAJOR, x, n, xs,
                                        OPTEX_CHANNEL_MAJOR, C, C, n, n_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream)))
                    return rc;
            }
        }
        return OPTEX_OK;
    }
    int rc;
    if (w.hoist) {
        // optex.py:171 for every iteration at once: rotated_style[it] = style_feature @ rotation[it] — one batched GEMM (the
        // "segments" of the launch are the iterations: one matrix each, all reading the same style map) — and what the
        // matcher needs of each: its per-channel min / max (cdf, histmatch.py:52-53) or its sorted columns (sort), one
        // launch for all iterations.  Same arithmetic as inside the loop, 2-4 launches per CALL instead of per iteration.
        if ((rc = optex_gemm_tn(R32, C, (long)C * C, style, ns, 0, OPTEX_CHANNEL_MAJOR, w.ys, ns, ss, OPTEX_CHANNEL_MAJOR, C,
                                C, ns, iters, nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream)))
            return rc;
        if (mode == MODE_CDF) {
            // ... and its histogram over its own range: the joint range of histmatch.py:52-53 IS the style's range for every
            // pastiche column the style's range contains, and then histc(style, 256, lo, hi) is the same for all textures
            if ((rc = col_minmax_launch(w.ys, ns, ss, ns, C, iters * rs_seg, w.smn_all, w.smx_all, st))) return rc;
            rc = col_hist_launch(w.ys, ns, ss, ns, C, iters * rs_seg, w.smn_all, w.smx_all, w.shist_all, st);
        } else
            rc = sort_columns_inplace(w.ys, ns, iters * rs_seg * C, w.sort_flags, st);
        if (rc) return rc;
    }
    if (mode == MODE_CDF && (rc = cdf_ws_clear(w.mode_ws, C, n_seg, st))) return rc;  // the pipeline leaves its counters clear
    for (int it = 0; it < iters; it++) {
        const float* R = R32 + (size_t)it * C * C;
        const float* Rt = Rt32 + (size_t)it * C * C;
        // optex.py:170  rotated_pastiche = pastiche_feature @ rotation   (+ per-channel min / max in the epilogue: it saves the
        // cdf matcher histmatch.py:52-53's pass over the rotated map, and the sort matcher its in-kernel range reduction)
        bool mm = false;
        if (fold && it == 0)   // the PCA projection rides in the first rotation: E R_0 on the un-projected map
            rc = rotate_with_stats(fold->ER0, 0, fold->xin, w.y, C, n, n_seg, 1, w.rs_a, w.rs_b, &mm, st, 0, fold->Cf);
        else
            rc = rotate_with_stats(R, r_seg_stride, x, w.y, C, n, n_seg, 1, w.rs_a, w.rs_b, &mm, st);
        if (rc) return rc;
        // optex.py:171  rotated_style = style_feature @ rotation   (one copy per rotation set; hoisted: done above)
        const float* ys = w.hoist ? w.ys + (size_t)it * rs_seg * ss : w.ys;
        if (!w.hoist &&
            (rc = optex_gemm_tn(R, C, r_seg_stride, style, ns, src_n_seg > 1 ? ss : 0, OPTEX_CHANNEL_MAJOR, w.ys, ns, ss,
                                OPTEX_CHANNEL_MAJOR, C, C, ns, rs_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream)))
            return rc;
        // optex.py:173  hist_match(rotated_pastiche, rotated_style), in place
        if (mode == MODE_CDF)
            rc = cdf_match_parts_impl(w.y, n, xs, n, ys, ns, ss, ns, rs_seg, C, n_seg, w.y, n, xs, w.mode_ws, nullptr,
                                      mm ? w.rs_a : nullptr, mm ? w.rs_b : nullptr, w.rs_parts, st,
                                      w.hoist ? w.smn_all + (size_t)it * rs_seg * C : nullptr,
                                      w.hoist ? w.smx_all + (size_t)it * rs_seg * C : nullptr, true,
                                      w.hoist ? w.shist_all + (size_t)it * rs_seg * C * kBins : nullptr);
        else
            rc = sort_match_impl(w.y, n, xs, n, ys, ns, ss, ns, rs_seg, C, n_seg, w.y, n, xs, w.mode_ws, st,
                                 mm ? w.rs_a : nullptr, mm ? w.rs_b : nullptr, w.rs_parts, w.hoist ? ys : nullptr);
        if (rc) return rc;
        // optex.py:175 + 115-117  pastiche = matched @ rotation.T ; content blend
        if (fold && fold->xout && it == iters - 1)   // ... and the PCA unprojection rides in the last one: (E R_l)^T
            rc = optex_gemm_tn(fold->G, fold->Cf, 0, w.y, n, xs, OPTEX_CHANNEL_MAJOR, fold->xout, n, (long)fold->Cf * n,
                               OPTEX_CHANNEL_MAJOR, fold->Cf, C, n, n_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream);
        else
            rc = optex_gemm_tn(Rt, C, r_seg_stride, w.y, n, xs, OPTEX_CHANNEL_MAJOR, x, n, xs, OPTEX_CHANNEL_MAJOR, C, C, n, n_seg,
                               nullptr, 0, nullptr, 0, content, strength, 0u, stream);
        if (rc) return rc;
    }
    return OPTEX_OK;
}

extern "C" int optex_ot_loop(int mode, float* x, long n, int n_seg, const float* style, long ns, int src_n_seg, int C,
                             const float* R32, const float* Rt32, long r_seg_stride, int iters, const float* content,
                             float strength, int fuse_rotations, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    CallScope call_scope(flags);
    return ot_loop_impl(mode, x, n, n_seg, style, ns, src_n_seg, C, R32, Rt32, r_seg_stride, iters, content, strength,
                        fuse_rotations, ws, ws_bytes, stream, nullptr);
}

// ---- the loop between the PCA projection and unprojection of optex.py:110,120 -----------------------------------------------
static size_t pca_extra_bytes(long n, int C, int C_full, int n_seg) {
    return align_up((size_t)n_seg * C * n * sizeof(float), 256) + 2 * align_up((size_t)C_full * C * sizeof(float), 256);
}

extern "C" size_t optex_ot_loop_pca_ws_bytes(int mode, long n, long ns, int C, int C_full, int n_seg, int src_n_seg, int iters) {
    return align_up(optex_ot_loop_ws_bytes(mode, n, ns, C, n_seg, src_n_seg, iters, 0, 0), 256) + pca_extra_bytes(n, C, C_full, n_seg);
}

extern "C" int optex_ot_loop_pca(int mode, float* x_full, int C_full, const float* eig, const float* eig_t, long n, int n_seg,
                                 const float* style, long ns, int src_n_seg, int C, const float* R32, const float* Rt32, int iters,
                                 const float* content, float strength, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    CallScope call_scope(flags);
    if (!x_full || !eig || !eig_t || !style || (iters > 0 && (!R32 || !Rt32)) || !ws || n <= 0 || ns <= 0 || C < 2 || C_full < C ||
        n_seg <= 0 || iters < 0) {
        set_error("optex_ot_loop_pca: bad argument (n=%ld ns=%ld C=%d C_full=%d n_seg=%d iters=%d)", n, ns, C, C_full, n_seg, iters);
        return OPTEX_E_ARG;
    }
    // everything ot_loop_impl would refuse is refused HERE, before the E R_0 / (E R_l)^T products go onto the stream (ADVICE r4:
    // a rejected call used to return after work had been enqueued)
    if (mode < MODE_CDF || mode > MODE_SYM) {
        set_error("optex_ot_loop_pca: mode %d (0 = cdf, 1 = sort, 2 = chol, 3 = pca, 4 = sym)", mode);
        return OPTEX_E_ARG;
    }
    if (src_n_seg != 1 && src_n_seg != n_seg) {
        set_error("optex_ot_loop_pca: style has %d segments, expected 1 or %d", src_n_seg, n_seg);
        return OPTEX_E_ARG;
    }
    if (mode >= MODE_CHOL && C > 512) {
        set_error("optex_ot_loop_pca: the linear modes support C <= 512 channels (got %d)", C);
        return OPTEX_E_UNSUPPORTED;
    }
    if (int rc = check_ws("optex_ot_loop_pca", ws, ws_bytes, optex_ot_loop_pca_ws_bytes(mode, n, ns, C, C_full, n_seg, src_n_seg, iters)))
        return rc;
    const size_t base = align_up(optex_ot_loop_ws_bytes(mode, n, ns, C, n_seg, src_n_seg, iters, 0, 0), 256);
    char* p = static_cast<char*>(ws) + base;
    float* xk = reinterpret_cast<float*>(p);                                  // the loop's state between iterations, k-space
    p += align_up((size_t)n_seg * C * n * sizeof(float), 256);
    float* ER0 = reinterpret_cast<float*>(p);
    p += align_up((size_t)C_full * C * sizeof(float), 256);
    float* G = reinterpret_cast<float*>(p);
    const long xfs = (long)C_full * n, xs = (long)C * n;
    int rc;
    if (iters == 0) {
        // optex.py:110 then :120 with nothing in between: x_full <- (x_full @ E) @ E^T, the projector onto the kept subspace
        if ((rc = optex_gemm_tn(eig, C, 0, x_full, n, xfs, OPTEX_CHANNEL_MAJOR, xk, n, xs, OPTEX_CHANNEL_MAJOR, C, C_full, n, n_seg,
                                nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream)))
            return rc;
        return optex_gemm_tn(eig_t, C_full, 0, xk, n, xs, OPTEX_CHANNEL_MAJOR, x_full, n, xfs, OPTEX_CHANNEL_MAJOR, C_full, C, n, n_seg,
                             nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream);
    }
    // E R_0 as the [C_full, C] matrix of the first rotation: (E R_0)[c][m] = sum_j E[c][j] R_0[j][m] — the transposing GEMM reads
    // E as a "pixel-major" map of C_full pixels and stores the result pixel-major, i.e. row-major [C_full, C]
    if ((rc = optex_gemm_tn(R32, C, 0, eig, C, 0, OPTEX_PIXEL_MAJOR, ER0, C, 0, OPTEX_PIXEL_MAJOR, C, C, C_full, 1, nullptr, 0, nullptr,
                            0, nullptr, 0.f, 0u, stream)))
        return rc;
    Fold f;
    f.xin = x_full;
    f.ER0 = ER0;
    f.Cf = C_full;
    // the unprojection folds where the last step is a plain rotation back: cdf / sort without a content blend (the blend
    // works on the un-rotated k-space map; the linear modes' last GEMM already carries the transfer operator and the means)
    const bool fold_out = mode < MODE_CHOL && content == nullptr;
    if (fold_out) {
        // (E R_l)^T as the [C, C_full] matrix of the last rotation back: the same product stored channel-major
        if ((rc = optex_gemm_tn(R32 + (size_t)(iters - 1) * C * C, C, 0, eig, C, 0, OPTEX_PIXEL_MAJOR, G, C_full, 0,
                                OPTEX_CHANNEL_MAJOR, C, C, C_full, 1, nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream)))
            return rc;
        f.xout = x_full;
        f.G = G;
    }
    if ((rc = ot_loop_impl(mode, xk, n, n_seg, style, ns, src_n_seg, C, R32, Rt32, 0, iters, content, strength, 0, ws, base, stream, &f)))
        return rc;
    if (fold_out) return OPTEX_OK;
    // optex.py:120  pastiche_feature @ eigvecs.T
    return optex_gemm_tn(eig_t, C_full, 0, xk, n, xs, OPTEX_CHANNEL_MAJOR, x_full, n, xfs, OPTEX_CHANNEL_MAJOR, C_full, C, n, n_seg,
                         nullptr, 0, nullptr, 0, nullptr, 0.f, 0u, stream);
}
