/*
 * optex_oracle.c — CPU restatement of the reference's sliced-OT hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker* for the HIP kernels in optimaltextures_amd/csrc.  It is never linked into,
 * imported by or called from the product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it (oracle/oracle.py is its ctypes wrapper).
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py against golden
 * vectors captured by importing the reference itself (tests/golden/gen_golden.py, run in the build
 * container against /root/reference with torch 2.10 CPU / numpy 2.2.6 / scipy 1.15.3).
 *
 * What each function restates (file:line in the reference repository):
 *   orc_mt_seed / orc_normals     numpy legacy RandomState (MT19937 + polar gauss) that drives
 *                                 scipy.stats.special_ortho_group.rvs at optex.py:149 (third-party, unpinned by
 *                                 the reference; numpy 2.2.6 `_legacy.pyx`/`legacy-distributions.c` algorithm)
 *   orc_random_rotation           optex.py:142-149 -> scipy 1.15.3 special_ortho_group_gen.rvs (Householder chain)
 *   orc_gemm_tn                   the three matmuls of optex.py:170,171,175 as a k-ordered fp32 fma chain
 *                                 (this is bit-for-bit what v_mfma_f32_32x32x2_f32 computes)
 *   orc_histc / orc_linspace      torch.histc / torch.linspace CPU semantics used at histmatch.py:57-59
 *   orc_interp                    histmatch.py:72-92 (NOT numpy.interp: right-anchored, 3-stage non-finite fallback)
 *   orc_cdf_match                 histmatch.py:49-69
 *   orc_linear_match              histmatch.py:16-44 (chol | pca | sym)
 *   orc_sort_columns / orc_sort_match   no reference item: exact 1-D OT by stable sort; this file
 *                                 IS the specification of that mode
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp -ffp-contract=off -o liboptex_oracle.so optex_oracle.c -lm
 * (-ffp-contract=off matters: the reference evaluates a*b+c with two roundings.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_BINS 256

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * numpy legacy RandomState: MT19937 + polar-method gaussian with a one-value cache.
 * State layout (uint32[627]): key[624], pos, has_gauss, then the cached gauss as two uint32 (a double).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t key[624];
    int32_t pos;
    int32_t has_gauss;
    double gauss;
} orc_rng;

int orc_rng_size(void) { return (int)sizeof(orc_rng); }

void orc_mt_seed(orc_rng* st, uint32_t seed) {
    for (int pos = 0; pos < 624; pos++) {
        st->key[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
    }
    st->pos = 624;
    st->has_gauss = 0;
    st->gauss = 0.0;
}

static void mt_gen(orc_rng* st) {
    uint32_t* mt = st->key;
    int i;
    uint32_t y;
    for (i = 0; i < 624 - 397; i++) {
        y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
        mt[i] = mt[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    for (; i < 623; i++) {
        y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
        mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    st->pos = 0;
}

static uint32_t mt_next(orc_rng* st) {
    if (st->pos == 624) mt_gen(st);
    uint32_t y = st->key[st->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double mt_double(orc_rng* st) {
    int32_t a = (int32_t)(mt_next(st) >> 5), b = (int32_t)(mt_next(st) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static double legacy_gauss(orc_rng* st) {
    if (st->has_gauss) {
        const double tmp = st->gauss;
        st->has_gauss = 0;
        st->gauss = 0.0;
        return tmp;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_double(st) - 1.0;
        x2 = 2.0 * mt_double(st) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    st->gauss = f * x1;
    st->has_gauss = 1;
    return f * x2;
}

/* RandomState.normal(size=n) with loc 0, scale 1 */
void orc_normals(orc_rng* st, double* out, long n) {
    for (long i = 0; i < n; i++) out[i] = 0.0 + 1.0 * legacy_gauss(st);
}

/* number of normals one SO(N) draw consumes: sum_{n=0}^{N-2} (N-n) */
long orc_rotation_normals(int N) { return (long)N * (N + 1) / 2 - 1; }

/* ------------------------------------------------------------------------------------------------
 * scipy special_ortho_group.rvs(N): Householder chain, fp64.  `normals` holds the N(N+1)/2-1 draws in the
 * order scipy makes them; it is modified in place (scipy modifies x in place too).  H is [N,N] row-major.
 * ---------------------------------------------------------------------------------------------- */
void orc_random_rotation(double* normals, int N, double* H) {
    double* D = (double*)malloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) H[(size_t)i * N + j] = (i == j) ? 1.0 : 0.0;
    /* phase 1 (sequential, O(N^2)): normalise every Householder vector in place */
    double* x = normals;
    for (int n = 0; n < N - 1; n++) {
        const int len = N - n;
        double norm2 = 0.0;
        for (int j = 0; j < len; j++) norm2 += x[j] * x[j];
        const double x0 = x[0];
        D[n] = (x0 != 0.0) ? ((x0 > 0.0) ? 1.0 : -1.0) : 1.0;
        x[0] += D[n] * sqrt(norm2);
        const double den = sqrt((norm2 - x0 * x0 + x[0] * x[0]) / 2.);
        for (int j = 0; j < len; j++) x[j] /= den;
        x += len;
    }
    /* phase 2 (O(N^3)): the reflections act on each row of H independently, in order n = 0..N-2 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        const double* v = normals;
        for (int n = 0; n < N - 1; n++) {
            const int len = N - n;
            double* h = H + (size_t)i * N + n;
            double s = 0.0;
            for (int j = 0; j < len; j++) s += h[j] * v[j];
            for (int j = 0; j < len; j++) h[j] -= s * v[j];
            v += len;
        }
    }
    double prod = 1.0;
    for (int i = 0; i < N - 1; i++) prod *= D[i];
    D[N - 1] = (((N - 1) & 1) ? -1.0 : 1.0) * prod;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) H[(size_t)i * N + j] *= D[i];
    free(D);
}

/* ------------------------------------------------------------------------------------------------
 * OUT[m][n] = sum_k At[k][m] * (B[k][n] - bsub[k]) (+ badd[m]), k ascending, ONE fp32 fma per product.
 * At is [K, M] (ld = lda), B is [K, N] (ld = ldb), OUT is [M, N] (ld = ldo).  bsub/badd may be NULL.
 * With At = R this is optex.py:170 in channel-major form: Y[c][n] = sum_k X[k][n] R[k][c].
 * ---------------------------------------------------------------------------------------------- */
void orc_gemm_tn(const float* At, long lda, const float* B, long ldb, float* OUT, long ldo, int M, int K, long N,
                 const float* bsub, const float* badd) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; m++) {
        float* o = OUT + (size_t)m * ldo;
        for (long n = 0; n < N; n++) o[n] = 0.0f;
        for (int k = 0; k < K; k++) {
            const float a = At[(size_t)k * lda + m];
            const float* b = B + (size_t)k * ldb;
            if (bsub) {
                const float s = bsub[k];
                for (long n = 0; n < N; n++) o[n] = fmaf(a, b[n] - s, o[n]);
            } else {
                for (long n = 0; n < N; n++) o[n] = fmaf(a, b[n], o[n]);
            }
        }
        if (badd) {
            const float s = badd[m];
            for (long n = 0; n < N; n++) o[n] = o[n] + s;
        }
    }
}

/* caller epilogue optex.py:115-117:  feat += strength * (content - feat)   (separate mul and add) */
void orc_content_blend(float* feat, const float* content, float strength, long n) {
    for (long i = 0; i < n; i++) {
        const float d = content[i] - feat[i];
        const float sd = strength * d;
        feat[i] = feat[i] + sd;
    }
}

/* ------------------------------------------------------------------------------------------------
 * torch.histc(x, 256, lo, hi) on CPU: linear-interpolation binning, x == hi -> last bin, lo == hi widens by 1.
 * Counts are returned as fp32 like torch does (exact integers below 2^24).
 * ---------------------------------------------------------------------------------------------- */
void orc_histc(const float* x, long n, float lo, float hi, float* hist) {
    for (int i = 0; i < ORC_BINS; i++) hist[i] = 0.0f;
    if (lo == hi) {
        lo -= 1.0f;
        hi += 1.0f;
    }
    const float range = hi - lo;
    for (long i = 0; i < n; i++) {
        const float v = x[i];
        if (!(v >= lo && v <= hi)) continue;
        const float scaled = (v - lo) * (float)ORC_BINS;
        long pos = (long)(scaled / range);
        if (pos == ORC_BINS) pos = ORC_BINS - 1;
        hist[pos] += 1.0f;
    }
}

/* torch.linspace(lo, hi, 257) fp32 on CPU: step = (hi-lo)/256; first half lo + step*i, second half
 * hi - step*(256-i), each evaluated as ONE fma (the vectorised ATen kernel contracts them). */
void orc_linspace257(float lo, float hi, float* e) {
    const float step = (hi - lo) / (float)ORC_BINS;
    for (int i = 0; i <= ORC_BINS; i++)
        e[i] = (i < (ORC_BINS + 1) / 2) ? fmaf(step, (float)i, lo) : fmaf(-step, (float)(ORC_BINS - i), hi);
}

/* torch.searchsorted(xp, v) (right=False): first index with xp[idx] >= v, same bisection as ATen */
static long lower_bound_f(const float* xp, long n, float v) {
    long start = 0, end = n;
    while (start < end) {
        const long mid = start + ((end - start) >> 1);
        if (!(xp[mid] >= v)) start = mid + 1;
        else end = mid;
    }
    return start;
}

/* histmatch.py:72-92.  idx == n (x above every knot or NaN) would index out of range in the reference and raise;
 * here it is clamped to n-1 so that the function is total (documented deviation, never hit on the hot path). */
void orc_interp(const float* x, long nx, const float* xp, const float* fp, long n, float* out) {
    for (long i = 0; i < nx; i++) {
        const float v = x[i];
        long idx = lower_bound_f(xp, n, v);
        if (idx > n - 1) idx = n - 1;
        const long nxt = (idx + 1 > n - 1) ? n - 1 : idx + 1;
        const float slope = (fp[nxt] - fp[idx]) / (xp[nxt] - xp[idx]);
        float f = slope * (v - xp[idx]);
        f = f + fp[idx];
        if (!isfinite(f)) {
            float f2 = slope * (v - xp[nxt]);
            f2 = f2 + fp[nxt];
            f = isfinite(f2) ? f2 : fp[idx];
        }
        out[i] = f;
    }
}

/* one channel of histmatch.py:49-69; dbg (may be NULL) receives lo,hi,hist_t[256],hist_s[256],edges[256],remapped[256] */
static void cdf_channel(const float* t, long nt, const float* s, long ns, float* out, float* dbg) {
    float lo = t[0], hi = t[0];
    for (long i = 0; i < nt; i++) {
        lo = t[i] < lo ? t[i] : lo;
        hi = t[i] > hi ? t[i] : hi;
    }
    for (long i = 0; i < ns; i++) {
        lo = s[i] < lo ? s[i] : lo;
        hi = s[i] > hi ? s[i] : hi;
    }
    float ht[ORC_BINS], hs[ORC_BINS], e[ORC_BINS + 1], tc[ORC_BINS], sc[ORC_BINS], rm[ORC_BINS];
    orc_histc(t, nt, lo, hi, ht);
    orc_histc(s, ns, lo, hi, hs);
    orc_linspace257(lo, hi, e);
    const float* edges = e + 1;
    float acc = 0.0f;
    for (int i = 0; i < ORC_BINS; i++) {
        acc += ht[i];
        tc[i] = acc;
    }
    const float tl = tc[ORC_BINS - 1];
    for (int i = 0; i < ORC_BINS; i++) tc[i] = tc[i] / tl;
    acc = 0.0f;
    for (int i = 0; i < ORC_BINS; i++) {
        acc += hs[i];
        sc[i] = acc;
    }
    const float sl = sc[ORC_BINS - 1];
    for (int i = 0; i < ORC_BINS; i++) sc[i] = sc[i] / sl;
    orc_interp(tc, ORC_BINS, sc, edges, ORC_BINS, rm);
    orc_interp(t, nt, edges, rm, ORC_BINS, out);
    if (dbg) {
        dbg[0] = lo;
        dbg[1] = hi;
        memcpy(dbg + 2, ht, sizeof(ht));
        memcpy(dbg + 2 + ORC_BINS, hs, sizeof(hs));
        memcpy(dbg + 2 + 2 * ORC_BINS, edges, sizeof(float) * ORC_BINS);
        memcpy(dbg + 2 + 3 * ORC_BINS, rm, sizeof(rm));
    }
}

/* ------------------------------------------------------------------------------------------------
 * histmatch.py:49-69 with any `bins` (the reference's third argument; every caller in the reference leaves it at 256).
 * Same statements as above with the bin count as a variable: torch.histc(x, bins, lo, hi), torch.linspace(lo, hi, bins + 1)
 * (step = (hi - lo) / bins, first half fma(step, i, lo), second half fma(-step, bins - i, hi), halfway = (bins + 1) / 2),
 * cumsum / last, the two interp calls.  Pinned by tests/golden/cdf_match_bins.npz (reference outputs for several bins).
 * ---------------------------------------------------------------------------------------------- */
static void cdf_channel_bins(const float* t, long nt, const float* s, long ns, int bins, float* out, float* work) {
    float lo = t[0], hi = t[0];
    for (long i = 0; i < nt; i++) {
        lo = t[i] < lo ? t[i] : lo;
        hi = t[i] > hi ? t[i] : hi;
    }
    for (long i = 0; i < ns; i++) {
        lo = s[i] < lo ? s[i] : lo;
        hi = s[i] > hi ? s[i] : hi;
    }
    float *ht = work, *hs = ht + bins, *e = hs + bins, *tc = e + bins + 1, *sc = tc + bins, *rm = sc + bins;
    for (int which = 0; which < 2; which++) {
        float* h = which ? hs : ht;
        const float* x = which ? s : t;
        const long n = which ? ns : nt;
        float l = lo, u = hi;
        for (int i = 0; i < bins; i++) h[i] = 0.0f;
        if (l == u) {
            l -= 1.0f;
            u += 1.0f;
        }
        const float range = u - l;
        for (long i = 0; i < n; i++) {
            const float v = x[i];
            if (!(v >= l && v <= u)) continue;
            const float scaled = (v - l) * (float)bins;
            long pos = (long)(scaled / range);
            if (pos == bins) pos = bins - 1;
            h[pos] += 1.0f;
        }
    }
    const float step = (hi - lo) / (float)bins;
    for (int i = 0; i <= bins; i++) e[i] = (i < (bins + 1) / 2) ? fmaf(step, (float)i, lo) : fmaf(-step, (float)(bins - i), hi);
    const float* edges = e + 1;
    float acc = 0.0f;
    for (int i = 0; i < bins; i++) {
        acc += ht[i];
        tc[i] = acc;
    }
    const float tl = tc[bins - 1];
    for (int i = 0; i < bins; i++) tc[i] = tc[i] / tl;
    acc = 0.0f;
    for (int i = 0; i < bins; i++) {
        acc += hs[i];
        sc[i] = acc;
    }
    const float sl = sc[bins - 1];
    for (int i = 0; i < bins; i++) sc[i] = sc[i] / sl;
    orc_interp(tc, bins, sc, edges, bins, rm);
    orc_interp(t, nt, edges, rm, bins, out);
}

void orc_cdf_match_bins(const float* t, long ldt, long nt, const float* s, long lds, long ns, int C, int bins, float* out) {
#pragma omp parallel for schedule(dynamic)
    for (int c = 0; c < C; c++) {
        float* work = (float*)malloc(sizeof(float) * (size_t)(6 * bins + 1));
        cdf_channel_bins(t + (size_t)c * ldt, nt, s + (size_t)c * lds, ns, bins, out + (size_t)c * ldt, work);
        free(work);
    }
}

/* target [C, nt] (row stride ldt), source [C, ns] (row stride lds), out [C, nt] (row stride ldt).
 * dbg: NULL or [C, 2 + 4*256]. */
void orc_cdf_match(const float* target, long ldt, long nt, const float* source, long lds, long ns, int C, float* out,
                   float* dbg) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < C; c++)
        cdf_channel(target + (size_t)c * ldt, nt, source + (size_t)c * lds, ns, out + (size_t)c * ldt,
                    dbg ? dbg + (size_t)c * (2 + 4 * ORC_BINS) : NULL);
}

/* ------------------------------------------------------------------------------------------------
 * Sort mode (no reference item; this is its specification).
 * Key order: IEEE-754 totalOrder on the fp32 bit pattern (-0 < +0, NaNs at the ends by sign), ties keep index
 * order (stable).  orc_sort_columns returns sorted keys and the permutation; orc_sort_match maps the i-th
 * smallest target value of a column to the source order statistic j = floor((2i+1)*ns / (2*nt)).
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t f2key(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static void stable_sort_pairs(uint32_t* key, uint32_t* idx, long n, uint32_t* tk, uint32_t* ti) {
    /* LSD radix, 4 x 8 bit: stable by construction */
    for (int pass = 0; pass < 4; pass++) {
        long cnt[257];
        memset(cnt, 0, sizeof(cnt));
        const int sh = pass * 8;
        for (long i = 0; i < n; i++) cnt[((key[i] >> sh) & 0xff) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (long i = 0; i < n; i++) {
            const long p = cnt[(key[i] >> sh) & 0xff]++;
            tk[p] = key[i];
            ti[p] = idx[i];
        }
        uint32_t* sw = key;
        key = tk;
        tk = sw;
        sw = idx;
        idx = ti;
        ti = sw;
    }
}

/* keys [C, n] (row stride ld) -> out_keys [C, n], out_idx [C, n] (contiguous) */
void orc_sort_columns(const float* keys, long ld, long n, int C, float* out_keys, uint32_t* out_idx) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < C; c++) {
        uint32_t* k = (uint32_t*)malloc(sizeof(uint32_t) * 4 * (size_t)n);
        uint32_t *i0 = k + n, *tk = k + 2 * n, *ti = k + 3 * n;
        const float* src = keys + (size_t)c * ld;
        for (long i = 0; i < n; i++) {
            k[i] = f2key(src[i]);
            i0[i] = (uint32_t)i;
        }
        stable_sort_pairs(k, i0, n, tk, ti);
        for (long i = 0; i < n; i++) {
            out_idx[(size_t)c * n + i] = i0[i];
            out_keys[(size_t)c * n + i] = src[i0[i]];
        }
        free(k);
    }
}

void orc_sort_match(const float* target, long ldt, long nt, const float* source, long lds, long ns, int C, float* out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < C; c++) {
        float* tk = (float*)malloc(sizeof(float) * (size_t)(nt + ns));
        uint32_t* ti = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(nt + ns));
        float* sk = tk + nt;
        uint32_t* si = ti + nt;
        orc_sort_columns(target + (size_t)c * ldt, nt, nt, 1, tk, ti);
        orc_sort_columns(source + (size_t)c * lds, ns, ns, 1, sk, si);
        float* o = out + (size_t)c * ldt;
        for (long i = 0; i < nt; i++) {
            const long j = (long)(((2 * (uint64_t)i + 1) * (uint64_t)ns) / (2 * (uint64_t)nt));
            o[ti[i]] = sk[j];
        }
        free(tk);
        free(ti);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Linear modes, histmatch.py:16-44.  Statistics in fp64 accumulators rounded to fp32 where the reference
 * holds fp32 tensors; the small C x C factorizations in fp64 (the reference uses fp32 LAPACK: parity is by
 * tolerance).  mode: 0 chol, 1 pca, 2 sym.
 * target [C, nb*np_] where each of the nb batch items has its own mean (histmatch.py:16) but the covariance is
 * pooled (histmatch.py:18); source likewise with sb items of sp pixels.  mu_s is added per batch item when
 * sb == nb, else item 0's (broadcast, histmatch.py:44).
 * ---------------------------------------------------------------------------------------------- */
static void chol_lower(const double* A, double* L, int n) {
    memset(L, 0, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        d = sqrt(d);
        L[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / d;
        }
    }
}

static void inv_lower(const double* L, double* X, int n) {
    memset(X, 0, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; j++) {
        X[(size_t)j * n + j] = 1.0 / L[(size_t)j * n + j];
        for (int i = j + 1; i < n; i++) {
            double s = 0.0;
            for (int k = j; k < i; k++) s -= L[(size_t)i * n + k] * X[(size_t)k * n + j];
            X[(size_t)i * n + j] = s / L[(size_t)i * n + i];
        }
    }
}

static void matmul_d(const double* A, const double* B, double* C, int n) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int k = 0; k < n; k++) s += A[(size_t)i * n + k] * B[(size_t)k * n + j];
            C[(size_t)i * n + j] = s;
        }
}

/* cyclic Jacobi: A symmetric -> eigenvalues w, eigenvectors V (columns) */
static void jacobi_eigh(const double* Ain, double* w, double* V, int n) {
    double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
    memcpy(A, Ain, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[(size_t)i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                if (i != j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
                else diag += A[(size_t)i * n + j] * A[(size_t)i * n + j];
            }
        if (off <= 1e-30 * diag) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[(size_t)p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[(size_t)i * n + i];
    free(A);
}

/* out = V diag(f(w)) V^T with f = sqrt (pw=0.5) or 1/sqrt (pw=-0.5) */
static void sym_fun(const double* S, double* out, int n, double pw) {
    double* w = (double*)malloc(sizeof(double) * (size_t)n);
    double* V = (double*)malloc(sizeof(double) * (size_t)n * n);
    jacobi_eigh(S, w, V, n);
    for (int i = 0; i < n; i++) w[i] = pow(w[i], pw);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int k = 0; k < n; k++) s += V[(size_t)i * n + k] * w[k] * V[(size_t)j * n + k];
            out[(size_t)i * n + j] = s;
        }
    free(w);
    free(V);
}

/* means per (channel, item) + pooled covariance (+eps on the diagonal) of x [C, nb*np_] */
static void stats(const float* x, long ld, int C, int nb, long np_, float eps, float* mu /*[C,nb]*/, double* cov) {
    const long N = (long)nb * np_;
    float* h = (float*)malloc(sizeof(float) * (size_t)C * N);
    for (int c = 0; c < C; c++)
        for (int b = 0; b < nb; b++) {
            double s = 0.0;
            const float* p = x + (size_t)c * ld + (size_t)b * np_;
            for (long i = 0; i < np_; i++) s += p[i];
            const float m = (float)(s / (double)np_);
            mu[(size_t)c * nb + b] = m;
            for (long i = 0; i < np_; i++) h[(size_t)c * N + (size_t)b * np_ + i] = p[i] - m;
        }
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < C; i++)
        for (int j = i; j < C; j++) {
            double s = 0.0;
            for (long k = 0; k < N; k++) s += (double)h[(size_t)i * N + k] * (double)h[(size_t)j * N + k];
            const double v = (double)(float)(s / (double)N) + ((i == j) ? (double)eps : 0.0);
            cov[(size_t)i * C + j] = v;
            cov[(size_t)j * C + i] = v;
        }
    free(h);
}

/* T (fp64 [C,C]) such that matched = T @ hist_t, histmatch.py:24-42 */
void orc_transfer_operator(const double* cov_t, const double* cov_s, int C, int mode, double* T) {
    const size_t nn = (size_t)C * C;
    double* a = (double*)malloc(sizeof(double) * nn * 4);
    double *b = a + nn, *c = a + 2 * nn, *d = a + 3 * nn;
    if (mode == 0) {
        chol_lower(cov_t, a, C);
        chol_lower(cov_s, b, C);
        inv_lower(a, c, C);
        matmul_d(b, c, T, C);
    } else if (mode == 1) {
        sym_fun(cov_s, a, C, 0.5);
        sym_fun(cov_t, b, C, -0.5);
        matmul_d(a, b, T, C);
    } else {
        sym_fun(cov_t, a, C, 0.5);  /* Qt */
        sym_fun(cov_t, b, C, -0.5); /* Qt^-1 */
        matmul_d(a, cov_s, c, C);
        matmul_d(c, a, d, C);    /* Qt Cs Qt */
        sym_fun(d, c, C, 0.5);   /* (Qt Cs Qt)^1/2 */
        matmul_d(b, c, d, C);
        matmul_d(d, b, T, C);
    }
    free(a);
}

void orc_linear_match(const float* target, long ldt, int nb, long np_, const float* source, long lds, int sb, long sp,
                      int C, int mode, float eps, float* out, float* T_out /* NULL or [C,C] fp32 */) {
    const long N = (long)nb * np_;
    float* mu_t = (float*)malloc(sizeof(float) * (size_t)C * nb);
    float* mu_s = (float*)malloc(sizeof(float) * (size_t)C * sb);
    double* cov_t = (double*)malloc(sizeof(double) * (size_t)C * C * 3);
    double *cov_s = cov_t + (size_t)C * C, *T = cov_t + 2 * (size_t)C * C;
    stats(target, ldt, C, nb, np_, eps, mu_t, cov_t);
    stats(source, lds, C, sb, sp, eps, mu_s, cov_s);
    orc_transfer_operator(cov_t, cov_s, C, mode, T);
    float* Tt = (float*)malloc(sizeof(float) * (size_t)C * C); /* Tt[k][m] = T[m][k] */
    for (int m = 0; m < C; m++)
        for (int k = 0; k < C; k++) {
            Tt[(size_t)k * C + m] = (float)T[(size_t)m * C + k];
            if (T_out) T_out[(size_t)m * C + k] = (float)T[(size_t)m * C + k];
        }
    float* bs = (float*)malloc(sizeof(float) * (size_t)C * 2);
    float* ba = bs + C;
    for (int b = 0; b < nb; b++) {
        for (int c = 0; c < C; c++) {
            bs[c] = mu_t[(size_t)c * nb + b];
            ba[c] = mu_s[(size_t)c * sb + ((sb == nb) ? b : 0)];
        }
        orc_gemm_tn(Tt, C, target + (size_t)b * np_, ldt, out + (size_t)b * np_, ldt, C, C, np_, bs, ba);
    }
    (void)N;
    free(bs);
    free(Tt);
    free(cov_t);
    free(mu_s);
    free(mu_t);
}
