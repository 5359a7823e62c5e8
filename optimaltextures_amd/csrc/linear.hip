// linear.hip — K4: statistics of the linear modes (histmatch.py:16-22): per-(segment, channel) spatial means and the
// centred covariance  cov = hist @ hist.T / N + eps * I  as a split-K Gram GEMM on the fp32 matrix cores.
//
// The Gram matrix is symmetric: only the upper-triangular 64x64 tile pairs are computed (C(C+64)/2 * n * 2 flop
// instead of 2*n*C^2) and mirrored in the finalize kernel.  Centring happens while staging tiles into LDS, exactly
// like the reference centres before the GEMM (no E[x^2] - mu^2 cancellation).  Partials of the K (= pixel) split
// are reduced in a fixed order, so the result is deterministic.

#include "optex_common.h"

namespace optex {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GT = 64;        // small Gram output tile (GT x GT), 4 waves as 2 x 2 of one 32x32 MFMA tile each (C <= 64)
constexpr int GT2 = 128;      // large tile: 4 waves as 2 x 2 of 64 x 64 (2 x 2 MFMA tiles): one LDS read per MFMA instead of two
constexpr int GK = 32;        // pixels staged per chunk
constexpr int GSTR = GK + 1;  // odd LDS row stride: conflict-free column reads
constexpr int G_MAX_SPLITS = 64;   // pixel splits per segment ... (gram_split_cap: up to 512 where a partial is small against its pixels)

__global__ __launch_bounds__(256) void col_mean_kernel(const float* __restrict__ x, long ld, long seg_stride, long n,
                                                       int C, float* __restrict__ mu, int vec) {
    const int col = blockIdx.x, seg = col / C, c = col % C;
    const float* p = x + (size_t)seg * seg_stride + (size_t)c * ld;
    double s = 0.0;
    if (vec) {
        const long nv = n / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (long i = threadIdx.x; i < nv; i += blockDim.x) {
            const float4 v = p4[i];
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        }
        for (long i = nv * 4 + threadIdx.x; i < n; i += blockDim.x) s += (double)p[i];
    } else {
        for (long i = threadIdx.x; i < n; i += blockDim.x) s += (double)p[i];
    }
    s = wave_sum(s);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) mu[col] = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / (double)n);
}

// The same for FEW long columns (one texture: 64 channels x 262144 pixels on 64 workgroups took 87 us): `P` workgroups per column
// over pixel ranges of `chunk` (a multiple of 4) pixels, partial sums in double, finished in a fixed order by
// mean_from_dparts_kernel.  grid (columns, P)
__global__ __launch_bounds__(256) void col_sum_parts_kernel(const float* __restrict__ x, long ld, long seg_stride, long n, int C,
                                                            long chunk, double* __restrict__ part, int vec) {
    const int col = blockIdx.x, seg = col / C, c = col % C;
    const long beg = (long)blockIdx.y * chunk, end = beg + chunk < n ? beg + chunk : n;
    const float* p = x + (size_t)seg * seg_stride + (size_t)c * ld;
    double s = 0.0;
    if (vec) {
        const long v0 = beg / 4, v1 = end / 4;   // beg is a multiple of 4; the last range ends at n
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (long i = v0 + threadIdx.x; i < v1; i += blockDim.x) {
            const float4 v = p4[i];
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        }
        for (long i = v1 * 4 + threadIdx.x; i < end; i += blockDim.x) s += (double)p[i];
    } else {
        for (long i = beg + threadIdx.x; i < end; i += blockDim.x) s += (double)p[i];
    }
    s = wave_sum(s);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + col] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void mean_from_dparts_kernel(const double* __restrict__ part, int P, int ncols, long n,
                                                               float* __restrict__ mu) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols) return;
    double s = 0.0;
    for (int q = 0; q < P; q++) s += part[(size_t)q * ncols + col];
    mu[col] = (float)(s / (double)n);
}

// spatial mean of every (segment, channel) from the per-tile row sums the forward rotation GEMM left behind
// (GemmArgs::rowstat = 2: part [n_seg][parts][C]), summed in double in a fixed order
__global__ __launch_bounds__(256) void mean_from_parts_kernel(const float* __restrict__ psum, int parts, int C, int ncols, long n,
                                                              float* __restrict__ mu) {
    // 64 columns per block (coalesced along the channel), 4 threads per column over interleaved partials
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (col < ncols) {
        const float* a = psum + (size_t)(col / C) * parts * C + (col % C);
#pragma unroll 8
        for (int p = g; p < parts; p += 4) s += (double)a[(size_t)p * C];
    }
    __shared__ double sh[4][64];
    sh[g][cl] = s;
    __syncthreads();
    if (g == 0 && col < ncols) mu[col] = (float)(((sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl])) / (double)n);
}

// grid = (tile pairs, splits, n_seg).  part[seg][split][C][C] receives the (ti, tj) tile of this pixel range.
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ x, long ld, long seg_stride, long n, int C,
                                                   const float* __restrict__ mu, long chunk, int tiles,
                                                   float* __restrict__ part, int vec) {
    // decode the upper-triangular pair index
    int pi = blockIdx.x, ti = 0;
    while (pi >= tiles - ti) {
        pi -= tiles - ti;
        ti++;
    }
    const int tj = ti + pi;
    const int seg = blockIdx.z, split = blockIdx.y;
    const float* xs = x + (size_t)seg * seg_stride;
    const float* mus = mu + (size_t)seg * C;
    const long p_beg = (long)split * chunk, p_end = (p_beg + chunk < n) ? p_beg + chunk : n;

    __shared__ float Xi[2][GT * GSTR], Xj[2][GT * GSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1, l31 = lane & 31, h = lane >> 5;

    float4 ri[2], rj[2];
    auto load_global = [&](long p0) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int idx = tid + q * 256;
            const int row = idx / (GK / 4), px = (idx % (GK / 4)) * 4;
            const long pp = p0 + px;
#pragma unroll
            for (int which = 0; which < 2; which++) {
                const int ch = (which ? tj : ti) * GT + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ch < C) {
                    const float* p = xs + (size_t)ch * ld + pp;
                    const float m = mus[ch];
                    if (vec && pp + 3 < p_end) {
                        v = *reinterpret_cast<const float4*>(p);
                        v.x -= m; v.y -= m; v.z -= m; v.w -= m;
                    } else {
                        if (pp + 0 < p_end) v.x = p[0] - m;
                        if (pp + 1 < p_end) v.y = p[1] - m;
                        if (pp + 2 < p_end) v.z = p[2] - m;
                        if (pp + 3 < p_end) v.w = p[3] - m;
                    }
                }
                if (which) rj[q] = v; else ri[q] = v;
            }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int idx = tid + q * 256;
            const int row = idx / (GK / 4), px = (idx % (GK / 4)) * 4;
            float* di = &Xi[buf][row * GSTR + px];
            di[0] = ri[q].x; di[1] = ri[q].y; di[2] = ri[q].z; di[3] = ri[q].w;
            float* dj = &Xj[buf][row * GSTR + px];
            dj[0] = rj[q].x; dj[1] = rj[q].y; dj[2] = rj[q].z; dj[3] = rj[q].w;
        }
    };

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    const long nchunks = (p_end - p_beg + GK - 1) / GK;
    if (nchunks > 0) {
        load_global(p_beg);
        store_lds(0);
    }
    __syncthreads();
    for (long kc = 0; kc < nchunks; kc++) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_global(p_beg + (kc + 1) * GK);
        const float* ai = &Xi[buf][(wi * 32 + l31) * GSTR];
        const float* bj = &Xj[buf][(wj * 32 + l31) * GSTR];
#pragma unroll
        for (int j = 0; j < GK / 2; j++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[2 * j + h], bj[2 * j + h], acc, 0, 0, 0);
        if (kc + 1 < nchunks) store_lds(buf ^ 1);
        __syncthreads();
    }
    float* o = part + ((size_t)seg * gridDim.y + split) * C * C;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = ti * GT + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int j = tj * GT + wj * 32 + l31;
        if (i < C && j < C) o[(size_t)i * C + j] = acc[r];
    }
}

// The same on 128 x 128 output tiles (C > 64): every wave owns a 64 x 64 block = 2 x 2 MFMA tiles, so an LDS fragment
// feeds two MFMAs instead of one, and a 256-channel map is covered by 3 tile pairs that read 768 rows from L2 instead of
// 10 pairs reading 1280.  On a diagonal pair the wave below the diagonal (wi = 1, wj = 0) would only recompute the mirror
// image of its neighbour: it stores nothing and skips its MFMAs.  GK pixels per staged chunk; dynamic LDS:
// 2 x 2 x 128 x (GK + 1) floats = 66 KiB (GK = 32, two blocks per CU) or 34 KiB (GK = 16, four blocks per CU).
template <int GK>
__global__ __launch_bounds__(256) void gram128_kernel(const float* __restrict__ x, long ld, long seg_stride, long n, int C,
                                                      const float* __restrict__ mu, long chunk, int tiles,
                                                      float* __restrict__ part, int vec) {
    constexpr int GSTR = GK + 1;
    extern __shared__ __align__(16) float g_smem[];
    float* Xi = g_smem;                       // [2][GT2 * GSTR]
    float* Xj = g_smem + 2 * GT2 * GSTR;      // [2][GT2 * GSTR]
    int pi = blockIdx.x, ti = 0;
    while (pi >= tiles - ti) {
        pi -= tiles - ti;
        ti++;
    }
    const int tj = ti + pi;
    const int seg = blockIdx.z, split = blockIdx.y;
    const float* xs = x + (size_t)seg * seg_stride;
    const float* mus = mu + (size_t)seg * C;
    const long p_beg = (long)split * chunk, p_end = (p_beg + chunk < n) ? p_beg + chunk : n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1, l31 = lane & 31, h = lane >> 5;
    const bool diag = ti == tj;
    const bool mirror = diag && wi > wj;      // uniform per wave

    constexpr int NQ = GT2 * GK / 4 / 256;    // float4 per operand per thread and chunk
    // a thread stages the same NQ rows of each operand in every chunk: their means are read once.  Launched only with
    // 16-byte aligned rows and n % 4 == 0, so every load is one unconditional float4 (clamped address, zeroed by selects
    // past the end of this block's pixel range): all 2 * NQ loads of a chunk are in flight together — the first build
    // branched per load on the tail and serialised a mean load, a wait, a data load and another wait per row.
    float mi[NQ], mj[NQ];
    const float* rowi[NQ];
    const float* rowj[NQ];
    bool oki[NQ], okj[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int row = (tid + q * 256) / (GK / 4);
        const int ci = ti * GT2 + row, cj = tj * GT2 + row;
        oki[q] = ci < C;
        okj[q] = cj < C;
        mi[q] = oki[q] ? mus[ci] : 0.f;
        mj[q] = okj[q] ? mus[cj] : 0.f;
        rowi[q] = xs + (size_t)(oki[q] ? ci : 0) * ld;
        rowj[q] = xs + (size_t)(okj[q] ? cj : 0) * ld;
    }
    float4 ri[NQ], rj[NQ];
    auto centred = [&](float4 v, float m, bool ok, long pp) {
        v.x = (ok && pp + 0 < p_end) ? v.x - m : 0.f;
        v.y = (ok && pp + 1 < p_end) ? v.y - m : 0.f;
        v.z = (ok && pp + 2 < p_end) ? v.z - m : 0.f;
        v.w = (ok && pp + 3 < p_end) ? v.w - m : 0.f;
        return v;
    };
    auto load_global = [&](long p0) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const long pp = p0 + ((tid + q * 256) % (GK / 4)) * 4;
            const long pc = pp < n ? pp : 0;
            ri[q] = *reinterpret_cast<const float4*>(rowi[q] + pc);
            if (!diag) rj[q] = *reinterpret_cast<const float4*>(rowj[q] + pc);
        }
    };
    auto store_lds = [&](int buf, long p0) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int idx = tid + q * 256;
            const int row = idx / (GK / 4), px = (idx % (GK / 4)) * 4;
            const long pp = p0 + px;
            const float4 vi = centred(ri[q], mi[q], oki[q], pp);
            float* di = &Xi[buf * GT2 * GSTR + row * GSTR + px];
            di[0] = vi.x; di[1] = vi.y; di[2] = vi.z; di[3] = vi.w;
            if (!diag) {
                const float4 vj = centred(rj[q], mj[q], okj[q], pp);
                float* dj = &Xj[buf * GT2 * GSTR + row * GSTR + px];
                dj[0] = vj.x; dj[1] = vj.y; dj[2] = vj.z; dj[3] = vj.w;
            }
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const long nchunks = (p_end - p_beg + GK - 1) / GK;
    if (nchunks > 0) {
        load_global(p_beg);
        store_lds(0, p_beg);
    }
    __syncthreads();
    for (long kc = 0; kc < nchunks; kc++) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_global(p_beg + (kc + 1) * GK);
        if (!mirror) {
            const float* ai = &Xi[buf * GT2 * GSTR + (wi * 64 + l31) * GSTR];
            const float* bj = &(diag ? Xi : Xj)[buf * GT2 * GSTR + (wj * 64 + l31) * GSTR];
#pragma unroll
            for (int j = 0; j < GK / 2; j++) {
                const float a0 = ai[2 * j + h], a1 = ai[32 * GSTR + 2 * j + h];
                const float b0 = bj[2 * j + h], b1 = bj[32 * GSTR + 2 * j + h];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (kc + 1 < nchunks) store_lds(buf ^ 1, p_beg + (kc + 1) * GK);
        __syncthreads();
    }
    if (mirror) return;
    float* o = part + ((size_t)seg * gridDim.y + split) * C * C;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = ti * GT2 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int j = tj * GT2 + wj * 64 + b * 32 + l31;
                if (i < C && j < C) o[(size_t)i * C + j] = acc[a][b][r];
            }
}

// The whole upper triangle of one segment's Gram matrix per workgroup (128 < C <= 256, round 3).  NB = 8 (192 < C): the
// 8 x 8 grid of 32 x 32 MFMA tiles has 36 tiles on or above the diagonal, nine per wave, grouped so that a wave reads five
// to seven 32-channel fragments per k-step for its nine MFMAs (row blocks A = {0,1,2}, B = {3,4,5}, C = {6,7}: wave 0
// takes A x B, wave 1 the triangles of A and C, wave 2 A x C and three tiles of B x C, wave 3 the triangle of B and the
// rest of B x C).  NB = 6 (C <= 192, the PCA ranks of relu3_1): 21 tiles as 6 + 6 + 5 + 4.  Against the 128 x 128 tile
// pairs of gram128_kernel: no idle wave on diagonal tiles (36 tile products per pixel pair instead of 48 issue slots for
// 40), and every pixel chunk is staged once per segment instead of once per tile pair (256 rows instead of 512).
// grid = (splits, n_seg); part[seg][split][C][C] receives the upper 32 x 32 tiles.
constexpr int TRI_K = 32, TRI_STR = TRI_K + 1;
constexpr size_t tri_lds_bytes(int NB) { return (size_t)2 * NB * 32 * TRI_STR * sizeof(float); }
__device__ constexpr unsigned char TRI8_I[4][9] = {{0, 0, 0, 1, 1, 1, 2, 2, 2}, {0, 0, 0, 1, 1, 2, 6, 6, 7},
                                                   {0, 0, 1, 1, 2, 2, 3, 3, 4}, {3, 3, 3, 4, 4, 5, 4, 5, 5}};
__device__ constexpr unsigned char TRI8_J[4][9] = {{3, 4, 5, 3, 4, 5, 3, 4, 5}, {0, 1, 2, 1, 2, 2, 6, 7, 7},
                                                   {6, 7, 6, 7, 6, 7, 6, 7, 6}, {3, 4, 5, 4, 5, 5, 7, 6, 7}};
__device__ constexpr unsigned char TRI6_I[4][6] = {{0, 0, 0, 1, 1, 2}, {3, 3, 3, 4, 4, 5}, {0, 0, 0, 1, 1, 0}, {1, 2, 2, 2, 0, 0}};
__device__ constexpr unsigned char TRI6_J[4][6] = {{0, 1, 2, 1, 2, 2}, {3, 4, 5, 4, 5, 5}, {3, 4, 5, 3, 4, 0}, {5, 3, 4, 5, 0, 0}};
template <int NB, int W> constexpr int tri_cnt() { return NB == 8 ? 9 : (W < 2 ? 6 : (W == 2 ? 5 : 4)); }
template <int NB, int W> constexpr int tri_i(int t) { return NB == 8 ? TRI8_I[W][t] : TRI6_I[W][t]; }
template <int NB, int W> constexpr int tri_j(int t) { return NB == 8 ? TRI8_J[W][t] : TRI6_J[W][t]; }

template <int NB, int W>
__device__ __forceinline__ void tri_chunk(const float* __restrict__ X, int l31, int h, floatx16 (&acc)[tri_cnt<NB, W>()]) {
    constexpr int CNT = tri_cnt<NB, W>();
    constexpr unsigned need = [] {
        unsigned m = 0;
        for (int t = 0; t < CNT; t++) m |= (1u << tri_i<NB, W>(t)) | (1u << tri_j<NB, W>(t));
        return m;
    }();
#pragma unroll 2
    for (int j = 0; j < TRI_K / 2; j++) {
        float f[NB];
#pragma unroll
        for (int b = 0; b < NB; b++)
            if (need >> b & 1) f[b] = X[(b * 32 + l31) * TRI_STR + 2 * j + h];
        // (channels past C hold a copy of a real row: their tiles only reach Gram entries that are never stored — no branch)
#pragma unroll
        for (int t = 0; t < CNT; t++)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[tri_i<NB, W>(t)], f[tri_j<NB, W>(t)], acc[t], 0, 0, 0);
    }
}

template <int NB, int W>
__device__ __forceinline__ void tri_store(float* __restrict__ o, int C, int l31, int h, const floatx16 (&acc)[tri_cnt<NB, W>()]) {
#pragma unroll
    for (int t = 0; t < tri_cnt<NB, W>(); t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = tri_i<NB, W>(t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int j = tri_j<NB, W>(t) * 32 + l31;
            if (i < C && j < C) o[(size_t)i * C + j] = acc[t][r];
        }
}

// the whole kernel body of wave W: every wave runs its own copy of the chunk loop (the same number of barriers in each), so
// that its accumulators stay in one place for the whole launch — with the wave switch inside the loop the compiler
// moved all 144 accumulator registers in and out of the branch once per chunk
template <int NB, int W>
__device__ __forceinline__ void tri_main(const float* __restrict__ xs, long ld, long n, int C, const float* __restrict__ mus,
                                         long p_beg, long p_end, float* __restrict__ o, float* smem) {
    constexpr int ROWS = NB * 32, CNT = tri_cnt<NB, W>();
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    // a thread stages the same NQ rows (tid / 8 + 32 q) of every chunk, 4 pixels each: one unconditional float4 per row
    // (clamped address, zeroed by selects past the end of this block's pixel range)
    constexpr int NQ = ROWS * TRI_K / 4 / 256;
    const int row0 = tid >> 3, px = (tid & 7) * 4;
    float m[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) m[q] = mus[row0 + 32 * q < C ? row0 + 32 * q : 0];
    const float* base = xs + (size_t)row0 * ld + px;
    const long qstride = 32 * ld;
    float4 rv[NQ];
    auto load_global = [&](long p0) {
        const long pc = (p0 + px < n) ? p0 : 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) rv[q] = *reinterpret_cast<const float4*>(base + (row0 + 32 * q < C ? q * qstride : 0) + pc);
    };
    auto store_lds = [&](int buf, long p0) {
        const long left = p_end - (p0 + px);  // pixels of this thread's quad inside the block's range
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            float* d = &smem[buf * ROWS * TRI_STR + (row0 + 32 * q) * TRI_STR + px];
            d[0] = left > 0 ? rv[q].x - m[q] : 0.f;
            d[1] = left > 1 ? rv[q].y - m[q] : 0.f;
            d[2] = left > 2 ? rv[q].z - m[q] : 0.f;
            d[3] = left > 3 ? rv[q].w - m[q] : 0.f;
        }
    };
    floatx16 acc[CNT];
#pragma unroll
    for (int t = 0; t < CNT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const long nchunks = (p_end - p_beg + TRI_K - 1) / TRI_K;
    if (nchunks > 0) {
        load_global(p_beg);
        store_lds(0, p_beg);
    }
    __syncthreads();
    for (long kc = 0; kc < nchunks; kc++) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_global(p_beg + (kc + 1) * TRI_K);
        tri_chunk<NB, W>(&smem[buf * ROWS * TRI_STR], l31, h, acc);
        if (kc + 1 < nchunks) store_lds(buf ^ 1, p_beg + (kc + 1) * TRI_K);
        __syncthreads();
    }
    tri_store<NB, W>(o, C, l31, h, acc);
}

template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gram_tri_kernel(
    const float* __restrict__ x, long ld, long seg_stride, long n, int C, const float* __restrict__ mu, long chunk,
    float* __restrict__ part) {
    extern __shared__ __align__(16) float g_smem[];  // [2][NB * 32 * TRI_STR]
    const int seg = blockIdx.y, split = blockIdx.x;
    const float* xs = x + (size_t)seg * seg_stride;
    const float* mus = mu + (size_t)seg * C;
    const long p_beg = (long)split * chunk, p_end = (p_beg + chunk < n) ? p_beg + chunk : n;
    float* o = part + ((size_t)seg * gridDim.x + split) * C * C;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) tri_main<NB, 0>(xs, ld, n, C, mus, p_beg, p_end, o, g_smem);
    else if (wave == 1) tri_main<NB, 1>(xs, ld, n, C, mus, p_beg, p_end, o, g_smem);
    else if (wave == 2) tri_main<NB, 2>(xs, ld, n, C, mus, p_beg, p_end, o, g_smem);
    else tri_main<NB, 3>(xs, ld, n, C, mus, p_beg, p_end, o, g_smem);
}

// cov[s][i][j] = sum_split part / N + eps * (i == j); lower triangle mirrored from the upper tiles.
// pool: one covariance over all segments (the reference's batch semantics), N = n * n_seg.
__global__ void cov_finalize_kernel(const float* __restrict__ part, int C, int n_seg, int splits, int pool, float N,
                                    float eps, int gt, float* __restrict__ cov) {
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.z;
    if (j >= C) return;
    // element (i, j) lives in the gt x gt block (i/gt, j/gt) if that is an upper block, else read (j, i).  The tile-pair
    // kernels leave exactly the upper 64-blocks behind (the 128-tile kernel skips the block below the diagonal of a diagonal
    // tile), the whole-triangle kernel the upper 32-blocks.
    const bool upper = (i / gt) <= (j / gt);
    const int ri = upper ? i : j, rj = upper ? j : i;
    float sum = 0.f;
    const int s_beg = pool ? 0 : s, s_end = pool ? n_seg : s + 1;
    for (int ss = s_beg; ss < s_end; ss++) {
        const float* p = part + (size_t)ss * splits * C * C + (size_t)ri * C + rj;
        int k = 0;
        for (; k + 8 <= splits; k += 8) {   // eight partials in flight, added in the same fixed order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = p[(size_t)(k + u) * C * C];
#pragma unroll
            for (int u = 0; u < 8; u++) sum += v[u];
        }
        for (; k < splits; k++) sum += p[(size_t)k * C * C];
    }
    float v = __fdiv_rn(sum, N);
    if (i == j) v = v + eps;
    cov[(size_t)s * C * C + (size_t)i * C + j] = v;
}

// The same by upper blocks (round 6): one workgroup per upper GT x GT block sums its partials along the rows (coalesced), stores the
// block, and stores the mirror image of an off-diagonal block through an LDS transpose.  cov_finalize_kernel read the lower
// half of the matrix at the transposed address — one 4-byte element per 64-byte sector — and took 96 us per launch for 64
// matrices of 256^2 (1.5 TB/s), 5.5 ms of a 64-texture chol step.  Same sums in the same order: same bits.
// grid (upper blocks, pool ? 1 : n_seg)
template <int GT_>
__global__ __launch_bounds__(256) void cov_finalize_blocks_kernel(const float* __restrict__ part, int C, int n_seg, int splits, int pool,
                                                                  float N, float eps, float* __restrict__ cov) {
    const int nb = (C + GT_ - 1) / GT_;
    int bi = 0, rest = blockIdx.x;   // upper blocks row by row: (0, 0 .. nb - 1), (1, 1 .. nb - 1), ...
    while (rest >= nb - bi) {
        rest -= nb - bi;
        bi++;
    }
    const int bj = bi + rest, s = blockIdx.y;
    __shared__ float t[GT_][GT_ + 1];
    const int s_beg = pool ? 0 : s, s_end = pool ? n_seg : s + 1;
    const size_t cc = (size_t)C * C;
    for (int e = threadIdx.x; e < GT_ * GT_; e += 256) {
        const int r = e / GT_, c = e % GT_, i = bi * GT_ + r, j = bj * GT_ + c;
        float v = 0.f;
        if (i < C && j < C) {
            float sum = 0.f;
            for (int ss = s_beg; ss < s_end; ss++) {
                const float* p = part + (size_t)ss * splits * cc + (size_t)i * C + j;
                int k = 0;
                for (; k + 8 <= splits; k += 8) {   // eight partials in flight, added in the same fixed order
                    float q[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) q[u] = p[(size_t)(k + u) * cc];
#pragma unroll
                    for (int u = 0; u < 8; u++) sum += q[u];
                }
                for (; k < splits; k++) sum += p[(size_t)k * cc];
            }
            v = __fdiv_rn(sum, N);
            if (i == j) v = v + eps;
            cov[(size_t)s * cc + (size_t)i * C + j] = v;
        }
        t[r][c] = v;
    }
    if (bi == bj) return;
    __syncthreads();
    for (int e = threadIdx.x; e < GT_ * GT_; e += 256) {
        const int r = e / GT_, c = e % GT_, i = bj * GT_ + r, j = bi * GT_ + c;   // the mirror block: row = a column of the upper one
        if (i < C && j < C) cov[(size_t)s * cc + (size_t)i * C + j] = t[c][r];
    }
}

int device_cu_count();
bool gram_tri_enabled = true;  // (internal, not ABI: tests compare the whole-triangle kernel with the tile-pair kernel)

template <int NB>
static int launch_tri(const float* x, long ld, long seg_stride, long n, int C, const float* mu, long chunk, float* part, int splits,
                      int n_seg, hipStream_t st) {
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gram_tri_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)tri_lds_bytes(NB));
        if (e != hipSuccess) {
            set_error("gram_tri_kernel: cannot reserve LDS: %s", hipGetErrorString(e));
            return OPTEX_E_LAUNCH;
        }
        attr_done[dev & 63] = true;
    }
    hipLaunchKernelGGL(gram_tri_kernel<NB>, dim3(splits, n_seg), dim3(256), tri_lds_bytes(NB), st, x, ld, seg_stride, n, C, mu, chunk,
                       part);
    return OPTEX_OK;
}

// How many pixel ranges a segment's Gram matrix may be split into.  64 where a C x C partial is large against the pixels it
// covers (C = 256, n = 16384: 64 partials are already as many bytes as the feature map); up to 512 where it is small (round 6:
// ONE texture, relu1_1, n = 262144 pixels of <= 64 PCA channels ran on 64 workgroups = a quarter of the chip, 150 us per launch).
static int gram_split_cap(long n, int C) {
    long cap = n / (8L * C);
    if (cap < G_MAX_SPLITS) cap = G_MAX_SPLITS;
    if (cap > 512) cap = 512;
    return (int)cap;
}

static int gram_splits(long n, int C, int n_seg, bool big) {
    // 64-tiles: two blocks' worth of work per CU; 128-tiles (two resident blocks per CU, long blocks): about four rounds of
    // resident blocks, so that the last round's idle CUs cost a few per cent instead of a third
    const int gt = big ? GT2 : GT;
    const int tiles = (C + gt - 1) / gt, pairs = tiles * (tiles + 1) / 2;
    const long target = (big ? 6L : 2L) * device_cu_count();
    long want = (target + (long)pairs * n_seg - 1) / ((long)pairs * n_seg);
    long maxs = (n + 1023) / 1024;
    if (want > maxs) want = maxs;
    const int cap = gram_split_cap(n, C);
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace optex

using namespace optex;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the split-K partials of the Gram kernels: every launch of <= n_seg segments and <= n pixels uses at most
// s * min(cap, ceil(6 CUs / s)) partials (gram_splits / the whole-triangle kernel's count below), s = its segment count
extern "C" size_t optex_linear_stats_ws_bytes(long n, int C, int n_seg) {
    const long cap = gram_split_cap(n, C), target = 6L * device_cu_count();
    long most = 0;
    for (long s = 1; s <= n_seg; s++) {
        long per = (target + s - 1) / s;
        if (per > cap) per = cap;
        if (per < 1) per = 1;
        if (s * per > most) most = s * per;
    }
    return align_up((size_t)most * C * C * sizeof(float), 256);
}

extern "C" int optex_linear_stats(const float* x, long ld, long seg_stride, long n, int C, int n_seg, int pool,
                                  float eps, float* mu, float* cov, void* ws, size_t ws_bytes, void* stream) {
    return optex::linear_stats_parts(x, ld, seg_stride, n, C, n_seg, pool, eps, mu, cov, ws, ws_bytes, nullptr, 0, stream);
}

// sum_parts [n_seg][parts][C]: per-tile row sums of x the producing GEMM already took (or NULL: col_mean_kernel reads x)
int optex::linear_stats_parts(const float* x, long ld, long seg_stride, long n, int C, int n_seg, int pool, float eps, float* mu,
                              float* cov, void* ws, size_t ws_bytes, const float* sum_parts, int parts, void* stream) {
    if (!x || !mu || !cov || !ws || n <= 0 || C <= 0 || n_seg <= 0 || ld < n) {
        set_error("optex_linear_stats: bad argument (n=%ld C=%d n_seg=%d ld=%ld)", n, C, n_seg, ld);
        return OPTEX_E_ARG;
    }
    if (int rc = check_ws("optex_linear_stats", ws, ws_bytes, optex_linear_stats_ws_bytes(n, C, n_seg))) return rc;
    hipStream_t st = as_stream(stream);
    const int vec = aligned16(x) && ld % 4 == 0 && seg_stride % 4 == 0;
    if (sum_parts) {
        ProfScope prof(KC_MEAN, st, 0.0, 4.0 * (double)parts * C * n_seg);
        hipLaunchKernelGGL(mean_from_parts_kernel, dim3((C * n_seg + 63) / 64), dim3(256), 0, st, sum_parts, parts, C,
                           C * n_seg, n, mu);
    } else {
        ProfScope prof(KC_MEAN, st, 0.0, 4.0 * (double)n * C * n_seg);
        // few long columns: several workgroups per column (their double partial sums borrow the head of the Gram workspace,
        // which the Gram kernel behind them on the stream overwrites)
        const long ncols = (long)C * n_seg;
        long P = ncols < 2L * device_cu_count() ? n / 16384 : 1;
        if (P > 16) P = 16;
        if (P >= 2 && (size_t)P * ncols * sizeof(double) <= ws_bytes) {
            const long chunk = ((n + P - 1) / P + 3) / 4 * 4;
            double* dpart = static_cast<double*>(ws);
            hipLaunchKernelGGL(col_sum_parts_kernel, dim3((unsigned)ncols, (unsigned)P), dim3(256), 0, st, x, ld, seg_stride, n, C,
                               chunk, dpart, vec);
            hipLaunchKernelGGL(mean_from_dparts_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, st, dpart, (int)P,
                               (int)ncols, n, mu);
        } else {
            hipLaunchKernelGGL(col_mean_kernel, dim3(C * n_seg), dim3(256), 0, st, x, ld, seg_stride, n, C, mu, vec);
        }
    }
    int rc = check_launch("col_mean_kernel");
    if (rc) return rc;
    const bool big = C > GT && vec && n % 4 == 0;  // the 128-tile kernel loads unconditional float4s
    const bool tri = big && C > 128 && C <= 256 && gram_tri_enabled;
    const int gt = big ? GT2 : GT;
    const int tiles = (C + gt - 1) / gt, pairs = tiles * (tiles + 1) / 2;
    int splits = gram_splits(n, C, n_seg, big);
    if (tri) {
        // one workgroup per (split, segment), two resident per CU: whole rounds of resident workgroups
        const long target = 2L * device_cu_count();
        long want = (target + n_seg - 1) / n_seg, maxs = (n + 255) / 256;
        want = want > maxs ? maxs : want;
        const long cap = gram_split_cap(n, C);
        splits = (int)(want > cap ? cap : (want < 1 ? 1 : want));
    }
    long chunk = (n + splits - 1) / splits;
    chunk = (chunk + GK - 1) / GK * GK;  // chunk starts stay multiples of 4 pixels (float4 loads)
    float* part = static_cast<float*>(ws);
    {
        // upper-triangular tile pairs only: pairs * GT*GT * 2n flop per segment
        // algorithmic flops = the upper-triangular 64 x 64 blocks: b (b + 1) / 2 * 64 * 64 * 2n per segment
        const int b64 = (C + GT - 1) / GT;
        ProfScope prof(KC_GRAM, st, 2.0 * (b64 * (b64 + 1) / 2) * GT * GT * (double)n * n_seg, 4.0 * (double)n * C * n_seg);
        if (tri) {
            if (C > 192) rc = launch_tri<8>(x, ld, seg_stride, n, C, mu, chunk, part, splits, n_seg, st);
            else rc = launch_tri<6>(x, ld, seg_stride, n, C, mu, chunk, part, splits, n_seg, st);
            if (rc) return rc;
        } else if (big) {
            constexpr int gk = 32;
            const size_t lds = (size_t)4 * GT2 * (gk + 1) * sizeof(float);
            auto kern = gram128_kernel<gk>;
            static bool attr_done[64] = {};
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (!attr_done[dev & 63]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds);
                if (e != hipSuccess) {
                    set_error("gram128_kernel: cannot reserve LDS: %s", hipGetErrorString(e));
                    return OPTEX_E_LAUNCH;
                }
                attr_done[dev & 63] = true;
            }
            hipLaunchKernelGGL(kern, dim3(pairs, splits, n_seg), dim3(256), lds, st, x, ld, seg_stride, n, C, mu, chunk, tiles,
                               part, vec);
        } else {
            hipLaunchKernelGGL(gram_kernel, dim3(pairs, splits, n_seg), dim3(256), 0, st, x, ld, seg_stride, n, C, mu, chunk,
                               tiles, part, vec);
        }
    }
    if ((rc = check_launch("gram_kernel"))) return rc;
    const float N = pool ? (float)((double)n * n_seg) : (float)n;
    ProfScope prof(KC_COVFIN, st, 0.0, 4.0 * (double)C * C * n_seg * (splits + 1));
    const int fgt = tri ? 32 : GT, fnb = (C + fgt - 1) / fgt;
    if ((long)(fnb * (fnb + 1) / 2) * (pool ? 1 : n_seg) < 2L * device_cu_count()) {
        // few matrices (one texture): one thread per element, a workgroup per row — more workgroups than blocks
        hipLaunchKernelGGL(cov_finalize_kernel, dim3((C + 255) / 256, C, pool ? 1 : n_seg), dim3(256), 0, st, part, C, n_seg,
                           splits, pool, N, eps, fgt, cov);
    } else {
        const dim3 grid((unsigned)(fnb * (fnb + 1) / 2), (unsigned)(pool ? 1 : n_seg));
        if (tri) hipLaunchKernelGGL(cov_finalize_blocks_kernel<32>, grid, dim3(256), 0, st, part, C, n_seg, splits, pool, N, eps, cov);
        else hipLaunchKernelGGL(cov_finalize_blocks_kernel<GT>, grid, dim3(256), 0, st, part, C, n_seg, splits, pool, N, eps, cov);
    }
    return check_launch("cov_finalize_kernel");
}
