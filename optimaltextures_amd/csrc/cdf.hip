// cdf.hip — K2a/K2b/K2c/K3: the reference's `cdf` histogram matching (histmatch.py:49-92) as four streaming
// kernels over channel-major columns.  All of them are HBM-bound (0 flop/B); what matters is that every
// column is read in whole 16-byte vectors and that the per-column state (256-bin histogram, 3 KB LUT) lives
// in LDS.  Algorithmic bytes per (pixel, channel): min/max 4, histogram 4, apply 8.
//
// Exactness rules (so that the result is bit-identical to the reference on identical inputs):
//   * bin = trunc((x - lo) * 256 / (hi - lo)) with a true IEEE division, x == hi -> bin 255, lo == hi -> [lo-1, hi+1]
//   * bin_edges = torch.linspace(lo, hi, 257)[1:]: step = (hi-lo)/256, first half fma(step, i, lo), second half
//     fma(-step, 256-i, hi)
//   * interp (histmatch.py:72-92): slope * (x - xp[idx]) + fp[idx] with SEPARATE multiply and add (the library is
//     built with -ffp-contract=off), IEEE division for the slope, and the reference's 3-stage non-finite fallback
//   * counts are integers (uint32 atomics), so the histogram does not depend on the order of accumulation
#include "optex_common.h"

namespace optex {

__device__ __forceinline__ bool finite_f(float f) { return (__float_as_uint(f) & 0x7f800000u) != 0x7f800000u; }

// histmatch.py:77-90 for one x given idx = searchsorted(xp, x) (already clamped to n-1)
__device__ __forceinline__ float interp_eval(float x, int idx, const float* xp, const float* fp, int n) {
    const int nxt = (idx + 1 > n - 1) ? n - 1 : idx + 1;
    const float slope = __fdiv_rn(fp[nxt] - fp[idx], xp[nxt] - xp[idx]);
    float f = __fadd_rn(__fmul_rn(slope, x - xp[idx]), fp[idx]);
    if (!finite_f(f)) {
        const float f2 = __fadd_rn(__fmul_rn(slope, x - xp[nxt]), fp[nxt]);
        f = finite_f(f2) ? f2 : fp[idx];
    }
    return f;
}

// torch.searchsorted(xp, v), right=False: same bisection as ATen's cus_lower_bound
__device__ __forceinline__ int lower_bound_f(const float* xp, int n, float v) {
    int start = 0, end = n;
    while (start < end) {
        const int mid = start + ((end - start) >> 1);
        if (!(xp[mid] >= v)) start = mid + 1;
        else end = mid;
    }
    return start;
}

__device__ __forceinline__ float linspace_edge(float lo, float hi, float step, int i) {  // i in [0, 256]
    return (i < (kBins + 1) / 2) ? __fmaf_rn(step, (float)i, lo) : __fmaf_rn(-step, (float)(kBins - i), hi);
}

// ------------------------------------------------------------------------------------------------ K2a min/max
// grid = (columns, chunks).  chunks == 1: plain float stores.  chunks > 1: atomics on totalOrder keys held in the
// output arrays themselves (initialised by minmax_init_kernel, decoded by minmax_decode_kernel).
template <bool ATOMIC>
__global__ __launch_bounds__(256) void col_minmax_kernel(const float* __restrict__ x, long ld, long seg_stride, long n,
                                                         int C, long chunk, const float* __restrict__ omn,
                                                         const float* __restrict__ omx, int o_n_seg, float* mn,
                                                         float* mx, int vec) {
    const int col = blockIdx.x, seg = col / C, c = col % C;
    const float* p = x + (size_t)seg * seg_stride + (size_t)c * ld;
    const long beg = (long)blockIdx.y * chunk, end = (beg + chunk < n) ? beg + chunk : n;
    float lo = INFINITY, hi = -INFINITY;
    if (vec) {  // beg is a multiple of 4 and rows are 16-byte aligned
        const long nv = (end - beg) / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p + beg);
        for (long i = threadIdx.x; i < nv; i += blockDim.x) {
            const float4 v = p4[i];
            lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
            hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
        }
        for (long i = beg + nv * 4 + threadIdx.x; i < end; i += blockDim.x) {
            lo = fminf(lo, p[i]);
            hi = fmaxf(hi, p[i]);
        }
    } else {
        for (long i = beg + threadIdx.x; i < end; i += blockDim.x) {
            lo = fminf(lo, p[i]);
            hi = fmaxf(hi, p[i]);
        }
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    __shared__ float slo[4], shi[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        slo[w] = lo;
        shi[w] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
        if (omn) {  // joint range with the (already reduced) other distribution, histmatch.py:52-53
            const int oc = ((o_n_seg == 1) ? 0 : seg) * C + c;
            lo = fminf(lo, omn[oc]);
            hi = fmaxf(hi, omx[oc]);
        }
        if (ATOMIC) {
            atomicMin(reinterpret_cast<unsigned*>(mn) + col, f2key(lo));
            atomicMax(reinterpret_cast<unsigned*>(mx) + col, f2key(hi));
        } else {
            mn[col] = lo;
            mx[col] = hi;
        }
    }
}

// min / max of every (segment, channel) from the per-tile partials the forward rotation GEMM left behind
// (GemmArgs::rowstat = 1: part [n_seg][parts][C]), joined with the other distribution's range like col_minmax_kernel
__global__ __launch_bounds__(256) void minmax_from_parts_kernel(const float* __restrict__ pmn, const float* __restrict__ pmx,
                                                                int parts, int C, int ncols, const float* __restrict__ omn,
                                                                const float* __restrict__ omx, int o_n_seg,
                                                                float* __restrict__ mn, float* __restrict__ mx) {
    // 64 columns per block (coalesced along the channel), 4 threads per column over interleaved partials
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float lo = INFINITY, hi = -INFINITY;
    int seg = 0, c = 0;
    if (col < ncols) {
        seg = col / C;
        c = col % C;
        const float* a = pmn + (size_t)seg * parts * C + c;
        const float* b = pmx + (size_t)seg * parts * C + c;
#pragma unroll 8
        for (int p = g; p < parts; p += 4) {
            lo = fminf(lo, a[(size_t)p * C]);
            hi = fmaxf(hi, b[(size_t)p * C]);
        }
    }
    __shared__ float slo[4][64], shi[4][64];
    slo[g][cl] = lo;
    shi[g][cl] = hi;
    __syncthreads();
    if (g == 0 && col < ncols) {
        lo = fminf(fminf(slo[0][cl], slo[1][cl]), fminf(slo[2][cl], slo[3][cl]));
        hi = fmaxf(fmaxf(shi[0][cl], shi[1][cl]), fmaxf(shi[2][cl], shi[3][cl]));
        if (omn) {
            const int oc = ((o_n_seg == 1) ? 0 : seg) * C + c;
            lo = fminf(lo, omn[oc]);
            hi = fmaxf(hi, omx[oc]);
        }
        mn[col] = lo;
        mx[col] = hi;
    }
}

__global__ void minmax_init_kernel(float* mn, float* mx, int ncols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncols) {
        reinterpret_cast<unsigned*>(mn)[i] = 0xffffffffu;
        reinterpret_cast<unsigned*>(mx)[i] = 0u;
    }
}

__global__ void minmax_decode_kernel(float* mn, float* mx, int ncols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncols) {
        mn[i] = key2f(reinterpret_cast<unsigned*>(mn)[i]);
        mx[i] = key2f(reinterpret_cast<unsigned*>(mx)[i]);
    }
}

// ------------------------------------------------------------------------------------------------ K2b histogram
__device__ __forceinline__ void hist_add(unsigned* h, float v, float lo, float hi, float range) {
    if (!(v >= lo && v <= hi)) return;  // also skips NaN, like ATen
    int pos = (int)__fdiv_rn((v - lo) * (float)kBins, range);
    pos = pos > kBins - 1 ? kBins - 1 : pos;
    atomicAdd(&h[pos], 1u);
}

// grid = (columns, chunks).  lohi_seg_div: the (lo, hi) of column (seg, c) is read at [(seg / lohi_seg_div), c] so that a
// shared source (one segment) can be binned with every target segment's range: x_n_seg == 1 -> x segment 0 always.
__global__ __launch_bounds__(256) void col_hist_kernel(const float* __restrict__ x, long ld, long seg_stride, long n,
                                                       int C, int x_n_seg, long chunk, const float* __restrict__ lo_,
                                                       const float* __restrict__ hi_, unsigned* __restrict__ hist,
                                                       int vec) {
    const int col = blockIdx.x, seg = col / C, c = col % C;
    const int xseg = (x_n_seg == 1) ? 0 : seg;
    const float* p = x + (size_t)xseg * seg_stride + (size_t)c * ld;
    __shared__ unsigned sh[4][kBins];
    for (int i = threadIdx.x; i < 4 * kBins; i += blockDim.x) (&sh[0][0])[i] = 0u;
    __syncthreads();
    float lo = lo_[col], hi = hi_[col];
    if (lo == hi) {
        lo -= 1.0f;
        hi += 1.0f;
    }
    const float range = hi - lo;
    unsigned* h = sh[threadIdx.x >> 6];
    const long beg = (long)blockIdx.y * chunk, end = (beg + chunk < n) ? beg + chunk : n;
    if (vec) {
        const long nv = (end - beg) / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p + beg);
        for (long i = threadIdx.x; i < nv; i += blockDim.x) {
            const float4 v = p4[i];
            hist_add(h, v.x, lo, hi, range);
            hist_add(h, v.y, lo, hi, range);
            hist_add(h, v.z, lo, hi, range);
            hist_add(h, v.w, lo, hi, range);
        }
        for (long i = beg + nv * 4 + threadIdx.x; i < end; i += blockDim.x) hist_add(h, p[i], lo, hi, range);
    } else {
        for (long i = beg + threadIdx.x; i < end; i += blockDim.x) hist_add(h, p[i], lo, hi, range);
    }
    __syncthreads();
    unsigned* g = hist + (size_t)col * kBins;
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) {
        const unsigned v = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
        if (gridDim.y == 1) g[i] = v;
        else if (v) atomicAdd(&g[i], v);
    }
}

// ------------------------------------------------------------------------------------------------ K2c CDFs -> LUT
// one 256-thread block per column, thread i = bin i.  lut[col] = { bin_edges[256], remapped_cdf[256], slope[256] }
__global__ __launch_bounds__(256) void cdf_lut_kernel(const unsigned* __restrict__ hist_t,
                                                      const unsigned* __restrict__ hist_s,
                                                      const float* __restrict__ lo_, const float* __restrict__ hi_,
                                                      float* __restrict__ lut, float* __restrict__ dbg) {
    const int col = blockIdx.x, i = threadIdx.x;
    __shared__ unsigned ct[kBins], cs[kBins];
    __shared__ float tcdf[kBins], scdf[kBins], edges[kBins], rm[kBins];
    const unsigned ht = hist_t[(size_t)col * kBins + i], hs = hist_s[(size_t)col * kBins + i];
    ct[i] = ht;
    cs[i] = hs;
    __syncthreads();
    // inclusive scan (Hillis-Steele); integer, hence exact and equal to torch's fp32 cumsum while totals < 2^24
    for (int off = 1; off < kBins; off <<= 1) {
        const unsigned a = (i >= off) ? ct[i - off] : 0u, b = (i >= off) ? cs[i - off] : 0u;
        __syncthreads();
        ct[i] += a;
        cs[i] += b;
        __syncthreads();
    }
    float ft = (float)ct[i], fs = (float)cs[i];
    float tl = (float)ct[kBins - 1], sl = (float)cs[kBins - 1];
    if (ct[kBins - 1] >= (1u << 24) || cs[kBins - 1] >= (1u << 24)) {
        // beyond 2^24 the reference's sequential fp32 cumsum rounds: replay it literally
        __syncthreads();
        if (i == 0) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < kBins; k++) {
                const unsigned hk = hist_t[(size_t)col * kBins + k], sk = hist_s[(size_t)col * kBins + k];
                a = a + (float)hk;
                b = b + (float)sk;
                tcdf[k] = a;
                scdf[k] = b;
            }
        }
        __syncthreads();
        ft = tcdf[i];
        fs = scdf[i];
        tl = tcdf[kBins - 1];
        sl = scdf[kBins - 1];
        __syncthreads();
    }
    const float lo = lo_[col], hi = hi_[col];
    const float step = __fdiv_rn(hi - lo, (float)kBins);
    tcdf[i] = __fdiv_rn(ft, tl);
    scdf[i] = __fdiv_rn(fs, sl);
    edges[i] = linspace_edge(lo, hi, step, i + 1);
    __syncthreads();
    // remapped_cdf = interp(target_cdf, source_cdf, bin_edges)   histmatch.py:67
    const float x = tcdf[i];
    int idx = lower_bound_f(scdf, kBins, x);
    idx = idx > kBins - 1 ? kBins - 1 : idx;
    const float r = interp_eval(x, idx, scdf, edges, kBins);
    rm[i] = r;
    __syncthreads();
    const int nxt = (i + 1 > kBins - 1) ? kBins - 1 : i + 1;
    const float slope = __fdiv_rn(rm[nxt] - rm[i], edges[nxt] - edges[i]);
    float* l = lut + (size_t)col * 3 * kBins;
    l[i] = edges[i];
    l[kBins + i] = r;
    l[2 * kBins + i] = slope;
    if (dbg) {
        float* d = dbg + (size_t)col * (2 + 4 * kBins);
        if (i == 0) {
            d[0] = lo;
            d[1] = hi;
        }
        d[2 + i] = (float)ht;
        d[2 + kBins + i] = (float)hs;
        d[2 + 2 * kBins + i] = edges[i];
        d[2 + 3 * kBins + i] = r;
    }
}

// ------------------------------------------------------------------------------------------------ K3 apply
// out = interp(target_channel, bin_edges, remapped_cdf)   histmatch.py:68
__device__ __forceinline__ float lut_apply(float x, float lo, float range256, const float* e, const float* rm,
                                           const float* sl) {
    // candidate bin from the histogram formula, then an exact fix-up to idx = searchsorted_left(edges, x)
    int idx = 0;
    if (range256 > 0.f) {
        const float t = (x - lo) * range256;
        idx = (t >= 0.f) ? ((t < 255.f) ? (int)t : 255) : 0;
    }
    while (idx > 0 && e[idx - 1] >= x) idx--;
    while (idx < kBins - 1 && !(e[idx] >= x)) idx++;
    const int nxt = (idx + 1 > kBins - 1) ? kBins - 1 : idx + 1;
    const float slope = sl[idx];
    float f = __fadd_rn(__fmul_rn(slope, x - e[idx]), rm[idx]);
    if (!finite_f(f)) {
        const float f2 = __fadd_rn(__fmul_rn(slope, x - e[nxt]), rm[nxt]);
        f = finite_f(f2) ? f2 : rm[idx];
    }
    return f;
}

__global__ __launch_bounds__(256) void cdf_apply_kernel(const float* __restrict__ x, long ld, long seg_stride, long n,
                                                        int C, long chunk, const float* __restrict__ lo_,
                                                        const float* __restrict__ hi_, const float* __restrict__ lut,
                                                        float* __restrict__ out, long ldo, long o_seg_stride, int vec) {
    const int col = blockIdx.x, seg = col / C, c = col % C;
    __shared__ float e[kBins], rm[kBins], sl[kBins];
    const float* l = lut + (size_t)col * 3 * kBins;
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) {
        e[i] = l[i];
        rm[i] = l[kBins + i];
        sl[i] = l[2 * kBins + i];
    }
    __syncthreads();
    const float lo = lo_[col], hi = hi_[col];
    const float range = hi - lo;
    const float range256 = (range > 0.f) ? 256.f / range : 0.f;  // only seeds the search; exactness comes from the fix-up
    const float* p = x + (size_t)seg * seg_stride + (size_t)c * ld;
    float* o = out + (size_t)seg * o_seg_stride + (size_t)c * ldo;
    const long beg = (long)blockIdx.y * chunk, end = (beg + chunk < n) ? beg + chunk : n;
    if (vec) {
        const long nv = (end - beg) / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p + beg);
        float4* o4 = reinterpret_cast<float4*>(o + beg);
        for (long i = threadIdx.x; i < nv; i += blockDim.x) {
            const float4 v = p4[i];
            float4 r;
            r.x = lut_apply(v.x, lo, range256, e, rm, sl);
            r.y = lut_apply(v.y, lo, range256, e, rm, sl);
            r.z = lut_apply(v.z, lo, range256, e, rm, sl);
            r.w = lut_apply(v.w, lo, range256, e, rm, sl);
            o4[i] = r;
        }
        for (long i = beg + nv * 4 + threadIdx.x; i < end; i += blockDim.x)
            o[i] = lut_apply(p[i], lo, range256, e, rm, sl);
    } else {
        for (long i = beg + threadIdx.x; i < end; i += blockDim.x) o[i] = lut_apply(p[i], lo, range256, e, rm, sl);
    }
}

// ------------------------------------------------------------------------------------------------ any bin count
// histmatch.py:49-69 with the reference's third argument `bins` left free (every caller inside the reference keeps 256, the
// kernels above; this one serves a direct cdf_match(target, source, bins) call).  One 256-thread workgroup per column does the
// whole function: joint range, both histograms, both CDFs, the remapped CDF and the final interpolation.  The six per-column
// arrays of `bins` words live in LDS up to kBinsLds bins and in the caller's workspace (L2-resident) beyond that.
constexpr int kBinsLds = 2048;

// cdf[k] = cumsum(h)[k] / cumsum(h)[-1] as torch does it in fp32: integer prefix sums are exact (and equal) below 2^24,
// beyond that the sequential fp32 accumulation is replayed
__device__ void hist_to_cdf(const unsigned* h, float* cdf, int bins, unsigned* part) {
    const int i = threadIdx.x, per = (bins + 255) / 256;
    const int b = i * per < bins ? i * per : bins, e = b + per < bins ? b + per : bins;
    unsigned local = 0;
    for (int k = b; k < e; k++) local += h[k];
    __syncthreads();
    part[i] = local;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned a = (i >= off) ? part[i - off] : 0u;
        __syncthreads();
        part[i] += a;
        __syncthreads();
    }
    const unsigned total = part[255];
    if (total < (1u << 24)) {
        unsigned run = part[i] - local;
        const float tl = (float)total;
        for (int k = b; k < e; k++) {
            run += h[k];
            cdf[k] = __fdiv_rn((float)run, tl);
        }
    } else {
        if (i == 0) {
            float acc = 0.f;
            for (int k = 0; k < bins; k++) {
                acc = acc + (float)h[k];
                cdf[k] = acc;
            }
        }
        __syncthreads();
        const float tl = cdf[bins - 1];
        __syncthreads();
        for (int k = b; k < e; k++) cdf[k] = __fdiv_rn(cdf[k], tl);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void cdf_bins_kernel(const float* __restrict__ target, long ldt, long tss, long nt,
                                                       const float* __restrict__ source, long lds, long sss, long ns,
                                                       int src_n_seg, int C, int bins, float* gws, float* __restrict__ out,
                                                       long ldo, long oss) {
    extern __shared__ float dyn[];
    __shared__ float slo[4], shi[4];
    __shared__ unsigned part[256];
    const int col = blockIdx.x, seg = col / C, c = col % C;
    const float* t = target + (size_t)seg * tss + (size_t)c * ldt;
    const float* s = source + (size_t)((src_n_seg == 1) ? 0 : seg) * sss + (size_t)c * lds;
    float* o = out + (size_t)seg * oss + (size_t)c * ldo;
    float* base = (bins <= kBinsLds) ? dyn : gws + (size_t)col * 6 * bins;
    unsigned *ht = reinterpret_cast<unsigned*>(base), *hs = ht + bins;
    float *edges = base + 2 * (size_t)bins, *tc = edges + bins, *sc = tc + bins, *rm = sc + bins;

    // histmatch.py:52-53 joint range
    float lo = INFINITY, hi = -INFINITY;
    for (long i = threadIdx.x; i < nt; i += 256) {
        lo = fminf(lo, t[i]);
        hi = fmaxf(hi, t[i]);
    }
    for (long i = threadIdx.x; i < ns; i += 256) {
        lo = fminf(lo, s[i]);
        hi = fmaxf(hi, s[i]);
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
    }
    for (int k = threadIdx.x; k < 2 * bins; k += 256) ht[k] = 0u;
    __syncthreads();
    lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
    hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));

    // histmatch.py:55-56 torch.histc(x, bins, lo, hi)
    float hl = lo, hu = hi;
    if (hl == hu) {
        hl -= 1.0f;
        hu += 1.0f;
    }
    const float range = hu - hl, fbins = (float)bins;
    for (int which = 0; which < 2; which++) {
        const float* x = which ? s : t;
        const long n = which ? ns : nt;
        unsigned* h = which ? hs : ht;
        for (long i = threadIdx.x; i < n; i += 256) {
            const float v = x[i];
            if (!(v >= hl && v <= hu)) continue;
            int pos = (int)__fdiv_rn((v - hl) * fbins, range);
            pos = pos > bins - 1 ? bins - 1 : pos;
            atomicAdd(&h[pos], 1u);
        }
    }
    __syncthreads();
    // histmatch.py:58-65
    hist_to_cdf(ht, tc, bins, part);
    hist_to_cdf(hs, sc, bins, part);
    const float step = __fdiv_rn(hi - lo, fbins);
    const int half = (bins + 1) / 2;
    for (int k = threadIdx.x; k < bins; k += 256) {
        const int i = k + 1;  // torch.linspace(lo, hi, bins + 1)[1:]
        edges[k] = (i < half) ? __fmaf_rn(step, (float)i, lo) : __fmaf_rn(-step, (float)(bins - i), hi);
    }
    __syncthreads();
    // histmatch.py:67 remapped_cdf = interp(target_cdf, source_cdf, bin_edges)
    for (int k = threadIdx.x; k < bins; k += 256) {
        const float x = tc[k];
        int idx = lower_bound_f(sc, bins, x);
        idx = idx > bins - 1 ? bins - 1 : idx;
        rm[k] = interp_eval(x, idx, sc, edges, bins);
    }
    __syncthreads();
    // histmatch.py:68 interp(target_channel, bin_edges, remapped_cdf); the search starts at the histogram bin and is then made
    // exact (idx = searchsorted_left(edges, x)) by the two walks
    const float seed = (hi - lo > 0.f) ? fbins / (hi - lo) : 0.f;
    for (long i = threadIdx.x; i < nt; i += 256) {
        const float x = t[i];
        const float f = (x - lo) * seed;
        int idx = (f >= 0.f) ? ((f < (float)(bins - 1)) ? (int)f : bins - 1) : 0;
        while (idx > 0 && edges[idx - 1] >= x) idx--;
        while (idx < bins - 1 && !(edges[idx] >= x)) idx++;
        o[i] = interp_eval(x, idx, edges, rm, bins);
    }
}

// ------------------------------------------------------------------------------------------------ generic interp
__global__ void interp_kernel(const float* __restrict__ x, long nx, const float* __restrict__ xp,
                              const float* __restrict__ fp, int np_, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx) return;
    const float v = x[i];
    int idx = lower_bound_f(xp, np_, v);
    idx = idx > np_ - 1 ? np_ - 1 : idx;  // the reference would raise IndexError here (x above every knot)
    out[i] = interp_eval(v, idx, xp, fp, np_);
}

// ------------------------------------------------------------------------------------------------ host side
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static long pick_chunk(long n, int ncols, int n_cu) {
    // aim for >= 8 blocks per CU in flight; chunks are multiples of 1024 elements (256 threads x float4)
    const long want_blocks = 8L * n_cu;
    long chunks = (want_blocks + ncols - 1) / ncols;
    if (chunks < 1) chunks = 1;
    long chunk = (n + chunks - 1) / chunks;
    chunk = (chunk + 1023) / 1024 * 1024;
    if (chunk < 4096) chunk = 4096;
    return chunk;
}

int device_cu_count();

static int launch_minmax(const float* x, long ld, long ss, long n, int C, int n_seg, const float* omn, const float* omx,
                         int o_n_seg, float* mn, float* mx, hipStream_t st) {
    const int ncols = C * n_seg;
    const int vec = aligned16(x) && ld % 4 == 0 && ss % 4 == 0;
    long chunk = pick_chunk(n, ncols, device_cu_count());
    const int chunks = (int)((n + chunk - 1) / chunk);
    ProfScope prof(KC_MINMAX, st, 0.0, 4.0 * (double)n * ncols);
    if (chunks <= 1) {
        hipLaunchKernelGGL(col_minmax_kernel<false>, dim3(ncols, 1), dim3(256), 0, st, x, ld, ss, n, C, n, omn, omx,
                           o_n_seg, mn, mx, vec);
    } else {
        hipLaunchKernelGGL(minmax_init_kernel, dim3((ncols + 255) / 256), dim3(256), 0, st, mn, mx, ncols);
        hipLaunchKernelGGL(col_minmax_kernel<true>, dim3(ncols, chunks), dim3(256), 0, st, x, ld, ss, n, C, chunk, omn,
                           omx, o_n_seg, mn, mx, vec);
        hipLaunchKernelGGL(minmax_decode_kernel, dim3((ncols + 255) / 256), dim3(256), 0, st, mn, mx, ncols);
    }
    return check_launch("col_minmax_kernel");
}

static int launch_hist(const float* x, long ld, long ss, long n, int C, int x_n_seg, int n_seg, const float* lo,
                       const float* hi, unsigned* hist, hipStream_t st) {
    const int ncols = C * n_seg;
    const int vec = aligned16(x) && ld % 4 == 0 && ss % 4 == 0;
    long chunk = pick_chunk(n, ncols, device_cu_count());
    const int chunks = (int)((n + chunk - 1) / chunk);
    if (chunks > 1) {
        hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned) * (size_t)ncols * kBins, st);
        if (e != hipSuccess) {
            set_error("col_hist: memset failed: %s", hipGetErrorString(e));
            return OPTEX_E_LAUNCH;
        }
    }
    // algorithmic bytes: every DISTINCT column once — a shared source (x_n_seg == 1) is binned with each target segment's
    // range (n_seg blocks per channel re-read it through L2) but comes from HBM once
    ProfScope prof(KC_HIST, st, 0.0, 4.0 * (double)n * C * (x_n_seg == 1 ? 1 : n_seg));
    hipLaunchKernelGGL(col_hist_kernel, dim3(ncols, chunks < 1 ? 1 : chunks), dim3(256), 0, st, x, ld, ss, n, C,
                       x_n_seg, chunk, lo, hi, hist, vec);
    return check_launch("col_hist_kernel");
}

// workspace layout of optex_cdf_match (all [n_seg, C, ...]):
struct CdfWs {
    float *smn, *smx;   // source min/max          [src_n_seg <= n_seg, C]
    float *lo, *hi;     // joint range             [n_seg, C]
    unsigned *ht, *hs;  // histograms              [n_seg, C, 256]
    float* lut;         // edges, remapped, slope  [n_seg, C, 3, 256]
    static size_t bytes(int C, int n_seg) {
        const size_t cols = (size_t)C * n_seg;
        return align_up(cols * 4 * sizeof(float), 256) + align_up(cols * 2 * kBins * sizeof(unsigned), 256) +
               align_up(cols * 3 * kBins * sizeof(float), 256);
    }
    CdfWs(void* ws, int C, int n_seg) {
        const size_t cols = (size_t)C * n_seg;
        char* p = static_cast<char*>(ws);
        smn = reinterpret_cast<float*>(p);
        smx = smn + cols;
        lo = smx + cols;
        hi = lo + cols;
        p += align_up(cols * 4 * sizeof(float), 256);
        ht = reinterpret_cast<unsigned*>(p);
        hs = ht + cols * kBins;
        p += align_up(cols * 2 * kBins * sizeof(unsigned), 256);
        lut = reinterpret_cast<float*>(p);
    }
};

int cdf_match_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                   int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, float* dbg,
                   hipStream_t st) {
    return cdf_match_parts_impl(target, ldt, tss, nt, source, lds, sss, ns, src_n_seg, C, n_seg, out, ldo, oss, ws, dbg,
                                nullptr, nullptr, 0, st);
}

// tmn_parts / tmx_parts [n_seg][parts][C]: per-tile min / max of the target the producing GEMM already took (or NULL)
int cdf_match_parts_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                         int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, float* dbg,
                         const float* tmn_parts, const float* tmx_parts, int parts, hipStream_t st) {
    CdfWs w(ws, C, n_seg);
    int rc;
    if ((rc = launch_minmax(source, lds, sss, ns, C, src_n_seg, nullptr, nullptr, 1, w.smn, w.smx, st))) return rc;
    if (tmn_parts) {
        const int ncols = C * n_seg;
        // reads the partials (parts / n of the map's bytes) instead of the map
        ProfScope prof(KC_MINMAX, st, 0.0, 8.0 * (double)parts * ncols);
        hipLaunchKernelGGL(minmax_from_parts_kernel, dim3((ncols + 63) / 64), dim3(256), 0, st, tmn_parts, tmx_parts, parts, C,
                           ncols, w.smn, w.smx, src_n_seg, w.lo, w.hi);
        if ((rc = check_launch("minmax_from_parts_kernel"))) return rc;
    } else if ((rc = launch_minmax(target, ldt, tss, nt, C, n_seg, w.smn, w.smx, src_n_seg, w.lo, w.hi, st))) {
        return rc;
    }
    if ((rc = launch_hist(target, ldt, tss, nt, C, n_seg, n_seg, w.lo, w.hi, w.ht, st))) return rc;
    if ((rc = launch_hist(source, lds, sss, ns, C, src_n_seg, n_seg, w.lo, w.hi, w.hs, st))) return rc;
    const int ncols = C * n_seg;
    {
        ProfScope prof(KC_LUT, st, 0.0, (2.0 * 4 + 3.0 * 4) * kBins * ncols);
        hipLaunchKernelGGL(cdf_lut_kernel, dim3(ncols), dim3(256), 0, st, w.ht, w.hs, w.lo, w.hi, w.lut, dbg);
    }
    if ((rc = check_launch("cdf_lut_kernel"))) return rc;
    const int vec = aligned16(target) && ldt % 4 == 0 && tss % 4 == 0 && aligned16(out) && ldo % 4 == 0 && oss % 4 == 0;
    long chunk = pick_chunk(nt, ncols, device_cu_count());
    const int chunks = (int)((nt + chunk - 1) / chunk);
    ProfScope prof(KC_APPLY, st, 0.0, 8.0 * (double)nt * ncols);
    hipLaunchKernelGGL(cdf_apply_kernel, dim3(ncols, chunks < 1 ? 1 : chunks), dim3(256), 0, st, target, ldt, tss, nt, C,
                       chunk, w.lo, w.hi, w.lut, out, ldo, oss, vec);
    return check_launch("cdf_apply_kernel");
}

}  // namespace optex

using namespace optex;

extern "C" int optex_col_minmax(const float* x, long ld, long seg_stride, long n, int C, int n_seg, float* mn, float* mx,
                                void* stream) {
    if (!x || !mn || !mx || n <= 0 || C <= 0 || n_seg <= 0 || ld < n) {
        set_error("optex_col_minmax: bad argument (n=%ld C=%d n_seg=%d ld=%ld)", n, C, n_seg, ld);
        return OPTEX_E_ARG;
    }
    return launch_minmax(x, ld, seg_stride, n, C, n_seg, nullptr, nullptr, 1, mn, mx, as_stream(stream));
}

extern "C" int optex_col_histc(const float* x, long ld, long seg_stride, long n, int C, int n_seg, const float* lo,
                               const float* hi, uint32_t* hist, void* stream) {
    if (!x || !lo || !hi || !hist || n <= 0 || C <= 0 || n_seg <= 0 || ld < n) {
        set_error("optex_col_histc: bad argument (n=%ld C=%d n_seg=%d ld=%ld)", n, C, n_seg, ld);
        return OPTEX_E_ARG;
    }
    return launch_hist(x, ld, seg_stride, n, C, n_seg, n_seg, lo, hi, hist, as_stream(stream));
}

extern "C" int optex_interp(const float* x, long nx, const float* xp, const float* fp, long np_, float* out,
                            void* stream) {
    if (!x || !xp || !fp || !out || nx < 0 || np_ <= 0 || np_ > 0x7fffffffL) {
        set_error("optex_interp: bad argument (nx=%ld np=%ld)", nx, np_);
        return OPTEX_E_ARG;
    }
    if (nx == 0) return OPTEX_OK;
    ProfScope prof(KC_INTERP, as_stream(stream), 0.0, 8.0 * (double)nx);
    hipLaunchKernelGGL(interp_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, as_stream(stream), x, nx, xp,
                       fp, (int)np_, out);
    return check_launch("interp_kernel");
}

extern "C" size_t optex_cdf_ws_bytes(int C, int n_seg) { return CdfWs::bytes(C, n_seg); }

extern "C" int optex_cdf_match(const float* target, long ldt, long t_seg_stride, long nt, const float* source, long lds,
                               long s_seg_stride, long ns, int src_n_seg, int C, int n_seg, float* out, long ldo,
                               long o_seg_stride, void* ws, size_t ws_bytes, float* dbg, void* stream) {
    if (!target || !source || !out || !ws || nt <= 0 || ns <= 0 || C <= 0 || n_seg <= 0 || ldt < nt || lds < ns ||
        ldo < nt) {
        set_error("optex_cdf_match: bad argument (nt=%ld ns=%ld C=%d n_seg=%d)", nt, ns, C, n_seg);
        return OPTEX_E_ARG;
    }
    if (src_n_seg != 1 && src_n_seg != n_seg) {
        set_error("optex_cdf_match: source has %d segments, expected 1 or %d", src_n_seg, n_seg);
        return OPTEX_E_ARG;
    }
    if (int rc = check_ws("optex_cdf_match", ws, ws_bytes, optex_cdf_ws_bytes(C, n_seg))) return rc;
    return cdf_match_impl(target, ldt, t_seg_stride, nt, source, lds, s_seg_stride, ns, src_n_seg, C, n_seg, out, ldo,
                          o_seg_stride, ws, dbg, as_stream(stream));
}

extern "C" size_t optex_cdf_bins_ws_bytes(int C, int n_seg, int bins) {
    if (bins <= kBinsLds) return 256;
    return (size_t)6 * sizeof(float) * (size_t)bins * (size_t)C * (size_t)n_seg;
}

extern "C" int optex_cdf_match_bins(const float* target, long ldt, long t_seg_stride, long nt, const float* source, long lds,
                                    long s_seg_stride, long ns, int src_n_seg, int C, int n_seg, int bins, float* out,
                                    long ldo, long o_seg_stride, void* ws, size_t ws_bytes, void* stream) {
    if (!target || !source || !out || !ws || nt <= 0 || ns <= 0 || C <= 0 || n_seg <= 0 || ldt < nt || lds < ns ||
        ldo < nt || bins <= 0) {
        set_error("optex_cdf_match_bins: bad argument (nt=%ld ns=%ld C=%d n_seg=%d bins=%d)", nt, ns, C, n_seg, bins);
        return OPTEX_E_ARG;
    }
    if (src_n_seg != 1 && src_n_seg != n_seg) {
        set_error("optex_cdf_match_bins: source has %d segments, expected 1 or %d", src_n_seg, n_seg);
        return OPTEX_E_ARG;
    }
    if (int rc = check_ws("optex_cdf_match_bins", ws, ws_bytes, optex_cdf_bins_ws_bytes(C, n_seg, bins))) return rc;
    const int ncols = C * n_seg;
    const size_t dyn = bins <= kBinsLds ? (size_t)6 * sizeof(float) * bins : 0;
    ProfScope prof(KC_APPLY, as_stream(stream), 0.0, (16.0 * (double)nt + 8.0 * (double)ns) * ncols);
    hipLaunchKernelGGL(cdf_bins_kernel, dim3(ncols), dim3(256), dyn, as_stream(stream), target, ldt, t_seg_stride, nt, source,
                       lds, s_seg_stride, ns, src_n_seg, C, bins, static_cast<float*>(ws), out, ldo, o_seg_stride);
    return check_launch("cdf_bins_kernel");
}
