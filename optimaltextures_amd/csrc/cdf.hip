// cdf.hip — K2a/K2b/K2c/K3: the reference's `cdf` histogram matching (histmatch.py:49-92) over channel-major columns: as
// ONE kernel that keeps a column in registers (cdf_fused_kernel, round 5: columns of up to 16384 values, every batched call of
// the hot loop: 8 bytes per element) or as streaming kernels (min/max, histograms + LUT, apply: longer, chunked or unaligned
// columns: min/max 4, histogram 4, apply 8 bytes per element).  0 flop/B: what matters is that every column is
// read in whole 16-byte vectors, that enough of them are in flight, and that the per-column state (256-bin histograms, the
// LUT as 16-byte entries) lives in LDS.
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
// The bin index is trunc of the IEEE quotient (x - lo) * 256 / range (torch.histc; a multiply by a rounded reciprocal mis-bins
// ~1.4 ppm).  The compiler's correctly rounded division is ~11 instructions per element, most of them of the 4-cycle class
// (v_div_scale x 2, v_rcp, v_div_fmas, v_div_fixup) — and this kernel is VALU-bound on it (round 5: 2.56 M elements per CU per
// launch at 0.27 cycles each = the 290 of its 326 us).  For a divisor b whose correctly rounded reciprocal y = RN(1 / b) is
// known — one true division per column — the quotient takes a multiply and four fmas, all of the 2-cycle class:
//     q0 = RN(a y);  r0 = a - b q0 (exact in the fma);  q1 = RN(q0 + r0 y);  r1 = a - b q1;  q = RN(q1 + r1 y)
// q1 is within half an ulp (+ 2^-24 ulp) of a / b, and the last step then rounds correctly (Markstein's theorem: the fma's
// argument differs from a / b by less than 2^-25 ulp, closer than a quotient of two 24-bit numbers can come to a rounding
// boundary without lying on one of them, which it never does).  Checked against the division itself on 1.2e9 operand
// pairs incl. all-ones mantissas, powers of two and +-4 ulp around every bin edge k * range: 0 differing quotients.  Needs
// normal operands: used when 2^-100 < range < 2^100 (uniform per column), the plain division otherwise.
__device__ __forceinline__ float div_by(float a, float b, float y) {
    const float q0 = __fmul_rn(a, y);
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}
__device__ __forceinline__ bool div_by_ok(float range) { return range > 7.8886091e-31f && range < 1.2676506e30f; }  // 2^-100, 2^100

template <bool FAST = false>
__device__ __forceinline__ void hist_add(unsigned* h, float v, float lo, float hi, float range, float inv = 0.f) {
    // branch-free: a value outside [lo, hi] (NaN included, like ATen) adds 0 to a clamped bin — four independent chains per
    // 16-byte load instead of four exec-mask branches
    const unsigned inc = (v >= lo && v <= hi) ? 1u : 0u;
    const float a = (v - lo) * (float)kBins;
    // clamped in the FLOAT domain (one v_med3_f32; NaN -> 0), then converted: a float -> int cast of NaN or of an out-of-range
    // value is undefined in C++, and the two integer clamps it replaces were one instruction more
    const int pos = (int)__builtin_amdgcn_fmed3f(FAST ? div_by(a, range, inv) : __fdiv_rn(a, range), 0.f, (float)(kBins - 1));
#ifdef CDF_PROBE_NOATOMIC   // scripts/cdf_probe.hip: the binning arithmetic without its LDS atomic
    if (pos == 0x7fffffff) atomicAdd(&h[pos & 255], inc);
#else
    atomicAdd(&h[pos], inc);
#endif
}

template <bool FAST>
__device__ __forceinline__ void hist_add4(unsigned* h, const float4 v, float hl, float hu, float range, float inv) {
    hist_add<FAST>(h, v.x, hl, hu, range, inv);
    hist_add<FAST>(h, v.y, hl, hu, range, inv);
    hist_add<FAST>(h, v.z, hl, hu, range, inv);
    hist_add<FAST>(h, v.w, hl, hu, range, inv);
}

// Four 16-byte loads in flight per thread (round 5: the plain loop compiled to load -> s_waitcnt vmcnt(0) -> 4 atomics, ONE
// kilobyte in flight per wavefront — 32 KB per CU where ~60 KB cover the HBM round trip at full rate)
template <bool FAST, int NT = 256>
__device__ __forceinline__ void hist_chunk_t(unsigned* h, const float* __restrict__ p, long beg, long end, int vec, float hl, float hu,
                                             float range, float inv) {
    const int tid = threadIdx.x;
    if (vec) {  // beg is a multiple of 4 and rows are 16-byte aligned
        const long nv = (end - beg) / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p + beg);
        long i = tid;
        for (; i + 3 * NT < nv; i += 4 * NT) {
            const float4 v0 = p4[i], v1 = p4[i + NT], v2 = p4[i + 2 * NT], v3 = p4[i + 3 * NT];
            hist_add4<FAST>(h, v0, hl, hu, range, inv);
            hist_add4<FAST>(h, v1, hl, hu, range, inv);
            hist_add4<FAST>(h, v2, hl, hu, range, inv);
            hist_add4<FAST>(h, v3, hl, hu, range, inv);
        }
        for (; i < nv; i += NT) hist_add4<FAST>(h, p4[i], hl, hu, range, inv);
        for (long j = beg + nv * 4 + tid; j < end; j += NT) hist_add<FAST>(h, p[j], hl, hu, range, inv);
    } else {
        for (long i = beg + tid; i < end; i += NT) hist_add<FAST>(h, p[i], hl, hu, range, inv);
    }
}
// (the reciprocal is taken once per call: one division per thread and chunk)
template <int NT = 256>
__device__ __forceinline__ void hist_chunk(unsigned* h, const float* __restrict__ p, long beg, long end, int vec, float hl, float hu,
                                           float range) {
    if (div_by_ok(range)) hist_chunk_t<true, NT>(h, p, beg, end, vec, hl, hu, range, __fdiv_rn(1.0f, range));
    else hist_chunk_t<false, NT>(h, p, beg, end, vec, hl, hu, range, 0.f);
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
    hist_chunk(h, p, beg, end, vec, lo, hi, range);   // (blockDim.x == 256)
    __syncthreads();
    unsigned* g = hist + (size_t)col * kBins;
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) {
        const unsigned v = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
        if (gridDim.y == 1) g[i] = v;
        else if (v) atomicAdd(&g[i], v);
    }
}

// ------------------------------------------------------------------------------------------------ K2c CDFs -> LUT
// 256 threads, thread i = bin i.  lut[col] = { bin_edges[256], remapped_cdf[256], slope[256] }
struct LutShared {
    unsigned ct[kBins], cs[kBins], rt[kBins], rs[kBins];
    float tcdf[kBins], scdf[kBins], edges[kBins], rm[kBins];
};

// histmatch.py:59-67 for one column: ht / hs = the two counts of bin i = threadIdx.x; l = the column's LUT in global memory
// (or NULL: the caller keeps it on the CU); d = NULL or the column's debug record.  All 256 threads of the block call it;
// thread i returns with remapped_cdf[i] in `r_out` and the slope of bin i in `slope_out`, S.edges / S.rm hold the tables.
__device__ __forceinline__ void lut_column(LutShared& S, unsigned ht, unsigned hs, float lo, float hi, float* __restrict__ l,
                                           float* __restrict__ d, float& r_out, float& slope_out) {
    // (a 512-thread workgroup calls it with all its threads: thread 256 + i mirrors thread i's arithmetic — it must pass the
    // barriers anyway — and only the lower half writes)
    const int i = threadIdx.x & (kBins - 1), lane = i & 63, w = i >> 6;
    const bool wr = threadIdx.x < kBins;
    if (wr) {
        S.rt[i] = ht;
        S.rs[i] = hs;
    }
    // inclusive scan: inside the wavefront by lane shifts, the three wave totals through LDS (one barrier; the Hillis-Steele
    // scan over LDS this replaces had sixteen); integer, hence exact and equal to torch's fp32 cumsum while totals < 2^24
    unsigned ct = ht, cs = hs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned a = __shfl_up(ct, o), b = __shfl_up(cs, o);
        if (lane >= o) {
            ct += a;
            cs += b;
        }
    }
    if (lane == 63 && wr) {
        S.ct[w] = ct;
        S.cs[w] = cs;
    }
    __syncthreads();
    unsigned tot_t = 0, tot_s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned a = S.ct[k], b = S.cs[k];
        if (k < w) {
            ct += a;
            cs += b;
        }
        tot_t += a;
        tot_s += b;
    }
    float ft = (float)ct, fs = (float)cs;
    float tl = (float)tot_t, sl = (float)tot_s;
    if (tot_t >= (1u << 24) || tot_s >= (1u << 24)) {
        // beyond 2^24 the reference's sequential fp32 cumsum rounds: replay it literally
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < kBins; k++) {
                a = a + (float)S.rt[k];
                b = b + (float)S.rs[k];
                S.tcdf[k] = a;
                S.scdf[k] = b;
            }
        }
        __syncthreads();
        ft = S.tcdf[i];
        fs = S.scdf[i];
        tl = S.tcdf[kBins - 1];
        sl = S.scdf[kBins - 1];
        __syncthreads();
    }
    const float step = __fdiv_rn(hi - lo, (float)kBins);
    if (wr) {
        S.tcdf[i] = __fdiv_rn(ft, tl);
        S.scdf[i] = __fdiv_rn(fs, sl);
        S.edges[i] = linspace_edge(lo, hi, step, i + 1);
    }
    __syncthreads();
    // remapped_cdf = interp(target_cdf, source_cdf, bin_edges)   histmatch.py:67
    const float x = S.tcdf[i];
    int idx = lower_bound_f(S.scdf, kBins, x);
    idx = idx > kBins - 1 ? kBins - 1 : idx;
    const float r = interp_eval(x, idx, S.scdf, S.edges, kBins);
    if (wr) S.rm[i] = r;
    __syncthreads();
    const int nxt = (i + 1 > kBins - 1) ? kBins - 1 : i + 1;
    const float slope = __fdiv_rn(S.rm[nxt] - S.rm[i], S.edges[nxt] - S.edges[i]);
    r_out = r;
    slope_out = slope;
    if (l && wr) {
        l[i] = S.edges[i];
        l[kBins + i] = r;
        l[2 * kBins + i] = slope;
    }
    if (d && wr) {
        if (i == 0) {
            d[0] = lo;
            d[1] = hi;
        }
        d[2 + i] = (float)ht;
        d[2 + kBins + i] = (float)hs;
        d[2 + 2 * kBins + i] = S.edges[i];
        d[2 + 3 * kBins + i] = r;
    }
}

// ------------------------------------------------------------------------------------------------ K2a + K2b + K2c in one launch
// The joint range, both histograms and the LUT of every column in ONE launch (an OT iteration at 8 textures per step is
// launch-bound: range, two histograms with their clears and the LUT were seven launches of 5-30 us each):
//   * range: every block folds the column's min / max itself — from the per-tile partials the rotation GEMM's epilogue left
//     (pmn / pmx) joined with the source's range (smn / smx), or reads the joint range somebody computed (lo / hi);
//     min / max do not depend on the order, so every block of a column gets the same bits;
//   * grid = (columns, 1), the case of every batched call (one block per column fills the chip): the block bins the target
//     column, then the source column, and goes straight on to the LUT — nothing but the LUT leaves the CU;
//   * grid = (columns, chunks_t + chunks_s), few long columns: block (col, y) bins one chunk of the target (y < chunks_t) or of
//     the source and stores its 256 counts in its own slot of `part`; the LAST block of a column to finish (one ticket atomic
//     per block; the counter resets itself) adds the slots up — integers, exact in any order — and computes the LUT.
//     (First version: atomics on one global histogram per column.  Device-scope atomics are served behind the per-XCD L2s:
//     8 M of them per launch took 5 ms, the whole step went from 320 to 597 ms.)
struct HistLutArgs {
    const float* t; long ldt, tss, nt;
    const float* s; long lds, sss, ns; int src_n_seg;
    int C; long chunk_t; int chunks_t; long chunk_s; int chunks_s;
    const float* pmn; const float* pmx; int parts;     // target min / max partials [n_seg][parts][C], or NULL:
    const float* smn; const float* smx;                // source min / max [src_n_seg, C] (joined with the partials)
    float* lo; float* hi;                              // joint range [ncols]: read if pmn == NULL, written for the apply kernel
    unsigned* part; unsigned* done;                    // multi-chunk only: [ncols][chunks_t + chunks_s][256] and the tickets
    // optional: the source columns' histograms over their OWN range [smn, smx], [src_n_seg, C, 256].  A column whose joint
    // range IS the source's range (the source's range contains the target's: the usual case once the pastiche has been matched
    // a few times) takes them instead of binning the source again — the same counts, bin for bin, since the same range gives
    // the same arithmetic; optex_ot_loop computes them once per (iteration, channel) for all textures of a batch.
    const unsigned* shist;
    float* lut; float* dbg;
    int vec_t, vec_s;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void cdf_hist_lut_kernel(HistLutArgs a) {
    const int col = blockIdx.x, seg = col / a.C, c = col % a.C, tid = threadIdx.x;
    __shared__ unsigned sh[4][kBins];
    __shared__ float slo[4], shi[4];
    __shared__ unsigned s_last;
    __shared__ LutShared S;
    for (int i = tid; i < 4 * kBins; i += 256) (&sh[0][0])[i] = 0u;
    float lo, hi;
    bool src_range = false;   // the joint range is the source's own range: its precomputed histogram applies
    if (a.pmn) {
        lo = INFINITY;
        hi = -INFINITY;
        const float* pa = a.pmn + (size_t)seg * a.parts * a.C + c;
        const float* pb = a.pmx + (size_t)seg * a.parts * a.C + c;
        for (int p = tid; p < a.parts; p += 256) {
            lo = fminf(lo, pa[(size_t)p * a.C]);
            hi = fmaxf(hi, pb[(size_t)p * a.C]);
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        if ((tid & 63) == 0) {
            slo[tid >> 6] = lo;
            shi[tid >> 6] = hi;
        }
        __syncthreads();
        const int oc = ((a.src_n_seg == 1) ? 0 : seg) * a.C + c;
        const float smn = a.smn[oc], smx = a.smx[oc];
        lo = fminf(fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3])), smn);   // histmatch.py:52-53
        hi = fmaxf(fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3])), smx);
        src_range = a.shist != nullptr && lo == smn && hi == smx;   // uniform over the block
    } else {
        lo = a.lo[col];
        hi = a.hi[col];
        __syncthreads();
    }
    float hl = lo, hu = hi;
    if (hl == hu) {  // torch.histc widens an empty range
        hl -= 1.0f;
        hu += 1.0f;
    }
    const float range = hu - hl;
    const float* pt = a.t + (size_t)seg * a.tss + (size_t)c * a.ldt;
    const float* ps = a.s + (size_t)((a.src_n_seg == 1) ? 0 : seg) * a.sss + (size_t)c * a.lds;
    float* l = a.lut + (size_t)col * 3 * kBins;
    float* d = a.dbg ? a.dbg + (size_t)col * (2 + 4 * kBins) : nullptr;
    unsigned* h = sh[tid >> 6];
    if (gridDim.y == 1) {
        // the whole column pair in this block: target counts to registers, then the source, then the LUT
        hist_chunk(h, pt, 0, a.nt, a.vec_t, hl, hu, range);
        __syncthreads();
        const unsigned ht = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
        __syncthreads();
        unsigned hs;
        if (src_range) {
            hs = a.shist[((size_t)((a.src_n_seg == 1) ? 0 : seg) * a.C + c) * kBins + tid];
        } else {
            for (int i = tid; i < 4 * kBins; i += 256) (&sh[0][0])[i] = 0u;
            __syncthreads();
            hist_chunk(h, ps, 0, a.ns, a.vec_s, hl, hu, range);
            __syncthreads();
            hs = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
        }
        if (tid == 0) {
            a.lo[col] = lo;
            a.hi[col] = hi;
        }
        float r_, sl_;
        lut_column(S, ht, hs, lo, hi, l, d, r_, sl_);
        return;
    }
    const int nblk = a.chunks_t + a.chunks_s, y = (int)blockIdx.y;
    if (y < a.chunks_t) {
        const long beg = (long)y * a.chunk_t;
        hist_chunk(h, pt, beg, (beg + a.chunk_t < a.nt) ? beg + a.chunk_t : a.nt, a.vec_t, hl, hu, range);
    } else {
        const long beg = (long)(y - a.chunks_t) * a.chunk_s;
        hist_chunk(h, ps, beg, (beg + a.chunk_s < a.ns) ? beg + a.chunk_s : a.ns, a.vec_s, hl, hu, range);
    }
    __syncthreads();
    unsigned* mine = a.part + ((size_t)col * nblk + y) * kBins;
    __hip_atomic_store(mine + tid, sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();   // this block's counts are visible device-wide before its ticket is
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&a.done[col], 1u) == (unsigned)(nblk - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    unsigned ht = 0u, hs = 0u;
    const unsigned* base = a.part + (size_t)col * nblk * kBins + tid;
    for (int k = 0; k < a.chunks_t; k++) ht += __hip_atomic_load(base + (size_t)k * kBins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int k = a.chunks_t; k < nblk; k++) hs += __hip_atomic_load(base + (size_t)k * kBins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        a.done[col] = 0u;   // the next launch finds its tickets at zero
        a.lo[col] = lo;
        a.hi[col] = hi;
    }
    float r_, sl_;
    lut_column(S, ht, hs, lo, hi, l, d, r_, sl_);
}

// ------------------------------------------------------------------------------------------------ K3 apply
// out = interp(target_channel, bin_edges, remapped_cdf)   histmatch.py:68
// The LUT sits in LDS as one 16-byte entry per bin: T[idx] = (edges[idx - 1] (NaN for idx = 0), edges[idx], remapped[idx],
// slope[idx]) — the candidate bin's entry answers "is this searchsorted_left(edges, x)?" AND carries everything the
// interpolation needs, ONE ds_read_b128 per element where three separate tables took five to six ds_read_b32 (round 5: the
// kernel was bound by those reads, 0.65 LDS cycles per element against 0.61 for the HBM stream at peak).
__device__ __forceinline__ float lut_apply(float x, float lo, float range256, const float4* T) {
    // candidate bin from the histogram formula, then an exact fix-up to idx = searchsorted_left(edges, x)
    int idx = 0;
    if (range256 > 0.f) {
        const float t = (x - lo) * range256;
        idx = (t >= 0.f) ? ((t < 255.f) ? (int)t : 255) : 0;
    }
    float4 q = T[idx];
    // (the common case is evaluated straight from the candidate's entry, all four components consumed at once: with the
    // evaluation behind the fix-up branch the compiler split the entry into four separate LDS reads)
    float f = __fadd_rn(__fmul_rn(q.w, x - q.y), q.z);
    if (q.x >= x || (idx < kBins - 1 && !(q.y >= x))) {   // rare: the candidate is a neighbour of the bin (or x is NaN)
        while (idx > 0 && T[idx].x >= x) idx--;
        while (idx < kBins - 1 && !(T[idx].y >= x)) idx++;
        q = T[idx];
        f = __fadd_rn(__fmul_rn(q.w, x - q.y), q.z);
    }
    if (!finite_f(f)) {
        const float4 q2 = T[(idx + 1 > kBins - 1) ? kBins - 1 : idx + 1];
        const float f2 = __fadd_rn(__fmul_rn(q.w, x - q2.y), q2.z);
        f = finite_f(f2) ? f2 : q.z;
    }
    return f;
}
// candidate bin of x from the histogram formula (only seeds the search)
__device__ __forceinline__ int lut_guess(float x, float lo, float range256) {
    int idx = 0;
    if (range256 > 0.f) {
        const float t = (x - lo) * range256;
        idx = (t >= 0.f) ? ((t < 255.f) ? (int)t : 255) : 0;
    }
    return idx;
}
// the rest of lut_apply given the candidate's entry q
__device__ __forceinline__ float lut_finish(float x, int idx, float4 q, const float4* T) {
    float f = __fadd_rn(__fmul_rn(q.w, x - q.y), q.z);
    if (q.x >= x || (idx < kBins - 1 && !(q.y >= x))) {   // rare: the candidate is a neighbour of the bin (or x is NaN)
        while (idx > 0 && T[idx].x >= x) idx--;
        while (idx < kBins - 1 && !(T[idx].y >= x)) idx++;
        q = T[idx];
        f = __fadd_rn(__fmul_rn(q.w, x - q.y), q.z);
    }
    if (!finite_f(f)) {
        const float4 q2 = T[(idx + 1 > kBins - 1) ? kBins - 1 : idx + 1];
        const float f2 = __fadd_rn(__fmul_rn(q.w, x - q2.y), q2.z);
        f = finite_f(f2) ? f2 : q.z;
    }
    return f;
}
// Four elements: their four candidate entries as four ds_read_b128 issued back to back.  Written in C++ the compiler splits
// every entry into three or four narrower reads (it wants q.x / q.y early for the branch and sinks the rest): 13 LDS
// instructions per 16-byte vector instead of four, on the pipe that bounds this step.
typedef float lut_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lut_apply4(const float4 v, float lo, float range256, const float4* T) {
    const int i0 = lut_guess(v.x, lo, range256), i1 = lut_guess(v.y, lo, range256), i2 = lut_guess(v.z, lo, range256),
              i3 = lut_guess(v.w, lo, range256);
    const unsigned base = (unsigned)(uintptr_t)T;   // (the low half of a flat LDS address is the LDS byte offset)
    lut_f4 q0, q1, q2, q3;
    asm volatile(
        "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
        : "v"(base + 16u * (unsigned)i0), "v"(base + 16u * (unsigned)i1), "v"(base + 16u * (unsigned)i2), "v"(base + 16u * (unsigned)i3)
        : "memory");
    float4 r;
    r.x = lut_finish(v.x, i0, make_float4(q0[0], q0[1], q0[2], q0[3]), T);
    r.y = lut_finish(v.y, i1, make_float4(q1[0], q1[1], q1[2], q1[3]), T);
    r.z = lut_finish(v.z, i2, make_float4(q2[0], q2[1], q2[2], q2[3]), T);
    r.w = lut_finish(v.w, i3, make_float4(q3[0], q3[1], q3[2], q3[3]), T);
    return r;
}

__global__ __launch_bounds__(256) void cdf_apply_kernel(const float* __restrict__ x, long ld, long seg_stride, long n,
                                                        int C, long chunk, const float* __restrict__ lo_,
                                                        const float* __restrict__ hi_, const float* __restrict__ lut,
                                                        float* __restrict__ out, long ldo, long o_seg_stride, int vec) {
    const int col = blockIdx.x, seg = col / C, c = col % C;
    __shared__ float4 T[kBins];
    const float* l = lut + (size_t)col * 3 * kBins;
    for (int i = threadIdx.x; i < kBins; i += blockDim.x)
        T[i] = make_float4(i > 0 ? l[i - 1] : __uint_as_float(0x7fc00000u), l[i], l[kBins + i], l[2 * kBins + i]);
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
        long i = threadIdx.x;
        for (; i + 768 < nv; i += 1024) {   // four 16-byte loads in flight per thread (all read before the first store: in place is safe)
            const float4 v0 = p4[i], v1 = p4[i + 256], v2 = p4[i + 512], v3 = p4[i + 768];
            o4[i] = lut_apply4(v0, lo, range256, T);
            o4[i + 256] = lut_apply4(v1, lo, range256, T);
            o4[i + 512] = lut_apply4(v2, lo, range256, T);
            o4[i + 768] = lut_apply4(v3, lo, range256, T);
        }
        for (; i < nv; i += 256) o4[i] = lut_apply4(p4[i], lo, range256, T);
        for (long j = beg + nv * 4 + threadIdx.x; j < end; j += 256) o[j] = lut_apply(p[j], lo, range256, T);
    } else {
        for (long i = beg + threadIdx.x; i < end; i += 256) o[i] = lut_apply(p[i], lo, range256, T);
    }
}

// ------------------------------------------------------------------------------------------------ K2 + K3 in one launch
// Range, both histograms, the LUT AND the interpolation of a column in one workgroup that keeps the column in REGISTERS
// (round 5): a column of the hot loop is at most 16384 values, so the target is read from HBM once — binned from the
// registers, matched from the registers, stored — instead of once by the histogram kernel and once more by the apply kernel:
// 8 bytes per element instead of 12 (24 instead of 28 per element and iteration of the whole cdf step), three launches per
// iteration instead of four.  The LUT never leaves the CU.  One workgroup per column, several columns per CU in different
// phases (what the arithmetic of one hides is the memory phase of another; profiles/r05_cdf_probe.md):
//   * a PERSISTENT, double-buffered variant (next column requested before the current one is worked on, two workgroups per CU)
//     was built and measured slower — 640 against 514 us at 16384 values per column: a column's phases are a serial chain of
//     ~20 us on one workgroup, and two workgroups per CU overlap less than four;
//   * the range comes in as ONE pair per column (a.lo / a.hi, or the source range joined with a folded pair): folding the GEMM's
//     per-tile partials in here — 256 strided 4-byte reads per column, each a 64-byte line — cost 100 us per launch at 64
//     textures per step (617 against 514 us); the fold is a 12 us kernel of its own (minmax_from_parts_kernel, coalesced).
// Same arithmetic as cdf_hist_lut_kernel + cdf_apply_kernel, statement for statement (they stay: columns longer than 16384
// values or cut into chunks, unaligned rows).  grid = (columns); nt % 4 == 0, nt <= 4 NT NV, rows 16-byte aligned.
template <int NV, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) void cdf_fused_kernel(HistLutArgs a, float* __restrict__ out, long ldo, long oss) {
    constexpr int NW = NT / 64;
    const int col = blockIdx.x, seg = col / a.C, c = col - seg * a.C, tid = threadIdx.x;
    __shared__ unsigned sh[NW][kBins];
    __shared__ LutShared S;
    __shared__ float4 T[kBins];
    const int nv = (int)(a.nt / 4);
    const float4* p4 = reinterpret_cast<const float4*>(a.t + (size_t)seg * a.tss + (size_t)c * a.ldt);
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (tid + NT * k < nv) v[k] = p4[tid + NT * k];
    for (int i = tid; i < NW * kBins; i += NT) (&sh[0][0])[i] = 0u;
    float lo = a.lo[col], hi = a.hi[col];
    bool src_range = false;
    {  // the target's range (folded from the GEMM's partials) joined with the source's   histmatch.py:52-53
        const int oc = ((a.src_n_seg == 1) ? 0 : seg) * a.C + c;
        const float smn = a.smn[oc], smx = a.smx[oc];
        lo = fminf(lo, smn);
        hi = fmaxf(hi, smx);
        src_range = a.shist != nullptr && lo == smn && hi == smx;   // uniform over the block
    }
    __syncthreads();
    float hl = lo, hu = hi;
    if (hl == hu) {  // torch.histc widens an empty range
        hl -= 1.0f;
        hu += 1.0f;
    }
    const float range = hu - hl;
    unsigned* h = sh[tid >> 6];
    const int bin = tid & (kBins - 1);
    const bool fast = div_by_ok(range);
    const float inv = fast ? __fdiv_rn(1.0f, range) : 0.f;
    unsigned hs = 0;
    if (!src_range) {
        // The column's range sticks out of the style's: the style's counts over the JOINT range are binned here (the shared
        // histogram is over the style's own range).  Round 6: FIRST, while the target column is still on its way from HBM, four
        // 16-byte loads in flight and the reciprocal-refinement quotient (the same counts as IEEE division) — it used to run behind
        // the target's histogram as a one-load-in-flight loop with fdiv: 555-580 us per [64 x 256 x 16384] launch against 430-440
        // for columns that take the shared histogram, and inside optex_ot_loop MOST columns stick out (a column matched under the
        // previous rotation is not inside the style's range under the next): the bench's "in-loop gap" of the fused matcher.
        // (Also built and measured, not kept: the style sorted once per call and thread k bisecting for the first value of bin k —
        // the same counts, no binning at all, and SLOWER: 616 against 541 us, fourteen dependent L2 reads per workgroup are not
        // hidden by its three or four neighbours; profiles/r06_cdf_inloop.md.)
        const float* ps = a.s + (size_t)((a.src_n_seg == 1) ? 0 : seg) * a.sss + (size_t)c * a.lds;
        if (a.vec_s) {
            const float4* s4 = reinterpret_cast<const float4*>(ps);
            const int nvs = (int)(a.ns / 4);
            int i = tid;
            if (fast) {
                // loads in flight: four where the column's own registers set the budget anyway (16 vectors per thread), two
                // for the short columns (their occupancy is worth more than the deeper queue)
                constexpr int SU = NV >= 16 ? 4 : 2;
#pragma unroll 1
                for (; i + (SU - 1) * NT < nvs; i += SU * NT) {
                    float4 q[SU];
#pragma unroll
                    for (int u = 0; u < SU; u++) q[u] = s4[i + u * NT];
#pragma unroll
                    for (int u = 0; u < SU; u++) {
                        hist_add4<true>(h, q[u], hl, hu, range, inv);
                        asm volatile("" ::: "memory");  // (one vector's four chains at a time)
                    }
                }
#pragma unroll 1
                for (; i < nvs; i += NT) hist_add4<true>(h, s4[i], hl, hu, range, inv);
            } else {
#pragma unroll 1
                for (; i < nvs; i += NT) hist_add4<false>(h, s4[i], hl, hu, range, 0.f);
            }
#pragma unroll 1
            for (long j = 4L * nvs + tid; j < a.ns; j += NT) hist_add<false>(h, ps[j], hl, hu, range);
        } else {
#pragma unroll 1
            for (long j = tid; j < a.ns; j += NT) hist_add<false>(h, ps[j], hl, hu, range);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NW; k++) hs += sh[k][bin];
        __syncthreads();
        for (int i = tid; i < NW * kBins; i += NT) (&sh[0][0])[i] = 0u;
        __syncthreads();
    }
#ifdef CDF_PROBE_NOHIST
    if (range == 12345.f) {
#else
    if (fast) {
#endif
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (tid + NT * k < nv) hist_add4<true>(h, v[k], hl, hu, range, inv);
    } else {
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (tid + NT * k < nv) hist_add4<false>(h, v[k], hl, hu, range, 0.f);
    }
    __syncthreads();
    unsigned ht = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) ht += sh[k][bin];
    if (src_range) hs = a.shist[((size_t)((a.src_n_seg == 1) ? 0 : seg) * a.C + c) * kBins + bin];
    if (tid == 0) {   // (the joint range: the two-kernel pipeline leaves it here too)
        a.lo[col] = lo;
        a.hi[col] = hi;
    }
    float* d = a.dbg ? a.dbg + (size_t)col * (2 + 4 * kBins) : nullptr;
    float r, slope;
#ifdef CDF_PROBE_NOLUT
    r = (float)(ht + hs);
    slope = 1.f;
    if (tid < kBins) S.edges[tid] = lo + (float)(tid + 1) * (hi - lo) * (1.f / 256.f);
    __syncthreads();
#else
    lut_column(S, ht, hs, lo, hi, nullptr, d, r, slope);
#endif
    if (tid < kBins) T[tid] = make_float4(tid > 0 ? S.edges[tid - 1] : __uint_as_float(0x7fc00000u), S.edges[tid], r, slope);
    __syncthreads();
    const float arange = hi - lo;
    const float range256 = (arange > 0.f) ? 256.f / arange : 0.f;
    float4* o4 = reinterpret_cast<float4*>(out + (size_t)seg * oss + (size_t)c * ldo);
#pragma unroll
    for (int k = 0; k < NV; k++) {
#ifdef CDF_PROBE_NOAPPLY
        if (tid + NT * k < nv) o4[tid + NT * k] = v[k] * T[k].z;
#else
        if (tid + NT * k < nv) o4[tid + NT * k] = lut_apply4(v[k], lo, range256, T);
#endif
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

static int cdf_fused_enabled = 1;   // optex_cdf_fused (include/optex.h, ABI 9)

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
        if (int rc = device_fill_u32(hist, 0u, (size_t)ncols * kBins, st)) return rc;
    }
    // algorithmic bytes: every DISTINCT column once — a shared source (x_n_seg == 1) is binned with each target segment's
    // range (n_seg blocks per channel re-read it through L2) but comes from HBM once
    ProfScope prof(KC_HIST, st, 0.0, 4.0 * (double)n * C * (x_n_seg == 1 ? 1 : n_seg));
    hipLaunchKernelGGL(col_hist_kernel, dim3(ncols, chunks < 1 ? 1 : chunks), dim3(256), 0, st, x, ld, ss, n, C,
                       x_n_seg, chunk, lo, hi, hist, vec);
    return check_launch("col_hist_kernel");
}

// chunks a column of any length is cut into when `ncols` columns share the chip (pick_chunk): never more than this
static int max_chunks(int ncols, int n_cu) {
    const long want_blocks = 8L * n_cu;
    const long k = (want_blocks + ncols - 1) / ncols;
    return k < 1 ? 1 : (int)k;
}

// workspace layout of optex_cdf_match (all [n_seg, C, ...]):
struct CdfWs {
    float *smn, *smx;   // source min/max          [src_n_seg <= n_seg, C]
    float *lo, *hi;     // joint range             [n_seg, C]
    unsigned* done;     // tickets                 [n_seg, C]   zero between launches of cdf_hist_lut_kernel
    unsigned* part;     // per-block histograms    [n_seg, C, chunks_t + chunks_s, 256] (columns cut into chunks only)
    float* lut;         // edges, remapped, slope  [n_seg, C, 3, 256]
    size_t cols;
    // sized for kMaxCuForWs compute units, NOT for the current device: optex_cdf_ws_bytes / optex_ot_loop_ws_bytes are pure
    // functions of their arguments (include/optex.h), whatever device is current when a caller sizes its scratch.
    // A launch cuts columns into at most max_chunks(cols, real CU count) <= this many chunks.
    static constexpr int kMaxCuForWs = 512;
    static size_t part_words(size_t cols) {
        const int k = max_chunks((int)cols, kMaxCuForWs);
        return k <= 1 ? 0 : cols * 2 * (size_t)k * kBins;
    }
    static size_t bytes(int C, int n_seg) {
        const size_t cols = (size_t)C * n_seg;
        return align_up(cols * 5 * sizeof(float), 256) + align_up(part_words(cols) * sizeof(unsigned), 256) +
               align_up(cols * 3 * kBins * sizeof(float), 256);
    }
    CdfWs(void* ws, int C, int n_seg) {
        cols = (size_t)C * n_seg;
        char* p = static_cast<char*>(ws);
        smn = reinterpret_cast<float*>(p);
        smx = smn + cols;
        lo = smx + cols;
        hi = lo + cols;
        done = reinterpret_cast<unsigned*>(hi + cols);
        p += align_up(cols * 5 * sizeof(float), 256);
        part = reinterpret_cast<unsigned*>(p);
        p += align_up(part_words(cols) * sizeof(unsigned), 256);
        lut = reinterpret_cast<float*>(p);
    }
};

int cdf_ws_clear(void* ws, int C, int n_seg, hipStream_t st) {
    CdfWs w(ws, C, n_seg);
    return device_fill_u32(w.done, 0u, w.cols, st);
}

int col_minmax_launch(const float* x, long ld, long ss, long n, int C, int n_seg, float* mn, float* mx, hipStream_t st) {
    return launch_minmax(x, ld, ss, n, C, n_seg, nullptr, nullptr, 1, mn, mx, st);
}

// torch.histc(column, 256, lo[col], hi[col]) of every column of n_seg segments, hist [n_seg, C, 256]
int col_hist_launch(const float* x, long ld, long ss, long n, int C, int n_seg, const float* lo, const float* hi, unsigned* hist,
                    hipStream_t st) {
    return launch_hist(x, ld, ss, n, C, n_seg, n_seg, lo, hi, hist, st);
}

#ifndef CDF_FUSED_NT_BIG
#define CDF_FUSED_NT_BIG 256   // threads per column above 8192 values: 256 measured faster than 512 at every size (503 against 588 us at
                               // 16384 values, scripts/cdf_probe.hip builds both)
#endif
template <int NV, int NT>
static int launch_fused_t(const HistLutArgs& a, float* out, long ldo, long oss, int ncols, hipStream_t st) {
    hipLaunchKernelGGL((cdf_fused_kernel<NV, NT>), dim3((unsigned)ncols), dim3(NT), 0, st, a, out, ldo, oss);
    return check_launch("cdf_fused_kernel");
}

static int launch_fused(const HistLutArgs& a, float* out, long ldo, long oss, int ncols, hipStream_t st) {
    const long nv = a.nt / 4;
    if (nv <= 2048 || CDF_FUSED_NT_BIG == 256) {
        const int per = (int)((nv + 255) / 256);
        if (per <= 2) return launch_fused_t<2, 256>(a, out, ldo, oss, ncols, st);
        if (per <= 4) return launch_fused_t<4, 256>(a, out, ldo, oss, ncols, st);
        if (per <= 6) return launch_fused_t<6, 256>(a, out, ldo, oss, ncols, st);
        if (per <= 8) return launch_fused_t<8, 256>(a, out, ldo, oss, ncols, st);
        if (per <= 12) return launch_fused_t<12, 256>(a, out, ldo, oss, ncols, st);
        return launch_fused_t<16, 256>(a, out, ldo, oss, ncols, st);
    }
#if CDF_FUSED_NT_BIG == 512   // (probe build only: the library has no 512-thread instantiation)
    const int per = (int)((nv + 511) / 512);
    if (per <= 6) return launch_fused_t<6, 512>(a, out, ldo, oss, ncols, st);
    return launch_fused_t<8, 512>(a, out, ldo, oss, ncols, st);
#else
    set_error("cdf_fused_kernel: no instantiation for %ld values per column", a.nt);
    return OPTEX_E_UNSUPPORTED;
#endif
}

int cdf_match_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                   int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, float* dbg,
                   hipStream_t st) {
    return cdf_match_parts_impl(target, ldt, tss, nt, source, lds, sss, ns, src_n_seg, C, n_seg, out, ldo, oss, ws, dbg,
                                nullptr, nullptr, 0, st);
}

// tmn_parts / tmx_parts [n_seg][parts][C]: per-tile min / max of the target the producing GEMM already took (or NULL);
// smn_given / smx_given [src_n_seg, C]: the source's min / max when the caller has them (optex_ot_loop takes them for all
// iterations of a call in one launch), else they are taken here; ws_clean: the caller cleared the scratch's counters
// (cdf_ws_clear) — the pipeline leaves them clear, so a loop clears them once.
int cdf_match_parts_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                         int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, float* dbg,
                         const float* tmn_parts, const float* tmx_parts, int parts, hipStream_t st, const float* smn_given,
                         const float* smx_given, bool ws_clean, const unsigned* shist_given) {
    CdfWs w(ws, C, n_seg);
    const int ncols = C * n_seg, n_cu = device_cu_count() < CdfWs::kMaxCuForWs ? device_cu_count() : CdfWs::kMaxCuForWs;
    int rc;
    if (!ws_clean && (rc = device_fill_u32(w.done, 0u, w.cols, st))) return rc;
    const float *smn = smn_given, *smx = smx_given;
    if (!smn || !smx) {
        if ((rc = launch_minmax(source, lds, sss, ns, C, src_n_seg, nullptr, nullptr, 1, w.smn, w.smx, st))) return rc;
        smn = w.smn;
        smx = w.smx;
    }
    if (!tmn_parts) {  // no partials: the joint range from a pass over the target (histmatch.py:52-53)
        if ((rc = launch_minmax(target, ldt, tss, nt, C, n_seg, smn, smx, src_n_seg, w.lo, w.hi, st))) return rc;
    }
    HistLutArgs a;
    a.t = target; a.ldt = ldt; a.tss = tss; a.nt = nt;
    a.s = source; a.lds = lds; a.sss = sss; a.ns = ns; a.src_n_seg = src_n_seg;
    a.C = C;
    a.chunk_t = pick_chunk(nt, ncols, n_cu);
    a.chunks_t = (int)((nt + a.chunk_t - 1) / a.chunk_t);
    a.chunk_s = pick_chunk(ns, ncols, n_cu);
    a.chunks_s = (int)((ns + a.chunk_s - 1) / a.chunk_s);
    a.pmn = tmn_parts; a.pmx = tmn_parts ? tmx_parts : nullptr; a.parts = parts;
    a.smn = smn; a.smx = smx;
    a.lo = w.lo; a.hi = w.hi; a.part = w.part; a.done = w.done; a.lut = w.lut; a.dbg = dbg;
    a.shist = (smn_given && smx_given) ? shist_given : nullptr;   // (they go with the given source range)
    a.vec_t = aligned16(target) && ldt % 4 == 0 && tss % 4 == 0;
    a.vec_s = aligned16(source) && lds % 4 == 0 && sss % 4 == 0;
    const int vec = aligned16(target) && ldt % 4 == 0 && tss % 4 == 0 && aligned16(out) && ldo % 4 == 0 && oss % 4 == 0;
    const int blocks_y = (a.chunks_t == 1 && a.chunks_s == 1) ? 1 : a.chunks_t + a.chunks_s;
    // algorithmic bytes of the histogram stage: every DISTINCT column once — a shared source (src_n_seg == 1) is binned with
    // each target segment's range (n_seg blocks per channel re-read it through L2) but comes from HBM once — and the 3 KB LUT
    // per column that reaches HBM (the histograms, CDFs and edges stay in LDS)
    const double hist_bytes = 4.0 * ((double)nt * ncols + (double)ns * C * src_n_seg) + 3.0 * 4 * kBins * ncols;
    if ((tl_call.cdf_two_kernel >= 0 ? tl_call.cdf_two_kernel == 0 : cdf_fused_enabled != 0) && blocks_y == 1 && vec && a.vec_t && nt % 4 == 0 && nt <= 16384 && ns <= 65536) {
        // the whole matcher in one launch, the column in registers: the target is read once and written once (KC_CDF_FUSED)
        if (tmn_parts) {   // the target's own range, one pair per column (joined with the source's inside the kernel)
            if ((rc = minmax_fold_parts(tmn_parts, tmx_parts, parts, C, ncols, w.lo, w.hi, st))) return rc;
            a.pmn = a.pmx = nullptr;
        }   // (without partials w.lo / w.hi hold the joint range already: joining it with the source's range again changes nothing)
        ProfScope prof(KC_CDF_FUSED, st, 0.0, 8.0 * (double)nt * ncols + 4.0 * (double)ns * C * src_n_seg);
        return launch_fused(a, out, ldo, oss, ncols, st);
    }
    {
        ProfScope prof(KC_HIST, st, 0.0, hist_bytes);
        hipLaunchKernelGGL(cdf_hist_lut_kernel, dim3(ncols, blocks_y), dim3(256), 0, st, a);
    }
    if ((rc = check_launch("cdf_hist_lut_kernel"))) return rc;
    long chunk = pick_chunk(nt, ncols, n_cu);
    const int chunks = (int)((nt + chunk - 1) / chunk);
    ProfScope prof(KC_APPLY, st, 0.0, 8.0 * (double)nt * ncols);
    hipLaunchKernelGGL(cdf_apply_kernel, dim3(ncols, chunks < 1 ? 1 : chunks), dim3(256), 0, st, target, ldt, tss, nt, C,
                       chunk, w.lo, w.hi, w.lut, out, ldo, oss, vec);
    return check_launch("cdf_apply_kernel");
}

int minmax_fold_parts(const float* pmn, const float* pmx, int parts, int C, int ncols, float* mn, float* mx, hipStream_t st) {
    ProfScope prof(KC_MINMAX, st, 0.0, 8.0 * (double)parts * ncols);
    hipLaunchKernelGGL(minmax_from_parts_kernel, dim3((ncols + 63) / 64), dim3(256), 0, st, pmn, pmx, parts, C, ncols, nullptr,
                       nullptr, 1, mn, mx);
    return check_launch("minmax_from_parts_kernel");
}

}  // namespace optex

using namespace optex;

extern "C" int optex_cdf_fused(int on) {
    const int old = cdf_fused_enabled;
    cdf_fused_enabled = on != 0;
    return old;
}

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
                               long o_seg_stride, void* ws, size_t ws_bytes, float* dbg, unsigned flags, void* stream) {
    CallScope call_scope(flags);
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
