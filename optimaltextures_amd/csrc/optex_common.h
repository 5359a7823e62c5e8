// optex_common.h — shared helpers for the gfx950 kernels behind include/optex.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/optex.h"

namespace optex {

constexpr int kBins = OPTEX_BINS;
constexpr int kWave = 64;       // CDNA4 wavefront
constexpr int kNumXcd = 8;      // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only, never correctness)

void set_error(const char* fmt, ...);
int check_launch(const char* what);
// caller-owned scratch: refuse a buffer smaller than the *_ws_bytes helper asks for (include/optex.h, ABI 3)
int check_ws(const char* fn, const void* ws, size_t have, size_t need);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Per-call options (the `flags` word of include/optex.h, ABI 10), held THREAD-LOCALLY for the duration of one extern "C" call:
// the launchers underneath read them, an entry point called from inside another one (flags = 0) inherits the outer call's.
// -1 = not given: the library default (for two of them a deprecated process-wide setter can still move the default).
struct CallOpts { int spare_cus = -1, cdf_two_kernel = -1, sort_rank4 = -1; };
extern thread_local CallOpts tl_call;
struct CallScope {
    CallOpts saved;
    explicit CallScope(unsigned flags) : saved(tl_call) {
        if (flags & 0xffu) tl_call.spare_cus = (int)(flags & 0xffu) - 1;
        if (flags & OPTEX_F_CDF_TWO_KERNEL) tl_call.cdf_two_kernel = 1;
        if (flags & OPTEX_F_SORT_RANK4) tl_call.sort_rank4 = 1;
    }
    ~CallScope() { tl_call = saved; }
    CallScope(const CallScope&) = delete;
    CallScope& operator=(const CallScope&) = delete;
};
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bijective XCD-aware remap of a 1-D block id: consecutive logical ids land on the same XCD (and share its L2).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks / kNumXcd, r = nblocks % kNumXcd;
    const unsigned xcd = bid % kNumXcd, slot = bid / kNumXcd;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// IEEE totalOrder key: unsigned compare of f2key(a), f2key(b) orders floats with -0 < +0.
__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Optional per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline leg).
// Off by default: no events, no overhead.  Classes are stable ABI (optex_prof_class_name).
enum KClass { KC_GEMM = 0, KC_MINMAX, KC_HIST, KC_LUT, KC_APPLY, KC_SORT, KC_SORT_MATCH, KC_MEAN, KC_GRAM, KC_COVFIN,
              KC_ROTGEN, KC_INTERP, KC_SORT_FALLBACK, KC_GLUE, KC_SMALL_GEMM, KC_CHOL, KC_NS_INIT, KC_NORMALS, KC_CDF_FUSED, KC_COUNT };
struct ProfScope {
    int cls;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(int cls, hipStream_t st, double flops, double bytes);
    ~ProfScope();
};

// implemented in cdf.hip / sort.hip, shared with the fused loop in ot_loop.hip
int cdf_match_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                   int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, float* dbg,
                   hipStream_t st);
int cdf_match_parts_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                         int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, float* dbg,
                         const float* tmn_parts, const float* tmx_parts, int parts, hipStream_t st,
                         const float* smn_given = nullptr, const float* smx_given = nullptr, bool ws_clean = false,
                         const unsigned* shist_given = nullptr);
// shist_given [src_n_seg, C, 256]: the source columns' histograms over [smn_given, smx_given] (col_hist_launch), taken by every
// column whose joint range is the source's own
int col_hist_launch(const float* x, long ld, long ss, long n, int C, int n_seg, const float* lo, const float* hi, unsigned* hist,
                    hipStream_t st);
// clears the counters of a cdf scratch (its pipeline leaves them clear: once per loop); per-column min / max of segments
int cdf_ws_clear(void* ws, int C, int n_seg, hipStream_t st);
int col_minmax_launch(const float* x, long ld, long ss, long n, int C, int n_seg, float* mn, float* mx, hipStream_t st);
// tmn_parts / tmx_parts (optional): per-tile min / max partials of the TARGET columns [n_seg][parts][C], as the rotation GEMM's
// row-statistics epilogue leaves them — the rank kernel then starts from the folded range instead of reducing the column
int sort_match_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                    int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, hipStream_t st,
                    const float* tmn_parts = nullptr, const float* tmx_parts = nullptr, int parts = 0,
                    const float* src_sorted_given = nullptr);
// src_sorted_given: the source columns sorted already, [src_n_seg, C, ns] contiguous (`source` is then not read)
int sort_columns_inplace(float* keys, long n, int ncols, int* flags, hipStream_t st);
// fold per-tile min / max partials [n_seg][parts][C] into one range per column, mn / mx [n_seg * C] (cdf.hip)
int minmax_fold_parts(const float* pmn, const float* pmx, int parts, int C, int ncols, float* mn, float* mx, hipStream_t st);
int linear_stats_parts(const float* x, long ld, long seg_stride, long n, int C, int n_seg, int pool, float eps, float* mu,
                       float* cov, void* ws, size_t ws_bytes, const float* sum_parts, int parts, void* stream);
int device_cu_count();
// device-to-device copy / 32-bit fill as plain kernels on `st` (api.hip): the library enqueues nothing but kernel launches,
// so a captured hipGraph of any call holds kernel nodes only (tests/test_gpu_parity.py::test_ot_loop_is_hipgraph_capturable).
int device_copy(float* dst, const float* src, size_t count, hipStream_t st);
int device_fill_u32(uint32_t* dst, uint32_t value, size_t count, hipStream_t st);

}  // namespace optex
