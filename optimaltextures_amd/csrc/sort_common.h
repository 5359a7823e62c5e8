// sort_common.h — shared pieces of the column-sort kernels (sort.hip: LDS-resident columns; sort_large.hip: global
// multi-pass radix for columns longer than one LDS)
#pragma once
#include "optex_common.h"

namespace optex {

constexpr int SORT_NT = 1024;            // 16 wavefronts
constexpr int SORT_NW = SORT_NT / 64;
constexpr int SORT_RADIX = 256;
constexpr int SORT_MAX_N = 16384;
constexpr int SORT_CSTR = SORT_RADIX + 1;

constexpr int RK_COARSE = 256;           // equalisation bins
constexpr int RK_BIG = 48;               // buckets above this size take the all-equal path or the radix fallback
constexpr int RK_MAXBIG = 8;
constexpr int RK_MIN_N = 512;            // shorter columns go straight to the radix kernel

enum SortMode { SORT_EMIT = 0, SORT_MATCH = 1 };

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: remember it per device, so a
// process that drives several GPUs sets it on each (a benign race: the call is idempotent).
struct DeviceOnce {
    bool done[64] = {};
    bool* slot() {
        int d = 0;
        (void)hipGetDevice(&d);
        return &done[d & 63];
    }
};

struct SortArgs {
    const float* keys; long ld, ss; long n; int C; int x_n_seg;
    float* out_keys; uint32_t* out_idx;                       // SORT_EMIT, contiguous [n_seg, C, n]
    const float* src_sorted; long ns; int src_n_seg;          // SORT_MATCH: sorted source keys [src_n_seg, C, ns]
    float* out; long ldo, oss;                                // SORT_MATCH
    int* flags;                                               // per column: 1 = needs the radix kernel
    int only_flagged;                                         // radix kernel: skip columns whose flag is 0
    double inv_2nt;                                           // 1 / (2 * n) for the quantile index
    int ncols;                                                // C * n_seg
    int out_vec;                                              // rank_match4_kernel: `out` takes 16-byte stores
    // optional, rank_match4_kernel: the column's min / max [ncols] when the caller has them already (optex_ot_loop: the
    // rotation GEMM's row-statistics epilogue) — the kernel then skips its own reduction and the barrier behind it
    const float* rng_lo; const float* rng_hi;
    // rank_match5w_kernel: the quantile index floor((2 rank + 1) ns / (2 n)) as a multiply-high (sort_rank5.hip, quantile_magic)
    unsigned qmul; int qshr;
#ifdef OPTEX_SORT_PROBE
    long long* probe;                                         // [ncols, 16] phase timestamps (scripts/sort_phase_probe.hip)
#endif
};

#ifdef OPTEX_SORT_PROBE
#define SORT_PROBE(i) do { if (threadIdx.x == 0) a.probe[(size_t)blockIdx.x * 16 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define SORT_PROBE(i) do { } while (0)
#endif

// floor((2 * rank + 1) * ns / (2 * nt)), exact: the quotient is < 2^24, a non-integer quotient is at least 2^-15 away
// from an integer, the double product carries < 2^-28 of error and the 2^-27 bias lifts exact integers over the edge.
__device__ __forceinline__ unsigned quantile_index(unsigned rank, unsigned ns, unsigned nt, double inv_2nt) {
    if (ns == nt) return rank;
    const double a = (double)(2u * rank + 1u) * (double)ns;
    return (unsigned)__builtin_fma(a, inv_2nt, 7.450580596923828e-09);
}

// block-wide exclusive scan of one value per thread (all SORT_NT threads must call); red: >= 17 words of LDS scratch
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, uint32_t* red, unsigned* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) red[w] = incl;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SORT_NW; k++) {
        const unsigned x = red[k];
        if (k < w) base += x;
        tot += x;
    }
    __syncthreads();
    if (total) *total = tot;
    return base + incl - v;
}

// lanes holding the same NBITS-bit digit: NBITS ballots
template <int NBITS = 8>
__device__ __forceinline__ unsigned long long match_digit(unsigned d) {
    unsigned long long m = ~0ull;
#pragma unroll
    for (int b = 0; b < NBITS; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// columns longer than SORT_MAX_N keys (sort_large.hip)
size_t sort_large_ws_bytes(long n, int ncols);
int sort_large(int mode, const SortArgs& a, int ncols, void* ws, hipStream_t st);

// owner-ranked ranking-by-counting kernel in the float domain (sort_rank4.hip): mode = SORT_MATCH (the transport match) or
// SORT_EMIT (sorted keys / pixel indices, optex_sort_columns)
int launch_rank4(int mode, const SortArgs& a, int ncols, hipStream_t st);
// ranking with over-provisioned 8-bit buckets (sort_rank5.hip): 16-byte aligned rows, 2048 < n <= 16384; the match additionally a
// staged source column
bool rank5w_supported(int mode, const SortArgs& a);
int launch_rank5w(int mode, const SortArgs& a, int ncols, hipStream_t st);

}  // namespace optex
