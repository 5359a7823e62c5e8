// sort.hip — K6: segmented stable radix sort of rotated feature columns, LDS-resident, one workgroup per column,
// and the exact 1-D optimal-transport match built on it (north-star addition; specification =
// oracle/optex_oracle.c orc_sort_columns / orc_sort_match).
//
// A column of n <= 16384 fp32 keys + uint32 pixel indices is 128 KiB: it fits the 160 KiB LDS of a CDNA4 CU, so the
// column is read from HBM once and written once (algorithmic traffic 12*n*C bytes for key+index output).
// LSD radix, 4 passes of 8 bits.  Each of the 16 wavefronts owns a contiguous slice of the column; stability
// comes from (digit-major, wave-minor) exclusive offsets plus a match-any ranking inside each 64-key round.
#include "optex_common.h"

namespace optex {

constexpr int SORT_NT = 1024;            // 16 wavefronts
constexpr int SORT_NW = SORT_NT / 64;
constexpr int SORT_RADIX = 256;
constexpr int SORT_MAX_N = 16384;

enum SortMode { SORT_EMIT = 0, SORT_MATCH = 1 };

struct SortArgs {
    const float* keys; long ld, ss; long n; int C; int x_n_seg;
    float* out_keys; uint32_t* out_idx;                       // SORT_EMIT, contiguous [n_seg, C, n]
    const float* src_sorted; long ns; int src_n_seg;          // SORT_MATCH: sorted source keys [src_n_seg, C, ns]
    float* out; long ldo, oss;                                // SORT_MATCH
};

// lanes holding the same 8-bit digit: 8 ballots
__device__ __forceinline__ unsigned long long match_digit(unsigned d) {
    unsigned long long m = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

template <int ITEMS, int MODE>
__global__ __launch_bounds__(SORT_NT) void sort_columns_kernel(SortArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CAP = ITEMS * SORT_NT;  // padded column length held in LDS
    uint32_t* skey = reinterpret_cast<uint32_t*>(smem);
    uint32_t* sidx = skey + CAP;
    uint32_t* cnt = sidx + CAP;             // [256 digits][16 waves], digit-major
    uint32_t* wtot = cnt + SORT_RADIX * SORT_NW;  // [16] scan scratch

    const int col = blockIdx.x, seg = col / a.C, c = col % a.C;
    const int xseg = (a.x_n_seg == 1) ? 0 : seg;
    const float* src = a.keys + (size_t)xseg * a.ss + (size_t)c * a.ld;
    const int n = (int)a.n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // element e of wave w, round r:  e = w * (ITEMS*64) + r*64 + lane  (monotone in (w, r, lane): pads are the tail)
    uint32_t key[ITEMS], idx[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const int e = w * (ITEMS * 64) + r * 64 + lane;
        key[r] = (e < n) ? f2key(src[e]) : 0xffffffffu;
        idx[r] = (uint32_t)e;
    }

#pragma unroll 1
    for (int pass = 0; pass < 4; pass++) {
        const int sh = pass * 8;
        for (int i = tid; i < SORT_RADIX * SORT_NW; i += SORT_NT) cnt[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; r++) atomicAdd(&cnt[((key[r] >> sh) & 255u) * SORT_NW + w], 1u);
        __syncthreads();
        // exclusive scan of the 4096 counters in (digit, wave) order: 4 per thread
        {
            uint4 v = reinterpret_cast<uint4*>(cnt)[tid];
            const unsigned s4 = v.x + v.y + v.z + v.w;
            unsigned incl = s4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wtot[w] = incl;
            __syncthreads();
            unsigned base = 0;
            for (int k = 0; k < w; k++) base += wtot[k];
            unsigned ex = base + incl - s4;
            uint4 o4;
            o4.x = ex; ex += v.x;
            o4.y = ex; ex += v.y;
            o4.z = ex; ex += v.z;
            o4.w = ex;
            reinterpret_cast<uint4*>(cnt)[tid] = o4;
        }
        __syncthreads();
        // ranked scatter, rounds in order (the counter row of this wave is private to it; LDS ops of one wave are ordered)
        volatile uint32_t* vcnt = cnt;
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const unsigned d = (key[r] >> sh) & 255u;
            const unsigned long long m = match_digit(d);
            const unsigned rank = __popcll(m & lt_mask);
            const unsigned base = vcnt[d * SORT_NW + w];
            const unsigned pos = base + rank;
            skey[pos] = key[r];
            sidx[pos] = idx[r];
            if (rank == 0) vcnt[d * SORT_NW + w] = base + (unsigned)__popcll(m);
        }
        __syncthreads();
        if (pass < 3) {
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const int e = w * (ITEMS * 64) + r * 64 + lane;
                key[r] = skey[e];
                idx[r] = sidx[e];
            }
            // the next pass zeroes cnt and syncs before anyone scatters again, so these reads are safe
        }
    }

    if (MODE == SORT_EMIT) {
        float* ok = a.out_keys ? a.out_keys + (size_t)col * n : nullptr;
        uint32_t* oi = a.out_idx ? a.out_idx + (size_t)col * n : nullptr;
        for (int i = tid; i < n; i += SORT_NT) {
            if (ok) ok[i] = key2f(skey[i]);
            if (oi) oi[i] = sidx[i];
        }
    } else {
        // out[pixel holding the i-th smallest target] = source order statistic floor((2i+1)*ns / (2*nt))
        const int sseg = (a.src_n_seg == 1) ? 0 : seg;
        const float* ssrt = a.src_sorted + ((size_t)sseg * a.C + c) * a.ns;
        const unsigned long long ns = (unsigned long long)a.ns, nt2 = 2ull * (unsigned long long)n;
        float* sval = reinterpret_cast<float*>(skey);  // keys are dead: reuse as the scatter target
        uint32_t myidx[ITEMS];
        float myval[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const int i = tid + r * SORT_NT;
            if (i < n) {
                myidx[r] = sidx[i];
                myval[r] = ssrt[((2ull * (unsigned long long)i + 1ull) * ns) / nt2];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const int i = tid + r * SORT_NT;
            if (i < n) sval[myidx[r]] = myval[r];
        }
        __syncthreads();
        float* o = a.out + (size_t)seg * a.oss + (size_t)c * a.ldo;
        for (int i = tid; i < n; i += SORT_NT) o[i] = sval[i];
    }
}

template <int ITEMS, int MODE>
static int launch_sort_items(const SortArgs& a, int ncols, hipStream_t st) {
    const size_t lds = (size_t)ITEMS * SORT_NT * 8 + (size_t)SORT_RADIX * SORT_NW * 4 + SORT_NW * 4;
    auto kern = sort_columns_kernel<ITEMS, MODE>;
    static thread_local bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("sort_columns_kernel: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return OPTEX_E_LAUNCH;
        }
        attr_done = true;
    }
    // algorithmic bytes: read key 4 + write key 4 + write index 4 per element; the match reads the
    // column (4), reads one source order statistic per pixel (4) and writes the matched column (4)
    const double per_elem = (MODE == SORT_EMIT) ? (4.0 + (a.out_keys ? 4.0 : 0.0) + (a.out_idx ? 4.0 : 0.0)) : 12.0;
    ProfScope prof(MODE == SORT_EMIT ? KC_SORT : KC_SORT_MATCH, st, 0.0, per_elem * (double)a.n * ncols);
    hipLaunchKernelGGL(kern, dim3(ncols), dim3(SORT_NT), lds, st, a);
    return check_launch("sort_columns_kernel");
}

template <int MODE>
static int launch_sort(const SortArgs& a, int ncols, hipStream_t st) {
    if (a.n <= 2 * SORT_NT) return launch_sort_items<2, MODE>(a, ncols, st);
    if (a.n <= 4 * SORT_NT) return launch_sort_items<4, MODE>(a, ncols, st);
    if (a.n <= 8 * SORT_NT) return launch_sort_items<8, MODE>(a, ncols, st);
    if (a.n <= 16 * SORT_NT) return launch_sort_items<16, MODE>(a, ncols, st);
    set_error("sort: columns longer than %d keys are not supported yet (n = %ld)", SORT_MAX_N, a.n);
    return OPTEX_E_UNSUPPORTED;
}

int sort_match_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                    int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, hipStream_t st) {
    // 1. sort the source columns (keys only) into ws: [src_n_seg, C, ns]
    float* ssorted = static_cast<float*>(ws);
    SortArgs s{};
    s.keys = source; s.ld = lds; s.ss = sss; s.n = ns; s.C = C; s.x_n_seg = src_n_seg;
    s.out_keys = ssorted; s.out_idx = nullptr;
    int rc = launch_sort<SORT_EMIT>(s, C * src_n_seg, st);
    if (rc) return rc;
    // 2. sort each target column with its pixel indices and scatter the source quantiles
    SortArgs t{};
    t.keys = target; t.ld = ldt; t.ss = tss; t.n = nt; t.C = C; t.x_n_seg = n_seg;
    t.src_sorted = ssorted; t.ns = ns; t.src_n_seg = src_n_seg;
    t.out = out; t.ldo = ldo; t.oss = oss;
    return launch_sort<SORT_MATCH>(t, C * n_seg, st);
}

}  // namespace optex

using namespace optex;

extern "C" size_t optex_sort_ws_bytes(long n, int C, int n_seg) {
    (void)n; (void)C; (void)n_seg;
    return 256;  // the LDS-resident path needs no global scratch
}

extern "C" int optex_sort_columns(const float* keys, long ld, long seg_stride, long n, int C, int n_seg,
                                  float* out_keys, uint32_t* out_idx, void* ws, void* stream) {
    (void)ws;
    if (!keys || n <= 0 || C <= 0 || n_seg <= 0 || ld < n) {
        set_error("optex_sort_columns: bad argument (n=%ld C=%d n_seg=%d ld=%ld)", n, C, n_seg, ld);
        return OPTEX_E_ARG;
    }
    SortArgs a{};
    a.keys = keys; a.ld = ld; a.ss = seg_stride; a.n = n; a.C = C; a.x_n_seg = n_seg;
    a.out_keys = out_keys; a.out_idx = out_idx;
    return launch_sort<SORT_EMIT>(a, C * n_seg, as_stream(stream));
}

extern "C" size_t optex_sort_match_ws_bytes(long nt, long ns, int C, int n_seg, int src_n_seg) {
    (void)nt; (void)n_seg;
    return align_up((size_t)src_n_seg * C * ns * sizeof(float), 256);
}

extern "C" int optex_sort_match(const float* target, long ldt, long t_seg_stride, long nt, const float* source,
                                long lds, long s_seg_stride, long ns, int src_n_seg, int C, int n_seg, float* out,
                                long ldo, long o_seg_stride, void* ws, void* stream) {
    if (!target || !source || !out || !ws || nt <= 0 || ns <= 0 || C <= 0 || n_seg <= 0 || ldt < nt || lds < ns ||
        ldo < nt) {
        set_error("optex_sort_match: bad argument (nt=%ld ns=%ld C=%d n_seg=%d)", nt, ns, C, n_seg);
        return OPTEX_E_ARG;
    }
    if (src_n_seg != 1 && src_n_seg != n_seg) {
        set_error("optex_sort_match: source has %d segments, expected 1 or %d", src_n_seg, n_seg);
        return OPTEX_E_ARG;
    }
    return sort_match_impl(target, ldt, t_seg_stride, nt, source, lds, s_seg_stride, ns, src_n_seg, C, n_seg, out, ldo,
                           o_seg_stride, ws, as_stream(stream));
}
