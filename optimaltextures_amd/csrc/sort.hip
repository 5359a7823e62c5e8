// sort.hip — K6: segmented stable sort of rotated feature columns, LDS-resident, one workgroup per column, and the
// exact 1-D optimal-transport match built on it (north-star addition; specification =
// oracle/optex_oracle.c orc_sort_columns / orc_sort_match).
//
// A column of n <= 16384 fp32 keys lives in the 160 KiB LDS of one CDNA4 CU: it is read from HBM once and the result
// written once (algorithmic traffic 12 B per element: key in, key + index out, or for the match: key in, one source
// order statistic in, matched value out).
//
// The fast paths are rank_match4_kernel<..., SORT_MATCH / SORT_EMIT> (sort_rank4.hip): RANKING BY COUNTING instead of
// moving data through radix passes.  They flag every column they cannot take (non-finite keys, massive distinct ties, ...)
// and the kernel below sweeps those up right behind them on the same stream (no host round trip):
//
//  sort_columns_kernel — the general LSD radix sort (4 passes of 8 bits, 16 wavefronts, stability from (digit-major,
//    wave-minor) offsets plus ballot-based match-any ranking).  It is the specification-conformant path for every input,
//    and the only one for columns shorter than RK_MIN_N keys.
#include "sort_common.h"

namespace optex {

// ================================================================================================ radix kernel
template <int ITEMS, int MODE>
__global__ __launch_bounds__(SORT_NT) void sort_columns_kernel(SortArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CAP = ITEMS * SORT_NT;  // padded column length held in LDS
    uint32_t* skey = reinterpret_cast<uint32_t*>(smem);
    uint32_t* sidx = skey + CAP;
    // counters: element (digit d, wave w) at cnt[w * SORT_CSTR + d].  Wave-major with an odd stride: the 64 lanes of a
    // wave (same w, random d) spread over all banks.
    uint32_t* cnt = sidx + CAP;
    uint32_t* wtot = cnt + SORT_CSTR * SORT_NW;  // [16] scan scratch

    // As the sweep behind a rank kernel (only_flagged) the launch has one workgroup per CU, each walking the columns with
    // a stride and taking the flagged ones: 16384 workgroups of 1024 threads and 128 KiB of LDS that start only to find
    // their flag clear cost 30 us per launch (5 % of the match they follow).
    // (64 flags per round, one per lane, then the set bits: a serial walk is 64 dependent loads per workgroup, ~25 us)
    for (int base = blockIdx.x; base < a.ncols; base += (int)gridDim.x * 64) {
    const long mycol = (long)base + (long)(threadIdx.x & 63) * (long)gridDim.x;
    unsigned long long todo = __ballot(mycol < a.ncols && (!a.only_flagged || a.flags[mycol] != 0));
    while (todo) {
    const int col = base + __builtin_ctzll(todo) * (int)gridDim.x;
    todo &= todo - 1ull;
    const int seg = col / a.C, c = col % a.C;
    const int xseg = (a.x_n_seg == 1) ? 0 : seg;
    const float* src = a.keys + (size_t)xseg * a.ss + (size_t)c * a.ld;
    const int n = (int)a.n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // element e of wave w, round r:  e = w * (ITEMS*64) + r*64 + lane  (monotone in (w, r, lane): pads are the tail)
    uint32_t key[ITEMS], idx[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {  // unconditional clamped loads: all in flight at once
        const int e = w * (ITEMS * 64) + r * 64 + lane;
        const float v = src[e < n ? e : n - 1];
        key[r] = f2key(v);
        idx[r] = (uint32_t)e;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++)
        if ((int)idx[r] >= n) key[r] = 0xffffffffu;

#pragma unroll 1
    for (int pass = 0; pass < 4; pass++) {
        const int sh = pass * 8;
        for (int i = tid; i < SORT_CSTR * SORT_NW; i += SORT_NT) cnt[i] = 0u;
        __syncthreads();
        uint32_t* mycnt = cnt + w * SORT_CSTR;
#pragma unroll
        for (int r = 0; r < ITEMS; r++) atomicAdd(&mycnt[(key[r] >> sh) & 255u], 1u);
        __syncthreads();
        // exclusive scan of the 4096 counters in (digit-major, wave-minor) order: thread t owns digit t / 4, waves
        // 4 * (t % 4) .. + 3
        {
            uint32_t* c4 = cnt + (4 * (tid & 3)) * SORT_CSTR + (tid >> 2);
            uint4 v;
            v.x = c4[0];
            v.y = c4[SORT_CSTR];
            v.z = c4[2 * SORT_CSTR];
            v.w = c4[3 * SORT_CSTR];
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
            c4[0] = o4.x;
            c4[SORT_CSTR] = o4.y;
            c4[2 * SORT_CSTR] = o4.z;
            c4[3 * SORT_CSTR] = o4.w;
        }
        __syncthreads();
        // ranked scatter, rounds in order (the counter row of this wave is private to it; LDS ops of one wave are ordered)
        volatile uint32_t* vcnt = mycnt;
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const unsigned d = (key[r] >> sh) & 255u;
            const unsigned long long m = match_digit(d);
            const unsigned rank = __popcll(m & lt_mask);
            const unsigned base = vcnt[d];
            const unsigned pos = base + rank;
            skey[pos] = key[r];
            sidx[pos] = idx[r];
            if (rank == 0) vcnt[d] = base + (unsigned)__popcll(m);
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
        float* sval = reinterpret_cast<float*>(skey);  // keys are dead: reuse as the scatter target
        uint32_t myidx[ITEMS];
        float myval[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const int i = tid + r * SORT_NT;
            if (i < n) {
                myidx[r] = sidx[i];
                myval[r] = ssrt[quantile_index((unsigned)i, (unsigned)a.ns, (unsigned)n, a.inv_2nt)];
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
    __syncthreads();  // the LDS arrays are reused by the next column of this workgroup
    }
    }
}


int device_cu_count();

// ================================================================================================ host side
template <typename KernT>
static int set_lds(KernT kern, size_t lds, bool* done) {
    if (*done) return OPTEX_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
        set_error("sort: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        return OPTEX_E_LAUNCH;
    }
    *done = true;
    return OPTEX_OK;
}

template <int ITEMS, int MODE>
static int launch_sort_items(SortArgs a, int ncols, int* flags, hipStream_t st) {
    // algorithmic bytes: read key 4 + write key 4 + write index 4 per element; the match reads the
    // column (4), reads one source order statistic per pixel (4) and writes the matched column (4)
    const double per_elem = (MODE == SORT_EMIT) ? (4.0 + (a.out_keys ? 4.0 : 0.0) + (a.out_idx ? 4.0 : 0.0)) : 12.0;
    const bool use_rank = flags != nullptr && a.n >= RK_MIN_N;
    a.flags = flags;
    a.ncols = ncols;
    a.inv_2nt = 1.0 / (2.0 * (double)a.n);
    int rc;
    if (use_rank) {
        if ((rc = device_fill_u32(reinterpret_cast<uint32_t*>(flags), 0u, (size_t)ncols, st))) return rc;
        {
            ProfScope prof(MODE == SORT_EMIT ? KC_SORT : KC_SORT_MATCH, st, 0.0, per_elem * (double)a.n * ncols);
            // owner-ranked, float domain: over-provisioned 8-bit buckets where the shape allows it (sort_rank5.hip, round 6),
            // the 8-slot-window kernel otherwise (sort_rank4.hip) — both flag what they cannot take
            if (tl_call.sort_rank4 <= 0 && rank5w_supported(MODE, a)) {
                if ((rc = launch_rank5w(MODE, a, ncols, st))) return rc;
                if (!a.rng_lo) {
                    // No caller-given range = not the rotated pastiche of the hot loop: the columns may hold massive ties (the
                    // zeros of un-rotated ReLU features), which the 8-bit counters flag and rank_match4_kernel ranks itself
                    // (all-equal buckets by pixel index).  It takes the flagged columns only (30 us of empty workgroups per
                    // launch otherwise) and un-flags the ones it ranks; what it cannot take either goes to the radix sweep.
                    a.only_flagged = 1;
                    if ((rc = launch_rank4(MODE, a, ncols, st))) return rc;
                }
            } else if ((rc = launch_rank4(MODE, a, ncols, st))) return rc;
        }
    }
    const size_t lds = (size_t)ITEMS * SORT_NT * 8 + (size_t)SORT_CSTR * SORT_NW * 4 + SORT_NW * 4;
    static DeviceOnce attr_radix;
    auto kern = sort_columns_kernel<ITEMS, MODE>;
    if ((rc = set_lds(kern, lds, attr_radix.slot()))) return rc;
    a.only_flagged = use_rank ? 1 : 0;
    // when it only sweeps up flagged columns the radix launch is accounted with zero algorithmic bytes
    ProfScope prof(use_rank ? KC_SORT_FALLBACK : (MODE == SORT_EMIT ? KC_SORT : KC_SORT_MATCH), st, 0.0,
                   use_rank ? 0.0 : per_elem * (double)a.n * ncols);
    int grid = ncols;
    if (use_rank && device_cu_count() < grid) grid = device_cu_count();  // sweep: one workgroup per CU walks the flags
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SORT_NT), lds, st, a);
    return check_launch("sort_columns_kernel");
}

static size_t flags_bytes(int ncols) { return align_up(sizeof(int) * (size_t)ncols, 256); }
static size_t range_bytes(int ncols) { return 2 * align_up(sizeof(float) * (size_t)ncols, 256); }  // lo, hi per target column

template <int MODE>
static int launch_sort(const SortArgs& a, int ncols, int* flags, void* large_ws, hipStream_t st) {
    if (a.n <= 2 * SORT_NT) return launch_sort_items<2, MODE>(a, ncols, flags, st);
    if (a.n <= 4 * SORT_NT) return launch_sort_items<4, MODE>(a, ncols, flags, st);
    if (a.n <= 8 * SORT_NT) return launch_sort_items<8, MODE>(a, ncols, flags, st);
    if (a.n <= 12 * SORT_NT) return launch_sort_items<12, MODE>(a, ncols, flags, st);
    if (a.n <= 16 * SORT_NT) return launch_sort_items<16, MODE>(a, ncols, flags, st);
    // longer than one LDS: global multi-pass radix (sort_large.hip)
    return sort_large(MODE, a, ncols, large_ws, st);
}

// keys [ncols][n] contiguous, sorted IN PLACE (keys only; n <= SORT_MAX_N): the rank kernel holds a whole column in
// registers before it writes, and so does the radix sweep behind it.  flags: ncols ints of scratch.
int sort_columns_inplace(float* keys, long n, int ncols, int* flags, hipStream_t st) {
    if (n > SORT_MAX_N) {
        set_error("sort_columns_inplace: columns of %ld keys (at most %d)", n, SORT_MAX_N);
        return OPTEX_E_UNSUPPORTED;
    }
    SortArgs s{};
    s.keys = keys; s.ld = n; s.ss = n; s.n = n; s.C = 1; s.x_n_seg = ncols;
    s.out_keys = keys; s.out_idx = nullptr;
    return launch_sort<SORT_EMIT>(s, ncols, flags, nullptr, st);
}

int sort_match_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                    int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, hipStream_t st,
                    const float* tmn_parts, const float* tmx_parts, int parts, const float* src_sorted_given) {
    // ws: [flags for the larger launch][lo, hi per target column][sorted source keys [src_n_seg, C, ns]][scratch of the
    //      large-column path]
    const int maxcols = C * (n_seg > src_n_seg ? n_seg : src_n_seg);
    int* flags = static_cast<int*>(ws);
    float* rlo = reinterpret_cast<float*>(static_cast<char*>(ws) + flags_bytes(maxcols));
    float* rhi = reinterpret_cast<float*>(reinterpret_cast<char*>(rlo) + range_bytes(C * n_seg) / 2);
    float* ssorted = reinterpret_cast<float*>(reinterpret_cast<char*>(rlo) + range_bytes(C * n_seg));
    void* large_ws = reinterpret_cast<char*>(ssorted) + align_up((size_t)src_n_seg * C * ns * sizeof(float), 256);
    // 1. sort the source columns (keys only) — unless the caller did (optex_ot_loop sorts the rotated style of every
    //    iteration of a call in one launch)
    int rc;
    if (src_sorted_given) {
        ssorted = const_cast<float*>(src_sorted_given);  // [src_n_seg, C, ns] contiguous, read only
    } else {
        SortArgs s{};
        s.keys = source; s.ld = lds; s.ss = sss; s.n = ns; s.C = C; s.x_n_seg = src_n_seg;
        s.out_keys = ssorted; s.out_idx = nullptr;
        if ((rc = launch_sort<SORT_EMIT>(s, C * src_n_seg, flags, large_ws, st))) return rc;
    }
    // 2. rank each target column and fetch the source quantiles
    SortArgs t{};
    t.keys = target; t.ld = ldt; t.ss = tss; t.n = nt; t.C = C; t.x_n_seg = n_seg;
    t.src_sorted = ssorted; t.ns = ns; t.src_n_seg = src_n_seg;
    t.out = out; t.ldo = ldo; t.oss = oss;
    if (tmn_parts && tmx_parts && parts > 0 && nt >= RK_MIN_N && nt <= SORT_MAX_N) {
        // the range of every target column from the partials the rotation GEMM left (optex_ot_loop): the rank kernel skips
        // its own min / max reduction and the barrier behind it
        if ((rc = minmax_fold_parts(tmn_parts, tmx_parts, parts, C, C * n_seg, rlo, rhi, st))) return rc;
        t.rng_lo = rlo;
        t.rng_hi = rhi;
    }
    return launch_sort<SORT_MATCH>(t, C * n_seg, flags, large_ws, st);
}

}  // namespace optex

using namespace optex;

extern "C" size_t optex_sort_ws_bytes(long n, int C, int n_seg) {
    return flags_bytes(C * n_seg) + (n > SORT_MAX_N ? sort_large_ws_bytes(n, C * n_seg) : 0);
}

extern "C" int optex_sort_columns(const float* keys, long ld, long seg_stride, long n, int C, int n_seg,
                                  float* out_keys, uint32_t* out_idx, void* ws, size_t ws_bytes, void* stream) {
    if (!keys || n <= 0 || C <= 0 || n_seg <= 0 || ld < n) {
        set_error("optex_sort_columns: bad argument (n=%ld C=%d n_seg=%d ld=%ld)", n, C, n_seg, ld);
        return OPTEX_E_ARG;
    }
    if (int rc = check_ws("optex_sort_columns", ws, ws_bytes, optex_sort_ws_bytes(n, C, n_seg))) return rc;
    SortArgs a{};
    a.keys = keys; a.ld = ld; a.ss = seg_stride; a.n = n; a.C = C; a.x_n_seg = n_seg;
    a.out_keys = out_keys; a.out_idx = out_idx;
    void* large_ws = ws ? static_cast<char*>(ws) + flags_bytes(C * n_seg) : nullptr;
    return launch_sort<SORT_EMIT>(a, C * n_seg, static_cast<int*>(ws), large_ws, as_stream(stream));
}

extern "C" size_t optex_sort_match_ws_bytes(long nt, long ns, int C, int n_seg, int src_n_seg) {
    // [flags][sorted source keys][scratch of the large-column path, shared by the source and the target sort]
    size_t large = 0;
    if (nt > SORT_MAX_N) large = sort_large_ws_bytes(nt, C * n_seg);
    if (ns > SORT_MAX_N && sort_large_ws_bytes(ns, C * src_n_seg) > large) large = sort_large_ws_bytes(ns, C * src_n_seg);
    return flags_bytes(C * (n_seg > src_n_seg ? n_seg : src_n_seg)) + range_bytes(C * n_seg) +
           align_up((size_t)src_n_seg * C * ns * sizeof(float), 256) + large;
}

extern "C" int optex_sort_match(const float* target, long ldt, long t_seg_stride, long nt, const float* source,
                                long lds, long s_seg_stride, long ns, int src_n_seg, int C, int n_seg, float* out,
                                long ldo, long o_seg_stride, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    CallScope call_scope(flags);
    if (!target || !source || !out || !ws || nt <= 0 || ns <= 0 || C <= 0 || n_seg <= 0 || ldt < nt || lds < ns ||
        ldo < nt) {
        set_error("optex_sort_match: bad argument (nt=%ld ns=%ld C=%d n_seg=%d)", nt, ns, C, n_seg);
        return OPTEX_E_ARG;
    }
    if (src_n_seg != 1 && src_n_seg != n_seg) {
        set_error("optex_sort_match: source has %d segments, expected 1 or %d", src_n_seg, n_seg);
        return OPTEX_E_ARG;
    }
    if (int rc = check_ws("optex_sort_match", ws, ws_bytes, optex_sort_match_ws_bytes(nt, ns, C, n_seg, src_n_seg))) return rc;
    return sort_match_impl(target, ldt, t_seg_stride, nt, source, lds, s_seg_stride, ns, src_n_seg, C, n_seg, out, ldo,
                           o_seg_stride, ws, as_stream(stream));
}
