// sort_large.hip — K6 for columns LONGER than one LDS (n > 16384: relu1_1..relu3_1 at 1024^2 / 2048^2): a global,
// segmented, stable LSD radix sort of (totalOrder key, pixel index) pairs, 4 passes of 8 bits.  Each pass is
//   count   : per (column, 8192-key chunk) 256-bin digit histogram                     (LDS atomics)
//   scan    : per column, exclusive scan of the counts in (digit-major, chunk-minor) order
//   scatter : every chunk re-reads its keys in order, ranks them stably inside the block (per-wave counters in
//             (digit, wave) order + ballot match-any inside each 64-key round, exactly the in-LDS radix kernel's pass)
//             and writes the pairs to  base[digit][chunk] + rank.
// The last pass writes the final outputs directly (sorted keys / indices, or for the match the source order statistic of
// every pixel).  Same specification as the short-column kernels (oracle orc_sort_columns / orc_sort_match): stable,
// IEEE totalOrder, any input.  Traffic is ~4x (8 B in + 8 B out) per element instead of the single pass of the
// LDS-resident kernels; this path exists for completeness at the large BASELINE sizes, not for the headline number.
#include "sort_common.h"

namespace optex {

constexpr int LG_ITEMS = 8;
constexpr int LG_CHUNK = LG_ITEMS * SORT_NT;  // 8192 keys per block

struct LargeArgs {
    // input of this pass: raw fp32 column (pass 0) or pairs
    const float* keys; long ld, ss; int C, x_n_seg;
    const uint32_t* in_k; const uint32_t* in_i;
    uint32_t* out_k; uint32_t* out_i;             // pairs out (passes 0..2)
    uint32_t* counts;                             // [ncols][256][nchunks]
    long n; int nchunks; int shift; int first; int last;
    // final outputs (last pass)
    int mode; float* fkeys; uint32_t* fidx;       // SORT_EMIT
    const float* src_sorted; long ns; int src_n_seg; float* out; long ldo, oss; double inv_2nt;  // SORT_MATCH
};

__device__ __forceinline__ void lg_load(const LargeArgs& a, int col, int chunk, int tid, uint32_t (&key)[LG_ITEMS],
                                        uint32_t (&idx)[LG_ITEMS]) {
    const int lane = tid & 63, w = tid >> 6;
    const long base = (long)chunk * LG_CHUNK;
    if (a.first) {
        const int seg = col / a.C, c = col % a.C;
        const int xseg = (a.x_n_seg == 1) ? 0 : seg;
        const float* src = a.keys + (size_t)xseg * a.ss + (size_t)c * a.ld;
#pragma unroll
        for (int r = 0; r < LG_ITEMS; r++) {
            const long e = base + w * (LG_ITEMS * 64) + r * 64 + lane;
            key[r] = f2key(src[e < a.n ? e : a.n - 1]);
            idx[r] = (uint32_t)e;
        }
    } else {
        const uint32_t* pk = a.in_k + (size_t)col * a.n;
        const uint32_t* pi = a.in_i + (size_t)col * a.n;
#pragma unroll
        for (int r = 0; r < LG_ITEMS; r++) {
            const long e = base + w * (LG_ITEMS * 64) + r * 64 + lane;
            const long ec = e < a.n ? e : a.n - 1;
            key[r] = pk[ec];
            idx[r] = pi[ec];
        }
    }
}

__global__ __launch_bounds__(SORT_NT) void lg_count_kernel(LargeArgs a) {
    __shared__ uint32_t h[SORT_RADIX];
    const int chunk = blockIdx.x, col = blockIdx.y, tid = threadIdx.x;
    if (tid < SORT_RADIX) h[tid] = 0u;
    uint32_t key[LG_ITEMS], idx[LG_ITEMS];
    lg_load(a, col, chunk, tid, key, idx);
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int r = 0; r < LG_ITEMS; r++) {
        const long e = (long)chunk * LG_CHUNK + w * (LG_ITEMS * 64) + r * 64 + lane;
        if (e < a.n) atomicAdd(&h[(key[r] >> a.shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < SORT_RADIX) a.counts[((size_t)col * SORT_RADIX + tid) * a.nchunks + chunk] = h[tid];
}

// one block per column: exclusive scan of counts[col][.][.] in place
__global__ __launch_bounds__(SORT_NT) void lg_scan_kernel(uint32_t* counts, int nchunks) {
    __shared__ uint32_t red[32];
    uint32_t* c = counts + (size_t)blockIdx.x * SORT_RADIX * nchunks;
    const int total = SORT_RADIX * nchunks;
    const int per = (total + SORT_NT - 1) / SORT_NT;
    const int beg = threadIdx.x * per, end = beg + per < total ? beg + per : total;
    unsigned sum = 0;
    for (int i = beg; i < end; i++) sum += c[i];
    unsigned ex = block_excl_scan(sum, red, nullptr);
    for (int i = beg; i < end; i++) {
        const unsigned v = c[i];
        c[i] = ex;
        ex += v;
    }
}

__global__ __launch_bounds__(SORT_NT) void lg_scatter_kernel(LargeArgs a) {
    __shared__ uint32_t cnt[SORT_CSTR * SORT_NW];
    const int chunk = blockIdx.x, col = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int i = tid; i < SORT_CSTR * SORT_NW; i += SORT_NT) cnt[i] = 0u;
    uint32_t key[LG_ITEMS], idx[LG_ITEMS];
    lg_load(a, col, chunk, tid, key, idx);
    __syncthreads();
    const long ebase = (long)chunk * LG_CHUNK + w * (LG_ITEMS * 64) + lane;
    uint32_t* mycnt = cnt + w * SORT_CSTR;
#pragma unroll
    for (int r = 0; r < LG_ITEMS; r++)
        if (ebase + r * 64 < a.n) atomicAdd(&mycnt[(key[r] >> a.shift) & 255u], 1u);
    __syncthreads();
    // per digit: exclusive prefix over the 16 waves, plus the global base of (digit, chunk).  Thread t owns digit t / 4
    // and waves 4 * (t % 4) .. + 3; the four threads of a digit are neighbouring lanes.
    {
        const int d = tid >> 2, g = tid & 3;
        uint32_t* c4 = cnt + (4 * g) * SORT_CSTR + d;
        const unsigned v0 = c4[0], v1 = c4[SORT_CSTR], v2 = c4[2 * SORT_CSTR], v3 = c4[3 * SORT_CSTR];
        const unsigned s4 = v0 + v1 + v2 + v3;
        unsigned incl = s4;
        unsigned t = __shfl_up(incl, 1);
        if (g >= 1) incl += t;
        t = __shfl_up(incl, 2);
        if (g >= 2) incl += t;
        unsigned ex = incl - s4 + a.counts[((size_t)col * SORT_RADIX + d) * a.nchunks + chunk];
        c4[0] = ex; ex += v0;
        c4[SORT_CSTR] = ex; ex += v1;
        c4[2 * SORT_CSTR] = ex; ex += v2;
        c4[3 * SORT_CSTR] = ex;
    }
    __syncthreads();
    const int seg = col / a.C, c = col % a.C;
    uint32_t* ok = a.out_k ? a.out_k + (size_t)col * a.n : nullptr;
    uint32_t* oi = a.out_i ? a.out_i + (size_t)col * a.n : nullptr;
    float* fk = a.fkeys ? a.fkeys + (size_t)col * a.n : nullptr;
    uint32_t* fi = a.fidx ? a.fidx + (size_t)col * a.n : nullptr;
    const float* ssrt = nullptr;
    float* mo = nullptr;
    if (a.last && a.mode == SORT_MATCH) {
        const int sseg = (a.src_n_seg == 1) ? 0 : seg;
        ssrt = a.src_sorted + ((size_t)sseg * a.C + c) * a.ns;
        mo = a.out + (size_t)seg * a.oss + (size_t)c * a.ldo;
    }
    volatile uint32_t* vcnt = mycnt;
#pragma unroll
    for (int r = 0; r < LG_ITEMS; r++) {
        const bool valid = ebase + r * 64 < a.n;
        // pads (beyond n) are masked out of every match group
        const unsigned d = (key[r] >> a.shift) & 255u;
        const unsigned long long vm = __ballot(valid);
        const unsigned long long m = match_digit(d) & vm;
        if (valid) {
            const unsigned rank = __popcll(m & lt_mask);
            const unsigned base = vcnt[d];
            const unsigned pos = base + rank;
            if (!a.last) {
                ok[pos] = key[r];
                oi[pos] = idx[r];
            } else if (a.mode == SORT_EMIT) {
                if (fk) fk[pos] = key2f(key[r]);
                if (fi) fi[pos] = idx[r];
            } else {
                mo[idx[r]] = ssrt[quantile_index(pos, (unsigned)a.ns, (unsigned)a.n, a.inv_2nt)];
            }
            if (rank == 0) vcnt[d] = base + (unsigned)__popcll(m);
        }
    }
}

size_t sort_large_ws_bytes(long n, int ncols) {
    const size_t nch = (size_t)((n + LG_CHUNK - 1) / LG_CHUNK);
    return 4 * align_up((size_t)ncols * n * sizeof(uint32_t), 256) + align_up((size_t)ncols * SORT_RADIX * nch * 4, 256);
}

int sort_large(int mode, const SortArgs& s, int ncols, void* ws, hipStream_t st) {
    if (!ws) {
        set_error("sort: columns of %ld keys need the workspace of optex_sort_ws_bytes", s.n);
        return OPTEX_E_ARG;
    }
    if (s.n >= (1L << 31) || ncols > 65535) {
        set_error("sort: n = %ld / %d columns exceed the large-column kernel's limits", s.n, ncols);
        return OPTEX_E_UNSUPPORTED;
    }
    const int nch = (int)((s.n + LG_CHUNK - 1) / LG_CHUNK);
    const size_t arr = align_up((size_t)ncols * s.n * sizeof(uint32_t), 256);
    char* p = static_cast<char*>(ws);
    uint32_t* bufk[2] = {reinterpret_cast<uint32_t*>(p), reinterpret_cast<uint32_t*>(p + arr)};
    uint32_t* bufi[2] = {reinterpret_cast<uint32_t*>(p + 2 * arr), reinterpret_cast<uint32_t*>(p + 3 * arr)};
    uint32_t* counts = reinterpret_cast<uint32_t*>(p + 4 * arr);
    const double per_elem = (mode == SORT_EMIT) ? (4.0 + (s.out_keys ? 4.0 : 0.0) + (s.out_idx ? 4.0 : 0.0)) : 12.0;
    ProfScope prof(mode == SORT_EMIT ? KC_SORT : KC_SORT_MATCH, st, 0.0, per_elem * (double)s.n * ncols);
    for (int pass = 0; pass < 4; pass++) {
        LargeArgs a{};
        a.keys = s.keys; a.ld = s.ld; a.ss = s.ss; a.C = s.C; a.x_n_seg = s.x_n_seg;
        a.in_k = bufk[(pass + 1) & 1]; a.in_i = bufi[(pass + 1) & 1];
        a.out_k = bufk[pass & 1]; a.out_i = bufi[pass & 1];
        a.counts = counts; a.n = s.n; a.nchunks = nch; a.shift = 8 * pass; a.first = pass == 0; a.last = pass == 3;
        a.mode = mode; a.fkeys = s.out_keys; a.fidx = s.out_idx;
        a.src_sorted = s.src_sorted; a.ns = s.ns; a.src_n_seg = s.src_n_seg; a.out = s.out; a.ldo = s.ldo; a.oss = s.oss;
        a.inv_2nt = 1.0 / (2.0 * (double)s.n);
        dim3 grid((unsigned)nch, (unsigned)ncols);
        hipLaunchKernelGGL(lg_count_kernel, grid, dim3(SORT_NT), 0, st, a);
        hipLaunchKernelGGL(lg_scan_kernel, dim3((unsigned)ncols), dim3(SORT_NT), 0, st, counts, nch);
        hipLaunchKernelGGL(lg_scatter_kernel, grid, dim3(SORT_NT), 0, st, a);
        const int rc = check_launch("sort_large pass");
        if (rc) return rc;
    }
    return OPTEX_OK;
}

}  // namespace optex
