// sort.hip — K6: segmented stable sort of rotated feature columns, LDS-resident, one workgroup per column, and the
// exact 1-D optimal-transport match built on it (north-star addition; specification =
// oracle/optex_oracle.c orc_sort_columns / orc_sort_match).
//
// A column of n <= 16384 fp32 keys lives in the 160 KiB LDS of one CDNA4 CU: it is read from HBM once and the result
// written once (algorithmic traffic 12 B per element: key in, key + index out, or for the match: key in, one source
// order statistic in, matched value out).
//
// Two kernels:
//
//  rank_columns_kernel — the fast path: RANKING BY COUNTING instead of moving data through radix passes.
//    The stable sorted position of element i is  rank_i = #{j : key_j < key_i} + #{j < i : key_j == key_i}.
//    Keys are continuous feature projections, so a monotone bucket function splits the column into ~n/2 buckets of a
//    few keys each; rank_i = (keys in lower buckets) + (count inside its own bucket, by direct comparison).
//      1. min / max of the column (totalOrder keys)
//      2. 256-bin histogram over [lo, hi] (LDS u32 atomics)
//      3. histogram equalisation: coarse bin b gets w_b = 1 + cnt_b * NB / n fine buckets.  The map
//         x -> base_b + min(w_b - 1, int(frac * w_b)) is monotone non-decreasing in x by construction, whatever the
//         rounding, so bucket order == key order; equalisation keeps bucket sizes ~Poisson(2) for ANY continuous
//         distribution (outliers and heavy tails included)
//      4. fine-bucket histogram with slot assignment (LDS atomic-add-return), exclusive scan -> bucket starts
//      5. keys (and 16-bit pixel indices, for ties) are written ONCE to their bucket's slots
//      6. every element counts the smaller keys (ties: smaller indices) in its own bucket -> exact rank, independent of
//         the order in which the atomics of step 4 happened to resolve: deterministic, bit-exact indices
//      7. match: out[i] = sorted_source[q(rank_i)] straight from registers, coalesced (no second scatter);
//         emit: keys / indices are staged by rank in LDS and stored linearly.
//    Buckets of more than RK_BIG keys only arise from exact ties (e.g. the zeros of un-rotated ReLU features): when all
//    keys of such a bucket are equal the rank inside it is the rank of the pixel index, computed with a bitmap and
//    popcounts; anything else (several distinct values with massive ties, non-finite keys) flags the column for the
//    radix kernel below, which runs right behind it on the same stream (no host round trip).
//
//  sort_columns_kernel — the general LSD radix sort (4 passes of 8 bits, 16 wavefronts, stability from (digit-major,
//    wave-minor) offsets plus ballot-based match-any ranking).  It runs over the flagged columns only and is the
//    specification-conformant fallback for every input.
#include <cstdlib>

#include "sort_common.h"

namespace optex {

// ================================================================================================ rank kernel
// LDS map (ITEMS = 16: ~150 KiB): grouped keys gk [CAP] u32, their pixel indices gi [CAP] u16, rank by pixel rk [CAP] u16
// (match mode), packed u16 bucket counters / starts / cursors sc [NBT / 2] u32, bucket-start bitmap bs [CAP/32 + 1],
// index bitmap + its prefix for oversized all-equal buckets, coarse table c1 [256], scratch.
template <int ITEMS, int MODE>
__global__ __launch_bounds__(SORT_NT) void rank_columns_kernel(SortArgs a) {
    constexpr int CAP = ITEMS * SORT_NT;
    constexpr int NB = CAP / 2;                     // fine buckets handed out by the equalisation (+ 1 per coarse bin)
    constexpr int NBT = NB + RK_COARSE;             // even
    constexpr int NW2 = NBT / 2;                    // packed counter words
    constexpr int PER = (NW2 + SORT_NT - 1) / SORT_NT;
    constexpr int NWORDS = CAP / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* gk = reinterpret_cast<uint32_t*>(smem);            // [CAP]
    uint32_t* sc = gk + CAP;                                     // [NW2]
    uint32_t* c1 = sc + NW2;                                     // [256] coarse histogram, then base | width << 16
    uint32_t* bs = c1 + RK_COARSE;                               // [NWORDS + 4] bit p = position p starts a bucket
    uint32_t* bitmap = bs + NWORDS + 4;                          // [NWORDS] pixel indices of one oversized bucket
    uint32_t* bpre = bitmap + NWORDS;                            // [NWORDS]
    uint32_t* red = bpre + NWORDS;                               // [32] reduction / scan scratch
    uint32_t* misc = red + 32;                                   // [32] nbig, noteq, (start, count) x RK_MAXBIG
    uint16_t* gi = reinterpret_cast<uint16_t*>(misc + 32);       // [CAP]
    uint16_t* rk = gi + CAP;                                     // [CAP] rank by pixel (SORT_MATCH only)

    const int col = blockIdx.x, seg = col / a.C, c = col % a.C;
    const int xseg = (a.x_n_seg == 1) ? 0 : seg;
    const float* src = a.keys + (size_t)xseg * a.ss + (size_t)c * a.ld;
    const int n = (int)a.n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    SORT_PROBE(0);
    // ---- 0. the column, in pixel order: element e = r * 1024 + tid (coalesced), pads beyond n are ignored everywhere
    uint32_t key[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const int e = r * SORT_NT + tid;
        const float v = src[e < n ? e : n - 1];
        key[r] = f2key(v);
    }
    // zero the tables while the loads are in flight
    for (int i = tid; i < NW2; i += SORT_NT) sc[i] = 0u;
    for (int i = tid; i < NWORDS + 4; i += SORT_NT) bs[i] = 0u;
    if (tid < RK_COARSE) c1[tid] = 0u;
    if (tid < 32) misc[tid] = 0u;

    // ---- 1. min / max
    uint32_t klo = 0xffffffffu, khi = 0u;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (r * SORT_NT + tid < n) {
            klo = key[r] < klo ? key[r] : klo;
            khi = key[r] > khi ? key[r] : khi;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t l2 = __shfl_xor(klo, o), h2 = __shfl_xor(khi, o);
        klo = l2 < klo ? l2 : klo;
        khi = h2 > khi ? h2 : khi;
    }
    if (lane == 0) {
        red[w] = klo;
        red[16 + w] = khi;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_NW; k++) {
        klo = red[k] < klo ? red[k] : klo;
        khi = red[16 + k] > khi ? red[16 + k] : khi;
    }
    __syncthreads();  // red is reused by the scans
    SORT_PROBE(1);
    // non-finite keys (inf / nan of either sign) cannot be bucketed by value: radix kernel
    if (khi >= 0xff800000u || klo <= 0x007fffffu) {
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    const float lo = key2f(klo), hi = key2f(khi);
    const bool all_equal = (klo == khi);
    float s1 = 0.f;
    if (!all_equal) {
        s1 = __fdiv_rn((float)RK_COARSE, hi - lo);
        if (!(s1 > 0.f) || !(s1 < 3.0e38f)) {  // range over/underflow (or only -0 / +0): radix kernel
            if (tid == 0) a.flags[col] = 1;
            return;
        }
    }

    float* ok = (MODE == SORT_EMIT && a.out_keys) ? a.out_keys + (size_t)col * n : nullptr;
    uint32_t* oi = (MODE == SORT_EMIT && a.out_idx) ? a.out_idx + (size_t)col * n : nullptr;
    // what happens once the stable rank of (key k, pixel idx) is known
    // what happens once the stable rank of (key k, pixel idx) is known.  Emit mode stores straight to HBM: the ranks of
    // neighbouring grouped positions fall into the same few cache lines, and the stores overlap with the ranking of the
    // other waves (staging the permutation in LDS and storing linearly measured 15 % slower: it serialises a
    // store-only tail that every CU reaches at the same moment).
    auto emit = [&](uint32_t k, uint32_t idx, uint32_t rank) {
        if (MODE == SORT_MATCH) {
            rk[idx] = (uint16_t)rank;
        } else {
            if (ok) ok[rank] = key2f(k);
            if (oi) oi[rank] = idx;
        }
    };

    if (all_equal) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const int e = r * SORT_NT + tid;
            if (e < n) {  // already sorted: rank = pixel index
                if (MODE == SORT_MATCH) rk[e] = (uint16_t)e;
                if (ok) ok[e] = key2f(key[r]);
                if (oi) oi[e] = (uint32_t)e;
            }
        }
    } else {
        // ---- 2. coarse histogram of a SAMPLE (every RS-th row of 1024 pixels, i.e. rows spread over the whole image).
        //         Any table of widths gives a monotone bucket map; the sample only has to balance the bucket sizes.
        constexpr int RS = ITEMS >= 8 ? 4 : 1;
        unsigned mine = 0;
#pragma unroll
        for (int r = 0; r < ITEMS; r += RS) {
            if (r * SORT_NT + tid < n) {
                const float t = (key2f(key[r]) - lo) * s1;
                int bin = (int)t;
                bin = bin > RK_COARSE - 1 ? RK_COARSE - 1 : bin;
                atomicAdd(&c1[bin], 1u);
                mine++;
            }
        }
        __syncthreads();
        SORT_PROBE(2);
        // ---- 3. equalisation widths and bases (all threads take part in the scans)
        {
            unsigned nsamp = 0;
            (void)block_excl_scan(mine, red, &nsamp);
            const unsigned cnt = tid < RK_COARSE ? c1[tid] : 0u;
            // cnt * NB < 2^27: exact 32-bit quotient via a float estimate and one correction step
            unsigned q = 0;
            if (tid < RK_COARSE) {
                const unsigned x = cnt * (unsigned)NB;
                q = (unsigned)((float)x / (float)nsamp);
                if (q * nsamp > x) q--;
                else if ((q + 1u) * nsamp <= x) q++;
            }
            const unsigned wd = tid < RK_COARSE ? 1u + q : 0u;
            const unsigned base = block_excl_scan(wd, red, nullptr);
            if (tid < RK_COARSE) c1[tid] = base | (wd << 16);
        }
        __syncthreads();
        SORT_PROBE(3);
        // ---- 4. fine bucket of every element (16 bits, two to a register), packed u16 bucket counts
        uint32_t st[(ITEMS + 1) / 2];
#pragma unroll
        for (int q = 0; q < (ITEMS + 1) / 2; q++) st[q] = 0u;
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            if (r * SORT_NT + tid < n) {
                const float t = (key2f(key[r]) - lo) * s1;
                int bin = (int)t;
                bin = bin > RK_COARSE - 1 ? RK_COARSE - 1 : bin;
                const float frac = t - (float)bin;
                const uint32_t bw = c1[bin];
                const int wd = (int)(bw >> 16);
                int sub = (int)(frac * (float)wd);
                sub = sub > wd - 1 ? wd - 1 : sub;
                const uint32_t b = (bw & 0xffffu) + (uint32_t)sub;
                atomicAdd(&sc[b >> 1], (b & 1u) ? 0x10000u : 1u);
                st[r >> 1] |= b << ((r & 1) * 16);
            }
        }
        __syncthreads();
        SORT_PROBE(4);
        // ---- 5. exclusive scan of the bucket counts -> bucket starts (in place), start bitmap, oversized buckets
        {
            uint32_t wv[PER];
            unsigned sum = 0;
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int i = tid * PER + q;
                wv[q] = i < NW2 ? sc[i] : 0u;
                sum += (wv[q] & 0xffffu) + (wv[q] >> 16);
            }
            unsigned ex = block_excl_scan(sum, red, nullptr);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int i = tid * PER + q;
                if (i < NW2) {
                    const unsigned c0 = wv[q] & 0xffffu, c1v = wv[q] >> 16;
                    const unsigned s0 = ex, s1v = ex + c0;
                    sc[i] = s0 | (s1v << 16);
                    if (c0) atomicOr(&bs[s0 >> 5], 1u << (s0 & 31u));
                    if (c1v) atomicOr(&bs[s1v >> 5], 1u << (s1v & 31u));
                    if (c0 > RK_BIG) {
                        const unsigned k = atomicAdd(&misc[0], 1u);
                        if (k < RK_MAXBIG) { misc[2 + 2 * k] = s0; misc[3 + 2 * k] = c0; }
                    }
                    if (c1v > RK_BIG) {
                        const unsigned k = atomicAdd(&misc[0], 1u);
                        if (k < RK_MAXBIG) { misc[2 + 2 * k] = s1v; misc[3 + 2 * k] = c1v; }
                    }
                }
                ex += (wv[q] & 0xffffu) + (wv[q] >> 16);
            }
            if (tid == 0) atomicOr(&bs[n >> 5], 1u << (n & 31)); // sentinel: the position after the last bucket
        }
        __syncthreads();
        SORT_PROBE(5);
        const unsigned nbig = misc[0];
        if (nbig > RK_MAXBIG) {
            if (tid == 0) a.flags[col] = 1;
            return;
        }
        // ---- 6a. keys and pixel indices into their bucket (slot = arrival order; the ranking does not depend on it)
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const int e = r * SORT_NT + tid;
            if (e < n) {
                const uint32_t b = (st[r >> 1] >> ((r & 1) * 16)) & 0xffffu;
                const uint32_t old = atomicAdd(&sc[b >> 1], (b & 1u) ? 0x10000u : 1u);
                const uint32_t pos = (b & 1u) ? (old >> 16) : (old & 0xffffu);
                gk[pos] = key[r];
                gi[pos] = (uint16_t)e;
            }
        }
        __syncthreads();
        SORT_PROBE(6);
        // ---- 6b. GROUPED order from here on (registers of steps 0-6a are dead): position p looks up the bounds of its
        //          bucket in the start bitmap and counts the smaller keys in it; neighbouring lanes share buckets, so the
        //          LDS reads are broadcasts.  4 positions per thread and trip: 4 reads in flight.
        constexpr int G = ITEMS < 8 ? ITEMS : 8;
        const unsigned long long* bs64 = reinterpret_cast<const unsigned long long*>(bs);  // [NWORDS / 2 + 1]
#pragma unroll 1
        for (int p0 = 0; p0 < CAP; p0 += G * SORT_NT) {
            uint32_t ps[G], pc[G], pk[G], lt[G], eq[G];
            uint32_t trips = 0;
#pragma unroll
            for (int q = 0; q < G; q++) {
                const int p = p0 + q * SORT_NT + tid;
                ps[q] = 0u; pc[q] = 0u; pk[q] = 0u; lt[q] = 0u; eq[q] = 0u;
                if (p < n) {
                    bool in_big = false;
                    for (unsigned k = 0; k < nbig; k++) {
                        const uint32_t bs0 = misc[2 + 2 * k], bc0 = misc[3 + 2 * k];
                        in_big = in_big || ((uint32_t)p - bs0 < bc0);
                    }
                    if (!in_big) {
                        // the 64 positions of this wave trip are one aligned 64-bit word of the bitmap (wave-uniform);
                        // a bucket of <= RK_BIG < 64 positions starts in it or in the word before and ends in it or the next
                        const int wq = p >> 6, lb = p & 63;
                        const unsigned long long B = bs64[wq];
                        const unsigned long long A = wq > 0 ? bs64[wq - 1] : 0ull;
                        const unsigned long long Cw = bs64[wq + 1];
                        const unsigned long long le = B & (~0ull >> (63 - lb));          // starts at or before p
                        const unsigned long long gt = lb == 63 ? 0ull : (B & (~0ull << (lb + 1)));  // starts after p
                        const uint32_t s = le ? (uint32_t)(wq * 64 + 63 - __clzll(le)) : (uint32_t)((wq - 1) * 64 + 63 - __clzll(A));
                        const uint32_t e2 = gt ? (uint32_t)(wq * 64 + __builtin_ctzll(gt)) : (uint32_t)((wq + 1) * 64 + __builtin_ctzll(Cw));
                        ps[q] = s;
                        pc[q] = e2 - s;
                        pk[q] = gk[p];
                        trips = pc[q] > trips ? pc[q] : trips;
                    }
                }
            }
            for (uint32_t j = 0; j < trips; j++) {
#pragma unroll
                for (int q = 0; q < G; q++) {
                    const uint32_t live = j < pc[q] ? 1u : 0u;
                    const uint32_t kj = gk[ps[q] + (live ? j : 0u)];
                    lt[q] += (kj < pk[q]) ? live : 0u;
                    eq[q] += (kj == pk[q]) ? live : 0u;
                }
            }
#pragma unroll
            for (int q = 0; q < G; q++) {
                const int p = p0 + q * SORT_NT + tid;
                if (pc[q]) {
                    const uint32_t idx = gi[p];
                    if (eq[q] > 1u) {  // equal keys in the bucket (itself included): order them by pixel index
                        for (uint32_t j = 0; j < pc[q]; j++)
                            if (gk[ps[q] + j] == pk[q] && (uint32_t)gi[ps[q] + j] < idx) lt[q]++;
                    }
                    emit(pk[q], idx, ps[q] + lt[q]);
                }
            }
        }
        SORT_PROBE(7);
        // ---- 6c. oversized buckets: all keys equal -> rank of the pixel index through a bitmap; otherwise radix kernel
        for (unsigned bi = 0; bi < nbig; bi++) {
            const uint32_t s = misc[2 + 2 * bi], cnt = misc[3 + 2 * bi];
            const uint32_t k0 = gk[s];
            for (int i = tid; i < NWORDS; i += SORT_NT) bitmap[i] = 0u;
            __syncthreads();
            for (uint32_t j = tid; j < cnt; j += SORT_NT) {
                if (gk[s + j] != k0) misc[1] = 1u;
                const uint32_t idx = gi[s + j];
                atomicOr(&bitmap[idx >> 5], 1u << (idx & 31u));
            }
            __syncthreads();
            if (misc[1]) {
                if (tid == 0) a.flags[col] = 1;
                return;
            }
            {
                const unsigned pcn = tid < NWORDS ? (unsigned)__popc(bitmap[tid]) : 0u;
                const unsigned ex = block_excl_scan(pcn, red, nullptr);
                if (tid < NWORDS) bpre[tid] = ex;
            }
            __syncthreads();
            for (uint32_t j = tid; j < cnt; j += SORT_NT) {
                const uint32_t idx = gi[s + j];
                emit(k0, idx, s + bpre[idx >> 5] + (uint32_t)__popc(bitmap[idx >> 5] & ((1u << (idx & 31u)) - 1u)));
            }
            __syncthreads();
        }
    }

    SORT_PROBE(8);
    // ---- 7. match: out[i] = sorted_source[q(rank_i)] in pixel order (coalesced)
    if (MODE == SORT_MATCH) {
        __syncthreads();
        const int sseg = (a.src_n_seg == 1) ? 0 : seg;
        const float* ssrt = a.src_sorted + ((size_t)sseg * a.C + c) * a.ns;
        float* o = a.out + (size_t)seg * a.oss + (size_t)c * a.ldo;
        const unsigned ns = (unsigned)a.ns;
        for (int e = tid; e < n; e += SORT_NT) o[e] = ssrt[quantile_index((uint32_t)rk[e], ns, (unsigned)n, a.inv_2nt)];
    }
    SORT_PROBE(9);
}

template <int ITEMS>
static constexpr size_t rank_lds_bytes(bool match) {
    constexpr int CAP = ITEMS * SORT_NT;
    return (size_t)CAP * 4 + (size_t)(CAP / 2 + RK_COARSE) / 2 * 4 + RK_COARSE * 4 + (size_t)(CAP / 32 + 4) * 4 +
           2 * (size_t)(CAP / 32) * 4 + 32 * 4 + 32 * 4 + (size_t)CAP * 2 + (match ? (size_t)CAP * 2 : 0);
}

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
// OPTEX_SORT_PATH=radix forces the general kernel, =rank1 the one-column-per-CU ranking kernel for the match as well,
// =rank2 the slot-ranked two-per-CU match kernel of sort_rank2.hip, =rank3 the integer-key owner-ranked one of
// sort_rank3.hip instead of the float-domain owner-ranked default of sort_rank4.hip (tests, comparisons)
static int sort_path_override() {
    static const int v = [] {
        const char* e = getenv("OPTEX_SORT_PATH");
        if (!e) return 0;
        if (e[0] == 'r' && e[1] == 'a' && e[2] == 'd') return 1;
        if (e[0] == 'r' && e[1] == 'a' && e[2] == 'n' && e[3] == 'k') return e[4] == '1' ? 2 : (e[4] == '2' ? 3 : (e[4] == '3' ? 4 : 0));
        return 0;
    }();
    return v;
}

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
    const bool use_rank = flags != nullptr && a.n >= RK_MIN_N && sort_path_override() != 1;
    a.flags = flags;
    a.ncols = ncols;
    a.inv_2nt = 1.0 / (2.0 * (double)a.n);
    int rc;
    if (use_rank) {
        static DeviceOnce attr_rank;
        auto rkern = rank_columns_kernel<ITEMS, MODE>;
        if ((rc = set_lds(rkern, rank_lds_bytes<ITEMS>(MODE == SORT_MATCH), attr_rank.slot()))) return rc;
        hipError_t e = hipMemsetAsync(flags, 0, sizeof(int) * (size_t)ncols, st);
        if (e != hipSuccess) {
            set_error("sort: memset failed: %s", hipGetErrorString(e));
            return OPTEX_E_LAUNCH;
        }
        {
            ProfScope prof(MODE == SORT_EMIT ? KC_SORT : KC_SORT_MATCH, st, 0.0, per_elem * (double)a.n * ncols);
            if (MODE == SORT_MATCH && sort_path_override() == 3) {
                if ((rc = launch_rank_match(ITEMS, a, ncols, st))) return rc;   // slot-ranked, two columns per CU (sort_rank2.hip)
            } else if (MODE == SORT_MATCH && sort_path_override() == 4) {
                if ((rc = launch_rank_match3(ITEMS, a, ncols, st))) return rc;  // owner-ranked, integer keys (sort_rank3.hip)
            } else if (MODE == SORT_MATCH && sort_path_override() != 2) {
                if ((rc = launch_rank_match4(ITEMS, a, ncols, st))) return rc;  // owner-ranked, float domain (sort_rank4.hip)
            } else if (MODE == SORT_EMIT && sort_path_override() == 0) {
                if ((rc = launch_rank_emit4(ITEMS, a, ncols, st))) return rc;   // the same kernel, keys / indices by rank
            } else {
                hipLaunchKernelGGL(rkern, dim3(ncols), dim3(SORT_NT), rank_lds_bytes<ITEMS>(MODE == SORT_MATCH), st, a);
            }
        }
        if ((rc = check_launch("rank_columns_kernel"))) return rc;
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

int sort_match_impl(const float* target, long ldt, long tss, long nt, const float* source, long lds, long sss, long ns,
                    int src_n_seg, int C, int n_seg, float* out, long ldo, long oss, void* ws, hipStream_t st) {
    // ws: [flags for the larger launch][sorted source keys [src_n_seg, C, ns]][scratch of the large-column path]
    int* flags = static_cast<int*>(ws);
    float* ssorted = reinterpret_cast<float*>(static_cast<char*>(ws) + flags_bytes(C * n_seg));
    void* large_ws = reinterpret_cast<char*>(ssorted) + align_up((size_t)src_n_seg * C * ns * sizeof(float), 256);
    // 1. sort the source columns (keys only)
    SortArgs s{};
    s.keys = source; s.ld = lds; s.ss = sss; s.n = ns; s.C = C; s.x_n_seg = src_n_seg;
    s.out_keys = ssorted; s.out_idx = nullptr;
    int rc = launch_sort<SORT_EMIT>(s, C * src_n_seg, flags, large_ws, st);
    if (rc) return rc;
    // 2. rank each target column and fetch the source quantiles
    SortArgs t{};
    t.keys = target; t.ld = ldt; t.ss = tss; t.n = nt; t.C = C; t.x_n_seg = n_seg;
    t.src_sorted = ssorted; t.ns = ns; t.src_n_seg = src_n_seg;
    t.out = out; t.ldo = ldo; t.oss = oss;
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
    return flags_bytes(C * (n_seg > src_n_seg ? n_seg : src_n_seg)) + align_up((size_t)src_n_seg * C * ns * sizeof(float), 256) +
           large;
}

extern "C" int optex_sort_match(const float* target, long ldt, long t_seg_stride, long nt, const float* source,
                                long lds, long s_seg_stride, long ns, int src_n_seg, int C, int n_seg, float* out,
                                long ldo, long o_seg_stride, void* ws, size_t ws_bytes, void* stream) {
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
