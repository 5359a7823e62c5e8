// sort_rank2.hip — rank_match_kernel: the exact 1-D transport match (north-star addition; specification =
// oracle/optex_oracle.c orc_sort_match) of one column per 1024-thread workgroup in < 80 KiB of LDS, so that TWO columns
// are resident per CU: while one workgroup waits on HBM (column load, source quantiles, store tail) or on a barrier,
// the other one computes.  rank_columns_kernel (sort.hip) needs ~150 KiB for a 16384-key column and therefore runs one
// column per CU with nothing to overlap its seven barrier-separated phases with.
//
// Same idea as sort.hip — RANKING BY COUNTING through a histogram-equalised monotone bucket map — with these changes:
//   * the bucket slots hold ONE packed word per key, (sub << 14 | pixel): `sub` is an 18-bit monotone refinement of the
//     key's place inside its bucket, so an unsigned compare of two words of the same bucket orders them by key and,
//     for equal keys, by pixel index — the stable order — in one instruction.  Two DIFFERENT keys of one bucket share
//     a `sub` a dozen times per column (fp32 resolution of the bucket coordinate), equal keys always do: those slots
//     are queued, and one thread per queued slot re-reads the real keys from the (L2-resident) column and compares
//     (key, pixel) exactly — all queued slots at once, so their memory latencies overlap.  No separate key / index /
//     rank arrays: 4 B of LDS per key.
//   * the matched value is fetched where the RANK is known: neighbouring slots have neighbouring ranks, so
//     sorted_source[q(rank)] is a coalesced read; values are scattered by pixel into the (dead) slot array and the
//     column leaves with 16-byte stores.
//   * 16-byte loads; the sample histogram size is known analytically (one block scan less).
// A variant that ranked the slots of a run across lanes with DPP (v_mov_b32_dpp wave_shr:1 + lane masks in SGPRs)
// instead of probing LDS was built and measured: correct, but every lane pays the whole window set-up and every
// compare trip costs ~20 instructions for < 1 slot per lane — 3x the instructions per key of the LDS loop below
// (DESIGN.md 4.2).
// Columns this kernel cannot take (non-finite keys, many distinct massive ties) are flagged for the radix kernel of
// sort.hip, which runs right behind it on the stream.
#include "sort_common.h"

namespace optex {

constexpr int R2_IDX_BITS = 14;  // pixel index inside a column, n <= 16384
constexpr uint32_t R2_IDX_MASK = (1u << R2_IDX_BITS) - 1u;
constexpr int R2_SUB_BITS = 18;
constexpr uint32_t R2_NONE = 0xffffffffu;
#ifndef R2_TMAX_VALUE
#define R2_TMAX_VALUE 6
#endif
constexpr int R2_TMAX = R2_TMAX_VALUE;  // runs longer than this are ranked one slot per thread (6d), not in the G-wide loop

template <int ITEMS>
struct R2 {
    static constexpr int CAP = ITEMS * SORT_NT;
    // fine buckets handed out by the equalisation: as many as 80 KiB of LDS (16 keys per thread) and the 13-bit bucket id
    // allow — fewer keys per bucket = fewer compare trips and fewer queued slots
    static constexpr int NB = ITEMS == 16 ? CAP * 3 / 8 : (ITEMS == 12 ? CAP * 5 / 8 : (ITEMS == 8 ? CAP * 15 / 16 : CAP));
    static constexpr int NBT = NB + RK_COARSE;                       // + 1 per coarse bin; even; < 2^13
    static constexpr int NW2 = NBT / 2;                              // packed u16 counters
    static constexpr int PER = (NW2 + SORT_NT - 1) / SORT_NT;
    static constexpr int NWORDS = CAP / 32;
    static constexpr int QCAP = NW2;  // queued slots per column (the queue aliases the dead counters)
    static constexpr size_t LDS = (size_t)(CAP + NW2 + RK_COARSE + NWORDS + 4 + 32 + 32) * 4;
    static_assert(NBT < (1 << (32 - R2_SUB_BITS - 1)), "bucket id and sub must fit 31 bits");
    static_assert(2 * NWORDS <= NW2, "big-bucket scratch aliases the counters");
};

// acc += (a < b): compare + add-with-carry, two instructions
__device__ __forceinline__ void add_if_less(uint32_t& acc, uint32_t a, uint32_t b) {
    asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <int ITEMS, bool VEC>
__global__ __launch_bounds__(SORT_NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void rank_match_kernel(SortArgs a) {
    using K = R2<ITEMS>;
    constexpr int CAP = K::CAP, NB = K::NB, NW2 = K::NW2, PER = K::PER, NWORDS = K::NWORDS, QCAP = K::QCAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* slot = reinterpret_cast<uint32_t*>(smem);  // [CAP] sub << 14 | pixel by bucket position; later the output
    uint32_t* cnt = slot + CAP;                          // [NW2] packed u16 bucket counts -> starts -> cursors
    uint32_t* c1 = cnt + NW2;                            // [256] coarse histogram, then base | width << 16
    uint32_t* bs = c1 + RK_COARSE;                       // [NWORDS + 4] bit p = slot p starts a bucket
    uint32_t* red = bs + NWORDS + 4;                     // [32]
    uint32_t* misc = red + 32;                           // [32] nbig, noteq, (start, count) x RK_MAXBIG, [20] queue length
    uint32_t* queue = cnt;                               // [QCAP] slots for the one-per-thread pass (6c/6d), then their results
    uint32_t* bitmap = cnt;                              // [NWORDS] big-bucket pass (counters are dead by then)
    uint32_t* bpre = cnt + NWORDS;                       // [NWORDS]

    const int col = blockIdx.x, seg = col / a.C, c = col % a.C;
    const int xseg = (a.x_n_seg == 1) ? 0 : seg;
    const float* src = a.keys + (size_t)xseg * a.ss + (size_t)c * a.ld;
    const int sseg = (a.src_n_seg == 1) ? 0 : seg;
    const float* ssrt = a.src_sorted + ((size_t)sseg * a.C + c) * a.ns;
    float* o = a.out + (size_t)seg * a.oss + (size_t)c * a.ldo;
    const unsigned ns = (unsigned)a.ns;
    const int n = (int)a.n;
    const int tid = threadIdx.x, lane = tid & 63, w = (int)uniform((uint32_t)tid >> 6);
    // pixel held in register r: 16-byte loads put 4 neighbouring pixels into one thread
    auto elem = [&](int r) { return VEC ? ((r >> 2) * SORT_NT + tid) * 4 + (r & 3) : r * SORT_NT + tid; };

    SORT_PROBE(0);
    // ---- 0. the column
    uint32_t key[ITEMS];
    if (VEC) {
#pragma unroll
        for (int q = 0; q < ITEMS / 4; q++) {
            const int e0 = (q * SORT_NT + tid) * 4;
            const float4 v = *reinterpret_cast<const float4*>(src + (e0 < n ? e0 : 0));
            key[4 * q + 0] = f2key(v.x);
            key[4 * q + 1] = f2key(v.y);
            key[4 * q + 2] = f2key(v.z);
            key[4 * q + 3] = f2key(v.w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const int e = r * SORT_NT + tid;
            key[r] = f2key(src[e < n ? e : n - 1]);
        }
    }
    for (int i = tid; i < NW2; i += SORT_NT) cnt[i] = 0u;
    for (int i = tid; i < NWORDS + 4; i += SORT_NT) bs[i] = 0u;
    if (tid < RK_COARSE) c1[tid] = 0u;
    if (tid < 32) misc[tid] = 0u;

    // ---- 1. min / max
    uint32_t klo = 0xffffffffu, khi = 0u;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (elem(r) < n) {
            klo = key[r] < klo ? key[r] : klo;
            khi = key[r] > khi ? key[r] : khi;
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const uint32_t l2 = __shfl_xor(klo, s), h2 = __shfl_xor(khi, s);
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
    // (red is next written by the scan of step 5, two barriers from here)
    if (khi >= 0xff800000u || klo <= 0x007fffffu) {  // non-finite keys cannot be bucketed by value: radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    const float lo = key2f(klo), hi = key2f(khi);
    if (klo == khi) {  // constant column: already sorted, rank = pixel index
        for (int e = tid; e < n; e += SORT_NT) o[e] = ssrt[quantile_index((uint32_t)e, ns, (unsigned)n, a.inv_2nt)];
        return;
    }
    const float s1 = __fdiv_rn((float)RK_COARSE, hi - lo);
    if (!(s1 > 0.f) || !(s1 < 3.0e38f)) {  // range over/underflow (or only -0 / +0): radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }

    SORT_PROBE(1);
    // ---- 2. coarse histogram of a spatially spread quarter sample (any widths give a monotone map; the sample only
    //         balances the bucket sizes)
    constexpr int RS = VEC ? 4 : (ITEMS >= 8 ? 4 : 1);
    unsigned nsamp = 0;
    if (VEC) {
        nsamp = (unsigned)(n + 3) / 4u;
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; r += RS) {
            const int left = n - r * SORT_NT;
            nsamp += (unsigned)(left < 0 ? 0 : (left > SORT_NT ? SORT_NT : left));
        }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r += RS) {
        if (elem(r) < n) {
            const float t = (key2f(key[r]) - lo) * s1;
            int bin = (int)t;
            bin = bin > RK_COARSE - 1 ? RK_COARSE - 1 : bin;
            atomicAdd(&c1[bin], 1u);
        }
    }
    __syncthreads();
    SORT_PROBE(2);
    // ---- 3. equalisation: coarse bin b gets w_b = 1 + cnt_b * NB / nsamp fine buckets.  One wavefront, four bins per
    //         lane: no block-wide scan, one barrier
    if (w == 0) {
        const uint4 c = *reinterpret_cast<const uint4*>(c1 + 4 * lane);
        auto width = [&](unsigned cnt) {
            const unsigned x = cnt * (unsigned)NB;  // < 2^27: exact quotient via a float estimate + one correction
            unsigned q = (unsigned)((float)x / (float)nsamp);
            if (q * nsamp > x) q--;
            else if ((q + 1u) * nsamp <= x) q++;
            return 1u + q;
        };
        const unsigned w0 = width(c.x), w1 = width(c.y), w2 = width(c.z), w3 = width(c.w);
        const unsigned sum = w0 + w1 + w2 + w3;
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const unsigned b0 = incl - sum, b1 = b0 + w0, b2 = b1 + w1, b3 = b2 + w2;
        *reinterpret_cast<uint4*>(c1 + 4 * lane) = make_uint4(b0 | (w0 << 16), b1 | (w1 << 16), b2 | (w2 << 16), b3 | (w3 << 16));
    }
    __syncthreads();
    SORT_PROBE(3);
    // ---- 4. fine bucket b and refinement sub of every key; the register now holds b << 18 | sub.  x -> (b, sub) is
    //         monotone non-decreasing whatever the rounding: every step (subtract, scale, truncate, clamp) is.
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint32_t packed = R2_NONE;
        if (elem(r) < n) {
            const float t = (key2f(key[r]) - lo) * s1;
            int bin = (int)t;
            bin = bin > RK_COARSE - 1 ? RK_COARSE - 1 : bin;
            const float frac = t - (float)bin;
            const uint32_t bw = c1[bin];
            const int wd = (int)(bw >> 16);
            const float u = frac * (float)wd;
            int sub = (int)u;
            sub = sub > wd - 1 ? wd - 1 : sub;
            int fine = (int)((u - (float)sub) * (float)(1 << R2_SUB_BITS));
            fine = fine > (1 << R2_SUB_BITS) - 1 ? (1 << R2_SUB_BITS) - 1 : fine;
            const uint32_t b = (bw & 0xffffu) + (uint32_t)sub;
            atomicAdd(&cnt[b >> 1], (b & 1u) ? 0x10000u : 1u);
            packed = (b << R2_SUB_BITS) | (uint32_t)fine;
        }
        key[r] = packed;
    }
    __syncthreads();
    SORT_PROBE(4);
    // ---- 5. exclusive scan of the bucket counts -> starts (in place), start bitmap, oversized buckets
    {
        uint32_t wv[PER];
        unsigned sum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int i = tid * PER + q;
            wv[q] = i < NW2 ? cnt[i] : 0u;
            sum += (wv[q] & 0xffffu) + (wv[q] >> 16);
        }
        unsigned ex = block_excl_scan(sum, red, nullptr);
        uint32_t curw = 0u, bits = 0u;  // a thread's buckets start at increasing positions: one atomicOr per word
        auto mark = [&](unsigned pos) {
            const uint32_t wd = pos >> 5;
            if (wd != curw) {
                if (bits) atomicOr(&bs[curw], bits);
                curw = wd;
                bits = 0u;
            }
            bits |= 1u << (pos & 31u);
        };
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int i = tid * PER + q;
            if (i < NW2) {
                const unsigned c0 = wv[q] & 0xffffu, c1v = wv[q] >> 16;
                const unsigned s0 = ex, s1v = ex + c0;
                cnt[i] = s0 | (s1v << 16);
                if (c0) mark(s0);
                if (c1v) mark(s1v);
                if (c0 > RK_BIG) {
                    const unsigned k = atomicAdd(&misc[0], 1u);
                    if (k < RK_MAXBIG) { misc[2 + 2 * k] = s0; misc[3 + 2 * k] = c0; }
                }
                if (c1v > RK_BIG) {
                    const unsigned k = atomicAdd(&misc[0], 1u);
                    if (k < RK_MAXBIG) { misc[2 + 2 * k] = s1v; misc[3 + 2 * k] = c1v; }
                }
                ex += c0 + c1v;
            }
        }
        if (bits) atomicOr(&bs[curw], bits);
        if (tid == 0) atomicOr(&bs[n >> 5], 1u << (n & 31));  // sentinel: the position after the last bucket
    }
    __syncthreads();
    const unsigned nbig = misc[0];
    if (nbig > RK_MAXBIG) {
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    SORT_PROBE(5);
    // ---- 6a. every key takes a slot of its bucket (arrival order; the ranking does not depend on it)
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (key[r] != R2_NONE) {
            const uint32_t b = key[r] >> R2_SUB_BITS;
            const uint32_t old = atomicAdd(&cnt[b >> 1], (b & 1u) ? 0x10000u : 1u);
            const uint32_t pos = (b & 1u) ? (old >> 16) : (old & 0xffffu);
            slot[pos] = (key[r] << R2_IDX_BITS) | (uint32_t)elem(r);  // the shift drops b, keeps sub
        }
    }
    __syncthreads();
#ifdef R2_DEBUG
    if (blockIdx.x == 0) {
        for (int i = tid; i < n; i += SORT_NT) a.dbg[i] = slot[i];
        for (int i = tid; i < NWORDS + 4; i += SORT_NT) a.dbg[CAP + i] = bs[i];
    }
    __syncthreads();
#endif
    // ---- 6b. oversized buckets only come from exact ties: if all keys of such a bucket are equal its ranks are the
    //          ranks of the pixel indices (bitmap + popcount prefix); the slots become pixel << 14 | rank in place.
    //          Anything else -> radix kernel.
    for (unsigned bi = 0; bi < nbig; bi++) {
        const uint32_t s = misc[2 + 2 * bi], cb = misc[3 + 2 * bi];
        const uint32_t k0 = f2key(src[slot[s] & R2_IDX_MASK]);
        for (int i = tid; i < NWORDS; i += SORT_NT) bitmap[i] = 0u;
        __syncthreads();
        for (uint32_t j = tid; j < cb; j += SORT_NT) {
            const uint32_t idx = slot[s + j] & R2_IDX_MASK;
            if (f2key(src[idx]) != k0) misc[1] = 1u;
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
        for (uint32_t j = tid; j < cb; j += SORT_NT) {
            const uint32_t idx = slot[s + j] & R2_IDX_MASK;
            const uint32_t rank = s + bpre[idx >> 5] + (uint32_t)__popc(bitmap[idx >> 5] & ((1u << (idx & 31u)) - 1u));
            slot[s + j] = (idx << R2_IDX_BITS) | rank;
        }
        __syncthreads();
    }
    SORT_PROBE(6);
    // ---- 6c. ranks, slot side: slot p finds the bounds of its run in a 14-bit window of the start bitmap and counts
    //          the smaller words of the run; neighbouring lanes share runs, so the LDS
    //          reads are broadcasts.  G slots per thread and trip keep G reads in flight.  res[r] = pixel << 14 | rank
    //          of slot r * 1024 + tid.  A run in which two words share a sub goes to the exact path (6d).
    constexpr int G = ITEMS < 4 ? ITEMS : 4;
    uint32_t res[ITEMS];
#pragma unroll
    for (int r0 = 0; r0 < ITEMS; r0 += G) {
        uint32_t ps[G], pc[G], pk[G], lt[G], eq[G];
        uint32_t trips = 0;
#pragma unroll
        for (int q = 0; q < G; q++) {
            const int p = (r0 + q) * SORT_NT + tid;
            ps[q] = 0u; pc[q] = 0u; pk[q] = 0u; lt[q] = 0u; eq[q] = 0u;
            res[r0 + q] = R2_NONE;
            if (p < n) {
                pk[q] = slot[p];
                // bucket starts among the slots lo .. lo + 2 T + 1 around p (lo = p - T, T = R2_TMAX): runs of <= T slots
                // have both ends inside this window; everything longer is ranked by the one-per-thread pass
                const int lo = p > R2_TMAX ? p - R2_TMAX : 0, d = p - lo;
                const int wi = lo >> 5;
                const unsigned long long two = (unsigned long long)bs[wi] | ((unsigned long long)bs[wi + 1] << 32);
                const uint32_t win = (uint32_t)(two >> (lo & 31));                 // bit i: slot lo + i starts a run
                const uint32_t below = win & ((2u << d) - 1u);                     // starts at lo .. p
                const uint32_t above = (win >> (d + 1)) & ((2u << R2_TMAX) - 1u);  // starts at p + 1 .. p + R2_TMAX + 1
                bool in_big = false;  // the all-equal pass left pixel << 14 | rank in the slots of oversized buckets
                for (unsigned bi = 0; bi < nbig; bi++) in_big = in_big || ((uint32_t)p - misc[2 + 2 * bi] < misc[3 + 2 * bi]);
                const uint32_t s = (uint32_t)(lo + 31 - __clz(below | 0u)), e2 = (uint32_t)(p + 1 + __builtin_ctz(above | (2u << R2_TMAX)));
                if (in_big) {
                    res[r0 + q] = pk[q];
                } else if (below == 0u || above == 0u || e2 - s > (uint32_t)R2_TMAX) {
                    const uint32_t qi = atomicAdd(&misc[20], 1u);  // long run: one-per-thread pass
                    if (qi < (uint32_t)QCAP) queue[qi] = (uint32_t)p;
                } else {
                    ps[q] = s;
                    pc[q] = e2 - s;
                    trips = pc[q] > trips ? pc[q] : trips;
                }
            }
        }
        // trip j looks at member j of the run; a lane whose run is shorter re-reads its own slot, which adds nothing
        // to lt and one to eq — no per-lane masking of the two counters
#pragma unroll 1
        for (uint32_t j = 0; j < trips; j++) {
#pragma unroll
            for (int q = 0; q < G; q++) {
                const uint32_t kj = slot[j < pc[q] ? ps[q] + j : (uint32_t)((r0 + q) * SORT_NT + tid)];
                add_if_less(lt[q], kj, pk[q]);
                add_if_less(eq[q], kj ^ pk[q], 1u << R2_IDX_BITS);
            }
        }
#pragma unroll
        for (int q = 0; q < G; q++) {
            if (pc[q]) {
                if (eq[q] == trips - pc[q] + 1u) {  // only itself shares its sub
                    res[r0 + q] = ((pk[q] & R2_IDX_MASK) << R2_IDX_BITS) | (ps[q] + lt[q]);
                } else {
                    const uint32_t qi = atomicAdd(&misc[20], 1u);
                    if (qi < (uint32_t)QCAP) queue[qi] = (uint32_t)((r0 + q) * SORT_NT + tid);
                }
            }
        }
        asm volatile("" ::: "memory");  // keep the next trip's bitmap reads below this trip's loop (64-VGPR budget)
    }
    SORT_PROBE(7);
    // ---- 6d. queued slots, one per thread: runs longer than R2_TMAX (the G-wide loop above would make every lane wait
    //          for the longest run of its wave) and runs in which two words share a sub — there the real keys decide
    //          (re-read from the L2-resident column), then the pixels.  The queue entry becomes pixel << 14 | rank.
    __syncthreads();
    const uint32_t qn = misc[20];
    if (qn > (uint32_t)QCAP) {  // tie-heavy column: radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    for (uint32_t i = tid; i < qn; i += SORT_NT) {
        const int p = (int)queue[i];
        const uint32_t my = slot[p];
        int wa = p >> 5;
        uint32_t m = bs[wa] & (0xffffffffu >> (31 - (p & 31)));
        while (m == 0u) m = bs[--wa];
        const int S = wa * 32 + 31 - __clz(m);
        const int q1 = p + 1;
        int wb = q1 >> 5;
        m = bs[wb] & (0xffffffffu << (q1 & 31));
        while (m == 0u) m = bs[++wb];  // the sentinel at n ends the search
        const int E = wb * 32 + __builtin_ctz(m);
        uint32_t lt = 0u, ties = 0u;
        for (int j = S; j < E; j++) {
            const uint32_t oj = slot[j];
            add_if_less(lt, oj, my);
            add_if_less(ties, oj ^ my, 1u << R2_IDX_BITS);
        }
        if (ties > 1u) {  // some other word shares the sub: decide those pairs by the keys
            const uint32_t myk = f2key(src[my & R2_IDX_MASK]);
            lt = 0u;
            for (int j = S; j < E; j++) {
                const uint32_t oj = slot[j];
                if (j == p) continue;
                if (((oj ^ my) >> R2_IDX_BITS) == 0u) {
                    const uint32_t ok = f2key(src[oj & R2_IDX_MASK]);
                    lt += (ok < myk || (ok == myk && (oj & R2_IDX_MASK) < (my & R2_IDX_MASK))) ? 1u : 0u;
                } else {
                    lt += oj < my ? 1u : 0u;
                }
            }
        }
        queue[i] = ((my & R2_IDX_MASK) << R2_IDX_BITS) | (uint32_t)(S + (int)lt);
    }
#ifdef R2_DEBUG
    if (blockIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) a.dbg[2 * CAP + k * SORT_NT + tid] = res[k];
    }
#endif
    SORT_PROBE(8);
    // ---- 7. out[pixel] = sorted_source[q(rank)]: read where the ranks are neighbours (coalesced), scatter by pixel
    //         into the slot array (every slot has been read: barrier), leave with 16-byte stores
    __syncthreads();
    float* val = reinterpret_cast<float*>(slot);
    // the first queued result of this thread: its LDS read -> global read chain starts before the main loads
    const uint32_t qr = (uint32_t)tid < qn ? queue[tid] : R2_NONE;
    const float qv = ssrt[quantile_index(qr != R2_NONE ? (qr & R2_IDX_MASK) : 0u, ns, (unsigned)n, a.inv_2nt)];
    constexpr int FCH = 16;  // loads of a chunk are all in flight before the first LDS write waits for one
#pragma unroll
    for (int k0 = 0; k0 < ITEMS; k0 += FCH) {
        float v[FCH];
#pragma unroll
        for (int k = k0; k < k0 + FCH && k < ITEMS; k++) {
            const uint32_t rank = res[k] != R2_NONE ? (res[k] & R2_IDX_MASK) : 0u;
            v[k - k0] = ssrt[quantile_index(rank, ns, (unsigned)n, a.inv_2nt)];
        }
#pragma unroll
        for (int k = k0; k < k0 + FCH && k < ITEMS; k++)
            if (res[k] != R2_NONE) val[res[k] >> R2_IDX_BITS] = v[k - k0];
    }
    if (qr != R2_NONE) val[qr >> R2_IDX_BITS] = qv;
    for (uint32_t i = tid + SORT_NT; i < qn; i += SORT_NT) {  // each thread reads back the entries it wrote
        const uint32_t r = queue[i];
        val[r >> R2_IDX_BITS] = ssrt[quantile_index(r & R2_IDX_MASK, ns, (unsigned)n, a.inv_2nt)];
    }
    __syncthreads();
    SORT_PROBE(9);
    if (VEC && a.out_vec) {
#pragma unroll
        for (int q = 0; q < ITEMS / 4; q++) {
            const int e0 = (q * SORT_NT + tid) * 4;
            if (e0 < n) *reinterpret_cast<float4*>(o + e0) = *reinterpret_cast<const float4*>(val + e0);
        }
    } else {
        for (int e = tid; e < n; e += SORT_NT) o[e] = val[e];
    }
    SORT_PROBE(10);
}

template <int ITEMS>
static int launch_rank_match_items(SortArgs a, int ncols, hipStream_t st) {
    const bool in_vec = ITEMS >= 4 && a.n % 4 == 0 && a.ld % 4 == 0 && a.ss % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(a.keys) & 15u) == 0;
    a.out_vec = (a.ldo % 4 == 0 && a.oss % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0) ? 1 : 0;
    const size_t lds = R2<ITEMS>::LDS;
    hipError_t e;
    if (in_vec) {
        auto kern = rank_match_kernel<ITEMS, (ITEMS >= 4)>;
        static DeviceOnce once;
        bool& attr = *once.slot();
        if (!attr) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error("sort: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return OPTEX_E_LAUNCH; }
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(ncols), dim3(SORT_NT), lds, st, a);
    } else {
        auto kern = rank_match_kernel<ITEMS, false>;
        static DeviceOnce once;
        bool& attr = *once.slot();
        if (!attr) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error("sort: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return OPTEX_E_LAUNCH; }
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(ncols), dim3(SORT_NT), lds, st, a);
    }
    return check_launch("rank_match_kernel");
}

// called by launch_sort_items<ITEMS, SORT_MATCH> (sort.hip) with flags cleared and the prof scope open
int launch_rank_match(int items, const SortArgs& a, int ncols, hipStream_t st) {
    switch (items) {
        case 2: return launch_rank_match_items<2>(a, ncols, st);
        case 4: return launch_rank_match_items<4>(a, ncols, st);
        case 8: return launch_rank_match_items<8>(a, ncols, st);
        case 12: return launch_rank_match_items<12>(a, ncols, st);
        default: return launch_rank_match_items<16>(a, ncols, st);
    }
}

}  // namespace optex
