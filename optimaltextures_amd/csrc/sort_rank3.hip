// sort_rank3.hip — rank_match3_kernel: the exact 1-D transport match (north-star addition; specification =
// oracle/optex_oracle.c orc_sort_match) of one column per 1024-thread workgroup, two columns resident per CU, with the
// ranking done BY THE THREAD THAT OWNS THE KEY.
//
// Same skeleton as sort_rank2.hip — a histogram-equalised MONOTONE bucket map, bucket counts, exclusive scan, every key
// written once into a slot of its bucket, exact rank = position of the bucket + comparisons inside it — but the slots hold
// the plain 32-bit totalOrder key and three observations remove most of the instructions of the older kernel
// (measured there: 212 VALU + 68 SALU + 19 LDS wave-instructions per 64 keys, profiles/r02_sort_match_sq_counters.md):
//
//   1. The count atomic RETURNS the key's arrival number a inside its bucket, so after the scan the slot is
//      start[b] + a: one LDS read instead of a second atomic, and the owner keeps (b, a) in a register.
//   2. The bucket map is monotone over the whole column: every slot before a bucket holds a smaller key, every slot
//      behind it a larger one.  Hence  rank = W0 + #{ j in [W0, W0 + 8) : slot[j] < key }  for ANY 8-slot window that
//      covers the bucket — in particular the 16-byte aligned one W0 = start & ~3, two ds_read_b128 — with no bucket
//      bounds, no start bitmap, no per-trip addressing or clamping: 2 LDS instructions and 8 compare / add-carry pairs
//      per key.  Buckets that do not fit the window ((start & 3) + count > 8, ~2-3 % of the keys) are flagged by the
//      scan and ranked one key per thread from a queue.
//   3. The owner knows its pixel: it computes q(rank), reads the matched value from the sorted source column staged in
//      the (then dead) slot array and stores its own four neighbouring pixels with one 16-byte store — no scatter by
//      pixel, no (key, pixel) packing, hence no `sub` refinement and no false collisions.
//
// Ties: equal keys always share a bucket; an owner that finds another slot equal to its key joins the queue, where equal
// keys are ordered by pixel index (the stable order of the specification).  A bucket of more than RK_BIG keys must be
// all-equal (the zeros of un-rotated ReLU features): ranks = ranks of the pixel indices (bitmap + popcount prefix).
// Anything else (non-finite keys, many distinct massive ties, queue overflow) flags the column for the radix kernel of
// sort.hip, which runs right behind this one on the stream.
#include "sort_common.h"

namespace optex {

constexpr uint32_t R3_TAG = 0x80000000u;   // res: rank pending in queue entry (low bits)
constexpr uint32_t R3_DONE = 0x40000000u;  // meta: rank already final (all-equal big bucket)
constexpr uint32_t R3_LONG = 0x8000u;      // start entry: bucket does not fit the aligned 8-slot window
constexpr uint32_t R3_BIGF = 0x4000u;      // start entry: bucket larger than RK_BIG
constexpr uint32_t R3_SMASK = 0x3fffu;     // start entry: first slot of the bucket (mod 16384)
constexpr int R3_WIN = 8;
constexpr int R3_QWIN = 52;                // window of a queued key, slots: covers (start & 3) + RK_BIG
constexpr int R3_BBITS = 13;               // bucket id bits in the owner's (b, a) register

template <int ITEMS, int NT>
struct R3 {
    static constexpr int CAP = ITEMS * NT;
    // buckets + 1 spare per coarse bin; 16 keys per thread: what fits 80 KiB next to the 64 KiB of slots
    static constexpr int NBT = CAP == 16384 ? 7872 : (CAP < 8192 ? CAP : 8192);
    static constexpr int NB = NBT - RK_COARSE;
    static constexpr int NW2 = NBT / 2;                               // packed u16 counters -> start entries
    static constexpr int PER = (NW2 + NT - 1) / NT;
    static constexpr int NWORDS = CAP / 32;
    static constexpr int TCAP = 256;                                  // queue entries with an equal partner
    static constexpr int QCAP = (NW2 - TCAP) / 4;                     // key, window, pixel, result per queue entry
    static constexpr int SLOTW = CAP + R3_QWIN + 4;                   // + window padding behind the last key + a dummy slot
    static constexpr int CNTW = NW2 + 4;                              // + the entry behind the last bucket
    static constexpr size_t LDS = (size_t)(SLOTW + CNTW + 32 + 32) * 4;
    static_assert(NBT <= (1 << R3_BBITS), "bucket id must fit its bit field");
    static_assert(2 * NWORDS <= NW2, "big-bucket scratch aliases the counters");
    static_assert(NBT % 2 == 0 && RK_COARSE <= CAP, "layout");
    static_assert(R3_QWIN % 4 == 0 && R3_QWIN >= 3 + RK_BIG, "a queued key's window must cover its bucket");
};

// acc += (a < b), acc += (a <= b): compare + add-with-carry, two instructions each
__device__ __forceinline__ void r3_add_lt(uint32_t& acc, uint32_t a, uint32_t b) {
    asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void r3_add_le(uint32_t& acc, uint32_t a, uint32_t b) {
    asm("v_cmp_le_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
// the whole 8-slot window against key k in ONE asm block (16 compare / add-carry pairs; separate asm statements make
// the compiler pad every one of them with an s_nop):  lt += #{x < k},  le += #{x <= k}
__device__ __forceinline__ void r3_window(uint32_t& lt, uint32_t& le, const uint4& x0, const uint4& x1, uint32_t k) {
    asm("v_cmp_lt_u32 vcc, %2, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %2, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_u32 vcc, %3, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %3, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_u32 vcc, %4, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %4, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_u32 vcc, %5, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %5, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_u32 vcc, %6, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %6, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_u32 vcc, %7, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %7, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_u32 vcc, %8, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %8, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_u32 vcc, %9, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %9, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lt), "+v"(le)
        : "v"(x0.x), "v"(x0.y), "v"(x0.z), "v"(x0.w), "v"(x1.x), "v"(x1.y), "v"(x1.z), "v"(x1.w), "v"(k)
        : "vcc");
}

// NT threads per workgroup (1024, or 512 / 256 for short columns: the per-column steps are barrier-to-barrier latency
// chains, and smaller workgroups let more columns overlap on a CU — up to the 2048-thread limit — at 16 keys per thread)
template <int ITEMS, bool VEC, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void rank_match3_kernel(SortArgs a) {
    using K = R3<ITEMS, NT>;
    constexpr int NW = NT / 64;
    constexpr int CAP = K::CAP, NB = K::NB, NW2 = K::NW2, PER = K::PER, NWORDS = K::NWORDS, QCAP = K::QCAP, TCAP = K::TCAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* slot = reinterpret_cast<uint32_t*>(smem);   // [CAP + 52 + 4] keys by bucket position; later the sorted source column
    uint32_t* cnt = slot + K::SLOTW;                      // [NW2 + 4] packed u16 bucket counts -> start entries
    uint32_t* red = cnt + K::CNTW;                        // [32]
    uint32_t* misc = red + 32;                            // [32] nbig, noteq, (start, count) x RK_MAXBIG, [20] queue length
    uint32_t* c1 = slot;                                  // [256] coarse histogram, then base | width << 16 (dead before the slots fill)
    const unsigned short* st16 = reinterpret_cast<const unsigned short*>(cnt);
    uint32_t* qkey = cnt;                                 // queue (the start entries are dead by then)
    uint32_t* qwin = cnt + QCAP;
    uint32_t* qpix = cnt + 2 * QCAP;
    uint32_t* qres = cnt + 3 * QCAP;
    uint32_t* tlist = cnt + 4 * QCAP;                     // [TCAP] queue entries whose key has an equal partner
    uint32_t* bitmap = cnt;                               // [NWORDS] big-bucket pass
    uint32_t* bpre = cnt + NWORDS;                        // [NWORDS]

    const int col = blockIdx.x, seg = col / a.C, c = col % a.C;
    const int xseg = (a.x_n_seg == 1) ? 0 : seg;
    const float* src = a.keys + (size_t)xseg * a.ss + (size_t)c * a.ld;
    const int sseg = (a.src_n_seg == 1) ? 0 : seg;
    const float* ssrt = a.src_sorted + ((size_t)sseg * a.C + c) * a.ns;
    float* o = a.out + (size_t)seg * a.oss + (size_t)c * a.ldo;
    const unsigned ns = (unsigned)a.ns;
    const int n = (int)a.n;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // pixel held in register r: 16-byte loads put 4 neighbouring pixels into one thread
    auto elem = [&](int r) { return VEC ? ((r >> 2) * NT + tid) * 4 + (r & 3) : r * NT + tid; };
    // register r holds a pixel of the column (VEC: n % 4 == 0, the four pixels of a 16-byte load stand or fall together);
    // a compare of tid with a scalar, so that neither 16 pixel numbers nor 16 lane masks have to stay live
    auto valid = [&](int r) { return VEC ? tid < (n >> 2) - (r >> 2) * NT : tid < n - r * NT; };

    SORT_PROBE(0);
    // ---- 0. the column
    uint32_t key[ITEMS];
    if (VEC) {
#pragma unroll
        for (int q = 0; q < ITEMS / 4; q++) {
            const int e0 = (q * NT + tid) * 4;
            const float4 v = *reinterpret_cast<const float4*>(src + (e0 < n ? e0 : 0));
            key[4 * q + 0] = f2key(v.x);
            key[4 * q + 1] = f2key(v.y);
            key[4 * q + 2] = f2key(v.z);
            key[4 * q + 3] = f2key(v.w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const int e = r * NT + tid;
            key[r] = f2key(src[e < n ? e : n - 1]);
        }
    }
    for (int i = tid; i < K::CNTW; i += NT) cnt[i] = 0u;
    if (tid < RK_COARSE) c1[tid] = 0u;
    if (tid < 32) misc[tid] = 0u;
    if (tid >= NT - R3_QWIN) slot[n + (tid - (NT - R3_QWIN))] = 0xffffffffu;  // larger than every finite key

    // ---- 1. min / max
    uint32_t klo = 0xffffffffu, khi = 0u;
    // (registers past the end of a short column hold a copy of a real key — clamped loads: harmless for min / max — and
    // stay out of every LDS update below through selects, not branches: per-register branches make the compiler keep the
    // register arrays as 16-wide tuples and spill them whole)
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        klo = key[r] < klo ? key[r] : klo;
        khi = key[r] > khi ? key[r] : khi;
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
    for (int k = 0; k < NW; k++) {
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
        for (int e = tid; e < n; e += NT) o[e] = ssrt[quantile_index((uint32_t)e, ns, (unsigned)n, a.inv_2nt)];
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
            const int left = n - r * NT;
            nsamp += (unsigned)(left < 0 ? 0 : (left > NT ? NT : left));
        }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r += RS) {
        if (valid(r)) {
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
        const uint4 cc = *reinterpret_cast<const uint4*>(c1 + 4 * lane);
        auto width = [&](unsigned cn) {
            const unsigned x = cn * (unsigned)NB;  // < 2^27: exact quotient via a float estimate + one correction
            unsigned q = (unsigned)((float)x / (float)nsamp);
            if (q * nsamp > x) q--;
            else if ((q + 1u) * nsamp <= x) q++;
            return 1u + q;
        };
        const unsigned w0 = width(cc.x), w1 = width(cc.y), w2 = width(cc.z), w3 = width(cc.w);
        const unsigned sum = w0 + w1 + w2 + w3;
        unsigned incl = sum;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const unsigned t = __shfl_up(incl, o2);
            if (lane >= o2) incl += t;
        }
        const unsigned b0 = incl - sum, b1 = b0 + w0, b2 = b1 + w1, b3 = b2 + w2;
        *reinterpret_cast<uint4*>(c1 + 4 * lane) = make_uint4(b0 | (w0 << 16), b1 | (w1 << 16), b2 | (w2 << 16), b3 | (w3 << 16));
    }
    __syncthreads();
    SORT_PROBE(3);
    // ---- 4. fine bucket b of every key; the returning count atomic gives the key's arrival number a inside the bucket.
    //         x -> b is monotone non-decreasing whatever the rounding: every step (subtract, scale, truncate, clamp) is.
    //         ba[r] = b | a << 13
    uint32_t ba[ITEMS];
    constexpr int G4 = ITEMS < 4 ? ITEMS : 4;  // LDS operations of G4 keys in flight together (the wave has no other ILP)
#pragma unroll
    for (int g = 0; g < ITEMS; g += G4) {
        float fr[G4];
        uint32_t bw[G4], bb[G4], old[G4];
#pragma unroll
        for (int j = 0; j < G4; j++) {
            if (g + j >= ITEMS) continue;  // compile-time: ITEMS need not be a multiple of four
            const float t = (key2f(key[g + j]) - lo) * s1;
            int bin = (int)t;
            bin = bin > RK_COARSE - 1 ? RK_COARSE - 1 : bin;
            fr[j] = t - (float)bin;
            bw[j] = c1[bin];
        }
#pragma unroll
        for (int j = 0; j < G4; j++) {
            if (g + j >= ITEMS) continue;
            const int wd = (int)(bw[j] >> 16);
            const float u = fr[j] * (float)wd;
            int sub = (int)u;
            sub = sub > wd - 1 ? wd - 1 : sub;
            bb[j] = (bw[j] & 0xffffu) + (uint32_t)sub;
        }
#pragma unroll
        for (int j = 0; j < G4; j++)
            if (g + j < ITEMS) old[j] = atomicAdd(&cnt[bb[j] >> 1], (valid(g + j) ? 1u : 0u) << ((bb[j] & 1u) << 4));
#pragma unroll
        for (int j = 0; j < G4; j++)
            if (g + j < ITEMS) ba[g + j] = bb[j] | (((old[j] >> ((bb[j] & 1u) << 4)) & 0xffffu) << R3_BBITS);
        // the (b, a) words packed here: otherwise the compiler keeps b and the atomic's result apart until step 6 and
        // spills both (64-VGPR budget)
#pragma unroll
        for (int j = 0; j < G4; j++)
            if (g + j < ITEMS) asm volatile("" : "+v"(ba[g + j]));
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    SORT_PROBE(4);
    // ---- 5. exclusive scan of the bucket counts -> start entries (in place, u16: start | big << 14 | long << 15)
    {
        uint32_t wv[PER];
        unsigned sum = 0;
        if (PER == 4) {  // one 16-byte read (the counters start on a 16-byte boundary)
            const uint4 c4 = tid * 4 < NW2 ? *reinterpret_cast<const uint4*>(cnt + tid * 4) : make_uint4(0u, 0u, 0u, 0u);
            wv[0] = c4.x; wv[1 % PER] = c4.y; wv[2 % PER] = c4.z; wv[3 % PER] = c4.w;
        } else {
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int i = tid * PER + q;
                wv[q] = i < NW2 ? cnt[i] : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < PER; q++) sum += (wv[q] & 0xffffu) + (wv[q] >> 16);
        // block-wide exclusive scan with ONE barrier: `red` is not written again before the barrier behind this step
        unsigned incl = sum;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const unsigned t = __shfl_up(incl, o2);
            if (lane >= o2) incl += t;
        }
        if (lane == 63) red[w] = incl;
        __syncthreads();
        unsigned ex = incl - sum;
#pragma unroll
        for (int k = 0; k < NW; k++) ex += k < w ? red[k] : 0u;
        auto entry = [&](unsigned s, unsigned cb) {
            uint32_t e = s & R3_SMASK;
            if ((s & 3u) + cb > (unsigned)R3_WIN) e |= R3_LONG;
            if (cb > (unsigned)RK_BIG) {
                e |= R3_BIGF | R3_LONG;
                const unsigned k = atomicAdd(&misc[0], 1u);
                if (k < RK_MAXBIG) { misc[2 + 2 * k] = s; misc[3 + 2 * k] = cb; }
            }
            return e;
        };
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int i = tid * PER + q;
            if (i < NW2) {
                const unsigned c0 = wv[q] & 0xffffu, c1v = wv[q] >> 16;
                const uint32_t e0 = entry(ex, c0), e1 = entry(ex + c0, c1v);
                cnt[i] = e0 | (e1 << 16);
                ex += c0 + c1v;
            }
        }
        if (tid == 0) cnt[NW2] = (uint32_t)n & R3_SMASK;  // the entry behind the last bucket
    }
    __syncthreads();
    const unsigned nbig = misc[0];
    if (nbig > RK_MAXBIG) {
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    SORT_PROBE(5);
    // ---- 6. every key takes its slot start[b] + a.  ba[r] becomes the bucket's start entry
#pragma unroll
    for (int g = 0; g < ITEMS; g += G4) {
        uint32_t e[G4];
#pragma unroll
        for (int j = 0; j < G4; j++)
            if (g + j < ITEMS) e[j] = st16[ba[g + j] & ((1u << R3_BBITS) - 1u)];
#pragma unroll
        for (int j = 0; j < G4; j++) {
            if (g + j >= ITEMS) continue;
            const uint32_t arr = ba[g + j] >> R3_BBITS;
            slot[valid(g + j) ? (e[j] & R3_SMASK) + arr : (uint32_t)(CAP + R3_QWIN)] = key[g + j];
            ba[g + j] = e[j];
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    // ---- 6b. oversized buckets only come from exact ties: if all keys of such a bucket are equal its ranks are the
    //          ranks of the pixel indices (bitmap + popcount prefix).  Anything else -> radix kernel.
    for (unsigned bi = 0; bi < nbig; bi++) {
        const uint32_t s = misc[2 + 2 * bi];
        const uint32_t k0 = slot[s];
        for (int i = tid; i < NWORDS; i += NT) bitmap[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            if (valid(r) && (ba[r] & (R3_DONE | R3_BIGF)) == R3_BIGF && (ba[r] & R3_SMASK) == (s & R3_SMASK)) {
                if (key[r] != k0) misc[1] = 1u;
                const uint32_t idx = (uint32_t)elem(r);
                atomicOr(&bitmap[idx >> 5], 1u << (idx & 31u));
            }
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
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const bool mine = valid(r) && (ba[r] & (R3_DONE | R3_BIGF)) == R3_BIGF && (ba[r] & R3_SMASK) == (s & R3_SMASK);
            const uint32_t idx = (uint32_t)(valid(r) ? elem(r) : 0);
            const uint32_t rk = s + bpre[idx >> 5] + (uint32_t)__popc(bitmap[idx >> 5] & ((1u << (idx & 31u)) - 1u));
            ba[r] = mine ? (R3_DONE | rk) : ba[r];
        }
        __syncthreads();
    }
    SORT_PROBE(6);
    // ---- 7. ranks, owner side: every slot before the bucket holds a smaller key, every slot behind it a larger one, so
    //         rank = W0 + #{slot[W0 .. W0 + 7] < key} for the aligned window W0 = start & ~3 whenever it covers the bucket.
    //         ba[r] becomes the rank (or R3_TAG | queue entry).
    auto rank_one = [&](int r, const uint4& x0, const uint4& x1) {
        const uint32_t e = ba[r], k = key[r];
        const uint32_t w0p = e & (R3_SMASK & ~3u);
        uint32_t lt = 0u, le = 0u;
        r3_window(lt, le, x0, x1, k);
        const bool done = (e & R3_DONE) != 0u;
        // a bucket wider than the window, or another slot with the same key (the pixels decide): the queue
        const bool queued = valid(r) && !done && ((e & R3_LONG) != 0u || le - lt > 1u);
        uint32_t qi = 0u;
        if (queued) {
            qi = atomicAdd(&misc[20], 1u);
            if (qi < (uint32_t)QCAP) {
                qkey[qi] = k;
                qwin[qi] = w0p;
                qpix[qi] = (uint32_t)elem(r);
            }
        }
        ba[r] = queued ? (R3_TAG | qi) : (done ? (e & R3_SMASK) : w0p + lt);
    };
#pragma unroll
    for (int g = 0; g + 1 < ITEMS; g += 2) {  // two windows in flight: keeps the unrolled loop inside the 64-VGPR budget
        const uint4* wa = reinterpret_cast<const uint4*>(slot + (ba[g] & (R3_SMASK & ~3u)));
        const uint4* wb = reinterpret_cast<const uint4*>(slot + (ba[g + 1] & (R3_SMASK & ~3u)));
        const uint4 a0 = wa[0], a1 = wa[1], b0 = wb[0], b1 = wb[1];
        rank_one(g, a0, a1);
        rank_one(g + 1, b0, b1);
        asm volatile("" ::: "memory");
    }
    if (ITEMS & 1) {
        const uint4* wa = reinterpret_cast<const uint4*>(slot + (ba[ITEMS - 1] & (R3_SMASK & ~3u)));
        const uint4 a0 = wa[0], a1 = wa[1];
        rank_one(ITEMS - 1, a0, a1);
    }
    asm volatile("" ::: "memory");
    SORT_PROBE(7);
    // the sorted source column on its way into registers while the queue is worked off (the key registers are dead)
    const bool stage = ns <= (unsigned)CAP;
    const bool svec = VEC && (ns % 4u == 0u) && ((reinterpret_cast<uintptr_t>(ssrt) & 15u) == 0u);
    float4 sv[VEC ? ITEMS / 4 : 1];
    if (VEC && stage && svec) {
#pragma unroll
        for (int q = 0; q < ITEMS / 4; q++) {
            const unsigned e0 = (unsigned)(q * NT + tid) * 4u;
            sv[q] = *reinterpret_cast<const float4*>(ssrt + (e0 < ns ? e0 : 0u));
        }
    }
    // ---- 8. queued keys, one per thread, against a 52-slot window (a bucket has at most RK_BIG keys here): buckets wider
    //         than the 8-slot window and keys with an equal partner.  Equal keys are all in the queue (each of them saw the
    //         other): their order is the order of their pixels, settled among the (few) entries of the tie list.
    __syncthreads();
    const uint32_t qn = misc[20];
    if (qn > (uint32_t)QCAP) {  // tie-heavy column: radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    for (uint32_t i = tid; i < qn; i += NT) {
        const uint32_t k = qkey[i], w0p = qwin[i];
        const uint4* wp = reinterpret_cast<const uint4*>(slot + w0p);
        uint32_t lt = 0u, le = 0u;
        static_assert(R3_QWIN == 52, "two trips of six 16-byte reads + one");
#pragma unroll 1
        for (int j = 0; j < 2; j++) {  // six reads in flight per trip
            const uint4 x0 = wp[6 * j], x1 = wp[6 * j + 1], x2 = wp[6 * j + 2], x3 = wp[6 * j + 3], x4 = wp[6 * j + 4],
                        x5 = wp[6 * j + 5];
            r3_window(lt, le, x0, x1, k);
            r3_window(lt, le, x2, x3, k);
            r3_window(lt, le, x4, x5, k);
        }
        {
            const uint4 x = wp[R3_QWIN / 4 - 1];
            r3_add_lt(lt, x.x, k); r3_add_lt(lt, x.y, k); r3_add_lt(lt, x.z, k); r3_add_lt(lt, x.w, k);
            r3_add_le(le, x.x, k); r3_add_le(le, x.y, k); r3_add_le(le, x.z, k); r3_add_le(le, x.w, k);
        }
        qres[i] = w0p + lt;
        if (le - lt > 1u) {  // a handful per column (exactly equal fp32 keys): compacted, so that nobody scans the whole queue
            const uint32_t t = atomicAdd(&misc[21], 1u);
            if (t < (uint32_t)TCAP) tlist[t] = i;
        }
    }
    __syncthreads();
    const uint32_t tn = misc[21];
    if (tn > (uint32_t)TCAP) {  // tie-heavy column: radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    for (uint32_t t = tid; t < tn; t += NT) {
        const uint32_t i = tlist[t], k = qkey[i], pix = qpix[i];
        uint32_t before = 0u;
        for (uint32_t u = 0; u < tn; u++) {
            const uint32_t j = tlist[u];
            before += (qkey[j] == k && qpix[j] < pix) ? 1u : 0u;
        }
        qres[i] += before;  // only this thread touches qres[i]
    }
    __syncthreads();
    SORT_PROBE(8);
    // ---- 9. out[pixel] = sorted_source[q(rank)]: the source column is staged in the slot array (every slot has been
    //         read), each owner picks its values and leaves with 16-byte stores
    float* val = reinterpret_cast<float*>(slot);
    if (stage) {
        if (VEC && svec) {
#pragma unroll
            for (int q = 0; q < ITEMS / 4; q++) {
                const unsigned e0 = (unsigned)(q * NT + tid) * 4u;
                if (e0 < ns) *reinterpret_cast<float4*>(val + e0) = sv[q];
            }
        } else {
            for (unsigned e = tid; e < ns; e += NT) val[e] = ssrt[e];
        }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const bool tagged = valid(r) && (ba[r] & R3_TAG) != 0u;
        const uint32_t got = qres[tagged ? (ba[r] & ~R3_TAG) : 0u];
        ba[r] = tagged ? got : ba[r];
    }
    __syncthreads();
    SORT_PROBE(9);
    float v[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const unsigned qi = quantile_index(valid(r) ? ba[r] : 0u, ns, (unsigned)n, a.inv_2nt);
        v[r] = stage ? val[qi] : ssrt[qi];
        if ((r & 3) == 3) asm volatile("" ::: "memory");
    }
    if (VEC && a.out_vec) {
#pragma unroll
        for (int q = 0; q < ITEMS / 4; q++) {
            const int e0 = (q * NT + tid) * 4;
            if (e0 < n) *reinterpret_cast<float4*>(o + e0) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; r++)
            if (valid(r)) o[elem(r)] = v[r];
    }
    SORT_PROBE(10);
}

template <int ITEMS, int NT>
static int launch_rank_match3_items(SortArgs a, int ncols, hipStream_t st) {
    const bool in_vec = ITEMS >= 4 && ITEMS % 4 == 0 && a.n % 4 == 0 && a.ld % 4 == 0 && a.ss % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(a.keys) & 15u) == 0;
    a.out_vec = (a.ldo % 4 == 0 && a.oss % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0) ? 1 : 0;
    const size_t lds = R3<ITEMS, NT>::LDS;
    hipError_t e;
    if (in_vec) {
        auto kern = rank_match3_kernel<ITEMS, (ITEMS >= 4 && ITEMS % 4 == 0), NT>;
        static DeviceOnce once;
        bool& attr = *once.slot();
        if (!attr) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error("sort: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return OPTEX_E_LAUNCH; }
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(ncols), dim3(NT), lds, st, a);
    } else {
        auto kern = rank_match3_kernel<ITEMS, false, NT>;
        static DeviceOnce once;
        bool& attr = *once.slot();
        if (!attr) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error("sort: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return OPTEX_E_LAUNCH; }
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(ncols), dim3(NT), lds, st, a);
    }
    return check_launch("rank_match3_kernel");
}

// 9 .. 16 keys per thread on NT threads
template <int NT>
static int launch_rank_match3_nt(int need, const SortArgs& a, int ncols, hipStream_t st) {
    switch (need) {
        case 9: return launch_rank_match3_items<9, NT>(a, ncols, st);
        case 10: return launch_rank_match3_items<10, NT>(a, ncols, st);
        case 11: return launch_rank_match3_items<11, NT>(a, ncols, st);
        case 12: return launch_rank_match3_items<12, NT>(a, ncols, st);
        case 13: return launch_rank_match3_items<13, NT>(a, ncols, st);
        case 14: return launch_rank_match3_items<14, NT>(a, ncols, st);
        case 15: return launch_rank_match3_items<15, NT>(a, ncols, st);
        default: return launch_rank_match3_items<16, NT>(a, ncols, st);
    }
}

// called by launch_sort_items<ITEMS, SORT_MATCH> (sort.hip) with flags cleared and the prof scope open.  Workgroup size and
// keys per thread are chosen here: 9 .. 16 keys per thread, as few as hold the column (registers past the end of a column
// still cost their instructions: a 12544-key column runs 13 keys on 1024 threads, not 16), on 256 / 512 / 1024 threads for
// columns up to 4096 / 8192 / 16384 keys (smaller workgroups = more columns resident per CU for the short ones).
int launch_rank_match3(int items, const SortArgs& a, int ncols, hipStream_t st) {
    (void)items;
    const long n = a.n;
    if (n <= 2048) return launch_rank_match3_items<2, SORT_NT>(a, ncols, st);
    if (n <= 4096) return launch_rank_match3_nt<256>((int)((n + 255) / 256), a, ncols, st);
    if (n <= 8192) return launch_rank_match3_nt<512>((int)((n + 511) / 512), a, ncols, st);
    return launch_rank_match3_nt<1024>((int)((n + 1023) / 1024), a, ncols, st);
}

}  // namespace optex
