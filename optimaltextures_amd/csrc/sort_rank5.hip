// sort_rank5.hip — rank_match5_kernel: the exact 1-D transport match (north-star addition; specification =
// oracle/optex_oracle.c orc_sort_match) with a ranking step built for FEWER LDS OPERATIONS PER KEY than rank_match4_kernel
// (sort_rank4.hip), whose 8-slot window costs every key a place write and four 8-byte reads at random addresses.
//
// Round 6: the bucket table is over-provisioned — FOUR fine buckets per key, 8-bit counters packed four to a
// 32-bit word — so that with the histogram-equalised map ~78 % of the keys are ALONE in their bucket.  For those keys
//      rank = group_start[bucket >> 2] + (sum of the counter bytes below bucket & 3 in the word)
// is final: no place write, no window read.  Only keys that share a bucket (~22 %) are written to the (then dead) counter
// array and compared with their bucket mates; exact ties go through a short list ordered by (totalOrder key, pixel).
//
// Per key on the LDS pipe: bucket-table read, returning count atomic, counter word, group start, source pick = 5 random-address
// operations on all lanes (rank_match4_kernel: table, atomic, start entry, place, 4 window reads, pick = 9) + place / mate reads
// on a fifth of the lanes + coalesced 16-byte passes (zero, scan, group starts, source staging).
//
// Layout: ONE persistent workgroup per CU (the counters alone are 4 bytes per key: 64 KiB at 16384 keys, + 32 KiB of group
// starts), 16 wavefronts at a 128-register budget; the next column's keys are requested as soon as this column's are dead
// (behind the mate compare) and arrive while the source is staged, picked and stored.  A workgroup walks the columns
// blockIdx.x, + gridDim.x, ...: with C = gridDim.x = 256 it sees the SAME channel of every texture, i.e. the same sorted source
// column every time (L2-resident).
//
// Everything the kernel cannot take is flagged for the radix sweep behind it (sort.hip), like rank_match4_kernel does:
// non-finite keys, a bucket of 256 or more keys (8-bit counter), more than R5_TCAP keys with an equal partner.
#include "sort_common.h"
#include <type_traits>

namespace optex {

constexpr int R5_TCAP = 256;               // keys with an equal partner, per column
constexpr uint32_t R5_TAG = 0x80000000u;   // rank register: index into the tie list, result pending
constexpr uint32_t R5_INF = 0x7f800000u;

typedef float r5_v4f __attribute__((ext_vector_type(4)));
typedef unsigned r5_v4u __attribute__((ext_vector_type(4)));
typedef unsigned r5_v2u __attribute__((ext_vector_type(2)));
typedef float r5_v2f __attribute__((ext_vector_type(2)));
#define R5_LDS(T, off) (*reinterpret_cast<__attribute__((address_space(3))) T*>((uint32_t)(off)))

template <int ITEMS, int NT>
struct R5 {
    static constexpr int CAP = ITEMS * NT;
    static constexpr int QR = (ITEMS + 3) / 4;        // 16-byte rows of counter words per thread
    static constexpr int NWRD = 4 * NT * QR;          // counter words, four 8-bit buckets each (>= CAP)
    static constexpr int NBK = 4 * NWRD;              // fine buckets
    static constexpr int WMIN = 4;                    // fine buckets every coarse bin gets whatever the sample says (a bin the quarter
                                                      // sample missed can hold a handful of keys: 0.44 % of the keys in buckets of four
                                                      // and more with 1, 0.23 % with 4 — the Poisson figure is 0.22 %)
    static constexpr int NBE = NBK - WMIN * RK_COARSE; // to distribute by the equalisation
    // byte offsets inside the workgroup's LDS (dynamic LDS starts at 0: checked once per workgroup).  Everything a DS
    // instruction addresses with a register + constant lies below 64 KiB + register, so the constant fits the offset field.
    static constexpr uint32_t MISC_B = 0;             // [32] words
    static constexpr uint32_t RED_B = 128;            // [64] scan partials (QR x wavefronts)
    static constexpr uint32_t C1_B = 384;             // [256] coarse histogram
    static constexpr uint32_t TAB_B = 1408;           // [257] (ww, base + 1/16) per coarse bin
    static constexpr uint32_t TIE_B = 3472;           // [3][R5_TCAP] key bits, pixel, rank
    static constexpr uint32_t GS_B = 8192;            // [NWRD] u16 group starts; from here on: the staged source column
    static constexpr uint32_t CW_B = GS_B + 2u * NWRD;  // [NWRD] counter words -> slots of the keys that share a bucket
    static constexpr size_t LDS = (size_t)CW_B + 4u * NWRD;
    static constexpr unsigned SRC_MAX = (6u * NWRD) / 4u;  // staged source values (over the group starts and the counters)
    static constexpr int SQ = (int)((SRC_MAX / 4u + NT - 1) / NT);  // 16-byte source loads per thread
    static_assert(NBK <= 65536, "bucket index is 16 bits");
    static_assert(QR * (NT / 64) <= 64, "one lane per (row, wavefront) partial in the scan");
    static_assert(TAB_B % 8 == 0 && TIE_B % 16 == 0 && TIE_B + 12u * R5_TCAP <= GS_B && TAB_B + 8u * (RK_COARSE + 1) <= TIE_B, "layout");
};

enum { R5_M_BAD = 0, R5_M_TN = 2, R5_M_HEAVY = 4 };

__device__ __forceinline__ unsigned r5_wave_incl_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// phase stamps of thread 0 (probe builds only: -DOPTEX_SORT_PROBE, scripts/sort5_probe.hip)
#ifdef OPTEX_SORT_PROBE
#define R5_STAMP(i) do { if (threadIdx.x == 0) a.probe[(size_t)col * 16 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define R5_STAMP(i) do { } while (0)
#endif

#ifdef R5_PERSISTENT_VARIANT  // the one-workgroup-per-CU experiment: probe builds only (scripts/sort5_probe.hip), not in the library
#ifndef R5_G
#define R5_G 8   // keys whose LDS operations are in flight together in the count / decode steps
#endif

// NT threads, ITEMS = ceil(n / NT) keys per thread: the first 4 * (ITEMS / 4) registers are 16-byte loads (4 neighbouring
// pixels), the rest scalar rows; only the last register row (or quad) can be ragged.  FULL: n == ITEMS * NT.
template <int ITEMS, int NT, bool FULL>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void rank_match5_kernel(SortArgs a) {
    using K = R5<ITEMS, NT>;
    constexpr int NW = NT / 64, QR = K::QR, CAP = K::CAP;
    constexpr uint32_t MISC_B = K::MISC_B, RED_B = K::RED_B, C1_B = K::C1_B, TAB_B = K::TAB_B, TIE_B = K::TIE_B, GS_B = K::GS_B,
                       CW_B = K::CW_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* misc = reinterpret_cast<uint32_t*>(smem + MISC_B);
    uint32_t* red = reinterpret_cast<uint32_t*>(smem + RED_B);
    uint32_t* c1 = reinterpret_cast<uint32_t*>(smem + C1_B);
    float2* tab = reinterpret_cast<float2*>(smem + TAB_B);
    uint32_t* tkey = reinterpret_cast<uint32_t*>(smem + TIE_B);
    uint32_t* tpix = tkey + R5_TCAP;
    uint32_t* tres = tpix + R5_TCAP;

    const int n = FULL ? CAP : (int)a.n;
    const unsigned ns = (unsigned)a.ns;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int Q = ITEMS / 4, T = ITEMS - 4 * Q;
    auto ragged = [](int r) { return !FULL && ((T == 0) ? r >= ITEMS - 4 : r == ITEMS - 1); };
    auto valid = [&](int r) { return !ragged(r) || ((T == 0) ? tid < (n >> 2) - (r >> 2) * NT : tid < n - r * NT); };

    // a thread index the compiler cannot tie to the other phases' (or the previous column's): otherwise it hoists every
    // address of every phase out of the column loop and keeps them — 60+ registers — alive (and spilled) for the whole kernel
    auto otid = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) {  // never on this toolchain
        for (int cc = blockIdx.x; cc < a.ncols; cc += (int)gridDim.x)
            if (threadIdx.x == 0) a.flags[cc] = 1;
        return;
    }

    // the column's keys (registers past the end hold a copy of a real key: they stay out of every LDS update)
    auto load_keys = [&](int cc, float (&xx)[ITEMS]) {
        const int sg = cc / a.C, ch = cc - sg * a.C;
        const float* src = a.keys + (size_t)((a.x_n_seg == 1) ? 0 : sg) * a.ss + (size_t)ch * a.ld;
        const int tid = otid();
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int e0 = (q * NT + tid) * 4;
            const float4 v = *reinterpret_cast<const float4*>(src + (ragged(4 * q) ? (e0 < n ? e0 : 0) : e0));
            xx[4 * q + 0] = v.x;
            xx[(4 * q + 1) % ITEMS] = v.y;
            xx[(4 * q + 2) % ITEMS] = v.z;
            xx[(4 * q + 3) % ITEMS] = v.w;
        }
#pragma unroll
        for (int r = 4 * Q; r < ITEMS; r++) {
            const int e = r * NT + tid;
            xx[r] = src[ragged(r) ? (e < n ? e : n - 1) : e];
        }
    };

    float x[ITEMS];
    int col = blockIdx.x;
    if (col < a.ncols) load_keys(col, x);

    for (; col < a.ncols; col += (int)gridDim.x) {
        bool fetched = false;  // the next column's keys are on their way into x
        const int nxt = col + (int)gridDim.x;
        do {
            const int seg = col / a.C, c = col - seg * a.C;
            const int sseg = (a.src_n_seg == 1) ? 0 : seg;
            const float* ssrt = a.src_sorted + ((size_t)sseg * a.C + c) * a.ns;
            float* o = a.out + (size_t)seg * a.oss + (size_t)c * a.ldo;
            const float lo = a.rng_lo[col], hi = a.rng_hi[col];
            R5_STAMP(0);

            // ---- 0. clear the counters (the previous column's staged source lies there), the coarse histogram, the flags
            {
                const int tz = otid();
#pragma unroll
                for (int j = 0; j < QR; j++) R5_LDS(r5_v4u, CW_B + (uint32_t)(j * NT + tz) * 16u) = r5_v4u{0u, 0u, 0u, 0u};
            }
            if (tid < RK_COARSE) c1[tid] = 0u;
            if (tid < 32) misc[tid] = 0u;

            if (!(hi < __uint_as_float(R5_INF)) || !(lo > -__uint_as_float(R5_INF))) {  // non-finite range: radix kernel
                if (tid == 0) a.flags[col] = 1;
                break;
            }
            if (lo == hi) {
                if (lo == 0.f) {  // zeros of both signs may be mixed (-0 < +0 in the specification): radix kernel
                    if (tid == 0) a.flags[col] = 1;
                    break;
                }
                // constant column: already sorted, rank = pixel index
                for (int e = tid; e < n; e += NT) o[e] = ssrt[quantile_index((uint32_t)e, ns, (unsigned)n, a.inv_2nt)];
                break;
            }
            const float s1 = __fdiv_rn((float)RK_COARSE, hi - lo);
            if (!(s1 > 0.f) || !(s1 < 1.0e37f)) {  // range over / underflow: radix kernel
                if (tid == 0) a.flags[col] = 1;
                break;
            }
            // non-finite keys inside a finite range cannot happen with an exact range, NaN can (min / max drop it): x * 0
            float nf = 0.f;
#pragma unroll
            for (int r = 0; r < ITEMS; r++) nf = __builtin_fmaf(x[r], 0.f, nf);
            __syncthreads();  // B0: cleared
            R5_STAMP(1);
            if (__any(!(nf == 0.f)) && lane == 0) misc[R5_M_BAD] = 1u;

            // ---- 1. coarse histogram of a spatially spread quarter sample
            constexpr int RS = 4;
            unsigned nsamp = 0;
            if (T == 0) {
                nsamp = (unsigned)(n + 3) / 4u;
            } else {
#pragma unroll
                for (int r = 0; r < ITEMS; r += RS) {
                    const int left = (r < 4 * Q) ? (n / 4 - (r >> 2) * NT) : (n - r * NT);  // rows: quads, then scalars
                    nsamp += (unsigned)(left < 0 ? 0 : (left > NT ? NT : left));
                }
            }
#pragma unroll
            for (int r = 0; r < ITEMS; r += RS) {
                if (valid(r)) {
                    const float t = (x[r] - lo) * s1;
                    int bin = (int)t;
                    bin = bin > RK_COARSE - 1 ? RK_COARSE - 1 : bin;
                    atomicAdd(&c1[bin], 1u);
                }
            }
            __syncthreads();  // B1
            R5_STAMP(2);
            // ---- 2. equalisation: coarse bin b gets w_b = WMIN + cnt_b * NBE / nsamp fine buckets from base_b on.  For
            //         t = (x - lo) * s1 in [b, b + 1):  u = fract(t) * (w_b - 1/8) + (base_b + 1/16)  lies in
            //         [base + 1/16, base + w - 1/16): both table values are exact in fp32 (17 + 4 bits), the fma rounds once by
            //         at most 2^-9, so int(u) stays inside the bin's bucket range — monotone over the column, no clamp.
            if (w == 0) {
                const uint4 cc = *reinterpret_cast<const uint4*>(c1 + 4 * lane);
                auto width = [&](unsigned cn) {
                    const unsigned xx = cn * (unsigned)K::NBE;  // < 2^28: exact quotient via a float estimate + one correction
                    unsigned q = (unsigned)((float)xx / (float)nsamp);
                    if (q * nsamp > xx) q--;
                    else if ((q + 1u) * nsamp <= xx) q++;
                    return (unsigned)K::WMIN + q;
                };
                const unsigned w0 = width(cc.x), w1 = width(cc.y), w2 = width(cc.z), w3 = width(cc.w);
                const unsigned sum = w0 + w1 + w2 + w3;
                const unsigned incl = r5_wave_incl_scan(sum);
                const unsigned b0 = incl - sum, b1 = b0 + w0, b2 = b1 + w1, b3 = b2 + w2;
                auto entry = [&](unsigned base, unsigned wd) { return make_float2((float)wd - 0.125f, (float)base + 0.0625f); };
                tab[4 * lane + 0] = entry(b0, w0);
                tab[4 * lane + 1] = entry(b1, w1);
                tab[4 * lane + 2] = entry(b2, w2);
                tab[4 * lane + 3] = entry(b3, w3);
                // t rounds up to RK_COARSE itself for x = hi: the last bucket of the last bin
                if (lane == 63) tab[RK_COARSE] = make_float2(0.f, (float)(b3 + w3 - 1u) + 0.0625f);
            }
            __syncthreads();  // B2
            R5_STAMP(3);
            if (misc[R5_M_BAD] != 0u) {
                if (tid == 0) a.flags[col] = 1;
                break;
            }
            // ---- 3. fine bucket b of every key; the returning count atomic on the bucket's byte of word b >> 2 gives the
            //         key's arrival number inside the bucket.  st[r] = b | arrival << 16
            uint32_t st[ITEMS];
            constexpr int G = ITEMS < R5_G ? ITEMS : R5_G;
#pragma unroll
            for (int g = 0; g < ITEMS; g += G) {
                float fr[G];
                r5_v2f e2[G];
                uint32_t b[G], old[G];
#pragma unroll
                for (int j = 0; j < G; j++) {
                    if (g + j >= ITEMS) continue;
                    const float t = (x[g + j] - lo) * s1;
                    fr[j] = __builtin_amdgcn_fractf(t);
                    e2[j] = R5_LDS(const r5_v2f, TAB_B + ((uint32_t)t << 3));
                }
#pragma unroll
                for (int j = 0; j < G; j++) {
                    if (g + j >= ITEMS) continue;
                    b[j] = (uint32_t)__builtin_fmaf(fr[j], e2[j].x, e2[j].y);
                    // 1 << 8 * (b & 3):  {0x01000000, 0x01000000} >> 8 * (~b & 3)
                    uint32_t inc = __builtin_amdgcn_alignbyte(0x01000000u, 0x01000000u, ~b[j]);
                    if (ragged(g + j)) inc = valid(g + j) ? inc : 0u;
                    old[j] = __hip_atomic_fetch_add(&R5_LDS(uint32_t, CW_B + (b[j] & ~3u)), inc, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int j = 0; j < G; j++) {
                    if (g + j >= ITEMS) continue;
                    const uint32_t arr = __builtin_amdgcn_alignbyte(0u, old[j], b[j]) & 255u;  // byte b & 3 of the old word
                    st[g + j] = b[j] | (arr << 16);
                }
                asm volatile("" ::: "memory");
            }
            __syncthreads();  // B3: all counts in
            R5_STAMP(4);
            // the sorted source column on its way into registers (staged in step 8)
            // (rank5_supported: ns % 4 == 0, 16-byte aligned columns, ns <= SRC_MAX — unconditional loads: a conditional one
            // makes every later use a phi of 4 * SQ registers)
            // The first SQE 16-byte rows (ns <= 4 * NT * SQE values: every call with ns <= n) now, the rest behind the mate compare,
            // when the registers are free again (at 16 keys per thread all six rows in flight here spill)
            constexpr int SQE = K::SQ < QR ? K::SQ : QR;
            r5_v4f sv[K::SQ];
            {
                const int ts = otid();
#pragma unroll
                for (int q = 0; q < SQE; q++) {
                    const unsigned e0 = (unsigned)(q * NT + ts) * 4u;
                    sv[q] = *reinterpret_cast<const r5_v4f*>(ssrt + (e0 < ns ? e0 : 0u));
                }
            }
            bool ovf = false;
            // ---- 4. exclusive scan of the counter bytes -> one 16-bit start per word (group of four buckets).  Thread t owns
            //         the 16-byte rows t, NT + t, ...: conflict-free reads; order = (row, wavefront, lane, word, byte)
            {
                r5_v4u cq[QR];
                unsigned p1[QR], p2[QR], p3[QR], tot[QR], incl[QR];
                const int tc = otid();
#pragma unroll
                for (int j = 0; j < QR; j++) cq[j] = R5_LDS(const r5_v4u, CW_B + (uint32_t)(j * NT + tc) * 16u);
#pragma unroll
                for (int j = 0; j < QR; j++) {
                    p1[j] = __builtin_amdgcn_sad_u8(cq[j].x, 0u, 0u);
                    p2[j] = __builtin_amdgcn_sad_u8(cq[j].y, 0u, p1[j]);
                    p3[j] = __builtin_amdgcn_sad_u8(cq[j].z, 0u, p2[j]);
                    tot[j] = __builtin_amdgcn_sad_u8(cq[j].w, 0u, p3[j]);
                    incl[j] = r5_wave_incl_scan(tot[j]);
                    if (lane == 63) red[j * NW + w] = incl[j];
                }
                __syncthreads();  // B4a
                R5_STAMP(5);
                // every wavefront scans the QR x NW partials itself (one lane each): no second barrier
                const unsigned pv = lane < QR * NW ? red[lane] : 0u;
                const unsigned pi = r5_wave_incl_scan(pv);
                // a bucket of 256 or more keys (massive ties) wraps its 8-bit counter: the carry adds 1 to the next byte (or is
                // lost) where 256 keys arrived, so the bytes no longer sum to n — the column goes to the radix kernel
                ovf = (unsigned)__builtin_amdgcn_readlane((int)pi, QR * NW - 1) != (unsigned)n;
#pragma unroll
                for (int j = 0; j < QR; j++) {
                    const int k = j * NW + w;  // wave-uniform
                    const unsigned base = k == 0 ? 0u : (unsigned)__builtin_amdgcn_readlane((int)pi, k - 1);
                    const unsigned ex = base + incl[j] - tot[j];
                    const r5_v2u gs = {ex | ((ex + p1[j]) << 16), (ex + p2[j]) | ((ex + p3[j]) << 16)};
                    R5_LDS(r5_v2u, GS_B + (uint32_t)(j * NT + tc) * 8u) = gs;
                }
            }
            __syncthreads();  // B4: group starts in place
            R5_STAMP(6);
            if (ovf) {  // uniform: every wavefront saw the same total
                if (tid == 0) a.flags[col] = 1;
                break;
            }
            // ---- 5. every key reads its counter word and its group start:
            //         start = group start + sum of the bytes below its own, cnt = its own byte.
            //         st[r] becomes  start | arrival << 16 | cnt << 24  (the bucket index is not needed again)
#pragma unroll
            for (int g = 0; g < ITEMS; g += G) {
                uint32_t cw[G], gs[G];
#pragma unroll
                for (int j = 0; j < G; j++) {
                    if (g + j >= ITEMS) continue;
                    cw[j] = R5_LDS(const uint32_t, CW_B + (st[g + j] & 0xfffcu));
                    gs[j] = R5_LDS(const unsigned short, GS_B + ((st[g + j] >> 1) & 0x7ffeu));
                }
#pragma unroll
                for (int j = 0; j < G; j++) {
                    if (g + j >= ITEMS) continue;
                    const uint32_t below = __builtin_amdgcn_alignbyte(cw[j], 0u, st[g + j]);   // bytes 0 .. (b & 3) - 1 of cw, moved up
                    const uint32_t start = __builtin_amdgcn_sad_u8(below, 0u, gs[j]);
                    uint32_t cnt = __builtin_amdgcn_alignbyte(0u, cw[j], st[g + j]) << 24;     // byte b & 3 of cw, on top
                    if (ragged(g + j)) cnt = valid(g + j) ? cnt : (1u << 24);
                    st[g + j] = ((st[g + j] & 0x00ff0000u) | start) | cnt;
                }
                asm volatile("" ::: "memory");
            }
            __syncthreads();  // B5: the counters are dead
            R5_STAMP(7);
            // ---- 6. keys that share a bucket take slot start + arrival in the counter array
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                if (st[r] >= (2u << 24)) {  // cnt >= 2
                    const uint32_t pos = (st[r] & 0xffffu) + ((st[r] >> 16) & 255u);
                    R5_LDS(float, CW_B + (pos << 2)) = x[r];
                }
            }
            __syncthreads();  // B6
            R5_STAMP(8);
            // ---- 7. ... and count their smaller bucket mates.  Two mates are read blindly (all keys' reads in flight
            //         together); buckets of four and more keys loop.  The low half of st[r] becomes the rank (in place: a rank
            //         stays below 16384), or st[r] = R5_TAG | tie-list entry
            uint32_t tie = 0u;  // bit r: key r has an equal bucket mate
            float m1[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                m1[r] = __uint_as_float(R5_INF);
                if (st[r] >= (2u << 24)) {
                    const uint32_t j1 = (st[r] & 0x00ff0000u) == 0u ? 1u : 0u;  // first mate: slot 0, or slot 1 for arrival 0
                    m1[r] = R5_LDS(const float, CW_B + (((st[r] & 0xffffu) + j1) << 2));
                }
            }
            // buckets of four and more keys (0.2 % of the keys, one wavefront row in seven): the whole bucket, four
            // slots in flight at a time — then out of the way (cnt := 1)
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                if (st[r] >= (4u << 24)) {
                    const uint32_t start = st[r] & 0xffffu, cnt = st[r] >> 24;
                    uint32_t lt = 0u, eq = 0u;
#pragma unroll
                    for (int j0 = 0; j0 < 8; j0 += 4) {
                        float mm[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            mm[j] = __uint_as_float(R5_INF);
                            if ((uint32_t)(j0 + j) < cnt) mm[j] = R5_LDS(const float, CW_B + ((start + (uint32_t)(j0 + j)) << 2));
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            lt += (mm[j] < x[r]) ? 1u : 0u;
                            eq += (mm[j] == x[r]) ? 1u : 0u;
                        }
                        if (j0 == 0 && !__any(cnt > 4u)) break;
                    }
                    for (uint32_t j = 8; j < cnt; j++) {  // (only ties fill a bucket like this)
                        const float m = R5_LDS(const float, CW_B + ((start + j) << 2));
                        lt += (m < x[r]) ? 1u : 0u;
                        eq += (m == x[r]) ? 1u : 0u;
                    }
                    st[r] = (start + lt) | (1u << 24);
                    tie |= eq > 1u ? (1u << r) : 0u;
                    m1[r] = __uint_as_float(R5_INF);
                }
            }
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const uint32_t start = st[r] & 0xffffu;
                const float mm = m1[r];
                m1[r] = __uint_as_float(R5_INF);
                if (st[r] >= (3u << 24)) {  // cnt == 3: the second mate is slot 2, or slot 1 for arrival 2
                    const uint32_t j2 = (st[r] & 0x00fe0000u) != 0u ? 1u : 2u;
                    m1[r] = R5_LDS(const float, CW_B + ((start + j2) << 2));
                }
                st[r] += (mm < x[r]) ? 1u : 0u;
                tie |= (mm == x[r]) ? (1u << r) : 0u;
            }
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                st[r] += (m1[r] < x[r]) ? 1u : 0u;
                tie |= (m1[r] == x[r]) ? (1u << r) : 0u;
            }
            if (tie != 0u) {  // rare: exact ties (and -0 / +0) are ordered by (totalOrder key, pixel) among themselves
                const int tid = otid();  // (elem(r) below: not sixteen pixel numbers kept alive for the whole kernel)
                auto elem = [&](int r) { return r < 4 * Q ? ((r >> 2) * NT + tid) * 4 + (r & 3) : r * NT + tid; };
#pragma unroll
                for (int r = 0; r < ITEMS; r++) {
                    if ((tie >> r) & 1u) {
                        const uint32_t ti = atomicAdd(&misc[R5_M_TN], 1u);
                        if (ti < (uint32_t)R5_TCAP) {
                            tkey[ti] = __float_as_uint(x[r]);
                            tpix[ti] = (uint32_t)elem(r);
                            tres[ti] = st[r] & 0xffffu;
                        }
                        st[r] = R5_TAG | ti;
                    }
                }
            }
            {
                const int ts = otid();
#pragma unroll
                for (int q = SQE; q < K::SQ; q++) {
                    const unsigned e0 = (unsigned)(q * NT + ts) * 4u;
                    sv[q] = *reinterpret_cast<const r5_v4f*>(ssrt + (e0 < ns ? e0 : 0u));
                }
            }
            // the keys are dead: the next column's are requested now and arrive while this one is staged, picked and stored
            if (nxt < a.ncols) load_keys(nxt, x);
            fetched = true;
            __syncthreads();  // B7: every slot has been read, the tie list is complete
            R5_STAMP(9);
            const uint32_t tn = misc[R5_M_TN];
            if (tn > (uint32_t)R5_TCAP) {  // tie-heavy column: radix kernel
                if (tid == 0) a.flags[col] = 1;
                break;
            }
            for (uint32_t t = tid; t < tn; t += NT) {
                const uint32_t kb = tkey[t], pix = tpix[t];
                const float kf = __uint_as_float(kb);
                const uint32_t kk = f2key(kf);
                uint32_t before = 0u;
                for (uint32_t u = 0; u < tn; u++) {
                    const float jf = __uint_as_float(tkey[u]);
                    const uint32_t jk = f2key(jf);
                    // float-equal partners (this includes -0 / +0) ordered by (totalOrder key, pixel); rk counted the
                    // float-smaller mates only
                    before += (jf == kf && (jk < kk || (jk == kk && tpix[u] < pix))) ? 1u : 0u;
                }
                tres[t] += before;  // only this thread touches tres[t]
            }
            // ---- 8. the sorted source column staged over the (dead) group starts and counters
            {
                const int tg = otid();
#pragma unroll
                for (int q = 0; q < K::SQ; q++) {
                    const unsigned e0 = (unsigned)(q * NT + tg) * 4u;
                    if (e0 < ns) R5_LDS(r5_v4f, GS_B + (e0 << 2)) = sv[q];
                }
            }
            __syncthreads();  // B8
            R5_STAMP(10);
            if (tn != 0u) {
#pragma unroll
                for (int r = 0; r < ITEMS; r++)
                    if ((st[r] & R5_TAG) != 0u) st[r] = tres[st[r] & ~R5_TAG];
            }
            // ---- 9. out[pixel] = sorted_source[q(rank)]
            float v[ITEMS];
            auto pick = [&](auto same) {
#pragma unroll
                for (int r = 0; r < ITEMS; r++) {
                    const unsigned rr = ragged(r) ? (valid(r) ? (st[r] & 0xffffu) : 0u) : (st[r] & 0xffffu);
                    unsigned qi = rr;
                    if (!decltype(same)::value) {
                        const double aa = (double)(2u * rr + 1u) * (double)ns;
                        qi = (unsigned)__builtin_fma(aa, a.inv_2nt, 7.450580596923828e-09);  // quantile_index (sort_common.h)
                    }
                    v[r] = R5_LDS(const float, GS_B + (qi << 2));
                }
            };
            if (ns == (unsigned)n) pick(std::true_type{});
            else pick(std::false_type{});
            const int to = otid();
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int e0 = (q * NT + to) * 4;
                if (!ragged(4 * q) || e0 < n)
                    *reinterpret_cast<float4*>(o + e0) =
                        make_float4(v[4 * q], v[(4 * q + 1) % ITEMS], v[(4 * q + 2) % ITEMS], v[(4 * q + 3) % ITEMS]);
            }
#pragma unroll
            for (int r = 4 * Q; r < ITEMS; r++)
                if (valid(r)) o[r * NT + to] = v[r];
            R5_STAMP(11);
        } while (0);
        if (!fetched && nxt < a.ncols) load_keys(nxt, x);
        __syncthreads();  // B9: the staged source has been read; the next column clears it
        R5_STAMP(12);
    }
}

int device_cu_count();

// ---------------------------------------------------------------------------------------------------------------------------
#endif  // R5_PERSISTENT_VARIANT

#ifndef R5W_H16
#define R5W_H16 4   // mates read together at 14 .. 16 keys per thread (8 below)
#endif
#ifndef R5W_BRANCHFREE
#define R5W_BRANCHFREE 0   // 1: place / mate steps without exec-mask branches (selects + a +inf word): measured SLOWER (profiles/r06_sort_experiments.md)
#endif
#ifndef R5W_PACKED_T
#define R5W_PACKED_T 0   // 1: count step with (x - lo) * s1 of two keys per v_pk_add_f32 / v_pk_mul_f32: measured SLOWER (0.541 against 0.548 weighted)
#endif
#ifndef R5W_G
#define R5W_G 4     // keys whose LDS operations are in flight together in the count and decode steps
#endif
// rank_match5w_kernel: the same ranking step at a 64-register budget, ONE column per workgroup, TWO (or more) workgroups per CU
// like rank_match4_kernel — the counters shrink to three buckets per key (NWRD = 12 * NT words at 13 .. 16 keys per thread:
// 8 KiB + 72 KiB = exactly half the LDS), so that while one workgroup waits at a barrier or for its column the other one issues.
// (profiles/r06_sort_experiments.md: the one-workgroup-per-CU kernel above has 30 % fewer LDS cycles and 15 % fewer VALU
// instructions than rank_match4_kernel and is still slower — 16 wavefronts per CU leave the VALU idle 40 % of the time.)
template <int ITEMS, int NT>
struct R5W {
    static constexpr int CAP = ITEMS * NT;
    static constexpr int QR = (ITEMS * 3 + 15) / 16;  // 16-byte rows of counter words per thread: ~3 words = 12 buckets per 4 keys
    static constexpr int NWRD = 4 * NT * QR;
    static constexpr int NBK = 4 * NWRD;
    static constexpr int WMIN = 4;
    static constexpr int NBE = NBK - WMIN * RK_COARSE;
    static constexpr uint32_t MISC_B = 0, RED_B = 128, C1_B = 384, TAB_B = 1408, TIE_B = 3472, GS_B = 8192;
    static constexpr uint32_t CW_B = GS_B + 2u * NWRD;
    static constexpr size_t LDS = (size_t)CW_B + 4u * NWRD;
    static constexpr int SQ = (6 * NWRD / 16 + NT - 1) / NT < 4 ? (6 * NWRD / 16 + NT - 1) / NT : 4;  // 16-byte source loads per thread
    static constexpr unsigned SRC_MAX = (unsigned)(6 * NWRD / 4) < (unsigned)(16 * NT * SQ / 4) ? (unsigned)(6 * NWRD / 4) : (unsigned)(4 * NT * SQ);
    static_assert(NBK <= 65536 && QR * (NT / 64) <= 64 && 6 * QR >= ITEMS, "layout: 6 * NWRD bytes hold one 4-byte slot per key");
};

template <int ITEMS, int NT, bool FULL, bool EMIT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void rank_match5w_kernel(SortArgs a) {
    using K = R5W<ITEMS, NT>;
    constexpr int NW = NT / 64, QR = K::QR, CAP = K::CAP;
    constexpr uint32_t MISC_B = K::MISC_B, RED_B = K::RED_B, C1_B = K::C1_B, TAB_B = K::TAB_B, TIE_B = K::TIE_B, GS_B = K::GS_B,
                       CW_B = K::CW_B;
    // a key that shares its bucket sits in slot start + arrival < n: the slots need n words, the counters have only NWRD >= 2 n / 3 —
    // they lie over the group starts AND the counters (6 * NWRD bytes, both dead behind the decode step's barrier)
    constexpr uint32_t SLOT_B = GS_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* misc = reinterpret_cast<uint32_t*>(smem + MISC_B);
    uint32_t* red = reinterpret_cast<uint32_t*>(smem + RED_B);
    uint32_t* c1 = reinterpret_cast<uint32_t*>(smem + C1_B);
    float2* tab = reinterpret_cast<float2*>(smem + TAB_B);
    uint32_t* tkey = reinterpret_cast<uint32_t*>(smem + TIE_B);
    uint32_t* tpix = tkey + R5_TCAP;
    uint32_t* tres = tpix + R5_TCAP;

    const int n = FULL ? CAP : (int)a.n;
    const unsigned ns = (unsigned)a.ns;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int Q = ITEMS / 4, T = ITEMS - 4 * Q;
    auto ragged = [](int r) { return !FULL && ((T == 0) ? r >= ITEMS - 4 : r == ITEMS - 1); };
    auto valid = [&](int r) { return !ragged(r) || ((T == 0) ? tid < (n >> 2) - (r >> 2) * NT : tid < n - r * NT); };
    auto otid = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    const int col = blockIdx.x;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) {  // never on this toolchain
        if (threadIdx.x == 0) a.flags[col] = 1;
        return;
    }
    const int seg = col / a.C, c = col - seg * a.C;
    const float* src = a.keys + (size_t)((a.x_n_seg == 1) ? 0 : seg) * a.ss + (size_t)c * a.ld;
    // the match (a.out: out[pixel] = sorted source order statistic of its rank) or the sort itself (optex_sort_columns: keys and /
    // or pixel indices by rank, contiguous [column, n]) — the ranking is the same; a template parameter all the same: with both
    // epilogues in one kernel the keys stay alive to the end and the 16-key instantiation of the MATCH spills in its mate step
    constexpr bool emit = EMIT;
    const float* ssrt = emit ? nullptr : a.src_sorted + ((size_t)((a.src_n_seg == 1) ? 0 : seg) * a.C + c) * a.ns;
    float* o = emit ? nullptr : a.out + (size_t)seg * a.oss + (size_t)c * a.ldo;

    float x[ITEMS];
    {
        const int tl = otid();
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int e0 = (q * NT + tl) * 4;
            const float4 v = *reinterpret_cast<const float4*>(src + (ragged(4 * q) ? (e0 < n ? e0 : 0) : e0));
            x[4 * q + 0] = v.x;
            x[(4 * q + 1) % ITEMS] = v.y;
            x[(4 * q + 2) % ITEMS] = v.z;
            x[(4 * q + 3) % ITEMS] = v.w;
        }
#pragma unroll
        for (int r = 4 * Q; r < ITEMS; r++) {
            const int e = r * NT + tl;
            x[r] = src[ragged(r) ? (e < n ? e : n - 1) : e];
        }
    }
    // The column's range: the caller's (optex_ot_loop: the rotation GEMM's epilogue took min / max of exactly these values — the
    // barrier below is then passed while the column is still on its way) or the kernel's own reduction (v_min / v_max drop NaN:
    // a wavefront that sees a non-finite key reports hi = +inf and the column goes to the radix kernel).
    const bool own_range = a.rng_lo == nullptr;   // uniform
    float lo = 0.f, hi = 0.f;
    if (!own_range) {
        lo = a.rng_lo[col];
        hi = a.rng_hi[col];
    } else {
        // per thread: v_min3 / v_max3 over the keys; NaN (dropped by both) through the sum of x * 0; equal neighbours (tie-heavy columns
        // — ReLU features before any rotation: half the keys are exactly 0 — overflow this kernel's 8-bit counters: the sooner they
        // leave for rank_match4_kernel, which ranks an all-equal bucket by pixel index, the better)
        float nf = 0.f;
        uint32_t eqp = 0u;
        lo = hi = x[0];
#pragma unroll
        for (int r = 1; r + 1 < ITEMS; r += 2) {
            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(lo) : "v"(x[r]), "v"(x[r + 1]));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(hi) : "v"(x[r]), "v"(x[r + 1]));
        }
        if (ITEMS % 2 == 0) {
            asm("v_min_f32 %0, %0, %1" : "+v"(lo) : "v"(x[ITEMS - 1]));
            asm("v_max_f32 %0, %0, %1" : "+v"(hi) : "v"(x[ITEMS - 1]));
        }
#pragma unroll
        for (int r = 0; r < ITEMS; r++) nf = __builtin_fmaf(x[r], 0.f, nf);
#pragma unroll
        for (int r = 0; r + 1 < ITEMS; r += 2) eqp |= (x[r] == x[r + 1]) ? 1u : 0u;
        if (__any(!(nf == 0.f))) hi = __uint_as_float(R5_INF);
        const uint32_t heavy = __popcll(__ballot(eqp != 0u)) >= 16 ? 1u : 0u;
        // wavefront: DPP row shifts, then the two row broadcasts: lane 63 holds the result
#define R5_DPP_STEP(ctrl, rmask)                                                                                          \
        {                                                                                                                 \
            const float pl = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lo), __float_as_int(lo), ctrl, rmask, 0xf, false)); \
            const float ph = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(hi), __float_as_int(hi), ctrl, rmask, 0xf, false)); \
            asm("v_min_f32 %0, %0, %1" : "+v"(lo) : "v"(pl));                                                         \
            asm("v_max_f32 %0, %0, %1" : "+v"(hi) : "v"(ph));                                                         \
        }
        R5_DPP_STEP(0x111, 0xf) R5_DPP_STEP(0x112, 0xf) R5_DPP_STEP(0x114, 0xf) R5_DPP_STEP(0x118, 0xf)
        R5_DPP_STEP(0x142, 0xa) R5_DPP_STEP(0x143, 0xc)
#undef R5_DPP_STEP
        if (lane == 63) {
            red[w] = __float_as_uint(lo);
            red[16 + w] = __float_as_uint(hi);
            red[32 + w] = heavy;
        }
    }
    // ---- 0. clear the counters, the coarse histogram, the flags
    {
        const int tz = otid();
#pragma unroll
        for (int j = 0; j < QR; j++) R5_LDS(r5_v4u, CW_B + (uint32_t)(j * NT + tz) * 16u) = r5_v4u{0u, 0u, 0u, 0u};
    }
    if (tid < RK_COARSE) c1[tid] = 0u;
    if (tid < 32) misc[tid] = tid >= 28 ? R5_INF : 0u;  // words 28 .. 31: +inf, what a key without that mate reads
    if (own_range) {
        __syncthreads();   // (the partial ranges; the cleared counters ride along: B0 below is then passed at once)
        uint32_t nheavy = 0u;
        lo = __uint_as_float(red[0]);
        hi = __uint_as_float(red[16]);
#pragma unroll
        for (int kk = 0; kk < NW; kk += 4) {
            const r5_v4u pl = R5_LDS(const r5_v4u, RED_B + kk * 4), ph = R5_LDS(const r5_v4u, RED_B + 64 + kk * 4),
                         pe = R5_LDS(const r5_v4u, RED_B + 128 + kk * 4);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                if (kk + j + 1 < NW) {
                    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(lo) : "v"(pl[j]), "v"(pl[j + 1]));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(hi) : "v"(ph[j]), "v"(ph[j + 1]));
                    nheavy += pe[j] + pe[j + 1];
                } else if (kk + j < NW) {
                    asm("v_min_f32 %0, %0, %1" : "+v"(lo) : "v"(pl[j]));
                    asm("v_max_f32 %0, %0, %1" : "+v"(hi) : "v"(ph[j]));
                    nheavy += pe[j];
                }
            }
        }
        lo = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(lo)));   // (the builtin takes an int: the bits, not the value)
        hi = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(hi)));
        if (2u * nheavy >= (uint32_t)NW) {  // tie-heavy: rank_match4_kernel's (it ranks an all-equal bucket by pixel index)
            if (tid == 0) a.flags[col] = 1;
            return;
        }
    }
    if (!(hi < __uint_as_float(R5_INF)) || !(lo > -__uint_as_float(R5_INF))) {  // non-finite range: radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    if (lo == hi) {
        if (lo == 0.f) {  // zeros of both signs may be mixed (-0 < +0 in the specification): radix kernel
            if (tid == 0) a.flags[col] = 1;
            return;
        }
        // constant column: already sorted, rank = pixel index
        if (emit) {
            for (int e = tid; e < n; e += NT) {
                if (a.out_keys) a.out_keys[(size_t)col * n + e] = lo;
                if (a.out_idx) a.out_idx[(size_t)col * n + e] = (uint32_t)e;
            }
        } else {
            for (int e = tid; e < n; e += NT) o[e] = ssrt[quantile_index((uint32_t)e, ns, (unsigned)n, a.inv_2nt)];
        }
        if (a.only_flagged && tid == 0) a.flags[col] = 0;
        return;
    }
    const float s1 = __fdiv_rn((float)RK_COARSE, hi - lo);
    if (!(s1 > 0.f) || !(s1 < 1.0e37f)) {  // range over / underflow: radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    __syncthreads();  // B0: cleared (passed while the column is still on its way)
    if (!own_range) {
        float nf = 0.f;
#pragma unroll
        for (int r = 0; r < ITEMS; r++) nf = __builtin_fmaf(x[r], 0.f, nf);
        if (__any(!(nf == 0.f)) && lane == 0) misc[R5_M_BAD] = 1u;
    }
    // ---- 1. coarse histogram of a spatially spread quarter sample
    constexpr int RS = 4;
    unsigned nsamp = 0;
    if (T == 0) {
        nsamp = (unsigned)(n + 3) / 4u;
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; r += RS) {
            const int left = (r < 4 * Q) ? (n / 4 - (r >> 2) * NT) : (n - r * NT);
            nsamp += (unsigned)(left < 0 ? 0 : (left > NT ? NT : left));
        }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r += RS) {
        if (valid(r)) {
            const float t = (x[r] - lo) * s1;
            int bin = (int)t;
            bin = bin > RK_COARSE - 1 ? RK_COARSE - 1 : bin;
            atomicAdd(&c1[bin], 1u);
        }
    }
    __syncthreads();  // B1
    // ---- 2. equalisation (see rank_match5_kernel)
    if (w == 0) {
        const uint4 cc = *reinterpret_cast<const uint4*>(c1 + 4 * lane);
        auto width = [&](unsigned cn) {
            const unsigned xx = cn * (unsigned)K::NBE;
            unsigned q = (unsigned)((float)xx / (float)nsamp);
            if (q * nsamp > xx) q--;
            else if ((q + 1u) * nsamp <= xx) q++;
            return (unsigned)K::WMIN + q;
        };
        const unsigned w0 = width(cc.x), w1 = width(cc.y), w2 = width(cc.z), w3 = width(cc.w);
        const unsigned sum = w0 + w1 + w2 + w3;
        const unsigned incl = r5_wave_incl_scan(sum);
        const unsigned b0 = incl - sum, b1 = b0 + w0, b2 = b1 + w1, b3 = b2 + w2;
        auto entry = [&](unsigned base, unsigned wd) { return make_float2((float)wd - 0.125f, (float)base + 0.0625f); };
        tab[4 * lane + 0] = entry(b0, w0);
        tab[4 * lane + 1] = entry(b1, w1);
        tab[4 * lane + 2] = entry(b2, w2);
        tab[4 * lane + 3] = entry(b3, w3);
        if (lane == 63) tab[RK_COARSE] = make_float2(0.f, (float)(b3 + w3 - 1u) + 0.0625f);
    }
    __syncthreads();  // B2
    if (misc[R5_M_BAD] != 0u) {
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    // ---- 3. fine bucket + returning count atomic.  st[r] = b | arrival << 16
    uint32_t st[ITEMS];
    constexpr int G = ITEMS < R5W_G ? ITEMS : R5W_G;
#pragma unroll
    for (int g = 0; g < ITEMS; g += G) {
        float fr[G];
        r5_v2f e2[G];
        uint32_t b[G], old[G];
#if R5W_PACKED_T
        // t = (x - lo) * s1 for two keys per instruction (v_pk_add_f32, v_pk_mul_f32: the same IEEE operations, same bits)
        float tj[G];
#pragma unroll
        for (int j = 0; j < G; j += 2) {
            if (g + j + 1 < ITEMS) {
                r5_v2f xx = {x[g + j], x[g + j + 1]};
                asm volatile("" : "+v"(xx));  // (keeps the differences x - lo from being formed ahead, next to the sample's)
                x[g + j] = xx.x;
                x[g + j + 1] = xx.y;
                const r5_v2f t2 = (xx - r5_v2f{lo, lo}) * r5_v2f{s1, s1};
                tj[j] = t2.x;
                tj[j + 1] = t2.y;
            } else if (g + j < ITEMS) {
                asm volatile("" : "+v"(x[g + j]));
                tj[j] = (x[g + j] - lo) * s1;
            }
        }
#endif
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (g + j >= ITEMS) continue;
#if R5W_PACKED_T
            const float t = tj[j];
#else
            asm volatile("" : "+v"(x[g + j]));  // (keeps the differences x - lo from being formed ahead, next to the sample's)
            const float t = (x[g + j] - lo) * s1;
#endif
            fr[j] = __builtin_amdgcn_fractf(t);
            e2[j] = R5_LDS(const r5_v2f, TAB_B + ((uint32_t)t << 3));
        }
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (g + j >= ITEMS) continue;
            b[j] = (uint32_t)__builtin_fmaf(fr[j], e2[j].x, e2[j].y);
            uint32_t inc = __builtin_amdgcn_alignbyte(0x01000000u, 0x01000000u, ~b[j]);
            if (ragged(g + j)) inc = valid(g + j) ? inc : 0u;
            old[j] = __hip_atomic_fetch_add(&R5_LDS(uint32_t, CW_B + (b[j] & ~3u)), inc, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (g + j >= ITEMS) continue;
            // byte b & 3 of the old word in bits 16 .. 23; what alignbyte leaves above it goes to bits 24 .. 31, which step 5 overwrites
            st[g + j] = b[j] | (__builtin_amdgcn_alignbyte(0u, old[j], b[j]) << 16);
        }
#pragma unroll
        for (int j = 0; j < G; j++)
            if (g + j < ITEMS) asm volatile("" : "+v"(st[g + j]));
        asm volatile("" ::: "memory");
    }
    __syncthreads();  // B3
    // ---- 4. scan of the counter bytes -> 16-bit group starts
    bool ovf = false;
    {
        r5_v4u cq[QR];
        unsigned p1[QR], p2[QR], p3[QR], tot[QR], incl[QR];
        const int tc = otid();
#pragma unroll
        for (int j = 0; j < QR; j++) cq[j] = R5_LDS(const r5_v4u, CW_B + (uint32_t)(j * NT + tc) * 16u);
#pragma unroll
        for (int j = 0; j < QR; j++) {
            p1[j] = __builtin_amdgcn_sad_u8(cq[j].x, 0u, 0u);
            p2[j] = __builtin_amdgcn_sad_u8(cq[j].y, 0u, p1[j]);
            p3[j] = __builtin_amdgcn_sad_u8(cq[j].z, 0u, p2[j]);
            tot[j] = __builtin_amdgcn_sad_u8(cq[j].w, 0u, p3[j]);
            incl[j] = r5_wave_incl_scan(tot[j]);
            if (lane == 63) red[j * NW + w] = incl[j];
        }
        if (own_range) {
            // Not the hot loop's rotated pastiche: a quantised column (a few hundred distinct values, dozens of keys in each of as many
            // buckets — no 8-bit counter overflows) would walk its buckets key by key in step 7 (5.4 ms instead of 0.85 for [64 x 256]
            // columns of 16384 keys) only to overflow the tie list at the end.  Threads that see a bucket of >= 16 keys are counted:
            // 32 of them and the column goes to rank_match4_kernel now.  (One flat patch of <= 255 equal keys is one such thread.)
            uint32_t m = 0u;
#pragma unroll
            for (int j = 0; j < QR; j++) m |= cq[j].x | cq[j].y | cq[j].z | cq[j].w;
            const unsigned hv = (unsigned)__popcll(__ballot((m & 0xf0f0f0f0u) != 0u));
            if (lane == 0 && hv != 0u) atomicAdd(&misc[R5_M_HEAVY], hv);
        }
        __syncthreads();  // B4a
        const unsigned pv = lane < QR * NW ? red[lane] : 0u;
        const unsigned pi = r5_wave_incl_scan(pv);
        ovf = (unsigned)__builtin_amdgcn_readlane((int)pi, QR * NW - 1) != (unsigned)n;
#pragma unroll
        for (int j = 0; j < QR; j++) {
            const int k = j * NW + w;
            const unsigned base = k == 0 ? 0u : (unsigned)__builtin_amdgcn_readlane((int)pi, k - 1);
            const unsigned ex = base + incl[j] - tot[j];
            const r5_v2u gs = {ex | ((ex + p1[j]) << 16), (ex + p2[j]) | ((ex + p3[j]) << 16)};
            R5_LDS(r5_v2u, GS_B + (uint32_t)(j * NT + tc) * 8u) = gs;
        }
    }
    __syncthreads();  // B4
    if (ovf || (own_range && misc[R5_M_HEAVY] >= 32u)) {
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    // ---- 5. decode: st[r] = start | arrival << 16 | cnt << 24
#pragma unroll
    for (int g = 0; g < ITEMS; g += G) {
        uint32_t cw[G], gs[G];
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (g + j >= ITEMS) continue;
            const uint32_t wo = st[g + j] & 0xfffcu;  // byte offset of counter word b >> 2; half of it: of its 16-bit group start
            cw[j] = R5_LDS(const uint32_t, CW_B + wo);
            gs[j] = R5_LDS(const unsigned short, GS_B + (wo >> 1));
        }
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (g + j >= ITEMS) continue;
            const uint32_t below = __builtin_amdgcn_alignbyte(cw[j], 0u, st[g + j]);
            const uint32_t start = __builtin_amdgcn_sad_u8(below, 0u, gs[j]);
            // cnt (byte b & 3 of the counter word) on top, the arrival number where it is, zeros below: ONE v_perm_b32 of
            // (counter byte in the low byte of the alignbyte result, arrival in byte 2 of st) instead of shift + and + or
            uint32_t hi16 = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(0u, cw[j], st[g + j]), st[g + j], 0x04020c0cu);
            if (ragged(g + j)) hi16 = valid(g + j) ? hi16 : (1u << 24);
            st[g + j] = hi16 | start;
        }
#pragma unroll
        for (int j = 0; j < G; j++)
            if (g + j < ITEMS) asm volatile("" : "+v"(st[g + j]));
        asm volatile("" ::: "memory");
    }
    __syncthreads();  // B5
#if R5W_BRANCHFREE
    // ---- 6. place the keys that share a bucket.  No branches: a key alone in its bucket writes a per-lane dummy word (the
    //         dead coarse histogram) — one select instead of an exec-mask round trip and a branch per key
    {
        const uint32_t dummy = C1_B + ((uint32_t)lane << 2);
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t pos = (st[r] & 0xffffu) + ((st[r] >> 16) & 255u);
            const uint32_t addr = st[r] >= (2u << 24) ? SLOT_B + (pos << 2) : dummy;
            R5_LDS(float, addr) = x[r];
        }
    }
    __syncthreads();  // B6
    // ---- 7. mates.  A key with cnt >= 2 reads its first mate (slot 0, or slot 1 for arrival 0), with cnt >= 3 its second (slot 2,
    //         or slot 1 for arrivals 2 and up); every other key reads +inf from a fixed word (one address for all of them: a
    //         broadcast): both reads of every key of a group in flight, no branch.  Equal mates (exact ties, -0 / +0) are noticed
    //         in a scalar lane mask — v_cmp_eq into an SGPR pair, s_or — and sorted out by the lanes it names afterwards.
    //         Buckets of four and more keys (0.2 % of the keys) recount their whole bucket.
    uint32_t tie = 0u;  // bit r: key r has an equal bucket mate (set in the rare paths only)
    constexpr int H = ITEMS <= 8 ? (ITEMS < 4 ? ITEMS : 4) : (ITEMS > 13 ? R5W_H16 : 8);  // (one group of eight at eight keys per thread spills)
    constexpr uint32_t INF_B = MISC_B + 28u * 4u;
#pragma unroll
    for (int h = 0; h < ITEMS; h += H) {
        float m1[H], m2[H];
        uint32_t res[H];
#pragma unroll
        for (int k = 0; k < H; k++) {
            const int r = h + k;
            if (r >= ITEMS) continue;
            const uint32_t start = st[r] & 0xffffu;
            const uint32_t a1 = start + ((st[r] & 0x00ff0000u) == 0u ? 1u : 0u);
            const uint32_t a2 = start + ((st[r] & 0x00fe0000u) != 0u ? 1u : 2u);
            m1[k] = R5_LDS(const float, st[r] >= (2u << 24) ? SLOT_B + (a1 << 2) : INF_B);
            m2[k] = R5_LDS(const float, st[r] >= (3u << 24) ? SLOT_B + (a2 << 2) : INF_B);
        }
        unsigned long long eqm = 0ull;  // lanes whose key of this group has an equal mate
#pragma unroll
        for (int k = 0; k < H; k++) {
            const int r = h + k;
            if (r >= ITEMS) continue;
            res[k] = st[r] + ((m1[k] < x[r]) ? 1u : 0u) + ((m2[k] < x[r]) ? 1u : 0u);
            eqm |= __builtin_amdgcn_fcmpf(m1[k], x[r], 1) | __builtin_amdgcn_fcmpf(m2[k], x[r], 1);  // FCMP_OEQ
        }
        unsigned long long big = 0ull;
#pragma unroll
        for (int k = 0; k < H; k++)
            if (h + k < ITEMS) big |= __builtin_amdgcn_uicmp(st[h + k], 4u << 24, 35);  // ICMP_UGE: cnt >= 4
        if ((big | eqm) != 0ull) {  // uniform, rare per group: recount the whole bucket of the keys concerned (start still intact in st)
            const bool mine = (((big | eqm) >> lane) & 1ull) != 0ull;
#pragma unroll
            for (int k = 0; k < H; k++) {
                const int r = h + k;
                if (r >= ITEMS) continue;
                if (mine && st[r] >= (2u << 24)) {
                    const uint32_t start = st[r] & 0xffffu, cnt = st[r] >> 24;
                    uint32_t lt = 0u, eq = 0u;
                    for (uint32_t j = 0; j < cnt; j += 4u) {
                        float mm[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            mm[i] = __uint_as_float(R5_INF);
                            if (j + (uint32_t)i < cnt) mm[i] = R5_LDS(const float, SLOT_B + ((start + j + (uint32_t)i) << 2));
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            lt += (mm[i] < x[r]) ? 1u : 0u;
                            eq += (mm[i] == x[r]) ? 1u : 0u;
                        }
                    }
                    res[k] = (start + lt) | (1u << 24);  // (cnt := 1: bit 31 must stay clear for the tie tag; eq counts the key itself once)
                    tie |= eq > 1u ? (1u << r) : 0u;
                }
                asm volatile("" ::: "memory");
            }
        }
        // (this group's ranks are final HERE: left to itself the compiler sinks every group's compares to the end of the step,
        // keeps all the mates alive until then and spills them)
#pragma unroll
        for (int k = 0; k < H; k++) {
            if (h + k >= ITEMS) continue;
            st[h + k] = res[k];
            asm volatile("" : "+v"(st[h + k]));
        }
        asm volatile("" : "+v"(tie)::"memory");
    }
#else
    // ---- 6. place the keys that share a bucket
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (st[r] >= (2u << 24)) {
            const uint32_t pos = (st[r] & 0xffffu) + ((st[r] >> 16) & 255u);
            R5_LDS(float, SLOT_B + (pos << 2)) = x[r];
        }
    }
    __syncthreads();  // B6
    // ---- 7. mates, eight keys at a time (register budget)
    uint32_t tie = 0u;
    constexpr int H = ITEMS <= 8 ? (ITEMS < 4 ? ITEMS : 4) : (ITEMS > 13 ? R5W_H16 : 8);  // (one group of eight at eight keys per thread spills)
#pragma unroll
    for (int h = 0; h < ITEMS; h += H) {
        float m1[H];
        uint32_t res[H];   // the group's ranks: st keeps the bucket starts until the end of the group (the tie recount needs them)
        uint32_t eqc = 0u; // equal mates seen by this lane in this group (exact ties, -0 / +0): v_cmp_eq + v_addc per compare
#pragma unroll
        for (int k = 0; k < H; k++) {
            const int r = h + k;
            if (r >= ITEMS) continue;
            m1[k] = __uint_as_float(R5_INF);
            if (st[r] >= (2u << 24)) {
                const uint32_t j1 = (st[r] & 0x00ff0000u) == 0u ? 1u : 0u;
                m1[k] = R5_LDS(const float, SLOT_B + (((st[r] & 0xffffu) + j1) << 2));
            }
        }
#pragma unroll
        for (int k = 0; k < H; k++) {
            const int r = h + k;
            if (r >= ITEMS) continue;
            if (st[r] >= (4u << 24)) {  // rare: the whole bucket
                const uint32_t start = st[r] & 0xffffu, cnt = st[r] >> 24;
                uint32_t lt = 0u, eq = 0u;
                for (uint32_t j = 0; j < cnt; j += 4u) {
                    float mm[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        mm[i] = __uint_as_float(R5_INF);
                        if (j + (uint32_t)i < cnt) mm[i] = R5_LDS(const float, SLOT_B + ((start + j + (uint32_t)i) << 2));
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        lt += (mm[i] < x[r]) ? 1u : 0u;
                        eq += (mm[i] == x[r]) ? 1u : 0u;
                    }
                }
                st[r] = (start + lt) | (1u << 24);
                tie |= eq > 1u ? (1u << r) : 0u;
                m1[k] = __uint_as_float(R5_INF);
            }
        }
#pragma unroll
        for (int k = 0; k < H; k++) {
            const int r = h + k;
            if (r >= ITEMS) continue;
            const uint32_t start = st[r] & 0xffffu;
            const float mm = m1[k];
            m1[k] = __uint_as_float(R5_INF);
            if (st[r] >= (3u << 24)) {
                const uint32_t j2 = (st[r] & 0x00fe0000u) != 0u ? 1u : 2u;
                m1[k] = R5_LDS(const float, SLOT_B + ((start + j2) << 2));
            }
            res[k] = st[r] + ((mm < x[r]) ? 1u : 0u);
            eqc += (mm == x[r]) ? 1u : 0u;
        }
#pragma unroll
        for (int k = 0; k < H; k++) {
            const int r = h + k;
            if (r >= ITEMS) continue;
            res[k] += (m1[k] < x[r]) ? 1u : 0u;
            eqc += (m1[k] == x[r]) ? 1u : 0u;
        }
        if (__any(eqc != 0u)) {  // rare (a tie or two per column): which keys of the lane it was — the whole bucket again
#pragma unroll
            for (int k = 0; k < H; k++) {
                const int r = h + k;
                if (r >= ITEMS) continue;
                if (eqc != 0u && st[r] >= (2u << 24)) {
                    const uint32_t start = st[r] & 0xffffu, cnt = st[r] >> 24;
                    uint32_t eq = 0u;
                    for (uint32_t j = 0; j < cnt; j++) eq += (R5_LDS(const float, SLOT_B + ((start + j) << 2)) == x[r]) ? 1u : 0u;
                    tie |= eq > 1u ? (1u << r) : 0u;  // (eq counts the key itself once)
                }
                asm volatile("" ::: "memory");
            }
        }
        // (this group's ranks are final HERE: left to itself the compiler sinks every group's compares to the end of the step,
        // keeps all the mates alive until then and spills them — one ds_read + s_waitcnt + scratch_store per key)
#pragma unroll
        for (int k = 0; k < H; k++) {
            if (h + k >= ITEMS) continue;
            st[h + k] = res[k];
            asm volatile("" : "+v"(st[h + k]));
        }
        asm volatile("" : "+v"(tie)::"memory");
    }
#endif
    if (tie != 0u) {
        const int tt = otid();
        auto elem = [&](int r) { return r < 4 * Q ? ((r >> 2) * NT + tt) * 4 + (r & 3) : r * NT + tt; };
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            if ((tie >> r) & 1u) {
                const uint32_t ti = atomicAdd(&misc[R5_M_TN], 1u);
                if (ti < (uint32_t)R5_TCAP) {
                    tkey[ti] = __float_as_uint(x[r]);
                    tpix[ti] = (uint32_t)elem(r);
                    tres[ti] = st[r] & 0xffffu;
                }
                st[r] = R5_TAG | ti;
            }
            asm volatile("" ::: "memory");
        }
    }
    // (match) the keys are dead: the sorted source column on its way into their registers
    r5_v4f sv[K::SQ];
    if (!emit) {
        const int ts = otid();
#pragma unroll
        for (int q = 0; q < K::SQ; q++) {
            const unsigned e0 = (unsigned)(q * NT + ts) * 4u;
            sv[q] = *reinterpret_cast<const r5_v4f*>(ssrt + (e0 < ns ? e0 : 0u));
        }
    }
    __syncthreads();  // B7
    const uint32_t tn = misc[R5_M_TN];
    if (tn > (uint32_t)R5_TCAP) {
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    for (uint32_t t = tid; t < tn; t += NT) {
        const uint32_t kb = tkey[t], pix = tpix[t];
        const float kf = __uint_as_float(kb);
        const uint32_t kk = f2key(kf);
        uint32_t before = 0u;
        for (uint32_t u = 0; u < tn; u++) {
            const float jf = __uint_as_float(tkey[u]);
            const uint32_t jk = f2key(jf);
            before += (jf == kf && (jk < kk || (jk == kk && tpix[u] < pix))) ? 1u : 0u;
        }
        tres[t] += before;
    }
    if (emit) {
        // ---- 8E / 9E. sorted keys / pixel indices: every owner writes its key (then its pixel number) to slot[rank] — every slot
        //               has been read — and the column leaves the LDS in order with 16-byte stores
        __syncthreads();  // (the tie results)
        if (tn != 0u) {
#pragma unroll
            for (int r = 0; r < ITEMS; r++)
                if ((st[r] & R5_TAG) != 0u) st[r] = tres[st[r] & ~R5_TAG];
        }
        const int te = otid();
        auto elem = [&](int r) { return r < 4 * Q ? ((r >> 2) * NT + te) * 4 + (r & 3) : r * NT + te; };
        const size_t obase = (size_t)col * (size_t)n;
        const bool ovec = (n % 4 == 0) && a.out_vec;
        auto drain = [&](uint32_t* dst) {
            __syncthreads();
            if (ovec) {
                for (int e = te * 4; e < n; e += NT * 4)
                    *reinterpret_cast<r5_v4u*>(dst + obase + e) = R5_LDS(const r5_v4u, SLOT_B + ((uint32_t)e << 2));
            } else {
                for (int e = te; e < n; e += NT) dst[obase + e] = R5_LDS(const uint32_t, SLOT_B + ((uint32_t)e << 2));
            }
        };
        if (a.out_keys) {
#pragma unroll
            for (int r = 0; r < ITEMS; r++)
                if (valid(r)) R5_LDS(float, SLOT_B + ((st[r] & 0xffffu) << 2)) = x[r];
            drain(reinterpret_cast<uint32_t*>(a.out_keys));
        }
        if (a.out_idx) {
            if (a.out_keys) __syncthreads();  // the keys have left the slot array
#pragma unroll
            for (int r = 0; r < ITEMS; r++)
                if (valid(r)) R5_LDS(uint32_t, SLOT_B + ((st[r] & 0xffffu) << 2)) = (uint32_t)elem(r);
            drain(a.out_idx);
        }
        if (a.only_flagged && tid == 0) a.flags[col] = 0;
        return;
    }
    // ---- 8. stage the source
    {
        const int tg = otid();
#pragma unroll
        for (int q = 0; q < K::SQ; q++) {
            const unsigned e0 = (unsigned)(q * NT + tg) * 4u;
            if (e0 < ns) R5_LDS(r5_v4f, GS_B + (e0 << 2)) = sv[q];
        }
    }
    __syncthreads();  // B8
    if (tn != 0u) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++)
            if ((st[r] & R5_TAG) != 0u) st[r] = tres[st[r] & ~R5_TAG];
    }
    // ---- 9. pick and store
    float v[ITEMS];
    auto pick = [&](auto same) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const unsigned rr = ragged(r) ? (valid(r) ? (st[r] & 0xffffu) : 0u) : (st[r] & 0xffffu);
            unsigned qi = rr;
            if (!decltype(same)::value)  // floor((2 rank + 1) ns / (2 n)), exact: launch_rank5w's multiplier (below)
                qi = __umulhi(__umul24(rr, 2u * ns) + ns, a.qmul) >> a.qshr;
            v[r] = R5_LDS(const float, GS_B + (qi << 2));
            if ((r & 3) == 3) asm volatile("" ::: "memory");
        }
    };
    if (ns == (unsigned)n) pick(std::true_type{});
    else pick(std::false_type{});
    const int to = otid();
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int e0 = (q * NT + to) * 4;
        if (!ragged(4 * q) || e0 < n)
            *reinterpret_cast<float4*>(o + e0) = make_float4(v[4 * q], v[(4 * q + 1) % ITEMS], v[(4 * q + 2) % ITEMS], v[(4 * q + 3) % ITEMS]);
    }
#pragma unroll
    for (int r = 4 * Q; r < ITEMS; r++)
        if (valid(r)) o[r * NT + to] = v[r];
}

// 6400 keys (a pass size of the 512^2 schedule) fill 640 threads x 10 keys exactly, three workgroups to a CU (as in sort_rank4.hip)
// Workgroup shape by column length (profiles/r06_sort_experiments.md): 13 .. 16 keys per thread and as many workgroups per CU as
// the LDS takes beat fewer, larger workgroups with 8 .. 10 keys per thread — 6400 keys: 448 threads x 15 keys, four workgroups
// per CU, 281 us against 343 us for 640 x 10 x three.  Probe builds move the class boundaries' shapes with -DR5W_NT_*.
#ifndef R5W_NT_A
#define R5W_NT_A 256    // 2048 < n <= 4096: six workgroups of four wavefronts per CU (0.570 against 0.556 with 512 threads)
#endif
#ifndef R5W_NT_B
#define R5W_NT_B 448    // 4096 < n <= 7168
#endif
#ifndef R5W_NT_C
#define R5W_NT_C 1024   // 7168 < n <= 9216
#endif
#ifndef R5W_NT_D
#define R5W_NT_D 1024   // 9216 < n <= 13312
#endif
static int rank5w_threads(long n) { return n <= 4096 ? R5W_NT_A : (n <= 7168 ? R5W_NT_B : (n <= 9216 ? R5W_NT_C : (n <= 13312 ? R5W_NT_D : 1024))); }

// Can launch_rank5w take this call?  mode = SORT_MATCH (with or without a caller-given range) or SORT_EMIT (keys and / or pixel
// indices).  Everything else stays with rank_match4_kernel.
// (The library compiles this file twice — R5W_TU = 0: the match instantiations and everything else, R5W_TU = 1: the EMIT
//  instantiations — so that the 156 kernels build side by side; a build without R5W_TU, e.g. a probe, holds everything.)
#if !defined(R5W_TU) || R5W_TU == 0
bool rank5w_supported(int mode, const SortArgs& a) {
    if (a.n <= 2048 || a.n > SORT_MAX_N) return false;
    const long nt = rank5w_threads(a.n), items = (a.n + nt - 1) / nt;
    // 16-byte loads: rows on 16-byte boundaries; a keys-per-thread count without scalar rows needs whole quads
    if (a.ld % 4 != 0 || a.ss % 4 != 0 || (reinterpret_cast<uintptr_t>(a.keys) & 15u) != 0) return false;
    if (items % 4 == 0 && a.n % 4 != 0) return false;
    if (items < 4 || items > 16) return false;
    if ((a.rng_lo == nullptr) != (a.rng_hi == nullptr)) return false;
    if (mode == SORT_EMIT) return a.out_keys != nullptr || a.out_idx != nullptr;
    if (!a.src_sorted || !a.out) return false;
    if (a.ldo % 4 != 0 || a.oss % 4 != 0 || (reinterpret_cast<uintptr_t>(a.out) & 15u) != 0) return false;
    const long qr = (items * 3 + 15) / 16, nwrd = 4 * nt * qr;
    long sq = (6 * nwrd / 16 + nt - 1) / nt;
    if (sq > 4) sq = 4;
    const long src_max = 6 * nwrd / 4 < 4 * nt * sq ? 6 * nwrd / 4 : 4 * nt * sq;
    if (a.ns % 4 != 0 || (reinterpret_cast<uintptr_t>(a.src_sorted) & 15u) != 0 || a.ns > src_max) return false;
    return true;
}

#endif

// floor(A / d) for every A < 2^30 as (A * m) >> (30 + l) with l = ceil(log2 d), m = ceil(2^(30 + l) / d) < 2^32 (Granlund &
// Montgomery 1994, Theorem 4.2 for 30-bit dividends: 2^(30+l) <= m d <= 2^(30+l) + 2^l).  Here d = 2 n and
// A = (2 rank + 1) ns <= 32767 * 18432 < 2^30:  quantile index = umulhi(A, qmul) >> qshr,  qshr = l - 2.
static void quantile_magic(SortArgs& a) {
    const unsigned long long d = 2ull * (unsigned long long)a.n;
    int l = 0;
    while ((1ull << l) < d) l++;
    a.qmul = (unsigned)(((1ull << (30 + l)) + d - 1ull) / d);
    a.qshr = l - 2;
}

template <int ITEMS, int NT, bool EMIT>
static int launch_rank5w_items(const SortArgs& a0, int ncols, hipStream_t st) {
    SortArgs a = a0;
    if (EMIT) {
        a.out = nullptr;   // contiguous [column, n] outputs: 16-byte stores when the columns start on 16-byte boundaries
        a.out_vec = (a.n % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out_keys) & 15u) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.out_idx) & 15u) == 0) ? 1 : 0;
    } else {
        quantile_magic(a);
    }
    const size_t lds = R5W<ITEMS, NT>::LDS;
    const bool full = a.n == (long)ITEMS * NT;
    auto go = [&](auto kern, DeviceOnce& once) {
        bool& attr = *once.slot();
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error("sort: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)OPTEX_E_LAUNCH; }
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(ncols), dim3(NT), lds, st, a);
        return (int)OPTEX_OK;
    };
    int rc;
    if (full) {
        static DeviceOnce once;
        rc = go(rank_match5w_kernel<ITEMS, NT, true, EMIT>, once);
    } else {
        static DeviceOnce once;
        rc = go(rank_match5w_kernel<ITEMS, NT, false, EMIT>, once);
    }
    if (rc) return rc;
    return check_launch("rank_match5w_kernel");
}

template <int NT, bool EMIT>
static int launch_rank5w_nt(const SortArgs& a, int ncols, hipStream_t st) {
    switch ((int)((a.n + NT - 1) / NT)) {
        case 4: return launch_rank5w_items<4, NT, EMIT>(a, ncols, st);
        case 5: return launch_rank5w_items<5, NT, EMIT>(a, ncols, st);
        case 6: return launch_rank5w_items<6, NT, EMIT>(a, ncols, st);
        case 7: return launch_rank5w_items<7, NT, EMIT>(a, ncols, st);
        case 8: return launch_rank5w_items<8, NT, EMIT>(a, ncols, st);
        case 9: return launch_rank5w_items<9, NT, EMIT>(a, ncols, st);
        case 10: return launch_rank5w_items<10, NT, EMIT>(a, ncols, st);
        case 11: return launch_rank5w_items<11, NT, EMIT>(a, ncols, st);
        case 12: return launch_rank5w_items<12, NT, EMIT>(a, ncols, st);
        case 13: return launch_rank5w_items<13, NT, EMIT>(a, ncols, st);
        case 14: return launch_rank5w_items<14, NT, EMIT>(a, ncols, st);
        case 15: return launch_rank5w_items<15, NT, EMIT>(a, ncols, st);
        default: return launch_rank5w_items<16, NT, EMIT>(a, ncols, st);
    }
}

template <bool EMIT>
static int launch_rank5w_mode(const SortArgs& a, int ncols, hipStream_t st) {
    const int nt = rank5w_threads(a.n);
    if (nt == R5W_NT_A) return launch_rank5w_nt<R5W_NT_A, EMIT>(a, ncols, st);
    if (nt == R5W_NT_B) return launch_rank5w_nt<R5W_NT_B, EMIT>(a, ncols, st);
    if (nt == R5W_NT_C) return launch_rank5w_nt<R5W_NT_C, EMIT>(a, ncols, st);
    if (nt == R5W_NT_D) return launch_rank5w_nt<R5W_NT_D, EMIT>(a, ncols, st);
    return launch_rank5w_nt<1024, EMIT>(a, ncols, st);
}

int launch_rank5w_emit(const SortArgs& a, int ncols, hipStream_t st);
#if !defined(R5W_TU) || R5W_TU == 1
int launch_rank5w_emit(const SortArgs& a, int ncols, hipStream_t st) { return launch_rank5w_mode<true>(a, ncols, st); }
#endif
#if !defined(R5W_TU) || R5W_TU == 0
int launch_rank5w(int mode, const SortArgs& a, int ncols, hipStream_t st) {
    return mode == SORT_MATCH ? launch_rank5w_mode<false>(a, ncols, st) : launch_rank5w_emit(a, ncols, st);
}
#endif

#ifdef R5_PERSISTENT_VARIANT
// Workgroup shape by column length: one 1024-thread workgroup per CU above 8192 keys, two of 512 threads down to 4097, four
// of 256 below — always 16 wavefronts per CU at a 128-register budget, 9 .. 16 keys per thread (8 .. 16 with 256 threads).
static int rank5_threads(long n) { return n > 8192 ? 1024 : (n > 4096 ? 512 : 256); }

// Can launch_rank5 take this match?  (everything else stays with rank_match4_kernel)
bool rank5_supported(const SortArgs& a) {
    if (!a.rng_lo || !a.rng_hi || !a.src_sorted || !a.out) return false;
    if (a.n <= 2048 || a.n > SORT_MAX_N) return false;
    const long nt = rank5_threads(a.n), items = (a.n + nt - 1) / nt;
    // 16-byte loads and stores: rows on 16-byte boundaries; a keys-per-thread count without scalar rows needs whole quads
    if (a.ld % 4 != 0 || a.ss % 4 != 0 || (reinterpret_cast<uintptr_t>(a.keys) & 15u) != 0) return false;
    if (a.ldo % 4 != 0 || a.oss % 4 != 0 || (reinterpret_cast<uintptr_t>(a.out) & 15u) != 0) return false;
    if (items % 4 == 0 && a.n % 4 != 0) return false;
    if (items < 4 || items > 16) return false;
    // the sorted source column is staged with 16-byte loads over the group starts and the counters
    const long qr = (items + 3) / 4;
    if (a.ns % 4 != 0 || (reinterpret_cast<uintptr_t>(a.src_sorted) & 15u) != 0 || a.ns > 6 * (4 * nt * qr) / 4) return false;
    return true;
}

template <int ITEMS, int NT>
static int launch_rank5_items(const SortArgs& a, int ncols, hipStream_t st) {
    const size_t lds = R5<ITEMS, NT>::LDS;
    const bool full = a.n == (long)ITEMS * NT;
    int grid = device_cu_count() * (1024 / NT);
    if (grid > ncols) grid = ncols;
    auto go = [&](auto kern, DeviceOnce& once) {
        bool& attr = *once.slot();
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error("sort: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)OPTEX_E_LAUNCH; }
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a);
        return (int)OPTEX_OK;
    };
    int rc;
    if (full) {
        static DeviceOnce once;
        rc = go(rank_match5_kernel<ITEMS, NT, true>, once);
    } else {
        static DeviceOnce once;
        rc = go(rank_match5_kernel<ITEMS, NT, false>, once);
    }
    if (rc) return rc;
    return check_launch("rank_match5_kernel");
}

template <int NT>
static int launch_rank5_nt(const SortArgs& a, int ncols, hipStream_t st) {
    switch ((int)((a.n + NT - 1) / NT)) {
        case 8: if (NT == 256) return launch_rank5_items<8, NT>(a, ncols, st);  // (2048 < n: only 256 threads get here)
        case 9: return launch_rank5_items<9, NT>(a, ncols, st);
        case 10: return launch_rank5_items<10, NT>(a, ncols, st);
        case 11: return launch_rank5_items<11, NT>(a, ncols, st);
        case 12: return launch_rank5_items<12, NT>(a, ncols, st);
        case 13: return launch_rank5_items<13, NT>(a, ncols, st);
        case 14: return launch_rank5_items<14, NT>(a, ncols, st);
        case 15: return launch_rank5_items<15, NT>(a, ncols, st);
        default: return launch_rank5_items<16, NT>(a, ncols, st);
    }
}

int launch_rank5(const SortArgs& a, int ncols, hipStream_t st) {
    switch (rank5_threads(a.n)) {
        case 256: return launch_rank5_nt<256>(a, ncols, st);
        case 512: return launch_rank5_nt<512>(a, ncols, st);
        default: return launch_rank5_nt<1024>(a, ncols, st);
    }
}

#endif  // R5_PERSISTENT_VARIANT

}  // namespace optex
