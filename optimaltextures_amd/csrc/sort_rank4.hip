// sort_rank4.hip — rank_match4_kernel: the exact 1-D transport match (north-star addition; specification =
// oracle/optex_oracle.c orc_sort_match), owner-ranked, built around what a VALU instruction COSTS on
// gfx950.  Measured (scripts/valu_lds_rate_probe.hip, profiles/r02_valu_lds_rate_probe.log, 8 waves per SIMD): a
// wave-instruction takes ~2.4 cycles of its SIMD if it is v_add_u32 / v_and_b32 / v_mov_b32 / v_mul_f32 / v_fma_f32 (and
// v_sub / v_add_f32), and ~4.1 cycles if it is anything else the previous generation (an integer-key owner-ranked kernel, round 2) was made of — v_cmp_*, v_addc_co_u32,
// v_cndmask, v_min/max, shifts, v_bfe, v_lshl_add, v_mad_u24, every v_cvt, every fp64 op.  At 141 mostly 4-cycle
// instructions per key that kernel kept the VALU busy ~90 % of its run (SQ_ACTIVE_INST_VALU x 4 / cycles): it was
// VALU-bound, not LDS-bound.  This kernel spends the fast class wherever the work allows it:
//
//   1. FLOAT-DOMAIN keys.  The slots hold the raw fp32 values and every comparison is a float comparison, so the
//      totalOrder key (3 instructions per key, and 3 more every time the value is needed back) disappears.  Float order
//      equals totalOrder except for -0 / +0, which compare equal and therefore take the tie path, where the full key
//      decides.  Non-finite keys are caught by one fma per key (x * 0 is NaN for inf / NaN).
//   2. COUNTING WITHOUT COMPARE INSTRUCTIONS.  For a window slot x_j and the owner's key k,
//          c_j = clamp(fma(k - x_j, 2^127, 1/16))   is 1 if x_j < k, 1/16 if x_j == k, 0 if x_j > k
//      (the clamp is a free output modifier; +-inf products clamp correctly).  Summed into an accumulator that starts at
//      2^19 (ulp = 1/16, every partial sum exact) the low mantissa byte IS 16 * #{x_j < k} + #{x_j == k}: three 2-cycle
//      instructions per slot give both counts, where v_cmp_lt + v_addc + v_cmp_le + v_addc cost four 4-cycle ones.
//      Exact unless two DISTINCT keys differ by less than 2^-127, which needs both below 2^-103 in magnitude: a column
//      holding a non-zero key below 2^-100 is handed to the radix kernel (three 2-cycle instructions per key to find out).
//   3. BUCKET MAP IN FIVE INSTRUCTIONS + ONE 8-BYTE TABLE READ: t = (x - lo) * s1, bin = int(t), bucket =
//      int(fma(t, ww_bin, K_bin)) with ww = w - 1/8 and K = base + 1/16 - bin * ww — both exactly representable, so the
//      fma is one rounding of the exact affine map and lands strictly inside (base, base + w): monotone non-decreasing
//      over the whole column without a single clamp.
//   4. 8-BYTE ALIGNED WINDOWS: four ds_read_b64 at start & ~1 instead of two ds_read_b128 at start & ~3 (fewer LDS cycles
//      at random addresses, see the probe) — and a bucket overflows the window only if (start & 1) + count > 8: ~0.3 % of
//      the keys are queued instead of ~3 %, so the per-key queue bookkeeping is branched over for most wavefront rows
//      and the queue phase shrinks to a fraction of a wavefront.
//   5. Validity tests only on the register rows that can be ragged (the last row, or the last 16-byte quad), none when
//      the column fills the workgroup exactly.
//
// Around that: histogram-equalised monotone bucket map, returning count atomics, one-barrier scan, all-equal big buckets,
// tie list, radix fallback through the flags (sort.hip), staged sorted source column, 16-byte stores.
#include "sort_common.h"
#include <type_traits>

namespace optex {

constexpr uint32_t R4_TAG = 0x80000000u;   // ba: rank pending in queue entry (low bits)
// start entry with BIGF set and LONG clear: rank already final (all-equal big bucket; the scan sets both for a big bucket)
constexpr uint32_t R4_LONG = 0x8000u;      // start entry: bucket does not fit the aligned 8-slot window
constexpr uint32_t R4_BIGF = 0x4000u;      // start entry: bucket larger than RK_BIG
constexpr uint32_t R4_SMASK = 0x3fffu;     // start entry: first slot of the bucket (mod 16384)
constexpr int R4_WIN = 8;
constexpr int R4_QWIN = 52;                // window of a queued key, slots: covers (start & 3) + RK_BIG
constexpr int R4_BBITS = 14;               // bits of 2 * bucket + half in the owner's (b2, a) register
constexpr int R4_TAB = 512;                // word offset of the (ww, K) table inside the (still empty) slot array
constexpr uint32_t R4_INF = 0x7f800000u;   // +inf: larger than every finite key

template <int ITEMS, int NT>
struct R4 {
    static constexpr int CAP = ITEMS * NT;
    // buckets + 1 spare per coarse bin; 16384 keys: what fits 80 KiB next to the 64 KiB of slots
    static constexpr int NBT = CAP == 16384 ? 7872 : (CAP < 8192 ? CAP : 8192);
    static constexpr int NB = NBT - RK_COARSE;
    static constexpr int NW2 = NBT / 2;                               // packed u16 counters -> start entries
    static constexpr int PER = (NW2 + NT - 1) / NT;
    static constexpr int NWORDS = CAP / 32;
    static constexpr int TCAP = 256;                                  // queue entries with an equal partner
    static constexpr int QCAP = (NW2 - TCAP) / 4;                     // key, window, pixel, result per queue entry
    static constexpr int SLOTW = CAP + R4_QWIN + 4;                   // + window padding behind the last key + a dummy slot
    static constexpr int CNTW = NW2 + 4;                              // + the entry behind the last bucket
    static constexpr size_t LDS = (size_t)(SLOTW + CNTW + 32 + 32) * 4;
    static_assert(2 * NBT <= (1 << R4_BBITS), "2 * bucket + 1 must fit its bit field");
    static_assert((NW2 + 4) % 4 == 0, "the arrays behind the counters stay 16-byte aligned");
    static_assert(2 * NWORDS <= NW2, "big-bucket scratch aliases the counters");
    static_assert(NBT % 2 == 0 && RK_COARSE <= CAP, "layout");
    static_assert(R4_QWIN % 4 == 0 && R4_QWIN >= 3 + RK_BIG, "a queued key's window must cover its bucket");
    static_assert(R4_TAB + 2 * (RK_COARSE + 1) <= CAP, "the bucket table lives in the slot array");
};

// LDS access by BYTE OFFSET from the start of the workgroup's LDS.  The kernel has no static LDS, so its dynamic LDS
// begins at address 0 (the kernel checks it once per workgroup): going through smem-derived pointers makes the compiler
// add that (link-time) zero to every address with a v_add_u32 of its own.
typedef unsigned r4_v2u __attribute__((ext_vector_type(2)));
typedef float r4_v2f __attribute__((ext_vector_type(2)));
typedef float r4_v4f __attribute__((ext_vector_type(4)));
#define R4_LDS(T, off) (*reinterpret_cast<__attribute__((address_space(3))) T*>((uint32_t)(off)))

__device__ __forceinline__ float r4_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float r4_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// wave-wide inclusive scan with six DPP adds, no LDS: row_shr 1 / 2 / 4 / 8 inside the 16-lane rows (lanes without a
// source add 0), then row_bcast 15 into rows 1 and 3 and row_bcast 31 into rows 2 and 3 (verified on gfx950 by
// scripts/valu_lds_rate_probe.hip)
__device__ __forceinline__ unsigned r4_wave_incl_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// acc += sum over the 8 window slots of clamp((k - x_j) * 2^127 + 1/16): 1 per smaller slot, 1/16 per equal slot.
// One asm block: 24 two-cycle instructions, two temporaries so that neighbouring chains overlap.
__device__ __forceinline__ void r4_window(float& acc, const r4_v2u& a0, const r4_v2u& a1, const r4_v2u& a2, const r4_v2u& a3, float k,
                                          float big, float c16) {
    float t0, t1;
    asm("v_sub_f32 %1, %11, %3\n\t"
        "v_sub_f32 %2, %11, %4\n\t"
        "v_fma_f32 %1, %1, %12, %13 clamp\n\t"
        "v_fma_f32 %2, %2, %12, %13 clamp\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_sub_f32 %1, %11, %5\n\t"
        "v_add_f32 %0, %0, %2\n\t"
        "v_sub_f32 %2, %11, %6\n\t"
        "v_fma_f32 %1, %1, %12, %13 clamp\n\t"
        "v_fma_f32 %2, %2, %12, %13 clamp\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_sub_f32 %1, %11, %7\n\t"
        "v_add_f32 %0, %0, %2\n\t"
        "v_sub_f32 %2, %11, %8\n\t"
        "v_fma_f32 %1, %1, %12, %13 clamp\n\t"
        "v_fma_f32 %2, %2, %12, %13 clamp\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_sub_f32 %1, %11, %9\n\t"
        "v_add_f32 %0, %0, %2\n\t"
        "v_sub_f32 %2, %11, %10\n\t"
        "v_fma_f32 %1, %1, %12, %13 clamp\n\t"
        "v_fma_f32 %2, %2, %12, %13 clamp\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_add_f32 %0, %0, %2"
        : "+v"(acc), "=&v"(t0), "=&v"(t1)
        : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(a2.x), "v"(a2.y), "v"(a3.x), "v"(a3.y), "v"(k), "s"(big), "v"(c16));
}

// lt += #{x < k}, le += #{x <= k} over four slots, float order (the queue: a handful of keys per column)
__device__ __forceinline__ void r4_count4(uint32_t& lt, uint32_t& le, const uint4& x, float k) {
    asm("v_cmp_lt_f32 vcc, %2, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_f32 vcc, %2, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_f32 vcc, %3, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_f32 vcc, %3, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_f32 vcc, %4, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_f32 vcc, %4, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cmp_lt_f32 vcc, %5, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_le_f32 vcc, %5, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lt), "+v"(le)
        : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "v"(k)
        : "vcc");
}

// NT threads per workgroup (1024, or 512 / 256 for short columns: more columns resident per CU).  FULL: the column
// fills all ITEMS * NT registers (no validity tests at all).
// MODE = SORT_MATCH: out[pixel] = sorted_source[q(rank)].  MODE = SORT_EMIT (optex_sort_columns): the sorted keys and / or
// their pixel indices, contiguous [column, n], written by rank through the (then dead) slot array.
template <int ITEMS, bool VEC, int NT, bool FULL, int MODE = SORT_MATCH>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void rank_match4_kernel(SortArgs a) {
    using K = R4<ITEMS, NT>;
    constexpr int NW = NT / 64;
    constexpr int CAP = K::CAP, NB = K::NB, NW2 = K::NW2, PER = K::PER, NWORDS = K::NWORDS, QCAP = K::QCAP,
                  TCAP = K::TCAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // counters first: every array but the slots then sits below 64 KiB, where a DS instruction's 16-bit offset field
    // reaches it without an address add
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);    // [NW2 + 4] packed u16 bucket counts -> start entries
    uint32_t* red = cnt + K::CNTW;                        // [32]
    uint32_t* misc = red + 32;                            // [32] nbig, noteq, (start, count) x RK_MAXBIG, [20] queue length
    uint32_t* slot = misc + 32;                           // [CAP + 52 + 4] fp32 keys by bucket position; later the sorted source column
    uint32_t* c1 = slot;                                  // [256] coarse histogram (dead before the slots fill)
    float2* tab = reinterpret_cast<float2*>(slot + R4_TAB);  // [257] (ww, K) per coarse bin (dead before the slots fill)
    uint32_t* qkey = cnt;                                 // queue (the start entries are dead by then)
    uint32_t* qwin = cnt + QCAP;
    uint32_t* qpix = cnt + 2 * QCAP;
    uint32_t* qres = cnt + 3 * QCAP;
    uint32_t* tlist = cnt + 4 * QCAP;                     // [TCAP] queue entries whose key has an equal partner
    uint32_t* bitmap = cnt;                               // [NWORDS] big-bucket pass
    uint32_t* bpre = cnt + NWORDS;                        // [NWORDS]

    constexpr uint32_t CNT_B = 0u, SLOT_B = (uint32_t)(K::CNTW + 64) * 4u, TAB_B = SLOT_B + (uint32_t)R4_TAB * 4u;

    const int col = blockIdx.x, seg = col / a.C, c = col % a.C;
    const int xseg = (a.x_n_seg == 1) ? 0 : seg;
    const float* src = a.keys + (size_t)xseg * a.ss + (size_t)c * a.ld;
    const int sseg = (a.src_n_seg == 1) ? 0 : seg;
    const float* ssrt = MODE == SORT_MATCH ? a.src_sorted + ((size_t)sseg * a.C + c) * a.ns : nullptr;
    float* o = MODE == SORT_MATCH ? a.out + (size_t)seg * a.oss + (size_t)c * a.ldo : nullptr;
    const unsigned ns = MODE == SORT_MATCH ? (unsigned)a.ns : 1u;
    const int n = FULL ? CAP : (int)a.n;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // VEC: the first 4 * Q registers are Q 16-byte loads (4 neighbouring pixels in one thread), the remaining T = ITEMS % 4
    // registers are scalar rows behind them (row r starts at pixel r * NT either way)
    constexpr int Q = VEC ? ITEMS / 4 : 0, T = ITEMS - 4 * Q;
    // pixel held in register r
    auto elem = [&](int r) { return r < 4 * Q ? ((r >> 2) * NT + tid) * 4 + (r & 3) : r * NT + tid; };
    // can register r lie past the end of the column?  ITEMS = ceil(n / NT) (the launcher guarantees it for ITEMS > 2):
    // only the last row — without scalar rows the last quad — can be ragged
    auto ragged = [](int r) { return !FULL && (ITEMS == 2 || ((VEC && T == 0) ? r >= ITEMS - 4 : r == ITEMS - 1)); };
    // register r holds a pixel of the column: a compare of tid with a scalar
    auto valid = [&](int r) { return !ragged(r) || ((VEC && T == 0) ? tid < (n >> 2) - (r >> 2) * NT : tid < n - r * NT); };

    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) {
        if (threadIdx.x == 0) a.flags[blockIdx.x] = 1;  // never on this toolchain: the radix kernel would take every column
        return;
    }
    // as the SECOND ranking kernel behind rank_match5w_kernel (sort.hip: calls whose columns may hold massive ties) only the
    // columns that one flagged are taken, and a column ranked here is un-flagged again for the radix sweep
    if (a.only_flagged && a.flags[blockIdx.x] == 0) return;  // (ranked by rank_match5w_kernel)
    SORT_PROBE(0);
    // ---- 0. the column (registers past the end hold a copy of a real key: harmless for min / max, and they stay out of
    //         every LDS update below through selects)
    float x[ITEMS];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int e0 = (q * NT + tid) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + (ragged(4 * q) ? (e0 < n ? e0 : 0) : e0));
        x[4 * q + 0] = v.x;
        x[(4 * q + 1) % ITEMS] = v.y;
        x[(4 * q + 2) % ITEMS] = v.z;
        x[(4 * q + 3) % ITEMS] = v.w;
    }
#pragma unroll
    for (int r = 4 * Q; r < ITEMS; r++) {
        const int e = r * NT + tid;
        x[r] = src[ragged(r) ? (e < n ? e : n - 1) : e];
    }
    for (int i = tid; i < K::CNTW; i += NT) cnt[i] = 0u;
    if (tid < RK_COARSE) c1[tid] = 0u;
    if (tid < 32) misc[tid] = 0u;

    // ---- 1. min / max; non-finite keys (x * 0 is NaN); non-zero keys below 2^-100 (the counting of step 7 is exact only
    //         if distinct keys differ by 2^-127 or more)
    float lo = x[0], hi = x[0], nf = 0.f, tz = 0.f;
    const bool range_given = a.rng_lo != nullptr;  // uniform: the caller knows the column's range (SortArgs::rng_lo)
    if (range_given) {
        // The range comes from the rotation GEMM's epilogue (min / max of exactly the values stored here): no reduction, and
        // the barrier below only publishes the zeroed counters — it is passed while the column is still on its way from
        // HBM, every wavefront then goes on as soon as ITS keys are there.  v_min / v_max drop NaN, so the non-finite /
        // tiny-key test stays per key; it reports through misc[22] (read behind the scan's barrier, before anything is stored).
        lo = a.rng_lo[col];
        hi = a.rng_hi[col];
        __syncthreads();
        const float p100 = 1.2676506e30f;  // 2^100
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            float p, z;
            asm("v_fma_f32 %0, %3, 0, %0\n\t"
                "v_mul_f32_e64 %1, |%3|, %4 clamp\n\t"
                "v_fma_f32 %2, -%1, %1, %1\n\t"
                : "+v"(nf), "=&v"(p), "=&v"(z)
                : "v"(x[r]), "s"(p100));
            tz += z;
        }
        const bool bad = !(nf == 0.f) || tz > 0.f;
        if (__any(bad) && lane == 0) misc[22] = 1u;
    } else {
    {
        const float p100 = 1.2676506e30f;  // 2^100
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            lo = r4_min(lo, x[r]);
            hi = r4_max(hi, x[r]);
            float p, z;
            asm("v_fma_f32 %0, %3, 0, %0\n\t"           // nf += x * 0
                "v_mul_f32_e64 %1, |%3|, %4 clamp\n\t"   // p = min(1, |x| * 2^100): 0 or 1 unless 0 < |x| < 2^-100
                "v_fma_f32 %2, -%1, %1, %1\n\t"          // z = p - p * p: positive exactly then
                : "+v"(nf), "=&v"(p), "=&v"(z)
                : "v"(x[r]), "s"(p100));
            tz += z;
        }
    }
    {
        // a wavefront that saw a non-finite or a tiny key reports hi = +inf: the column is handed to the radix kernel
        const bool bad = !(nf == 0.f) || tz > 0.f;
        if (__any(bad)) hi = __uint_as_float(R4_INF);
    }
    // wave minimum / maximum without LDS round trips: four DPP butterflies inside the 16-lane rows (quad_perm [1,0,3,2],
    // [2,3,0,1], row_half_mirror, row_mirror), then the four row results through readlane
    {
        auto dpp = [](float v, auto ctrl) {
            return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, false));
        };
        lo = r4_min(lo, dpp(lo, std::integral_constant<int, 0xb1>{}));
        hi = r4_max(hi, dpp(hi, std::integral_constant<int, 0xb1>{}));
        lo = r4_min(lo, dpp(lo, std::integral_constant<int, 0x4e>{}));
        hi = r4_max(hi, dpp(hi, std::integral_constant<int, 0x4e>{}));
        lo = r4_min(lo, dpp(lo, std::integral_constant<int, 0x141>{}));
        hi = r4_max(hi, dpp(hi, std::integral_constant<int, 0x141>{}));
        lo = r4_min(lo, dpp(lo, std::integral_constant<int, 0x140>{}));
        hi = r4_max(hi, dpp(hi, std::integral_constant<int, 0x140>{}));
        auto rl = [](float v, auto l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), decltype(l)::value)); };
        lo = r4_min(r4_min(rl(lo, std::integral_constant<int, 0>{}), rl(lo, std::integral_constant<int, 16>{})),
                    r4_min(rl(lo, std::integral_constant<int, 32>{}), rl(lo, std::integral_constant<int, 48>{})));
        hi = r4_max(r4_max(rl(hi, std::integral_constant<int, 0>{}), rl(hi, std::integral_constant<int, 16>{})),
                    r4_max(rl(hi, std::integral_constant<int, 32>{}), rl(hi, std::integral_constant<int, 48>{})));
    }
    if (lane == 0) {
        red[w] = __float_as_uint(lo);
        red[16 + w] = __float_as_uint(hi);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; k++) {
        lo = r4_min(lo, __uint_as_float(red[k]));
        hi = r4_max(hi, __uint_as_float(red[16 + k]));
    }
    }
    // (red is next written by the scan of step 5, two barriers from here)
    if (!(hi < __uint_as_float(R4_INF)) || !(lo > -__uint_as_float(R4_INF))) {  // non-finite / tiny keys: radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    if (lo == hi) {
        if (lo == 0.f) {  // zeros of both signs may be mixed (-0 < +0 in the specification): radix kernel
            if (tid == 0) a.flags[col] = 1;
            return;
        }
        // constant column: already sorted, rank = pixel index
        if (MODE == SORT_MATCH) {
            for (int e = tid; e < n; e += NT) o[e] = ssrt[quantile_index((uint32_t)e, ns, (unsigned)n, a.inv_2nt)];
        } else {
            for (int e = tid; e < n; e += NT) {
                if (a.out_keys) a.out_keys[(size_t)col * n + e] = lo;
                if (a.out_idx) a.out_idx[(size_t)col * n + e] = (uint32_t)e;
            }
        }
        if (a.only_flagged && tid == 0) a.flags[col] = 0;
        return;
    }
    const float s1 = __fdiv_rn((float)RK_COARSE, hi - lo);
    if (!(s1 > 0.f) || !(s1 < 1.0e37f)) {  // range over/underflow (8 * s1 must stay finite): radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }

    SORT_PROBE(1);
    // ---- 2. coarse histogram of a spatially spread quarter sample (any widths give a monotone map; the sample only
    //         balances the bucket sizes)
    constexpr int RS = VEC ? 4 : (ITEMS >= 8 ? 4 : 1);
    unsigned nsamp = 0;
    if (VEC && T == 0) {
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
            const float t = (x[r] - lo) * s1;
            int bin = (int)t;
            bin = bin > RK_COARSE - 1 ? RK_COARSE - 1 : bin;
            atomicAdd(&c1[bin], 1u);
        }
    }
    __syncthreads();
    SORT_PROBE(2);
    // ---- 3. equalisation: coarse bin b gets w_b = 1 + cnt_b * NB / nsamp fine buckets from base_b on.  With
    //         ww = w_b - 1/8 and K = base_b + 1/16 - b * ww,  u = t * ww + K  for t = (x - lo) * s1 in [b, b + 1) lies in
    //         [base + 1/16, base + w - 1/16): ww and b * ww are exact in fp32, K and the fma round by less than 1/16
    //         together, so int(u) stays inside its bin's bucket range — monotone over the column, no clamp.  The table
    //         holds (ww / 4, 2 K) for the argument t8 = 8 t (exact scalings): step 4 gets 2 u, whose integer part
    //         2 * bucket + half gives the counter's byte address and the u16 half with an AND each.
    //         One wavefront, four bins per lane.
    if (w == 0) {
        const uint4 cc = *reinterpret_cast<const uint4*>(c1 + 4 * lane);
        auto width = [&](unsigned cn) {
            const unsigned xx = cn * (unsigned)NB;  // < 2^27: exact quotient via a float estimate + one correction
            unsigned q = (unsigned)((float)xx / (float)nsamp);
            if (q * nsamp > xx) q--;
            else if ((q + 1u) * nsamp <= xx) q++;
            return 1u + q;
        };
        const unsigned w0 = width(cc.x), w1 = width(cc.y), w2 = width(cc.z), w3 = width(cc.w);
        const unsigned sum = w0 + w1 + w2 + w3;
        const unsigned incl = r4_wave_incl_scan(sum);
        const unsigned b0 = incl - sum, b1 = b0 + w0, b2 = b1 + w1, b3 = b2 + w2;
        auto entry = [&](unsigned base, unsigned wd, int bin) {
            const float ww = (float)wd - 0.125f;
            return make_float2(0.25f * ww, 2.f * (((float)base + 0.0625f) - (float)bin * ww));
        };
        tab[4 * lane + 0] = entry(b0, w0, 4 * lane + 0);
        tab[4 * lane + 1] = entry(b1, w1, 4 * lane + 1);
        tab[4 * lane + 2] = entry(b2, w2, 4 * lane + 2);
        tab[4 * lane + 3] = entry(b3, w3, 4 * lane + 3);
        // t can round up to RK_COARSE itself for x = hi: bin 256 is the last bucket of bin 255
        if (lane == 63) tab[RK_COARSE] = make_float2(0.f, 2.f * (float)(b3 + w3 - 1u) + 1.f);
    }
    __syncthreads();
    SORT_PROBE(3);
    // ---- 4. fine bucket of every key; the returning count atomic gives the key's arrival number a inside the bucket.
    //         b2 = 2 * bucket + half;  ba[r] = b2 | a << 14
    const float s8 = 8.f * s1;
    uint32_t ba[ITEMS];
#ifndef R4_G4
#define R4_G4 4
#endif
    constexpr int G4 = ITEMS < R4_G4 ? ITEMS : R4_G4;  // LDS operations of G4 keys in flight together (the wave has no other ILP)
#pragma unroll
    for (int g = 0; g < ITEMS; g += G4) {
        float t8[G4];
        float2 e2[G4];
        uint32_t b2[G4], old[G4], sh[G4];
#pragma unroll
        for (int j = 0; j < G4; j++) {
            if (g + j >= ITEMS) continue;  // compile-time: ITEMS need not be a multiple of four
            // (opaque: otherwise all 16 differences x - lo are formed before the barrier, next to the sample's, and stay
            // live beside x — 16 registers the 64-VGPR budget does not have)
            asm volatile("" : "+v"(x[g + j]));
            t8[j] = (x[g + j] - lo) * s8;  // = 8 t exactly
            // &tab[int(t)]: byte offset int(8 t) & ~7
            const r4_v2f tv = R4_LDS(const r4_v2f, TAB_B + ((uint32_t)(int)t8[j] & ~7u));
            e2[j] = make_float2(tv.x, tv.y);
        }
#pragma unroll
        for (int j = 0; j < G4; j++) {
            if (g + j >= ITEMS) continue;
            b2[j] = (uint32_t)__builtin_fmaf(t8[j], e2[j].x, e2[j].y);
            sh[j] = (b2[j] << 3) & 16u;  // 16 * (bucket & 1)
        }
#pragma unroll
        for (int j = 0; j < G4; j++) {
            if (g + j >= ITEMS) continue;
            uint32_t inc = 1u << sh[j];
            if (ragged(g + j)) inc = valid(g + j) ? inc : 0u;
            // counter word (bucket >> 1): byte offset b2 & ~3
            old[j] = __hip_atomic_fetch_add(&R4_LDS(uint32_t, CNT_B + (b2[j] & ~3u)), inc, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int j = 0; j < G4; j++)
            if (g + j < ITEMS) ba[g + j] = b2[j] | (__builtin_amdgcn_ubfe(old[j], sh[j], 16) << R4_BBITS);
        // the (b2, a) words packed here: otherwise the compiler keeps b2 and the atomic's result apart until step 6 and
        // spills both (64-VGPR budget)
#pragma unroll
        for (int j = 0; j < G4; j++)
            if (g + j < ITEMS) asm volatile("" : "+v"(ba[g + j]));
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    SORT_PROBE(4);
    // ---- 5. exclusive scan of the bucket counts -> start entries (in place, u16: start | big << 14 | long << 15).
    //         The bucket table is dead: the +inf slots behind the last key are written here.
    if (tid >= NT - R4_QWIN) slot[n + (tid - (NT - R4_QWIN))] = R4_INF;
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
        const unsigned incl = r4_wave_incl_scan(sum);
        if (lane == 63) red[w] = incl;
        __syncthreads();
        unsigned ex = incl - sum;
#pragma unroll
        for (int k = 0; k < NW; k++) ex += k < w ? red[k] : 0u;
        // start entry of a bucket from s on with cb keys, flags by carries instead of compares (2-cycle adds and ands):
        // bit 15 of (s & 1) + cb + (0x8000 - 9) is "(s & 1) + cb > 8", bit 14 of cb + (0x4000 - 49) is "cb > 48"
        // (cb <= 16384; an oversized bucket is long as well)
        uint32_t anybig = 0u;
        auto entry = [&](unsigned s, unsigned cb) {
            const uint32_t lng = ((s & 1u) + cb + (0x8000u - (unsigned)R4_WIN - 1u)) & R4_LONG;
            const uint32_t bigf = (cb + (0x4000u - (unsigned)RK_BIG - 1u)) & R4_BIGF;
            anybig |= bigf;
            return (s & R4_SMASK) | lng | bigf;
        };
        uint32_t ev[2 * PER];
        unsigned ex0 = ex;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const unsigned c0 = wv[q] & 0xffffu, c1v = wv[q] >> 16;
            ev[2 * q] = entry(ex, c0);
            ev[2 * q + 1] = entry(ex + c0, c1v);
            ex += c0 + c1v;
        }
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int i = tid * PER + q;
            if (i < NW2) cnt[i] = ev[2 * q] | (ev[2 * q + 1] << 16);
        }
        if (anybig) {  // rare: the oversized buckets register themselves (words past NW2 hold zero counts: never big)
#pragma unroll
            for (int q = 0; q < 2 * PER; q++) {
                const unsigned cb = (q & 1) ? (wv[q >> 1] >> 16) : (wv[q >> 1] & 0xffffu);
                if (cb > (unsigned)RK_BIG) {
                    const unsigned k = atomicAdd(&misc[0], 1u);
                    if (k < RK_MAXBIG) { misc[2 + 2 * k] = ex0; misc[3 + 2 * k] = cb; }
                }
                ex0 += cb;
            }
        }
        if (tid == 0) cnt[NW2] = (uint32_t)n & R4_SMASK;  // the entry behind the last bucket
    }
    __syncthreads();
    const unsigned nbig = misc[0];
    if (nbig > RK_MAXBIG || misc[22] != 0u) {  // (misc[22]: a non-finite or tiny key under a caller-given range, step 1)
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
            if (g + j < ITEMS)
                e[j] = R4_LDS(const unsigned short, CNT_B + (ba[g + j] & 0x3ffeu));
#pragma unroll
        for (int j = 0; j < G4; j++) {
            if (g + j >= ITEMS) continue;
            const uint32_t arr = ba[g + j] >> R4_BBITS;
            uint32_t pos = (e[j] & R4_SMASK) + arr;
            if (ragged(g + j)) pos = valid(g + j) ? pos : (uint32_t)(CAP + R4_QWIN);
            R4_LDS(float, SLOT_B + (pos << 2)) = x[g + j];
            ba[g + j] = e[j];
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    // ---- 6b. oversized buckets only come from exact ties: if all keys of such a bucket are equal (bit for bit) its
    //          ranks are the ranks of the pixel indices (bitmap + popcount prefix).  Anything else -> radix kernel.
    for (unsigned bi = 0; bi < nbig; bi++) {
        const uint32_t s = misc[2 + 2 * bi];
        const uint32_t k0 = slot[s];
        for (int i = tid; i < NWORDS; i += NT) bitmap[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            if (valid(r) && (ba[r] & (R4_LONG | R4_BIGF)) == (R4_LONG | R4_BIGF) && (ba[r] & R4_SMASK) == (s & R4_SMASK)) {
                if (__float_as_uint(x[r]) != k0) misc[1] = 1u;
                const uint32_t idx = (uint32_t)elem(r);
                atomicOr(&bitmap[idx >> 5], 1u << (idx & 31u));
            }
            asm volatile("" ::: "memory");  // one key at a time: this rare path must not set the kernel's register pressure
        }
        __syncthreads();
        if (misc[1]) {
            if (tid == 0) a.flags[col] = 1;
            return;
        }
        {
            const unsigned pcn = tid < NWORDS ? (unsigned)__popc(bitmap[tid]) : 0u;
            // block-wide exclusive scan on NT threads (block_excl_scan of sort_common.h assumes 1024)
            const unsigned incl = r4_wave_incl_scan(pcn);
            if (lane == 63) red[w] = incl;
            __syncthreads();
            unsigned ex = incl - pcn;
#pragma unroll
            for (int k = 0; k < NW; k++) ex += k < w ? red[k] : 0u;
            __syncthreads();
            if (tid < NWORDS) bpre[tid] = ex;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const bool mine = valid(r) && (ba[r] & (R4_LONG | R4_BIGF)) == (R4_LONG | R4_BIGF) && (ba[r] & R4_SMASK) == (s & R4_SMASK);
            const uint32_t idx = (uint32_t)(valid(r) ? elem(r) : 0);
            const uint32_t rk = s + bpre[idx >> 5] + (uint32_t)__popc(bitmap[idx >> 5] & ((1u << (idx & 31u)) - 1u));
            ba[r] = mine ? (R4_BIGF | (rk & R4_SMASK)) : ba[r];
            asm volatile("" ::: "memory");
        }
        __syncthreads();
    }
    SORT_PROBE(6);
    // ---- 7. ranks, owner side: every slot before the bucket holds a smaller key, every slot behind it a larger one, so
    //         rank = W0 + #{slot[W0 .. W0 + 7] < key} for the 8-byte aligned window W0 = start & ~1 whenever it covers the
    //         bucket.  ba[r] becomes the rank (or R4_TAG | queue entry).
    const float big = 1.7014118e38f;  // 2^127
    const float c16 = 0.0625f;
    // The 16-bit start entries are packed two to a register for this step: with the keys (16), the results (16, filling
    // up as the entries drain) and two windows in flight (16) the unpacked entries (16 more) do not fit the 64 VGPRs.
    uint32_t pe[(ITEMS + 1) / 2];
#pragma unroll
    for (int k = 0; k < (ITEMS + 1) / 2; k++) pe[k] = (2 * k + 1 < ITEMS) ? (ba[2 * k] & 0xffffu) | (ba[2 * k + 1] << 16) : ba[2 * k];
#pragma unroll
    for (int k = 0; k < (ITEMS + 1) / 2; k++) asm volatile("" : "+v"(pe[k]));
    // Common case in two 4-cycle instructions (v_bfe, one v_cmp): the start entry's LONG / BIG bits and the "exactly one
    // equal slot (myself)" test are merged into one word that is zero for a key ranked by its window.
    auto entry_of = [&](int r) { return (r & 1) ? pe[r >> 1] >> 16 : (pe[r >> 1] & 0xffffu); };
    auto rank_one = [&](int r, const r4_v2u& a0, const r4_v2u& a1, const r4_v2u& a2, const r4_v2u& a3) {
        const uint32_t e = entry_of(r);
        const uint32_t w0p = e & (R4_SMASK & ~1u);
        float acc = 524288.f;  // 2^19: ulp 1/16
        r4_window(acc, a0, a1, a2, a3, x[r], big, c16);
        const uint32_t bits = __float_as_uint(acc);
        uint32_t res = w0p + __builtin_amdgcn_ubfe(bits, 4, 4);
        // (bits - 1) & 15 != 0: another slot holds the same key (the pixels decide) -> the queue, like a bucket wider
        // than the window
        const uint32_t special = ((bits - 1u) & 15u) | (e & (R4_LONG | R4_BIGF));
        if (special != 0u) {
            if ((e & (R4_LONG | R4_BIGF)) == R4_BIGF) {  // final already
                res = e & R4_SMASK;
            } else if (!ragged(r) || valid(r)) {
                const uint32_t qi = atomicAdd(&misc[20], 1u);
                if (qi < (uint32_t)QCAP) {
                    qkey[qi] = __float_as_uint(x[r]);
                    qwin[qi] = e & (R4_SMASK & ~3u);
                    qpix[qi] = (uint32_t)elem(r);
                }
                res = R4_TAG | qi;
            }
        }
        ba[r] = res;
    };
    // four single 8-byte reads per window (volatile: merged into ds_read2_b64 they lose the 16-bit offset field that
    // holds the slot array's base, and the pairing is slower at random addresses — scripts/valu_lds_rate_probe.hip)
    auto wload = [&](int r, r4_v2u& a0, r4_v2u& a1, r4_v2u& a2, r4_v2u& a3) {
        const uint32_t off = SLOT_B + ((entry_of(r) & (R4_SMASK & ~1u)) << 2);
        a0 = R4_LDS(const volatile r4_v2u, off);
        a1 = R4_LDS(const volatile r4_v2u, off + 8u);
        a2 = R4_LDS(const volatile r4_v2u, off + 16u);
        a3 = R4_LDS(const volatile r4_v2u, off + 24u);
    };
    // One workgroup alone on a CU spends 66 cycles per wavefront and key here (LDS 29 + VALU 22 if they did not overlap at
    // all: profiles/r02_sort_rank4_one_vs_two_workgroups.log): with one window in flight per wavefront the step is bound by
    // the LDS round trip (16 windows in flight per workgroup), not by a pipe.  A second window in flight per wavefront
    // helps (9216 keys 549 -> 521 us, 8192 keys 484 -> 462 us, 16384 keys 904 -> 886 us) once the registers are there for
    // it: with unpacked start entries it spilled at 11 and more keys per thread and cost more than it hid (16 keys per
    // thread: 904 -> 917 us).  Three windows in flight are slower at every size.  Starting every other wavefront half a
    // round late, wave priorities and a first-generation stagger of the two workgroups of a CU were measured too: no gain.
#ifndef R4_PAIRMAX
#define R4_PAIRMAX 16
#endif
    if (ITEMS <= R4_PAIRMAX) {
#pragma unroll
    for (int g = 0; g + 1 < ITEMS; g += 2) {
        r4_v2u a0, a1, a2, a3, b0, b1, b2, b3;
        wload(g, a0, a1, a2, a3);
        wload(g + 1, b0, b1, b2, b3);
        rank_one(g, a0, a1, a2, a3);
        rank_one(g + 1, b0, b1, b2, b3);
        asm volatile("" ::: "memory");
    }
    if (ITEMS & 1) {
        r4_v2u a0, a1, a2, a3;
        wload(ITEMS - 1, a0, a1, a2, a3);
        rank_one(ITEMS - 1, a0, a1, a2, a3);
    }
    } else {
#pragma unroll
    for (int g = 0; g < ITEMS; g++) {
        r4_v2u a0, a1, a2, a3;
        wload(g, a0, a1, a2, a3);
        rank_one(g, a0, a1, a2, a3);
        asm volatile("" ::: "memory");
    }
    }
    asm volatile("" ::: "memory");
    SORT_PROBE(7);
    // the sorted source column on its way into registers while the queue is worked off (the key registers are dead)
    const bool stage = ns <= (unsigned)CAP;
    const bool svec = VEC && (ns % 4u == 0u) && ((reinterpret_cast<uintptr_t>(ssrt) & 15u) == 0u);
    // (native vector values, and a thread index the compiler cannot tie to the one of step 0: as a float4 array sv was
    // kept in scratch memory — the wavefront then WAITED for these loads right here to store them there — and the
    // element offsets of step 0 were kept alive, spilled, for the whole kernel)
    int tid9 = tid;
    asm volatile("" : "+v"(tid9));
    r4_v4f sv[Q > 0 ? Q : 1];
    float svt[T > 0 ? T : 1];
    if (MODE == SORT_MATCH && VEC && stage && svec) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const unsigned e0 = (unsigned)(q * NT + tid9) * 4u;
            sv[q] = *reinterpret_cast<const r4_v4f*>(ssrt + (e0 < ns ? e0 : 0u));
        }
#pragma unroll
        for (int r = 4 * Q; r < ITEMS; r++) {
            const unsigned e = (unsigned)(r * NT + tid9);
            svt[r - 4 * Q] = ssrt[e < ns ? e : 0u];
        }
    }
    // ---- 8. queued keys, one per thread, against a 52-slot window (a bucket has at most RK_BIG keys here): buckets wider
    //         than the 8-slot window and keys with an equal partner.  Equal keys are all in the queue (each of them saw the
    //         other): their order is the order of their (totalOrder key, pixel), settled among the (few) entries of the
    //         tie list.
    __syncthreads();
    const uint32_t qn = misc[20];
    if (qn > (uint32_t)QCAP) {  // tie-heavy column: radix kernel
        if (tid == 0) a.flags[col] = 1;
        return;
    }
    if (qn != 0u) {  // uniform: a column without a crossing bucket or an equal pair of keys skips the phase and its barriers
    for (uint32_t i = tid; i < qn; i += NT) {
        const float k = __uint_as_float(qkey[i]);
        const uint32_t w0p = qwin[i];
        const uint4* wp = reinterpret_cast<const uint4*>(slot + w0p);
        uint32_t lt = 0u, le = 0u;
        static_assert(R4_QWIN == 52, "thirteen 16-byte reads");
#pragma unroll 1
        for (int j = 0; j < 4; j++) {  // three reads in flight per trip (the sorted source column is live in registers)
            const uint4 x0 = wp[3 * j], x1 = wp[3 * j + 1], x2 = wp[3 * j + 2];
            r4_count4(lt, le, x0, k);
            r4_count4(lt, le, x1, k);
            r4_count4(lt, le, x2, k);
        }
        {
            const uint4 xl = wp[R4_QWIN / 4 - 1];
            r4_count4(lt, le, xl, k);
        }
        qres[i] = w0p + lt;
        if (le - lt > 1u) {  // equal fp32 keys, -0 / +0: compacted, so that nobody scans the whole queue
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
        const uint32_t i = tlist[t], kb = qkey[i], pix = qpix[i];
        const float kf = __uint_as_float(kb);
        const uint32_t kk = f2key(kf);
        uint32_t before = 0u;
        for (uint32_t u = 0; u < tn; u++) {
            const uint32_t j = tlist[u];
            const float jf = __uint_as_float(qkey[j]);
            const uint32_t jk = f2key(jf);
            // float-equal partners (this includes -0 / +0) ordered by (totalOrder key, pixel)
            before += (jf == kf && (jk < kk || (jk == kk && qpix[j] < pix))) ? 1u : 0u;
        }
        qres[i] += before;  // only this thread touches qres[i]
    }
    __syncthreads();
    }
    SORT_PROBE(8);
    if (MODE == SORT_EMIT) {
        // ---- 9E. sorted keys / pixel indices: every owner writes its key (then its pixel number) to slot[rank] — every
        //          slot has been read — and the column leaves the LDS in order with 16-byte stores
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            if ((ba[r] & R4_TAG) != 0u) ba[r] = qres[ba[r] & ~R4_TAG];  // rare: most wavefront rows branch over it
        }
        const size_t obase = (size_t)col * (size_t)n;
        const bool ovec = (n % 4 == 0) && a.out_vec;
        auto drain = [&](uint32_t* dst) {
            __syncthreads();
            if (ovec) {
                for (int e = tid9 * 4; e < n; e += NT * 4)
                    *reinterpret_cast<uint4*>(dst + obase + e) = *reinterpret_cast<const uint4*>(slot + e);
            } else {
                for (int e = tid9; e < n; e += NT) dst[obase + e] = slot[e];
            }
        };
        if (a.out_keys) {
#pragma unroll
            for (int r = 0; r < ITEMS; r++)
                if (valid(r)) R4_LDS(float, SLOT_B + (ba[r] << 2)) = x[r];
            drain(reinterpret_cast<uint32_t*>(a.out_keys));
        }
        if (a.out_idx) {
            if (a.out_keys) __syncthreads();  // the keys have left the slot array
#pragma unroll
            for (int r = 0; r < ITEMS; r++)
                if (valid(r)) R4_LDS(uint32_t, SLOT_B + (ba[r] << 2)) = (uint32_t)elem(r);
            drain(a.out_idx);
        }
        if (a.only_flagged && tid == 0) a.flags[col] = 0;
        SORT_PROBE(10);
        return;
    }
    // ---- 9. out[pixel] = sorted_source[q(rank)]: the source column is staged in the slot array (every slot has been
    //         read), each owner picks its values and leaves with 16-byte stores
    float* val = reinterpret_cast<float*>(slot);
    if (stage) {
        if (VEC && svec) {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const unsigned e0 = (unsigned)(q * NT + tid9) * 4u;
                if (e0 < ns) R4_LDS(r4_v4f, SLOT_B + (e0 << 2)) = sv[q];
            }
#pragma unroll
            for (int r = 4 * Q; r < ITEMS; r++) {
                const unsigned e = (unsigned)(r * NT + tid9);
                if (e < ns) R4_LDS(float, SLOT_B + (e << 2)) = svt[r - 4 * Q];
            }
        } else {
            for (unsigned e = tid; e < ns; e += NT) val[e] = ssrt[e];
        }
    }
    if (qn != 0u) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            if ((ba[r] & R4_TAG) != 0u) ba[r] = qres[ba[r] & ~R4_TAG];  // rare: most wavefront rows branch over it
        }
    }
    if (stage) __syncthreads();  // the staged column is complete (without staging nothing was written since the last barrier)
    SORT_PROBE(9);
    // the uniform decisions — staged column (LDS) or global gather, identity quantile when ns == n — are branches around
    // whole loops: selected per key they turn the LDS read into a flat load with a 64-bit address select
    float v[ITEMS];
    auto pick = [&](auto staged, auto same) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const unsigned rk = ragged(r) ? (valid(r) ? ba[r] : 0u) : ba[r];
            unsigned qi = rk;
            if (!decltype(same)::value) {
                const double aa = (double)(2u * rk + 1u) * (double)ns;
                qi = (unsigned)__builtin_fma(aa, a.inv_2nt, 7.450580596923828e-09);  // quantile_index (sort_common.h)
            }
            if (decltype(staged)::value) v[r] = R4_LDS(const float, SLOT_B + (qi << 2));
            else v[r] = ssrt[qi];
            if ((r & 3) == 3) asm volatile("" ::: "memory");
        }
    };
    if (stage) {
        if (ns == (unsigned)n) pick(std::true_type{}, std::true_type{});
        else pick(std::true_type{}, std::false_type{});
    } else {
        if (ns == (unsigned)n) pick(std::false_type{}, std::true_type{});
        else pick(std::false_type{}, std::false_type{});
    }
    if (VEC && a.out_vec) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int e0 = (q * NT + tid9) * 4;
            if (!ragged(4 * q) || e0 < n)
                *reinterpret_cast<float4*>(o + e0) =
                    make_float4(v[4 * q], v[(4 * q + 1) % ITEMS], v[(4 * q + 2) % ITEMS], v[(4 * q + 3) % ITEMS]);
        }
#pragma unroll
        for (int r = 4 * Q; r < ITEMS; r++)
            if (valid(r)) o[r * NT + tid9] = v[r];
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; r++)
            if (valid(r)) o[r < 4 * Q ? ((r >> 2) * NT + tid9) * 4 + (r & 3) : r * NT + tid9] = v[r];
    }
    if (a.only_flagged && tid == 0) a.flags[col] = 0;
    SORT_PROBE(10);
}

template <typename KernT>
static int launch_one4(KernT kern, DeviceOnce& once, size_t lds, const SortArgs& a, int ncols, int nt, hipStream_t st) {
#ifdef OPTEX_SORT_PROBE
    if (const char* e = getenv("OPTEX_SORT_LDS_PAD")) lds += (size_t)atol(e);  // fewer workgroups per CU (scripts/sort_rank_probe.hip)
#endif
    bool& attr = *once.slot();
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("sort: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return OPTEX_E_LAUNCH; }
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(ncols), dim3(nt), lds, st, a);
    return OPTEX_OK;
}

template <int ITEMS, int NT, int MODE>
static int launch_rank4_items(SortArgs a, int ncols, hipStream_t st) {
    // 16-byte loads for the first 4 * (ITEMS / 4) registers; without scalar rows behind them the column must end on a quad
    constexpr bool CANVEC = ITEMS >= 4;
    const bool in_vec = CANVEC && (ITEMS % 4 != 0 || a.n % 4 == 0) && a.ld % 4 == 0 && a.ss % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(a.keys) & 15u) == 0;
    if (MODE == SORT_MATCH)
        a.out_vec = (a.ldo % 4 == 0 && a.oss % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0) ? 1 : 0;
    else  // contiguous [column, n] outputs: 16-byte stores when the columns start on 16-byte boundaries
        a.out_vec = (a.n % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out_keys) & 15u) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.out_idx) & 15u) == 0) ? 1 : 0;
    const size_t lds = R4<ITEMS, NT>::LDS;
    const bool full = a.n == (long)ITEMS * NT;
    int rc;
    if (in_vec && full) {
        static DeviceOnce once;
        rc = launch_one4(rank_match4_kernel<ITEMS, CANVEC, NT, true, MODE>, once, lds, a, ncols, NT, st);
    } else if (in_vec) {
        static DeviceOnce once;
        rc = launch_one4(rank_match4_kernel<ITEMS, CANVEC, NT, false, MODE>, once, lds, a, ncols, NT, st);
    } else if (full) {
        static DeviceOnce once;
        rc = launch_one4(rank_match4_kernel<ITEMS, false, NT, true, MODE>, once, lds, a, ncols, NT, st);
    } else {
        static DeviceOnce once;
        rc = launch_one4(rank_match4_kernel<ITEMS, false, NT, false, MODE>, once, lds, a, ncols, NT, st);
    }
    if (rc) return rc;
    return check_launch("rank_match4_kernel");
}

template <int MODE>
static int launch_rank4_mode(const SortArgs& a0, int ncols, hipStream_t st) {
    SortArgs a = a0;
    const long n = a.n;
    // Workgroup size and keys per thread, always exactly ceil(n / threads) keys (the kernel relies on it: only the last
    // register row can be ragged).  Measured at [64, 256, n] (profiles/r02_sort_rank4_shapes.md): a column wants 6 .. 10
    // keys per thread — shorter register chains, no spills — as long as that leaves the CU its 32 wavefronts: 4096 keys
    // 227 us on 512 x 8 against 287 us on 256 x 16 and 323 us on 1024 x 4; 8192 keys 486 us on 1024 x 8 against 580 us on
    // 512 x 16; 5120 keys 287 us on 512 x 10 against 381 us on 1024 x 5.  Above 10240 keys a workgroup cannot have more
    // threads, and the keys per thread grow to 16.
    if (n <= 2048) return launch_rank4_items<2, SORT_NT, MODE>(a, ncols, st);
    if (n <= 2560) {
        if (n <= 9 * 256) return launch_rank4_items<9, 256, MODE>(a, ncols, st);
        return launch_rank4_items<10, 256, MODE>(a, ncols, st);
    }
    if (n <= 5120) {
        switch ((int)((n + 511) / 512)) {
            case 6: return launch_rank4_items<6, 512, MODE>(a, ncols, st);
            case 7: return launch_rank4_items<7, 512, MODE>(a, ncols, st);
            case 8: return launch_rank4_items<8, 512, MODE>(a, ncols, st);
            case 9: return launch_rank4_items<9, 512, MODE>(a, ncols, st);
            default: return launch_rank4_items<10, 512, MODE>(a, ncols, st);
        }
    }
    // 6400 keys (a pass size of the 512^2 schedule) fill 640 threads x 10 keys exactly, three workgroups to a CU.  (Tried
    // and slower: 9216 keys on 576 x 16, three workgroups of nine wavefronts to a CU, 822 us against 573 us on 1024 x 9;
    // 12544 keys on 896 x 14, 936 us against 826 us on 1024 x 13.)
    if (n == 10 * 640) return launch_rank4_items<10, 640, MODE>(a, ncols, st);
    switch ((int)((n + 1023) / 1024)) {
        case 6: return launch_rank4_items<6, 1024, MODE>(a, ncols, st);
        case 7: return launch_rank4_items<7, 1024, MODE>(a, ncols, st);
        case 8: return launch_rank4_items<8, 1024, MODE>(a, ncols, st);
        case 9: return launch_rank4_items<9, 1024, MODE>(a, ncols, st);
        case 10: return launch_rank4_items<10, 1024, MODE>(a, ncols, st);
        case 11: return launch_rank4_items<11, 1024, MODE>(a, ncols, st);
        case 12: return launch_rank4_items<12, 1024, MODE>(a, ncols, st);
        case 13: return launch_rank4_items<13, 1024, MODE>(a, ncols, st);
        case 14: return launch_rank4_items<14, 1024, MODE>(a, ncols, st);
        case 15: return launch_rank4_items<15, 1024, MODE>(a, ncols, st);
        default: return launch_rank4_items<16, 1024, MODE>(a, ncols, st);
    }
}

int launch_rank4(int mode, const SortArgs& a, int ncols, hipStream_t st) {
    return mode == SORT_MATCH ? launch_rank4_mode<SORT_MATCH>(a, ncols, st) : launch_rank4_mode<SORT_EMIT>(a, ncols, st);
}

}  // namespace optex
