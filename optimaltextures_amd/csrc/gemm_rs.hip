// gemm_rs.hip — the rotation GEMM of the hot loop as an R-STATIONARY, LDS-free, barrier-free kernel.
//
//   OUT[s][m][i] = sum_k At[s][k][m] * B[s][k][i]          (optex.py:170,171,175; channel-major in and out)
//
// The left operand of every rotation is a C x C matrix (C <= 256) shared by all pixels of a segment; the right operand, the
// feature map, streams through once.  gemm16_cm_kernel (gemm.hip) stages both through LDS in 16-deep K chunks: one barrier
// per chunk with two waves per SIMD, 0.73-0.76 of the fp32-MFMA peak.  This kernel removes the staging instead of tuning it
// — and lands on the same TFLOP/s: both kernels keep the matrix pipe busy 0.73-0.77 of the cycles and run at the clock the
// power management allows for the data (2.39 GHz on zeros, 2.12 GHz on dense operands; DESIGN.md 4.1,
// profiles/r03_gemm_power_dvfs.md):
//
//   * one workgroup of four wavefronts per CU (one wave per SIMD, the whole 512-register file), PERSISTENT over a contiguous
//     range of pixel tiles;
//   * wave w owns the output rows [16 MT w, 16 MT (w + 1)) and keeps its slice of the matrix — MT x KS fragments of
//     v_mfma_f32_16x16x4_f32's A operand, up to 256 registers — for the whole launch (a workgroup never crosses matrices:
//     with per-segment matrices the grid's y dimension selects the segment, so there is no reload);
//   * the feature map goes from HBM straight into the MFMA's B-operand registers: lane (i = lane & 15, q = lane >> 4) loads
//     the 16 bytes B[4 ks + q][p0 + 4 i .. + 3] of k-step ks — component j of that float4 IS the fragment of pixel sub-tile j
//     (sub-tile j = pixels p0 + 4 i + j: which 16 pixels form a 16-column MFMA tile is free to choose), so one
//     global_load_dwordx4 feeds 4 MT MFMAs and no feature-map byte passes through LDS.  A ring of DEPTH k-steps is in flight per wave,
//     across tile boundaries; the four waves of a CU read the same lines, HBM sees them once;
//   * fed (B fragment, A fragment) the MFMA returns the transposed block: lane (m = lane & 15, g = lane >> 4) ends up with
//     the 16 CONSECUTIVE pixels p0 + 16 g .. + 15 of channel m of each of its MT row tiles — four 16-byte stores each;
//   * no barrier in the loop: the waves of a workgroup never exchange data (they share the staging of the matrix in the
//     prologue, round 5).
//
// Round 5 (profiles/r05_gemm_timeline.md, s_memtime stamps per 4 k-steps):
//   * THE MATRIX LIVES IN ACCUMULATION REGISTERS.  The register allocator used to keep ~3/4 of the 256 A fragments in AGPRs as
//     "spill slots" and copy each one back with v_accvgpr_read_b32 (+ s_nop 1: VALU write -> MFMA read) in front of the four
//     MFMAs that use it: k-steps whose fragments sat in VGPRs ran 2180 cycles per 64 MFMAs (0.94 of the pipe), the others
//     2490-2620 (0.78-0.82) — the "13 points the loads cost" of round 3's NOLOAD probe were these copies (without the B ring
//     the fragments fitted the VGPRs).  Every fragment is now DEFINED in an AGPR (v_accvgpr_write_b32 in the prologue): its
//     live range has the AGPR class and the MFMA reads it from there (gfx950's MFMA takes SrcA / SrcB from either file).
//   * the matrix comes in through LDS (load_matrix below): the workgroup fetches 32-row blocks together and every wave picks
//     its fragments with one 16-byte LDS read per k-step — the wave's row tile t is the rows mw + MT i + t (i = lane & 15;
//     which 16 rows form an MFMA row tile is as free to choose as the pixel sub-tiles), so a lane's fragments of a k-step
//     are neighbours.  Fetched per wave straight from the L2 the prologue took 29-70 thousand cycles (14-30 us: every CU of
//     the chip asks for the same 256 KB in the same order), a third of a launch at 8 textures per step.
//
// Numerics: every output element is the k-ordered fmaf chain of the other GEMM kernels and of the oracle (a * b commutes
// exactly; k-steps ascend, four k per step in MFMA order) — bit-identical.  Rows k >= K enter as exact zeros on BOTH
// operands (a zero A fragment alone would turn a non-finite pad read into NaN).
// Optional per-row statistics of the output (GemmArgs::rowstat, 1 = min / max, 2 = sum) are taken from the accumulators:
// one partial per 64-pixel tile, [n_seg][n / 64][M].
#include "gemm_args.h"
#include <type_traits>
#include "timeline.h"

namespace optex {

typedef float rs_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char* rs_gptr;      // global address space kept explicit: a pointer rebuilt
typedef const __attribute__((address_space(1))) float* rs_gfptr;    // from integers would otherwise be loaded with flat_load

// a uniform global pointer pinned to an SGPR pair (the 64-bit tile arithmetic is uniform, but not always provably so)
__device__ __forceinline__ rs_gptr rs_uniform(const void* p) {
    const uintptr_t u = reinterpret_cast<uintptr_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<rs_gptr>(((uintptr_t)hi << 32) | (uintptr_t)lo);
}

// the value of lane (l ^ 16) / (l ^ 32), on the VALU: v_permlane16_swap / v_permlane32_swap of a register with itself (odd rows of
// the first operand <-> even rows of the second; upper half of the first <-> lower half of the second) leave the partner's value in
// one of the two results on every lane.  __shfl_xor is a ds_bpermute: an LDS round trip behind s_waitcnt lgkmcnt(0) — inside the
// k-loop of the double-buffered kernels each of them stalled the matrix pipe (round 5: the row-statistics variant ran 39.1
// thousand cycles per tile against 36.4 without statistics)
// (both results are combined, so the code does not depend on which of the two is the lane's own value: min, max and + commute)
__device__ __forceinline__ void rs_pair16(float v, float& p, float& q) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    p = __uint_as_float(r[0]);   // (row0, row0, row2, row2)
    q = __uint_as_float(r[1]);   // (row1, row1, row3, row3)
}
__device__ __forceinline__ void rs_pair32(float v, float& p, float& q) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    p = __uint_as_float(r[0]);   // (lower half, lower half)
    q = __uint_as_float(r[1]);   // (upper half, upper half)
}

constexpr int RS_BN = 64;     // pixels per tile (4 sub-tiles of 16)
#ifndef RS_DEPTH_VALUE
#define RS_DEPTH_VALUE 16
#endif
constexpr int RS_DEPTH = RS_DEPTH_VALUE;   // k-steps of B in flight per wave (scripts/gemm_rs_probe.hip builds variants)
#ifndef RS_DEPTH_DB_VALUE
#define RS_DEPTH_DB_VALUE 8
#endif
constexpr int RS_DEPTH_DB = RS_DEPTH_DB_VALUE;
#ifndef RS_FLUSH0
#define RS_FLUSH0 8    // first k-step behind which a row tile of the previous pixel tile is stored (DB kernels) ...
#define RS_FLUSHD 8    // ... and the distance to the next one
#endif
#ifndef RS_DB
#define RS_DB 1        // 0: probe build without the double-buffered accumulators
#endif
#ifndef RS_DB_RAGGED
#define RS_DB_RAGGED 0
#endif
#ifndef RS_SPLIT_LOAD
#define RS_SPLIT_LOAD 0   // in-loop refill as two 8-byte halves: measured SLOWER (111.6 against 115.1 TFLOP/s, profiles/r05_gemm_probes.md)
#endif
constexpr int RS_RAG = 16;    // the last RS_RAG k-steps of an instantiation may lie (partly) beyond K

extern int gemm_rs_spare_cus;   // CUs left out of the persistent grid (optex_gemm_spare_cus)

struct RsArgs {
    GemmArgs g;
    int tiles_n;          // pixel tiles per segment (n / 64)
    int segs_per_group;   // segments sharing one matrix: n_seg (at_seg_stride == 0) or 1; gridDim.y = n_seg / segs_per_group
};

// MT: 16-row tiles per wave (rows per workgroup = 64 MT: 128 / 192 / 256);  KS: k-steps of 4 (32 / 48 / 64), 4 (KS - RS_RAG) <= K <= 4 KS: the last RS_RAG
// k-steps test their rows, and a k-step entirely beyond K is branched over (uniform);  EXTRA: bias (badd) and content
// blend in the epilogue (1), and the operand centring `B[k][i] - bsub[k]` of the linear modes' apply step as well (2:
// KS more registers, one subtraction per fragment component, the same single rounding as the other kernels)
// WPE: wavefronts per SIMD (workgroup = 256 WPE threads, 64 MT WPE rows).  The library runs WPE = 1; WPE = 2 with MT = 2 is the
// probe variant that lets one wave's epilogue and load waits hide under the other's MFMAs (scripts/Makefile, -DRS_WPE=2).
#ifndef RS_WPE
#define RS_WPE 1
#endif
// KFULL: K == 4 KS exactly (the launcher checks): no k-step is ragged, so the masks, selects and uniform branches of the last
// RS_RAG k-steps disappear and the whole k-loop is one basic block (C = 256 and C = 128, the un-projected hot shapes)
// DB: double-buffered accumulators — the previous tile's results leave during this tile's k-loop (run_tile / epilogue_row)
template <int MT, int KS, int ROWSTAT, int EXTRA, int WPE = 1, bool KFULL = false, bool DB = false>
__global__ __attribute__((amdgpu_flat_work_group_size(256 * WPE, 256 * WPE), amdgpu_waves_per_eu(WPE, WPE))) void gemm_rs_kernel(RsArgs ra) {
    // B ring: DEPTH k-steps; the double-buffered kernels take RS_DEPTH_DB (their second accumulator set needs the registers:
    // with 16 k-steps the hot loop spilled; 7 k-steps = 3800 cycles of lead against ~900 cycles of HBM latency)
    constexpr int DEPTH = DB ? RS_DEPTH_DB : RS_DEPTH;
    static_assert(KS % DEPTH == 0, "the B ring keeps its phase across tiles");
    constexpr int RAG = KFULL ? 0 : RS_RAG;   // the last RAG k-steps may lie (partly) beyond K
    const GemmArgs& a = ra.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int mw = wave * 16 * MT;  // 16 MT rows per wave; row tile t = the rows mw + MT i + t (i = lane & 15): a lane's MT rows are neighbours
    const int M = a.M, K = a.K;

    // A workgroup keeps ONE matrix: blockIdx.y selects a group of `segs_per_group` segments that share it (all of them
    // when at_seg_stride == 0, one otherwise), blockIdx.x a contiguous, balanced range of the group's tiles (segment-major).
    const long T = (long)ra.tiles_n * ra.segs_per_group;
    const long t0 = (long)blockIdx.y * T;
    const long tb = t0 + T * (long)blockIdx.x / (long)gridDim.x, te = t0 + T * ((long)blockIdx.x + 1) / (long)gridDim.x;
    if (tb >= te) return;
#ifdef OPTEX_TIMELINE
    // slab: [0] = XCC id | tiles << 8, [1] = real time at entry, [2] = entry, [3] = matrix + ring loads issued, then per tile
    // 18 stamps (tile start, after k-steps 3, 7, ..., 63, stores issued), at the end the real time again
    tl_ptr tl = tl_begin((blockIdx.y * gridDim.x + blockIdx.x) * 4 * WPE + wave);
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tl_word(tl, (unsigned long long)(xcc & 0xff) | ((unsigned long long)(te - tb) << 8));
    }
    tl_stamp_real(tl);
    tl_stamp(tl);
#endif

    // the wave's slice of the matrix: fragment (t, ks) = At[4 ks + q][mw + MT i + t] (row tile t = rows mw + MT i + t), kept in
    // ACCUMULATION registers for the whole launch (see the header); rows / columns beyond K / M are exact zeros.
    float af[MT][KS];
    auto to_agpr = [](float v) {
        float r;
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(v));
        return r;
    };
    // The matrix reaches the registers THROUGH LDS, 8 k-steps (32 rows) at a time.  Straight from the L2 it took 29-70 thousand
    // cycles (14-30 us, profiles/r05_gemm_timeline.md): every wavefront of the chip asks for the same 256 KB in the same order,
    // so at any moment all 256 CUs queue at the one or two L2 channels that hold the current rows.  Here the workgroup fetches
    // a 32-row block together — whole 4 KB spans per instruction, the eight pieces of a block in an order rotated by the
    // workgroup's number, two blocks ahead of the one being unpacked — zero-fills what lies beyond K / M on the way, and every
    // wave then picks its fragments with one 16-byte LDS read per k-step (MT = 4; MT scalar reads otherwise).
    constexpr int NTH = 256 * WPE, LROW = 260, PH = KS / 8, NG = 2048 / NTH;   // threads, staged row stride (floats), phases, 16-byte pieces per thread and phase
    __shared__ __attribute__((aligned(16))) float As[32 * LROW];
    auto load_matrix = [&](int seg) {
        const rs_gfptr At = reinterpret_cast<rs_gfptr>(rs_uniform(a.At + (size_t)seg * a.at_ss));
        const int rot = (int)(blockIdx.x + blockIdx.y) & (NG - 1);
        // VECA (one uniform decision for the whole staging — taken per piece it put a branch and an s_waitcnt vmcnt(0) behind
        // every single load: 64 dependent L2 round trips, 25-30 us): 16-byte global loads, a piece is all inside M or all outside
        auto stage = [&](auto veca) {
            constexpr bool VECA = decltype(veca)::value;
            rs_f4 g[2][NG];
            // piece i of block p: row 32 p + (idx >> 6), columns 4 (idx & 63) .. + 3; inside K x M?
            auto where = [&](int p, int i, int& row, int& c0, int& k) {
                const int idx = (int)threadIdx.x + NTH * ((i + rot) & (NG - 1));
                row = idx >> 6;
                c0 = (idx & 63) * 4;
                k = 32 * p + row;
            };
            // the loads of a block go out back to back, RAW (clamped addresses); what lies beyond K / M is zeroed when the block
            // is written to LDS — masked right behind its load, every piece waited for its own round trip (s_waitcnt vmcnt(0)
            // 64 times over: the 25-30 us the first versions of this prologue took)
            auto gload = [&](int p, rs_f4 (&dst)[NG]) {
#pragma unroll
                for (int i = 0; i < NG; i++) {
                    int row, c0, k;
                    where(p, i, row, c0, k);
                    if (VECA) {
                        const unsigned ok = (k < K && c0 < M) ? 0xffffffffu : 0u;
                        const unsigned off = ((unsigned)k * (unsigned)a.lda + (unsigned)c0) & ok;
                        dst[i] = *reinterpret_cast<const __attribute__((address_space(1))) rs_f4*>(At + off);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const unsigned ok = (k < K && c0 + e < M) ? 0xffffffffu : 0u;
                            dst[i][e] = At[((unsigned)k * (unsigned)a.lda + (unsigned)(c0 + e)) & ok];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto lstore = [&](int p, const rs_f4 (&src)[NG]) {
#pragma unroll
                for (int i = 0; i < NG; i++) {
                    int row, c0, k;
                    where(p, i, row, c0, k);
                    rs_f4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const unsigned ok = (k < K && c0 + (VECA ? 0 : e) < M) ? 0xffffffffu : 0u;
                        v[e] = __uint_as_float(__float_as_uint(src[i][e]) & ok);
                    }
                    *reinterpret_cast<rs_f4*>(&As[row * LROW + c0]) = v;
                }
            };
            gload(0, g[0]);
            if (PH > 1) gload(1, g[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < PH; p++) {
                lstore(p, g[p & 1]);
                __syncthreads();
                if (p + 2 < PH) gload(p + 2, g[p & 1]);
                __builtin_amdgcn_sched_barrier(0);   // the loads of two blocks ahead are on their way before this block is unpacked
                const float* fr = &As[kq * LROW + mw + MT * l15];
                rs_f4 fv[8];   // the block's eight fragment reads in flight together, then into the accumulation registers
#pragma unroll
                for (int ksl = 0; ksl < 8; ksl++) {
                    if (MT == 4) {
                        fv[ksl] = *reinterpret_cast<const rs_f4*>(fr + 4 * ksl * LROW);
                    } else {
#pragma unroll
                        for (int t = 0; t < MT; t++) fv[ksl][t] = fr[4 * ksl * LROW + t];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ksl = 0; ksl < 8; ksl++)
#pragma unroll
                    for (int t = 0; t < MT; t++) af[t][8 * p + ksl] = to_agpr(fv[ksl][t]);
                __syncthreads();
            }
        };
        if (a.a_vec && (M & 3) == 0) stage(std::true_type{});
        else stage(std::false_type{});
    };

    // B addressing: ONE uniform running byte pointer (advanced by four rows per k-step and re-based at tile boundaries)
    // + ONE 32-bit lane offset (row q of the k-step, pixels 4 i .. 4 i + 3).
    const unsigned lane_off = ((unsigned)kq * (unsigned)a.ldb + 4u * (unsigned)l15) * 4u;
    // the one k-step that K may cut (rows 4 ks + q >= K): those lanes read row 4 ks (in bounds) and zero the value
    const bool ok_p = (K & ~3) + kq < K;
    const unsigned lane_off_p = ok_p ? lane_off : 4u * (unsigned)l15 * 4u;
    const size_t step_bytes = (size_t)a.ldb * 16u;
    // (seg, pt) of the current and of the next tile are carried along as counters: a 64-bit division per tile is ~350
    // scalar instructions during which the matrix pipe idles
    auto tile_base = [&](int seg, int pt) {  // uniform
        return rs_uniform(a.B + (size_t)seg * a.b_ss + (size_t)pt * RS_BN);
    };
    int seg = (int)(tb / ra.tiles_n), pt = (int)(tb - (long)seg * ra.tiles_n);
    rs_gptr pk = tile_base(seg, pt);
    // load k-step ks (compile-time position) at the running pointer into `dst`, advance the pointer
    auto load_next = [&](int ks, rs_f4& dst) {
        if (ks >= KS - RAG) {
            if (4 * ks < K) {  // uniform; a k-step entirely beyond K is neither loaded nor multiplied
                const bool partial = 4 * ks + 4 > K;
                // (the rows beyond K are zeroed where the k-step is CONSUMED: masking here would wait for the load at once)
                dst = *reinterpret_cast<const __attribute__((address_space(1))) rs_f4*>(pk + (partial ? lane_off_p : lane_off));
            }
        } else {
            dst = *reinterpret_cast<const __attribute__((address_space(1))) rs_f4*>(pk + lane_off);
        }
        pk += step_bytes;
    };

    // the same k-step in two 8-byte halves (-DRS_SPLIT_LOAD=1 probe build, KFULL loop only).  With its loads a k-step takes ~30
    // cycles more than without (round 5 timeline: 2196 cycles per 64 MFMAs against 2076, wherever the load is placed);
    // the guess that a 1 KB wave-instruction overruns its 32-cycle MFMA shadow and two 512-byte pieces would not was WRONG:
    // two instructions cost more than one (2244 cycles per 64 MFMAs) — the cost is per VMEM instruction, not per byte
    typedef float rs_f2 __attribute__((ext_vector_type(2)));
    auto load_half = [&](rs_f4& dst, int h) {
        const rs_f2 v = *reinterpret_cast<const __attribute__((address_space(1))) rs_f2*>(pk + lane_off + 8u * (unsigned)h);
        dst[2 * h] = v[0];
        dst[2 * h + 1] = v[1];
        if (h == 1) pk += step_bytes;
    };

    // EXTRA == 2: the centring value bsub[4 ks + q] of every k-step travels through the ring beside its fragment (one more
    // dword load per k-step, L1 / L2 resident; KS registers of it kept for the whole launch spill).  The launcher keeps a
    // workgroup inside one segment when bsub varies with the segment.
    const rs_gfptr sub = EXTRA == 2 ? reinterpret_cast<rs_gfptr>(rs_uniform(a.bsub + (size_t)seg * a.bsub_ss)) : nullptr;
    float bsr[EXTRA == 2 ? DEPTH : 1];
    auto load_sub = [&](int ks, float& dst) {
        if (EXTRA == 2) {
            const int k = 4 * ks + kq;
            dst = sub[(ks < KS - RAG || k < K) ? k : 0];
        }
    };

    rs_f4 br[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) br[d] = rs_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) {  // k-steps 0 .. DEPTH - 2; the loop requests k-step ks - 1 + DEPTH at k-step ks
        load_next(d, br[d]);
        load_sub(d, bsr[EXTRA == 2 ? d : 0]);
    }
    // (the matrix behind the first B fragments: their HBM latency passes while the matrix comes from the L2)
    __builtin_amdgcn_sched_barrier(0);
    load_matrix(seg);
    TL_STAMP(tl);

    // One output row tile (t) of a finished pixel tile (eseg, ept): the lane's 16 consecutive pixels of channel m as four 16-byte
    // stores, bias / blend (EXTRA), row statistics (ROWSTAT).  acc[t][j][r] = OUT[m = mw + MT l15 + t][pixel p0 + 16 kq + 4 r + j]
    auto epilogue_row = [&](rs_f4 (&acc)[MT][4], int t, int eseg, int ept) {
        float* __restrict__ Op = a.O + (size_t)eseg * a.o_ss + (size_t)ept * RS_BN + 16 * kq;
        const float* __restrict__ Cp = (EXTRA && a.content) ? a.content + (size_t)eseg * a.o_ss + (size_t)ept * RS_BN + 16 * kq : nullptr;
        const float* __restrict__ badd = (EXTRA && a.badd) ? a.badd + (size_t)eseg * a.badd_ss : nullptr;
        const int seg = eseg, pt = ept;
        (void)seg; (void)pt;
            const int m = mw + MT * l15 + t;
            const bool ok = m < M;
            const size_t row = (size_t)(ok ? m : 0) * a.ldo;
            rs_f4 v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = rs_f4{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
            if (EXTRA) {
                if (badd) {
                    const float bias = badd[ok ? m : 0];
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = v[r] + bias;
                }
                if (Cp) {  // optex.py:115-117, same arithmetic in the same order as the other kernels: v + strength * (content - v)
                    rs_f4 c[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) c[r] = *reinterpret_cast<const rs_f4*>(Cp + row + 4 * r);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const rs_f4 d = c[r] - v[r];
                        const rs_f4 sd = d * a.strength;
                        v[r] = v[r] + sd;
                    }
                }
            }
            if (ok) {
#pragma unroll
                for (int r = 0; r < 4; r++) *reinterpret_cast<rs_f4*>(Op + row + 4 * r) = v[r];
            }
            if (ROWSTAT != 0) {
                // statistics of the plain product (this path runs without bias / blend): the lane's 16 pixels, then the
                // four lane groups holding the other pixels of the same channel
                const size_t pidx = ((size_t)seg * ra.tiles_n + pt) * (size_t)M + (size_t)m;
                if (ROWSTAT == 1) {
                    // Round 6: v_min3 / v_max3 written out (fminf / fmaxf on values that crossed a permlane swap brought a
                    // canonicalising v_max x, x each), and BOTH statistics through ONE pair of swaps: rows 0 / 2 of the wave
                    // carry the minimum, rows 1 / 3 the NEGATED maximum, so a single v_min folds either — 25 instead of 56
                    // VALU instructions per row tile of a kernel that has one wavefront per SIMD (nothing hides them).
                    float mn = v[0][0], mx = v[0][0];
                    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn) : "v"(v[0][1]), "v"(v[0][2]));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(v[0][1]), "v"(v[0][2]));
                    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn) : "v"(v[0][3]), "v"(v[1][0]));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(v[0][3]), "v"(v[1][0]));
#pragma unroll
                    for (int r = 1; r < 4; r++) {
                        if (r > 1) {
                            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn) : "v"(v[r][0]), "v"(v[r][1]));
                            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(v[r][0]), "v"(v[r][1]));
                            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn) : "v"(v[r][2]), "v"(v[r][3]));
                            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(v[r][2]), "v"(v[r][3]));
                        } else {
                            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn) : "v"(v[1][1]), "v"(v[1][2]));
                            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(v[1][1]), "v"(v[1][2]));
                            asm("v_min_f32 %0, %0, %1" : "+v"(mn) : "v"(v[1][3]));
                            asm("v_max_f32 %0, %0, %1" : "+v"(mx) : "v"(v[1][3]));
                        }
                    }
                    const float nx = -mx;
                    // (row0: mn, row1: -mx, row2: mn, row3: -mx) against (row1: mn ... ) of the neighbouring rows, then the halves
                    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mn), __float_as_uint(nx), false, false);
                    float m = __uint_as_float(s16[0]);
                    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(__uint_as_float(s16[1])));
                    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                    m = __uint_as_float(s32[0]);
                    asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(__uint_as_float(s32[1])));
                    // rows 0 / 2 now hold min over the 64 pixels, rows 1 / 3 minus the max: lane group kq = 0 stores one, kq = 1 the other
                    if (kq < 2 && ok) {
                        float* dst = kq == 0 ? a.rs_a : a.rs_b;
                        dst[pidx] = kq == 0 ? m : -m;
                    }
                } else {
                    float sm = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int j = 0; j < 4; j++) sm = sm + v[r][j];
                    float p, q;
                    rs_pair16(sm, p, q); sm = p + q;
                    rs_pair32(sm, p, q); sm = p + q;
                    if (kq == 0 && ok) a.rs_a[pidx] = sm;
                }
            }
    };
    // the k-loop of one pixel tile into `acc`.  FLUSH (double-buffered accumulators, round 5): row tile t of the PREVIOUS pixel
    // tile (`old`, finished at (oseg, opt)) is stored behind k-step RS_FLUSH0 + RS_FLUSHD t of this one — the 64 KB a
    // workgroup writes per tile leave while the matrix pipe works instead of in a burst between two tiles (the epilogue was
    // 3-6 % of a tile at 64 textures per step and 14-19 % at 8, where all CUs finish their few tiles in lockstep and the
    // stores of the whole chip collide: profiles/r05_gemm_timeline.md)
    auto run_tile = [&](rs_f4 (&acc)[MT][4], rs_f4 (&old)[MT][4], auto flush, int oseg, int opt, const rs_gptr nbase) {
        constexpr bool FLUSH = decltype(flush)::value;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            rs_f4 b = br[ks % DEPTH];
            if (EXTRA == 2) b = b - bsr[EXTRA == 2 ? ks % DEPTH : 0];
            if (ks >= KS - RAG) {  // the k-step that K cuts: its rows beyond K enter as exact zeros
                const unsigned pm = (4 * ks + 4 > K && !ok_p) ? 0u : 0xffffffffu;
#pragma unroll
                for (int j = 0; j < 4; j++) b[j] = __uint_as_float(__float_as_uint(b[j]) & pm);
            }
            // refill the slot of the PREVIOUS k-step — k-step ks - 1 + DEPTH of this tile, or the head of the next one.  One
            // k-step late on purpose (round 5 timeline): a load whose destination registers were read by the MFMA issued just
            // before it waits for that MFMA — one 32-cycle slot of the matrix pipe per k-step (2180 against 2076 cycles per
            // 64 MFMAs); a k-step later the readers are long done.  DEPTH - 1 k-steps stay in flight.
            if (ks - 1 + DEPTH == KS) pk = nbase;
            constexpr bool SPLIT = KFULL && EXTRA != 2 && RS_SPLIT_LOAD != 0 && MT >= 2;
#ifndef RS_PROBE_NOLOAD
            if (SPLIT) {
                load_half(br[(ks - 1 + DEPTH) % DEPTH], 0);
            } else {
                load_next((ks - 1 + DEPTH) % KS, br[(ks - 1 + DEPTH) % DEPTH]);
                load_sub((ks - 1 + DEPTH) % KS, bsr[EXTRA == 2 ? (ks - 1 + DEPTH) % DEPTH : 0]);
            }
#endif
            if (ks < KS - RAG || 4 * ks < K) {  // uniform: a k-step entirely beyond K does nothing
#pragma unroll
                for (int t = 0; t < MT; t++) {
#ifndef RS_PROBE_NOLOAD
                    if (SPLIT && t == MT / 2) {  // the second half of the refill, half a k-step later
                        __builtin_amdgcn_sched_barrier(0);
                        load_half(br[(ks - 1 + DEPTH) % DEPTH], 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#endif
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const rs_f4 c = ks == 0 ? rs_f4{0.f, 0.f, 0.f, 0.f} : acc[t][j];
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], af[t][ks], c, 0, 0, 0);
                    }
                }
            }
            // pin the software pipeline: unfenced, the machine scheduler sinks every load down to its consumer, eight
            // k-steps later, and the loop becomes load -> s_waitcnt vmcnt(0) -> 16 MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (FLUSH && ks >= RS_FLUSH0 && (ks - RS_FLUSH0) % RS_FLUSHD == 0 && (ks - RS_FLUSH0) / RS_FLUSHD < MT) {
                epilogue_row(old, (ks - RS_FLUSH0) / RS_FLUSHD, oseg, opt);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef OPTEX_TIMELINE
            if (ks % 4 == 3) {
                tl_stamp(tl);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }

    };

    auto next_of = [&](long tile, int& nseg, int& npt) {
        nseg = seg;
        npt = pt;
        if (tile + 1 < te) {  // (the last tile prefetches itself once more: in bounds, unused)
            npt = pt + 1;
            if (npt == ra.tiles_n) { npt = 0; nseg = seg + 1; }
        }
    };
    if constexpr (DB) {
        rs_f4 accA[MT][4], accB[MT][4];
        int oseg = seg, opt = pt, nseg, npt;
        long tile = tb;
        TL_STAMP(tl);
        next_of(tile, nseg, npt);
        run_tile(accA, accB, std::false_type{}, 0, 0, tile_base(nseg, npt));   // the first tile has nothing to flush
        for (;;) {
            oseg = seg; opt = pt; seg = nseg; pt = npt;
#ifdef OPTEX_TIMELINE
            __builtin_amdgcn_sched_barrier(0);
            tl_stamp(tl);
#endif
            if (++tile >= te) {
#pragma unroll
                for (int t = 0; t < MT; t++) epilogue_row(accA, t, oseg, opt);
                break;
            }
            TL_STAMP(tl);
            next_of(tile, nseg, npt);
            run_tile(accB, accA, std::true_type{}, oseg, opt, tile_base(nseg, npt));
            oseg = seg; opt = pt; seg = nseg; pt = npt;
#ifdef OPTEX_TIMELINE
            __builtin_amdgcn_sched_barrier(0);
            tl_stamp(tl);
#endif
            if (++tile >= te) {
#pragma unroll
                for (int t = 0; t < MT; t++) epilogue_row(accB, t, oseg, opt);
                break;
            }
            TL_STAMP(tl);
            next_of(tile, nseg, npt);
            run_tile(accA, accB, std::true_type{}, oseg, opt, tile_base(nseg, npt));
        }
    } else {
        for (long tile = tb; tile < te; tile++) {
            TL_STAMP(tl);
            int nseg, npt;
            next_of(tile, nseg, npt);
            rs_f4 acc[MT][4];
            run_tile(acc, acc, std::false_type{}, 0, 0, tile_base(nseg, npt));
#pragma unroll
            for (int t = 0; t < MT; t++) epilogue_row(acc, t, seg, pt);
            seg = nseg;
            pt = npt;
#ifdef OPTEX_TIMELINE
            __builtin_amdgcn_sched_barrier(0);
            tl_stamp(tl);
#endif
        }
    }
#ifdef OPTEX_TIMELINE
    tl_stamp_real(tl);
    tl_end();
#endif
}

static inline bool rs_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// shapes / layouts the R-stationary kernel takes (channel-major on both sides is the caller's business)
bool gemm_rs_supported(const GemmArgs& a, int n_cu) {
    if (a.epi || a.sym) return false;
    if (a.bsub && a.bsub_ss != 0 && a.at_ss == 0) return false;  // a workgroup keeps one matrix AND one centring vector
    // centring with four row tiles per wave (M > 192): built for K = 256 exactly and for K <= 192; the ragged 192 < K < 256
    // instantiation spills 32 registers (the LDS-tiled kernel takes those)
    if (a.bsub && a.M > 192 && a.K > 192 && a.K != 256) return false;
    if (a.M <= 64 || a.M > 256 || a.K < 64 || a.K > 256) return false;
    if (a.n % RS_BN != 0 || a.n <= 0) return false;
    if (!rs_aligned16(a.B) || a.ldb % 4 != 0 || a.b_ss % 4 != 0) return false;
    if (!rs_aligned16(a.O) || a.ldo % 4 != 0 || a.o_ss % 4 != 0) return false;
    if (a.content && !rs_aligned16(a.content)) return false;
    // 32-bit lane offsets: a lane addresses row kq <= 3 of a k-step, ((kq * ldb + 4 * l15) * 4) bytes past the wave's base
    if (12ull * (unsigned long long)a.ldb + 256ull >= (1ull << 32)) return false;
    if (a.rowstat && (a.bsub || a.badd || a.content)) return false;  // the row statistics ride in the plain epilogue only
    const long long total = (long long)(a.n / RS_BN) * a.n_seg;
    return total >= 2LL * n_cu;  // every CU streams at least two tiles behind one matrix load
}

int gemm_rs_parts(long n) { return (int)(n / RS_BN); }

template <int MT, int KS>
static int rs_launch_mk(const RsArgs& ra, dim3 grid, hipStream_t st) {
    const GemmArgs& a = ra.g;
#if RS_WPE == 2
    if (MT == 4 && !a.bsub && !a.badd && !a.content && !a.rowstat) {   // probe build: 8 waves x 32 rows instead of 4 x 64
        if (KS == 64 && a.K == 4 * KS) hipLaunchKernelGGL((gemm_rs_kernel<2, KS, 0, 0, 2, KS == 64>), grid, dim3(512), 0, st, ra);
        else hipLaunchKernelGGL((gemm_rs_kernel<2, KS, 0, 0, 2>), grid, dim3(512), 0, st, ra);
        return check_launch("gemm_rs_kernel");
    }
#endif
    if (a.rowstat && (a.bsub || a.badd || a.content)) {
        set_error("gemm_rs_kernel: row statistics cannot be combined with bsub / badd / content (gemm_rs_supported says so)");
        return OPTEX_E_UNSUPPORTED;
    }
    if (a.bsub) {
        // (round 5: with the matrix in accumulation registers the centring ring fits beside four row tiles too)
        if ((MT == 4 && KS == 64) && a.K == 4 * KS)
            hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 0, 2, 1, (MT == 4 && KS == 64)>), grid, dim3(256), 0, st, ra);
        else
            hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 0, 2>), grid, dim3(256), 0, st, ra);
    } else if (a.badd || a.content) {
        // (round 6: double-buffered accumulators here too — the bias / blend of a finished row tile leaves behind the next tile's
        //  k-steps like the plain stores; the linear modes' apply GEMM takes this path with the centring folded into its bias)
        if ((MT == 4 && KS == 64) && a.K == 4 * KS)
            hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 0, 1, 1, (MT == 4 && KS == 64), (MT == 4 && KS == 64) && RS_DB != 0>), grid, dim3(256), 0, st, ra);
        else
            hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 0, 1>), grid, dim3(256), 0, st, ra);
    } else {
        // un-projected feature maps (C = 256, C = 128): K fills the instantiation exactly
        constexpr bool HOT = (MT == 4 && KS == 64) || (MT == 2 && KS == 32);
        const bool kfull = HOT && a.K == 4 * KS;
        // double-buffered accumulators for the ragged (PCA-rank) instantiations too (-DRS_DB_RAGGED=1 probe build): three row
        // tiles per wave and 48 k-steps leave the registers for it (M, K <= 192)
        constexpr bool DBR = RS_DB_RAGGED != 0 && RS_DB != 0 && MT == 3 && KS == 48;
        if (a.rowstat == 1) {
            if (kfull) hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 1, 0, 1, HOT, HOT && MT == 4 && RS_DB != 0>), grid, dim3(256), 0, st, ra);
            else hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 1, 0, 1, false, DBR>), grid, dim3(256), 0, st, ra);
        } else if (a.rowstat == 2) {
            if (kfull) hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 2, 0, 1, HOT, HOT && MT == 4 && RS_DB != 0>), grid, dim3(256), 0, st, ra);
            else hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 2, 0, 1, false, DBR>), grid, dim3(256), 0, st, ra);
        } else {
            if (kfull) hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 0, 0, 1, HOT, HOT && MT == 4 && RS_DB != 0>), grid, dim3(256), 0, st, ra);
            else hipLaunchKernelGGL((gemm_rs_kernel<MT, KS, 0, 0, 1, false, DBR>), grid, dim3(256), 0, st, ra);
        }
    }
    return check_launch("gemm_rs_kernel");
}

int gemm_rs_launch(const GemmArgs& a, int n_cu, hipStream_t st) {
    RsArgs ra;
    ra.g = a;
    ra.tiles_n = (int)(a.n / RS_BN);
    ra.segs_per_group = a.at_ss == 0 ? a.n_seg : 1;
    const int groups = a.n_seg / ra.segs_per_group;
    const long per_group = (long)ra.tiles_n * ra.segs_per_group;
    // one workgroup per CU over all groups, minus the spare CUs (include/optex.h, optex_gemm_spare_cus: a workgroup needs a whole
    // CU, and a CU held by a small kernel of another stream would make it the launch's straggler)
    int cus = n_cu - (tl_call.spare_cus >= 0 ? tl_call.spare_cus : gemm_rs_spare_cus);
    if (cus < n_cu / 2) cus = n_cu / 2;
    if (cus < 1) cus = 1;
    // `cus` bounds the TOTAL grid: with several groups (per-segment matrices) the workgroups per group are rounded DOWN, or
    // 64 groups x ceil(255 / 64) would be 256 workgroups again and the spare CU's straggler with them
    long gx = groups <= cus ? cus / groups : 1;
    if (gx > per_group) gx = per_group;
    ProfScope prof(a.prof_cls, st, 2.0 * a.M * a.K * (double)a.n * a.n_seg,
                   4.0 * ((double)(a.K + a.M) * a.n * a.n_seg + (double)a.K * a.M));
    const dim3 grid((unsigned)gx, (unsigned)groups);
        const int mt = a.M > 192 ? 4 : (a.M > 128 ? 3 : 2);   // rows: (64, 128] / (128, 192] / (192, 256]
    const int ks = a.K > 192 ? 64 : (a.K > 128 ? 48 : 32);  // k-steps: K in [64, 128] / (128, 192] / (192, 256]
    switch (mt * 100 + ks) {
        case 464: return rs_launch_mk<4, 64>(ra, grid, st);
        case 448: return rs_launch_mk<4, 48>(ra, grid, st);
        case 432: return rs_launch_mk<4, 32>(ra, grid, st);
        case 364: return rs_launch_mk<3, 64>(ra, grid, st);
        case 348: return rs_launch_mk<3, 48>(ra, grid, st);
        case 332: return rs_launch_mk<3, 32>(ra, grid, st);
        case 264: return rs_launch_mk<2, 64>(ra, grid, st);
        case 248: return rs_launch_mk<2, 48>(ra, grid, st);
        default: return rs_launch_mk<2, 32>(ra, grid, st);
    }
}

}  // namespace optex

namespace optex { int gemm_rs_spare_cus = 1; }

extern "C" int optex_gemm_spare_cus(int spare) {
    const int old = optex::gemm_rs_spare_cus;
    optex::gemm_rs_spare_cus = spare < 0 ? 0 : spare;
    return old;
}

TL_DEFINE_SETTER(tl_set_rs)
