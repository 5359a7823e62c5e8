// gemm.hip — K1: rotation / apply GEMM on the fp32 matrix cores of gfx950.
//
//   OUT[s][m][i] = sum_k At[s][k][m] * (B[s][k][i] - bsub[s][k]) + badd[s][m]   (+ content blend)
//
// Replaces optex.py:170,171,175 (the three rotation matmuls) and histmatch.py:27/34/42,44 (T @ hist_t + mu_s).
// The small left matrix (a rotation, C x C) is shared by all pixels; the big operand is the feature map,
// kept channel-major ([C, n] rows of contiguous pixels = NCHW memory) so that every kernel downstream
// (min/max, histogram, LUT apply, sort) streams whole cache lines.
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain.  Lanes 0-31 hold k = 2j, lanes 32-63 hold
// k = 2j+1 of MFMA step j, steps and K-chunks run in ascending order, one accumulator per output element:
// the result is bit-identical to   for k in 0..K-1: acc = fmaf(a[k], b[k], acc)   (oracle orc_gemm_tn).
//
// Roofline: 2*M*K*n flop against 4*(K + M)*n + 4*K*M bytes; at C = 256 the intensity is 64 flop/B, i.e.
// MFMA-bound (157 TFLOP/s fp32 matrix peak); below C ~ 80 it turns HBM-bound.
#include "gemm_args.h"
#include "timeline.h"

namespace optex {

// BK = K-chunk staged per LDS buffer (BK / 2 MFMA steps of k = 2); the block has WGM x WGN waves (m x n)
// VEC: 16-byte accesses on the feature map (and a pixel-major output); AVEC: on the small matrix (its own alignment: a
// ragged PCA rank, lda = k = 181, costs the matrix its vector loads, not the feature map)
template <int BM, int BN, int BK, int WGM, int WGN, bool BPM, bool OPM, bool VEC, bool AVEC = VEC>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_tn_kernel(GemmArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;  // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;   // 32x32 MFMA tiles per wave
    constexpr int BSTR = BPM ? (BK + 1) : BN;   // pixel-major B is staged [pixel][k] with an odd stride
    constexpr int NA = BK * BM / 4 / NT, NB = BK * BN / 4 / NT;
    static_assert(NA >= 1 && NB >= 1 && NA * NT * 4 == BK * BM && NB * NT * 4 == BK * BN, "tile / thread count mismatch");
    static_assert(TM >= 1 && TN >= 1 && TM * 32 * WGM == BM && TN * 32 * WGN == BN, "wave tile must be a multiple of 32x32");

    __shared__ float As[2][BK * BM];
    __shared__ float Bs[2][BPM ? BN * (BK + 1) : BK * BN];

    if (a.live_until && a.live_idx >= *a.live_until) return;  // uniform for the launch
    const unsigned nblocks = gridDim.x;
    const unsigned L = xcd_remap(blockIdx.x, nblocks);
    const int tm_idx = L % a.tiles_m;
    const int tn_idx = (L / a.tiles_m) % a.tiles_n;
    const int seg = L / (a.tiles_m * a.tiles_n);
    const int m0 = tm_idx * BM;
    const long n0 = (long)tn_idx * BN;
    const bool sym = !OPM && !BPM && BM == BN && a.sym;
    if (sym && (long)m0 > n0) return;  // the mirror image of a tile above the diagonal (uniform for the block)

    // (second operand set of a two-batch launch: uniform for the block)
    const bool second = a.half > 0 && seg >= a.half;
    const int oseg = second ? seg - a.half : seg;
    const float* __restrict__ At = (second ? a.At2 : a.At) + (size_t)oseg * a.at_ss;
    const float* __restrict__ Bp = (second ? a.B2 : a.B) + (size_t)oseg * a.b_ss;
    const float* __restrict__ bsub = a.bsub ? a.bsub + (size_t)seg * a.bsub_ss : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, h = lane >> 5;

    float4 ra[NA], rb[NB];

    auto load_global = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = tid + i * NT;
            const int k = idx / (BM / 4), m = m0 + (idx % (BM / 4)) * 4;
            const int kk = k0 + k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < a.K) {
                const float* p = At + (size_t)kk * a.lda + m;
                if (AVEC && m + 3 < a.M) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (m + 0 < a.M) v.x = p[0];
                    if (m + 1 < a.M) v.y = p[1];
                    if (m + 2 < a.M) v.z = p[2];
                    if (m + 3 < a.M) v.w = p[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int idx = tid + i * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!BPM) {
                const int k = idx / (BN / 4);
                const long nn = n0 + (idx % (BN / 4)) * 4;
                const int kk = k0 + k;
                if (kk < a.K) {
                    const float* p = Bp + (size_t)kk * a.ldb + nn;
                    if (VEC && nn + 3 < a.n) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (nn + 0 < a.n) v.x = p[0];
                        if (nn + 1 < a.n) v.y = p[1];
                        if (nn + 2 < a.n) v.z = p[2];
                        if (nn + 3 < a.n) v.w = p[3];
                    }
                    if (bsub) {
                        const float s = bsub[kk];
                        v.x -= s; v.y -= s; v.z -= s; v.w -= s;
                        // columns past n hold -s: they only feed outputs that are never stored
                    }
                }
            } else {
                const int px = idx / (BK / 4);
                const int kk = k0 + (idx % (BK / 4)) * 4;
                const long nn = n0 + px;
                if (nn < a.n) {
                    const float* p = Bp + (size_t)nn * a.ldb + kk;
                    if (VEC && kk + 3 < a.K) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (kk + 0 < a.K) v.x = p[0];
                        if (kk + 1 < a.K) v.y = p[1];
                        if (kk + 2 < a.K) v.z = p[2];
                        if (kk + 3 < a.K) v.w = p[3];
                    }
                    if (bsub) {
                        if (kk + 0 < a.K) v.x -= bsub[kk + 0];
                        if (kk + 1 < a.K) v.y -= bsub[kk + 1];
                        if (kk + 2 < a.K) v.z -= bsub[kk + 2];
                        if (kk + 3 < a.K) v.w -= bsub[kk + 3];
                    }
                }
            }
            rb[i] = v;
        }
    };

    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = tid + i * NT;
            const int k = idx / (BM / 4), m = (idx % (BM / 4)) * 4;
            *reinterpret_cast<float4*>(&As[buf][k * BM + m]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int idx = tid + i * NT;
            if (!BPM) {
                const int k = idx / (BN / 4), nn = (idx % (BN / 4)) * 4;
                *reinterpret_cast<float4*>(&Bs[buf][k * BN + nn]) = rb[i];
            } else {
                const int px = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
                float* d = &Bs[buf][px * (BK + 1) + kk];
                d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
            }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nchunks = (a.K + BK - 1) / BK;
    load_global(0);
    store_lds(0);
    __syncthreads();

    for (int kc = 0; kc < nchunks; kc++) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_global((kc + 1) * BK);
        const float* as = &As[buf][wm * WM + l31];
        const float* bs = BPM ? &Bs[buf][(wn * WN + l31) * BSTR] : &Bs[buf][wn * WN + l31];
#pragma unroll
        for (int j = 0; j < BK / 2; j++) {
            const int k = 2 * j + h;
            float av[TM], bv[TN];
#pragma unroll
            for (int t = 0; t < TM; t++) av[t] = as[k * BM + t * 32];
#pragma unroll
            for (int t = 0; t < TN; t++) bv[t] = BPM ? bs[t * 32 * BSTR + k] : bs[k * BN + t * 32];
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
        }
        if (kc + 1 < nchunks) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* __restrict__ Op = (second ? a.O2 : a.O) + (size_t)oseg * a.o_ss;
    const float* __restrict__ Cp = a.content ? a.content + (size_t)seg * a.o_ss : nullptr;
    const float* __restrict__ badd = a.badd ? a.badd + (size_t)seg * a.badd_ss : nullptr;
    const float strength = a.strength;
    const bool epi = a.epi;
    const float* aseg = second ? a.alpha_seg2 : a.alpha_seg;
    const float alpha = epi ? (aseg ? a.alpha * aseg[oseg] : a.alpha) : 1.f;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
        const int mb = m0 + wm * WM + tm * 32 + 4 * h;
        // the 16 row biases of this lane, loaded together (not one dependent load per element)
        float bias[16];
        if (!OPM && badd) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                bias[r] = badd[m < a.M ? m : 0];
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
            const long nn = n0 + wn * WN + tn * 32 + l31;
            if (nn >= a.n) continue;
            if (!OPM) {
                float cv[16];
                if (Cp) {  // the 16 content values of this lane's column, in flight together
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        cv[r] = Cp[(size_t)(m < a.M ? m : 0) * a.ldo + nn];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < a.M) {
                        float v = acc[tm][tn][r];
                        if (epi) {
                            v = v * alpha;
                            if ((long)m == nn) v = v + a.diag;
                        }
                        if (badd) v = v + bias[r];
                        const size_t off = (size_t)m * a.ldo + nn;
                        if (Cp) {
                            const float d = cv[r] - v;
                            const float sd = strength * d;
                            v = v + sd;
                        }
                        if (sym) {  // diagonal tiles hold both (m, i) and (i, m): keep the upper one
                            if ((long)m <= nn) {
                                Op[off] = v;
                                if (nn < a.M) Op[(size_t)nn * a.ldo + m] = v;
                            }
                            continue;
                        }
                        Op[off] = v;
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int m = mb + 8 * g;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        v[q] = acc[tm][tn][g * 4 + q];
                        if (epi) {
                            v[q] = v[q] * alpha;
                            if ((long)(m + q) == nn) v[q] = v[q] + a.diag;
                        }
                        if (badd && m + q < a.M) v[q] = v[q] + badd[m + q];
                    }
                    const size_t off = (size_t)nn * a.ldo + m;
                    if (Cp) {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (m + q < a.M) {
                                const float d = Cp[off + q] - v[q];
                                const float sd = strength * d;
                                v[q] = v[q] + sd;
                            }
                    }
                    if (VEC && m + 3 < a.M) {
                        *reinterpret_cast<float4*>(Op + off) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (m + q < a.M) Op[off + q] = v[q];
                    }
                }
            }
        }
    }
}

// ---- Hot-loop specialisation (channel-major in / out, no epilogue extras, full pixel tiles): the same block structure
// on v_mfma_f32_16x16x4_f32 — 4 accumulator registers per MFMA instead of 16, i.e. half the accumulator read / write
// traffic per flop.  Bit-identical results (both MFMA shapes are k-ordered fma chains); measured 118.6 vs 116.0 TFLOP/s
// at M = K = 256, n = 32 x 16384.
typedef float floatx4 __attribute__((ext_vector_type(4)));

// EXTRA = true adds the operand centring (bsub), the output bias (badd) and the content blend of the general kernel, with
// the same arithmetic in the same order (bit-identical to gemm_tn_kernel): the apply GEMM of the linear modes
// (histmatch.py:27/34/42,44) and the blending inverse rotation of style transfer (optex.py:115-117, 175) take this path too.
// ROWSTAT: 0 = off, 1 = per-row min / max, 2 = per-row sum of the outputs (GemmArgs::rowstat), reduced over the wave's 64
// pixel columns in registers (16 in-lane, then two cross-lane steps) instead of a separate pass that re-reads the whole
// rotated map from HBM.
template <int BM, int BN, int BK, int WGM, int WGN, bool EXTRA, int ROWSTAT = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm16_cm_kernel(GemmArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    constexpr int SA = BM + 16, SB = BN + 16;  // row strides = 16 mod 32 banks: the four k-rows of a fragment read hit disjoint banks
    constexpr int NA = BK * BM / 4 / NT, NB = BK * BN / 4 / NT;
    static_assert(NA >= 1 && NB >= 1 && NA * NT * 4 == BK * BM && NB * NT * 4 == BK * BN, "tile / thread count mismatch");
    __shared__ float As[2][BK * SA];
    __shared__ float Bs[2][BK * SB];
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int tm_idx = L % a.tiles_m;
    const int tn_idx = (L / a.tiles_m) % a.tiles_n;
    const int seg = L / (a.tiles_m * a.tiles_n);
    const int m0 = tm_idx * BM;
    const long n0 = (long)tn_idx * BN;
    const float* __restrict__ At = a.At + (size_t)seg * a.at_ss;
    const float* __restrict__ Bp = a.B + (size_t)seg * a.b_ss;
    const float* __restrict__ bsub = (EXTRA && a.bsub) ? a.bsub + (size_t)seg * a.bsub_ss : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l15 = lane & 15, kq = lane >> 4;
    float4 ra[NA], rb[NB];
    auto load_global = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = tid + i * NT;
            const int k = idx / (BM / 4), m = m0 + (idx % (BM / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + k < a.K && m + 3 < a.M) v = *reinterpret_cast<const float4*>(At + (size_t)(k0 + k) * a.lda + m);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int idx = tid + i * NT;
            const int k = idx / (BN / 4);
            const long nn = n0 + (idx % (BN / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + k < a.K) {
                v = *reinterpret_cast<const float4*>(Bp + (size_t)(k0 + k) * a.ldb + nn);
                if (EXTRA && bsub) {
                    const float s = bsub[k0 + k];
                    v.x -= s; v.y -= s; v.z -= s; v.w -= s;
                }
            }
            rb[i] = v;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = tid + i * NT;
            *reinterpret_cast<float4*>(&As[buf][(idx / (BM / 4)) * SA + (idx % (BM / 4)) * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int idx = tid + i * NT;
            *reinterpret_cast<float4*>(&Bs[buf][(idx / (BN / 4)) * SB + (idx % (BN / 4)) * 4]) = rb[i];
        }
    };
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int nchunks = (a.K + BK - 1) / BK;
#ifdef OPTEX_TIMELINE
    // slab: [0] = XCC id, [1] = real time at entry, [2] = entry, [3] = first chunk staged (behind the barrier), then per K chunk
    // 4 stamps (global loads issued, MFMAs issued, LDS refilled, barrier passed), then stores issued, then the real time
    tl_ptr tl = tl_begin(blockIdx.x * (NT / 64) + __builtin_amdgcn_readfirstlane(wave));
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tl_word(tl, (unsigned long long)(xcc & 0xff));
    }
    tl_stamp_real(tl);
    tl_stamp(tl);
#endif
    load_global(0);
    store_lds(0);
    __syncthreads();
    TL_STAMP(tl);
    for (int kc = 0; kc < nchunks; kc++) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_global((kc + 1) * BK);
#ifdef OPTEX_TIMELINE
        __builtin_amdgcn_sched_barrier(0);
        tl_stamp(tl);
        __builtin_amdgcn_sched_barrier(0);
#endif
        const float* as = &As[buf][kq * SA + wm * WM + l15];
        const float* bs = &Bs[buf][kq * SB + wn * WN + l15];
#pragma unroll
        for (int j = 0; j < BK / 4; j++) {
            float av[TM], bv[TN];
#pragma unroll
            for (int t = 0; t < TM; t++) av[t] = as[(4 * j) * SA + t * 16];
#pragma unroll
            for (int t = 0; t < TN; t++) bv[t] = bs[(4 * j) * SB + t * 16];
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[tn], av[tm], acc[tm][tn], 0, 0, 0);
        }
#ifdef OPTEX_TIMELINE
        __builtin_amdgcn_sched_barrier(0);
        tl_stamp(tl);
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (kc + 1 < nchunks) store_lds(buf ^ 1);
#ifdef OPTEX_TIMELINE
        __builtin_amdgcn_sched_barrier(0);
        tl_stamp(tl);
        __builtin_amdgcn_sched_barrier(0);
#endif
        __syncthreads();
        TL_STAMP(tl);
    }
    // The MFMA is fed (B fragment, A fragment), i.e. it computes the TRANSPOSED 16 x 16 block: in its C/D layout
    // (col = lane & 15, row = 4 * (lane >> 4) + r) a lane then holds FOUR CONSECUTIVE PIXELS (r) of ONE channel (lane & 15),
    // so the tile leaves with 16-byte stores along the pixel rows (16 store instructions per wave instead of 64 dword
    // stores) and the bias / content of a lane are one scalar / one 16-byte load.  Every output element is still the same
    // k-ordered fma chain (a * b commutes exactly): bit-identical to the other operand order.
    float* __restrict__ Op = a.O + (size_t)seg * a.o_ss;
    const float* __restrict__ Cp = (EXTRA && a.content) ? a.content + (size_t)seg * a.o_ss : nullptr;
    const float* __restrict__ badd = (EXTRA && a.badd) ? a.badd + (size_t)seg * a.badd_ss : nullptr;
    const float strength = a.strength;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
        const int m = m0 + wm * WM + tm * 16 + l15;
        const bool ok = m < a.M;
        const size_t row = (size_t)(ok ? m : 0) * a.ldo;
        float bias = 0.f;
        if (EXTRA && badd) bias = badd[ok ? m : 0];
        float4 cv[TN];
        if (EXTRA && Cp) {  // the content values of this lane, in flight together
#pragma unroll
            for (int tn = 0; tn < TN; tn++)
                cv[tn] = *reinterpret_cast<const float4*>(Cp + row + n0 + wn * WN + tn * 16 + 4 * kq);
        }
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
            const long nn = n0 + wn * WN + tn * 16 + 4 * kq;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                v[r] = acc[tm][tn][r];
                if (EXTRA && badd) v[r] = v[r] + bias;
            }
            if (EXTRA && Cp) {  // same arithmetic in the same order as gemm_tn_kernel: v + strength * (content - v)
                const float c[4] = {cv[tn].x, cv[tn].y, cv[tn].z, cv[tn].w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float d = c[r] - v[r];
                    const float sd = strength * d;
                    v[r] = v[r] + sd;
                }
            }
            if (ok) *reinterpret_cast<float4*>(Op + row + nn) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (ROWSTAT != 0) {
        // one partial per (pixel tile, wave column): [seg][tn_idx * WGN + wn][m]; the statistics are those of the plain
        // product (this path is only taken without bias / blend).  In-lane over the lane's 16 pixels, then over the four
        // lane groups that hold the other pixels of the same channel.
        const size_t pbase = ((size_t)seg * a.tiles_n * WGN + (size_t)tn_idx * WGN + wn) * a.M;
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
            const int m = m0 + wm * WM + tm * 16 + l15;
            if (ROWSTAT == 1) {
                float mn = acc[tm][0][0], mx = acc[tm][0][0];
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        mn = fminf(mn, acc[tm][tn][r]);
                        mx = fmaxf(mx, acc[tm][tn][r]);
                    }
                mn = fminf(mn, __shfl_xor(mn, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mn = fminf(mn, __shfl_xor(mn, 32));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (kq == 0 && m < a.M) {
                    a.rs_a[pbase + m] = mn;
                    a.rs_b[pbase + m] = mx;
                }
            } else {
                float sm = 0.f;
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
#pragma unroll
                    for (int r = 0; r < 4; r++) sm = sm + acc[tm][tn][r];
                sm = sm + __shfl_xor(sm, 16);
                sm = sm + __shfl_xor(sm, 32);
                if (kq == 0 && m < a.M) a.rs_a[pbase + m] = sm;
            }
        }
    }
#ifdef OPTEX_TIMELINE
    __builtin_amdgcn_sched_barrier(0);
    tl_stamp(tl);
    tl_stamp_real(tl);
    tl_end();
#endif
}

template <int BM, int BN, int BK, int WGM, int WGN, bool BPM, bool OPM>
static int launch_cfg(GemmArgs& a, bool vec, hipStream_t st) {
    constexpr int NT = 64 * WGM * WGN;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (int)((a.n + BN - 1) / BN);
    const long long total = (long long)a.tiles_m * a.tiles_n * a.n_seg;
    if (total <= 0 || total > 0x7fffffffLL) {
        set_error("optex_gemm_tn: bad grid (%lld tiles)", total);
        return OPTEX_E_ARG;
    }
    ProfScope prof(a.prof_cls, st, 2.0 * a.M * a.K * (double)a.n * a.n_seg,
                   4.0 * ((double)(a.K + a.M) * a.n * a.n_seg + (double)a.K * a.M));
    if (vec && a.a_vec)
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, BK, WGM, WGN, BPM, OPM, true>), dim3((unsigned)total), dim3(NT), 0, st, a);
    else if (vec)
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, BK, WGM, WGN, BPM, OPM, true, false>), dim3((unsigned)total), dim3(NT), 0, st, a);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, BK, WGM, WGN, BPM, OPM, false>), dim3((unsigned)total), dim3(NT), 0, st, a);
    return check_launch("gemm_tn_kernel");
}

// Tile choice (measured on MI355X at the hot-loop shape M = K = 256, n = 32 x 16384, 600 back-to-back launches):
//   256x128 tile / 512 threads  113 TFLOP/s   <- whole M in one block: the feature map is read from HBM exactly once
//   128x128 tile / 256 threads  109 TFLOP/s      (every pixel tile is fetched by two m-tiles)
// Software-pipelined LDS fragment reads, LDS refill under the MFMAs, BK = 8 / 32, and an LDS-free variant streaming
// fragments straight from L1/L2 were all tried and measured 72-99 TFLOP/s.  What bounds it (round 3, DESIGN.md 4.1,
// profiles/r03_gemm_power_dvfs.md): MFMA duty cycle 0.73-0.77 whatever the data, at the clock the power management allows
// for that data — 2.39 GHz on an all-zero feature map, 2.12 GHz on a dense (gaussian) one.  Round 2's "not power-limited,
// 1243 W at top clock" was a rocm-smi sample (0.28 s) over half-zero features and is withdrawn.
int device_cu_count();

// (internal, not ABI: scripts/gemm_rs_probe.hip times the kernels against each other at every shape)
bool gemm_rs_enabled = true, gemm_rs_force = false;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the hot-loop kernel stores (and reads the content) in 16-byte pieces along the pixel rows
static bool output_vec(const GemmArgs& a) {
    return aligned16(a.O) && a.ldo % 4 == 0 && a.o_ss % 4 == 0 && (!a.content || aligned16(a.content));
}

// shapes the hot-loop kernel (gemm16_cm_kernel, 256 x 128 tiles) takes; layouts and alignment are checked by the callers
static bool hot_shape(const GemmArgs& a, int n_cu) {
    const long long big = (long long)((a.M + 127) / 128) * ((a.n + 127) / 128) * a.n_seg;
    const long long huge = (long long)((a.M + 255) / 256) * ((a.n + 127) / 128) * a.n_seg;
    return big >= 2LL * n_cu && a.M > 64 && !a.epi && a.n % 128 == 0 && a.M % 4 == 0 && a.M > 128 &&
           huge >= 2LL * n_cu;
}

template <bool BPM, bool OPM>
static int launch_layout(GemmArgs& a, bool vec, int n_cu, hipStream_t st) {
    const long long big = (long long)((a.M + 127) / 128) * ((a.n + 127) / 128) * a.n_seg;
    if (a.sym) return launch_cfg<64, 64, 16, 2, 2, BPM, OPM>(a, vec, st);  // square tiles: the mirrored store needs BM == BN
    // Channel-major rotations of the hot loop: the R-stationary kernel (gemm_rs.hip) takes every launch it supports — since round 5
    // (matrix in AGPRs, LDS-staged prologue, double-buffered accumulators) it is ahead of the 256 x 128 LDS-tiled kernel at every
    // shape of the schedule: [64, 256, 16384] gaussian 122.0 against 112.1 TFLOP/s, [64, 256, 4096] 101.3 against 98.0, and at 8
    // textures per step 49 / 103 / 161 us against 57 / 122 / 170 us at 4096 / 9216 / 16384 pixels (profiles/r05_gemm_probes.md);
    // it has no tile quantisation in M or K beyond 16 rows and 4 k either (PCA ranks: M = K = 181: 86 against 60 TFLOP/s).
    // gemm16_cm_kernel keeps what the R-stationary kernel does not take (bias / blend / centring with M > 192, n % 64 != 0, ...).
    if (!BPM && !OPM && gemm_rs_enabled && gemm_rs_supported(a, n_cu)) return gemm_rs_launch(a, n_cu, st);
    if (big >= 2LL * n_cu && a.M > 64) {
        const long long huge = (long long)((a.M + 255) / 256) * ((a.n + 127) / 128) * a.n_seg;
        if (!BPM && !OPM && vec && a.a_vec && hot_shape(a, n_cu) && output_vec(a)) {
            a.tiles_m = (a.M + 255) / 256;
            a.tiles_n = (int)(a.n / 128);
            const long long total = (long long)a.tiles_m * a.tiles_n * a.n_seg;
            ProfScope prof(a.prof_cls, st, 2.0 * a.M * a.K * (double)a.n * a.n_seg,
                           4.0 * ((double)(a.K + a.M) * a.n * a.n_seg + (double)a.K * a.M));
            if (a.bsub || a.badd || a.content)
                hipLaunchKernelGGL((gemm16_cm_kernel<256, 128, 16, 4, 2, true>), dim3((unsigned)total), dim3(512), 0, st, a);
            else if (a.rowstat == 1)
                hipLaunchKernelGGL((gemm16_cm_kernel<256, 128, 16, 4, 2, false, 1>), dim3((unsigned)total), dim3(512), 0, st, a);
            else if (a.rowstat == 2)
                hipLaunchKernelGGL((gemm16_cm_kernel<256, 128, 16, 4, 2, false, 2>), dim3((unsigned)total), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((gemm16_cm_kernel<256, 128, 16, 4, 2, false>), dim3((unsigned)total), dim3(512), 0, st, a);
            return check_launch("gemm16_cm_kernel");
        }
        if (a.M > 128 && huge >= 2LL * n_cu) return launch_cfg<256, 128, 16, 4, 2, BPM, OPM>(a, vec, st);
        return launch_cfg<128, 128, 16, 2, 2, BPM, OPM>(a, vec, st);
    }
    // small problems: 64x64 tiles keep every CU busy.  The C x C products of linalg.hip (64 x [256, 256] per launch) take
    // 34 us each with 64x64, 64x128, 128x64 or 128x128 tiles alike (round 3 probe, 58-63 TFLOP/s): a launch is one round of
    // resident blocks, bounded by its prologue / epilogue latency, not by the tile's arithmetic intensity.
    return launch_cfg<64, 64, 16, 2, 2, BPM, OPM>(a, vec, st);
}

// 16-byte loads need 16-byte aligned rows — per operand: a ragged PCA rank (lda = k = 181) costs the small matrix its
// vector loads, not the feature map
static bool a_operand_vec(const GemmArgs& a) { return aligned16(a.At) && a.lda % 4 == 0 && a.at_ss % 4 == 0; }
static bool b_operand_vec(const GemmArgs& a) { return aligned16(a.B) && a.ldb % 4 == 0 && a.b_ss % 4 == 0; }
static bool operands_vec(const GemmArgs& a) { return a_operand_vec(a) && b_operand_vec(a); }

bool gemm_rowstat_supported(const GemmArgs& a) {
    if (a.bsub || a.badd || a.content) return false;
    const int n_cu = device_cu_count();
    return (gemm_rs_enabled && gemm_rs_supported(a, n_cu)) || (operands_vec(a) && output_vec(a) && hot_shape(a, n_cu));
}

// one partial per 64 pixels: the R-stationary kernel's tiles, or the 256 x 128 kernel's pixel tiles x its two wave columns
int gemm_rowstat_parts(long n) { return n % 64 == 0 ? (int)(n / 64) : 0; }

int gemm_tn_launch(GemmArgs& a, int b_layout, int o_layout, hipStream_t st) {
    const bool bpm = b_layout == OPTEX_PIXEL_MAJOR, opm = o_layout == OPTEX_PIXEL_MAJOR;
    // float4 paths need 16-byte aligned rows on every operand that is accessed with vectors: the template flag covers the
    // feature map (and a pixel-major output), the small matrix carries its own (runtime, uniform) flag
    a.a_vec = a_operand_vec(a) ? 1 : 0;
    bool vec = b_operand_vec(a);
    if (opm) vec = vec && aligned16(a.O) && a.ldo % 4 == 0 && a.o_ss % 4 == 0;
    const int n_cu = device_cu_count();
    if (!bpm && !opm) return launch_layout<false, false>(a, vec, n_cu, st);
    if (bpm && !opm) return launch_layout<true, false>(a, vec, n_cu, st);
    if (!bpm && opm) return launch_layout<false, true>(a, vec, n_cu, st);
    return launch_layout<true, true>(a, vec, n_cu, st);
}

}  // namespace optex

using namespace optex;

extern "C" int optex_gemm_tn(const float* At, long lda, long at_seg_stride, const float* B, long ldb,
                             long b_seg_stride, int b_layout, float* OUT, long ldo, long o_seg_stride, int o_layout,
                             int M, int K, long n, int n_seg, const float* bsub, long bsub_seg_stride,
                             const float* badd, long badd_seg_stride, const float* content, float strength,
                             unsigned flags, void* stream) {
    CallScope call_scope(flags);
    if (!At || !B || !OUT || M <= 0 || K <= 0 || n < 0 || n_seg < 0) {
        set_error("optex_gemm_tn: null pointer or non-positive size (M=%d K=%d n=%ld n_seg=%d)", M, K, n, n_seg);
        return OPTEX_E_ARG;
    }
    if (n == 0 || n_seg == 0) return OPTEX_OK;
    if (lda < M || (b_layout == OPTEX_CHANNEL_MAJOR ? ldb < n : ldb < K) ||
        (o_layout == OPTEX_CHANNEL_MAJOR ? ldo < n : ldo < M)) {
        set_error("optex_gemm_tn: leading dimension smaller than the row length (lda=%ld ldb=%ld ldo=%ld)", lda, ldb, ldo);
        return OPTEX_E_ARG;
    }
    GemmArgs a;
    a.At = At; a.lda = lda; a.at_ss = at_seg_stride;
    a.B = B; a.ldb = ldb; a.b_ss = b_seg_stride;
    a.O = OUT; a.ldo = ldo; a.o_ss = o_seg_stride;
    a.M = M; a.K = K; a.n = n; a.n_seg = n_seg;
    a.bsub = bsub; a.bsub_ss = bsub_seg_stride;
    a.badd = badd; a.badd_ss = badd_seg_stride;
    a.content = content; a.strength = strength;
    a.epi = 0; a.alpha = 1.f; a.alpha_seg = nullptr; a.diag = 0.f; a.sym = 0; a.prof_cls = KC_GEMM;
    a.rowstat = 0; a.rs_a = nullptr; a.rs_b = nullptr;
    return gemm_tn_launch(a, b_layout, o_layout, as_stream(stream));
}

TL_DEFINE_SETTER(tl_set_lds)
