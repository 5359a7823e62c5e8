// linalg.hip — the C x C algebra of the linear modes (histmatch.py:24-42) on the GPU, batched over independent
// segments, so that the hot loop of `chol` / `pca` / `sym` (the reference's default is chol, optex.py:229) runs without a
// host-side factorization between its kernels:
//
//   chol  T = L_s L_t^-1        chol_inv_kernel: one workgroup per matrix, left-looking blocked Cholesky that carries
//                               the rows of L^-T along as extra right-hand sides, so L^T (= U) and L^-1 come out of one
//                               pass; T^T = (L_t^-1)^T U_s is one batched C x C GEMM on the MFMA kernel of gemm.hip.
//   pca   T = S_s^1/2 S_t^-1/2  scaled coupled Newton-Schulz iteration  Y <- a Y W, Z <- a W Z, W = (3 I - a^2 Z Y) / 2
//   sym   T = S_t^-1/2 (S_t^1/2 S_s S_t^1/2)^1/2 S_t^-1/2        from Y0 = A / |A|_F, Z0 = I:  Y -> (A/|A|_F)^1/2, Z -> its inverse.
//                               Only GEMMs (MFMA-shaped work, batched over all segments) instead of a batched symmetric
//                               eigensolver: the reference's Q = V sqrt(L) V^T (histmatch.py:30-31) IS the principal
//                               square root, and eps bounds the spectrum below, which is what the scaling needs to
//                               reach fp32 round-off in 12 steps for |A|_F / lambda_min up to 1e7.
//
// Everything is fp32 like the reference's LAPACK calls; agreement with the reference is by tolerance (1e-4 per step,
// SURVEY 8c), not bit-exact — summation orders differ.

#include "gemm_args.h"

namespace optex {

// ------------------------------------------------------------------------------------------------ small batched products
// OUT[b] = alpha * alpha_seg[b] * (At[b]^T @ B[b]) + diag * I   for `batch` C x C matrices (strides in elements, 0 = shared)
// sym: the product is symmetric in exact arithmetic -> upper tiles only, stored with their mirror image (exactly symmetric)
int small_gemm(const float* At, long lda, long at_ss, const float* B, long ldb, long b_ss, float* O, long ldo, long o_ss, int C,
               int batch, bool epi, float alpha, const float* alpha_seg, float diag, hipStream_t st, bool sym) {
    GemmArgs a;
    a.At = At; a.lda = lda; a.at_ss = at_ss;
    a.B = B; a.ldb = ldb; a.b_ss = b_ss;
    a.O = O; a.ldo = ldo; a.o_ss = o_ss;
    a.M = C; a.K = C; a.n = C; a.n_seg = batch;
    a.bsub = nullptr; a.bsub_ss = 0; a.badd = nullptr; a.badd_ss = 0; a.content = nullptr; a.strength = 0.f;
    a.epi = epi ? 1 : 0; a.alpha = alpha; a.alpha_seg = alpha_seg; a.diag = diag;
    a.sym = sym ? 1 : 0;
    a.prof_cls = KC_SMALL_GEMM;
    a.rowstat = 0; a.rs_a = nullptr; a.rs_b = nullptr;
    return gemm_tn_launch(a, OPTEX_CHANNEL_MAJOR, OPTEX_CHANNEL_MAJOR, st);
}

// out[b][m] = sum_k R_g[it][k][m] * mu[s(g)][k]  for b = it * per + g — the rotated style mean  mu_s @ R  (optex.py:171 +
// histmatch.py:20); set g's rotations start at R + g * r_ss (r_ss = 0: shared), its style segment is g (mu_per_set) or 0
__global__ void rot_mean_kernel(const float* __restrict__ R, long r_ss, const float* __restrict__ mu, int mu_per_set, int C, int per,
                                float* __restrict__ out) {
    const int b = blockIdx.x, it = b / per, g = b % per;
    const float* r = R + (size_t)g * r_ss + (size_t)it * C * C;
    const float* m = mu + (size_t)(mu_per_set ? g : 0) * C;
    for (int j = threadIdx.x; j < C; j += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < C; k++) acc = __builtin_fmaf(r[(size_t)k * C + j], m[k], acc);
        out[(size_t)b * C + j] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ Cholesky + inverse
constexpr int CH_NB = 32;  // panel width

// One workgroup per matrix, blockDim = 2 * NP (NP = C rounded up to a multiple of 32, <= 512).
//   threads [0, NP):   row r of L                     (Cholesky:  A = L L^T)
//   threads [NP, 2NP): row i of W = L^-T, i.e. column i of L^-1   (the same recurrence with A := I)
// Left-looking by panels of 32 columns: a row's 32 panel entries live in registers; the already finished columns are read
// back from global memory (U = L^T and Linv are written row by row = contiguous over the thread index, and stay in L2).
// Outputs, both [NP, NP] with leading dimension NP, zero outside their triangle, identity in the padding:
//   U[k][r]    = L[r][k]      (upper triangular, = L^T)
//   Linv[k][i] = (L^-1)[k][i] (lower triangular)
__global__ __launch_bounds__(1024) void chol_inv_kernel(const float* __restrict__ A, long a_ss, int C, int NP,
                                                         float* __restrict__ U, float* __restrict__ Linv) {
    extern __shared__ __align__(16) float ch_smem[];
    float* pan = ch_smem;                          // [j0][32]: U[k][j0 .. j0+31] for the finished columns k < j0
    float* dg = ch_smem + (size_t)(NP - CH_NB) * CH_NB;  // [32][33]: the diagonal block of the current panel
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* Ab = A + (size_t)b * a_ss;
    float* Ub = U + (size_t)b * NP * NP;
    float* Wb = Linv + (size_t)b * NP * NP;
    const bool roleL = tid < NP;
    const int r = roleL ? tid : tid - NP;          // row of L, or row of W
    const int wave_first = r & ~63, wave_last = wave_first + 63;

    for (int j0 = 0; j0 < NP; j0 += CH_NB) {
        // ---- stage the panel's rows of the finished columns
        for (int idx = tid; idx < j0 * (CH_NB / 4); idx += blockDim.x) {
            const int k = idx / (CH_NB / 4), c4 = (idx % (CH_NB / 4)) * 4;
            *reinterpret_cast<float4*>(pan + k * CH_NB + c4) = *reinterpret_cast<const float4*>(Ub + (size_t)k * NP + j0 + c4);
        }
        __syncthreads();
        // ---- phase A: acc[c] = A[r][j0 + c] - sum_{k < j0} row[k] * L[j0 + c][k]
        float acc[CH_NB];
        const bool wave_active = roleL ? (wave_last >= j0) : (wave_first < j0 + CH_NB);
#pragma unroll
        for (int c = 0; c < CH_NB; c++) {
            const int col = j0 + c;
            float v;
            if (!roleL || r >= C || col >= C) v = (r == col) ? 1.f : 0.f;   // W rows and the identity padding
            else v = Ab[(size_t)col * C + r];                                // A is symmetric: read row `col`, contiguous in r
            acc[c] = v;
        }
        if (wave_active) {
            const float* src = roleL ? Ub : Wb;
            // W[i][k] = 0 for k < i: a wave of W rows starts at its first row
            const int k_beg = roleL ? 0 : wave_first;
            auto axpy = [&](float v, int k) {
                const float4* p = reinterpret_cast<const float4*>(pan + k * CH_NB);
#pragma unroll
                for (int q = 0; q < CH_NB / 4; q++) {
                    const float4 l = p[q];
                    acc[4 * q + 0] = __builtin_fmaf(-v, l.x, acc[4 * q + 0]);
                    acc[4 * q + 1] = __builtin_fmaf(-v, l.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = __builtin_fmaf(-v, l.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = __builtin_fmaf(-v, l.w, acc[4 * q + 3]);
                }
            };
            // the finished columns come back from L2: eight independent loads in flight per thread before the first use
            int k = k_beg;
            for (; k + 8 <= j0; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = src[(size_t)(k + u) * NP + r];
#pragma unroll
                for (int u = 0; u < 8; u++) axpy(v[u], k + u);
            }
            for (; k < j0; k++) axpy(src[(size_t)k * NP + r], k);
        }
        // ---- phase B: the panel itself, column by column (the pivot row publishes its entries through dg)
        float x[CH_NB];
        const int rb = r - j0;  // position inside the diagonal block (L role only)
#pragma unroll
        for (int c = 0; c < CH_NB; c++) {
            float t = acc[c];
#pragma unroll
            for (int k = 0; k < c; k++) t = __builtin_fmaf(-x[k], dg[c * (CH_NB + 1) + k], t);
            const bool pivot = roleL && rb == c;
            if (pivot) dg[c * (CH_NB + 1) + c] = sqrtf(t);
            __syncthreads();
            const float d = dg[c * (CH_NB + 1) + c];
            x[c] = pivot ? d : __fdiv_rn(t, d);
            if (roleL && rb > c && rb < CH_NB) dg[rb * (CH_NB + 1) + c] = x[c];
            __syncthreads();
        }
        // ---- write the panel: U rows j0..j0+31 (entries r >= row), Linv rows j0..j0+31 (entries i <= row)
        float* dst = roleL ? Ub : Wb;
#pragma unroll
        for (int c = 0; c < CH_NB; c++) {
            const int row = j0 + c;
            const bool keep = roleL ? (r >= row) : (r <= row);
            dst[(size_t)row * NP + r] = keep ? x[c] : 0.f;
        }
        __syncthreads();
    }
}

// The same factorization with the panel update on the matrix cores and a barrier-free panel solve (round 3).
//   phase A: acc[r][c] = A[r][j0 + c] - sum_{k < j0} row_r[k] * L[j0 + c][k] is a GEMM of the finished columns against the
//     staged panel rows: per 32-row group one v_mfma_f32_32x32x2_f32 per two k (operand A = pan[k][c] from LDS, operand B =
//     -row[k] from L2, coalesced), the same k-ordered fma chain as the VALU loop of chol_inv_kernel — which spent its time on
//     eight broadcast ds_read_b128 per 32 fmas.  The MFMA leaves a row's 32 values on lanes l and l + 32; sixteen
//     v_permlane32_swap hand every thread its own row.
//   phase B: the 32 x 32 diagonal block is factored by the wavefront that holds it, on the matrix core as well (round 6: one
//     rank-1 MFMA per column, see below); after ONE workgroup barrier every thread solves its row against the finished block
//     without further barriers (chol_inv_kernel: two workgroup barriers per column, 64 per panel).
typedef float chx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(1024) void chol_inv2_kernel(const float* __restrict__ A, long a_ss, int C, int NP,
                                                          float* __restrict__ U, float* __restrict__ Linv) {
    extern __shared__ __align__(16) float ch_smem[];
    float* pan = ch_smem;                                     // [j0][32]: U[k][j0 .. j0+31] for the finished columns k < j0
    float* dg = ch_smem + (size_t)(NP - CH_NB) * CH_NB;           // [32][33]: the diagonal block of the current panel
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const float* Ab = A + (size_t)b * a_ss;
    float* Ub = U + (size_t)b * NP * NP;
    float* Wb = Linv + (size_t)b * NP * NP;
    // 32-row groups: group g holds rows [32 g, 32 g + 32) of L for g < NP / 32, rows of W = L^-T behind them; a wave's two
    // lane halves are groups 2 wave and 2 wave + 1 (NP is a multiple of 32, not of 64: the halves may differ in role)
    const int ngl = NP / CH_NB;
    const int g0 = 2 * wave, g1 = 2 * wave + 1;
    const bool roleL0 = g0 < ngl, roleL1 = g1 < ngl;
    const int base0 = (roleL0 ? g0 : g0 - ngl) * CH_NB, base1 = (roleL1 ? g1 : g1 - ngl) * CH_NB;
    const bool roleL = h ? roleL1 : roleL0;
    const int r = (h ? base1 : base0) + l31;               // this thread's row of L, or of W

    for (int j0 = 0; j0 < NP; j0 += CH_NB) {
        for (int idx = tid; idx < j0 * (CH_NB / 4); idx += blockDim.x) {
            const int k = idx / (CH_NB / 4), c4 = (idx % (CH_NB / 4)) * 4;
            *reinterpret_cast<float4*>(pan + k * CH_NB + c4) = *reinterpret_cast<const float4*>(Ub + (size_t)k * NP + j0 + c4);
        }
        __syncthreads();
        // ---- phase A.  MFMA C/D layout: lane (j = lane & 31, h) holds D[i][j] for i = (q & 3) + 8 (q >> 2) + 4 h, q = 0..15;
        //      here i = panel column c, j = row of the group.  X: group 0 of the wave, Y: group 1.
        chx16 X, Y;
        {
            // initial values in that layout: A[row][col] (symmetric: read row `col`, contiguous over the lanes) for L rows,
            // the identity for W rows and for the padding.  Unconditional loads at clamped addresses + bit masks: 32 guarded
            // loads cost a branch and a wait each.
            const int r0 = base0 + l31, r1 = base1 + l31;
            const unsigned in0 = (roleL0 && r0 < C) ? 0xffffffffu : 0u, in1 = (roleL1 && r1 < C) ? 0xffffffffu : 0u;
            const float* a0 = Ab + (r0 < C ? r0 : 0);
            const float* a1 = Ab + (r1 < C ? r1 : 0);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int col = j0 + (q & 3) + 8 * (q >> 2) + 4 * h;
                const unsigned cm = col < C ? 0xffffffffu : 0u;
                const size_t off = (size_t)(col < C ? col : 0) * C;
                const unsigned v0 = __float_as_uint(a0[off]), v1 = __float_as_uint(a1[off]);
                const unsigned i0 = r0 == col ? 0x3f800000u : 0u, i1 = r1 == col ? 0x3f800000u : 0u;
                X[q] = __uint_as_float((v0 & in0 & cm) | (i0 & ~(in0 & cm)));
                Y[q] = __uint_as_float((v1 & in1 & cm) | (i1 & ~(in1 & cm)));
            }
        }
        {
            // a group of L rows takes part while it is not finished; W[i][k] = 0 for k < i: a group of W rows starts at its
            // first row and takes part once the panel reaches it
            const bool act0 = roleL0 ? (base0 + CH_NB - 1 >= j0) : (base0 < j0 + CH_NB);
            const bool act1 = roleL1 ? (base1 + CH_NB - 1 >= j0) : (base1 < j0 + CH_NB);
            const int kb0 = roleL0 ? 0 : base0, kb1 = roleL1 ? 0 : base1;
            const float* s0 = (roleL0 ? Ub : Wb) + base0 + l31;
            const float* s1 = (roleL1 ? Ub : Wb) + base1 + l31;
            const int kmin = !act0 ? kb1 : (!act1 ? kb0 : (kb0 < kb1 ? kb0 : kb1));
            if (act0 || act1) {
                // All bounds are multiples of 32.  The operands of a trip (four k pairs: four LDS reads + eight reads of finished
                // columns from L2) are requested one trip AHEAD, into the other of two register sets: the loop used to wait an L2
                // round trip (~1 us) in front of every four MFMAs — half the kernel's time at C = 256 (round 6).  Both groups of
                // the wave run every trip from kmin on: a group that is not active yet (W rows above the panel) or no longer
                // (finished L rows) reads the zeros that were stored for it and adds them — same fma chains for every entry that
                // is kept, same bits; the per-trip predicates cost more registers than the MFMAs they saved.
                constexpr int KP = 4;
                float b0[2][KP], b1[2][KP];  // (the LDS operand is read where it is used: its latency is a tenth of L2's)
                auto request = [&](int buf, int k) {
#pragma unroll
                    for (int u = 0; u < KP; u++) {
                        b0[buf][u] = s0[(size_t)(k + 2 * u + h) * NP];
                        b1[buf][u] = s1[(size_t)(k + 2 * u + h) * NP];
                    }
                };
                auto multiply = [&](int buf, int k) {
                    float a[KP];
#pragma unroll
                    for (int u = 0; u < KP; u++) a[u] = pan[(k + 2 * u + h) * CH_NB + l31];
#pragma unroll
                    for (int u = 0; u < KP; u++) {
                        X = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], -b0[buf][u], X, 0, 0, 0);
                        Y = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], -b1[buf][u], Y, 0, 0, 0);
                    }
                };
                int k = kmin;
                if (k < j0) request(0, k);
                for (; k < j0; k += 4 * KP) {   // two trips per round (bounds are multiples of 32 = 4 trips): set 0, then set 1
                    request(1, k + 2 * KP);
                    multiply(0, k);
                    if (k + 4 * KP < j0) request(0, k + 4 * KP);
                    multiply(1, k + 2 * KP);
                }
            }
        }
        // ---- phase B, step 1: the 32 x 32 diagonal block, by the wavefront that holds it — on the matrix core (round 6).  The
        //      block T (symmetric) is X or Y as phase A left it: row c of T lies in ONE register (q = (c & 3) + 4 (c >> 3)) across
        //      the 32 lanes of half (c >> 2) & 1.  Column c of the factor is that row over sqrt(T[c][c]) (one v_readlane), and the
        //      rank-1 update T -= y y^T of the whole block is one v_mfma_f32_32x32x2_f32 (k = 1 operands zero): 32 dependent
        //      {readlane, sqrt, divide, MFMA} instead of 496 LDS broadcasts + 496 fmas on one wavefront (7 us -> 3 us per panel).
        //      Same fma chain per entry (updates in ascending column order, the product -y_i y_j), same sqrt and division: same bits.
        const int gd = j0 / CH_NB;
        if (wave == (gd >> 1)) {
            auto diag = [&](chx16 T) {
#pragma unroll
                for (int c = 0; c < CH_NB; c++) {
                    const int hc = (c >> 2) & 1, qc = (c & 3) + 4 * (c >> 3);
                    const float v = T[qc];
                    const float tcc = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), c + 32 * hc));
                    const float d = sqrtf(tcc);
                    const float y = __fdiv_rn(v, d);
                    if (h == hc && l31 >= c) dg[l31 * (CH_NB + 1) + c] = (l31 == c) ? d : y;   // L[l31][c]
                    if (c + 1 < CH_NB) {
                        float up = y;   // the operand wants the vector on lanes 0 .. 31 (k = 0) and zeros on 32 .. 63 (k = 1)
                        if (hc) up = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false)[1]);
                        const float av = h ? 0.f : up;
                        T = __builtin_amdgcn_mfma_f32_32x32x2f32(av, -av, T, 0, 0, 0);
                    }
                }
            };
            if (gd & 1) diag(Y);
            else diag(X);
        }
        // hand the rows to their threads: after the swap X[q] is column (q & 3) + 8 (q >> 2) and Y[q] that column + 4 of THIS
        // thread's row, on every lane (v_permlane32_swap exchanges X's upper 32 lanes with Y's lower 32)
        float acc[CH_NB];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[q]), __float_as_uint(Y[q]), false, false);
            acc[(q & 3) + 8 * (q >> 2)] = __uint_as_float(sw[0]);
            acc[(q & 3) + 8 * (q >> 2) + 4] = __uint_as_float(sw[1]);
        }
        __syncthreads();
        // ---- step 2: every row against the finished block, no barrier
        float x[CH_NB];
        const int rb = r - j0;
#pragma unroll
        for (int c = 0; c < CH_NB; c++) {
            float t = acc[c];
#pragma unroll
            for (int k = 0; k < c; k++) t = __builtin_fmaf(-x[k], dg[c * (CH_NB + 1) + k], t);
            const float d = dg[c * (CH_NB + 1) + c];
            x[c] = (roleL && rb == c) ? d : __fdiv_rn(t, d);
        }
        float* dst = roleL ? Ub : Wb;
#pragma unroll
        for (int c = 0; c < CH_NB; c++) {
            const int row = j0 + c;
            const bool keep = roleL ? (r >= row) : (r <= row);
            dst[(size_t)row * NP + r] = keep ? x[c] : 0.f;
        }
        __syncthreads();
    }
}

static size_t chol_lds_bytes(int NP) { return ((size_t)(NP - CH_NB) * CH_NB + CH_NB * (CH_NB + 1)) * sizeof(float); }

int chol_np(int C) { return (C + CH_NB - 1) / CH_NB * CH_NB; }

bool chol_use_mfma = true;  // (internal, not ABI: scripts/chol_probe.hip times the two kernels against each other)
int chol_mfma_min_panels = 2;   // (round 6, diagonal block on the matrix core: 28 against 32 us at two panels, 15 against 14 at one)

// A [batch] (C x C, stride a_ss) -> U, Linv [batch, NP, NP]
int launch_chol_inv(const float* A, long a_ss, int C, int batch, float* U, float* Linv, hipStream_t st) {
    const int NP = chol_np(C);
    if (NP > 512) {
        set_error("linear modes: C = %d > 512 channels is not supported by the batched Cholesky", C);
        return OPTEX_E_UNSUPPORTED;
    }
    const size_t lds = chol_lds_bytes(NP);
    // raise the kernel's dynamic-LDS limit only as far as this NP needs (per device; a benign race: the call is idempotent
    // and the recorded size only grows)
    static size_t attr_lds[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > attr_lds[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chol_inv_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(chol_inv2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds);
        if (e != hipSuccess) {
            set_error("chol_inv_kernel: cannot reserve %zu bytes of LDS for C = %d: %s", lds, C, hipGetErrorString(e));
            return OPTEX_E_LAUNCH;
        }
        attr_lds[dev & 63] = lds;
    }
    // 2/3 C^3 flop for the factor + inverse; reads A, writes two triangles
    ProfScope prof(KC_CHOL, st, (2.0 / 3.0) * (double)C * C * C * batch, 12.0 * (double)C * C * batch);
    // the two kernels compute the same fma chains (bit-identical outputs, scripts/chol_probe.hip); the MFMA kernel pays from
    // two panels on: 1.13x at C = 64, 1.35x at 128, 1.56x at 256, 1.69x at 512 (batch 64), 0.96x at a single panel
    if (chol_use_mfma && NP >= chol_mfma_min_panels * CH_NB)
        hipLaunchKernelGGL(chol_inv2_kernel, dim3(batch), dim3(2 * NP), lds, st, A, a_ss, C, NP, U, Linv);
    else
        hipLaunchKernelGGL(chol_inv_kernel, dim3(batch), dim3(2 * NP), lds, st, A, a_ss, C, NP, U, Linv);
    return check_launch("chol_inv_kernel");
}

// ------------------------------------------------------------------------------------------------ Newton-Schulz square roots
// Coupled iteration (Higham, "Stable iterations for the matrix square root", 1997) with the interval scaling of
// Chen & Chow ("A stable scaling of Newton-Schulz for improving the sign function computation of a Hermitian matrix", 2014):
//   Y0 = A / |A|_F, Z0 = I;   W = (3 I - a^2 Z Y) / 2,  Y <- a Y W,  Z <- a W Z;   Y -> (A/|A|_F)^1/2, Z -> its inverse.
// The eigenvalues of Z Y start in [l0^2, 1] with l0^2 = lambda_min / |A|_F, and a_k = sqrt(3 / (1 + l + l^2)),
// l <- a l (3 - a^2 l^2) / 2 maps [l, 1] onto [l', 1] with l' ~ 2.6 l: 10 iterations reach fp32 round-off from
// |A|_F / lambda_min = 1e5, 12 from 1e7 (plain Newton-Schulz, a = 1: 20 and > 30).  Once l = 1 the scaling is 1 and
// further iterations sit on the (stable) fixed point, so a fixed count is safe for every better conditioned matrix.
//
// The products are TRUE products in the order above (small_gemm_nn): the iteration is stable only as long as Y and Z
// commute with W the way exact polynomials in A do.  Measured in fp32: evaluating Y^T W / W^T Z instead (what the
// transposing GEMM gives for free), or storing every product with its mirror image, amplifies the round-off
// asymmetry by 1.6-10x per iteration and diverges for |A|_F / lambda_min > 1e3.
constexpr int NS_MAX_ITERS = 64;

// A @ B for `batch` C x C row-major matrices on the MFMA kernel of gemm.hip, which evaluates At^T @ Bm:
// (A B)^T = B^T A^T, so At := B, Bm := A read "pixel-major" (element (k, i) at i * ld + k) and the result stored
// pixel-major (transposed back).  OUT = alpha * alpha_seg[b] * (A B) + diag * I.  a_ss / b_ss: matrix strides (0 = shared).
int small_gemm_nn(const float* A, long a_ss, const float* B, long b_ss, float* O, int C, int batch, float alpha,
                  const float* alpha_seg, float diag, const int* live_until, int live_idx, hipStream_t st) {
    const long cc = (long)C * C;
    GemmArgs a;
    a.At = B; a.lda = C; a.at_ss = b_ss;
    a.B = A; a.ldb = C; a.b_ss = a_ss;
    a.O = O; a.ldo = C; a.o_ss = cc;
    a.M = C; a.K = C; a.n = C; a.n_seg = batch;
    a.bsub = nullptr; a.bsub_ss = 0; a.badd = nullptr; a.badd_ss = 0; a.content = nullptr; a.strength = 0.f;
    a.epi = 1; a.alpha = alpha; a.alpha_seg = alpha_seg; a.diag = diag; a.sym = 0;
    a.prof_cls = KC_SMALL_GEMM;
    a.rowstat = 0; a.rs_a = nullptr; a.rs_b = nullptr;
    a.live_until = live_until; a.live_idx = live_idx;
    return gemm_tn_launch(a, OPTEX_PIXEL_MAJOR, OPTEX_PIXEL_MAJOR, st);
}

// A[b] @ B[b] with a leading dimension and matrix strides of its own on every operand (A: C x C inside rows of lda floats) — the
// style-side factor times R^T for every iteration of a loop in one launch (ot_loop.hip, prepare_style)
int small_gemm_nn_ld(const float* A, long lda, long a_ss, const float* B, long b_ss, float* O, long o_ss, int C, int batch,
                     hipStream_t st) {
    GemmArgs a;
    a.At = B; a.lda = C; a.at_ss = b_ss;
    a.B = A; a.ldb = lda; a.b_ss = a_ss;
    a.O = O; a.ldo = C; a.o_ss = o_ss;
    a.M = C; a.K = C; a.n = C; a.n_seg = batch;
    a.bsub = nullptr; a.bsub_ss = 0; a.badd = nullptr; a.badd_ss = 0; a.content = nullptr; a.strength = 0.f;
    a.epi = 0; a.alpha = 1.f; a.alpha_seg = nullptr; a.diag = 0.f; a.sym = 0;
    a.prof_cls = KC_SMALL_GEMM;
    a.rowstat = 0; a.rs_a = nullptr; a.rs_b = nullptr;
    return gemm_tn_launch(a, OPTEX_PIXEL_MAJOR, OPTEX_PIXEL_MAJOR, st);
}

// Two independent batches of true products in ONE launch:  O1[b] = alpha alpha1[b] (A1[b] B1[b]),  O2[b] = alpha alpha2[b]
// (A2[b] B2[b]),  b < batch — the Y W and W Z of a Newton-Schulz iteration (3 800 launches per sym-mode step were 2 500 of
// these pairs going out one by one).
int small_gemm_nn_pair(const float* A1, const float* B1, float* O1, const float* alpha1, const float* A2, const float* B2, float* O2,
                       const float* alpha2, int C, int batch, float alpha, const int* live_until, int live_idx, hipStream_t st) {
    const long cc = (long)C * C;
    GemmArgs a;
    a.At = B1; a.lda = C; a.at_ss = cc;
    a.B = A1; a.ldb = C; a.b_ss = cc;
    a.O = O1; a.ldo = C; a.o_ss = cc;
    a.M = C; a.K = C; a.n = C; a.n_seg = 2 * batch;
    a.bsub = nullptr; a.bsub_ss = 0; a.badd = nullptr; a.badd_ss = 0; a.content = nullptr; a.strength = 0.f;
    a.epi = 1; a.alpha = alpha; a.alpha_seg = alpha1; a.diag = 0.f; a.sym = 0;
    a.prof_cls = KC_SMALL_GEMM;
    a.rowstat = 0; a.rs_a = nullptr; a.rs_b = nullptr;
    a.live_until = live_until; a.live_idx = live_idx;
    a.half = batch; a.At2 = B2; a.B2 = A2; a.O2 = O2; a.alpha_seg2 = alpha2;
    return gemm_tn_launch(a, OPTEX_PIXEL_MAJOR, OPTEX_PIXEL_MAJOR, st);
}

// lower end of the spectrum of Z0 Y0 = A / |A|_F; without a bound from the caller: fp32 cannot resolve eigenvalues below
// ~2^-23 |A|_F anyway
__device__ __forceinline__ double ns_l0(float lambda_min, double fro) {
    double l2 = lambda_min > 0.f ? (double)lambda_min / fro : 0.0;
    if (l2 < 1.2e-7) l2 = 1.2e-7;
    if (l2 > 1.0) l2 = 1.0;
    return sqrt(l2);
}

// Y0 = A / |A|_F, Z0 = I, |A|_F, and how many iterations this matrix needs: the recurrence l <- a l (3 - a^2 l^2) / 2 is a
// guaranteed lower bound of the spectrum of Z Y, so once it is within fp32 round-off of 1 every eigenvalue is; one more
// iteration polishes.  The launch-wide count (the largest need, made even so that the ping-pong ends in the same buffers
// as the full count) is taken with an atomic: the host enqueues NS_ITERS iterations and the later ones switch themselves
// off (GemmArgs::live_until) — no device-to-host round trip.
constexpr int NS_INIT_PARTS = 4;
__global__ __launch_bounds__(256) void ns_init_kernel(const float* __restrict__ A, long a_ss, int C, int batch, int K,
                                                      int adaptive, float lambda_min, float* __restrict__ Y,
                                                      float* __restrict__ Z, float* __restrict__ fro_out,
                                                      int* __restrict__ k_need) {
    // grid = (batch, NS_INIT_PARTS): every part takes the norm of the whole matrix (the same sum in the same order, so all parts
    // scale by the same value) and writes its share of the rows of Y0 and Z0 — 4 x batch workgroups instead of batch
    const int b = blockIdx.x, part = blockIdx.y;
    const float* Ab = A + (size_t)b * a_ss;
    const int cc = C * C;
    double s = 0.0;
    if (cc % 4 == 0 && (reinterpret_cast<uintptr_t>(Ab) & 15) == 0) {
        const float4* A4 = reinterpret_cast<const float4*>(Ab);
        int i = threadIdx.x;
        for (; i + 7 * 256 < cc / 4; i += 8 * 256) {   // eight loads in flight, added in the same order (34 -> ~10 us at 64 x 256^2)
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) q[u] = A4[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; u++)
                s += ((double)q[u].x * (double)q[u].x + (double)q[u].y * (double)q[u].y) +
                     ((double)q[u].z * (double)q[u].z + (double)q[u].w * (double)q[u].w);
        }
        for (; i < cc / 4; i += 256) {
            const float4 v = A4[i];
            s += ((double)v.x * (double)v.x + (double)v.y * (double)v.y) + ((double)v.z * (double)v.z + (double)v.w * (double)v.w);
        }
    } else {
        for (int i = threadIdx.x; i < cc; i += 256) {
            const double v = (double)Ab[i];
            s += v * v;
        }
    }
    s = wave_sum(s);
    __shared__ double sh[4];
    __shared__ float fro_s;
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double fro = sqrt((sh[0] + sh[1]) + (sh[2] + sh[3]));
        fro_s = (float)fro;
    }
    if (threadIdx.x == 0 && part == 0) {
        const double fro = sqrt((sh[0] + sh[1]) + (sh[2] + sh[3]));
        fro_out[b] = (float)fro;
        double l = ns_l0(lambda_min, fro);
        int need = 0;
        while (need < K && l < 1.0 - 6e-8) {
            const double a = sqrt(3.0 / (1.0 + l + l * l));
            l = a * l * (3.0 - a * a * l * l) * 0.5;
            need++;
        }
        need += 1;
        need += need & 1;
        atomicMax(k_need, (adaptive && need < K) ? need : K);
    }
    __syncthreads();
    const float fro = fro_s;
    float* Yb = Y + (size_t)b * cc;
    float* Zb = Z + (size_t)b * cc;
    const int rows = (C + NS_INIT_PARTS - 1) / NS_INIT_PARTS, r0 = part * rows, r1 = (r0 + rows < C) ? r0 + rows : C;
    if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(Ab) & 15) == 0 && (reinterpret_cast<uintptr_t>(Yb) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(Zb) & 15) == 0) {
        // 16-byte pieces: a wavefront per row, 64 lanes x 4 columns (the same IEEE divisions)
        for (int r = r0 + (threadIdx.x >> 6); r < r1; r += 4)
            for (int c = (threadIdx.x & 63) * 4; c < C; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(Ab + r * C + c);
                *reinterpret_cast<float4*>(Yb + r * C + c) =
                    make_float4(__fdiv_rn(v.x, fro), __fdiv_rn(v.y, fro), __fdiv_rn(v.z, fro), __fdiv_rn(v.w, fro));
                *reinterpret_cast<float4*>(Zb + r * C + c) =
                    make_float4(r == c ? 1.f : 0.f, r == c + 1 ? 1.f : 0.f, r == c + 2 ? 1.f : 0.f, r == c + 3 ? 1.f : 0.f);
            }
        return;
    }
    for (int r = r0 + (threadIdx.x >> 6); r < r1; r += 4)       // (no integer division per element: it was most of this kernel)
        for (int c = threadIdx.x & 63; c < C; c += 64) {
            Yb[r * C + c] = __fdiv_rn(Ab[r * C + c], fro);
            Zb[r * C + c] = (r == c) ? 1.f : 0.f;
        }
}

// the per-matrix coefficients of the *k_need iterations that run:
//   cw[k][b] = a_k^2   (W = 1.5 I - 0.5 cw Z Y),   cy[k][b] = a_k (x sqrt|A|_F in the last iteration),
//   cz[k][b] = a_k (/ sqrt|A|_F in the last iteration) — the scale of A comes back in the last epilogue.
__global__ __launch_bounds__(256) void ns_coef_kernel(const float* __restrict__ fro_in, int batch, float lambda_min,
                                                      const int* __restrict__ k_need, float* __restrict__ cw,
                                                      float* __restrict__ cy, float* __restrict__ cz) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int K = *k_need;
    const double fro = (double)fro_in[b];
    double l = ns_l0(lambda_min, fro);
    const double r = sqrt(fro);
    for (int k = 0; k < K; k++) {
        const double a = sqrt(3.0 / (1.0 + l + l * l));
        l = a * l * (3.0 - a * a * l * l) * 0.5;
        if (l > 1.0) l = 1.0;
        const bool last = k == K - 1;
        cw[(size_t)k * batch + b] = (float)(a * a);
        cy[(size_t)k * batch + b] = (float)(last ? a * r : a);
        cz[(size_t)k * batch + b] = (float)(last ? a / r : a);
    }
}

// 12 iterations of the interval-scaled iteration reach fp32 round-off for |A|_F / lambda_min <= 1e7 and further
// iterations sit on the fixed point (tests/test_gpu_linalg.py): the count that is ENQUEUED is fixed, the count that RUNS is
// what the worst matrix of the launch needs (ns_init_kernel).  ns_adaptive = false runs all of them (tests, comparisons).
constexpr int NS_ITERS = 12;
static_assert(NS_ITERS <= NS_MAX_ITERS && NS_ITERS % 2 == 0, "coefficient tables; ping-pong parity");
bool ns_adaptive = true;

// Principal square root and inverse square root of `batch` symmetric positive definite C x C matrices whose spectrum
// is bounded below by lambda_min (<= 0: unknown).  buf: ns_ws_floats(C, batch) floats (Y, Z, W, their ping-pong
// partners and the coefficient tables).  The results land in *Yout / *Zout (pointers into buf).
size_t ns_ws_floats(int C, int batch) { return (size_t)5 * batch * C * C + (size_t)(3 * NS_MAX_ITERS + 1) * batch + 64; }

int ns_sqrt(const float* A, long a_ss, int C, int batch, float lambda_min, float* buf, float** Yout, float** Zout, hipStream_t st) {
    const size_t cc = (size_t)C * C, sz = cc * batch;
    float* Y = buf;
    float* Z = buf + sz;
    float* W = buf + 2 * sz;
    float* Y2 = buf + 3 * sz;
    float* Z2 = buf + 4 * sz;
    const int K = NS_ITERS;
    float* cw = buf + 5 * sz;
    float* cy = cw + (size_t)NS_MAX_ITERS * batch;
    float* cz = cy + (size_t)NS_MAX_ITERS * batch;
    float* fro = cz + (size_t)NS_MAX_ITERS * batch;
    int* k_need = reinterpret_cast<int*>(fro + batch);
    if (int rc0 = device_fill_u32(reinterpret_cast<uint32_t*>(k_need), 0u, 1, st)) return rc0;
    {
        ProfScope prof(KC_NS_INIT, st, 0.0, 12.0 * (double)cc * batch);
        hipLaunchKernelGGL(ns_init_kernel, dim3(batch, NS_INIT_PARTS), dim3(256), 0, st, A, a_ss, C, batch, K, ns_adaptive ? 1 : 0,
                           lambda_min, Y, Z, fro, k_need);
        hipLaunchKernelGGL(ns_coef_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, fro, batch, lambda_min, k_need, cw, cy, cz);
    }
    int rc = check_launch("ns_init_kernel");
    if (rc) return rc;
    for (int k = 0; k < K; k++) {
        const size_t o = (size_t)k * batch;
        if ((rc = small_gemm_nn(Z, (long)cc, Y, (long)cc, W, C, batch, -0.5f, cw + o, 1.5f, k_need, k, st))) return rc;   // W = 1.5 I - 0.5 a^2 Z Y
        if ((rc = small_gemm_nn_pair(Y, W, Y2, cy + o, W, Z, Z2, cz + o, C, batch, 1.f, k_need, k, st))) return rc;        // Y <- a Y W, Z <- a W Z
        float* t = Y; Y = Y2; Y2 = t;
        t = Z; Z = Z2; Z2 = t;
    }
    *Yout = Y;
    *Zout = Z;
    return OPTEX_OK;
}

// ------------------------------------------------------------------------------------------------ one transfer operator
static int dcopy(float* dst, const float* src, size_t count, hipStream_t st) { return device_copy(dst, src, count, st); }

struct TransferWs {
    float *Us, *Ls, *Ut, *Lt;      // chol
    float *ns_buf, *Ys, *Yt, *Zt, *G1, *G;  // pca / sym
    size_t bytes;
    TransferWs(void* ws, int mode, int C, int n_seg, int Ss) {
        char* base = static_cast<char*>(ws);
        size_t off = 0;
        auto take = [&](size_t floats) {
            float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
            off += align_up(floats * sizeof(float), 256);
            return p;
        };
        const size_t cc = (size_t)C * C, pp = (size_t)chol_np(C) * chol_np(C);
        Us = Ls = Ut = Lt = ns_buf = Ys = Yt = Zt = G1 = G = nullptr;
        if (mode == 2) {
            Us = take(Ss * pp); Ls = take(Ss * pp); Ut = take(n_seg * pp); Lt = take(n_seg * pp);
        } else {
            ns_buf = take(ns_ws_floats(C, n_seg > Ss ? n_seg : Ss));
            Ys = take(Ss * cc); Yt = take(n_seg * cc); Zt = take(n_seg * cc); G1 = take(n_seg * cc); G = take(n_seg * cc);
        }
        bytes = off;
    }
};

}  // namespace optex

using namespace optex;

extern "C" int optex_chol_ld(int C) { return chol_np(C); }

extern "C" int optex_chol_inv(const float* A, long a_seg_stride, int C, int batch, float* U, float* Linv, void* stream) {
    if (!A || !U || !Linv || C < 1 || batch < 1 || a_seg_stride < 0) {
        set_error("optex_chol_inv: bad argument (C=%d batch=%d)", C, batch);
        return OPTEX_E_ARG;
    }
    return launch_chol_inv(A, a_seg_stride, C, batch, U, Linv, as_stream(stream));
}

extern "C" size_t optex_spd_sqrt_ws_bytes(int C, int batch) { return ns_ws_floats(C, batch) * sizeof(float); }

extern "C" int optex_spd_sqrt(const float* A, long a_seg_stride, int C, int batch, float lambda_min, float* Y, float* Z, void* ws,
                              size_t ws_bytes, void* stream) {
    if (!A || C < 1 || batch < 1 || a_seg_stride < 0) {
        set_error("optex_spd_sqrt: bad argument (C=%d batch=%d)", C, batch);
        return OPTEX_E_ARG;
    }
    if (int rc = check_ws("optex_spd_sqrt", ws, ws_bytes, optex_spd_sqrt_ws_bytes(C, batch))) return rc;
    hipStream_t st = as_stream(stream);
    float *y, *z;
    int rc = ns_sqrt(A, a_seg_stride, C, batch, lambda_min, static_cast<float*>(ws), &y, &z, st);
    if (rc) return rc;
    if (Y && (rc = dcopy(Y, y, (size_t)batch * C * C, st))) return rc;
    if (Z && (rc = dcopy(Z, z, (size_t)batch * C * C, st))) return rc;
    return OPTEX_OK;
}

extern "C" size_t optex_transfer_operator_ws_bytes(int mode, int C, int n_seg, int src_n_seg) {
    return TransferWs(nullptr, mode, C, n_seg, src_n_seg).bytes;
}

extern "C" int optex_transfer_operator(int mode, const float* cov_t, const float* cov_s, int C, int n_seg, int src_n_seg,
                                       float eps, float* Tt, void* ws, size_t ws_bytes, void* stream) {
    if (!cov_t || !cov_s || !Tt || C < 1 || n_seg < 1 || (src_n_seg != 1 && src_n_seg != n_seg) || mode < 2 || mode > 4) {
        set_error("optex_transfer_operator: bad argument (mode=%d C=%d n_seg=%d src_n_seg=%d; modes 2 = chol, 3 = pca, 4 = sym)",
                  mode, C, n_seg, src_n_seg);
        return OPTEX_E_ARG;
    }
    if (int rc = check_ws("optex_transfer_operator", ws, ws_bytes, optex_transfer_operator_ws_bytes(mode, C, n_seg, src_n_seg)))
        return rc;
    hipStream_t st = as_stream(stream);
    TransferWs w(ws, mode, C, n_seg, src_n_seg);
    const size_t cc = (size_t)C * C;
    const int NP = chol_np(C), Ss = src_n_seg;
    const size_t pp = (size_t)NP * NP;
    int rc;
    if (mode == 2) {  // histmatch.py:24-27
        if ((rc = launch_chol_inv(cov_s, (long)cc, C, Ss, w.Us, w.Ls, st))) return rc;
        if ((rc = launch_chol_inv(cov_t, (long)cc, C, n_seg, w.Ut, w.Lt, st))) return rc;
        return small_gemm(w.Lt, NP, (long)pp, w.Us, NP, Ss > 1 ? (long)pp : 0, Tt, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false);
    }
    float *Y, *Z;
    if (mode == 3) {  // histmatch.py:29-34
        if ((rc = ns_sqrt(cov_s, (long)cc, C, Ss, eps, w.ns_buf, &Y, &Z, st))) return rc;
        if ((rc = dcopy(w.Ys, Y, Ss * cc, st))) return rc;
        if ((rc = ns_sqrt(cov_t, (long)cc, C, n_seg, eps, w.ns_buf, &Y, &Z, st))) return rc;
        return small_gemm(Z, C, (long)cc, w.Ys, C, Ss > 1 ? (long)cc : 0, Tt, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false);
    }
    // histmatch.py:36-42
    if ((rc = ns_sqrt(cov_t, (long)cc, C, n_seg, eps, w.ns_buf, &Y, &Z, st))) return rc;
    if ((rc = dcopy(w.Yt, Y, n_seg * cc, st))) return rc;
    if ((rc = dcopy(w.Zt, Z, n_seg * cc, st))) return rc;
    if ((rc = small_gemm(cov_s, C, Ss > 1 ? (long)cc : 0, w.Yt, C, (long)cc, w.G1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false)))
        return rc;
    if ((rc = small_gemm(w.Yt, C, (long)cc, w.G1, C, (long)cc, w.G, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, true))) return rc;
    if ((rc = ns_sqrt(w.G, (long)cc, C, n_seg, eps * eps, w.ns_buf, &Y, &Z, st))) return rc;
    if ((rc = small_gemm(Y, C, (long)cc, w.Zt, C, (long)cc, w.G1, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false))) return rc;
    return small_gemm(w.Zt, C, (long)cc, w.G1, C, (long)cc, Tt, C, (long)cc, C, n_seg, false, 1.f, nullptr, 0.f, st, false);
}
