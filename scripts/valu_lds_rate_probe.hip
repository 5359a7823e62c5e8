// Diagnostic: issue cost of the instructions the sort kernels are made of, on gfx950, at the occupancy they run at
// (8 waves per SIMD).  For each instruction kind a kernel issues REPS x 64 of them per wave; cycles per wave-instruction
// per SIMD = elapsed x clock x 1024 SIMDs / (waves x REPS x 64).  The clock is taken from hipDeviceProp (the ratios
// between kinds are what matters).   hipcc --offload-arch=gfx950 -O3 scripts/valu_lds_rate_probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REPS = 2000;

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BODY64(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)

#define VALU_KERNEL(NAME, ASMSTR, PER)                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned seed) {                     \
        unsigned a[8];                                                                              \
        for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i + seed;                              \
        unsigned b = seed + 3, c = threadIdx.x | 1;                                                 \
        for (int r = 0; r < REPS; r++) {                                                            \
            _Pragma("unroll") for (int u = 0; u < 8; u++) {                                         \
                asm volatile(ASMSTR : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]),   \
                             "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c) : "vcc");          \
            }                                                                                       \
        }                                                                                           \
        unsigned s = 0;                                                                             \
        for (int i = 0; i < 8; i++) s += a[i];                                                      \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                                 \
    }                                                                                               \
    static const int NAME##_per = PER;

// 8 instructions on 8 independent accumulators per asm block; the block is repeated 8 x REPS times
#define I8(OP) OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" \
               OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8"
VALU_KERNEL(k_add_u32, I8("v_add_u32"), 8)
VALU_KERNEL(k_and_b32, I8("v_and_b32"), 8)
VALU_KERNEL(k_min_u32, I8("v_min_u32"), 8)
VALU_KERNEL(k_mul_f32, I8("v_mul_f32"), 8)
VALU_KERNEL(k_lshlrev, I8("v_lshlrev_b32"), 8)
VALU_KERNEL(k_mul_lo_u32, I8("v_mul_lo_u32"), 8)
#define I8_3(OP) OP " %0, %0, %8, %9\n\t" OP " %1, %1, %8, %9\n\t" OP " %2, %2, %8, %9\n\t" OP " %3, %3, %8, %9\n\t" \
                 OP " %4, %4, %8, %9\n\t" OP " %5, %5, %8, %9\n\t" OP " %6, %6, %8, %9\n\t" OP " %7, %7, %8, %9"
VALU_KERNEL(k_fma_f32, I8_3("v_fma_f32"), 8)
VALU_KERNEL(k_lshl_add, I8_3("v_lshl_add_u32"), 8)
VALU_KERNEL(k_add3, I8_3("v_add3_u32"), 8)
VALU_KERNEL(k_bfe, I8_3("v_bfe_u32"), 8)
VALU_KERNEL(k_mad_u24, I8_3("v_mad_u32_u24"), 8)
VALU_KERNEL(k_med3, I8_3("v_med3_u32"), 8)
#define I8_CVT(OP) OP " %0, %0\n\t" OP " %1, %1\n\t" OP " %2, %2\n\t" OP " %3, %3\n\t" OP " %4, %4\n\t" OP " %5, %5\n\t" \
                   OP " %6, %6\n\t" OP " %7, %7"
VALU_KERNEL(k_cvt_i32_f32, I8_CVT("v_cvt_i32_f32"), 8)
VALU_KERNEL(k_cvt_f32_u32, I8_CVT("v_cvt_f32_u32"), 8)
VALU_KERNEL(k_mov, I8_CVT("v_mov_b32"), 8)
// compare + add-with-carry pairs (the sort kernel's counting idiom): 8 pairs = 16 instructions
#define P(N) "v_cmp_lt_u32 vcc, %" #N ", %8\n\tv_addc_co_u32 %" #N ", vcc, 0, %" #N ", vcc\n\t"
VALU_KERNEL(k_cmp_addc, P(0) P(1) P(2) P(3) P(4) P(5) P(6) "v_cmp_lt_u32 vcc, %7, %8\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc", 16)
#define Q(N) "v_cmp_lt_u32 vcc, %" #N ", %8\n\tv_cndmask_b32 %" #N ", %" #N ", %9, vcc\n\t"
VALU_KERNEL(k_cmp_cndmask, Q(0) Q(1) Q(2) Q(3) Q(4) Q(5) Q(6) "v_cmp_lt_u32 vcc, %7, %8\n\tv_cndmask_b32 %7, %7, %9, vcc", 16)
#define S(N) "v_sub_co_u32 %" #N ", vcc, %" #N ", %8\n\t"
VALU_KERNEL(k_sub_co, S(0) S(1) S(2) S(3) S(4) S(5) S(6) "v_sub_co_u32 %7, vcc, %7, %8", 8)

// packed: 4 instructions on 4 register pairs
__global__ __launch_bounds__(256) void k_pk_fma_f32(unsigned* out, unsigned seed) {
    double a[4];
    for (int i = 0; i < 4; i++) a[i] = (double)(threadIdx.x + i + seed);
    double b = 1.0000001, c = 0.5;
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\t"
                         "v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c));
        }
    }
    double s = a[0] + a[1] + a[2] + a[3];
    if (s == 1.2345) out[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_fma_f64(unsigned* out, unsigned seed) {
    double a[4];
    for (int i = 0; i < 4; i++) a[i] = (double)(threadIdx.x + i + seed);
    double b = 1.0000001, c = 0.5;
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\t"
                         "v_fma_f64 %3, %3, %4, %5"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c));
        }
    }
    double s = a[0] + a[1] + a[2] + a[3];
    if (s == 1.2345) out[threadIdx.x] = 1;
}
// the quantile index of the sort kernels: u32 -> f64, multiply, fma, f64 -> u32 (4 instructions per element)
__global__ __launch_bounds__(256) void k_quantile_f64(unsigned* out, unsigned seed) {
    unsigned a[4];
    for (int i = 0; i < 4; i++) a[i] = threadIdx.x + i + seed;
    double b = 1.0000001, c = 0.5;
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            double t0, t1, t2, t3;
            asm volatile("v_cvt_f64_u32 %4, %0\n\tv_cvt_f64_u32 %5, %1\n\tv_cvt_f64_u32 %6, %2\n\tv_cvt_f64_u32 %7, %3\n\t"
                         "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8\n\t"
                         "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t"
                         "v_cvt_u32_f64 %0, %4\n\tv_cvt_u32_f64 %1, %5\n\tv_cvt_u32_f64 %2, %6\n\tv_cvt_u32_f64 %3, %7"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                         : "v"(b), "v"(c));
        }
    }
    unsigned s = a[0] + a[1] + a[2] + a[3];
    if (s == 0x12345678u) out[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_pk_add_u16(unsigned* out, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i + seed;
    unsigned b = seed + 3;
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            asm volatile(I8("v_pk_sub_u16") : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),
                         "+v"(a[6]), "+v"(a[7]) : "v"(b));
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

// ---- LDS: 64 KiB of LDS per 256-thread block would cap occupancy; use 16 KiB per block (8 blocks per CU = 8 waves/SIMD)
enum { L_RD32, L_RD64, L_RD128, L_RD128X2, L_WR32, L_ATOM_RTN, L_ATOM_NORTN, L_RD32_SEQ, L_RD128_SEQ, L_RD_U16, L_BPERM };
template <int KIND>
__global__ __launch_bounds__(256) void k_lds(unsigned* out, unsigned seed, int reps) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    unsigned x = (threadIdx.x * 2654435761u + seed) ^ (blockIdx.x * 40503u);
    unsigned acc = 0;
    for (int r = 0; r < reps; r++) {
        unsigned addr[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {  // cheap LCG: the address stream is random and independent of the loaded data
            x = x * 1664525u + 1013904223u;
            addr[j] = x >> 20;  // 12 bits
        }
        if (KIND == L_RD32) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc += lds[addr[j]];
        } else if (KIND == L_RD32_SEQ) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc += lds[(threadIdx.x + 256 * j + r) & 4095];
        } else if (KIND == L_RD_U16) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc += reinterpret_cast<unsigned short*>(lds)[addr[j] * 2 + (j & 1)];
        } else if (KIND == L_RD64) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint2 v = *reinterpret_cast<uint2*>(lds + (addr[j] & ~1u));
                acc += v.x ^ v.y;
            }
        } else if (KIND == L_RD128) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint4 v = *reinterpret_cast<uint4*>(lds + (addr[j] & ~3u));
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        } else if (KIND == L_RD128_SEQ) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint4 v = *reinterpret_cast<uint4*>(lds + ((threadIdx.x * 4 + 1024 * j + 4 * r) & 4095));
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        } else if (KIND == L_RD128X2) {  // 8-slot aligned-by-4 windows: two 16-byte reads at consecutive addresses (4 windows)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned a0 = (addr[j] & ~3u) & 4087u;
                const uint4 v = *reinterpret_cast<uint4*>(lds + a0);
                const uint4 w = *reinterpret_cast<uint4*>(lds + a0 + 4);
                acc += v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
            }
        } else if (KIND == L_WR32) {
#pragma unroll
            for (int j = 0; j < 8; j++) lds[addr[j]] = x + j;
        } else if (KIND == L_ATOM_RTN) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc += atomicAdd(&lds[addr[j]], 1u);
        } else if (KIND == L_ATOM_NORTN) {
#pragma unroll
            for (int j = 0; j < 8; j++) atomicAdd(&lds[addr[j]], 1u);
        } else if (KIND == L_BPERM) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc += (unsigned)__builtin_amdgcn_ds_bpermute((int)(addr[j] << 2), (int)(x + j));
        }
    }
    __syncthreads();
    if (acc == 0x12345678u || lds[threadIdx.x] == 0x9abcdef0u) out[threadIdx.x] = acc;
}

// ---- wave-wide inclusive scan with DPP only (row_shr 1/2/4/8 inside the 16-lane rows, row_bcast 15 / 31 across them)
__device__ __forceinline__ unsigned wave_incl_scan_dpp(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__global__ void k_dpp_scan(unsigned* out) {
    const unsigned v = (threadIdx.x * 2654435761u >> 20) + 1u;
    out[threadIdx.x] = v;
    out[64 + threadIdx.x] = wave_incl_scan_dpp(v);
}
// float min / max butterflies with DPP (row_shr-free: quad_perm, row_half_mirror, row_mirror, row_bcast)
__global__ void k_dpp_minmax(float* out) {
    float v = (float)((threadIdx.x * 2654435761u >> 16) & 0xffff) - 30000.f;
    out[threadIdx.x] = v;
    float m = v;
    m = fminf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0xb1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
    m = fminf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0x4e, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
    m = fminf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0x141, 0xf, 0xf, false)));  // row_half_mirror
    m = fminf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0x140, 0xf, 0xf, false)));  // row_mirror
    // every lane of a row now holds the row minimum; across rows through readlane
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 48));
    out[64 + threadIdx.x] = fminf(fminf(r0, r1), fminf(r2, r3));
}

// ---- the two counting idioms of the sort kernels as they stand there (dependent chains), 8 slots per block
__global__ __launch_bounds__(256) void k_count_cmp(unsigned* out, unsigned seed) {
    unsigned xs[8];
    for (int i = 0; i < 8; i++) xs[i] = threadIdx.x * 7 + i * 977 + seed;
    unsigned lt = 0, le = 0, k = threadIdx.x * 13 + seed;
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            asm volatile("v_cmp_lt_u32 vcc, %2, %10\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
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
                : "v"(xs[0]), "v"(xs[1]), "v"(xs[2]), "v"(xs[3]), "v"(xs[4]), "v"(xs[5]), "v"(xs[6]), "v"(xs[7]), "v"(k) : "vcc");
        }
    }
    if (lt + le == 0x12345678u) out[threadIdx.x] = lt;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_count_fma(unsigned* out, unsigned seed) {
    float xs[8];
    for (int i = 0; i < 8; i++) xs[i] = (float)(threadIdx.x * 7 + i * 977 + seed);
    float k = (float)(threadIdx.x * 13 + seed), big = 1.7014118e38f, c16 = 0.0625f;
    float acc = 524288.f, acc2 = 0.f;
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float t0, t1;
            if (NACC == 1) {
                asm volatile("v_sub_f32 %1, %11, %3\n\tv_sub_f32 %2, %11, %4\n\tv_fma_f32 %1, %1, %12, %13 clamp\n\tv_fma_f32 %2, %2, %12, %13 clamp\n\t"
                    "v_add_f32 %0, %0, %1\n\tv_sub_f32 %1, %11, %5\n\tv_add_f32 %0, %0, %2\n\tv_sub_f32 %2, %11, %6\n\t"
                    "v_fma_f32 %1, %1, %12, %13 clamp\n\tv_fma_f32 %2, %2, %12, %13 clamp\n\t"
                    "v_add_f32 %0, %0, %1\n\tv_sub_f32 %1, %11, %7\n\tv_add_f32 %0, %0, %2\n\tv_sub_f32 %2, %11, %8\n\t"
                    "v_fma_f32 %1, %1, %12, %13 clamp\n\tv_fma_f32 %2, %2, %12, %13 clamp\n\t"
                    "v_add_f32 %0, %0, %1\n\tv_sub_f32 %1, %11, %9\n\tv_add_f32 %0, %0, %2\n\tv_sub_f32 %2, %11, %10\n\t"
                    "v_fma_f32 %1, %1, %12, %13 clamp\n\tv_fma_f32 %2, %2, %12, %13 clamp\n\t"
                    "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2"
                    : "+v"(acc), "=&v"(t0), "=&v"(t1)
                    : "v"(xs[0]), "v"(xs[1]), "v"(xs[2]), "v"(xs[3]), "v"(xs[4]), "v"(xs[5]), "v"(xs[6]), "v"(xs[7]), "v"(k), "s"(big), "v"(c16));
            } else {  // same 24 instructions without the clamp modifier and with an SGPR-free fma
                asm volatile("v_sub_f32 %1, %11, %3\n\tv_sub_f32 %2, %11, %4\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\t"
                    "v_add_f32 %0, %0, %1\n\tv_sub_f32 %1, %11, %5\n\tv_add_f32 %0, %0, %2\n\tv_sub_f32 %2, %11, %6\n\t"
                    "v_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\t"
                    "v_add_f32 %0, %0, %1\n\tv_sub_f32 %1, %11, %7\n\tv_add_f32 %0, %0, %2\n\tv_sub_f32 %2, %11, %8\n\t"
                    "v_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\t"
                    "v_add_f32 %0, %0, %1\n\tv_sub_f32 %1, %11, %9\n\tv_add_f32 %0, %0, %2\n\tv_sub_f32 %2, %11, %10\n\t"
                    "v_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\t"
                    "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2"
                    : "+v"(acc), "=&v"(t0), "=&v"(t1)
                    : "v"(xs[0]), "v"(xs[1]), "v"(xs[2]), "v"(xs[3]), "v"(xs[4]), "v"(xs[5]), "v"(xs[6]), "v"(xs[7]), "v"(k), "v"(big), "v"(c16));
            }
        }
    }
    if (acc + acc2 == 1.2345f) out[threadIdx.x] = 1;
}

// ---- the 8-slot window of the sort kernel, three ways (random bases, 4 windows per rep)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const unsigned* lds_cptr;
enum { W_B64X4, W_READ2X2, W_B128X2 };
template <int KIND>
__global__ __launch_bounds__(256) void k_win(unsigned* out, unsigned seed, int reps) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4096 + 16];
    for (int i = threadIdx.x; i < 4096 + 16; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    unsigned x = (threadIdx.x * 2654435761u + seed) ^ (blockIdx.x * 40503u);
    unsigned acc = 0;
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x = x * 1664525u + 1013904223u;
            const unsigned idx = x >> 20;  // 12 bits
            if (KIND == W_B64X4) {
                lds_cptr p = (lds_cptr)&lds[idx & ~1u];
                unsigned long long a0, a1, a2, a3;
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\t"
                             "ds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(p));
                acc += (unsigned)(a0 ^ a1 ^ a2 ^ a3) ^ (unsigned)((a0 ^ a1 ^ a2 ^ a3) >> 32);
            } else if (KIND == W_READ2X2) {
                lds_cptr p = (lds_cptr)&lds[idx & ~1u];
                v4u a0, a1;
                asm volatile("ds_read2_b64 %0, %2 offset1:1\n\tds_read2_b64 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(a0), "=&v"(a1) : "v"(p));
                acc += a0.x ^ a0.y ^ a0.z ^ a0.w ^ a1.x ^ a1.y ^ a1.z ^ a1.w;
            } else {
                lds_cptr p = (lds_cptr)&lds[idx & ~3u];
                v4u a0, a1;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(a0), "=&v"(a1) : "v"(p));
                acc += a0.x ^ a0.y ^ a0.z ^ a0.w ^ a1.x ^ a1.y ^ a1.z ^ a1.w;
            }
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// what HW_REG_LDS_ALLOC / HW_REG_HW_ID say for two 80 KiB workgroups per CU
__global__ __launch_bounds__(1024) void k_hwreg(unsigned* out) {
    extern __shared__ unsigned dyn[];
    dyn[threadIdx.x] = threadIdx.x;
    unsigned la, id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    __builtin_amdgcn_s_sleep(100);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = la; out[2 * blockIdx.x + 1] = id; }
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    launch();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < 3; i++) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3.0;
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const double clk = p.clockRate * 1e3;  // Hz
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %.0f MHz\n", p.name, cus, clk / 1e6);
    unsigned* out;
    CHK(hipMalloc(&out, 4096));
    const int blocks = cus * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
#define RUN_VALU(NAME, NINS)                                                                                     \
    {                                                                                                            \
        const double ms = time_ms([&] { hipLaunchKernelGGL(NAME, dim3(blocks), dim3(256), 0, 0, out, 1u); });     \
        const double inst = (double)blocks * 4 * REPS * 8.0 * NINS;                                              \
        printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", #NAME, ms,                         \
               ms * 1e-3 * clk * (cus * 4) / inst);                                                              \
    }
    RUN_VALU(k_mov, 8)
    RUN_VALU(k_add_u32, 8)
    RUN_VALU(k_and_b32, 8)
    RUN_VALU(k_min_u32, 8)
    RUN_VALU(k_lshlrev, 8)
    RUN_VALU(k_mul_f32, 8)
    RUN_VALU(k_fma_f32, 8)
    RUN_VALU(k_lshl_add, 8)
    RUN_VALU(k_add3, 8)
    RUN_VALU(k_bfe, 8)
    RUN_VALU(k_med3, 8)
    RUN_VALU(k_mad_u24, 8)
    RUN_VALU(k_mul_lo_u32, 8)
    RUN_VALU(k_cvt_i32_f32, 8)
    RUN_VALU(k_cvt_f32_u32, 8)
    RUN_VALU(k_cmp_addc, 16)
    RUN_VALU(k_cmp_cndmask, 16)
    RUN_VALU(k_sub_co, 8)
    RUN_VALU(k_pk_add_u16, 8)
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_pk_fma_f32, dim3(blocks), dim3(256), 0, 0, out, 1u); });
        const double inst = (double)blocks * 4 * REPS * 16.0 * 4;
        printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", "k_pk_fma_f32", ms, ms * 1e-3 * clk * (cus * 4) / inst);
    }
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_fma_f64, dim3(blocks), dim3(256), 0, 0, out, 1u); });
        const double inst = (double)blocks * 4 * REPS * 16.0 * 4;
        printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", "k_fma_f64", ms, ms * 1e-3 * clk * (cus * 4) / inst);
    }
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_quantile_f64, dim3(blocks), dim3(256), 0, 0, out, 1u); });
        const double inst = (double)blocks * 4 * REPS * 16.0 * 16;
        printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (cvt, mul, fma, cvt)\n", "k_quantile_f64", ms, ms * 1e-3 * clk * (cus * 4) / inst);
    }
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_count_cmp, dim3(blocks), dim3(256), 0, 0, out, 1u); });
        const double blk = (double)blocks * 4 * REPS * 4.0;
        printf("%-16s %8.3f ms  %6.2f cycles per 8-slot block per SIMD (32 instructions: cmp_lt + addc + cmp_le + addc per slot)\n", "k_count_cmp", ms, ms * 1e-3 * clk * (cus * 4) / blk);
    }
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_count_fma<1>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
        const double blk = (double)blocks * 4 * REPS * 4.0;
        printf("%-16s %8.3f ms  %6.2f cycles per 8-slot block per SIMD (24 instructions: sub + fma clamp (SGPR operand) + add per slot)\n", "k_count_fma", ms, ms * 1e-3 * clk * (cus * 4) / blk);
    }
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_count_fma<2>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
        const double blk = (double)blocks * 4 * REPS * 4.0;
        printf("%-16s %8.3f ms  %6.2f cycles per 8-slot block per SIMD (24 instructions: sub + fma (no clamp, VGPR operands) + add per slot)\n", "k_count_fma_nc", ms, ms * 1e-3 * clk * (cus * 4) / blk);
    }
    const int lreps = 1000;
#define RUN_LDS(KIND, NOPS)                                                                                            \
    {                                                                                                                  \
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_lds<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1u, lreps); }); \
        const double inst = (double)blocks * 4 * lreps * (double)NOPS;                                                 \
        printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction per CU (incl. 8 LCG steps of 2 VALU per rep)\n", #KIND, ms, \
               ms * 1e-3 * clk * cus / inst);                                                                          \
    }
    RUN_LDS(L_RD32_SEQ, 8)
    RUN_LDS(L_RD32, 8)
    RUN_LDS(L_RD_U16, 8)
    RUN_LDS(L_RD64, 8)
    RUN_LDS(L_RD128_SEQ, 8)
    RUN_LDS(L_RD128, 8)
    RUN_LDS(L_RD128X2, 8)
    RUN_LDS(L_WR32, 8)
    RUN_LDS(L_ATOM_RTN, 8)
    RUN_LDS(L_ATOM_NORTN, 8)
    RUN_LDS(L_BPERM, 8)
#define RUN_WIN(KIND)                                                                                                  \
    {                                                                                                                  \
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_win<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1u, lreps); }); \
        const double wins = (double)blocks * 4 * lreps * 4.0;                                                          \
        printf("%-16s %8.3f ms  %6.2f cycles per 8-slot window (wave) per CU (incl. the LCG step and 8 xor)\n", #KIND, ms, \
               ms * 1e-3 * clk * cus / wins);                                                                          \
    }
    {
        unsigned* d;
        const int nb = 1024;
        CHK(hipMalloc(&d, nb * 8));
        CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hwreg), hipFuncAttributeMaxDynamicSharedMemorySize, 81776));
        hipLaunchKernelGGL(k_hwreg, dim3(nb), dim3(1024), 81776, 0, d);
        std::vector<unsigned> h(nb * 2);
        CHK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
        printf("HW_REG_LDS_ALLOC / HW_ID of workgroups 0..7 and 512..515:");
        for (int i : {0, 1, 2, 3, 4, 5, 6, 7, 512, 513, 514, 515}) printf(" %08x/%08x", h[2 * i], h[2 * i + 1]);
        int nz = 0;
        for (int i = 0; i < nb; i++) nz += (h[2 * i] & 0xfff) != 0;
        printf("\nworkgroups with a non-zero LDS base field: %d of %d\n", nz, nb);
    }
    RUN_WIN(W_B64X4)
    RUN_WIN(W_READ2X2)
    RUN_WIN(W_B128X2)
    {
        unsigned* d;
        CHK(hipMalloc(&d, 128 * 4));
        hipLaunchKernelGGL(k_dpp_scan, dim3(1), dim3(64), 0, 0, d);
        unsigned h[128];
        CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        unsigned run = 0;
        int bad = 0;
        for (int i = 0; i < 64; i++) { run += h[i]; if (h[64 + i] != run) bad++; }
        printf("dpp inclusive scan: %s (%d lanes differ)\n", bad ? "WRONG" : "ok", bad);
        float* f = reinterpret_cast<float*>(d);
        hipLaunchKernelGGL(k_dpp_minmax, dim3(1), dim3(64), 0, 0, f);
        float hf[128];
        CHK(hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost));
        float mn = hf[0];
        for (int i = 1; i < 64; i++) mn = hf[i] < mn ? hf[i] : mn;
        bad = 0;
        for (int i = 0; i < 64; i++) if (hf[64 + i] != mn) bad++;
        printf("dpp wave minimum: %s (%d lanes differ)\n", bad ? "WRONG" : "ok", bad);
    }
    return 0;
}
