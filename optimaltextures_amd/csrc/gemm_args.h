// gemm_args.h — argument block of the GEMM kernels (gemm.hip)
#pragma once
#include "optex_common.h"

namespace optex {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* At; long lda, at_ss;
    const float* B;  long ldb, b_ss;
    float* O;        long ldo, o_ss;
    int M, K; long n; int n_seg;
    const float* bsub; long bsub_ss;
    const float* badd; long badd_ss;
    const float* content; float strength;
    int tiles_m, tiles_n;
    int a_vec;  // At rows take 16-byte loads (set by gemm_tn_launch)
    // optional scaling epilogue of the small C x C products of the linear modes (linalg.hip):
    //   OUT = alpha * alpha_seg[seg] * acc + diag * (m == i)        (epi = 0: untouched, the hot-loop arithmetic)
    // sym = 1 (64 x 64 tiles, square output): the product is symmetric in exact arithmetic — only the tiles on or above
    // the diagonal are computed and every element is stored together with its mirror image, so the output is EXACTLY
    // symmetric (half the flops; what keeps the Newton-Schulz iteration of linalg.hip on its stable branch).
    int epi; float alpha; const float* alpha_seg; float diag; int sym;
    int prof_cls;  // KC_GEMM for the feature-map GEMMs, KC_SMALL_GEMM for the C x C products of linalg.hip
    // optional per-row statistics of the OUTPUT, taken from the accumulators before they are stored (hot-loop kernel only,
    // gemm_rowstat_supported): 1 = min and max, 2 = sum of every output row over the block's pixel columns, one partial
    // per (pixel tile, wave column):  rs_a / rs_b [n_seg][gemm_rowstat_parts][M]  (min / sum in rs_a, max in rs_b).
    // Saves the separate pass over the rotated map that col_minmax_kernel / col_mean_kernel would make.
    int rowstat; float* rs_a; float* rs_b;
    // optional device-side switch of a launch that was enqueued before anyone knew whether it is needed (the later
    // Newton-Schulz iterations of linalg.hip): the kernel returns at once if live_idx >= *live_until.  gemm_tn_kernel only.
    const int* live_until = nullptr; int live_idx = 0;
    // optional second operand set (gemm_tn_kernel only): segments half .. n_seg - 1 run At2 / B2 -> O2 (same shapes, strides and
    // epilogue, scaled by alpha_seg2) — two independent batches of products in ONE launch (the Y W and W Z of a Newton-Schulz
    // iteration, linalg.hip).  half = 0: off.
    int half = 0; const float* At2 = nullptr; const float* B2 = nullptr; float* O2 = nullptr; const float* alpha_seg2 = nullptr;
};

// the launch of `a` (channel-major in and out) takes the hot-loop kernel, i.e. a.rowstat is honoured
bool gemm_rowstat_supported(const GemmArgs& a);
// partials per (segment, row) a rowstat launch writes for n pixels
int gemm_rowstat_parts(long n);

// the R-stationary, LDS-free hot-loop kernel (gemm_rs.hip): channel-major in / out, 64 < M <= 256, 64 <= K <= 256, n % 64 == 0
// (bsub: M <= 192 only; row statistics: without bsub / badd / content only — gemm_rs_supported is the authority)
bool gemm_rs_supported(const GemmArgs& a, int n_cu);
int gemm_rs_launch(const GemmArgs& a, int n_cu, hipStream_t st);
extern bool gemm_rs_enabled, gemm_rs_force;

// internal launcher behind optex_gemm_tn (gemm.hip): `a` fully filled except tiles_*; layouts are OPTEX_*_MAJOR
int gemm_tn_launch(GemmArgs& a, int b_layout, int o_layout, hipStream_t st);

}  // namespace optex
