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
    // optional scaling epilogue of the small C x C products of the linear modes (linalg.hip):
    //   OUT = alpha * alpha_seg[seg] * acc + diag * (m == i)        (epi = 0: untouched, the hot-loop arithmetic)
    // sym = 1 (64 x 64 tiles, square output): the product is symmetric in exact arithmetic — only the tiles on or above
    // the diagonal are computed and every element is stored together with its mirror image, so the output is EXACTLY
    // symmetric (half the flops; what keeps the Newton-Schulz iteration of linalg.hip on its stable branch).
    int epi; float alpha; const float* alpha_seg; float diag; int sym;
    int prof_cls;  // KC_GEMM for the feature-map GEMMs, KC_SMALL_GEMM for the C x C products of linalg.hip
};

// internal launcher behind optex_gemm_tn (gemm.hip): `a` fully filled except tiles_*; layouts are OPTEX_*_MAJOR
int gemm_tn_launch(GemmArgs& a, int b_layout, int o_layout, hipStream_t st);

}  // namespace optex
