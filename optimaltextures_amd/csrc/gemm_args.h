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
};

}  // namespace optex
