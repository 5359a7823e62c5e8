// Diagnostic (not part of the library): the R-stationary rotation GEMM (csrc/gemm_rs.hip) alone at the hot-loop shape,
// timed with HIP events, against the LDS-tiled kernels of gemm.hip on the same data; built in variants by scripts/Makefile
// (-DRS_DEPTH_VALUE=..., -DRS_PROBE_NOLOAD).
//   scripts/gemm_rs_probe_<variant>.bin [n_seg] [n] [M] [K] [reps] [1 = LDS-tiled kernels] [data: 0 gaussian, 1 zeros, 2 ReLU-like]
// (links csrc/gemm_rs.hip and csrc/gemm.hip as separate objects, compiled like the library's: scripts/Makefile)
#include "../optimaltextures_amd/csrc/gemm_args.h"
#ifndef RS_DEPTH_VALUE
#define RS_DEPTH_VALUE 16
#endif
namespace optex { int device_cu_count(); }

#include <random>
#include <vector>

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 64;
    const long n = argc > 2 ? atol(argv[2]) : 16384;
    const int M = argc > 3 ? atoi(argv[3]) : 256, K = argc > 4 ? atoi(argv[4]) : 256;
    const int reps = argc > 5 ? atoi(argv[5]) : 20;
    const bool old = argc > 6 && atoi(argv[6]) == 1;
    const int mode = argc > 7 ? atoi(argv[7]) : 0;  // feature-map data: 0 = gaussian, 1 = zeros, 2 = max(gaussian, 0)
    const int rowstat = argc > 8 ? atoi(argv[8]) : 0;  // 1 = min / max, 2 = sums of the output rows in the epilogue (GemmArgs::rowstat)
    optex::gemm_rs_enabled = !old;
    optex::gemm_rs_force = !old;
    std::vector<float> hb((size_t)S * K * n), ha((size_t)K * M);
    std::mt19937 g(1);
    std::normal_distribution<float> d(0.f, 1.f);
    for (auto& v : hb) { const float x = d(g); v = mode == 1 ? 0.f : (mode == 2 ? (x > 0.f ? x : 0.f) : x); }
    for (auto& v : ha) v = d(g) / 16.f;
    float *A, *B, *O;
    (void)hipMalloc(&A, ha.size() * 4); (void)hipMalloc(&B, hb.size() * 4); (void)hipMalloc(&O, (size_t)S * M * n * 4);
    (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    optex::GemmArgs a{};
    a.At = A; a.lda = M; a.at_ss = 0; a.B = B; a.ldb = n; a.b_ss = (long)K * n; a.O = O; a.ldo = n; a.o_ss = (long)M * n;
    a.M = M; a.K = K; a.n = n; a.n_seg = S; a.alpha = 1.f; a.prof_cls = optex::KC_GEMM;
    float *ra = nullptr, *rb = nullptr;
    if (rowstat) {
        (void)hipMalloc(&ra, (size_t)S * (n / 64) * M * 4); (void)hipMalloc(&rb, (size_t)S * (n / 64) * M * 4);
        a.rowstat = rowstat; a.rs_a = ra; a.rs_b = rb;
    }
    const int CM = OPTEX_CHANNEL_MAJOR;
    const int n_cu = optex::device_cu_count();
    if (!optex::gemm_rs_supported(a, n_cu)) { printf("shape not supported\n"); return 1; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) optex::gemm_tn_launch(a, CM, CM, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) optex::gemm_tn_launch(a, CM, CM, 0);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> ho((size_t)M * 64);
    (void)hipMemcpy(ho.data(), O, ho.size() * 4, hipMemcpyDeviceToHost);
    double chk = 0;
    for (float v : ho) chk += v;
    const double us = 1e3 * ms / reps, tf = 2.0 * M * K * (double)n * S / (us * 1e6);
    if (rowstat) {
        std::vector<float> hr((size_t)M * 4);
        (void)hipMemcpy(hr.data(), ra, hr.size() * 4, hipMemcpyDeviceToHost);
        for (float v : hr) chk += v;
        printf("[rowstat %d] ", rowstat);
    }
    printf("%s ring %d%s, %s data: S=%d n=%ld M=%d K=%d  %.1f us  %.1f TFLOP/s  (%.3f of 157.3)  checksum %.6f\n", old ? "LDS-tiled kernel," : "R-stationary kernel,", RS_DEPTH_VALUE,
#ifdef RS_PROBE_NOLOAD
           " NOLOAD",
#else
           "",
#endif
           mode == 1 ? "all-zero" : (mode == 2 ? "ReLU-like (half zeros)" : "gaussian"), S, n, M, K, us, tf, tf / 157.3, chk);
    return 0;
}
