// Diagnostic (not part of the library): s_memtime timeline of the two hot rotation-GEMM kernels — the R-stationary
// gemm_rs_kernel<4, 64> (one stamp per 4 k-steps = 64 MFMAs, and around the epilogue) and the LDS-tiled gemm16_cm_kernel
// (four stamps per 16-deep K chunk: global loads issued / MFMAs issued / LDS refilled / barrier passed).  Built with
// -DOPTEX_TIMELINE (csrc/timeline.h): every wavefront appends its stamps to a slab through the scalar store path.
//   scripts/gemm_timeline_probe.bin <out.bin> [kind: 0 = R-stationary, 1 = LDS-tiled] [n_seg] [n] [rowstat 0/1/2] [data: 0 gaussian, 1 zeros] [reps]
// The raw slabs go to <out.bin> (header: 8 x int64 = magic, kind, waves, words per wave, n_seg, n, launch us x 1000, reps);
// scripts/gemm_timeline_report.py turns them into profiles/r05_gemm_timeline.md.
// (links csrc/gemm_rs.hip and csrc/gemm.hip as separate objects, compiled like the library's + -DOPTEX_TIMELINE: scripts/Makefile)
#include "../optimaltextures_amd/csrc/gemm_args.h"
namespace optex {
int device_cu_count();
void tl_set_rs(unsigned long long* buf, int words);    // csrc/timeline.h, TL_DEFINE_SETTER at the end of gemm_rs.hip / gemm.hip
void tl_set_lds(unsigned long long* buf, int words);
}

#include <random>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s out.bin [kind] [n_seg] [n] [rowstat] [data] [reps]\n", argv[0]); return 2; }
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    const int S = argc > 3 ? atoi(argv[3]) : 64;
    const long n = argc > 4 ? atol(argv[4]) : 16384;
    const int rowstat = argc > 5 ? atoi(argv[5]) : 0;
    const int mode = argc > 6 ? atoi(argv[6]) : 0;
    const int reps = argc > 7 ? atoi(argv[7]) : 5;
    const int M = 256, K = 256;
    optex::gemm_rs_enabled = kind == 0;
    optex::gemm_rs_force = kind == 0;
    std::vector<float> hb((size_t)S * K * n), ha((size_t)K * M);
    std::mt19937 g(1);
    std::normal_distribution<float> d(0.f, 1.f);
    for (auto& v : hb) v = mode == 1 ? 0.f : d(g);
    for (auto& v : ha) v = d(g) / 16.f;
    float *A, *B, *O, *rsa, *rsb;
    (void)hipMalloc(&A, ha.size() * 4); (void)hipMalloc(&B, hb.size() * 4); (void)hipMalloc(&O, (size_t)S * M * n * 4);
    (void)hipMalloc(&rsa, (size_t)S * (n / 64) * M * 4); (void)hipMalloc(&rsb, (size_t)S * (n / 64) * M * 4);
    (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    optex::GemmArgs a{};
    a.At = A; a.lda = M; a.at_ss = 0; a.B = B; a.ldb = n; a.b_ss = (long)K * n; a.O = O; a.ldo = n; a.o_ss = (long)M * n;
    a.M = M; a.K = K; a.n = n; a.n_seg = S; a.alpha = 1.f; a.prof_cls = optex::KC_GEMM;
    a.rowstat = rowstat; a.rs_a = rsa; a.rs_b = rsb;
    const int CM = OPTEX_CHANNEL_MAJOR;
    const int n_cu = optex::device_cu_count();
    long waves, words;
    if (kind == 0) {
        if (!optex::gemm_rs_supported(a, n_cu)) { printf("shape not supported by the R-stationary kernel\n"); return 1; }
        const long tiles = (n / 64) * S, per = (tiles + n_cu - 1) / n_cu;
        waves = 4L * (n_cu < tiles ? n_cu : tiles);
        words = 4 + per * 18 + 2;
    } else {
        waves = 8L * (n / 128) * S;
        words = 4 + 16 * 4 + 3;
    }
    unsigned long long* tl;
    (void)hipMalloc(&tl, (size_t)waves * words * 8);
    (void)hipMemset(tl, 0, (size_t)waves * words * 8);
    const int w32 = (int)words;
    optex::tl_set_rs(tl, w32);
    optex::tl_set_lds(tl, w32);
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
    const double us = 1e3 * ms / reps, tf = 2.0 * M * K * (double)n * S / (us * 1e6);
    printf("%s rowstat %d, %s data: S=%d n=%ld  %.1f us per launch (stamped build)  %.1f TFLOP/s  waves %ld words %ld\n",
           kind == 0 ? "R-stationary" : "LDS-tiled", rowstat, mode == 1 ? "all-zero" : "gaussian", S, n, us, tf, waves, words);
    std::vector<unsigned long long> h((size_t)waves * words);
    (void)hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost);
    // LDS-tiled: 64 Ki wavefronts at the bench shape — keep every 16th workgroup (they are statistically alike)
    long keep_every = 1;
    if (kind == 1 && waves * words * 8 > (8L << 20)) keep_every = 16;
    FILE* f = fopen(argv[1], "wb");
    if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
    const long wg_waves = kind == 0 ? 4 : 8;
    long kept = 0;
    for (long w = 0; w < waves; w++) kept += ((w / wg_waves) % keep_every) == 0;
    const long long hdr[8] = {0x4c54474dLL, kind, kept, words, S, n, (long long)(us * 1000.0), reps};
    fwrite(hdr, 8, 8, f);
    for (long w = 0; w < waves; w++)
        if (((w / wg_waves) % keep_every) == 0) fwrite(&h[(size_t)w * words], 8, words, f);
    fclose(f);
    return 0;
}
