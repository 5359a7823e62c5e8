// Diagnostic (not part of the library): per-phase wall-clock breakdown of rank_match_kernel (sort_rank2.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOPTEX_SORT_PROBE scripts/sort_rank2_probe.hip \
//         optimaltextures_amd/csrc/api.hip -o /tmp/r2probe && /tmp/r2probe
#include "../optimaltextures_amd/csrc/sort_rank2.hip"

#include <algorithm>
#include <random>
#include <vector>

int main() {
    const int C = 256, S = 32, maxcols = C * S;
    const long n = 16384;
    std::vector<float> h((size_t)maxcols * n);
    std::mt19937 g(1);
    std::normal_distribution<float> d(0.f, 1.f);
    for (auto& v : h) v = d(g);
    std::vector<float> hs((size_t)C * n);
    for (int c = 0; c < C; c++) for (long i = 0; i < n; i++) hs[(size_t)c * n + i] = (float)i;
    float *x, *out, *ss; int* flags; long long* probe;
    hipMalloc(&x, h.size() * 4); hipMalloc(&out, h.size() * 4); hipMalloc(&ss, hs.size() * 4);
    hipMalloc(&flags, maxcols * 4); hipMalloc(&probe, (size_t)maxcols * 16 * 8);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ss, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    auto kern = optex::rank_match_kernel<16, true>;
    const size_t lds = optex::R2<16>::LDS;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 1024, lds);
    printf("LDS %zu B per workgroup, occupancy %d workgroups per CU\n", lds, occ);
    const char* names[] = {"load+minmax", "coarse hist", "equalise", "bucket+count", "scan+bitmap", "place(+big)", "rank slots", "queued slots",
                           "fetch+scatter", "store issue"};
    for (int ncols : {256, 512, maxcols}) {
        optex::SortArgs a{};
        a.keys = x; a.ld = n; a.ss = (long)C * n; a.n = n; a.C = C; a.x_n_seg = S;
        a.src_sorted = ss; a.ns = n; a.src_n_seg = 1; a.out = out; a.ldo = n; a.oss = (long)C * n;
        a.flags = flags; a.inv_2nt = 1.0 / (2.0 * n); a.probe = probe; a.ncols = ncols; a.out_vec = 1;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int it = 0; it < 3; it++) {
            hipMemset(flags, 0, maxcols * 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(ncols), dim3(1024), lds, 0, a);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<long long> p((size_t)ncols * 16);
        hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
        double tot[10] = {0}, all = 0;
        for (int c = 0; c < ncols; c++)
            for (int i = 0; i < 10; i++) tot[i] += (double)(p[(size_t)c * 16 + i + 1] - p[(size_t)c * 16 + i]);
        for (int i = 0; i < 10; i++) all += tot[i];
        printf("%d columns: kernel %.1f us = %.2f TB/s algorithmic; per column %.2f us in-kernel (100 MHz ticks)\n", ncols,
               ms * 1e3, 12.0 * n * ncols / (ms * 1e-3) / 1e12, all / ncols / 100.0);
        for (int i = 0; i < 10; i++) printf("  %-16s %8.1f ticks  %5.1f %%\n", names[i], tot[i] / ncols, 100.0 * tot[i] / all);
    }
    return 0;
}
