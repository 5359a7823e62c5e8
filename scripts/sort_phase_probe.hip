// Diagnostic (not part of the library): per-phase wall-clock breakdown of rank_columns_kernel on the headline shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOPTEX_SORT_PROBE scripts/sort_phase_probe.hip \
//         optimaltextures_amd/csrc/api.hip -o /tmp/sort_probe && /tmp/sort_probe
#include "../optimaltextures_amd/csrc/sort.hip"

#include <random>
#include <vector>

int main() {
    const int C = 256, S = 32, ncols = C * S;
    const long n = 16384;
    std::vector<float> h((size_t)ncols * n);
    std::mt19937 g(1);
    std::normal_distribution<float> d(0.f, 1.f);
    for (auto& v : h) v = d(g);
    float *x, *ok; uint32_t* oi; int* flags; long long* probe;
    hipMalloc(&x, h.size() * 4); hipMalloc(&ok, h.size() * 4); hipMalloc(&oi, h.size() * 4);
    hipMalloc(&flags, ncols * 4); hipMalloc(&probe, (size_t)ncols * 16 * 8);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    optex::SortArgs a{};
    a.keys = x; a.ld = n; a.ss = (long)C * n; a.n = n; a.C = C; a.x_n_seg = S;
    a.out_keys = ok; a.out_idx = oi; a.flags = flags; a.inv_2nt = 1.0 / (2.0 * n); a.probe = probe; a.ncols = ncols;
    auto kern = optex::rank_columns_kernel<16, optex::SORT_EMIT>;
    const size_t lds = optex::rank_lds_bytes<16>(false);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int it = 0; it < 3; it++) {
        hipMemset(flags, 0, ncols * 4);
        hipLaunchKernelGGL(kern, dim3(ncols), dim3(1024), lds, 0, a);
        hipDeviceSynchronize();
    }
    std::vector<long long> p((size_t)ncols * 16);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"load+minmax", "coarse hist", "equalise", "fine bucket+count", "scan", "place", "rank (6b)",
                           "big buckets", "output"};
    double tot[9] = {0};
    for (int c = 0; c < ncols; c++)
        for (int i = 0; i < 9; i++) tot[i] += (double)(p[(size_t)c * 16 + i + 1] - p[(size_t)c * 16 + i]);
    double all = 0;
    for (int i = 0; i < 9; i++) all += tot[i];
    printf("wall_clock64 ticks per column (100 MHz clock), mean over %d columns; total %.1f ticks = %.2f us\n", ncols,
           all / ncols, all / ncols / 100.0);
    for (int i = 0; i < 9; i++) printf("  %-20s %8.1f ticks  %5.1f %%\n", names[i], tot[i] / ncols, 100.0 * tot[i] / all);
    return 0;
}
