// Diagnostic (not part of the library): per-phase wall-clock breakdown of rank_match4_kernel, workgroups per CU as in the
// library.  Built by scripts/Makefile (-DOPTEX_SORT_PROBE):  make -C scripts sort_rank_probe.bin && scripts/sort_rank_probe.bin [n] [ns] [rg]
// (rg: the column ranges are handed to the kernel like optex_ot_loop does — SortArgs::rng_lo / rng_hi)
#include "../optimaltextures_amd/csrc/sort_rank4.hip"
#define LAUNCH(items, a, ncols, st) optex::launch_rank4(optex::SORT_MATCH, a, ncols, st)

#include <algorithm>
#include <random>
#include <vector>

int main(int argc, char** argv) {
    const int C = 256, S = 64, ncols = C * S;
    const long n = argc > 1 ? atol(argv[1]) : 16384, ns = argc > 2 ? atol(argv[2]) : 12288;
    const bool rg = argc > 3;
    std::vector<float> h((size_t)ncols * n), hs((size_t)C * ns);
    std::mt19937 g(1);
    std::normal_distribution<float> d(0.f, 1.f);
    for (auto& v : h) v = d(g);
    for (auto& v : hs) v = d(g);
    for (int c = 0; c < C; c++) std::sort(hs.begin() + (size_t)c * ns, hs.begin() + (size_t)(c + 1) * ns);
    float *x, *ss, *out; int* flags; long long* probe;
    hipMalloc(&x, h.size() * 4); hipMalloc(&out, h.size() * 4); hipMalloc(&ss, hs.size() * 4);
    hipMalloc(&flags, ncols * 4); hipMalloc(&probe, (size_t)ncols * 16 * 8);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ss, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    optex::SortArgs a{};
    a.keys = x; a.ld = n; a.ss = (long)C * n; a.n = n; a.C = C; a.x_n_seg = S;
    a.src_sorted = ss; a.ns = ns; a.src_n_seg = 1;
    a.out = out; a.ldo = n; a.oss = (long)C * n; a.out_vec = 1;
    a.flags = flags; a.inv_2nt = 1.0 / (2.0 * n); a.probe = probe; a.ncols = ncols;
    if (rg) {
        std::vector<float> lo(ncols), hi(ncols);
        for (int c = 0; c < ncols; c++) {
            auto mm = std::minmax_element(h.begin() + (size_t)c * n, h.begin() + (size_t)(c + 1) * n);
            lo[c] = *mm.first;
            hi[c] = *mm.second;
        }
        float *dlo, *dhi;
        hipMalloc(&dlo, ncols * 4); hipMalloc(&dhi, ncols * 4);
        hipMemcpy(dlo, lo.data(), ncols * 4, hipMemcpyHostToDevice);
        hipMemcpy(dhi, hi.data(), ncols * 4, hipMemcpyHostToDevice);
        a.rng_lo = dlo;
        a.rng_hi = dhi;
    }
    const int items = n <= 2048 ? 2 : n <= 4096 ? 4 : n <= 8192 ? 8 : n <= 12288 ? 12 : 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int it = 0; it < 4; it++) {
        hipMemset(flags, 0, ncols * 4);
        hipEventRecord(e0, 0);
        LAUNCH(items, a, ncols, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> p((size_t)ncols * 16);
    std::vector<int> fl(ncols);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(fl.data(), flags, ncols * 4, hipMemcpyDeviceToHost);
    int nflag = 0;
    for (int f : fl) nflag += f;
    const char* names[] = {"load+minmax", "coarse hist", "equalise", "bucket+count", "scan", "place (+big)", "rank", "queue",
                           "stage+resolve", "gather+store"};
    double tot[10] = {0};
    for (int c = 0; c < ncols; c++)
        for (int i = 0; i < 10; i++) tot[i] += (double)(p[(size_t)c * 16 + i + 1] - p[(size_t)c * 16 + i]);
    double all = 0;
    for (int i = 0; i < 10; i++) all += tot[i];
    printf("n = %ld, ns = %ld, range %s: kernel %.1f us (with probes), %d flagged columns\n", n, ns, rg ? "given" : "own", ms * 1e3, nflag);
    printf("wall_clock64 ticks per column (100 MHz clock), mean over %d columns; total %.1f ticks = %.2f us\n", ncols,
           all / ncols, all / ncols / 100.0);
    for (int i = 0; i < 10; i++) printf("  %-20s %8.1f ticks  %5.1f %%\n", names[i], tot[i] / ncols, 100.0 * tot[i] / all);
    return 0;
}
