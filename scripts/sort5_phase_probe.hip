// Diagnostic (not part of the library): wall_clock64 stamps of thread 0 at every barrier of rank_match5_kernel
// (-DOPTEX_SORT_PROBE), averaged over the columns of a [64 x 256] launch.  Built by scripts/Makefile.
//   scripts/sort5_phase_probe.bin [n ...]
#include "../optimaltextures_amd/csrc/sort_rank5.hip"

#include <algorithm>
#include <random>
#include <vector>

int main(int argc, char** argv) {
    const int C = 256, S = 64, ncols = C * S;
    std::vector<long> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back(atol(argv[i]));
    if (sizes.empty()) sizes = {16384, 12544, 9216, 6400, 4096};
    static const char* names[13] = {"(start)", "clear + keys arrive + nf", "sample histogram", "equalise (one wave)", "bucket + count",
                                    "scan: row sums", "scan: group starts", "decode (word + group start)", "place colliders",
                                    "mates + next keys requested", "ties + stage source", "pick + store", "end barrier"};
    for (long n : sizes) {
        const long ns = n * 3 / 4;
        std::vector<float> h((size_t)ncols * n), hs((size_t)C * ns), lo(ncols), hi(ncols);
        std::mt19937 g(5);
        std::normal_distribution<float> d(0.f, 1.f);
        for (auto& v : h) v = d(g);
        for (auto& v : hs) v = d(g);
        for (int c = 0; c < C; c++) std::sort(hs.begin() + (size_t)c * ns, hs.begin() + (size_t)(c + 1) * ns);
        for (int c = 0; c < ncols; c++) {
            auto mm = std::minmax_element(h.begin() + (size_t)c * n, h.begin() + (size_t)(c + 1) * n);
            lo[c] = *mm.first;
            hi[c] = *mm.second;
        }
        float *x, *ss, *out, *dlo, *dhi;
        int* flags;
        long long* probe;
        hipMalloc(&x, h.size() * 4); hipMalloc(&out, h.size() * 4); hipMalloc(&ss, hs.size() * 4);
        hipMalloc(&flags, ncols * 4); hipMalloc(&dlo, ncols * 4); hipMalloc(&dhi, ncols * 4);
        hipMalloc(&probe, (size_t)ncols * 16 * 8);
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(ss, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dlo, lo.data(), ncols * 4, hipMemcpyHostToDevice);
        hipMemcpy(dhi, hi.data(), ncols * 4, hipMemcpyHostToDevice);
        hipMemset(flags, 0, ncols * 4);
        optex::SortArgs a{};
        a.keys = x; a.ld = n; a.ss = (long)C * n; a.n = n; a.C = C; a.x_n_seg = S;
        a.src_sorted = ss; a.ns = ns; a.src_n_seg = 1;
        a.out = out; a.ldo = n; a.oss = (long)C * n; a.out_vec = 1;
        a.flags = flags; a.inv_2nt = 1.0 / (2.0 * n); a.ncols = ncols;
        a.rng_lo = dlo; a.rng_hi = dhi; a.probe = probe;
        if (!optex::rank5_supported(a)) { printf("n = %ld: not supported\n", n); continue; }
        for (int it = 0; it < 2; it++) { optex::launch_rank5(a, ncols, 0); hipDeviceSynchronize(); }
        std::vector<long long> p((size_t)ncols * 16);
        hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
        // wall_clock64: 100 MHz constant clock.  Skip each workgroup's first column (cold start).
        double sum[13] = {0}, tot = 0;
        long cnt = 0;
        for (int col = 256; col < ncols; col++) {
            const long long* q = &p[(size_t)col * 16];
            for (int i = 1; i <= 12; i++) sum[i] += (double)(q[i] - q[i - 1]);
            // column to column: stamp 0 of this column against stamp 0 of the workgroup's previous one
            tot += (double)(q[0] - p[(size_t)(col - 256) * 16]);
            cnt++;
        }
        printf("n = %ld: %.2f us per column per workgroup (stamp 0 to stamp 0), %ld columns\n", n, tot / cnt * 0.01, cnt);
        for (int i = 1; i <= 12; i++) printf("  %-32s %6.2f us  %5.1f %%\n", names[i], sum[i] / cnt * 0.01, 100.0 * sum[i] / tot);
        hipFree(x); hipFree(out); hipFree(ss); hipFree(flags); hipFree(dlo); hipFree(dhi); hipFree(probe);
    }
    return 0;
}
