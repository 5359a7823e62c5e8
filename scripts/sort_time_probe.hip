// Diagnostic (not part of the library): rank_match4_kernel alone at the five pass sizes of the 512^2 schedule, HIP-event
// timing, with and without a caller-given range (SortArgs::rng_lo), results checked against a host stable sort on sampled
// columns.  Built by scripts/Makefile.
//   scripts/sort_time_probe.bin [reps]
#include "../optimaltextures_amd/csrc/sort_rank4.hip"

#include <algorithm>
#include <numeric>
#include <random>
#include <vector>

int main(int argc, char** argv) {
    const int C = 256, S = 64, ncols = C * S, reps = argc > 1 ? atoi(argv[1]) : 5;
    const long sizes[5][2] = {{16384, 12288}, {12544, 9408}, {9216, 6912}, {6400, 4800}, {4096, 3072}};
    const double weight[5] = {8, 9, 10, 12, 13};  // iterations per pass size (relu3_1, 512^2)
    double tot_us[2] = {0, 0}, tot_bytes = 0;
    for (int si = 0; si < 5; si++) {
        const long n = sizes[si][0], ns = sizes[si][1];
        std::vector<float> h((size_t)ncols * n), hs((size_t)C * ns), lo(ncols), hi(ncols);
        std::mt19937 g(1 + si);
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
        hipMalloc(&x, h.size() * 4); hipMalloc(&out, h.size() * 4); hipMalloc(&ss, hs.size() * 4);
        hipMalloc(&flags, ncols * 4); hipMalloc(&dlo, ncols * 4); hipMalloc(&dhi, ncols * 4);
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(ss, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dlo, lo.data(), ncols * 4, hipMemcpyHostToDevice);
        hipMemcpy(dhi, hi.data(), ncols * 4, hipMemcpyHostToDevice);
        optex::SortArgs a{};
        a.keys = x; a.ld = n; a.ss = (long)C * n; a.n = n; a.C = C; a.x_n_seg = S;
        a.src_sorted = ss; a.ns = ns; a.src_n_seg = 1;
        a.out = out; a.ldo = n; a.oss = (long)C * n; a.out_vec = 1;
        a.flags = flags; a.inv_2nt = 1.0 / (2.0 * n); a.ncols = ncols;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rg = 0; rg < 2; rg++) {
            a.rng_lo = rg ? dlo : nullptr;
            a.rng_hi = rg ? dhi : nullptr;
            float best = 1e30f, ms = 0.f;
            for (int it = 0; it < reps + 1; it++) {
                hipMemset(flags, 0, ncols * 4);
                hipEventRecord(e0, 0);
                optex::launch_rank4(optex::SORT_MATCH, a, ncols, 0);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
                if (it > 0 && ms < best) best = ms;
            }
            std::vector<int> fl(ncols);
            hipMemcpy(fl.data(), flags, ncols * 4, hipMemcpyDeviceToHost);
            const int nflag = std::accumulate(fl.begin(), fl.end(), 0);
            // host check of 6 columns: out[pixel of rank i] = sorted_source[floor((2 i + 1) ns / (2 n))]
            int bad = 0;
            std::vector<float> got(n);
            std::vector<int> idx(n);
            for (int k = 0; k < 6; k++) {
                const int col = (int)(((long)k * 2731 + 17) % ncols);
                hipMemcpy(got.data(), out + (size_t)col * n, n * 4, hipMemcpyDeviceToHost);
                std::iota(idx.begin(), idx.end(), 0);
                const float* kc = h.data() + (size_t)col * n;
                std::stable_sort(idx.begin(), idx.end(), [&](int p, int q) { return kc[p] < kc[q]; });
                const float* sc = hs.data() + (size_t)(col % C) * ns;
                for (long i = 0; i < n; i++) {
                    const long qi = ((2 * i + 1) * ns) / (2 * n);
                    if (got[idx[i]] != sc[qi]) bad++;
                }
            }
            const double bytes = 12.0 * n * ncols;
            printf("n = %5ld ns = %5ld range %-6s %8.1f us  %6.2f TB/s  %.3f of 8 TB/s   flagged %d, mismatches on 6 sampled columns %d\n", n,
                   ns, rg ? "given" : "own", best * 1e3, bytes / (best * 1e9), bytes / (best * 1e9) / 8.0, nflag, bad);
            tot_us[rg] += weight[si] * best * 1e3;
            if (rg == 0) tot_bytes += weight[si] * bytes;
        }
        hipFree(x); hipFree(out); hipFree(ss); hipFree(flags); hipFree(dlo); hipFree(dhi);
    }
    for (int rg = 0; rg < 2; rg++)
        printf("schedule-weighted (13/12/10/9/8 iterations), range %-6s: %.2f ms per step, %.2f TB/s = %.3f of HBM peak\n",
               rg ? "given" : "own", tot_us[rg] * 1e-3, tot_bytes / (tot_us[rg] * 1e6), tot_bytes / (tot_us[rg] * 1e6) / 8.0);
    return 0;
}
