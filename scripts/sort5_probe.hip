// Diagnostic (not part of the library): rank_match5_kernel (sort_rank5.hip, round 6) against rank_match4_kernel at the five pass
// sizes of the 512^2 schedule — HIP-event timing, every sampled column checked against a host stable sort — and on adversarial
// columns (ties, outliers, constant, skewed).  Built by scripts/Makefile.
//   scripts/sort5_probe.bin [reps] [check_columns]
#include "../optimaltextures_amd/csrc/sort_rank4.hip"
#include "../optimaltextures_amd/csrc/sort_rank5.hip"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

static int check_columns(const std::vector<float>& h, const std::vector<float>& hs, const float* out, const std::vector<int>& fl, long n,
                         long ns, int C, int ncols, int ncheck, int* nchecked) {
    int bad = 0;
    std::vector<float> got(n);
    std::vector<int> idx(n);
    *nchecked = 0;
    for (int k = 0; k < ncheck; k++) {
        const int col = (int)(((long)k * 2731 + 17) % ncols);
        if (fl[col]) continue;  // flagged: the radix sweep's column
        (*nchecked)++;
        hipMemcpy(got.data(), out + (size_t)col * n, n * 4, hipMemcpyDeviceToHost);
        std::iota(idx.begin(), idx.end(), 0);
        const float* kc = h.data() + (size_t)col * n;
        // IEEE totalOrder: -0 < +0 (the specification); NaN-free here
        auto key = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        std::stable_sort(idx.begin(), idx.end(), [&](int p, int q) { return key(kc[p]) < key(kc[q]); });
        const float* sc = hs.data() + (size_t)(col % C) * ns;
        for (long i = 0; i < n; i++) {
            const long qi = ((2 * i + 1) * ns) / (2 * n);
            if (memcmp(&got[idx[i]], &sc[qi], 4) != 0) bad++;
        }
    }
    return bad;
}

int main(int argc, char** argv) {
    const int C = 256, S = 64, ncols = C * S, reps = argc > 1 ? atoi(argv[1]) : 5, ncheck = argc > 2 ? atoi(argv[2]) : 24;
    const long only_n = argc > 3 ? atol(argv[3]) : 0;  // one size, ns = 3 n / 4, no adversarial part (counter runs)
    // ns = 0.75 n (the harness of rounds 4 / 5) and ns = 23 n / 16 (bench.py: the style image is 736 x 512)
    const long sizes[5] = {16384, 12544, 9216, 6400, 4096};
    const double weight[5] = {8, 9, 10, 12, 13};  // iterations per pass size (relu3_1, 512^2)
    for (int ratio = 0; ratio < (only_n ? 1 : 2); ratio++) {
        double tot_us[3] = {0, 0, 0}, tot_bytes = 0;
        for (int si = 0; si < 5; si++) {
            const long n = sizes[si], ns = ratio == 0 ? n * 3 / 4 : n * 23 / 16;
            if (only_n && n != only_n) continue;
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
            a.rng_lo = dlo; a.rng_hi = dhi;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int kern = 0; kern < 3; kern++) {
                if (kern == 2 && !optex::rank5w_supported(optex::SORT_MATCH, a)) { printf("n = %5ld ns = %5ld rank5w: not supported\n", n, ns); continue; }
#ifdef R5_PERSISTENT_VARIANT
                if (kern == 1 && !optex::rank5_supported(a)) { printf("n = %5ld ns = %5ld rank5: not supported\n", n, ns); continue; }
#else
                if (kern == 1) continue;  // the one-workgroup-per-CU experiment: sort5_probe_persist.bin
#endif
                float best = 1e30f, ms = 0.f;
                for (int it = 0; it < reps + 1; it++) {
                    hipMemset(flags, 0, ncols * 4);
                    hipMemset(out, 0xff, h.size() * 4);
                    hipEventRecord(e0, 0);
                    if (kern == 0) optex::launch_rank4(optex::SORT_MATCH, a, ncols, 0);
#ifdef R5_PERSISTENT_VARIANT
                    else if (kern == 1) optex::launch_rank5(a, ncols, 0);
#endif
                    else optex::launch_rank5w(optex::SORT_MATCH, a, ncols, 0);
                    hipEventRecord(e1, 0);
                    hipError_t err = hipDeviceSynchronize();
                    if (err != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(err)); return 1; }
                    hipEventElapsedTime(&ms, e0, e1);
                    if (it > 0 && ms < best) best = ms;
                }
                std::vector<int> fl(ncols);
                hipMemcpy(fl.data(), flags, ncols * 4, hipMemcpyDeviceToHost);
                const int nflag = std::accumulate(fl.begin(), fl.end(), 0);
                int nchecked = 0;
                const int bad = check_columns(h, hs, out, fl, n, ns, C, ncols, ncheck, &nchecked);
                const double bytes = 12.0 * n * ncols;
                printf("n = %5ld ns = %5ld %-6s %8.1f us  %6.2f TB/s  %.3f of 8 TB/s   flagged %d, mismatches on %d checked columns %d\n", n, ns,
                       kern == 2 ? "rank5w" : (kern ? "rank5" : "rank4"), best * 1e3, bytes / (best * 1e9), bytes / (best * 1e9) / 8.0, nflag, nchecked, bad);
                fflush(stdout);
                tot_us[kern] += weight[si] * best * 1e3;
                if (kern == 0) tot_bytes += weight[si] * bytes;
            }
            hipFree(x); hipFree(out); hipFree(ss); hipFree(flags); hipFree(dlo); hipFree(dhi);
        }
        for (int kern = 0; kern < 3; kern++)
            if (tot_us[kern] > 0)
            printf("schedule-weighted (13/12/10/9/8 iterations), ns = %s, %-6s: %.2f ms per step, %.2f TB/s = %.3f of HBM peak\n",
                   ratio ? "23 n / 16" : "3 n / 4", kern == 2 ? "rank5w" : (kern ? "rank5" : "rank4"), tot_us[kern] * 1e-3, tot_bytes / (tot_us[kern] * 1e6),
                   tot_bytes / (tot_us[kern] * 1e6) / 8.0);
    }
    // ---- the sort itself (optex_sort_columns: keys + pixel indices by rank, SURVEY 8d's 12 B per element), own range: both kernels
    for (int si = 0; si < 5 && (!only_n || sizes[si] == only_n); si++) {
        const long n = sizes[si];
        std::vector<float> h((size_t)ncols * n);
        std::mt19937 g(11 + si);
        std::normal_distribution<float> d(0.f, 1.f);
        for (auto& v : h) v = d(g);
        float *x, *ok;
        uint32_t* oi;
        int* flags;
        hipMalloc(&x, h.size() * 4); hipMalloc(&ok, h.size() * 4); hipMalloc(&oi, h.size() * 4); hipMalloc(&flags, ncols * 4);
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        optex::SortArgs a{};
        a.keys = x; a.ld = n; a.ss = (long)C * n; a.n = n; a.C = C; a.x_n_seg = S;
        a.out_keys = ok; a.out_idx = oi; a.flags = flags; a.ncols = ncols;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        std::vector<float> hlo(ncols), hhi(ncols);
        for (int c2 = 0; c2 < ncols; c2++) {
            const float* kc = h.data() + (size_t)c2 * n;
            hlo[c2] = *std::min_element(kc, kc + n);
            hhi[c2] = *std::max_element(kc, kc + n);
        }
        float *dlo2, *dhi2;
        hipMalloc(&dlo2, ncols * 4); hipMalloc(&dhi2, ncols * 4);
        hipMemcpy(dlo2, hlo.data(), ncols * 4, hipMemcpyHostToDevice);
        hipMemcpy(dhi2, hhi.data(), ncols * 4, hipMemcpyHostToDevice);
        for (int kern = 0; kern < 5; kern++) {   // 0 rank4; rank5w: 1 own range, 2 given range, 3 keys only, 4 indices only
            a.rng_lo = kern == 2 ? dlo2 : nullptr; a.rng_hi = kern == 2 ? dhi2 : nullptr;
            a.out_keys = kern == 4 ? nullptr : ok; a.out_idx = kern == 3 ? nullptr : oi;
            if (kern >= 1 && !optex::rank5w_supported(optex::SORT_EMIT, a)) continue;
            hipMemset(ok, 0xff, h.size() * 4); hipMemset(oi, 0xff, h.size() * 4);
            float best = 1e30f, ms = 0.f;
            for (int it = 0; it < reps + 1; it++) {
                hipMemset(flags, 0, ncols * 4);
                hipEventRecord(e0, 0);
                if (kern == 0) optex::launch_rank4(optex::SORT_EMIT, a, ncols, 0);
                else optex::launch_rank5w(optex::SORT_EMIT, a, ncols, 0);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
                if (it > 0 && ms < best) best = ms;
            }
            std::vector<int> fl(ncols);
            hipMemcpy(fl.data(), flags, ncols * 4, hipMemcpyDeviceToHost);
            int bad = 0, nchecked = 0;
            std::vector<float> gk(n);
            std::vector<uint32_t> gi(n);
            std::vector<int> idx(n);
            for (int k = 0; k < ncheck; k++) {
                const int col = (int)(((long)k * 2731 + 17) % ncols);
                if (fl[col]) continue;
                nchecked++;
                hipMemcpy(gk.data(), ok + (size_t)col * n, n * 4, hipMemcpyDeviceToHost);
                hipMemcpy(gi.data(), oi + (size_t)col * n, n * 4, hipMemcpyDeviceToHost);
                const float* kc = h.data() + (size_t)col * n;
                std::iota(idx.begin(), idx.end(), 0);
                std::stable_sort(idx.begin(), idx.end(), [&](int p, int q) { return kc[p] < kc[q]; });
                for (long i = 0; i < n; i++)
                    if ((a.out_idx && gi[i] != (uint32_t)idx[i]) || (a.out_keys && memcmp(&gk[i], &kc[idx[i]], 4) != 0)) {
                        if (bad < 6) printf("    col %d rank %ld: got (pixel %u key %.9g) expected (pixel %d key %.9g)\n", col, i, gi[i], gk[i], idx[i], kc[idx[i]]);
                        bad++;
                    }
            }
            const double bytes = 12.0 * n * ncols;
            printf("sort_columns (keys + indices) n = %5ld %-6s %8.1f us  %6.2f TB/s  %.3f of 8 TB/s   flagged %d, mismatches on %d checked columns %d%s\n", n,
                   kern == 0 ? "rank4" : (kern == 1 ? "rank5w" : (kern == 2 ? "5w-rng" : (kern == 3 ? "5w-key" : "5w-idx"))), best * 1e3, bytes / (best * 1e9), bytes / (best * 1e9) / 8.0,
                   std::accumulate(fl.begin(), fl.end(), 0), nchecked, bad, bad ? "   <-- WRONG" : "");
            fflush(stdout);
        }
        hipFree(x); hipFree(ok); hipFree(oi); hipFree(flags); hipFree(dlo2); hipFree(dhi2);
    }
    // ---- adversarial columns, rank5 only, every column checked: [distribution][n]
    if (!only_n) {
        const int Cc = 16, Ss = 4, nc = Cc * Ss;
        const long ns_list[3] = {0, 1, 2};
        for (long n : {16384L, 12544L, 9216L, 6400L, 4096L, 15000L, 8196L, 5124L, 2052L, 3000L}) {
            for (int dist = 0; dist < 8; dist++) {
                for (long nsk : ns_list) {
                    const long ns = nsk == 0 ? n : (nsk == 1 ? (n * 3 / 4 + 3) / 4 * 4 : n * 23 / 16 / 4 * 4);
                    std::vector<float> h((size_t)nc * n), hs((size_t)Cc * ns), lo(nc), hi(nc);
                    std::mt19937 g(77 + dist);
                    std::normal_distribution<float> d(0.f, 1.f);
                    std::uniform_real_distribution<float> u(0.f, 1.f);
                    for (size_t i = 0; i < h.size(); i++) {
                        float v = d(g);
                        switch (dist) {
                            case 0: break;                                               // gaussian
                            case 1: v = std::floor(v * 64.f) / 64.f; break;              // ~500 distinct values: heavy ties
                            case 2: if (u(g) < 0.02f) v = 0.5f; break;                   // one big tie group + gaussian
                            case 3: v = (i % n == 7) ? 1.0e6f : v; break;                // one outlier sets the range
                            case 4: v = std::exp(3.f * v); break;                        // log-normal, heavy tail
                            case 5: v = v > 0.f ? v : 0.f; break;                        // ReLU: half zeros
                            case 6: v = (u(g) < 0.5f) ? -0.f : 0.f; if (i % 97 == 0) v = d(g); break;  // signed zeros + a few values
                            case 7: v = (float)(int)(u(g) * 3.f); break;                 // three values
                        }
                        h[i] = v;
                    }
                    for (auto& v : hs) v = d(g);
                    for (int c = 0; c < Cc; c++) std::sort(hs.begin() + (size_t)c * ns, hs.begin() + (size_t)(c + 1) * ns);
                    for (int c = 0; c < nc; c++) {
                        auto mm = std::minmax_element(h.begin() + (size_t)c * n, h.begin() + (size_t)(c + 1) * n);
                        lo[c] = *mm.first;
                        hi[c] = *mm.second;
                    }
                    float *x, *ss, *out, *dlo, *dhi;
                    int* flags;
                    hipMalloc(&x, h.size() * 4); hipMalloc(&out, h.size() * 4); hipMalloc(&ss, hs.size() * 4);
                    hipMalloc(&flags, nc * 4); hipMalloc(&dlo, nc * 4); hipMalloc(&dhi, nc * 4);
                    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
                    hipMemcpy(ss, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
                    hipMemcpy(dlo, lo.data(), nc * 4, hipMemcpyHostToDevice);
                    hipMemcpy(dhi, hi.data(), nc * 4, hipMemcpyHostToDevice);
                    optex::SortArgs a{};
                    a.keys = x; a.ld = n; a.ss = (long)Cc * n; a.n = n; a.C = Cc; a.x_n_seg = Ss;
                    a.src_sorted = ss; a.ns = ns; a.src_n_seg = 1;
                    a.out = out; a.ldo = n; a.oss = (long)Cc * n; a.out_vec = 1;
                    a.flags = flags; a.inv_2nt = 1.0 / (2.0 * n); a.ncols = nc;
                    a.rng_lo = dlo; a.rng_hi = dhi;
#ifdef R5_PERSISTENT_VARIANT
                    for (int kern = 1; kern < 3; kern++)
                    if (kern == 1 ? !optex::rank5_supported(a) : !optex::rank5w_supported(optex::SORT_MATCH, a)) {
#else
                    for (int kern = 2; kern < 3; kern++)
                    if (!optex::rank5w_supported(optex::SORT_MATCH, a)) {
#endif
                        printf("adversarial n = %5ld ns = %5ld dist %d %s: not supported\n", n, ns, dist, kern == 1 ? "rank5" : "rank5w");
                    } else {
                        hipMemset(flags, 0, nc * 4);
                        hipMemset(out, 0xff, h.size() * 4);
#ifdef R5_PERSISTENT_VARIANT
                        if (kern == 1) optex::launch_rank5(a, nc, 0);
                        else
#endif
                        optex::launch_rank5w(optex::SORT_MATCH, a, nc, 0);
                        hipError_t err = hipDeviceSynchronize();
                        if (err != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(err)); return 1; }
                        std::vector<int> fl(nc);
                        hipMemcpy(fl.data(), flags, nc * 4, hipMemcpyDeviceToHost);
                        const int nflag = std::accumulate(fl.begin(), fl.end(), 0);
                        int nchecked = 0;
                        // all columns: check_columns walks (k * 2731 + 17) % nc, a permutation for nc = 64
                        const int bad = check_columns(h, hs, out, fl, n, ns, Cc, nc, nc, &nchecked);
                        printf("adversarial n = %5ld ns = %5ld dist %d %-6s: flagged %2d of %d, mismatches on %2d checked columns %d%s\n", n, ns, dist,
                               kern == 1 ? "rank5" : "rank5w", nflag, nc, nchecked, bad, bad ? "   <-- WRONG" : "");
                    }
                    fflush(stdout);
                    hipFree(x); hipFree(out); hipFree(ss); hipFree(flags); hipFree(dlo); hipFree(dhi);
                }
            }
        }
    }
    return 0;
}
