// Diagnostic (not part of the library): dumps the slot array, the start bitmap and the per-window results of
// rank_match_kernel for one column and checks the invariants on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DR2_DEBUG scripts/sort_rank2_debug.hip \
//         optimaltextures_amd/csrc/api.hip -o /tmp/r2dbg && /tmp/r2dbg [n]
#define R2_DEBUG
#include "../optimaltextures_amd/csrc/sort_rank2.hip"

#include <algorithm>
#include <random>
#include <vector>

template <int ITEMS>
int run(int n) {
    using K = optex::R2<ITEMS>;
    std::vector<float> h(n), srcs(n);
    std::mt19937 g(n);
    std::normal_distribution<float> d(0.f, 1.f);
    for (auto& v : h) v = d(g);
    for (int i = 0; i < n; i++) srcs[i] = (float)i;
    float *x, *ss, *out; uint32_t* dbg; int* flags;
    const size_t ndbg = 2 * K::CAP + (size_t)K::TRIPS * 16 * 64 + 64;
    hipMalloc(&x, n * 4); hipMalloc(&ss, n * 4); hipMalloc(&out, n * 4); hipMalloc(&dbg, ndbg * 4); hipMalloc(&flags, 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(ss, srcs.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(flags, 0, 4); hipMemset(dbg, 0xff, ndbg * 4);
    optex::SortArgs a{};
    a.keys = x; a.ld = n; a.ss = n; a.n = n; a.C = 1; a.x_n_seg = 1;
    a.src_sorted = ss; a.ns = n; a.src_n_seg = 1; a.out = out; a.ldo = n; a.oss = n; a.flags = flags;
    a.inv_2nt = 1.0 / (2.0 * n); a.ncols = 1; a.out_vec = 1; a.dbg = dbg;
    auto kern = optex::rank_match_kernel<ITEMS, (ITEMS >= 4)>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS);
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), K::LDS, 0, a);
    hipDeviceSynchronize();
    std::vector<uint32_t> D(ndbg); std::vector<float> o(n); int fl;
    hipMemcpy(D.data(), dbg, ndbg * 4, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(&fl, flags, 4, hipMemcpyDeviceToHost);
    printf("n=%d ITEMS=%d flag=%d\n", n, ITEMS, fl);
    const uint32_t* slot = D.data(); const uint32_t* bs = D.data() + K::CAP; const uint32_t* res = D.data() + 2 * K::CAP;
    auto bit = [&](int p) { return (bs[p >> 5] >> (p & 31)) & 1u; };
    // 1. every pixel appears once
    std::vector<int> seen(n, 0);
    for (int p = 0; p < n; p++) seen[slot[p] & 0x3fff]++;
    int bad = 0; for (int i = 0; i < n; i++) bad += seen[i] != 1;
    printf(" pixels not placed exactly once: %d; bit0=%u sentinel=%u\n", bad, bit(0), bit(n));
    // 2. runs: keys of a run all <= keys of the next run; inside a run the packed order agrees with the key order
    int runs = 0, cross = 0, inner = 0, maxlen = 0;
    float prevmax = -1e30f;
    for (int s = 0; s < n;) {
        int e = s + 1; while (e < n && !bit(e)) e++;
        runs++; maxlen = std::max(maxlen, e - s);
        float mn = 1e30f, mx = -1e30f;
        for (int j = s; j < e; j++) { float k = h[slot[j] & 0x3fff]; mn = std::min(mn, k); mx = std::max(mx, k); }
        if (mn < prevmax) { if (cross < 5) printf("  run [%d,%d) min %.9g < previous max %.9g\n", s, e, mn, prevmax); cross++; }
        prevmax = std::max(prevmax, mx);
        for (int i = s; i < e; i++) for (int j = s; j < e; j++) {
            float ki = h[slot[i] & 0x3fff], kj = h[slot[j] & 0x3fff];
            if (ki < kj && (slot[i] >> 14) > (slot[j] >> 14)) inner++;
        }
        s = e;
    }
    printf(" runs %d (longest %d), runs out of key order: %d, packed-vs-key inversions inside runs: %d\n", runs, maxlen, cross, inner);
    // 3. results per window vs the true rank
    std::vector<int> order(n); for (int i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return h[p] < h[q]; });
    std::vector<int> trank(n); for (int i = 0; i < n; i++) trank[order[i]] = i;
    int wrong = 0, none = 0; std::vector<int> hits(n, 0);
    const int nwin = (n + 47) / 48;
    for (int t = 0; t < nwin; t++) for (int l = 0; l < 64; l++) {
        uint32_t r = res[t * 64 + l];
        if (r == 0xffffffffu) { none++; continue; }
        int idx = r >> 14, rk = r & 0x3fff; hits[idx]++;
        if (rk != trank[idx]) { if (wrong < 12) printf("  window %d lane %d: pixel %d rank %d, true %d (slot word %08x)\n", t, l, idx, rk, trank[idx], slot[t * 48 + l]); wrong++; }
    }
    // 4. details of the first wrong lanes: the run and the words in it
    int shown = 0;
    for (int t = 0; t < nwin && shown < 4; t++) for (int l = 0; l < 64 && shown < 4; l++) {
        uint32_t r = res[t * 64 + l];
        if (r == 0xffffffffu) continue;
        int idx = r >> 14, rk = r & 0x3fff;
        if (rk == trank[idx]) continue;
        int p = t * 48 + l, S = p, E = p + 1;
        while (!bit(S)) S--;
        while (!bit(E)) E++;
        printf("  window %d lane %d (slot %d) run [%d,%d) = lanes [%d,%d): got %d true %d\n", t, l, p, S, E, S - t * 48, E - t * 48, rk, trank[idx]);
        for (int j = S; j < E; j++) printf("     slot %d lane %d word %08x key %.9g true rank %d res %08x\n", j, j - t * 48, slot[j], h[slot[j] & 0x3fff], trank[slot[j] & 0x3fff], (j - t * 48 < 64) ? res[t * 64 + j - t * 48] : 0u);
        shown++;
    }
    int nohit = 0, multi = 0; for (int i = 0; i < n; i++) { nohit += hits[i] == 0; multi += hits[i] > 1; }
    printf(" res: wrong %d, pixels without a result %d, with several %d\n", wrong, nohit, multi);
    int ow = 0; for (int i = 0; i < n; i++) ow += (int)o[i] != trank[i];
    printf(" final output wrong: %d\n", ow);
    return 0;
}

int main(int argc, char** argv) {
    run<2>(1024); run<4>(4096); run<8>(5000); run<16>(16384);
    return 0;
}
