// Diagnostic (not part of the library): the two Cholesky + inverse kernels of csrc/linalg.hip against each other — results
// (they compute the same fma chains: expected bit-equal) and time.   make -C scripts chol_probe.bin && scripts/chol_probe.bin
#include "../optimaltextures_amd/csrc/linalg.hip"
#include "../optimaltextures_amd/csrc/gemm.hip"
#include "../optimaltextures_amd/csrc/gemm_rs.hip"

#include <cmath>
#include <random>
#include <vector>

int main() {
    optex::chol_mfma_min_panels = 1;   // time the MFMA kernel at every size (the library switches at launch_chol_inv's threshold)
    std::mt19937 g(3);
    std::normal_distribution<float> d(0.f, 1.f);
    for (int C : {23, 32, 48, 64, 96, 100, 128, 181, 256, 300, 512}) {
        for (int batch : {1, 64}) {
            const int NP = optex::chol_np(C);
            std::vector<float> h((size_t)batch * C * C);
            for (int b = 0; b < batch; b++) {  // A = X^T X / n + I
                std::vector<float> x((size_t)2 * C * C);
                for (auto& v : x) v = d(g);
                for (int i = 0; i < C; i++)
                    for (int j = 0; j <= i; j++) {
                        double s = 0;
                        for (int k = 0; k < 2 * C; k++) s += (double)x[(size_t)k * C + i] * x[(size_t)k * C + j];
                        const float v = (float)(s / (2 * C)) + (i == j ? 1.f : 0.f);
                        h[((size_t)b * C + i) * C + j] = v;
                        h[((size_t)b * C + j) * C + i] = v;
                    }
            }
            float *A, *U[2], *L[2];
            const size_t pp = (size_t)batch * NP * NP;
            (void)hipMalloc(&A, h.size() * 4);
            for (int v = 0; v < 2; v++) { (void)hipMalloc(&U[v], pp * 4); (void)hipMalloc(&L[v], pp * 4); }
            (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            float us[2];
            for (int v = 0; v < 2; v++) {
                optex::chol_use_mfma = v == 1;
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                for (int i = 0; i < 2; i++) optex::launch_chol_inv(A, (long)C * C, C, batch, U[v], L[v], 0);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, 0);
                for (int i = 0; i < 10; i++) optex::launch_chol_inv(A, (long)C * C, C, batch, U[v], L[v], 0);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                us[v] = ms * 100.f;
            }
            std::vector<float> u0(pp), u1(pp), l0(pp), l1(pp);
            (void)hipMemcpy(u0.data(), U[0], pp * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(u1.data(), U[1], pp * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(l0.data(), L[0], pp * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(l1.data(), L[1], pp * 4, hipMemcpyDeviceToHost);
            double du = 0, dl = 0; size_t nbits = 0;
            for (size_t i = 0; i < pp; i++) {
                du = std::fmax(du, std::fabs((double)u0[i] - u1[i]));
                dl = std::fmax(dl, std::fabs((double)l0[i] - l1[i]));
                nbits += (u0[i] != u1[i]) + (l0[i] != l1[i]);
            }
            printf("C = %3d batch %2d: VALU kernel %7.1f us, MFMA kernel %7.1f us  (%.2fx)   max |dU| %.2e  max |dLinv| %.2e  differing words %zu\n",
                   C, batch, us[0], us[1], us[0] / us[1], du, dl, nbits);
            (void)hipFree(A);
            for (int v = 0; v < 2; v++) { (void)hipFree(U[v]); (void)hipFree(L[v]); }
        }
    }
    return 0;
}
