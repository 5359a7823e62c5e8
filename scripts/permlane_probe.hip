// Diagnostic: lane semantics of v_permlane32_swap_b32 on gfx950 (used by chol_inv_kernel to hand an MFMA result's two lane
// halves to the threads that own the rows).   hipcc --offload-arch=gfx950 scripts/permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned x = 1000u + threadIdx.x, y = 2000u + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r[0]: lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n", h[0], h[31], h[32], h[63]);
    printf("r[1]: lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n", h[64], h[95], h[96], h[127]);
    return 0;
}
