// Diagnostic: semantics of v_mov_b32_dpp wave_shr:1 and v_addc with an SGPR carry-in on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned lane = threadIdx.x;
    unsigned v = 100 + lane;
    unsigned s1 = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x138, 0xf, 0xf, false);
    unsigned s2 = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)s1, 0x138, 0xf, 0xf, false);
    unsigned long long m = __builtin_amdgcn_ballot_w64((lane % 3) == 0);
    unsigned acc = 5;
    unsigned long long co;
    asm volatile("v_addc_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(acc), "=s"(co) : "s"(m));
    out[lane] = s1; out[64 + lane] = s2; out[128 + lane] = acc;
}
int main() {
    unsigned* d; hipMalloc(&d, 192 * 4);
    k<<<1, 64>>>(d);
    unsigned h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; r++) { for (int i = 0; i < 64; i++) printf("%u ", h[r * 64 + i]); printf("\n"); }
    return 0;
}
