// Diagnostic (not part of the library): what a "whole column per workgroup" copy costs on MI355X by its shape — threads per
// column, 16-byte loads in flight per thread, bursts against interleaved loads and stores — beside the plain streaming copy.
// [64 x 256] columns of n floats, in place or out of place.   scripts/colcopy_probe.bin [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NV, int NT>
__global__ __launch_bounds__(NT) void copy_burst(const float* __restrict__ x, float* __restrict__ y, long n) {
    const float4* p = reinterpret_cast<const float4*>(x + (size_t)blockIdx.x * n);
    float4* o = reinterpret_cast<float4*>(y + (size_t)blockIdx.x * n);
    const int nv = (int)(n / 4), tid = threadIdx.x;
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (tid + NT * k < nv) v[k] = p[tid + NT * k];
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (tid + NT * k < nv) o[tid + NT * k] = v[k];
}

// the same bytes per workgroup, but G loads then G stores at a time
template <int NV, int NT, int G>
__global__ __launch_bounds__(NT) void copy_groups(const float* __restrict__ x, float* __restrict__ y, long n) {
    const float4* p = reinterpret_cast<const float4*>(x + (size_t)blockIdx.x * n);
    float4* o = reinterpret_cast<float4*>(y + (size_t)blockIdx.x * n);
    const int nv = (int)(n / 4), tid = threadIdx.x;
#pragma unroll
    for (int g = 0; g < NV; g += G) {
        float4 v[G];
#pragma unroll
        for (int k = 0; k < G; k++)
            if (tid + NT * (g + k) < nv) v[k] = p[tid + NT * (g + k)];
#pragma unroll
        for (int k = 0; k < G; k++)
            if (tid + NT * (g + k) < nv) o[tid + NT * (g + k)] = v[k];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// a barrier between the load burst and the store burst (what a compute phase on the whole column implies)
template <int NV, int NT>
__global__ __launch_bounds__(NT) void copy_burst_barrier(const float* __restrict__ x, float* __restrict__ y, long n) {
    const float4* p = reinterpret_cast<const float4*>(x + (size_t)blockIdx.x * n);
    float4* o = reinterpret_cast<float4*>(y + (size_t)blockIdx.x * n);
    const int nv = (int)(n / 4), tid = threadIdx.x;
    __shared__ float s[NT];
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (tid + NT * k < nv) v[k] = p[tid + NT * k];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++) acc += v[k].x;
    s[tid] = acc;
    __syncthreads();
    const float t = s[(tid + 1) & (NT - 1)] * 1e-30f;
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (tid + NT * k < nv) {
            v[k].x += t;
            o[tid + NT * k] = v[k];
        }
}

__global__ __launch_bounds__(256) void copy_stream(const float4* __restrict__ x, float4* __restrict__ y, size_t nv) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) y[i] = x[i];
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 16384;
    const int ncols = 64 * 256;
    float *x, *y;
    hipMalloc(&x, (size_t)ncols * n * 4);
    hipMalloc(&y, (size_t)ncols * n * 4);
    hipMemset(x, 0, (size_t)ncols * n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = 8.0 * ncols * n;
    auto timed = [&](const char* what, auto fn) {
        float best = 1e30f, ms;
        for (int it = 0; it < 7; it++) {
            hipEventRecord(e0, 0);
            fn();
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
            if (it > 1 && ms < best) best = ms;
        }
        printf("n = %5ld  %-64s %8.1f us  %5.2f TB/s\n", n, what, best * 1e3, bytes / (best * 1e9));
    };
    timed("streaming copy, grid-stride, 2048 blocks", [&] { copy_stream<<<2048, 256>>>((const float4*)x, (float4*)y, (size_t)ncols * n / 4); });
    timed("streaming copy IN PLACE", [&] { copy_stream<<<2048, 256>>>((const float4*)x, (float4*)x, (size_t)ncols * n / 4); });
    if (n <= 16384) {
        timed("column per workgroup: 256 threads x 16 loads, burst", [&] { copy_burst<16, 256><<<ncols, 256>>>(x, y, n); });
        timed("column per workgroup: 512 threads x 8 loads, burst", [&] { copy_burst<8, 512><<<ncols, 512>>>(x, y, n); });
        timed("column per workgroup: 1024 threads x 4 loads, burst", [&] { copy_burst<4, 1024><<<ncols, 1024>>>(x, y, n); });
        timed("column per workgroup: 1024 x 4, IN PLACE", [&] { copy_burst<4, 1024><<<ncols, 1024>>>(x, x, n); });
        timed("column per workgroup: 256 x 16, IN PLACE", [&] { copy_burst<16, 256><<<ncols, 256>>>(x, x, n); });
        timed("column per workgroup: 256 x 16, groups of 4 loads + 4 stores", [&] { copy_groups<16, 256, 4><<<ncols, 256>>>(x, y, n); });
        timed("column per workgroup: 256 x 16, groups of 1", [&] { copy_groups<16, 256, 1><<<ncols, 256>>>(x, y, n); });
        timed("column per workgroup: 256 x 16, burst + barrier", [&] { copy_burst_barrier<16, 256><<<ncols, 256>>>(x, y, n); });
        timed("column per workgroup: 512 x 8, burst + barrier", [&] { copy_burst_barrier<8, 512><<<ncols, 512>>>(x, y, n); });
        timed("column per workgroup: 1024 x 4, burst + barrier", [&] { copy_burst_barrier<4, 1024><<<ncols, 1024>>>(x, y, n); });
        timed("column per workgroup: 1024 x 4, burst + barrier, IN PLACE", [&] { copy_burst_barrier<4, 1024><<<ncols, 1024>>>(x, x, n); });
    }
    return 0;
}
