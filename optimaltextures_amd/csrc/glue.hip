// glue.hip — the element-wise glue BETWEEN the VGG convolutions (reference vgg.py:14-135): bias add, ReLU, 2x2
// ceil-mode max-pool, nearest 2x upsampling and the 1-pixel reflection pad that precedes every 3x3 convolution, fused
// into ONE pass per convolution boundary.  The convolutions themselves stay on PyTorch-ROCm / MIOpen (north star);
// what PyTorch runs as 3-4 separate kernels per boundary (bias add, clamp, pool / upsample, reflection_pad2d — 26 % of a
// bench step in the round-1 profile) becomes one read and one write.
//
//   v   = x[n][c][y][x] (+ bias[c])          nn.Conv2d bias, applied here so the conv can run bias-free
//   v   = max(v, 0)                           nn.ReLU                      (if relu)
//   v   = max over the 2x2 window             nn.MaxPool2d(2, 2, ceil_mode=True)   (if pool)
//   v   = nearest-neighbour 2x                nn.UpsamplingNearest2d(2)    (if up)
//   out = reflect-pad(v, pad)                 nn.ReflectionPad2d(1)        (pad = 0 or 1)
//
// Pure data movement with one add and one max per element: results are bit-identical to the PyTorch op sequence.
// HBM-bound: algorithmic bytes = 4 * (input elements + output elements).
#include "optex_common.h"

namespace optex {

struct GlueArgs {
    const float* x; const float* bias; float* out;
    int C, H, W;          // input plane
    int Hm, Wm;           // after pool / upsample
    int Ho, Wo;           // after padding
    int relu, pool, up, pad, vec2;
};

__device__ __forceinline__ int reflect_index(int i, int n) {  // reflection without repeating the border, |i| < n
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

constexpr int GLUE_ROWS = 8;  // output rows per 256-thread block: two per wavefront

// One wavefront per output row: no integer division per element, source rows are read as contiguous spans (shifted
// by the padding), the reflected border costs two lanes per row.
template <bool POOL>
__global__ __launch_bounds__(256) void glue_kernel(GlueArgs a) {
    const int plane = blockIdx.y;             // n * C + c
    const int c = plane % a.C;
    const float b = a.bias ? a.bias[c] : 0.f;
    const float* __restrict__ xin = a.x + (size_t)plane * a.H * a.W;
    float* __restrict__ o = a.out + (size_t)plane * a.Ho * a.Wo;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int rr = 0; rr < GLUE_ROWS / 4; rr++) {
        const int oy = blockIdx.x * GLUE_ROWS + rr * 4 + w;
        if (oy >= a.Ho) break;
        int my = reflect_index(oy - a.pad, a.Hm);
        if (a.up) my >>= 1;
        float* __restrict__ orow = o + (size_t)oy * a.Wo;
        const int sh = a.up ? 1 : 0;
        const int y0 = POOL ? 2 * my : my;
        const float* __restrict__ r0 = xin + (size_t)y0 * a.W;
        const float* __restrict__ r1 = r0 + ((POOL && y0 + 1 < a.H) ? a.W : 0);   // ceil_mode: partial windows at odd edges
        auto value = [&](int ox) {
            float v;
            if (POOL) {
                const int x0 = 2 * reflect_index(ox - a.pad, a.Wm);
                const int x1 = (x0 + 1 < a.W) ? x0 + 1 : x0;
                v = fmaxf(fmaxf(r0[x0], r0[x1]), fmaxf(r1[x0], r1[x1])) + b;  // max(x_i) + b == max(x_i + b)
            } else {
                v = r0[reflect_index(ox - a.pad, a.Wm) >> sh] + b;
            }
            return a.relu ? fmaxf(v, 0.f) : v;
        };
        if (a.vec2) {  // even row length and 8-byte aligned planes: 8-byte stores (the glue is store-bound)
            for (int ox = 2 * lane; ox < a.Wo; ox += 128)
                *reinterpret_cast<float2*>(orow + ox) = make_float2(value(ox), value(ox + 1));
        } else {
            for (int ox = lane; ox < a.Wo; ox += 64) orow[ox] = value(ox);
        }
    }
}

}  // namespace optex

using namespace optex;

extern "C" int optex_vgg_glue(const float* x, const float* bias, float* out, int N, int C, int H, int W, int relu,
                              int pool, int up, int pad, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (pool && up) || pad < 0 || pad > 1) {
        set_error("optex_vgg_glue: bad argument (N=%d C=%d H=%d W=%d pool=%d up=%d pad=%d)", N, C, H, W, pool, up, pad);
        return OPTEX_E_ARG;
    }
    GlueArgs a;
    a.x = x; a.bias = bias; a.out = out;
    a.C = C; a.H = H; a.W = W;
    a.Hm = pool ? (H + 1) / 2 : (up ? 2 * H : H);
    a.Wm = pool ? (W + 1) / 2 : (up ? 2 * W : W);
    a.Ho = a.Hm + 2 * pad;
    a.Wo = a.Wm + 2 * pad;
    a.relu = relu; a.pool = pool; a.up = up; a.pad = pad;
    a.vec2 = (a.Wo % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % 8 == 0);
    if (pad && (a.Hm < 2 || a.Wm < 2)) {
        set_error("optex_vgg_glue: reflection padding needs at least 2 pixels per side (got %d x %d)", a.Hm, a.Wm);
        return OPTEX_E_ARG;
    }
    const long long planes = (long long)N * C;
    const long long per_plane = (long long)a.Ho * a.Wo;
    if (per_plane > 0x7fffffffLL) {
        set_error("optex_vgg_glue: tensor too large for one launch");
        return OPTEX_E_ARG;
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(KC_GLUE, st, 0.0, 4.0 * ((double)planes * H * W + (double)planes * per_plane));
    // blockIdx.y is limited to 65535: split the planes over launches of whole images' worth of channels
    const long long step = (65535 / C) * (long long)C;
    if (step == 0) {
        set_error("optex_vgg_glue: C = %d exceeds the grid limit", C);
        return OPTEX_E_UNSUPPORTED;
    }
    for (long long p0 = 0; p0 < planes; p0 += step) {
        const int np = (int)((planes - p0 < step) ? planes - p0 : step);
        GlueArgs b = a;
        b.x = x + (size_t)p0 * H * W;
        b.out = out + (size_t)p0 * per_plane;
        dim3 grid((unsigned)((a.Ho + GLUE_ROWS - 1) / GLUE_ROWS), (unsigned)np);
        if (pool) hipLaunchKernelGGL(glue_kernel<true>, grid, dim3(256), 0, st, b);
        else hipLaunchKernelGGL(glue_kernel<false>, grid, dim3(256), 0, st, b);
    }
    return check_launch("glue_kernel");
}
