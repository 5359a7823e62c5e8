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
//
// Layouts: MIOpen's fp32 3x3 convolutions with >= 64 channels on both sides run 8-20 % faster on channels-last tensors
// (MFMA implicit-GEMM kernels instead of the gfx9 Winograd assembly; scripts/conv_layout_probe.py), while the sliced-OT
// kernels and the 3-channel ends of the codec want planar NCHW.  The glue is where a layout change is free: it reads
// one layout and writes the other in the same pass (glue_layout_kernel: channels-last on both sides = 16-byte
// accesses along C; mixed = 32 x 32 (pixel, channel) tiles transposed through LDS so that both the read and the write
// are contiguous 128-byte spans).
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


// ---- layout-aware variants: input and output independently NCHW (planar) or NHWC (channels-last)
struct GlueLArgs {
    const float* x; const float* bias; float* out;
    int N, C, H, W, Hm, Wm, Ho, Wo;
    int relu, pool, up, pad;
};

template <bool IN_NHWC>
__device__ __forceinline__ size_t glue_in_index(const GlueLArgs& a, int n, int c, int y, int x) {
    return IN_NHWC ? (((size_t)n * a.H + y) * a.W + x) * a.C + c : (((size_t)n * a.C + c) * a.H + y) * a.W + x;
}

// value of output pixel (oy, ox) of channel c of image n, before the store
template <bool IN_NHWC, bool POOL>
__device__ __forceinline__ float glue_value(const GlueLArgs& a, int n, int c, int oy, int ox, float b) {
    int my = reflect_index(oy - a.pad, a.Hm), mx = reflect_index(ox - a.pad, a.Wm);
    float v;
    if (POOL) {
        const int y0 = 2 * my, x0 = 2 * mx;
        const int y1 = (y0 + 1 < a.H) ? y0 + 1 : y0, x1 = (x0 + 1 < a.W) ? x0 + 1 : x0;  // ceil_mode partial windows
        v = fmaxf(fmaxf(a.x[glue_in_index<IN_NHWC>(a, n, c, y0, x0)], a.x[glue_in_index<IN_NHWC>(a, n, c, y0, x1)]),
                  fmaxf(a.x[glue_in_index<IN_NHWC>(a, n, c, y1, x0)], a.x[glue_in_index<IN_NHWC>(a, n, c, y1, x1)])) + b;
    } else {
        if (a.up) { my >>= 1; mx >>= 1; }
        v = a.x[glue_in_index<IN_NHWC>(a, n, c, my, mx)] + b;
    }
    return a.relu ? fmaxf(v, 0.f) : v;
}

// 4 consecutive channels of output pixel (oy, ox) from a channels-last input (c % 4 == 0, 16-byte aligned)
template <bool POOL>
__device__ __forceinline__ float4 glue_value4_nhwc(const GlueLArgs& a, int n, int c, int oy, int ox) {
    const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    int my = reflect_index(oy - a.pad, a.Hm), mx = reflect_index(ox - a.pad, a.Wm);
    auto ld = [&](int y, int x) { return *reinterpret_cast<const float4*>(a.x + glue_in_index<true>(a, n, c, y, x)); };
    float4 v;
    if (POOL) {
        const int y0 = 2 * my, x0 = 2 * mx;
        const int y1 = (y0 + 1 < a.H) ? y0 + 1 : y0, x1 = (x0 + 1 < a.W) ? x0 + 1 : x0;
        const float4 p = ld(y0, x0), q = ld(y0, x1), r = ld(y1, x0), t = ld(y1, x1);
        v.x = fmaxf(fmaxf(p.x, q.x), fmaxf(r.x, t.x)) + b.x;
        v.y = fmaxf(fmaxf(p.y, q.y), fmaxf(r.y, t.y)) + b.y;
        v.z = fmaxf(fmaxf(p.z, q.z), fmaxf(r.z, t.z)) + b.z;
        v.w = fmaxf(fmaxf(p.w, q.w), fmaxf(r.w, t.w)) + b.w;
    } else {
        if (a.up) { my >>= 1; mx >>= 1; }
        v = ld(my, mx);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}

// channels-last in and out, C % 4 == 0: one thread = 4 channels of one output pixel; grid (ceil(Wo * C/4 / 256), Ho, N)
template <bool POOL>
__global__ __launch_bounds__(256) void glue_nhwc_kernel(GlueLArgs a) {
    const int c4n = a.C >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.Wo * c4n) return;
    const int ox = i / c4n, c = (i - ox * c4n) * 4;
    const int oy = blockIdx.y, n = blockIdx.z;
    const float4 v = glue_value4_nhwc<POOL>(a, n, c, oy, ox);
    *reinterpret_cast<float4*>(a.out + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.C + c) = v;
}

// The same for C / 4 a power of two (every VGG width: 64 / 128 / 256 / 512 channels) — round 5: the kernel above moves ONE
// 16-byte piece per thread behind an integer division (4 KB per block: 3.4-5.5 TB/s over the codec's tensors, the largest
// HBM-bound class of a bench step).  Here a thread moves U pieces of U consecutive output ROWS at the same (pixel, channel
// quad): the index arithmetic (a shift, the reflection of the column) is shared, U independent 16-byte loads are in
// flight before the first store, and a block covers U rows.  grid (ceil(Wo * C/4 / 256), ceil(Ho / U), N)
template <bool POOL, int U>
__global__ __launch_bounds__(256) void glue_nhwc_rows_kernel(GlueLArgs a, int c4shift) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ox = i >> c4shift;
    if (ox >= a.Wo) return;
    const int c = (i & ((1 << c4shift) - 1)) * 4;
    const int n = blockIdx.z, oy0 = blockIdx.y * U;
    const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    int mx = reflect_index(ox - a.pad, a.Wm);
    if (!POOL && a.up) mx >>= 1;
    const size_t row_in = (size_t)a.W * a.C;
    const float* __restrict__ xin = a.x + (size_t)n * a.H * row_in + (size_t)(POOL ? 2 * mx : mx) * a.C + c;
    const int dx = (POOL && 2 * mx + 1 < a.W) ? a.C : 0;   // ceil_mode partial windows
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int oy = oy0 + u < a.Ho ? oy0 + u : a.Ho - 1;   // (rows past the end re-read the last one: no divergent loads)
        int my = reflect_index(oy - a.pad, a.Hm);
        if (POOL) {
            const int y0 = 2 * my, y1 = (y0 + 1 < a.H) ? y0 + 1 : y0;
            const float4 p = *reinterpret_cast<const float4*>(xin + (size_t)y0 * row_in);
            const float4 q = *reinterpret_cast<const float4*>(xin + (size_t)y0 * row_in + dx);
            const float4 r = *reinterpret_cast<const float4*>(xin + (size_t)y1 * row_in);
            const float4 t = *reinterpret_cast<const float4*>(xin + (size_t)y1 * row_in + dx);
            v[u].x = fmaxf(fmaxf(p.x, q.x), fmaxf(r.x, t.x));
            v[u].y = fmaxf(fmaxf(p.y, q.y), fmaxf(r.y, t.y));
            v[u].z = fmaxf(fmaxf(p.z, q.z), fmaxf(r.z, t.z));
            v[u].w = fmaxf(fmaxf(p.w, q.w), fmaxf(r.w, t.w));
        } else {
            if (a.up) my >>= 1;
            v[u] = *reinterpret_cast<const float4*>(xin + (size_t)my * row_in);
        }
    }
    float* __restrict__ o = a.out + (((size_t)n * a.Ho + oy0) * a.Wo + ox) * a.C + c;
    const size_t row_out = (size_t)a.Wo * a.C;
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (oy0 + u < a.Ho) {
            float4 w = v[u];
            w.x += b.x; w.y += b.y; w.z += b.z; w.w += b.w;
            if (a.relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
            *reinterpret_cast<float4*>(o + (size_t)u * row_out) = w;
        }
    }
}

// mixed layouts: a 32 (pixels of one output row) x 32 (channels) tile per 256-thread block, transposed through LDS.
// Reads run along the input's fastest dimension, writes along the output's.  grid (ceil(Wo/32) * ceil(C/32), Ho, N)
template <bool IN_NHWC, bool POOL>
__global__ __launch_bounds__(256) void glue_transpose_kernel(GlueLArgs a) {
    __shared__ float tile[32][33];
    const int ctiles = (a.C + 31) / 32;
    const int ox0 = (blockIdx.x / ctiles) * 32, c0 = (blockIdx.x % ctiles) * 32;
    const int oy = blockIdx.y, n = blockIdx.z;
    const int lo = threadIdx.x & 31, hi = threadIdx.x >> 5;  // hi = 0..7
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // read side: the lane index runs along the input's fastest dimension (x for planar, c for channels-last)
        const int px = IN_NHWC ? hi + 8 * k : lo, ch = IN_NHWC ? lo : hi + 8 * k;
        const int ox = ox0 + px, c = c0 + ch;
        if (ox < a.Wo && c < a.C) tile[px][ch] = glue_value<IN_NHWC, POOL>(a, n, c, oy, ox, a.bias ? a.bias[c] : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // write side: the lane index runs along the output's fastest dimension (the other one)
        const int px = IN_NHWC ? lo : hi + 8 * k, ch = IN_NHWC ? hi + 8 * k : lo;
        const int ox = ox0 + px, c = c0 + ch;
        if (ox < a.Wo && c < a.C) {
            const size_t o = IN_NHWC ? (((size_t)n * a.C + c) * a.Ho + oy) * a.Wo + ox      // out planar
                                     : (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.C + c;     // out channels-last
            a.out[o] = tile[px][ch];
        }
    }
}

// mixed layouts, C % 4 == 0: TP (64 or 128) pixels x 32 channels per block; the channels-last side moves 16 bytes per
// lane (128-byte spans per pixel), the planar side TP * 4-byte spans per channel row.  TP = 128 for rows of 128 pixels
// and more: twice the bytes in flight per thread (the 64-pixel tile ran latency-bound: 3.5 TB/s on the 4.3 GB
// relu1 tensor against 5.6 TB/s of the same-layout kernel).
template <bool IN_NHWC, bool POOL, int TP, int TC = 32>
__global__ __launch_bounds__(256) void glue_transpose_wide_kernel(GlueLArgs a) {
    constexpr int GL = TC / 4;                           // lanes per pixel on the channels-last side
    __shared__ float tile[TC][TP + 1];
    const int ctiles = (a.C + TC - 1) / TC;
    const int ox0 = (blockIdx.x / ctiles) * TP, c0 = (blockIdx.x % ctiles) * TC;
    const int oy = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x;
    const int g = tid & (GL - 1), pp = tid / GL;         // channels-last side: GL lanes x float4 = TC channels of pixel pp (+256 / GL, ...)
    constexpr int PPS = 256 / GL;                        // pixels a pass on the channels-last side
    const int px = tid & (TP - 1), cq = tid / TP;        // planar side: TP pixels of channel cq (+ 256 / TP, ...)
    constexpr int ROWS = 256 / TP;                       // channel rows a pass on the planar side (4-byte accesses)
    constexpr int ROWS2 = 512 / TP;                      // ... with 8-byte accesses
    if (IN_NHWC) {
#pragma unroll
        for (int k = 0; k < TP / PPS; k++) {
            const int p = pp + PPS * k, ox = ox0 + p, c = c0 + 4 * g;
            if (ox < a.Wo && c < a.C) {
                const float4 v = glue_value4_nhwc<POOL>(a, n, c, oy, ox);
                tile[4 * g + 0][p] = v.x; tile[4 * g + 1][p] = v.y; tile[4 * g + 2][p] = v.z; tile[4 * g + 3][p] = v.w;
            }
        }
        __syncthreads();
        if (a.Wo % 2 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7u) == 0) {
            // even rows: 8-byte stores (the planar side is store-issue-bound), TP / 2 lanes x 2 pixels per channel row
            const int p2 = (tid & (TP / 2 - 1)) * 2, cr = tid / (TP / 2);
#pragma unroll
            for (int k = 0; k < TC / ROWS2; k++) {
                const int ch = cr + ROWS2 * k, ox = ox0 + p2, c = c0 + ch;
                if (ox < a.Wo && c < a.C)   // Wo and ox even: ox + 1 < Wo as well
                    *reinterpret_cast<float2*>(a.out + (((size_t)n * a.C + c) * a.Ho + oy) * a.Wo + ox) =
                        make_float2(tile[ch][p2], tile[ch][p2 + 1]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < TC / ROWS; k++) {
                const int ch = cq + ROWS * k, ox = ox0 + px, c = c0 + ch;
                if (ox < a.Wo && c < a.C) a.out[(((size_t)n * a.C + c) * a.Ho + oy) * a.Wo + ox] = tile[ch][px];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < TC / ROWS; k++) {
            const int ch = cq + ROWS * k, ox = ox0 + px, c = c0 + ch;
            if (ox < a.Wo && c < a.C) tile[ch][px] = glue_value<false, POOL>(a, n, c, oy, ox, a.bias ? a.bias[c] : 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TP / PPS; k++) {
            const int p = pp + PPS * k, ox = ox0 + p, c = c0 + 4 * g;
            if (ox < a.Wo && c < a.C)
                *reinterpret_cast<float4*>(a.out + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.C + c) =
                    make_float4(tile[4 * g + 0][p], tile[4 * g + 1][p], tile[4 * g + 2][p], tile[4 * g + 3][p]);
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

extern "C" int optex_vgg_glue_layout(const float* x, const float* bias, float* out, int N, int C, int H, int W, int relu,
                                     int pool, int up, int pad, int in_nhwc, int out_nhwc, void* stream) {
    if (!in_nhwc && !out_nhwc) return optex_vgg_glue(x, bias, out, N, C, H, W, relu, pool, up, pad, stream);
    if (!x || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (pool && up) || pad < 0 || pad > 1) {
        set_error("optex_vgg_glue_layout: bad argument (N=%d C=%d H=%d W=%d pool=%d up=%d pad=%d)", N, C, H, W, pool, up, pad);
        return OPTEX_E_ARG;
    }
    GlueLArgs a;
    a.x = x; a.bias = bias; a.out = out;
    a.N = N; a.C = C; a.H = H; a.W = W;
    a.Hm = pool ? (H + 1) / 2 : (up ? 2 * H : H);
    a.Wm = pool ? (W + 1) / 2 : (up ? 2 * W : W);
    a.Ho = a.Hm + 2 * pad;
    a.Wo = a.Wm + 2 * pad;
    a.relu = relu; a.pool = pool; a.up = up; a.pad = pad;
    if (pad && (a.Hm < 2 || a.Wm < 2)) {
        set_error("optex_vgg_glue_layout: reflection padding needs at least 2 pixels per side (got %d x %d)", a.Hm, a.Wm);
        return OPTEX_E_ARG;
    }
    if (a.Ho > 65535 || N > 65535) {
        set_error("optex_vgg_glue_layout: Ho = %d / N = %d exceed the grid limit", a.Ho, N);
        return OPTEX_E_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(KC_GLUE, st, 0.0, 4.0 * ((double)N * C * H * W + (double)N * C * a.Ho * a.Wo));
    const bool vec = in_nhwc && out_nhwc && C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(out) % 16 == 0) && (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0);
    if (vec) {
        const int c4 = C / 4;
        if ((c4 & (c4 - 1)) == 0 && a.Ho >= 8) {  // every VGG width; four rows per thread
            int sh = 0;
            while ((1 << sh) < c4) sh++;
            constexpr int U = 4;
            dim3 grid((unsigned)(((long long)a.Wo * c4 + 255) / 256), (unsigned)((a.Ho + U - 1) / U), (unsigned)N);
            if (pool) hipLaunchKernelGGL((glue_nhwc_rows_kernel<true, U>), grid, dim3(256), 0, st, a, sh);
            else hipLaunchKernelGGL((glue_nhwc_rows_kernel<false, U>), grid, dim3(256), 0, st, a, sh);
            return check_launch("glue_nhwc_rows_kernel");
        }
        dim3 grid((unsigned)(((long long)a.Wo * (C / 4) + 255) / 256), (unsigned)a.Ho, (unsigned)N);
        if (pool) hipLaunchKernelGGL(glue_nhwc_kernel<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(glue_nhwc_kernel<false>, grid, dim3(256), 0, st, a);
    } else if (in_nhwc != out_nhwc && C % 4 == 0 && (reinterpret_cast<uintptr_t>(in_nhwc ? x : out) % 16 == 0) &&
               (!bias || !in_nhwc || reinterpret_cast<uintptr_t>(bias) % 16 == 0)) {
        const bool wide = a.Wo >= 128;
        const int tp = wide ? 128 : 64;
        // channels-last -> planar with C % 64 == 0: 64 channels per tile, 256-byte spans per pixel on the read side (4.03 ->
        // 4.25 TB/s; the other direction loses with it: 3.6 -> 2.3 TB/s)
        const bool deep = wide && in_nhwc && C % 64 == 0;
        dim3 grid((unsigned)(((a.Wo + tp - 1) / tp) * ((C + (deep ? 63 : 31)) / (deep ? 64 : 32))), (unsigned)a.Ho, (unsigned)N);
        if (deep) {
            if (pool) hipLaunchKernelGGL((glue_transpose_wide_kernel<true, true, 128, 64>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((glue_transpose_wide_kernel<true, false, 128, 64>), grid, dim3(256), 0, st, a);
        } else if (wide) {
            if (in_nhwc) {
                if (pool) hipLaunchKernelGGL((glue_transpose_wide_kernel<true, true, 128>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((glue_transpose_wide_kernel<true, false, 128>), grid, dim3(256), 0, st, a);
            } else {
                if (pool) hipLaunchKernelGGL((glue_transpose_wide_kernel<false, true, 128>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((glue_transpose_wide_kernel<false, false, 128>), grid, dim3(256), 0, st, a);
            }
        } else if (in_nhwc) {
            if (pool) hipLaunchKernelGGL((glue_transpose_wide_kernel<true, true, 64>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((glue_transpose_wide_kernel<true, false, 64>), grid, dim3(256), 0, st, a);
        } else {
            if (pool) hipLaunchKernelGGL((glue_transpose_wide_kernel<false, true, 64>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((glue_transpose_wide_kernel<false, false, 64>), grid, dim3(256), 0, st, a);
        }
    } else if (in_nhwc != out_nhwc) {
        dim3 grid((unsigned)(((a.Wo + 31) / 32) * ((C + 31) / 32)), (unsigned)a.Ho, (unsigned)N);
        if (in_nhwc) {
            if (pool) hipLaunchKernelGGL((glue_transpose_kernel<true, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((glue_transpose_kernel<true, false>), grid, dim3(256), 0, st, a);
        } else {
            if (pool) hipLaunchKernelGGL((glue_transpose_kernel<false, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((glue_transpose_kernel<false, false>), grid, dim3(256), 0, st, a);
        }
    } else {
        set_error("optex_vgg_glue_layout: channels-last on both sides needs C %% 4 == 0 and 16-byte aligned tensors (C=%d)", C);
        return OPTEX_E_UNSUPPORTED;
    }
    return check_launch("glue_layout_kernel");
}
