// ot_loop.hip — the reference's hot loop (optex.py:112-117) for the modes that need no host-side factorization:
//   for it in range(iters):  x = hist_match(x @ R, style @ R, mode) @ R.T ;  x += strength * (content - x)
// enqueued back-to-back on one stream from C++ (no Python between launches).  Everything stays channel-major
// ([segment][channel][pixel] = NCHW memory), so no kernel in the loop transposes anything.
#include "optex_common.h"

using namespace optex;

namespace {
struct LoopWs {
    float* y;    // rotated pastiche [n_seg, C, n]
    float* ys;   // rotated style    [src_n_seg, C, ns]
    float* y2;   // second rotated buffer (fused rotations only: the re-rotation cannot run in place)
    float* P;    // [iters - 1, C, C] re-rotation matrices R_i^T R_{i+1} (fused rotations only)
    void* mode_ws;
    static size_t mode_bytes(int mode, long n, long ns, int C, int n_seg, int src_n_seg) {
        // the sort scratch depends on BOTH column lengths: pastiche columns longer than one LDS take the global radix
        return mode == 0 ? optex_cdf_ws_bytes(C, n_seg) : optex_sort_match_ws_bytes(n, ns, C, n_seg, src_n_seg);
    }
    static size_t bytes(int mode, long n, long ns, int C, int n_seg, int src_n_seg, int iters, int fused) {
        size_t b = align_up((size_t)n_seg * C * n * sizeof(float), 256) +
                   align_up((size_t)src_n_seg * C * ns * sizeof(float), 256) + mode_bytes(mode, n, ns, C, n_seg, src_n_seg);
        if (fused)
            b += align_up((size_t)n_seg * C * n * sizeof(float), 256) +
                 align_up((size_t)(iters > 1 ? iters - 1 : 1) * C * C * sizeof(float), 256);
        return b;
    }
    LoopWs(void* ws, long n, long ns, int C, int n_seg, int src_n_seg, int iters, int fused) {
        char* p = static_cast<char*>(ws);
        y = reinterpret_cast<float*>(p);
        p += align_up((size_t)n_seg * C * n * sizeof(float), 256);
        ys = reinterpret_cast<float*>(p);
        p += align_up((size_t)src_n_seg * C * ns * sizeof(float), 256);
        y2 = P = nullptr;
        if (fused) {
            y2 = reinterpret_cast<float*>(p);
            p += align_up((size_t)n_seg * C * n * sizeof(float), 256);
            P = reinterpret_cast<float*>(p);
            p += align_up((size_t)(iters > 1 ? iters - 1 : 1) * C * C * sizeof(float), 256);
        }
        mode_ws = p;
    }
};
}  // namespace

extern "C" size_t optex_ot_loop_ws_bytes(int mode, long n, long ns, int C, int n_seg, int src_n_seg, int iters,
                                         int fuse_rotations) {
    return LoopWs::bytes(mode, n, ns, C, n_seg, src_n_seg, iters, fuse_rotations);
}

extern "C" int optex_ot_loop(int mode, float* x, long n, int n_seg, const float* style, long ns, int src_n_seg, int C,
                             const float* R32, const float* Rt32, int iters, const float* content, float strength,
                             int fuse_rotations, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !style || !R32 || !Rt32 || !ws || n <= 0 || ns <= 0 || C < 2 || n_seg <= 0 || iters < 0) {
        set_error("optex_ot_loop: bad argument (n=%ld ns=%ld C=%d n_seg=%d iters=%d)", n, ns, C, n_seg, iters);
        return OPTEX_E_ARG;
    }
    if (mode != 0 && mode != 1) {
        set_error("optex_ot_loop: mode %d (0 = cdf, 1 = sort; the linear modes factorize on the host side)", mode);
        return OPTEX_E_ARG;
    }
    if (src_n_seg != 1 && src_n_seg != n_seg) {
        set_error("optex_ot_loop: style has %d segments, expected 1 or %d", src_n_seg, n_seg);
        return OPTEX_E_ARG;
    }
    if (fuse_rotations && content) {
        set_error("optex_ot_loop: fuse_rotations needs the un-rotated pastiche between iterations for the content blend");
        return OPTEX_E_ARG;
    }
    if (int rc = check_ws("optex_ot_loop", ws, ws_bytes,
                          optex_ot_loop_ws_bytes(mode, n, ns, C, n_seg, src_n_seg, iters, fuse_rotations)))
        return rc;
    LoopWs w(ws, n, ns, C, n_seg, src_n_seg, iters, fuse_rotations);
    hipStream_t st = as_stream(stream);
    const long xs = (long)C * n, ss = (long)C * ns;
    if (fuse_rotations && iters > 0) {
        // Re-association of optex.py:175 + :170 of the next iteration:  (m @ R_i^T) @ R_{i+1} == m @ (R_i^T R_{i+1}).
        // One feature-map GEMM per iteration instead of two; the C x C products P_i = R_i^T R_{i+1} cost nothing.
        // Same fp32 arithmetic contract (k-ordered fma chains), different association: results agree with the literal
        // loop to fp32 round-off per step (tests/test_gpu_parity.py), not bit for bit.
        int rc;
        if (iters > 1 &&
            (rc = optex_gemm_tn(R32, C, (long)C * C, R32 + (size_t)C * C, C, (long)C * C, OPTEX_CHANNEL_MAJOR, w.P, C,
                                (long)C * C, OPTEX_CHANNEL_MAJOR, C, C, C, iters - 1, nullptr, 0, nullptr, 0, nullptr, 0.f,
                                stream)))
            return rc;
        float* cur = w.y;
        float* nxt = w.y2;
        if ((rc = optex_gemm_tn(R32, C, 0, x, n, xs, OPTEX_CHANNEL_MAJOR, cur, n, xs, OPTEX_CHANNEL_MAJOR, C, C, n, n_seg,
                                nullptr, 0, nullptr, 0, nullptr, 0.f, stream)))
            return rc;
        for (int it = 0; it < iters; it++) {
            const float* R = R32 + (size_t)it * C * C;
            if ((rc = optex_gemm_tn(R, C, 0, style, ns, ss, OPTEX_CHANNEL_MAJOR, w.ys, ns, ss, OPTEX_CHANNEL_MAJOR, C, C,
                                    ns, src_n_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, stream)))
                return rc;
            if (mode == 0)
                rc = cdf_match_impl(cur, n, xs, n, w.ys, ns, ss, ns, src_n_seg, C, n_seg, cur, n, xs, w.mode_ws, nullptr, st);
            else
                rc = sort_match_impl(cur, n, xs, n, w.ys, ns, ss, ns, src_n_seg, C, n_seg, cur, n, xs, w.mode_ws, st);
            if (rc) return rc;
            if (it + 1 < iters) {  // straight into the next iteration's rotated frame
                if ((rc = optex_gemm_tn(w.P + (size_t)it * C * C, C, 0, cur, n, xs, OPTEX_CHANNEL_MAJOR, nxt, n, xs,
                                        OPTEX_CHANNEL_MAJOR, C, C, n, n_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, stream)))
                    return rc;
                float* t = cur; cur = nxt; nxt = t;
            } else {               // optex.py:175 of the last iteration
                if ((rc = optex_gemm_tn(Rt32 + (size_t)it * C * C, C, 0, cur, n, xs, OPTEX_CHANNEL_MAJOR, x, n, xs,
                                        OPTEX_CHANNEL_MAJOR, C, C, n, n_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, stream)))
                    return rc;
            }
        }
        return OPTEX_OK;
    }
    for (int it = 0; it < iters; it++) {
        const float* R = R32 + (size_t)it * C * C;
        const float* Rt = Rt32 + (size_t)it * C * C;
        int rc;
        // optex.py:170  rotated_pastiche = pastiche_feature @ rotation
        if ((rc = optex_gemm_tn(R, C, 0, x, n, xs, OPTEX_CHANNEL_MAJOR, w.y, n, xs, OPTEX_CHANNEL_MAJOR, C, C, n, n_seg,
                                nullptr, 0, nullptr, 0, nullptr, 0.f, stream)))
            return rc;
        // optex.py:171  rotated_style = style_feature @ rotation
        if ((rc = optex_gemm_tn(R, C, 0, style, ns, ss, OPTEX_CHANNEL_MAJOR, w.ys, ns, ss, OPTEX_CHANNEL_MAJOR, C, C,
                                ns, src_n_seg, nullptr, 0, nullptr, 0, nullptr, 0.f, stream)))
            return rc;
        // optex.py:173  hist_match(rotated_pastiche, rotated_style), in place
        if (mode == 0)
            rc = cdf_match_impl(w.y, n, xs, n, w.ys, ns, ss, ns, src_n_seg, C, n_seg, w.y, n, xs, w.mode_ws, nullptr, st);
        else
            rc = sort_match_impl(w.y, n, xs, n, w.ys, ns, ss, ns, src_n_seg, C, n_seg, w.y, n, xs, w.mode_ws, st);
        if (rc) return rc;
        // optex.py:175 + 115-117  pastiche = matched @ rotation.T ; content blend
        if ((rc = optex_gemm_tn(Rt, C, 0, w.y, n, xs, OPTEX_CHANNEL_MAJOR, x, n, xs, OPTEX_CHANNEL_MAJOR, C, C, n, n_seg,
                                nullptr, 0, nullptr, 0, content, strength, stream)))
            return rc;
    }
    return OPTEX_OK;
}
