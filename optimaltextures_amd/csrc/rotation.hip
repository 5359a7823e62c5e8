// rotation.hip — R0: Haar-random SO(N) matrices, optex.py:142-149 -> scipy.stats.special_ortho_group.rvs.
//
// scipy runs N-1 Householder reflections in a Python loop on the host (140 ms at N = 256).  The reflections act on
// the ROWS of H independently ( H[i, n:] -= (H[i, n:] . x_n) x_n ), so the O(N^3) part is embarrassingly parallel over
// rows: one wavefront per row keeps its row in registers and replays the N-1 reflections in order, in fp64 like scipy.
// The random stream itself (numpy's legacy MT19937 gaussian stream, so that np.random.seed reproduces the reference's
// matrices) stays on the host; the normals are uploaded once per batch of rotations.
// Differences to scipy are fp64 summation-order effects (~1e-16), invisible after optex.py:168's cast to fp32 except
// for rare 1-ulp flips.
#include "optex_common.h"

namespace optex {

__host__ __device__ inline long refl_offset(int N, int n) { return (long)n * N - (long)n * (n - 1) / 2; }

// one wave per (rotation, reflection n): normalised Householder vector v_n and sign D[n]
__global__ __launch_bounds__(64) void householder_prep_kernel(const double* __restrict__ normals, int N, long per_rot,
                                                              double* __restrict__ V, double* __restrict__ D) {
    const int n = blockIdx.x, rot = blockIdx.y, lane = threadIdx.x;
    const int len = N - n;
    const double* x = normals + (size_t)rot * per_rot + refl_offset(N, n);
    double* v = V + (size_t)rot * per_rot + refl_offset(N, n);
    double s = 0.0;
    for (int j = lane; j < len; j += 64) s += x[j] * x[j];
    const double norm2 = wave_sum(s);
    const double x0 = x[0];
    const double d = (x0 != 0.0) ? ((x0 > 0.0) ? 1.0 : -1.0) : 1.0;
    const double x0n = x0 + d * sqrt(norm2);
    const double den = sqrt((norm2 - x0 * x0 + x0n * x0n) / 2.);
    for (int j = lane; j < len; j += 64) v[j] = ((j == 0) ? x0n : x[j]) / den;
    if (lane == 0) D[(size_t)rot * N + n] = d;
}

// one wave per (rotation, row i)
template <int NQ>
__global__ __launch_bounds__(64) void householder_apply_kernel(const double* __restrict__ V, const double* __restrict__ D,
                                                               int N, long per_rot, double* __restrict__ R64,
                                                               float* __restrict__ R32, float* __restrict__ Rt32) {
    const int i = blockIdx.x, rot = blockIdx.y, lane = threadIdx.x;
    const double* vr = V + (size_t)rot * per_rot;
    double hrow[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) hrow[q] = (lane + 64 * q == i) ? 1.0 : 0.0;
    for (int n = 0; n < N - 1; n++) {
        const double* v = vr + refl_offset(N, n) - n;  // v[j] for column j >= n
        double vv[NQ];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int j = lane + 64 * q;
            vv[q] = (j >= n && j < N) ? v[j] : 0.0;
            s += hrow[q] * vv[q];
        }
        s = wave_sum(s);
#pragma unroll
        for (int q = 0; q < NQ; q++) hrow[q] -= s * vv[q];
    }
    // D[N-1] = (-1)^(N-1) * prod(D[:-1]); rows scaled by D
    const double* dr = D + (size_t)rot * N;
    double di;
    if (i < N - 1) {
        di = dr[i];
    } else {
        double p = 1.0;
        for (int k = 0; k < N - 1; k++) p *= dr[k];
        di = (((N - 1) & 1) ? -1.0 : 1.0) * p;
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int j = lane + 64 * q;
        if (j < N) {
            const double val = hrow[q] * di;
            const size_t o = (size_t)rot * N * N;
            if (R64) R64[o + (size_t)i * N + j] = val;
            if (R32) R32[o + (size_t)i * N + j] = (float)val;
            if (Rt32) Rt32[o + (size_t)j * N + i] = (float)val;
        }
    }
}

}  // namespace optex

using namespace optex;

extern "C" long optex_rotation_normals(int N) { return (long)N * (N + 1) / 2 - 1; }

extern "C" size_t optex_rotation_ws_bytes(int N, int count) {
    const size_t per = (size_t)optex_rotation_normals(N);
    return align_up((size_t)count * per * sizeof(double), 256) + align_up((size_t)count * N * sizeof(double), 256);
}

extern "C" int optex_rotations_from_normals(const double* normals, int N, int count, double* R64, float* R32,
                                            float* Rt32, void* ws, size_t ws_bytes, void* stream) {
    if (!normals || !ws || N < 2 || count <= 0) {
        set_error("optex_rotations_from_normals: Dimension of rotation must be specified, and must be a scalar greater "
                  "than 1 (N=%d count=%d)", N, count);
        return OPTEX_E_ARG;
    }
    if (N > 1024) {
        set_error("optex_rotations_from_normals: N = %d > 1024 is not supported", N);
        return OPTEX_E_UNSUPPORTED;
    }
    if (int rc = check_ws("optex_rotations_from_normals", ws, ws_bytes, optex_rotation_ws_bytes(N, count))) return rc;
    hipStream_t st = as_stream(stream);
    const long per = optex_rotation_normals(N);
    double* V = static_cast<double*>(ws);
    double* D = reinterpret_cast<double*>(static_cast<char*>(ws) + align_up((size_t)count * per * sizeof(double), 256));
    ProfScope prof(KC_ROTGEN, st, 2.0 * (double)N * N * N * count, 8.0 * (double)per * count + 16.0 * N * N * count);
    hipLaunchKernelGGL(householder_prep_kernel, dim3(N - 1, count), dim3(64), 0, st, normals, N, per, V, D);
    int rc = check_launch("householder_prep_kernel");
    if (rc) return rc;
    const int nq = (N + 63) / 64;
    dim3 grid(N, count);
#define OPTEX_HH(Q) hipLaunchKernelGGL(householder_apply_kernel<Q>, grid, dim3(64), 0, st, V, D, N, per, R64, R32, Rt32)
    if (nq <= 1) OPTEX_HH(1);
    else if (nq <= 2) OPTEX_HH(2);
    else if (nq <= 4) OPTEX_HH(4);
    else if (nq <= 8) OPTEX_HH(8);
    else OPTEX_HH(16);
#undef OPTEX_HH
    return check_launch("householder_apply_kernel");
}
