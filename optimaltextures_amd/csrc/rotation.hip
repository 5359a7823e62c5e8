// rotation.hip — R0: Haar-random SO(N) matrices, optex.py:142-149 -> scipy.stats.special_ortho_group.rvs.
//
// scipy runs N-1 Householder reflections in a Python loop on the host (140 ms at N = 256).  The reflections act on
// the ROWS of H independently ( H[i, n:] -= (H[i, n:] . x_n) x_n ), so the O(N^3) part is embarrassingly parallel over
// rows: one wavefront per row keeps its row in registers and replays the N-1 reflections in order, in fp64 like scipy.
// The random stream itself is numpy's legacy MT19937 gaussian stream (so that np.random.seed reproduces the reference's
// matrices).  It can stay on the host (the normals are then uploaded once per batch of rotations) or run on the device:
// legacy_normals_kernel below advances a stream from its 624-word state exactly as numpy's RandomState.normal does.
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

// ------------------------------------------------------------------------------------------------ numpy's legacy gaussian stream
// RandomState.normal(size) = MT19937 (Matsumoto & Nishimura) + numpy's legacy polar method with its one-value cache
// (numpy/random/src/legacy/legacy-distributions.c legacy_gauss; restated on the host in oracle/optex_oracle.c:60-131):
//     do { x1 = 2 u() - 1; x2 = 2 u() - 1; r2 = x1 x1 + x2 x2; } while (r2 >= 1 || r2 == 0);
//     f = sqrt(-2 log(r2) / r2);  return f x2, then (cached) f x1;       u() = ((w >> 5) 2^26 + (w' >> 6)) / 2^53
// Sequential as written, parallel in fact: an ATTEMPT always consumes exactly four 32-bit words whether it is accepted or
// not, so attempt j of a stretch of the stream is a pure function of words 4j .. 4j + 3, and the outputs are the accepted
// attempts in order (a prefix sum of the accept flags).  The MT19937 recurrence x[k + 624] = x[k + 397] ^ t(x[k], x[k + 1])
// is itself parallel 227 wide.  Two stages: one workgroup per stream regenerates the 624-word blocks (three 227-wide phases, out
// of place: no read-before-write hazards), tempers them, takes the accept / reject decisions of a block's <= 157 attempts
// and writes the accepted (x1, x2, r2) behind each other; then one thread per accepted pair, on all CUs, does the
// expensive part (the logarithm).  State in / out: 624 key words, the position inside the
// block, the cache flag and the cached value — numpy's RandomState.get_state() tuple, so a stream can be handed over
// from / to the host at any point.  Every operation is IEEE (no contraction, correctly rounded division and square root)
// and identical to the host's, except log(): log_unit_interval below is correctly rounded, glibc's log is accurate to
// 0.52 ulp and disagrees with the correctly rounded value for about one argument in 2000 (measured), so a draw equals
// numpy's bit for bit in >= 99.8 % of the cases and differs by a few units in the last place of the double otherwise
// (tests/test_gpu_parity.py::test_device_normals_follow_numpy_stream; the words, the accept / reject decisions and the
// state are exact).
constexpr int MT_N = 624, MT_M = 397, MT_STATE_WORDS = MT_N + 4;   // key, pos, has_gauss, gauss (2 words)

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t far) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// log(x) for 0 < x < 1, correctly rounded (error before the final rounding < 2^-67 relative; 0 misses in 15 000 arguments
// against a 50-digit reference, glibc's log: 7).  (The device library's log() is specified to 1 ulp, not enough here: a
// one-ulp logarithm can move f * x by two or three.)   x = m 2^e, m in [sqrt(1/2), sqrt(2)),  log x = e ln 2 + 2 atanh(s),
// s = (m - 1) / (m + 1), atanh(s) / s = 1 + z / 3 + z^2 / 5 + z^3 / 7 + z^4 (1/9 + z/11 + ...),  z = s^2 <= 0.0295 —
// the first four terms in double-double arithmetic (error-free transformations on fma), the tail in plain fp64.
struct dd2 { double h, l; };
__device__ __forceinline__ dd2 dd_fast(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }   // |a| >= |b|
__device__ __forceinline__ dd2 dd_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd2 dd_add(dd2 x, dd2 y) {
    const dd2 s = dd_sum(x.h, y.h);
    return dd_fast(s.h, s.l + (x.l + y.l));
}
__device__ __forceinline__ dd2 dd_mul(dd2 x, dd2 y) {
    const double p = x.h * y.h;
    const double e = __builtin_fma(x.h, y.h, -p) + (x.h * y.l + x.l * y.h);
    return dd_fast(p, e);
}
__device__ __forceinline__ dd2 dd_div(dd2 x, dd2 y) {   // three quotient digits; each remainder is exact to double-double
    const double inv = 1.0 / y.h;
    const double q1 = x.h * inv;
    dd2 r = dd_add(x, dd_mul(y, dd2{-q1, 0.0}));
    const double q2 = r.h * inv;
    r = dd_add(r, dd_mul(y, dd2{-q2, 0.0}));
    const double q3 = r.h * inv;
    return dd_add(dd_fast(q1, q2), dd2{q3, 0.0});
}

__device__ double log_unit_interval(double x) {
    int e;
    double m = frexp(x, &e);          // m in [0.5, 1)
    if (m < 0.70710678118654752440) {
        m *= 2.0;                     // exact
        e -= 1;
    }
    const dd2 num = {m - 1.0, 0.0};   // exact (Sterbenz)
    const dd2 den = dd_sum(m, 1.0);
    const dd2 sdd = dd_div(num, den);
    const dd2 z = dd_mul(sdd, sdd);
    // tail: z^4 (1/9 + z/11 + ... + z^9/27), plain fp64 (it is below 1e-6 of the sum)
    const double zh = z.h;
    double t = 1.0 / 27.0;
    t = t * zh + 1.0 / 25.0;
    t = t * zh + 1.0 / 23.0;
    t = t * zh + 1.0 / 21.0;
    t = t * zh + 1.0 / 19.0;
    t = t * zh + 1.0 / 17.0;
    t = t * zh + 1.0 / 15.0;
    t = t * zh + 1.0 / 13.0;
    t = t * zh + 1.0 / 11.0;
    t = t * zh + 1.0 / 9.0;
    // 1/3, 1/5, 1/7 as double-doubles (high part = the double nearest to the fraction, low part = the remainder)
    const dd2 c3 = {0.33333333333333331, 1.8503717077085941e-17};
    const dd2 c5 = {0.20000000000000001, -1.1102230246251566e-17};
    const dd2 c7 = {0.14285714285714285, 7.9301644616082610e-18};
    dd2 p = dd_add(c7, dd2{zh * t, 0.0});
    p = dd_add(c5, dd_mul(z, p));
    p = dd_add(c3, dd_mul(z, p));
    p = dd_add(dd2{1.0, 0.0}, dd_mul(z, p));
    dd2 lm = dd_mul(sdd, p);
    lm.h *= 2.0;
    lm.l *= 2.0;
    // e ln 2: the high part of ln 2 has its low 11 bits clear, so e * ln2_h is exact for |e| <= 1074
    const double ln2_h = 0.69314718055989033, ln2_l = 5.4979230187083712e-14 + 0.0;
    const dd2 el = dd_add(dd2{(double)e * ln2_h, 0.0}, dd_sum((double)e * ln2_l, 0.0));
    const dd2 r = dd_add(el, lm);
    return r.h + r.l;
}

__device__ __forceinline__ double mt_unit(uint32_t w0, uint32_t w1) {
    return ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) * (1.0 / 9007199254740992.0);   // (a power of two: exact)
}

// Stage 1, one workgroup of four wavefronts per stream (the only sequential part): the MT19937 words and the accept /
// reject decisions.  Accepted attempts leave (x1, x2, r2) behind each other in `pairs`; the stream stops exactly behind
// the attempt that completes the draw, so the state handed back is numpy's.  Per 624-word block: three 227-wide phases of
// the recurrence (out of place), the tempering, one attempt per thread, a ballot prefix across the four waves — five
// barriers, ~0.5 us.  (One wavefront alone, barrier-free, was tried: 2.2 us per block — a lone wave cannot issue its ~450
// dependent instructions per block any faster.)
struct NormalsMeta { long npairs; long lead; double cached; long pad; };   // per stream: pairs written, 1 if out[0] is the old cache

__global__ __launch_bounds__(256) void mt_accept_kernel(uint32_t* __restrict__ states, long count, double* __restrict__ pairs,
                                                        long pairs_stride, NormalsMeta* __restrict__ meta) {
    __shared__ uint32_t key[2][MT_N];   // the block, ping-pong
    __shared__ __attribute__((aligned(16))) uint32_t sw[MT_N + 4];   // tempered words still to be consumed: <= 3 carried over + the rest of the block
    __shared__ uint32_t wtot[4];
    __shared__ int s_stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* st = states + (size_t)blockIdx.x * MT_STATE_WORDS;
    double* pr = pairs + (size_t)blockIdx.x * pairs_stride;
    for (int i = tid; i < MT_N; i += 256) key[0][i] = st[i];
    // uniform bookkeeping: block in use, position inside it, words carried over, pairs delivered
    int cur = 0, pos = (int)st[MT_N], carry = 0;
    const int had = (int)st[MT_N + 1];
    const long lead = (count > 0 && had) ? 1 : 0;          // the cached second value of the last pair goes out first
    const long need = count - lead;                          // values to come from new pairs
    const long want = (need + 1) / 2;                        // pairs
    long done = 0;
    if (tid == 0) {
        NormalsMeta m;
        m.npairs = want;
        m.lead = lead;
        m.cached = *reinterpret_cast<const double*>(st + MT_N + 2);
        m.pad = 0;
        meta[blockIdx.x] = m;
    }
    __syncthreads();
    constexpr int W = MT_N - MT_M;  // 227: the recurrence is W wide
    while (done < want) {
        if (pos >= MT_N) {
            // next block, out of place: key[cur] -> key[cur ^ 1], three phases of up to 227 independent words
            const uint32_t* a = key[cur];
            uint32_t* b = key[cur ^ 1];
            if (tid < W) b[tid] = mt_twist(a[tid], a[tid + 1], a[tid + MT_M]);
            __syncthreads();
            if (tid < W) b[tid + W] = mt_twist(a[tid + W], a[tid + W + 1], b[tid]);
            __syncthreads();
            if (tid < MT_N - 2 * W) {
                const int k = tid + 2 * W;
                b[k] = mt_twist(a[k], k + 1 < MT_N ? a[k + 1] : b[0], b[k - W]);
            }
            __syncthreads();
            cur ^= 1;
            pos = 0;
        }
        // the rest of the block, tempered, behind the words carried over from the last one
        const int avail = MT_N - pos, total = carry + avail, nat = total >> 2;   // nat <= 157 attempts: one per thread
        {
            uint32_t v[3];
#pragma unroll
            for (int i = 0; i < 3; i++) v[i] = tid + 256 * i < avail ? key[cur][pos + tid + 256 * i] : 0u;
#pragma unroll
            for (int i = 0; i < 3; i++)
                if (tid + 256 * i < avail) sw[carry + tid + 256 * i] = mt_temper(v[i]);
        }
        __syncthreads();
        bool acc = false;
        double x1 = 0.0, x2 = 0.0, r2 = 0.0;
        if (tid < nat) {
            const uint4 wq = *reinterpret_cast<const uint4*>(sw + 4 * tid);   // the attempt's four words
            x1 = 2.0 * mt_unit(wq.x, wq.y) - 1.0;
            x2 = 2.0 * mt_unit(wq.z, wq.w) - 1.0;
            r2 = x1 * x1 + x2 * x2;
            acc = !(r2 >= 1.0 || r2 == 0.0);
        }
        const unsigned long long bal = __ballot(acc);
        if (lane == 0) wtot[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        long before = (long)__popcll(bal & ((1ull << lane) - 1ull));   // accepted attempts in front of mine
        for (int k = 0; k < wave; k++) before += wtot[k];
        const long accepted = (long)wtot[0] + wtot[1] + wtot[2] + wtot[3], room = want - done;
        if (acc && before < room) {
            double* p = pr + 3 * (done + before);
            p[0] = x1;
            p[1] = x2;
            p[2] = r2;
        }
        if (accepted >= room) {
            // the stream stops behind the attempt that delivers pair number want - 1: the words after it stay unconsumed
            if (acc && before == room - 1) s_stop = tid;
            __syncthreads();
            pos += 4 * (s_stop + 1) - carry;   // words of THIS block consumed from pos on (the carried ones were the last block's)
            carry = 0;
            done = want;
        } else {
            done += accepted;
            const int left = total - 4 * nat;  // up to three words are left over: they open the next round
            uint32_t keep = 0u;
            if (tid < left) keep = sw[4 * nat + tid];
            __syncthreads();
            if (tid < left) sw[tid] = keep;
            carry = left;
            pos = MT_N;
        }
        __syncthreads();
    }
    // hand the state back (the loop leaves through its stop branch: nothing is carried over here).  An odd number of new
    // values leaves the second value of the last pair in the cache: stage 2 knows it and stores it (state word 626).
    for (int i = tid; i < MT_N; i += 256) st[i] = key[cur][i];
    if (tid == 0) {
        st[MT_N] = (uint32_t)pos;
        if (count > 0) {
            st[MT_N + 1] = (uint32_t)(need & 1);
            if (!(need & 1)) {
                st[MT_N + 2] = 0u;
                st[MT_N + 3] = 0u;
            }
        }
    }
}

// Stage 2, one thread per accepted pair, all CUs: f = sqrt(-2 log(r2) / r2), values f x2 and f x1 in that order
__global__ __launch_bounds__(256) void normals_emit_kernel(uint32_t* __restrict__ states, long count, const double* __restrict__ pairs,
                                                           long pairs_stride, const NormalsMeta* __restrict__ meta,
                                                           double* __restrict__ out, long out_stride) {
    const int s = blockIdx.y;
    const NormalsMeta m = meta[s];
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    double* o = out + (size_t)s * out_stride;
    if (r == 0 && m.lead) o[0] = m.cached;
    if (r >= m.npairs) return;
    const double* p = pairs + (size_t)s * pairs_stride + 3 * r;
    const double x1 = p[0], x2 = p[1], r2 = p[2];
    const double f = sqrt(-2.0 * log_unit_interval(r2) / r2);
    const long at = m.lead + 2 * r;
    o[at] = f * x2;
    if (at + 1 < count) o[at + 1] = f * x1;
    else *reinterpret_cast<double*>(states + (size_t)s * MT_STATE_WORDS + MT_N + 2) = f * x1;   // the draw ends on a first value
}

// numpy's RandomState(seed) for a 32-bit integer seed (mt19937_seed / Knuth's init_genrand): key[0] = seed,
// key[i] = 1812433253 (key[i-1] ^ key[i-1] >> 30) + i;  pos = 624 (the first draw regenerates the block), cache empty.
// One lane per stream: 623 dependent steps, a few microseconds.
__global__ void mt_seed_kernel(uint32_t* __restrict__ states, int n_streams, uint32_t first_seed, uint32_t seed_stride) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    uint32_t* st = states + (size_t)s * MT_STATE_WORDS;
    uint32_t v = first_seed + seed_stride * (uint32_t)s;
    for (int i = 0; i < MT_N; i++) {
        st[i] = v;
        v = 1812433253u * (v ^ (v >> 30)) + (uint32_t)i + 1u;
    }
    st[MT_N] = (uint32_t)MT_N;
    st[MT_N + 1] = 0u;
    st[MT_N + 2] = 0u;
    st[MT_N + 3] = 0u;
}

}  // namespace optex

using namespace optex;

extern "C" int optex_mt19937_seed(void* states, int n_streams, uint32_t first_seed, uint32_t seed_stride, void* stream) {
    if (!states || n_streams <= 0) {
        set_error("optex_mt19937_seed: bad argument (n_streams=%d)", n_streams);
        return OPTEX_E_ARG;
    }
    hipLaunchKernelGGL(mt_seed_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, as_stream(stream), static_cast<uint32_t*>(states),
                       n_streams, first_seed, seed_stride);
    return check_launch("mt_seed_kernel");
}

extern "C" size_t optex_mt19937_state_bytes(void) { return (size_t)MT_STATE_WORDS * sizeof(uint32_t); }

static size_t normals_pairs(long count) { return (size_t)(count + 1) / 2 + 1; }

extern "C" size_t optex_legacy_normals_ws_bytes(int n_streams, long count) {
    if (n_streams <= 0 || count <= 0) return 0;
    return align_up((size_t)n_streams * sizeof(NormalsMeta), 256) + (size_t)n_streams * normals_pairs(count) * 3 * sizeof(double);
}

extern "C" int optex_legacy_normals(void* states, int n_streams, long count, double* out, long out_stride, void* ws, size_t ws_bytes,
                                    void* stream) {
    if (!states || !out || n_streams <= 0 || count < 0 || out_stride < count) {
        set_error("optex_legacy_normals: bad argument (n_streams=%d count=%ld out_stride=%ld)", n_streams, count, out_stride);
        return OPTEX_E_ARG;
    }
    if (count == 0) return OPTEX_OK;
    if (int rc = check_ws("optex_legacy_normals", ws, ws_bytes, optex_legacy_normals_ws_bytes(n_streams, count))) return rc;
    hipStream_t st = as_stream(stream);
    NormalsMeta* meta = static_cast<NormalsMeta*>(ws);
    double* pairs = reinterpret_cast<double*>(static_cast<char*>(ws) + align_up((size_t)n_streams * sizeof(NormalsMeta), 256));
    const long pstride = (long)normals_pairs(count) * 3;
    ProfScope prof(KC_NORMALS, st, 0.0, 8.0 * (double)count * n_streams);
    hipLaunchKernelGGL(mt_accept_kernel, dim3(n_streams), dim3(256), 0, st, static_cast<uint32_t*>(states), count, pairs, pstride, meta);
    int rc = check_launch("mt_accept_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(normals_emit_kernel, dim3((unsigned)((normals_pairs(count) + 255) / 256), n_streams), dim3(256), 0, st,
                       static_cast<uint32_t*>(states), count, pairs, pstride, meta, out, out_stride);
    return check_launch("normals_emit_kernel");
}

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
