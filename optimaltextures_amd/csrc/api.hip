// api.hip — error plumbing and device queries for the C ABI in include/optex.h
#include <mutex>
#include <vector>

#include "optex_common.h"

namespace optex {

static thread_local char g_err[512] = "";
thread_local CallOpts tl_call;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return OPTEX_E_LAUNCH;
    }
    return OPTEX_OK;
}

int check_ws(const char* fn, const void* ws, size_t have, size_t need) {
    if (need == 0) return OPTEX_OK;
    if (!ws || have < need) {
        set_error("%s: scratch buffer too small: ws_bytes = %zu, this call needs %zu (see the *_ws_bytes helper)", fn, have, need);
        return OPTEX_E_ARG;
    }
    return OPTEX_OK;
}

int device_cu_count() {
    static thread_local int cached_dev = -1, cached_cu = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cached_cu = v;
        cached_dev = dev;
    }
    return cached_cu;
}

// ---------------------------------------------------------------------------------------------- copies and fills as kernels
__global__ __launch_bounds__(256) void copy_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t count, int vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const size_t q = count / 4;
        for (size_t i = i0; i < q; i += stride) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        for (size_t i = 4 * q + i0; i < count; i += stride) dst[i] = src[i];
    } else {
        for (size_t i = i0; i < count; i += stride) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ dst, uint32_t value, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = value;
}

int device_copy(float* dst, const float* src, size_t count, hipStream_t st) {
    if (count == 0 || dst == src) return OPTEX_OK;
    const int vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0 ? 1 : 0;
    const size_t work = vec ? (count + 3) / 4 : count;
    size_t blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dst, src, count, vec);
    return check_launch("copy_kernel");
}

int device_fill_u32(uint32_t* dst, uint32_t value, size_t count, hipStream_t st) {
    if (count == 0) return OPTEX_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dst, value, count);
    return check_launch("fill_u32_kernel");
}

// ---------------------------------------------------------------------------------------------- profiler
struct ProfState {
    bool on = false;
    std::vector<hipEvent_t> pool;
    struct Rec { int cls; hipEvent_t e0, e1; };
    std::vector<Rec> recs;
    double flops[KC_COUNT] = {0}, bytes[KC_COUNT] = {0};
    long long launches[KC_COUNT] = {0};
    std::mutex mu;
};
static ProfState g_prof;

static hipEvent_t prof_event() {
    if (!g_prof.pool.empty()) {
        hipEvent_t e = g_prof.pool.back();
        g_prof.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int c, hipStream_t s, double flops, double bytes) : cls(c), st(s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    e0 = prof_event();
    e1 = prof_event();
    g_prof.flops[cls] += flops;
    g_prof.bytes[cls] += bytes;
    g_prof.launches[cls] += 1;
    if (e0) (void)hipEventRecord(e0, st);
}

ProfScope::~ProfScope() {
    if (!e0 || !e1) return;
    (void)hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.recs.push_back({cls, e0, e1});
}

static const char* kClassNames[KC_COUNT] = {"gemm_tn", "col_minmax", "col_hist", "cdf_lut", "cdf_apply", "sort_columns",
                                            "sort_match", "col_mean", "gram", "cov_finalize", "householder", "interp",
                                            "sort_radix_sweep", "vgg_glue", "linalg_gemm", "chol_inv", "ns_init",
                                            "legacy_normals", "cdf_match"};

}  // namespace optex

extern "C" int optex_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(optex::g_prof.mu);
    optex::g_prof.on = on != 0;
    return OPTEX_OK;
}

extern "C" int optex_prof_num_classes(void) { return optex::KC_COUNT; }

extern "C" const char* optex_prof_class_name(int cls) {
    return (cls >= 0 && cls < optex::KC_COUNT) ? optex::kClassNames[cls] : "?";
}

// Waits for the recorded events (this is the one call of the library that blocks), sums the elapsed time per class
// into ms[], copies launch counts and algorithmic flops / bytes, then resets the counters.
extern "C" int optex_prof_collect(int n, double* ms, long long* launches, double* flops, double* bytes) {
    using namespace optex;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int i = 0; i < n && i < KC_COUNT; i++) {
        if (ms) ms[i] = 0.0;
        if (launches) launches[i] = g_prof.launches[i];
        if (flops) flops[i] = g_prof.flops[i];
        if (bytes) bytes[i] = g_prof.bytes[i];
    }
    for (auto& r : g_prof.recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess && ms &&
            r.cls < n)
            ms[r.cls] += (double)t;
        g_prof.pool.push_back(r.e0);
        g_prof.pool.push_back(r.e1);
    }
    g_prof.recs.clear();
    for (int i = 0; i < KC_COUNT; i++) {
        g_prof.launches[i] = 0;
        g_prof.flops[i] = g_prof.bytes[i] = 0.0;
    }
    return OPTEX_OK;
}

extern "C" int optex_abi_version(void) { return OPTEX_ABI_VERSION; }

extern "C" const char* optex_last_error(void) { return optex::g_err; }

extern "C" int optex_device_info(int* n_cu, int* lds_bytes, int* wavefront) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        optex::set_error("optex_device_info: %s", hipGetErrorString(e));
        return OPTEX_E_LAUNCH;
    }
    int v = 0;
    if (n_cu) {
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        *n_cu = v;
    }
    if (lds_bytes) {
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        *lds_bytes = v;
    }
    if (wavefront) {
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeWarpSize, dev);
        *wavefront = v;
    }
    return OPTEX_OK;
}
