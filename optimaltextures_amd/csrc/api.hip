// api.hip — error plumbing and device queries for the C ABI in include/optex.h
#include "optex_common.h"

namespace optex {

static thread_local char g_err[512] = "";

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

}  // namespace optex

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
