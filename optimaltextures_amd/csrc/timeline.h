// timeline.h — PROBE BUILDS ONLY (-DOPTEX_TIMELINE, scripts/gemm_timeline_probe.hip): s_memtime stamps inside the GEMM
// kernels.  Every wavefront owns a slab of 64-bit words in tl_buf and appends one shader-clock stamp per call through the
// SCALAR store path: no vector register, no exec-mask branch, one s_memtime + s_store_dwordx2 per stamp (the library build
// compiles none of this: TL_STAMP expands to nothing).
#pragma once
#ifdef OPTEX_TIMELINE
namespace optex {

// one copy per translation unit (gemm_rs.hip and gemm.hip are compiled apart): the probe sets the copy of the kernel it
// launches through that unit's setter, TL_DEFINE_SETTER(name)
static __device__ unsigned long long* tl_buf;
static __device__ int tl_words;          // words per wavefront slab

typedef unsigned long long* tl_ptr;

// the wavefront's slab, as a uniform (SGPR) pointer
__device__ __forceinline__ tl_ptr tl_begin(unsigned wave_slot) {
    const uintptr_t u = reinterpret_cast<uintptr_t>(tl_buf + (size_t)wave_slot * (size_t)tl_words);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<tl_ptr>(((uintptr_t)hi << 32) | (uintptr_t)lo);
}
// (a cursor that has been through a spill comes back in vector registers: pin it to an SGPR pair again)
__device__ __forceinline__ tl_ptr tl_uniform(tl_ptr p) {
    const uintptr_t u = reinterpret_cast<uintptr_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<tl_ptr>(((uintptr_t)hi << 32) | (uintptr_t)lo);
}
// (the first wait drains the previous stamp's store — long gone — so that its data registers may be reused)
__device__ __forceinline__ void tl_stamp(tl_ptr& p) {
    unsigned long long t;
    const tl_ptr q = tl_uniform(p);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0" : "=&s"(t) : "s"(q));
    p += 1;
}
// the constant 100 MHz clock: what a stretch of shader cycles is in time (the effective clock)
__device__ __forceinline__ void tl_stamp_real(tl_ptr& p) {
    unsigned long long t;
    const tl_ptr q = tl_uniform(p);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0" : "=&s"(t) : "s"(q));
    p += 1;
}
__device__ __forceinline__ void tl_word(tl_ptr& p, unsigned long long v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const unsigned long long s = ((unsigned long long)hi << 32) | lo;
    const tl_ptr q = tl_uniform(p);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0" : : "s"(s), "s"(q));
    p += 1;
}
__device__ __forceinline__ void tl_end() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)"); }

}  // namespace optex
#define TL_STAMP(p) optex::tl_stamp(p)
#define TL_DEFINE_SETTER(name)                                                                    \
    namespace optex {                                                                             \
    void name(unsigned long long* buf, int words) {                                               \
        (void)hipMemcpyToSymbol(HIP_SYMBOL(tl_buf), &buf, sizeof(buf));                           \
        (void)hipMemcpyToSymbol(HIP_SYMBOL(tl_words), &words, sizeof(words));                     \
    }                                                                                             \
    }
#else
#define TL_DEFINE_SETTER(name)
#define TL_STAMP(p)
#endif
