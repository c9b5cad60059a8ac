// Shared device/host helpers for libc2s_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/c2s_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void c2s_set_error(const char* fmt, ...);

// Per-device one-time set-up (misc.hip).  Every source file registers a hook that raises the dynamic-LDS limit of its
// kernels; c2s_init(device) / c2s_ensure_init() run all hooks once per device under a mutex and cache the device's CU
// count.  Entry points call c2s_ensure_init() (one thread-local hipGetDevice + an atomic load when already done), so a
// caller that ran c2s_init() before capturing a hipGraph never triggers set-up work inside the capture.
typedef void (*c2s_init_hook)(void);
struct C2sInitRegistrar {
    explicit C2sInitRegistrar(c2s_init_hook hook);
};
void c2s_ensure_init();
int c2s_cus();   // compute units of the current device (256 when no device is present: CPU-side argument checks)
#define C2S_RAISE_LDS(kernel) \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)

#define C2S_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            c2s_set_error(__VA_ARGS__);   \
            return C2S_EINVAL;            \
        }                                 \
    } while (0)

#define C2S_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            c2s_set_error("%s: %s", name, hipGetErrorString(e__));          \
            return C2S_ELAUNCH;                                             \
        }                                                                   \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int reflect_idx(int i, int n) {
    // single reflection (pad < n): -1 -> 1, n -> n-2
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// counter-based RNG: two rounds of a 32-bit avalanche hash (multiply / xor-shift) over (index, key): uniform in [0,1).
// 14 integer VALU ops per draw (a 64-bit splitmix finaliser costs ~30 on this ISA: no 64-bit multiplier) -- the L-TAE
// kernels draw one number per attention element, 976 per pixel at T = 61.
__device__ __forceinline__ uint32_t c2s_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float c2s_uniform(uint64_t seed, uint64_t idx) {
    uint32_t h = c2s_hash32((uint32_t)idx ^ (uint32_t)seed);
    h = c2s_hash32(h ^ (uint32_t)(idx >> 32) * 0x9E3779B9u ^ (uint32_t)(seed >> 32));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
