// Shared device/host helpers for libc2s_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/c2s_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void c2s_set_error(const char* fmt, ...);

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

// counter-based RNG (splitmix64 finaliser of key ^ index): uniform in [0,1)
__device__ __forceinline__ float c2s_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
