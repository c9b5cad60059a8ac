// micro-benchmark: issue rate of the fp32 MFMA shapes on gfx950 (diagnostic, not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float av, float bv) {
    float a = av + threadIdx.x, b = bv + threadIdx.x;
    if constexpr (MODE >= 3) {                         // NV independent VALU ops after each MFMA; their results feed the MFMA 32 later
        constexpr int NV = MODE - 2;
        f32x4 acc[32];
        float bb[32][NV];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < NV; ++j) bb[i][j] = b + i + j;
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                float bsum = bb[i][0];
#pragma unroll
                for (int j = 1; j < NV; ++j) bsum = bb[i][j];      // (use the last: all chains stay live)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bsum, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) bb[i][j] = bb[i][j] * 1.0001f + a;
            }
        }
        f32x4 s = acc[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) s += acc[i];
        float e = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i)
#pragma unroll
            for (int j = 0; j < NV; ++j) e += bb[i][j];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + e;
    } else if constexpr (MODE == 0 || MODE == 2) {            // 32 accumulators of 16x16x4
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if constexpr (MODE == 2) { b = b * 1.0001f + a; }       // one VALU op per MFMA feeding B
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        }
        f32x4 s = acc[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) s += acc[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    } else {                                            // 8 accumulators of 32x32x2
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[i][j];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int MODE>
void run(const char* name, int threads, double flop_per_iter_per_wave) {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * threads / 64;
    printf("%-44s %4d threads/CU: %8.3f ms  %7.1f TFLOP/s\n", name, threads, ms, flop_per_iter_per_wave * iters * waves / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("16x16x4 f32, 32 independent accumulators", threads, 32 * 2048.0);
        run<2>("16x16x4 f32 + one dependent VALU op each", threads, 32 * 2048.0);
        run<1>("32x32x2 f32, 8 independent accumulators", threads, 8 * 4096.0);
        run<3>("16x16x4 f32 + 1 independent VALU op each", threads, 32 * 2048.0);
        run<4>("16x16x4 f32 + 2 independent VALU ops each", threads, 32 * 2048.0);
        run<5>("16x16x4 f32 + 3 independent VALU ops each", threads, 32 * 2048.0);
    }
    return 0;
}
