// micro-benchmark: does the Infinity Cache (MALL) keep the most recently WRITTEN part of a 512 MB tensor, so that a consumer
// reading it in the opposite order of the producer hits it?  (diagnostic, not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void writer(float4* p, size_t n4, int reverse) {
    const size_t nb = gridDim.x;
    for (size_t chunk = blockIdx.x; chunk < n4 / 256; chunk += nb) {
        const size_t c = reverse ? (n4 / 256 - 1 - chunk) : chunk;
        p[c * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
__global__ void reader(const float4* p, size_t n4, int reverse, float* out) {
    const size_t nb = gridDim.x;
    float s = 0.f;
    for (size_t chunk = blockIdx.x; chunk < n4 / 256; chunk += nb) {
        const size_t c = reverse ? (n4 / 256 - 1 - chunk) : chunk;
        const float4 v = p[c * 256 + threadIdx.x];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == -1.f) out[0] = s;
}
// the persistent-grid order: block b takes chunks b, b + nb, ...: at any time the grid works on a contiguous window that moves
// through the tensor (ascending or descending)

int main() {
    for (size_t mb : {128, 256, 512, 1024}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        float4* buf; float* out;
        hipMalloc(&buf, bytes); hipMalloc(&out, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rev = 0; rev < 2; ++rev) {
            float best = 1e9f;
            for (int it = 0; it < 5; ++it) {
                hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n4, 0);
                hipEventRecord(e0);
                hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, 0, (const float4*)buf, n4, rev, out);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%5zu MB written ascending, read %-10s: %7.1f us  %6.2f TB/s\n", mb, rev ? "descending" : "ascending", best * 1e3,
                   bytes / (best * 1e-3) / 1e12);
        }
        hipFree(buf); hipFree(out);
    }
    return 0;
}
