// Squeeze-and-excitation (constructor flag add_squeeze_excit, and inside every MBConv of use_mbconv).
//
// Reference: src/backbones/squeeze_and_excitation.py:7-30
//     y = x * sigmoid(W2 relu(W1 mean_hw(x)))       W1 [C/16, C], W2 [C, C/16], no biases, per frame
// appended to ConvLayer after its last conv-norm-ReLU (conv.py:90-91), applied after the residual sum of DownConvBlock
// (conv.py:286-294) and between the depthwise and the projection convolution of MBConv (mbconv.py:80-82).
//
// HBM-bound byte work in the row layout of norm.hip (row = one (frame, channel) plane, wave = (row, 2048-float segment)):
//   forward   pool: one read of x -> per-(row, segment) sums;  gate: one workgroup per frame (a few hundred MACs);
//             scale: x read again, y written (rows of padded frames are filled with pad_value)
//   backward  sums: ds[row] = sum g x (one read of g and x);  gate adjoint per frame -> d pooled, per-frame d W1 / d W2;
//             apply: dx = g s + d pooled / HW (g read again);  parameter gradients = fixed-order sum over the valid frames
// Fixed summation orders (bitwise reproducible), double accumulation in the per-frame stage.
#include "common.h"

namespace {

constexpr int SEG = 2048;
constexpr int MAXC = 1024, MAXR = 64;

__host__ __device__ inline int seg_len(int HW) { return HW < SEG ? HW : SEG; }
__host__ __device__ inline int n_segs(int HW) { return (HW + seg_len(HW) - 1) / seg_len(HW); }

// part[item] = sum over the segment of x (G == nullptr) or of g * x
__global__ __launch_bounds__(256) void se_row_sums_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          float* __restrict__ part, const int* __restrict__ valid, int C,
                                                          int HW, int segs, long nitems) {
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63;
    const long row = item / segs;
    const int seg = (int)(item % segs);
    if (valid != nullptr && valid[row / C] == 0) {
        if (lane == 0) part[item] = 0.f;
        return;
    }
    const int L = seg_len(HW);
    const int beg = seg * L;
    const int len = (HW - beg) < L ? (HW - beg) : L;
    const size_t base = (size_t)row * HW + beg;
    float s = 0.f;
    if ((len & 3) == 0) {
        for (int i = lane * 4; i < len; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + base + i);
            if (g != nullptr) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(g + base + i);
                s += (v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w);
            } else {
                s += (v.x + v.y) + (v.z + v.w);
            }
        }
    } else {
        for (int i = lane; i < len; i += 64) s += g != nullptr ? x[base + i] * g[base + i] : x[base + i];
    }
    s = wave_sum(s);
    if (lane == 0) part[item] = s;
}

// one workgroup per frame: pooled [N,C], hidden [N,R] (post-ReLU), scale [N,C]
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ part, const float* __restrict__ W1,
                                                      const float* __restrict__ W2, float* __restrict__ pooled,
                                                      float* __restrict__ hidden, float* __restrict__ scale,
                                                      const int* __restrict__ valid, int C, int R, int HW, int segs) {
    __shared__ float sp[MAXC];
    __shared__ float sh[MAXR];
    const int n = blockIdx.x, tid = threadIdx.x;
    if (valid != nullptr && valid[n] == 0) return;
    for (int c = tid; c < C; c += 256) {
        double s = 0.0;
        for (int k = 0; k < segs; ++k) s += part[((size_t)n * C + c) * segs + k];
        const float m = (float)(s / (double)HW);
        sp[c] = m;
        pooled[(size_t)n * C + c] = m;
    }
    __syncthreads();
    if (tid < R) {
        float v = 0.f;
        for (int c = 0; c < C; ++c) v = fmaf(W1[(size_t)tid * C + c], sp[c], v);
        v = fmaxf(v, 0.f);
        sh[tid] = v;
        hidden[(size_t)n * R + tid] = v;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float v = 0.f;
        for (int k = 0; k < R; ++k) v = fmaf(W2[(size_t)c * R + k], sh[k], v);
        scale[(size_t)n * C + c] = 1.f / (1.f + expf(-v));
    }
}

// forward: y = x * scale[row];  backward: dx = g * scale[row] + dpool[row] / HW
__global__ __launch_bounds__(256) void se_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ dpool, float* __restrict__ y,
                                                       const int* __restrict__ valid, int C, int HW, int segs, long nitems,
                                                       float fill) {
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63;
    const long row = item / segs;
    const int seg = (int)(item % segs);
    const int L = seg_len(HW);
    const int beg = seg * L;
    const int len = (HW - beg) < L ? (HW - beg) : L;
    const size_t base = (size_t)row * HW + beg;
    if (valid != nullptr && valid[row / C] == 0) {
        for (int i = lane; i < len; i += 64) y[base + i] = fill;
        return;
    }
    const float s = scale[row];
    const float a = dpool != nullptr ? dpool[row] / (float)HW : 0.f;
    if ((len & 3) == 0) {
        for (int i = lane * 4; i < len; i += 256) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + base + i);
            v.x = fmaf(v.x, s, a); v.y = fmaf(v.y, s, a); v.z = fmaf(v.z, s, a); v.w = fmaf(v.w, s, a);
            *reinterpret_cast<f32x4*>(y + base + i) = v;
        }
    } else {
        for (int i = lane; i < len; i += 64) y[base + i] = fmaf(x[base + i], s, a);
    }
}

// gate adjoint of one frame: ds from the (g x) sums; writes dpool [N,C] and the frame's parameter-gradient contributions
// pw [N][R*C + C*R] (d W1 | d W2)
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float* __restrict__ part, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, const float* __restrict__ pooled,
                                                          const float* __restrict__ hidden, const float* __restrict__ scale,
                                                          float* __restrict__ dpool, float* __restrict__ pw,
                                                          const int* __restrict__ valid, int C, int R, int segs) {
    __shared__ float dz2[MAXC];
    __shared__ float dz1[MAXR];
    const int n = blockIdx.x, tid = threadIdx.x;
    if (valid != nullptr && valid[n] == 0) return;
    float* pw1 = pw + (size_t)n * 2 * R * C;
    float* pw2 = pw1 + (size_t)R * C;
    for (int c = tid; c < C; c += 256) {
        double s = 0.0;
        for (int k = 0; k < segs; ++k) s += part[((size_t)n * C + c) * segs + k];
        const float sc = scale[(size_t)n * C + c];
        dz2[c] = (float)s * sc * (1.f - sc);
    }
    __syncthreads();
    for (int e = tid; e < C * R; e += 256) pw2[e] = dz2[e / R] * hidden[(size_t)n * R + e % R];      // d W2 [C][R]
    if (tid < R) {
        float v = 0.f;
        for (int c = 0; c < C; ++c) v = fmaf(W2[(size_t)c * R + tid], dz2[c], v);
        dz1[tid] = hidden[(size_t)n * R + tid] > 0.f ? v : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < R * C; e += 256) pw1[e] = dz1[e / C] * pooled[(size_t)n * C + e % C];      // d W1 [R][C]
    for (int c = tid; c < C; c += 256) {
        float v = 0.f;
        for (int k = 0; k < R; ++k) v = fmaf(W1[(size_t)k * C + c], dz1[k], v);
        dpool[(size_t)n * C + c] = v;
    }
}

__global__ __launch_bounds__(256) void se_param_reduce_kernel(const float* __restrict__ pw, float* __restrict__ gW1,
                                                              float* __restrict__ gW2, const int* __restrict__ valid, int N,
                                                              int RC, int acc1, int acc2) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 2 * RC) return;
    double s = 0.0;
    for (int n = 0; n < N; ++n)
        if (valid == nullptr || valid[n] != 0) s += pw[(size_t)n * 2 * RC + e];
    if (e < RC) gW1[e] = acc1 ? gW1[e] + (float)s : (float)s;
    else gW2[e - RC] = acc2 ? gW2[e - RC] + (float)s : (float)s;
}

// x[n,c,:] += bias[c] on the frames that are real (the depthwise convolution of MBConv has a bias, mbconv.py:71-79; the
// depthwise kernels of misc.hip serve DepthwiseSeparableConv2D, which has none)
__global__ __launch_bounds__(256) void channel_bias_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                           const int* __restrict__ valid, int C, int HW, long total) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / HW;
        if (valid != nullptr && valid[row / C] == 0) continue;
        x[e] += bias[row % C];
    }
}

}  // namespace

extern "C" int c2s_channel_bias_add(float* x, const float* bias, const int* valid, int N, int C, int HW, void* stream) {
    C2S_REQUIRE(x && bias && N > 0 && C > 0 && HW > 0, "channel_bias_add: bad args");
    const long total = (long)N * C * HW;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(channel_bias_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream, x,
                       bias, valid, C, HW, total);
    C2S_CHECK_LAUNCH("channel_bias_add");
    return C2S_OK;
}

// workspace: part [rows*segs] | pw [N * 2*R*C] | dpool [N*C]
extern "C" size_t c2s_se_workspace_floats(int N, int C, int HW) {
    const int R = C / 16;
    return (size_t)N * C * n_segs(HW) + (size_t)N * 2 * R * C + (size_t)N * C;
}

extern "C" int c2s_se_fwd(const float* x, const float* W1, const float* W2, float* pooled, float* hidden, float* scale,
                          float* y, const int* valid, int N, int C, int HW, float pad_value, float* workspace,
                          size_t ws_floats, void* stream) {
    C2S_REQUIRE(x && W1 && W2 && pooled && hidden && scale && y && workspace, "se_fwd: null pointer");
    C2S_REQUIRE(N > 0 && HW > 0 && C >= 16 && C <= MAXC && C / 16 <= MAXR, "se_fwd: 16 <= C <= 1024");
    C2S_REQUIRE(ws_floats >= c2s_se_workspace_floats(N, C, HW), "se_fwd: workspace too small");
    const int R = C / 16, segs = n_segs(HW);
    const long nitems = (long)N * C * segs;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(se_row_sums_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, (const float*)nullptr, workspace, valid, C,
                       HW, segs, nitems);
    C2S_CHECK_LAUNCH("se_pool");
    hipLaunchKernelGGL(se_gate_kernel, dim3(N), dim3(256), 0, st, workspace, W1, W2, pooled, hidden, scale, valid, C, R, HW, segs);
    C2S_CHECK_LAUNCH("se_gate");
    hipLaunchKernelGGL(se_apply_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, scale, (const float*)nullptr, y, valid, C, HW,
                       segs, nitems, pad_value);
    C2S_CHECK_LAUNCH("se_scale");
    return C2S_OK;
}

extern "C" int c2s_se_bwd(const float* x, const float* g, const float* W1, const float* W2, const float* pooled,
                          const float* hidden, const float* scale, float* gx, float* gW1, float* gW2, int acc_w1, int acc_w2,
                          const int* valid, int N, int C, int HW, float* workspace, size_t ws_floats, void* stream) {
    C2S_REQUIRE(x && g && W1 && W2 && pooled && hidden && scale && gx && gW1 && gW2 && workspace, "se_bwd: null pointer");
    C2S_REQUIRE(N > 0 && HW > 0 && C >= 16 && C <= MAXC && C / 16 <= MAXR, "se_bwd: 16 <= C <= 1024");
    C2S_REQUIRE(ws_floats >= c2s_se_workspace_floats(N, C, HW), "se_bwd: workspace too small");
    const int R = C / 16, segs = n_segs(HW);
    const long nitems = (long)N * C * segs;
    float* part = workspace;
    float* pw = part + (size_t)N * C * segs;
    float* dpool = pw + (size_t)N * 2 * R * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(se_row_sums_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, g, part, valid, C, HW, segs, nitems);
    C2S_CHECK_LAUNCH("se_bwd_sums");
    hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(N), dim3(256), 0, st, part, W1, W2, pooled, hidden, scale, dpool, pw, valid, C, R,
                       segs);
    C2S_CHECK_LAUNCH("se_gate_bwd");
    hipLaunchKernelGGL(se_apply_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, g, scale, dpool, gx, valid, C, HW, segs, nitems,
                       0.f);
    C2S_CHECK_LAUNCH("se_bwd_apply");
    hipLaunchKernelGGL(se_param_reduce_kernel, dim3(cdiv(2L * R * C, 256)), dim3(256), 0, st, pw, gW1, gW2, valid, N, R * C, acc_w1,
                       acc_w2);
    C2S_CHECK_LAUNCH("se_param_reduce");
    return C2S_OK;
}
