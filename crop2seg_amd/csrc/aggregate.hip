// Attention-weighted temporal aggregation of skip feature maps (mode att_group), forward and backward.
// Reference: src/backbones/temporal_aggregator.py:14-45,58-70 -- nn.Upsample(bilinear, align_corners=False)
// of the attention masks, x.chunk(n_head) * attn, sum over T -- which materialises [B,T,C,H,W] products.
// Here the 4-tap bilinear weights are computed once per thread and x is streamed exactly once (HBM-bound).
#include "common.h"

namespace {

struct Taps {
    int i0, i1;
    float w0, w1;
};
// torch area_pixel_compute_source_index (align_corners=False)
__device__ __forceinline__ Taps taps_for(int dst, int in, int out) {
    Taps t;
    if (in == out) { t.i0 = dst; t.i1 = dst; t.w0 = 1.f; t.w1 = 0.f; return t; }
    if (in == 1) { t.i0 = 0; t.i1 = 0; t.w0 = 1.f; t.w1 = 0.f; return t; }        // one weight per plane (mode "mean")
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    t.i0 = (int)src;
    t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
    t.w1 = src - (float)t.i0;
    t.w0 = 1.f - t.w1;
    return t;
}

template <int CPG>
__global__ __launch_bounds__(256) void agg_fwd_kernel(const float* __restrict__ x, const float* __restrict__ attn,
                                                      const int* __restrict__ valid, float* __restrict__ out,
                                                      c2s_agg_desc d) {
    const long total = (long)d.B * d.n_head * d.H * d.W;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int X = (int)(e % d.W);
    long r = e / d.W;
    const int Y = (int)(r % d.H); r /= d.H;
    const int g = (int)(r % d.n_head), b = (int)(r / d.n_head);
    const Taps ty = taps_for(Y, d.h, d.H), tx = taps_for(X, d.w, d.W);
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.h * d.w;
    float acc[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) acc[c] = 0.f;
    for (int t = 0; t < d.T; ++t) {
        if (valid != nullptr && valid[b * d.T + t] == 0) continue;
        const float* ap = attn + ((size_t)(g * d.B + b) * d.T + t) * hw;
        const float a = ty.w0 * (tx.w0 * ap[ty.i0 * d.w + tx.i0] + tx.w1 * ap[ty.i0 * d.w + tx.i1]) +
                        ty.w1 * (tx.w0 * ap[ty.i1 * d.w + tx.i0] + tx.w1 * ap[ty.i1 * d.w + tx.i1]);
        const float* xp = x + (((size_t)b * d.T + t) * d.C + g * CPG) * HW + (size_t)Y * d.W + X;
#pragma unroll
        for (int c = 0; c < CPG; ++c) acc[c] += a * xp[(size_t)c * HW];
    }
    float* op = out + ((size_t)b * d.C + g * CPG) * HW + (size_t)Y * d.W + X;
#pragma unroll
    for (int c = 0; c < CPG; ++c) op[(size_t)c * HW] = acc[c];
}

// gx = up(attn) * gout ;  gup[g,b,t,Y,X] = sum_{c in g} x * gout
template <int CPG>
__global__ __launch_bounds__(256) void agg_bwd_kernel(const float* __restrict__ x, const float* __restrict__ attn,
                                                      const int* __restrict__ valid, const float* __restrict__ gout,
                                                      float* __restrict__ gx, int gx_acc, float* __restrict__ gup,
                                                      c2s_agg_desc d) {
    const long total = (long)d.B * d.n_head * d.H * d.W;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int X = (int)(e % d.W);
    long r = e / d.W;
    const int Y = (int)(r % d.H); r /= d.H;
    const int g = (int)(r % d.n_head), b = (int)(r / d.n_head);
    const Taps ty = taps_for(Y, d.h, d.H), tx = taps_for(X, d.w, d.W);
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.h * d.w;
    float go[CPG];
    const float* gp = gout + ((size_t)b * d.C + g * CPG) * HW + (size_t)Y * d.W + X;
#pragma unroll
    for (int c = 0; c < CPG; ++c) go[c] = gp[(size_t)c * HW];
    for (int t = 0; t < d.T; ++t) {
        const size_t xo = (((size_t)b * d.T + t) * d.C + g * CPG) * HW + (size_t)Y * d.W + X;
        const size_t uo = ((size_t)(g * d.B + b) * d.T + t) * HW + (size_t)Y * d.W + X;
        if (valid != nullptr && valid[b * d.T + t] == 0) {
            if (!gx_acc) {
#pragma unroll
                for (int c = 0; c < CPG; ++c) gx[xo + (size_t)c * HW] = 0.f;
            }
            gup[uo] = 0.f;
            continue;
        }
        const float* ap = attn + ((size_t)(g * d.B + b) * d.T + t) * hw;
        const float a = ty.w0 * (tx.w0 * ap[ty.i0 * d.w + tx.i0] + tx.w1 * ap[ty.i0 * d.w + tx.i1]) +
                        ty.w1 * (tx.w0 * ap[ty.i1 * d.w + tx.i0] + tx.w1 * ap[ty.i1 * d.w + tx.i1]);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            s += x[xo + (size_t)c * HW] * go[c];
            const float v = a * go[c];
            if (gx_acc) gx[xo + (size_t)c * HW] += v; else gx[xo + (size_t)c * HW] = v;
        }
        gup[uo] = s;
    }
}

// adjoint of the bilinear upsample: one thread per low-resolution attention cell (gather, deterministic)
// Adjoint of the bilinear upsampling of the attention masks: gattn[plane][i][j] += sum_{Y,X} wy(Y,i) wx(X,j) gup[plane][Y][X].
// One workgroup per (plane, low-resolution row i), separable: thread X first folds the <= 3*H/h contributing rows
// of column X (coalesced row reads), then thread j folds the <= 3*W/w contributing columns out of LDS.
__global__ __launch_bounds__(256) void agg_upsample_adjoint_kernel(const float* __restrict__ gup, float* __restrict__ gattn,
                                                                   c2s_agg_desc d) {
    extern __shared__ float colsum[];               // [W]
    const long plane = blockIdx.x / d.h;
    const int i = blockIdx.x % d.h;
    const int sy = d.H / d.h, sx = d.W / d.w;
    const int y0 = max(0, (i - 1) * sy), y1 = min(d.H, (i + 2) * sy);
    const float* gp = gup + (size_t)plane * d.H * d.W;
    for (int X = threadIdx.x; X < d.W; X += blockDim.x) {
        float acc = 0.f;
        for (int Y = y0; Y < y1; ++Y) {
            const Taps ty = taps_for(Y, d.h, d.H);
            const float wy = (ty.i0 == i ? ty.w0 : 0.f) + (ty.i1 == i ? ty.w1 : 0.f);
            acc += wy * gp[(size_t)Y * d.W + X];
        }
        colsum[X] = acc;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d.w; j += blockDim.x) {
        const int x0 = max(0, (j - 1) * sx), x1 = min(d.W, (j + 2) * sx);
        float acc = 0.f;
        for (int X = x0; X < x1; ++X) {
            const Taps tx = taps_for(X, d.w, d.W);
            const float wx = (tx.i0 == j ? tx.w0 : 0.f) + (tx.i1 == j ? tx.w1 : 0.f);
            acc += wx * colsum[X];
        }
        gattn[((size_t)plane * d.h + i) * d.w + j] += acc;
    }
}

// ---- agg_mode "att_mean" / "mean" (temporal_aggregator.py:46-56,71-77) reuse the kernels above with a derived weight tensor:
//   att_mean: every channel group gets the head-averaged attention  v[g] = mean_h attn[h]   (g = 0..n_head-1)
//   mean    : v[g][b][t] = valid[b,t] / #valid frames of b          (one weight per frame: a 1x1 "attention map")
__global__ __launch_bounds__(256) void head_mean_kernel(const float* __restrict__ attn, float* __restrict__ v, int n_head, long n) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int h = 0; h < n_head; ++h) s += attn[(size_t)h * n + i];
    s /= (float)n_head;
    for (int h = 0; h < n_head; ++h) v[(size_t)h * n + i] = s;
}
__global__ __launch_bounds__(256) void head_mean_bwd_kernel(const float* __restrict__ gv, float* __restrict__ gattn, int n_head,
                                                            long n, int accumulate) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int h = 0; h < n_head; ++h) s += gv[(size_t)h * n + i];
    s /= (float)n_head;
    for (int h = 0; h < n_head; ++h) gattn[(size_t)h * n + i] = accumulate ? gattn[(size_t)h * n + i] + s : s;
}
__global__ void frame_mean_weights_kernel(const int* __restrict__ valid, float* __restrict__ v, int n_head, int B, int T) {
    const int b = blockIdx.x;
    int cnt = 0;
    for (int t = 0; t < T; ++t) cnt += (valid == nullptr || valid[b * T + t] != 0) ? 1 : 0;
    const float w = 1.f / (float)cnt;
    for (int e = threadIdx.x; e < n_head * T; e += blockDim.x) {
        const int h = e / T, t = e % T;
        v[((size_t)h * B + b) * T + t] = (valid == nullptr || valid[b * T + t] != 0) ? w : 0.f;
    }
}

int check(const c2s_agg_desc* d) {
    C2S_REQUIRE(d && d->B > 0 && d->T > 0 && d->C > 0 && d->H > 0 && d->W > 0, "aggregate: bad shape");
    C2S_REQUIRE(d->n_head > 0 && d->C % d->n_head == 0, "aggregate: C %% n_head != 0");
    C2S_REQUIRE(d->H >= d->h && d->W >= d->w && d->H % d->h == 0 && d->W % d->w == 0,
                "aggregate: feature maps must be an integer multiple of the attention resolution");
    const int cpg = d->C / d->n_head;
    C2S_REQUIRE(cpg == 1 || cpg == 2 || cpg == 4 || cpg == 8 || cpg == 16, "aggregate: C/n_head must be 1,2,4,8 or 16");
    return C2S_OK;
}

}  // namespace

extern "C" int c2s_attn_head_mean(const float* attn, float* v, int n_head, long n, void* stream) {
    C2S_REQUIRE(attn && v && n_head > 0 && n > 0, "attn_head_mean: bad args");
    hipLaunchKernelGGL(head_mean_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, attn, v, n_head, n);
    C2S_CHECK_LAUNCH("attn_head_mean");
    return C2S_OK;
}
extern "C" int c2s_attn_head_mean_bwd(const float* gv, float* gattn, int n_head, long n, int accumulate, void* stream) {
    C2S_REQUIRE(gv && gattn && n_head > 0 && n > 0, "attn_head_mean_bwd: bad args");
    hipLaunchKernelGGL(head_mean_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, gv, gattn, n_head, n, accumulate);
    C2S_CHECK_LAUNCH("attn_head_mean_bwd");
    return C2S_OK;
}
extern "C" int c2s_frame_mean_weights(const int* valid, float* v, int n_head, int B, int T, void* stream) {
    C2S_REQUIRE(v && n_head > 0 && B > 0 && T > 0, "frame_mean_weights: bad args");
    hipLaunchKernelGGL(frame_mean_weights_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, valid, v, n_head, B, T);
    C2S_CHECK_LAUNCH("frame_mean_weights");
    return C2S_OK;
}

extern "C" int c2s_temporal_aggregate_fwd(const c2s_agg_desc* d, const float* x, const float* attn, const int* valid,
                                          float* out, void* stream) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(x && attn && out, "aggregate_fwd: null pointer");
    const long total = (long)d->B * d->n_head * d->H * d->W;
    dim3 grid(cdiv(total, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (d->C / d->n_head) {
        case 1: hipLaunchKernelGGL(agg_fwd_kernel<1>, grid, block, 0, st, x, attn, valid, out, *d); break;
        case 2: hipLaunchKernelGGL(agg_fwd_kernel<2>, grid, block, 0, st, x, attn, valid, out, *d); break;
        case 4: hipLaunchKernelGGL(agg_fwd_kernel<4>, grid, block, 0, st, x, attn, valid, out, *d); break;
        case 8: hipLaunchKernelGGL(agg_fwd_kernel<8>, grid, block, 0, st, x, attn, valid, out, *d); break;
        default: hipLaunchKernelGGL(agg_fwd_kernel<16>, grid, block, 0, st, x, attn, valid, out, *d); break;
    }
    C2S_CHECK_LAUNCH("aggregate_fwd");
    return C2S_OK;
}

extern "C" size_t c2s_temporal_aggregate_bwd_workspace_floats(const c2s_agg_desc* d) {
    if (!d) return 0;
    return (size_t)d->n_head * d->B * d->T * d->H * d->W;
}

extern "C" int c2s_temporal_aggregate_bwd(const c2s_agg_desc* d, const float* x, const float* attn, const int* valid,
                                          const float* gout, float* gx, int gx_accumulate, float* gattn,
                                          float* workspace, size_t ws_floats, void* stream) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(x && attn && gout && gx && gattn && workspace, "aggregate_bwd: null pointer");
    C2S_REQUIRE(ws_floats >= c2s_temporal_aggregate_bwd_workspace_floats(d), "aggregate_bwd: workspace too small");
    const long total = (long)d->B * d->n_head * d->H * d->W;
    dim3 grid(cdiv(total, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (d->C / d->n_head) {
        case 1: hipLaunchKernelGGL(agg_bwd_kernel<1>, grid, block, 0, st, x, attn, valid, gout, gx, gx_accumulate, workspace, *d); break;
        case 2: hipLaunchKernelGGL(agg_bwd_kernel<2>, grid, block, 0, st, x, attn, valid, gout, gx, gx_accumulate, workspace, *d); break;
        case 4: hipLaunchKernelGGL(agg_bwd_kernel<4>, grid, block, 0, st, x, attn, valid, gout, gx, gx_accumulate, workspace, *d); break;
        case 8: hipLaunchKernelGGL(agg_bwd_kernel<8>, grid, block, 0, st, x, attn, valid, gout, gx, gx_accumulate, workspace, *d); break;
        default: hipLaunchKernelGGL(agg_bwd_kernel<16>, grid, block, 0, st, x, attn, valid, gout, gx, gx_accumulate, workspace, *d); break;
    }
    C2S_CHECK_LAUNCH("aggregate_bwd");
    const long rows = (long)d->n_head * d->B * d->T * d->h;
    const int threads = d->W >= 256 ? 256 : (d->W > 64 ? 128 : 64);
    hipLaunchKernelGGL(agg_upsample_adjoint_kernel, dim3(rows), dim3(threads), (size_t)d->W * sizeof(float), st, workspace,
                       gattn, *d);
    C2S_CHECK_LAUNCH("aggregate_upsample_adjoint");
    return C2S_OK;
}
