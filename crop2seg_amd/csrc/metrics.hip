// Metrics tail of the training / validation loop and the boundary-loss pieces (N4).
//
// Reference call sites (paths under the reference root):
//   src/learning/utils.py:332-336   pred = out.argmax(1); pred_ = out.topk(2, 1).indices
//   src/learning/utils.py:377-380   pred_top2 = where(y == pred_[:,1], pred_[:,1], pred_[:,0]); iou_meter.add(pred, y);
//                                   iou_meter_top2.add(pred_top2, y); loss_meter.add(loss.item())
//   src/learning/miou.py:55-117     ConfusionMatrix.add: bincount(pred + K * target) accumulated into a K x K matrix
//   src/learning/utils.py:198-222   get_dilated: one-hot -> 3x3 cross depthwise conv (zero padding) -> bool
//   src/learning/utils.py:283-285   y_b = where(dilated.sum(1) > 1, 1, 0)
//   src/learning/focal_loss.py:7-44 FocalCELoss(gamma): mean over kept pixels of -(1 - pt)^gamma log pt
//
// All of this is integer / index work on [B,K,H,W] logits read ONCE: HBM-bound at K*4 + 8 bytes per pixel (3.9 MB at
// B = 4, 128x128: microseconds), so the design goal is "no extra passes, no host synchronisation": confusion matrices and
// the loss sum accumulate in device memory; the host reads them when it displays (every display_step iterations).
#include "common.h"

namespace {

constexpr int MAXK = 32;    // classes (the reference uses 15; PASTIS 20)

// One thread per pixel.  argmax: first maximum (torch.argmax).  second: the largest of the rest, lowest index on ties --
// torch.topk leaves the order of tied values unspecified (its CPU introselect and CUDA radix-select disagree with each
// other); wherever the three largest logits are distinct this is exactly out.topk(2).indices.
__global__ __launch_bounds__(256) void metrics_update_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                             unsigned long long* __restrict__ conf,
                                                             unsigned long long* __restrict__ conf_top2,
                                                             int64_t* __restrict__ pred_out, int64_t* __restrict__ pred2_out,
                                                             int B, int K, int HW) {
    __shared__ unsigned int h1[MAXK * MAXK];
    __shared__ unsigned int h2[MAXK * MAXK];
    const int KK = K * K;
    for (int i = threadIdx.x; i < KK; i += 256) { h1[i] = 0; h2[i] = 0; }
    __syncthreads();
    const long total = (long)B * HW;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int pix = (int)(e % HW), b = (int)(e / HW);
        const float* lp = logits + (size_t)b * K * HW + pix;
        float v1 = lp[0], v2 = -INFINITY;
        int i1 = 0, i2 = -1;
        for (int k = 1; k < K; ++k) {
            const float v = lp[(size_t)k * HW];
            if (v > v1 || (v != v && v1 == v1)) {           // NaN counts as the maximum (torch semantics)
                v2 = v1; i2 = i1; v1 = v; i1 = k;
            } else if (i2 < 0 || v > v2 || (v != v && v2 == v2)) {
                v2 = v; i2 = k;
            }
        }
        if (i2 < 0) i2 = 0;                                   // K == 1
        const long long t = target[e];
        const int top2 = (t == i2) ? i2 : i1;                 // utils.py:377
        if (pred_out) pred_out[e] = i1;
        if (pred2_out) pred2_out[e] = top2;
        if (t >= 0 && t < K) {
            atomicAdd(&h1[(int)t * K + i1], 1u);
            atomicAdd(&h2[(int)t * K + top2], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KK; i += 256) {
        if (h1[i]) atomicAdd(conf + i, (unsigned long long)h1[i]);
        if (conf_top2 && h2[i]) atomicAdd(conf_top2 + i, (unsigned long long)h2[i]);
    }
}

// ConfusionMatrix.add for class-index predictions (miou.py:98-114): conf[t*K + p] += 1
__global__ __launch_bounds__(256) void confusion_add_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ target,
                                                            unsigned long long* __restrict__ conf, long n, int K) {
    __shared__ unsigned int h1[MAXK * MAXK];
    const int KK = K * K;
    for (int i = threadIdx.x; i < KK; i += 256) h1[i] = 0;
    __syncthreads();
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long long t = target[e], p = pred[e];
        if (t >= 0 && t < K && p >= 0 && p < K) atomicAdd(&h1[(int)t * K + (int)p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KK; i += 256)
        if (h1[i]) atomicAdd(conf + i, (unsigned long long)h1[i]);
}

// loss meter: acc[0] += loss, acc[1] += 1 (double) -- tnt AverageValueMeter.add(loss.item()) without the .item() sync
__global__ void loss_meter_kernel(const float* __restrict__ loss, double* __restrict__ acc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { acc[0] += (double)*loss; acc[1] += 1.0; }
}

// y_b[p] = 1 when the 4-neighbourhood (+ centre, zero padding: outside pixels contribute nothing) of p holds more than
// one class  ==  get_dilated(y, K, 4).sum(1) > 1 without the one-hot tensor and the depthwise convolution.
__global__ __launch_bounds__(256) void boundary_target_kernel(const int64_t* __restrict__ y, int64_t* __restrict__ yb, int B,
                                                              int H, int W) {
    const long total = (long)B * H * W;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % W), r = (int)((e / W) % H);
        const int64_t c = y[e];
        int diff = 0;
        if (r > 0) diff |= y[e - W] != c;
        if (r < H - 1) diff |= y[e + W] != c;
        if (x > 0) diff |= y[e - 1] != c;
        if (x < W - 1) diff |= y[e + 1] != c;
        yb[e] = diff;
    }
}

// test_region of iterate() (utils.py:362-373): 'boundary' keeps the boundary pixels (interior -> ignore label), 'interior'
// keeps the interior (boundary -> ignore label); boundary = get_dilated(y, K, 4).sum(1) > 1, as above.
__global__ __launch_bounds__(256) void region_relabel_kernel(const int64_t* __restrict__ y, int64_t* __restrict__ out, int B,
                                                             int H, int W, int keep_boundary, long long ignore_label) {
    const long total = (long)B * H * W;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % W), r = (int)((e / W) % H);
        const int64_t c = y[e];
        int diff = 0;
        if (r > 0) diff |= y[e - W] != c;
        if (r < H - 1) diff |= y[e + W] != c;
        if (x > 0) diff |= y[e - 1] != c;
        if (x < W - 1) diff |= y[e + 1] != c;
        out[e] = (diff != 0) == (keep_boundary != 0) ? c : ignore_label;
    }
}

// FocalCELoss (focal_loss.py:12-45): part[block] = (sum over kept pixels of f = -(1-pt)^g log pt, #kept, sum of w[target]).
// With class weights the reference multiplies a [N,1] column of weights by the [N] row of focal terms (focal_loss.py:36-39:
// `w.gather(1, target)` keeps the column shape) -- an N x N outer product, so its value is
//   size_average: mean_i(w_i) * mean_j(f_j),   else: sum_i(w_i) * sum_j(f_j)
// and that is what is computed here (without the N^2 tensor, which is 17 GB at B=4, 128x128); pinned by fixtures written from
// the imported module (oracle/make_golden_tail.py).  d/dz: the weights do not depend on the logits -- a constant factor.
__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        const float* __restrict__ class_w, float* __restrict__ part, int B, int K,
                                                        int HW, float gamma, long long ignore_index) {
    __shared__ float red[4][3];
    const long total = (long)B * HW;
    float num = 0.f, cnt = 0.f, wsum = 0.f;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long long t = target[e];
        if (t == ignore_index || t < 0 || t >= K) continue;
        const int pix = (int)(e % HW), b = (int)(e / HW);
        const float* lp = logits + (size_t)b * K * HW + pix;
        float mx = lp[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[(size_t)k * HW]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(lp[(size_t)k * HW] - mx);
        const float logpt = lp[(size_t)t * HW] - mx - logf(s);
        const float pt = expf(logpt);
        num += -powf(1.f - pt, gamma) * logpt;
        cnt += 1.f;
        wsum += class_w ? class_w[t] : 1.f;
    }
    num = wave_sum(num); cnt = wave_sum(cnt); wsum = wave_sum(wsum);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = num; red[threadIdx.x >> 6][1] = cnt; red[threadIdx.x >> 6][2] = wsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) part[blockIdx.x * 3 + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    }
}

// tot = (sum f, #kept, d loss / d (sum f))
__global__ void focal_finalize_kernel(const float* __restrict__ part, float* __restrict__ tot, float* __restrict__ loss,
                                      int blocks, int weighted, int size_average, int accumulate) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double num = 0, cnt = 0, wsum = 0;
    for (int i = 0; i < blocks; ++i) { num += part[i * 3]; cnt += part[i * 3 + 1]; wsum += part[i * 3 + 2]; }
    const double scale = weighted ? (size_average ? wsum / (cnt * cnt) : wsum) : (size_average ? 1.0 / cnt : 1.0);
    tot[0] = (float)num; tot[1] = (float)cnt; tot[2] = (float)scale;
    const float l = (float)(num * scale);
    *loss = accumulate ? *loss + l : l;
}

// d/dz_k of -(1-pt)^g log pt = [g (1-pt)^(g-1) pt log pt - (1-pt)^g] (delta_kt - p_k)
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        const float* __restrict__ tot, float* __restrict__ glogits, int B,
                                                        int K, int HW, float gamma, long long ignore_index) {
    const long total = (long)B * HW;
    const float inv_n = tot[2];
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int pix = (int)(e % HW), b = (int)(e / HW);
        const float* lp = logits + (size_t)b * K * HW + pix;
        float* gp = glogits + (size_t)b * K * HW + pix;
        const long long t = target[e];
        if (t == ignore_index || t < 0 || t >= K) {
            for (int k = 0; k < K; ++k) gp[(size_t)k * HW] = 0.f;
            continue;
        }
        float mx = lp[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[(size_t)k * HW]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(lp[(size_t)k * HW] - mx);
        const float inv_s = 1.f / s;
        const float logpt = lp[(size_t)t * HW] - mx - logf(s);
        const float pt = expf(logpt);
        const float om = 1.f - pt;
        // dL/dlogpt = g (1-pt)^(g-1) pt logpt - (1-pt)^g ; dlogpt/dz_k = delta_kt - p_k
        const float coef = (gamma * powf(om, gamma - 1.f) * pt * logpt - powf(om, gamma)) * inv_n;
        for (int k = 0; k < K; ++k) {
            const float pk = expf(lp[(size_t)k * HW] - mx) * inv_s;
            gp[(size_t)k * HW] = coef * ((k == t ? 1.f : 0.f) - pk);
        }
    }
}

// SmoothCrossEntropy2D (smooth_loss.py:58-84): soft targets from the 4-neighbourhood dilation of the label map -- the classes
// present at a pixel or its 4 neighbours (zero padding) share 1 - eps*(K - n) evenly, every other class gets eps = ls/K;
// pixels labelled `bg_index` take the fixed distribution `bg` instead (background_treatment) -- then CrossEntropyLoss with
// probability targets: mean over ALL B*H*W pixels of -sum_k w_k t_k log_softmax(z)_k.  One pass: the class set of a pixel is
// a bit mask built from five labels (no one-hot tensor, no depthwise convolution), the loss term and d/dz of the pixel come
// from the same softmax.  part[block] = (sum of pixel losses, labels outside [0,K) seen -- the reference's one_hot raises
// on those; here they are counted and the pixel contributes nothing).
__global__ __launch_bounds__(256) void smooth_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        const float* __restrict__ class_w, const float* __restrict__ bg,
                                                        float* __restrict__ part, float* __restrict__ glogits,
                                                        float* __restrict__ pixel_loss, int B, int K, int H, int W, float ls,
                                                        long long bg_index, float inv_n) {
    __shared__ float red[4][2];
    const int HW = H * W;
    const long total = (long)B * HW;
    const float eps = ls / (float)K;
    float num = 0.f, bad = 0.f;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int pix = (int)(e % HW), b = (int)(e / HW);
        const int x = pix % W, r = pix / W;
        const float* lp = logits + (size_t)b * K * HW + pix;
        float* gp = glogits ? glogits + (size_t)b * K * HW + pix : nullptr;
        const long long t = target[e];
        if (t < 0 || t >= K) {
            bad += 1.f;
            if (gp) for (int k = 0; k < K; ++k) gp[(size_t)k * HW] = 0.f;
            if (pixel_loss) pixel_loss[e] = 0.f;
            continue;
        }
        unsigned mask = 1u << (int)t;
        if (r > 0) { const long long v = target[e - W]; if (v >= 0 && v < K) mask |= 1u << (int)v; }
        if (r < H - 1) { const long long v = target[e + W]; if (v >= 0 && v < K) mask |= 1u << (int)v; }
        if (x > 0) { const long long v = target[e - 1]; if (v >= 0 && v < K) mask |= 1u << (int)v; }
        if (x < W - 1) { const long long v = target[e + 1]; if (v >= 0 && v < K) mask |= 1u << (int)v; }
        const float nd = (float)__popc(mask);
        const float exp_small = eps * ((float)K - nd);
        const float exp_large = (1.f - exp_small) / nd;
        const bool is_bg = bg != nullptr && t == bg_index;
        float mx = lp[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[(size_t)k * HW]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(lp[(size_t)k * HW] - mx);
        const float lse = mx + logf(s);
        const float inv_s = 1.f / s;
        float loss = 0.f, wsum = 0.f;
        for (int k = 0; k < K; ++k) {
            const float tk = is_bg ? bg[k] : (((mask >> k) & 1u) ? exp_large : eps);
            const float wt = (class_w ? class_w[k] : 1.f) * tk;
            loss -= wt * (lp[(size_t)k * HW] - lse);
            wsum += wt;
        }
        num += loss;
        if (pixel_loss) pixel_loss[e] = loss;
        if (gp) {
            for (int k = 0; k < K; ++k) {
                const float tk = is_bg ? bg[k] : (((mask >> k) & 1u) ? exp_large : eps);
                const float wt = (class_w ? class_w[k] : 1.f) * tk;
                const float pk = expf(lp[(size_t)k * HW] - mx) * inv_s;
                gp[(size_t)k * HW] = (pk * wsum - wt) * inv_n;
            }
        }
    }
    num = wave_sum(num); bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = num; red[threadIdx.x >> 6][1] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        part[blockIdx.x * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

__global__ void smooth_ce_finalize_kernel(const float* __restrict__ part, float* __restrict__ tot, float* __restrict__ loss,
                                          int blocks, double npix, int accumulate) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double num = 0, bad = 0;
    for (int i = 0; i < blocks; ++i) { num += part[i * 2]; bad += part[i * 2 + 1]; }
    tot[0] = (float)num; tot[1] = (float)bad;
    const float l = (float)(num / npix);
    *loss = accumulate ? *loss + l : l;
}

inline int grid_for(long n, int cap = 2048) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
constexpr int FOCAL_BLOCKS = 512;

}  // namespace

extern "C" int c2s_metrics_update(const float* logits, const long long* target, long long* conf, long long* conf_top2,
                                  long long* pred, long long* pred_top2, int B, int K, int HW, void* stream) {
    C2S_REQUIRE(logits && target && conf, "metrics_update: null pointer");
    C2S_REQUIRE(B > 0 && HW > 0 && K >= 1 && K <= MAXK, "metrics_update: 1 <= K <= 32 classes");
    hipLaunchKernelGGL(metrics_update_kernel, dim3(grid_for((long)B * HW, 1024)), dim3(256), 0, (hipStream_t)stream, logits,
                       (const int64_t*)target, (unsigned long long*)conf, (unsigned long long*)conf_top2, (int64_t*)pred,
                       (int64_t*)pred_top2, B, K, HW);
    C2S_CHECK_LAUNCH("metrics_update");
    return C2S_OK;
}

extern "C" int c2s_confusion_add(const long long* pred, const long long* target, long long* conf, long n, int K, void* stream) {
    C2S_REQUIRE(pred && target && conf && n > 0 && K >= 1 && K <= MAXK, "confusion_add: bad args (1 <= K <= 32)");
    hipLaunchKernelGGL(confusion_add_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, (const int64_t*)pred,
                       (const int64_t*)target, (unsigned long long*)conf, n, K);
    C2S_CHECK_LAUNCH("confusion_add");
    return C2S_OK;
}

extern "C" int c2s_loss_meter_add(const float* loss, double* acc, void* stream) {
    C2S_REQUIRE(loss && acc, "loss_meter_add: null pointer");
    hipLaunchKernelGGL(loss_meter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, acc);
    C2S_CHECK_LAUNCH("loss_meter_add");
    return C2S_OK;
}

extern "C" int c2s_boundary_target(const long long* y, long long* y_b, int B, int H, int W, void* stream) {
    C2S_REQUIRE(y && y_b && B > 0 && H > 0 && W > 0, "boundary_target: bad args");
    hipLaunchKernelGGL(boundary_target_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, (hipStream_t)stream,
                       (const int64_t*)y, (int64_t*)y_b, B, H, W);
    C2S_CHECK_LAUNCH("boundary_target");
    return C2S_OK;
}

extern "C" int c2s_region_relabel(const long long* y, long long* y_out, int B, int H, int W, int keep_boundary,
                                  long long ignore_label, void* stream) {
    C2S_REQUIRE(y && y_out && y != y_out && B > 0 && H > 0 && W > 0, "region_relabel: bad args (out of place only)");
    hipLaunchKernelGGL(region_relabel_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, (hipStream_t)stream,
                       (const int64_t*)y, (int64_t*)y_out, B, H, W, keep_boundary, ignore_label);
    C2S_CHECK_LAUNCH("region_relabel");
    return C2S_OK;
}

extern "C" size_t c2s_focal_ce_workspace_floats(void) { return 3 * FOCAL_BLOCKS + 4; }

extern "C" int c2s_focal_ce_ex(const float* logits, const long long* target, const float* class_w, float* loss, float* glogits,
                               int B, int K, int HW, float gamma, long long ignore_index, int size_average, int accumulate_loss,
                               float* workspace, size_t ws_floats, void* stream) {
    C2S_REQUIRE(logits && target && loss && workspace, "focal_ce: null pointer");
    C2S_REQUIRE(B > 0 && K > 0 && HW > 0 && gamma >= 0.f, "focal_ce: bad args");
    C2S_REQUIRE(ws_floats >= c2s_focal_ce_workspace_floats(), "focal_ce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = grid_for((long)B * HW, FOCAL_BLOCKS);
    float* tot = workspace + 3 * FOCAL_BLOCKS;
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(blocks), dim3(256), 0, st, logits, (const int64_t*)target, class_w, workspace, B, K,
                       HW, gamma, ignore_index);
    C2S_CHECK_LAUNCH("focal_fwd");
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, tot, loss, blocks, class_w != nullptr ? 1 : 0,
                       size_average, accumulate_loss);
    C2S_CHECK_LAUNCH("focal_finalize");
    if (glogits) {
        hipLaunchKernelGGL(focal_bwd_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, st, logits, (const int64_t*)target, tot,
                           glogits, B, K, HW, gamma, ignore_index);
        C2S_CHECK_LAUNCH("focal_bwd");
    }
    return C2S_OK;
}

extern "C" int c2s_focal_ce(const float* logits, const long long* target, float* loss, float* glogits, int B, int K, int HW,
                            float gamma, long long ignore_index, int accumulate_loss, float* workspace, size_t ws_floats,
                            void* stream) {
    return c2s_focal_ce_ex(logits, target, nullptr, loss, glogits, B, K, HW, gamma, ignore_index, 1, accumulate_loss, workspace,
                           ws_floats, stream);
}

extern "C" size_t c2s_smooth_ce_workspace_floats(void) { return 2 * FOCAL_BLOCKS + 2; }

// reduction: 0 'mean' (over all B*H*W pixels), 1 'sum', 2 'none' (pixel_loss [B,H,W] receives the per-pixel terms; *loss their
// sum and glogits the gradient of that sum)
extern "C" int c2s_smooth_ce_ex(const float* logits, const long long* target, const float* class_w, const float* bg_distrib,
                                float* loss, float* glogits, float* pixel_loss, int B, int K, int H, int W, float label_smoothing,
                                long long bg_index, int reduction, int accumulate_loss, float* workspace, size_t ws_floats,
                                void* stream) {
    C2S_REQUIRE(logits && target && loss && workspace, "smooth_ce: null pointer");
    C2S_REQUIRE(B > 0 && H > 0 && W > 0 && K >= 1 && K <= MAXK, "smooth_ce: 1 <= K <= 32 classes");
    C2S_REQUIRE(label_smoothing >= 0.f && label_smoothing <= 1.f, "smooth_ce: label_smoothing outside [0, 1]");
    C2S_REQUIRE(reduction >= 0 && reduction <= 2 && (reduction != 2 || pixel_loss != nullptr), "smooth_ce: reduction 0 mean / 1 sum / 2 none (needs pixel_loss)");
    C2S_REQUIRE(ws_floats >= c2s_smooth_ce_workspace_floats(), "smooth_ce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = grid_for((long)B * H * W, FOCAL_BLOCKS);
    float* tot = workspace + 2 * FOCAL_BLOCKS;
    const double npix = reduction == 0 ? (double)B * H * W : 1.0;
    hipLaunchKernelGGL(smooth_ce_kernel, dim3(blocks), dim3(256), 0, st, logits, (const int64_t*)target, class_w, bg_distrib,
                       workspace, glogits, pixel_loss, B, K, H, W, label_smoothing, bg_index, (float)(1.0 / npix));
    C2S_CHECK_LAUNCH("smooth_ce");
    hipLaunchKernelGGL(smooth_ce_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, tot, loss, blocks, npix, accumulate_loss);
    C2S_CHECK_LAUNCH("smooth_ce_finalize");
    return C2S_OK;
}

extern "C" int c2s_smooth_ce(const float* logits, const long long* target, const float* class_w, const float* bg_distrib,
                             float* loss, float* glogits, int B, int K, int H, int W, float label_smoothing,
                             long long bg_index, int accumulate_loss, float* workspace, size_t ws_floats, void* stream) {
    return c2s_smooth_ce_ex(logits, target, class_w, bg_distrib, loss, glogits, nullptr, B, K, H, W, label_smoothing, bg_index, 0,
                            accumulate_loss, workspace, ws_floats, stream);
}
