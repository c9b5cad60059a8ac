// The steps on either side of the hot path that move whole batches:
//
//  * c2s_collate_series : dataset __getitem__ tail + pad_collate as ONE pass from the raw patch series (host-pinned or
//    device memory) to the padded, normalised, band-reordered model input in HBM, plus the padded dates and the per-frame
//    flags.  Reference: src/datasets/s2_ts_cz_crop.py:366-374 (np.load(...).astype(float32)[:, channels_order]),
//    :393-398 ((d - mean[c]) / std[c], fp32), src/utils.py:14-32 (pad_tensor / pad_collate: zero frames appended up to
//    the longest series of the batch, for the data and for the dates), train.py:291 (band order [2,1,0,4,5,6,3,7,8,9]).
//  * c2s_softmax_stitch : tiled-inference tail, src/webapp/prediction.py:310-333: Softmax(dim=1) of every patch's logits,
//    top-1 class, the einops re-tiling '(h w) c h1 w1 -> c (h h1) (w w1)' and the crop to 1098 x 1098, written straight
//    into the tile rasters.
//
// Both are HBM/PCIe-bound byte movers (no reuse): the collate kernel reads each source element once with 16-byte lanes
// (8-byte for 16-bit sources) and writes 16-byte lanes; padding frames are written as zeros by the same grid.
#include "common.h"

namespace {

struct CollateParams {
    const void* src;            // concatenated series: [sum_b T_b][Cs][HW] in the dataset's storage type
    const long long* offsets;   // [B+1] frame offsets of the series inside src (device-accessible)
    const long long* src_dates; // [sum_b T_b] (device-accessible) or NULL
    float* x;                   // [B][T][C][HW]
    long long* dates;           // [B][T] or NULL
    int* valid;                 // [B*T] or NULL
    int B, T, C, Cs, HW;
    int order[16];              // output channel c reads source channel order[c]
    float mean[16], stdv[16];   // in OUTPUT channel order (reference norm_values, already re-ordered)
    int normalise;
    float pad_value;
    int ndvi_a, ndvi_b;         // >= 0: the LAST output channel is the NDVI of source channels (a, b), not normalised
};

template <typename S> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<short> { using type = short4; };
template <> struct Vec4<unsigned short> { using type = ushort4; };

// grid: (chunks of the HW plane, C, B*T).  Every thread converts 4 consecutive pixels.
template <typename S>
__global__ __launch_bounds__(256) void collate_kernel(CollateParams p) {
    const int n = blockIdx.z, c = blockIdx.y;
    const int b = n / p.T, t = n % p.T;
    const long long beg = p.offsets[b], len = p.offsets[b + 1] - beg;
    const bool real = t < len;
    if (c == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        if (p.valid) p.valid[n] = real ? 1 : 0;
        if (p.dates) p.dates[n] = (real && p.src_dates) ? p.src_dates[beg + t] : 0;       // dates are padded with 0 too
    }
    float4* out = reinterpret_cast<float4*>(p.x + ((size_t)n * p.C + c) * p.HW);
    const int q = p.HW >> 2;
    if (!real) {
        const float4 z = make_float4(p.pad_value, p.pad_value, p.pad_value, p.pad_value);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < q; i += gridDim.x * 256) out[i] = z;
        return;
    }
    using V = typename Vec4<S>::type;
    if (p.ndvi_a >= 0 && c == p.C - 1) {
        // add_ndvi (s2_ts_cz_crop.py:376-391,401-402): (NIR - red) / (NIR + red) of the raw bands, 0 where the sum is 0 and where
        // the quotient leaves [-1, 1]; appended after the normalised bands, itself not normalised.  IEEE fp32 add / sub / div.
        const V* ia = reinterpret_cast<const V*>(static_cast<const S*>(p.src) + ((size_t)(beg + t) * p.Cs + p.ndvi_a) * p.HW);
        const V* ib = reinterpret_cast<const V*>(static_cast<const S*>(p.src) + ((size_t)(beg + t) * p.Cs + p.ndvi_b) * p.HW);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < q; i += gridDim.x * 256) {
            const V va = ia[i], vb = ib[i];
            const float a4[4] = {(float)va.x, (float)va.y, (float)va.z, (float)va.w};
            const float b4[4] = {(float)vb.x, (float)vb.y, (float)vb.z, (float)vb.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float sum = __fadd_rn(a4[k], b4[k]);
                float v = sum == 0.f ? 0.f : __fdiv_rn(__fsub_rn(a4[k], b4[k]), sum);
                if (v < -1.f || v > 1.f) v = 0.f;
                o[k] = v;
            }
            out[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    const V* in = reinterpret_cast<const V*>(static_cast<const S*>(p.src) + ((size_t)(beg + t) * p.Cs + p.order[c]) * p.HW);
    const float m = p.mean[c], s = p.stdv[c];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < q; i += gridDim.x * 256) {
        const V v = in[i];
        float4 f = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);      // .astype(np.float32): exact for 16-bit ints
        if (p.normalise) {
            // (d - mean) / std with IEEE fp32 subtraction and division, as torch evaluates it (no reciprocal, no fma)
            f.x = __fdiv_rn(__fsub_rn(f.x, m), s);
            f.y = __fdiv_rn(__fsub_rn(f.y, m), s);
            f.z = __fdiv_rn(__fsub_rn(f.z, m), s);
            f.w = __fdiv_rn(__fsub_rn(f.w, m), s);
        }
        out[i] = f;
    }
}

// One thread per output pixel of the cropped raster: reads the K logits of its patch pixel once, writes K probabilities.
__global__ __launch_bounds__(256) void softmax_stitch_kernel(const float* __restrict__ logits, float* __restrict__ proba,
                                                             long long* __restrict__ top1, int first_patch, int npatch,
                                                             int K, int ph, int pw, int grid_w, int out_h, int out_w) {
    const long total = (long)npatch * ph * pw;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x1 = (int)(e % pw), y1 = (int)((e / pw) % ph), pi = (int)(e / ((long)pw * ph));
        const int patch = first_patch + pi;
        const int oy = (patch / grid_w) * ph + y1, ox = (patch % grid_w) * pw + x1;
        if (oy >= out_h || ox >= out_w) continue;                                      // the crop (prediction.py:330-331)
        const float* lp = logits + ((size_t)pi * K * ph + y1) * pw + x1;
        const size_t cs = (size_t)ph * pw;
        float mx = lp[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[k * cs]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(lp[k * cs] - mx);
        const size_t os = (size_t)out_h * out_w;
        float* op = proba + (size_t)oy * out_w + ox;
        // top-1 = first maximum of the PROBABILITIES (pred_.max(dim=1)[1], prediction.py:316): logits closer than one
        // ulp of exp() tie there and resolve to the lower class, as in the reference
        float best = -1.f;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            const float pk = expf(lp[k * cs] - mx) / s;
            op[k * os] = pk;
            if (pk > best) { best = pk; arg = k; }
        }
        if (top1) top1[(size_t)oy * out_w + ox] = arg;
    }
}

}  // namespace

extern "C" int c2s_collate_series(const void* src, int src_dtype, const long long* offsets, const long long* src_dates,
                                  float* x, long long* dates, int* valid, int B, int T, int C, int Cs, int HW,
                                  const int* host_channel_order, const float* host_mean, const float* host_std,
                                  float pad_value, void* stream) {
    return c2s_collate_series_ndvi(src, src_dtype, offsets, src_dates, x, dates, valid, B, T, C, Cs, HW, host_channel_order,
                                   host_mean, host_std, pad_value, -1, -1, stream);
}

extern "C" int c2s_collate_series_ndvi(const void* src, int src_dtype, const long long* offsets, const long long* src_dates,
                                       float* x, long long* dates, int* valid, int B, int T, int C, int Cs, int HW,
                                       const int* host_channel_order, const float* host_mean, const float* host_std,
                                       float pad_value, int ndvi_a, int ndvi_b, void* stream) {
    C2S_REQUIRE(src && offsets && x, "collate_series: null pointer");
    const bool ndvi = ndvi_a >= 0 || ndvi_b >= 0;
    C2S_REQUIRE(!ndvi || (ndvi_a >= 0 && ndvi_a < Cs && ndvi_b >= 0 && ndvi_b < Cs && C >= 2), "collate_series: NDVI source channels out of range");
    C2S_REQUIRE(B > 0 && T > 0 && C > 0 && C <= 16 && Cs >= C - (ndvi ? 1 : 0) && HW > 0 && HW % 4 == 0, "collate_series: bad shape (C <= 16, HW %% 4 == 0)");
    C2S_REQUIRE((long)B * T < 65536 && C < 65536, "collate_series: too many frames for one launch");
    C2S_REQUIRE((host_mean == nullptr) == (host_std == nullptr), "collate_series: mean and std come together");
    CollateParams p = {};
    p.src = src; p.offsets = offsets; p.src_dates = src_dates; p.x = x; p.dates = dates; p.valid = valid;
    p.B = B; p.T = T; p.C = C; p.Cs = Cs; p.HW = HW; p.pad_value = pad_value;
    p.normalise = host_mean != nullptr;
    p.ndvi_a = ndvi ? ndvi_a : -1; p.ndvi_b = ndvi ? ndvi_b : -1;
    for (int c = 0; c < C - (ndvi ? 1 : 0); ++c) {
        p.order[c] = host_channel_order ? host_channel_order[c] : c;
        C2S_REQUIRE(p.order[c] >= 0 && p.order[c] < Cs, "collate_series: channel_order entry out of range");
        p.mean[c] = host_mean ? host_mean[c] : 0.f;
        p.stdv[c] = host_std ? host_std[c] : 1.f;
    }
    int chunks = (HW / 4 + 255) / 256;
    if (chunks > 16) chunks = 16;
    dim3 grid(chunks, C, B * T);
    hipStream_t st = (hipStream_t)stream;
    switch (src_dtype) {
        case C2S_SRC_F32: hipLaunchKernelGGL(collate_kernel<float>, grid, dim3(256), 0, st, p); break;
        case C2S_SRC_I16: hipLaunchKernelGGL(collate_kernel<short>, grid, dim3(256), 0, st, p); break;
        case C2S_SRC_U16: hipLaunchKernelGGL(collate_kernel<unsigned short>, grid, dim3(256), 0, st, p); break;
        default: c2s_set_error("collate_series: unknown src_dtype %d", src_dtype); return C2S_EINVAL;
    }
    C2S_CHECK_LAUNCH("collate_series");
    return C2S_OK;
}

extern "C" int c2s_softmax_stitch(const float* logits, float* proba, long long* top1, int first_patch, int npatch, int K,
                                  int ph, int pw, int grid_w, int out_h, int out_w, void* stream) {
    C2S_REQUIRE(logits && proba, "softmax_stitch: null pointer");
    C2S_REQUIRE(first_patch >= 0 && npatch > 0 && K > 0 && ph > 0 && pw > 0 && grid_w > 0 && out_h > 0 && out_w > 0,
                "softmax_stitch: bad shape");
    const long total = (long)npatch * ph * pw;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(softmax_stitch_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, logits, proba, top1,
                       first_patch, npatch, K, ph, pw, grid_w, out_h, out_w);
    C2S_CHECK_LAUNCH("softmax_stitch");
    return C2S_OK;
}
