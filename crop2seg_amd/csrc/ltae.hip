// L-TAE temporal attention for gfx950, fused and re-associated:
//
//   reference (src/backbones/tae.py:451-481, 738-847) per pixel sequence x[T,C]:
//     xhat = GroupNorm_16(x over (C/16 x T))            e_t = Wc xhat_t + bc + pe_t        (C -> 256)
//     k_t  = Wk e_t + bk ; score[h,t] = q_h . k_t[4h:4h+4] / 2 ; attn = dropout(softmax_T(mask(score)))
//     emb[16h+j] = sum_t attn[h,t] e_t[16h+j]
//   here:
//     score[h,t] = U[h,:] . xhat_t + s0[b,t,h]          U = q_h^T Wk_h Wc / 2   (16 x C, folded on the host)
//     z[h,:]     = sum_t attn[h,t] xhat_t               (per head, C-vector)
//     emb[16h+j] = Wc[16h+j,:] . z[h,:] + (sum_t attn[h,t]) bc[16h+j] + sum_t attn[h,t] pe[b,t,j]
//   i.e. the C->256 projection runs once per pixel instead of once per (pixel, t): 1/T of the FLOPs and
//   the [P,T,256] tensors of the reference are never materialised.
//
// Layout: everything stays NCHW.  x [B,T,C,hw], attn [16,B,T,hw], emb [B,256,hw]: the pixel index is the
// fastest axis, so a workgroup owns 16 adjacent pixels and its 256 threads are (pixel, head|group)
// pairs: lanes 0..15 of every 16-lane group read/write 64 contiguous bytes.
#include "common.h"

namespace {

constexpr int NH = 16;   // heads == GroupNorm groups (tae.py:428-435)
constexpr int DV = 16;   // d_model / n_head

struct LtaeParams {
    const float* x; const float* gamma; const float* beta; const float* U; const float* s0;
    const float* Wc; const float* bc; const float* pe; const int* valid;
    float* attn; float* attn_pre; float* emb; float* stats;
    // backward
    const float* g_emb; const float* g_attn; const float* attn_in; const float* attn_pre_in; const float* stats_in;
    float* gx; float* GS; float* V; float* Z; float* part_s0; float* part_bc; float* part_gb;
    const float* keep;
    int B, T, C, HW;
    float eps, drop_p;
    uint64_t seed;
    const uint64_t* seed_dev;   // optional device-side step counter added to the seed (hipGraph replay)
};

__device__ __forceinline__ float keep_scale(const LtaeParams& p, int h, long P_total, long pidx, int t) {
    if (p.drop_p <= 0.f) return 1.f;
    const long idx = ((long)h * P_total + pidx) * p.T + t;
    const float inv = 1.f / (1.f - p.drop_p);
    if (p.keep != nullptr) return p.keep[idx] != 0.f ? inv : 0.f;
    const uint64_t seed = p.seed + (p.seed_dev != nullptr ? *p.seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    return c2s_uniform(seed, (uint64_t)idx) >= p.drop_p ? inv : 0.f;
}

// ------------------------------------------------------------------------------------------ forward
// Workgroup = 16 adjacent pixels.  Streaming phases use threads = (pixel quad q, slot) with float4 loads (4 pixels
// per lane; 8-16 independent 16-byte loads in flight per thread keep >= 32 KB per CU outstanding), the per-pixel
// scalar phases use threads = (pixel, head).  Every x element is loaded by exactly one thread per phase:
//   1 stats   : slot = (group, t mod 4)         partial sums -> LDS -> per-(pixel,group) mean / rstd
//   2 scores  : slot = time step(s)             loop c; 16 head accumulators x 4 pixels; U through the scalar cache
//   3 softmax : (pixel, head)                   loop t over LDS; writes attn / attn_pre
//   4 z       : slot = channel(s)               loop t; 16 head accumulators x 4 pixels -> LDS (CH channels at a time)
//   5 emb     : (pixel, head)                   loop c over LDS; Wc rows from L1/L2
__global__ __launch_bounds__(256) void ltae_fwd_kernel(LtaeParams p) {
    extern __shared__ float lds[];
    const int C = p.C, T = p.T, HW = p.HW;
    const int CH = C > 64 ? 64 : C;             // channels per z/emb pass
    float* ABl = lds;                           // [C][2][16]   per-(channel,pixel) scale, shift of the GroupNorm
    float* Sl = ABl + C * 32;                   // [T][16][16]  scores -> attention (post-dropout)
    float* ASl = Sl + T * 256;                  // [16][16]     sum_t attention
    float* Zl = ASl + 256;                      // [16][CH][16] z = sum_t attn * xhat   (also the stats scratch)
    const int tid = threadIdx.x;
    const int q = tid & 3, slot = tid >> 2;      // streaming mapping
    const int px = tid & 15, hh = tid >> 4;      // per-pixel mapping
    const int tiles_per_b = (HW + 15) / 16;
    const int b = blockIdx.x / tiles_per_b;
    const int pix0 = (blockIdx.x % tiles_per_b) * 16;
    const bool actq = pix0 + 4 * q < HW;                 // HW % 4 == 0: a quad is entirely inside or outside
    const int pixq = actq ? pix0 + 4 * q : 0;
    const bool act = pix0 + px < HW;
    const int pix = act ? pix0 + px : HW - 1;
    const long pidx = (long)b * HW + pix, Ptot = (long)p.B * HW;
    const int cpg = C / NH;
    const float* xq = p.x + (size_t)b * T * C * HW + pixq;     // + (t*C + c)*HW : float4 of 4 pixels

    // ---- phase 1: GroupNorm statistics (padded frames included, tae.py:461); shifted sums, 4 partials per group
    {
        const int g = slot & 15, tq = slot >> 4;
        const f32x4 K0 = *reinterpret_cast<const f32x4*>(xq + (size_t)(g * cpg) * HW);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
        for (int t = tq; t < T; t += 4) {
#pragma unroll 8
            for (int cc = 0; cc < cpg; ++cc) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(xq + (size_t)(t * C + g * cpg + cc) * HW) - K0;
                s += d;
                ss += d * d;
            }
        }
        // scratch [g][tq][2][16]
        *reinterpret_cast<f32x4*>(Zl + ((g * 4 + tq) * 2 + 0) * 16 + 4 * q) = s;
        *reinterpret_cast<f32x4*>(Zl + ((g * 4 + tq) * 2 + 1) * 16 + 4 * q) = ss;
    }
    __syncthreads();
    {
        const int g = hh;       // (pixel, group)
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            s += Zl[((g * 4 + tq) * 2 + 0) * 16 + px];
            ss += Zl[((g * 4 + tq) * 2 + 1) * 16 + px];
        }
        const float K0 = p.x[(size_t)b * T * C * HW + (size_t)(g * cpg) * HW + pix];
        const float inv_n = 1.f / (float)(cpg * T);
        const float md = s * inv_n;
        const float mean = K0 + md;
        const float var = fmaxf(ss * inv_n - md * md, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        if (act) {
            p.stats[(pidx * NH + g) * 2] = mean;
            p.stats[(pidx * NH + g) * 2 + 1] = rstd;
        }
        for (int cc = 0; cc < cpg; ++cc) {
            const int c = g * cpg + cc;
            const float a = p.gamma[c] * rstd;
            ABl[(c * 2 + 0) * 16 + px] = a;
            ABl[(c * 2 + 1) * 16 + px] = p.beta[c] - mean * a;
        }
    }
    __syncthreads();

    // ---- phase 2: scores.  U[h][c] is wave-uniform: it comes through the scalar cache, not LDS.
    for (int t = slot; t < T; t += 64) {
        f32x4 sc[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float v = p.s0[(b * T + t) * NH + h];
            sc[h] = (f32x4){v, v, v, v};
        }
        const float* xt = xq + (size_t)t * C * HW;
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xt + (size_t)c * HW);
            const f32x4 xh = *reinterpret_cast<const f32x4*>(ABl + (c * 2 + 0) * 16 + 4 * q) * xv +
                             *reinterpret_cast<const f32x4*>(ABl + (c * 2 + 1) * 16 + 4 * q);
#pragma unroll
            for (int h = 0; h < NH; ++h) sc[h] += p.U[h * C + c] * xh;
        }
        const bool padded = p.valid != nullptr && p.valid[b * T + t] == 0;
#pragma unroll
        for (int h = 0; h < NH; ++h)
            *reinterpret_cast<f32x4*>(Sl + (t * 16 + h) * 16 + 4 * q) = padded ? (f32x4){-1e6f, -1e6f, -1e6f, -1e6f} : sc[h];   // tae.py:831
    }
    __syncthreads();

    // ---- phase 3: softmax over T for (pixel, head), dropout
    {
        float mx = -3.0e38f;
        for (int t = 0; t < T; ++t) mx = fmaxf(mx, Sl[(t * 16 + hh) * 16 + px]);
        float den = 0.f;
        for (int t = 0; t < T; ++t) {
            const float e = __expf(Sl[(t * 16 + hh) * 16 + px] - mx);
            Sl[(t * 16 + hh) * 16 + px] = e;
            den += e;
        }
        const float inv_den = 1.f / den;
        float asum = 0.f;
        for (int t = 0; t < T; ++t) {
            const float a = Sl[(t * 16 + hh) * 16 + px] * inv_den;
            const float ad = a * keep_scale(p, hh, Ptot, pidx, t);
            const size_t o = ((size_t)(hh * p.B + b) * T + t) * HW + pix;
            if (act) {
                if (p.attn_pre != nullptr) p.attn_pre[o] = a;
                p.attn[o] = ad;
            }
            Sl[(t * 16 + hh) * 16 + px] = ad;
            asum += ad;
        }
        ASl[hh * 16 + px] = asum;
    }
    if (p.emb == nullptr) return;   // W-TAE: attention masks only (tae.py:619)
    __syncthreads();

    // ---- phases 4+5, CH channels at a time
    float o[DV];
#pragma unroll
    for (int j = 0; j < DV; ++j) o[j] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CH) {
        // 4: z[h] = sum_t attn[h,t] * x[t,c] for channel c = c0 + slot (4 pixels per thread)
        for (int c = c0 + slot; c < c0 + CH; c += 64) {
            f32x4 z[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) z[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int t = 0; t < T; ++t) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xq + (size_t)(t * C + c) * HW);
#pragma unroll
                for (int h = 0; h < NH; ++h) z[h] += *reinterpret_cast<const f32x4*>(Sl + (t * 16 + h) * 16 + 4 * q) * xv;
            }
            const f32x4 a = *reinterpret_cast<const f32x4*>(ABl + (c * 2 + 0) * 16 + 4 * q);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(ABl + (c * 2 + 1) * 16 + 4 * q);
#pragma unroll
            for (int h = 0; h < NH; ++h)
                *reinterpret_cast<f32x4*>(Zl + (h * CH + (c - c0)) * 16 + 4 * q) =
                    a * z[h] + bb * *reinterpret_cast<const f32x4*>(ASl + h * 16 + 4 * q);
        }
        __syncthreads();
        // 5: emb slice of (pixel, head)
        for (int c = 0; c < CH; ++c) {
            const float zz = Zl[(hh * CH + c) * 16 + px];
#pragma unroll
            for (int j = 0; j < DV; ++j) o[j] = fmaf(p.Wc[(size_t)(hh * DV + j) * C + c0 + c], zz, o[j]);
        }
        __syncthreads();
    }
    {
        const float asum = ASl[hh * 16 + px];
#pragma unroll
        for (int j = 0; j < DV; ++j) o[j] += asum * p.bc[hh * DV + j];
        for (int t = 0; t < T; ++t) {
            const float ad = Sl[(t * 16 + hh) * 16 + px];
#pragma unroll
            for (int j = 0; j < DV; ++j) o[j] = fmaf(ad, p.pe[(b * T + t) * DV + j], o[j]);
        }
        if (act) {
#pragma unroll
            for (int j = 0; j < DV; ++j) p.emb[((size_t)b * NH * DV + hh * DV + j) * HW + pix] = o[j];
        }
    }
}

// ------------------------------------------------------------------------------------------ backward, part 1
// Softmax/dropout backward -> GS (d score), V = sum_t gs xhat, Z = sum_t attn xhat, per-tile partials of d s0 and
// d bc.  Workgroup = 8 adjacent pixels (so that the [T][16][8] tiles fit in LDS up to T = 64).  Streaming phases use
// threads = (pixel quad q in {0,1}, slot in [0,128)) with float4 loads, per-pixel phases use (pixel, item):
//   A1 r[h][c] = sum_j ge[16h+j] Wc[16h+j][c]   (pixel, 32 items)      RC channels at a time
//   A2 dot[h,t] += r[h][c] xhat[t,c]            slot = (t, channel half of the chunk); halves add in turn
//   B  softmax / dropout backward               (pixel, head)
//   C  V (slots 0..63) and Z (slots 64..127)     slot % 64 = channel
constexpr int BPT = 8;      // pixels per backward tile

__global__ __launch_bounds__(256) void ltae_bwd_heads_kernel(LtaeParams p) {
    constexpr int PT = BPT;
    constexpr int RC = 32;                       // channels per r-chunk
    extern __shared__ float lds[];
    const int C = p.C, T = p.T, HW = p.HW;
    float* ABl = lds;                            // [C][2][PT]
    float* GEl = ABl + C * 2 * PT;               // [256][PT]
    float* Rl = GEl + 256 * PT;                  // [16][RC][PT]
    float* Dl = Rl + NH * RC * PT;               // [T][16][PT]  dot -> ga -> gs
    float* Al = Dl + T * NH * PT;                // [T][16][PT]  attention (post-dropout)
    float* APl = Al + T * NH * PT;               // [T][16][PT]  attention before dropout (softmax output)
    float* GAl = APl + T * NH * PT;              // [T][16][PT]  incoming d attn
    float* SUMl = GAl + T * NH * PT;             // [2][16][PT]  sum_t attn, sum_t gs
    const int tid = threadIdx.x;
    const int px = tid & 7, item = tid >> 3;     // per-pixel mapping: 8 pixels x 32 items
    const int q = tid & 1, slot = tid >> 1;      // streaming mapping: 2 quads x 128 slots
    const int tiles_per_b = (HW + PT - 1) / PT;
    const int b = blockIdx.x / tiles_per_b, pix0 = (blockIdx.x % tiles_per_b) * PT;
    const bool act = pix0 + px < HW;
    const float actf = act ? 1.f : 0.f;
    const int pix = act ? pix0 + px : HW - 1;
    const bool actq = pix0 + 4 * q < HW;
    const int pixq = actq ? pix0 + 4 * q : 0;
    const long pidx = (long)b * HW + pix, Ptot = (long)p.B * HW;
    const int cpg = C / NH;
    const float* xq = p.x + (size_t)b * T * C * HW + pixq;

    if (item < NH) {
        const int g = item;
        const float mean = p.stats_in[(pidx * NH + g) * 2], rstd = p.stats_in[(pidx * NH + g) * 2 + 1];
        for (int cc = 0; cc < cpg; ++cc) {
            const int c = g * cpg + cc;
            const float a = p.gamma[c] * rstd;
            ABl[(c * 2 + 0) * PT + px] = a;
            ABl[(c * 2 + 1) * PT + px] = p.beta[c] - mean * a;
        }
    }
    for (int ch = item; ch < NH * DV; ch += 32)
        GEl[ch * PT + px] = p.g_emb != nullptr ? p.g_emb[((size_t)b * NH * DV + ch) * HW + pix] : 0.f;
    // tiles of attn / attn_pre / d attn: float4 (pixel quad) loads, all issued back to back
    for (int i = slot; i < T * NH; i += 128) {
        const int t = i / NH, h = i % NH;
        const size_t o = ((size_t)(h * p.B + b) * T + t) * HW + pixq;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(Dl + i * PT + 4 * q) = z4;
        *reinterpret_cast<f32x4*>(Al + i * PT + 4 * q) = *reinterpret_cast<const f32x4*>(p.attn_in + o);
        *reinterpret_cast<f32x4*>(APl + i * PT + 4 * q) = *reinterpret_cast<const f32x4*>(p.attn_pre_in + o);
        *reinterpret_cast<f32x4*>(GAl + i * PT + 4 * q) = p.g_attn != nullptr ? *reinterpret_cast<const f32x4*>(p.g_attn + o) : z4;
    }
    __syncthreads();

    if (p.g_emb != nullptr) {
        for (int c0 = 0; c0 < C; c0 += RC) {
            // A1
            for (int it = item; it < NH * RC; it += 32) {
                const int h = it / RC, cc = it % RC;
                float r = 0.f;
#pragma unroll
                for (int j = 0; j < DV; ++j) r = fmaf(GEl[(h * DV + j) * PT + px], p.Wc[(size_t)(h * DV + j) * C + c0 + cc], r);
                Rl[it * PT + px] = r;
            }
            __syncthreads();
            // A2: slot = (t, half); each half covers RC/2 channels of the chunk
            const int t = slot >> 1, half = slot & 1;
            f32x4 acc[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t < T) {
                const float* xt = xq + (size_t)t * C * HW;
#pragma unroll 4
                for (int cc = half * (RC / 2); cc < (half + 1) * (RC / 2); ++cc) {
                    const int c = c0 + cc;
                    const f32x4 xh = *reinterpret_cast<const f32x4*>(ABl + (c * 2 + 0) * PT + 4 * q) *
                                         *reinterpret_cast<const f32x4*>(xt + (size_t)c * HW) +
                                     *reinterpret_cast<const f32x4*>(ABl + (c * 2 + 1) * PT + 4 * q);
#pragma unroll
                    for (int h = 0; h < NH; ++h) acc[h] += *reinterpret_cast<const f32x4*>(Rl + (h * RC + cc) * PT + 4 * q) * xh;
                }
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (t < T && half == hf) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        f32x4* d = reinterpret_cast<f32x4*>(Dl + (t * NH + h) * PT + 4 * q);
                        *d = *d + acc[h];
                    }
                }
                __syncthreads();
            }
        }
    }

    // B: (pixel, head)
    if (item < NH) {
        const int hh = item;
        float ge[DV];
        float gebc = 0.f;
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            ge[j] = GEl[(hh * DV + j) * PT + px];
            gebc = fmaf(ge[j], p.bc[hh * DV + j], gebc);
        }
        float dsum = 0.f, asum = 0.f;
        for (int t = 0; t < T; ++t) {
            const size_t o = ((size_t)(hh * p.B + b) * T + t) * HW + pix;
            float gap = Dl[(t * NH + hh) * PT + px] + gebc + GAl[(t * NH + hh) * PT + px];
#pragma unroll
            for (int j = 0; j < DV; ++j) gap = fmaf(ge[j], p.pe[(b * T + t) * DV + j], gap);
            const float ga = gap * keep_scale(p, hh, Ptot, pidx, t);
            dsum = fmaf(APl[(t * NH + hh) * PT + px], ga, dsum);
            Dl[(t * NH + hh) * PT + px] = ga;
            asum += Al[(t * NH + hh) * PT + px];
        }
        float gssum = 0.f;
        for (int t = 0; t < T; ++t) {
            const size_t o = ((size_t)(hh * p.B + b) * T + t) * HW + pix;
            const float gs = APl[(t * NH + hh) * PT + px] * (Dl[(t * NH + hh) * PT + px] - dsum);
            Dl[(t * NH + hh) * PT + px] = gs;
            gssum += gs;
            if (act) p.GS[o] = gs;
            float r = gs * actf;       // d s0[b,t,h]: sum over the pixels of the tile
            r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64);
            if (px == 0) p.part_s0[((size_t)blockIdx.x * T + t) * NH + hh] = r;
        }
        SUMl[hh * PT + px] = asum;
        SUMl[(NH + hh) * PT + px] = gssum;
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            float r = ge[j] * asum * actf;
            r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64);
            if (px == 0) p.part_bc[(size_t)blockIdx.x * NH * DV + hh * DV + j] = r;
        }
    }
    __syncthreads();

    // C: slots 0..63 produce V = sum_t gs xhat, slots 64..127 produce Z = sum_t attn xhat; channel = slot % 64 (+64)
    {
        const bool isz = slot >= 64;
        const float* Wl_ = isz ? Al : Dl;
        float* outp = isz ? p.Z : p.V;
        const float* suml = isz ? SUMl : SUMl + NH * PT;
        for (int c = slot & 63; c < C; c += 64) {
            f32x4 v[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) v[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int t = 0; t < T; ++t) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xq + (size_t)(t * C + c) * HW);
#pragma unroll
                for (int h = 0; h < NH; ++h) v[h] += *reinterpret_cast<const f32x4*>(Wl_ + (t * NH + h) * PT + 4 * q) * xv;
            }
            if (actq) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ABl + (c * 2 + 0) * PT + 4 * q);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(ABl + (c * 2 + 1) * PT + 4 * q);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const size_t o = (((size_t)b * NH + h) * C + c) * HW + pixq;
                    *reinterpret_cast<f32x4*>(outp + o) = a * v[h] + bb * *reinterpret_cast<const f32x4*>(suml + h * PT + 4 * q);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward, part 2
// d xhat[t,c] = sum_h (attn[h,t] r[h,c] + gs[h,t] U[h,c]), then the per-pixel GroupNorm backward.
// Streaming threads = (quad, slot): slot = (channel c = slot % 64 (+64), time half th = slot / 64).
__global__ __launch_bounds__(256) void ltae_bwd_gx_kernel(LtaeParams p) {
    constexpr int PT = BPT;
    extern __shared__ float lds[];
    const int C = p.C, T = p.T, HW = p.HW;
    float* GEl = lds;                            // [256][PT]
    float* Gl = GEl + 256 * PT;                  // [T][16][PT] gs
    float* Al = Gl + T * NH * PT;                // [T][16][PT] attn
    float* Ml = Al + T * NH * PT;                // [C][2 th][2][PT]  partial (sum dxn, sum dxn*xn)
    float* STl = Ml + C * 4 * PT;                // [16][4][PT] mean, rstd, m1, m2 of each group
    float* Pl = STl + NH * 4 * PT;               // [C][2 th][2 q][2]  partial (dgamma, dbeta)
    const int tid = threadIdx.x;
    const int px = tid & 7, item = tid >> 3;
    const int q = tid & 1, slot = tid >> 1;
    const int tiles_per_b = (HW + PT - 1) / PT;
    const int b = blockIdx.x / tiles_per_b, pix0 = (blockIdx.x % tiles_per_b) * PT;
    const bool act = pix0 + px < HW;
    const int pix = act ? pix0 + px : HW - 1;
    const bool actq = pix0 + 4 * q < HW;
    const float actqf = actq ? 1.f : 0.f;
    const int pixq = actq ? pix0 + 4 * q : 0;
    const long pidx = (long)b * HW + pix;
    const int cpg = C / NH;
    for (int ch = item; ch < NH * DV; ch += 32)
        GEl[ch * PT + px] = p.g_emb != nullptr ? p.g_emb[((size_t)b * NH * DV + ch) * HW + pix] : 0.f;
    {
        const int q_ = tid & 1, slot_ = tid >> 1;
        const bool actq_ = pix0 + 4 * q_ < HW;
        const int pixq_ = actq_ ? pix0 + 4 * q_ : 0;
        for (int i = slot_; i < T * NH; i += 128) {
            const int t = i / NH, h = i % NH;
            const size_t o = ((size_t)(h * p.B + b) * T + t) * HW + pixq_;
            *reinterpret_cast<f32x4*>(Gl + i * PT + 4 * q_) = *reinterpret_cast<const f32x4*>(p.GS + o);
            *reinterpret_cast<f32x4*>(Al + i * PT + 4 * q_) = *reinterpret_cast<const f32x4*>(p.attn_in + o);
        }
    }
    if (item < NH) {
        STl[(item * 4 + 0) * PT + px] = p.stats_in[(pidx * NH + item) * 2];
        STl[(item * 4 + 1) * PT + px] = p.stats_in[(pidx * NH + item) * 2 + 1];
    }
    __syncthreads();
    const float* xq = p.x + (size_t)b * T * C * HW + pixq;
    float* gxq = p.gx + (size_t)b * T * C * HW + pixq;
    const int th = slot >> 6;
    const int t_beg = th == 0 ? 0 : (T + 1) / 2, t_end = th == 0 ? (T + 1) / 2 : T;
    for (int c = slot & 63; c < C; c += 64) {
        const int g = c / cpg;
        const f32x4 mean = *reinterpret_cast<const f32x4*>(STl + (g * 4 + 0) * PT + 4 * q);
        const f32x4 rstd = *reinterpret_cast<const f32x4*>(STl + (g * 4 + 1) * PT + 4 * q);
        f32x4 r[NH];
        float u[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < DV; ++j)
                sacc += *reinterpret_cast<const f32x4*>(GEl + (h * DV + j) * PT + 4 * q) * p.Wc[(size_t)(h * DV + j) * C + c];
            r[h] = sacc;
            u[h] = p.U[h * C + c];
        }
        const float gm = p.gamma[c];
        f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = dg, m1 = dg, m2 = dg;
        for (int t = t_beg; t < t_end; ++t) {
            f32x4 gxh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < NH; ++h)
                gxh += *reinterpret_cast<const f32x4*>(Al + (t * NH + h) * PT + 4 * q) * r[h] +
                       *reinterpret_cast<const f32x4*>(Gl + (t * NH + h) * PT + 4 * q) * u[h];
            const f32x4 xn = (*reinterpret_cast<const f32x4*>(xq + (size_t)(t * C + c) * HW) - mean) * rstd;
            dg += gxh * xn;
            db += gxh;
            const f32x4 dxn = gxh * gm;
            m1 += dxn;
            m2 += dxn * xn;
            if (actq) *reinterpret_cast<f32x4*>(gxq + (size_t)(t * C + c) * HW) = dxn;
        }
        *reinterpret_cast<f32x4*>(Ml + ((c * 2 + th) * 2 + 0) * PT + 4 * q) = m1;
        *reinterpret_cast<f32x4*>(Ml + ((c * 2 + th) * 2 + 1) * PT + 4 * q) = m2;
        Pl[((c * 2 + th) * 2 + q) * 2 + 0] = ((dg.x + dg.y) + (dg.z + dg.w)) * actqf;
        Pl[((c * 2 + th) * 2 + q) * 2 + 1] = ((db.x + db.y) + (db.z + db.w)) * actqf;
    }
    __syncthreads();
    if (item < NH) {
        const int g = item;
        float m1 = 0.f, m2 = 0.f;
        for (int cc = 0; cc < cpg; ++cc)
            for (int h2 = 0; h2 < 2; ++h2) {
                m1 += Ml[(((g * cpg + cc) * 2 + h2) * 2 + 0) * PT + px];
                m2 += Ml[(((g * cpg + cc) * 2 + h2) * 2 + 1) * PT + px];
            }
        const float inv_n = 1.f / (float)(cpg * T);
        STl[(g * 4 + 2) * PT + px] = m1 * inv_n;
        STl[(g * 4 + 3) * PT + px] = m2 * inv_n;
    }
    for (int c = tid; c < C; c += 256) {
        const float dg = (Pl[((c * 2 + 0) * 2 + 0) * 2] + Pl[((c * 2 + 0) * 2 + 1) * 2]) + (Pl[((c * 2 + 1) * 2 + 0) * 2] + Pl[((c * 2 + 1) * 2 + 1) * 2]);
        const float db = (Pl[((c * 2 + 0) * 2 + 0) * 2 + 1] + Pl[((c * 2 + 0) * 2 + 1) * 2 + 1]) +
                         (Pl[((c * 2 + 1) * 2 + 0) * 2 + 1] + Pl[((c * 2 + 1) * 2 + 1) * 2 + 1]);
        p.part_gb[((size_t)blockIdx.x * C + c) * 2] = dg;
        p.part_gb[((size_t)blockIdx.x * C + c) * 2 + 1] = db;
    }
    __syncthreads();
    if (actq) {
        for (int c = slot & 63; c < C; c += 64) {
            const int g = c / cpg;
            const f32x4 mean = *reinterpret_cast<const f32x4*>(STl + (g * 4 + 0) * PT + 4 * q);
            const f32x4 rstd = *reinterpret_cast<const f32x4*>(STl + (g * 4 + 1) * PT + 4 * q);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(STl + (g * 4 + 2) * PT + 4 * q);
            const f32x4 m2 = *reinterpret_cast<const f32x4*>(STl + (g * 4 + 3) * PT + 4 * q);
            for (int t = t_beg; t < t_end; ++t) {
                const size_t o = (size_t)(t * C + c) * HW;
                const f32x4 xn = (*reinterpret_cast<const f32x4*>(xq + o) - mean) * rstd;
                f32x4* gp = reinterpret_cast<f32x4*>(gxq + o);
                *gp = rstd * (*gp - m1 - xn * m2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ reductions
// out[grp][k] = sum_{i<count} part[(grp*count + i)*K + k]; one wave per output element, lanes stride the tiles
// (fixed lane assignment + fixed shuffle tree: bitwise reproducible)
__global__ __launch_bounds__(64) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int count,
                                                             int K, long total) {
    const long e = blockIdx.x;
    if (e >= total) return;
    const long grp = e / K;
    const int k = (int)(e % K);
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 64) s += part[((size_t)grp * count + i) * K + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[e] = (float)s;
}

// gU[h,c] = sum_{b,pix} V[b,h,c,pix]     one wave per (h,c)
__global__ __launch_bounds__(64) void sum_over_pixels_kernel(const float* __restrict__ V, float* __restrict__ out, int B,
                                                             int rows, int HW) {
    const int row = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* v = V + ((size_t)b * rows + row) * HW;
        for (int i = lane; i < HW; i += 64) s += v[i];
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

// gWc[16h+j][c] = sum_{b,pix} ge[b,16h+j,pix] * Z[b,h,c,pix]    workgroup per (h,c): 16 j x 16 pixel lanes
__global__ __launch_bounds__(256) void gwc_kernel(const float* __restrict__ ge, const float* __restrict__ Z,
                                                  float* __restrict__ gWc, int B, int C, int HW) {
    const int h = blockIdx.x / C, c = blockIdx.x % C;
    const int j = threadIdx.x >> 4, q = threadIdx.x & 15;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* z = Z + (((size_t)b * NH + h) * C + c) * HW;
        const float* g = ge + ((size_t)b * NH * DV + h * DV + j) * HW;
        for (int i = q; i < HW; i += 16) s += g[i] * z[i];
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    if (q == 0) gWc[(size_t)(h * DV + j) * C + c] = s;
}

// per-pixel GroupNorm over channel groups of a [B,C,HW] tensor (tae.py:437-440,488)
__global__ void pixel_gn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ stats,
                                    int B, int C, int HW, int groups, float eps) {
    const long total = (long)B * groups * HW;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int pix = (int)(e % HW);
    const long bg = e / HW;
    const int g = (int)(bg % groups), b = (int)(bg / groups);
    const int cpg = C / groups;
    const float* xp = x + ((size_t)b * C + g * cpg) * HW + pix;
    float s = 0.f;
    for (int c = 0; c < cpg; ++c) s += xp[(size_t)c * HW];
    const float mean = s / cpg;
    float m2 = 0.f;
    for (int c = 0; c < cpg; ++c) { const float d = xp[(size_t)c * HW] - mean; m2 += d * d; }
    const float rstd = rsqrtf(m2 / cpg + eps);
    stats[e * 2] = mean; stats[e * 2 + 1] = rstd;
    float* yp = y + ((size_t)b * C + g * cpg) * HW + pix;
    for (int c = 0; c < cpg; ++c)
        yp[(size_t)c * HW] = (xp[(size_t)c * HW] - mean) * rstd * gamma[g * cpg + c] + beta[g * cpg + c];
}

// gx, and per-thread-block partials of dgamma/dbeta: part[block][C][2] via atomics-free two-level scheme:
// each thread handles one (b,g,pix); dgamma/dbeta partials are accumulated with a wave reduction when the
// 64 lanes share (b,g) (HW % 64 == 0) and written per wave.
__global__ __launch_bounds__(64) void pixel_gn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          const float* __restrict__ gamma, const float* __restrict__ stats,
                                                          float* __restrict__ gx, float* __restrict__ part, int B, int C,
                                                          int HW, int groups, int chunks) {
    // grid = B*groups*chunks waves; wave handles pixels [chunk*len, ...) of one (b,g)
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x % chunks;
    const int bg = blockIdx.x / chunks;
    const int g = bg % groups, b = bg / groups;
    const int cpg = C / groups;
    const int len = (HW + chunks - 1) / chunks;
    const int beg = chunk * len, end = (beg + len) < HW ? (beg + len) : HW;
    float dg[16], db[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { dg[c] = 0.f; db[c] = 0.f; }
    for (int pix = beg + lane; pix < end; pix += 64) {
        const long e = ((long)b * groups + g) * HW + pix;
        const float mean = stats[e * 2], rstd = stats[e * 2 + 1];
        const float* xp = x + ((size_t)b * C + g * cpg) * HW + pix;
        const float* gp = gy + ((size_t)b * C + g * cpg) * HW + pix;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < cpg) {
                const float xn = (xp[(size_t)c * HW] - mean) * rstd;
                const float gg = gp[(size_t)c * HW];
                dg[c] += gg * xn; db[c] += gg;
                const float dxn = gg * gamma[g * cpg + c];
                m1 += dxn; m2 += dxn * xn;
            }
        m1 /= cpg; m2 /= cpg;
        float* gxp = gx + ((size_t)b * C + g * cpg) * HW + pix;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < cpg) {
                const float xn = (xp[(size_t)c * HW] - mean) * rstd;
                gxp[(size_t)c * HW] = rstd * (gp[(size_t)c * HW] * gamma[g * cpg + c] - m1 - xn * m2);
            }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < cpg) {
            const float a = wave_sum(dg[c]), bb = wave_sum(db[c]);
            if (lane == 0) {
                // part[(b*chunks+chunk)][C][2]
                const size_t o = (((size_t)b * chunks + chunk) * C + g * cpg + c) * 2;
                part[o] = a; part[o + 1] = bb;
            }
        }
}

// y[b,c,pix] = x * keep / (1-p), keep indexed pixel-major like the reference's [P, C] activations (tae.py:448)
__global__ void dropout_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW, float p,
                                    uint64_t seed0, const uint64_t* __restrict__ seed_dev, const float* __restrict__ keep) {
    const uint64_t seed = seed0 + (seed_dev != nullptr ? *seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    const long total = (long)B * C * HW;
    const float inv = 1.f / (1.f - p);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(e % HW);
        const long bc = e / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        const long idx = ((long)b * HW + pix) * C + c;
        const bool k = keep != nullptr ? keep[idx] != 0.f : c2s_uniform(seed, (uint64_t)idx) >= p;
        y[e] = k ? x[e] * inv : 0.f;
    }
}

size_t fwd_lds(const c2s_ltae_desc* d) {
    const size_t CH = d->C > 64 ? 64 : d->C;
    return ((size_t)d->C * 32 + (size_t)d->T * 256 + 256 + NH * CH * 16) * 4;
}
int bwd_pt(const c2s_ltae_desc*) { return BPT; }
size_t bwd1_lds(const c2s_ltae_desc* d) {
    const size_t PT = BPT;
    return ((size_t)d->C * 2 * PT + 256 * PT + NH * 32 * PT + 4 * (size_t)d->T * NH * PT + 2 * NH * PT) * 4;
}
size_t bwd2_lds(const c2s_ltae_desc* d) {
    const size_t PT = BPT;
    return (256 * PT + 2 * (size_t)d->T * NH * PT + (size_t)d->C * 4 * PT + NH * 4 * PT + (size_t)d->C * 8) * 4;
}

int check(const c2s_ltae_desc* d) {
    C2S_REQUIRE(d && d->B > 0 && d->T > 0 && d->C > 0 && d->HW > 0, "ltae: bad shape");
    C2S_REQUIRE(d->n_head == NH && d->d_model == NH * DV, "ltae: only n_head=16, d_model=256 are built");
    C2S_REQUIRE(d->C % NH == 0 && d->C / NH <= 16 && d->T <= 64, "ltae: C must be a multiple of 16 and <= 256, T <= 64");
    C2S_REQUIRE(bwd1_lds(d) <= 160 * 1024 && bwd2_lds(d) <= 160 * 1024 && fwd_lds(d) <= 160 * 1024, "ltae: T*C too large for the LDS tile");
    C2S_REQUIRE(d->C % 64 == 0, "ltae: C must be a multiple of 64");
    C2S_REQUIRE(d->HW % 4 == 0, "ltae: h*w must be a multiple of 4");
    C2S_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "ltae: bad dropout p");
    return C2S_OK;
}

void fill(LtaeParams& p, const c2s_ltae_desc* d) {
    p.B = d->B; p.T = d->T; p.C = d->C; p.HW = d->HW; p.eps = d->eps; p.drop_p = d->dropout_p; p.seed = d->seed; p.seed_dev = d->seed_dev;
    p.keep = d->keep;
}

}  // namespace

extern "C" int c2s_ltae_attn_fwd(const c2s_ltae_desc* d, const float* x, const float* gamma, const float* beta,
                                 const float* U, const float* s0, const float* Wc, const float* bc, const float* pe,
                                 const int* valid, float* attn, float* attn_pre, float* emb, float* stats,
                                 void* stream) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(x && gamma && beta && U && s0 && attn && stats, "ltae_fwd: null pointer");
    C2S_REQUIRE(emb == nullptr || (Wc && bc && pe), "ltae_fwd: embedding output needs Wc, bc, pe");
    LtaeParams p = {};
    fill(p, d);
    p.x = x; p.gamma = gamma; p.beta = beta; p.U = U; p.s0 = s0; p.Wc = Wc; p.bc = bc; p.pe = pe; p.valid = valid;
    p.attn = attn; p.attn_pre = attn_pre; p.emb = emb; p.stats = stats;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&ltae_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(ltae_fwd_kernel, dim3(d->B * ((d->HW + 15) / 16)), dim3(256), fwd_lds(d), (hipStream_t)stream, p);
    C2S_CHECK_LAUNCH("ltae_fwd");
    return C2S_OK;
}

// workspace: GS [16,B,T,HW] | V [B,16,C,HW] | Z [B,16,C,HW] | part_s0 [tiles][T][16] | part_bc [tiles][256]
//            | part_gb [tiles][C][2]
extern "C" size_t c2s_ltae_bwd_workspace_floats(const c2s_ltae_desc* d) {
    if (!d) return 0;
    const size_t tiles = (size_t)d->B * ((d->HW + 7) / 8);   // upper bound (8-pixel tiles)
    return (size_t)NH * d->B * d->T * d->HW + 2 * (size_t)d->B * NH * d->C * d->HW + tiles * d->T * NH + tiles * 256 +
           tiles * d->C * 2;
}

extern "C" int c2s_ltae_attn_bwd(const c2s_ltae_desc* d, const float* x, const float* gamma, const float* beta,
                                 const float* U, const float* s0, const float* Wc, const float* bc, const float* pe,
                                 const int* valid, const float* attn, const float* attn_pre, const float* stats,
                                 const float* g_emb, const float* g_attn, float* gx, float* gU, float* gs0, float* gWc,
                                 float* gbc, float* ggamma, float* gbeta, float* workspace, size_t ws_floats,
                                 void* stream) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(x && gamma && beta && U && Wc && bc && pe && attn && attn_pre && stats && gx && gU && gs0 && gWc && gbc &&
                    ggamma && gbeta && workspace,
                "ltae_bwd: null pointer");
    C2S_REQUIRE(ws_floats >= c2s_ltae_bwd_workspace_floats(d), "ltae_bwd: workspace too small");
    (void)s0; (void)valid;
    const int PT = bwd_pt(d);
    const size_t tiles = (size_t)d->B * ((d->HW + PT - 1) / PT);
    const size_t tiles_ws = (size_t)d->B * ((d->HW + 7) / 8);
    LtaeParams p = {};
    fill(p, d);
    p.x = x; p.gamma = gamma; p.beta = beta; p.U = U; p.Wc = Wc; p.bc = bc; p.pe = pe;
    p.attn_in = attn; p.attn_pre_in = attn_pre; p.stats_in = stats; p.g_emb = g_emb; p.g_attn = g_attn; p.gx = gx;
    p.GS = workspace;
    p.V = p.GS + (size_t)NH * d->B * d->T * d->HW;
    p.Z = p.V + (size_t)d->B * NH * d->C * d->HW;
    p.part_s0 = p.Z + (size_t)d->B * NH * d->C * d->HW;
    p.part_bc = p.part_s0 + tiles_ws * d->T * NH;
    p.part_gb = p.part_bc + tiles_ws * 256;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&ltae_bwd_heads_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&ltae_bwd_gx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(ltae_bwd_heads_kernel, dim3(tiles), dim3(256), bwd1_lds(d), st, p);
    C2S_CHECK_LAUNCH("ltae_bwd_heads");
    hipLaunchKernelGGL(ltae_bwd_gx_kernel, dim3(tiles), dim3(256), bwd2_lds(d), st, p);
    C2S_CHECK_LAUNCH("ltae_bwd_gx");
    // reductions
    const int tpb = (d->HW + PT - 1) / PT;
    {   // gs0[b][t][h] = sum over the tiles of b
        const long total = (long)d->B * d->T * NH;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(total), dim3(64), 0, st, p.part_s0, gs0, tpb,
                           d->T * NH, total);
        C2S_CHECK_LAUNCH("ltae_reduce_s0");
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(256), dim3(64), 0, st, p.part_bc, gbc, (int)tiles, 256, (long)256);
    C2S_CHECK_LAUNCH("ltae_reduce_bc");
    {   // interleaved (dgamma, dbeta) -> two outputs: reduce into a [C][2] scratch then split (reuse part_bc tail)
        float* gb = p.part_bc;   // part_bc is consumed above; 2*C <= 256 floats fit in its first entries
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(2 * d->C), dim3(64), 0, st, p.part_gb, gb, (int)tiles,
                           2 * d->C, (long)2 * d->C);
        C2S_CHECK_LAUNCH("ltae_reduce_gb");
        hipMemcpy2DAsync(ggamma, sizeof(float), gb, 2 * sizeof(float), sizeof(float), d->C, hipMemcpyDeviceToDevice, st);
        hipMemcpy2DAsync(gbeta, sizeof(float), gb + 1, 2 * sizeof(float), sizeof(float), d->C, hipMemcpyDeviceToDevice, st);
    }
    hipLaunchKernelGGL(sum_over_pixels_kernel, dim3(NH * d->C), dim3(64), 0, st, p.V, gU, d->B, NH * d->C, d->HW);
    C2S_CHECK_LAUNCH("ltae_gU");
    if (g_emb != nullptr) {
        hipLaunchKernelGGL(gwc_kernel, dim3(NH * d->C), dim3(256), 0, st, g_emb, p.Z, gWc, d->B, d->C, d->HW);
        C2S_CHECK_LAUNCH("ltae_gWc");
    } else {
        hipMemsetAsync(gWc, 0, (size_t)256 * d->C * sizeof(float), st);
    }
    return C2S_OK;
}

extern "C" int c2s_pixel_gn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int B,
                                int C, int HW, int groups, float eps, void* stream) {
    C2S_REQUIRE(x && gamma && beta && y && stats && groups > 0 && C % groups == 0 && C / groups <= 16, "pixel_gn_fwd: bad args");
    const long total = (long)B * groups * HW;
    hipLaunchKernelGGL(pixel_gn_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                       stats, B, C, HW, groups, eps);
    C2S_CHECK_LAUNCH("pixel_gn_fwd");
    return C2S_OK;
}

extern "C" size_t c2s_pixel_gn_bwd_workspace_floats(int B, int C, int HW) {
    const int chunks = cdiv(HW, 256);
    return (size_t)B * chunks * C * 2 + 2 * (size_t)C;
}

extern "C" int c2s_pixel_gn_bwd(const float* x, const float* gy, const float* gamma, const float* stats, float* gx,
                                float* dgamma, float* dbeta, int B, int C, int HW, int groups, float* workspace,
                                size_t ws_floats, void* stream) {
    C2S_REQUIRE(x && gy && gamma && stats && gx && dgamma && dbeta && workspace, "pixel_gn_bwd: null pointer");
    C2S_REQUIRE(groups > 0 && C % groups == 0 && C / groups <= 16, "pixel_gn_bwd: bad groups");
    C2S_REQUIRE(ws_floats >= c2s_pixel_gn_bwd_workspace_floats(B, C, HW), "pixel_gn_bwd: workspace too small");
    const int chunks = cdiv(HW, 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pixel_gn_bwd_kernel, dim3(B * groups * chunks), dim3(64), 0, st, x, gy, gamma, stats, gx, workspace,
                       B, C, HW, groups, chunks);
    C2S_CHECK_LAUNCH("pixel_gn_bwd");
    float* gb = workspace + (size_t)B * chunks * C * 2;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(2 * C), dim3(64), 0, st, workspace, gb, B * chunks, 2 * C,
                       (long)2 * C);
    C2S_CHECK_LAUNCH("pixel_gn_reduce");
    hipMemcpy2DAsync(dgamma, sizeof(float), gb, 2 * sizeof(float), sizeof(float), C, hipMemcpyDeviceToDevice, st);
    hipMemcpy2DAsync(dbeta, sizeof(float), gb + 1, 2 * sizeof(float), sizeof(float), C, hipMemcpyDeviceToDevice, st);
    return C2S_OK;
}

extern "C" int c2s_dropout_nchw(const float* x, float* y, int B, int C, int HW, float p, uint64_t seed,
                                const uint64_t* seed_dev, const float* keep,
                                void* stream) {
    C2S_REQUIRE(x && y && p >= 0.f && p < 1.f, "dropout: bad args");
    const long total = (long)B * C * HW;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(dropout_nchw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, C, HW, p, seed, seed_dev, keep);
    C2S_CHECK_LAUNCH("dropout");
    return C2S_OK;
}
