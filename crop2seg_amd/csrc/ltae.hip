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

#define C2S_AS1 __attribute__((address_space(1)))
#define C2S_AS3 __attribute__((address_space(3)))

namespace {

#ifdef C2S_LT_STAMP
// diagnostic build only: s_memtime at the phase boundaries of the L-TAE kernels, per workgroup
__device__ unsigned long long lt_stamps[4096 * 8];
__device__ unsigned long long lt_stamps_bwd[4096 * 8];
#define LT_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) lt_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define LT_STAMP_B(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) lt_stamps_bwd[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LT_STAMP(k)
#define LT_STAMP_B(k)
#endif

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
    unsigned long long* keepbits;   // optional [P][16] keep flags as bits (c2s_ltae_desc.keep_bits)
};

// extra outputs of the tiled backward kernels (streaming / register-resident / LDS-resident)
struct StreamBwd {
    float* M;        // [P][16][2]  m1, m2
    float* part_U;   // [tiles][16][C]
    float* gb64;     // [64-pixel tiles][C][2]  d gamma / d beta partials written by the dx kernel (NULL: the heads kernel wrote part_gb)
};

// Attention dropout (tae.py:837).  Explicit keep mask [16,P,T] (tests) or a counter-based RNG: ONE 32-bit avalanche hash per
// pair of time steps (2u, 2u+1) of a (head, pixel) row, 16 bits per element -- drop probability round(p * 2^16) / 2^16 with the
// matching scale, so E[keep * scale] = 1 exactly.  (Two full hashes per element cost 4 quarter-rate v_mul_lo_u32 each:
// 6.6k of the 12k cycles the dropout + store phase of a 16-pixel tile took.)
struct DropCtx {
    uint32_t key, thr;
    float inv;
    int half_t;
};
__device__ __forceinline__ DropCtx drop_ctx(const LtaeParams& p) {
    DropCtx d;
    const uint64_t seed = p.seed + (p.seed_dev != nullptr ? *p.seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    d.key = c2s_hash32((uint32_t)seed ^ c2s_hash32((uint32_t)(seed >> 32) + 0x9E3779B9u));
    d.thr = (uint32_t)(p.drop_p * 65536.f + 0.5f);
    d.inv = 65536.f / (65536.f - (float)d.thr);
    d.half_t = (p.T + 1) >> 1;
    return d;
}
__device__ __forceinline__ uint32_t drop_bits(const DropCtx& d, long row, int u) {
    const uint64_t i2 = (uint64_t)row * (uint64_t)d.half_t + (uint64_t)u;
    return c2s_hash32((uint32_t)i2 ^ d.key ^ (uint32_t)(i2 >> 32) * 0x85EBCA6Bu);
}
__device__ __forceinline__ float drop_pick(const DropCtx& d, uint32_t bits, int t) {
    const uint32_t u16 = (t & 1) ? bits >> 16 : bits & 0xffffu;
    return u16 >= d.thr ? d.inv : 0.f;
}
__device__ __forceinline__ float keep_scale(const LtaeParams& p, int h, long P_total, long pidx, int t) {
    if (p.drop_p <= 0.f) return 1.f;
    if (p.keep != nullptr) return p.keep[((long)h * P_total + pidx) * p.T + t] != 0.f ? 1.f / (1.f - p.drop_p) : 0.f;
    const DropCtx d = drop_ctx(p);
    return drop_pick(d, drop_bits(d, (long)h * P_total + pidx, t >> 1), t);
}


// XCD-aware tile order: workgroups go to the 8 XCDs round-robin, so XCD k gets the k-th contiguous eighth of the tiles (tiles that
// share 128-byte lines sit behind the same L2).  (Visiting the eighth with a stride, so that the tiles running at the same time
// are spread over the [HW] rows instead of a 2 KB window, changed nothing: 1.36 vs 1.37 ms forward, 4.17 vs 4.20 ms backward.)
__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned grid) {
    return (grid & 7) ? bid : (bid & 7) * (grid >> 3) + (bid >> 3);
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

    // ---- phase 1: GroupNorm statistics (padded frames included, tae.py:461): 4 partials per group, each a Chan merge of
    // exact two-pass moments of the cpg values of one time step (error independent of the data, see the streaming kernel)
    {
        const int g = slot & 15, tq = slot >> 4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 mean = zero, m2 = zero;
        float cnt = 0.f;
        const float nb = (float)cpg;
        for (int t = tq; t < T; t += 4) {
            f32x4 v[16];                          // cpg <= 16 (check())
            f32x4 sb = zero;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
                if (cc < cpg) {
                    v[cc] = *reinterpret_cast<const f32x4*>(xq + (size_t)(t * C + g * cpg + cc) * HW);
                    sb += v[cc];
                }
            const f32x4 mb = sb / nb;
            f32x4 qb = zero;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
                if (cc < cpg) {
                    const f32x4 d = v[cc] - mb;
                    qb += d * d;
                }
            const float tot = cnt + nb;
            const f32x4 delta = mb - mean;
            mean += delta * (nb / tot);
            m2 += qb + delta * delta * (cnt * nb / tot);
            cnt = tot;
        }
        // scratch [g][tq][2][16] (mean, M2); the count of partial tq is a function of T
        *reinterpret_cast<f32x4*>(Zl + ((g * 4 + tq) * 2 + 0) * 16 + 4 * q) = mean;
        *reinterpret_cast<f32x4*>(Zl + ((g * 4 + tq) * 2 + 1) * 16 + 4 * q) = m2;
    }
    __syncthreads();
    {
        const int g = hh;       // (pixel, group)
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const float nb = (float)(((T - tq + 3) / 4) * cpg);       // time steps tq, tq+4, ... < T
            if (nb > 0.f) {
                const float mb = Zl[((g * 4 + tq) * 2 + 0) * 16 + px], qb = Zl[((g * 4 + tq) * 2 + 1) * 16 + px];
                const float tot = cnt + nb, delta = mb - mean;
                mean = fmaf(delta, nb / tot, mean);
                m2 += qb + delta * delta * (cnt * nb / tot);
                cnt = tot;
            }
        }
        const float var = fmaxf(m2 / cnt, 0.f);
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

    // ---- phases 4+5, CH channels at a time.  Phase 5 runs on the 16x16x4 MFMA (rows = pixel, columns = j, k = channel,
    // then k = time step for the positional term): wave w owns heads 4w..4w+3, one accumulator each.
    const int l15 = tid & 15, l4 = (tid & 63) >> 4, wv = tid >> 6;
    f32x4 eacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) eacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
        // 5: emb[px][16h+j] += sum_c z[h][c][px] Wc[16h+j][c]; k-step (m, i) lane-quarter l4 <-> channel 16m + 4*l4 + i
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int h = wv * 4 + i4;
            for (int m = 0; m < CH / 16; ++m) {
                const f32x4 wq = *reinterpret_cast<const f32x4*>(p.Wc + (size_t)(h * DV + l15) * C + c0 + 16 * m + 4 * l4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    eacc[i4] = __builtin_amdgcn_mfma_f32_16x16x4f32(Zl[(h * CH + 16 * m + 4 * l4 + i) * 16 + l15], wq[i], eacc[i4], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
        const int h = wv * 4 + i4;
        // + sum_t attn[h,t] pe[t][j]  (k = time step, zero beyond T)
        for (int t0 = 0; t0 < T; t0 += 4) {
            const int t = t0 + l4;
            const float av = Sl[((t < T ? t : 0) * 16 + h) * 16 + l15];
            const float pv = p.pe[(size_t)(b * T + (t < T ? t : 0)) * DV + l15];
            eacc[i4] = __builtin_amdgcn_mfma_f32_16x16x4f32(t < T ? av : 0.f, pv, eacc[i4], 0, 0, 0);
        }
        // D[row = pixel 4*l4 + r][col = j = l15];  + (sum_t attn) * bc[j]
        const float bcv = p.bc[h * DV + l15];
        const f32x4 as4 = *reinterpret_cast<const f32x4*>(ASl + h * 16 + 4 * l4);
        if (pix0 + 4 * l4 < HW)
            *reinterpret_cast<f32x4*>(p.emb + ((size_t)b * NH * DV + h * DV + l15) * HW + pix0 + 4 * l4) = eacc[i4] + as4 * bcv;
    }
}

// ------------------------------------------------------------------------------------------ forward, LDS-resident 4-pixel tiles
// The 16-pixel kernel above launches B*HW/16 workgroups: 64 on the 16 x 16 maps of U-TAE / W-TAE at B = 4 -- a quarter of the
// chip -- and each of them reads its x tile from global memory three times in dependent chunks (134 us at B=4, T=32, C=128:
// latency, not bandwidth).  Here a workgroup owns FOUR adjacent pixels: their whole series x[T][C][4] (64 KB at T=32, C=128) is
// fetched ONCE, by LDS-DMA with every request in flight at the same time, and all phases work on LDS:
//   0 load        global -> LDS rows [t][c] of one float4 (4 pixels); row pitch per time step C+1 (conflict-free across t)
//   1 statistics  exact two-pass (mean, then squared deviations) per (pixel, group) over (C/16 x T), padded frames included
//                 (tae.py:461); thread = (channel, time slice): the group's channels sit in adjacent lanes
//   2 normalise   xhat = gamma (x - mean) rstd + beta, in place
//   3 scores      thread = (t, pair of heads): S[t][h][px] = s0 + sum_c U[h][c] xhat[t][c][px]
//   4 softmax     thread = (px, head, quarter of T): masked softmax over T (tae.py:831), dropout, attn / attn_pre out
//   5 z           thread = (channel, group of heads): z[h][c][px] = sum_t a[h,t,px] xhat[t][c][px]  -> LDS (over x)
//   6 emb         16x16x4 MFMA as in the 16-pixel kernel (rows = pixel: 4 of 16 used), + pe and bias terms
// B*HW/4 workgroups (256 at the bench shape).
template <int C>
__global__ __launch_bounds__(256) void ltae_lds_fwd_kernel(LtaeParams p) {
    extern __shared__ float lds[];
    constexpr int CP = C + 1;
    constexpr int NTH = 256 / C, HPT = NH / NTH, CPG = C / NH;
    const int T = p.T, HW = p.HW;
    float* xs = lds;                              // [T][CP][4]   x -> xhat;  later z [16][C][4]
    float* Sl = xs + (size_t)(T * CP > NH * C ? T * CP : NH * C) * 4;     // [T][16][4]   scores -> post-dropout attention
    float* Ut = Sl + T * 64;                      // [C][16]      U transposed
    float* ABl = Ut + C * 16;                     // [C][2][4]    scale, mean per (channel, pixel);  beta in red[768 ..]
    float* red = ABl + C * 8;                     // [1024]       cross-wave exchanges
    float* ASl = red + 1024;                      // [16][4]      sum_t attention
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);      // the four tiles that share a 64-byte row segment: same L2
    const int tiles_per_b = HW / 4;
    const int b = (int)(tile / tiles_per_b), pix0 = (int)(tile % tiles_per_b) * 4;
    const long Ptot = (long)p.B * HW;
    // ---- 0: every row request of the tile in flight at once (T*C / 256 per lane), no registers involved
    {
        const float* xb = p.x + (size_t)b * T * C * HW + pix0;
        const int nrows = T * C;                  // a multiple of 64: a wave's 64 rows never straddle a time step
        for (int r0 = w * 64; r0 < nrows; r0 += 256) {
            const int t = r0 / C, c0 = r0 - t * C;
            __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(xb + (size_t)(r0 + lane) * HW),
                                             (C2S_AS3 void*)(xs + (size_t)(t * CP + c0) * 4), 16, 0, 0);
        }
        for (int i = tid; i < C * NH; i += 256) Ut[(i % C) * NH + i / C] = p.U[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- 1: statistics.  thread = (channel c, slice th of the time steps); group = CPG adjacent lanes
    const int sc_ = tid % C, sth = tid / C;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto group_total = [&](f32x4 v, float* scratch) -> f32x4 {          // sum over the group's channels and the NTH slices
#pragma unroll
        for (int o = 1; o < CPG; o <<= 1) {
            v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
        }
        if constexpr (NTH == 1) return v;
        if ((sc_ % CPG) == 0) *reinterpret_cast<f32x4*>(scratch + (sth * NH + sc_ / CPG) * 4) = v;
        __syncthreads();
        f32x4 tot = zero4;
#pragma unroll
        for (int k = 0; k < NTH; ++k) tot += *reinterpret_cast<const f32x4*>(scratch + (k * NH + sc_ / CPG) * 4);
        return tot;
    };
    f32x4 mean, rstd;
    {
        f32x4 s1 = zero4;
        for (int t = sth; t < T; t += NTH) s1 += *reinterpret_cast<const f32x4*>(xs + (size_t)(t * CP + sc_) * 4);
        const float inv_n = 1.f / (float)(CPG * T);
        mean = group_total(s1, red) * inv_n;
        f32x4 s2 = zero4;
        for (int t = sth; t < T; t += NTH) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(xs + (size_t)(t * CP + sc_) * 4) - mean;
            s2 += d * d;
        }
        const f32x4 var = group_total(s2, red + 512) * inv_n;
#pragma unroll
        for (int e = 0; e < 4; ++e) rstd[e] = rsqrtf(var[e] + p.eps);
        if (sth == 0) {
            const f32x4 a = rstd * p.gamma[sc_];
            *reinterpret_cast<f32x4*>(ABl + sc_ * 8) = a;
            *reinterpret_cast<f32x4*>(ABl + sc_ * 8 + 4) = mean;
            red[768 + sc_] = p.beta[sc_];
            if ((sc_ % CPG) == 0) {
                const int g = sc_ / CPG;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p.stats[(((long)b * HW + pix0 + e) * NH + g) * 2] = mean[e];
                    p.stats[(((long)b * HW + pix0 + e) * NH + g) * 2 + 1] = rstd[e];
                }
            }
        }
        __syncthreads();
    }
    // ---- 2: normalise in place: xhat = (x - mean) (rstd gamma) + beta -- the mean is subtracted first, not folded into the shift
    // (row = tid + 256 k: the lanes of a wave walk adjacent rows)
    for (int r = tid; r < T * C; r += 256) {
        const int t = r / C, c = r - t * C;
        f32x4* xp = reinterpret_cast<f32x4*>(xs + (size_t)(t * CP + c) * 4);
        *xp = (*xp - *reinterpret_cast<const f32x4*>(ABl + c * 8 + 4)) * *reinterpret_cast<const f32x4*>(ABl + c * 8) + red[768 + c];
    }
    __syncthreads();
    // ---- 3: scores.  thread = (t = tid & 31 (+32), heads 2 hg, 2 hg + 1)
    {
        const int tl = tid & 31, hg = tid >> 5;
        for (int t = tl; t < T; t += 32) {
            const float s00 = p.s0[(b * T + t) * NH + 2 * hg], s01 = p.s0[(b * T + t) * NH + 2 * hg + 1];
            f32x4 a0 = {s00, s00, s00, s00}, a1 = {s01, s01, s01, s01};
            const float* xr = xs + (size_t)t * CP * 4;
#pragma unroll 8
            for (int c = 0; c < C; ++c) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c * 4);
                const float u0 = Ut[c * NH + 2 * hg], u1 = Ut[c * NH + 2 * hg + 1];
                a0 += xv * u0;
                a1 += xv * u1;
            }
            const bool padded = p.valid != nullptr && p.valid[b * T + t] == 0;
            const f32x4 m = {-1e6f, -1e6f, -1e6f, -1e6f};                                      // tae.py:831
            *reinterpret_cast<f32x4*>(Sl + t * 64 + (2 * hg) * 4) = padded ? m : a0;
            *reinterpret_cast<f32x4*>(Sl + t * 64 + (2 * hg + 1) * 4) = padded ? m : a1;
        }
    }
    __syncthreads();
    // ---- 4: softmax over T, dropout, outputs.  thread = (px, head, tq): time steps tq, tq + 4, ...
    {
        const int hp = tid & 63, tq = tid >> 6, h = hp >> 2, px = hp & 3;
        const long pidx = (long)b * HW + pix0 + px;
        float mx = -3.0e38f;
        for (int t = tq; t < T; t += 4) mx = fmaxf(mx, Sl[t * 64 + hp]);
        red[tq * 64 + hp] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[hp], red[64 + hp]), fmaxf(red[128 + hp], red[192 + hp]));
        float den = 0.f;
        for (int t = tq; t < T; t += 4) {
            const float e = __expf(Sl[t * 64 + hp] - mx);
            Sl[t * 64 + hp] = e;
            den += e;
        }
        red[256 + tq * 64 + hp] = den;
        __syncthreads();
        den = (red[256 + hp] + red[320 + hp]) + (red[384 + hp] + red[448 + hp]);
        const float inv_den = 1.f / den;
        float asum = 0.f;
        for (int t = tq; t < T; t += 4) {
            const float a = Sl[t * 64 + hp] * inv_den;
            const float ad = a * keep_scale(p, h, Ptot, pidx, t);
            const size_t o = ((size_t)(h * p.B + b) * T + t) * HW + pix0 + px;
            if (p.attn_pre != nullptr) p.attn_pre[o] = a;
            p.attn[o] = ad;
            Sl[t * 64 + hp] = ad;
            asum += ad;
        }
        red[512 + tq * 64 + hp] = asum;
        __syncthreads();
        if (tq == 0) ASl[hp] = (red[512 + hp] + red[576 + hp]) + (red[640 + hp] + red[704 + hp]);
    }
    if (p.emb == nullptr) return;                 // W-TAE: attention masks only (tae.py:619)
    // ---- 5: z[h][c][px] = sum_t a[h,t,px] xhat[t][c][px].  thread = (channel, HPT heads)
    {
        f32x4 z[HPT];
#pragma unroll
        for (int k = 0; k < HPT; ++k) z[k] = zero4;
#pragma unroll 2
        for (int t = 0; t < T; ++t) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + (size_t)(t * CP + sc_) * 4);
#pragma unroll
            for (int k = 0; k < HPT; ++k) z[k] += *reinterpret_cast<const f32x4*>(Sl + t * 64 + (sth * HPT + k) * 4) * xv;
        }
        __syncthreads();                          // every thread has read its xhat rows: z takes their place
#pragma unroll
        for (int k = 0; k < HPT; ++k) *reinterpret_cast<f32x4*>(xs + (size_t)((sth * HPT + k) * C + sc_) * 4) = z[k];
        __syncthreads();
    }
    // ---- 6: emb[px][16h+j] = sum_c z[h][c][px] Wc[16h+j][c] + sum_t a[h,t,px] pe[t][j] + (sum_t a) bc[16h+j] on the 16x16x4
    // MFMA: rows = pixel (4 of 16 carry data), columns = j, k = channel / time step; wave w owns heads 4w .. 4w+3
    {
        const int l15 = lane & 15, l4 = lane >> 4;
        const bool rowok = l15 < 4;
        const int rl = rowok ? l15 : 0;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int h = w * 4 + i4;
            f32x4 acc = zero4;
            for (int m = 0; m < C / 16; ++m) {
                const f32x4 wq = *reinterpret_cast<const f32x4*>(p.Wc + (size_t)(h * DV + l15) * C + 16 * m + 4 * l4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float zv = xs[(size_t)(h * C + 16 * m + 4 * l4 + i) * 4 + rl];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rowok ? zv : 0.f, wq[i], acc, 0, 0, 0);
                }
            }
            for (int t0 = 0; t0 < T; t0 += 4) {
                const int t = t0 + l4, tc = t < T ? t : 0;
                const float av = Sl[tc * 64 + h * 4 + rl];
                const float pv = p.pe[(size_t)(b * T + tc) * DV + l15];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32((rowok && t < T) ? av : 0.f, pv, acc, 0, 0, 0);
            }
            // D: lane (column j = l15, rows 4 l4 + r): pixels 0..3 are rows 0..3 of the lanes with l4 == 0
            if (l4 == 0) {
                const float bcv = p.bc[h * DV + l15];
                const f32x4 as4 = *reinterpret_cast<const f32x4*>(ASl + h * 4);
                *reinterpret_cast<f32x4*>(p.emb + ((size_t)b * NH * DV + h * DV + l15) * HW + pix0) = acc + as4 * bcv;
            }
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
    LT_STAMP_B(0);

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
    LT_STAMP_B(1);

    if (p.g_emb != nullptr) {
        for (int c0 = 0; c0 < C; c0 += RC) {
            // A1 on the 16x16x4 MFMA (rows = pixel, padded 8 -> 16; k = j; columns = 16 channels): wave w owns heads
            // 4w..4w+3.  (One thread per (pixel, head, channel) with 16 scalar Wc loads each took 57% of this kernel.)
            {
                const int wv = tid >> 6, l15 = tid & 15, l4 = (tid & 63) >> 4;
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const int h = wv * 4 + hh;
#pragma unroll
                    for (int cg = 0; cg < RC / 16; ++cg) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s4 = 0; s4 < DV / 4; ++s4) {
                            const int j = 4 * s4 + l4;
                            const float gv = GEl[(h * DV + j) * PT + (l15 & (PT - 1))];
                            const float a = l15 < PT ? gv : 0.f;
                            const float bw = p.Wc[(size_t)(h * DV + j) * C + c0 + cg * 16 + l15];
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc, 0, 0, 0);
                        }
                        // D[row = 4*l4 + r][col = l15]; rows 0..7 are the pixels of the tile
                        if (l4 < PT / 4) *reinterpret_cast<f32x4*>(Rl + (h * RC + cg * 16 + l15) * PT + 4 * l4) = acc;
                    }
                }
            }
            __syncthreads();
            if (c0 == 0) LT_STAMP_B(5);
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
            if (c0 == 0) LT_STAMP_B(6);
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
            if (c0 == 0) LT_STAMP_B(7);
        }
    }

    LT_STAMP_B(2);
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
    LT_STAMP_B(3);

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
    LT_STAMP_B(4);
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

// ------------------------------------------------------------------------------------------ backward, LDS-resident 4-pixel tiles
// Counterpart of ltae_lds_fwd_kernel: the two kernels above run B*HW/8 workgroups (128 at the bench shape), read x from global
// memory in three dependent passes (dots, V/Z, dx) and hand GS through HBM.  Here a workgroup owns four pixels, keeps their
// normalised series xn[T][C][4] in LDS (one LDS-DMA fetch) next to r[16][C][4] and the attention tiles, and does the whole
// backward of the block:
//   0 load     x, attn, attn_pre, g_attn, g_emb tiles -> LDS (all requests in flight at once); xn = (x - mean) rstd in place
//   1 r        r[h][c][px] = sum_j ge[16h+j][px] Wc[16h+j][c]                        thread = (channel, group of heads)
//   2 dots     dot[h][t] = sum_c r[h][c] (gamma_c xn[t][c] + beta_c)                 thread = (t, pair of heads)
//   3 softmax  ga = (dot + ge.(bc + pe_t) + g_attn) keep;  gs = a' (ga - sum_t a' ga) thread = (px, head, quarter of T)
//              d s0 / d bc partials of the tile
//   4 V, Z     V = sum_t gs xhat (-> d U partial), Z = sum_t attn xhat (-> global, for d Wc)   thread = (channel, heads)
//   5 dx       d xhat[t][c] = sum_h attn r + gs U; GroupNorm backward with the means m1, m2 over (C/16 x T); d gamma / d beta
//              partials.  gamma d xhat is parked in gx (each thread re-reads only what it wrote) until the means are known.
template <int C>
__global__ __launch_bounds__(256) void ltae_lds_bwd_kernel(LtaeParams p, StreamBwd sb) {
    extern __shared__ float lds[];
    constexpr int CP = C + 1;
    constexpr int NTH = 256 / C, HPT = NH / NTH, CPG = C / NH;
    const int T = p.T, HW = p.HW;
    float* xs = lds;                              // [T][CP][4]   x -> xn
    float* Rl = xs + (size_t)T * CP * 4;          // [16][C][4]   r
    float* Al = Rl + NH * C * 4;                  // [T][16][4]   attention (post-dropout)
    float* APl = Al + T * 64;                     // [T][16][4]   attention before dropout
    float* GAl = APl + T * 64;                    // [T][16][4]   upstream gradient of the attention output
    float* Dl = GAl + T * 64;                     // [T][16][4]   dots -> ga -> gs
    float* GEl = Dl + T * 64;                     // [256][4]     tile of g_emb
    float* GBl = GEl + 1024;                      // [C][2]       gamma, beta
    float* STl = GBl + 2 * C;                     // [16][2][4]   mean, rstd
    float* red = STl + 128;                       // [1024]       exchanges
    float* SUMl = red + 1024;                     // [2][16][4]   sum_t attn, sum_t gs
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
    const int tiles_per_b = HW / 4;
    const int b = (int)(tile / tiles_per_b), pix0 = (int)(tile % tiles_per_b) * 4;
    const long Ptot = (long)p.B * HW;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool has_ge = p.g_emb != nullptr;
    // ---- 0
    {
        const float* xb = p.x + (size_t)b * T * C * HW + pix0;
        const int nrows = T * C;
        for (int r0 = w * 64; r0 < nrows; r0 += 256) {
            const int t = r0 / C, c0 = r0 - t * C;
            __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(xb + (size_t)(r0 + lane) * HW),
                                             (C2S_AS3 void*)(xs + (size_t)(t * CP + c0) * 4), 16, 0, 0);
        }
        // attention tiles: row i = (t, h) -> [16][B][T][HW] row (h, b, t); 64 consecutive i per wave request
        const int nat = T * NH;
        for (int i0 = w * 64; i0 < nat; i0 += 256) {
            const int i = i0 + lane;
            if (i < nat) {
                const size_t o = ((size_t)((i & 15) * p.B + b) * T + (i >> 4)) * HW + pix0;
                __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(p.attn_in + o), (C2S_AS3 void*)(Al + i0 * 4), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(p.attn_pre_in + o), (C2S_AS3 void*)(APl + i0 * 4), 16, 0, 0);
                if (p.g_attn != nullptr)
                    __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(p.g_attn + o), (C2S_AS3 void*)(GAl + i0 * 4), 16, 0, 0);
                else
                    *reinterpret_cast<f32x4*>(GAl + i * 4) = zero4;
            }
        }
        if (has_ge)
            __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(p.g_emb + ((size_t)b * NH * DV + tid) * HW + pix0),
                                             (C2S_AS3 void*)(GEl + w * 256), 16, 0, 0);
        else
            *reinterpret_cast<f32x4*>(GEl + tid * 4) = zero4;
        for (int c = tid; c < C; c += 256) { GBl[2 * c] = p.gamma[c]; GBl[2 * c + 1] = p.beta[c]; }
        if (tid < 128) {                          // (g, mean | rstd, px)
            const int g = tid >> 3, k = (tid >> 2) & 1, px = tid & 3;
            STl[tid] = p.stats_in[(((long)b * HW + pix0 + px) * NH + g) * 2 + k];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int r = tid; r < T * C; r += 256) {
        const int t = r / C, c = r - t * C, g = c / CPG;
        f32x4* xp = reinterpret_cast<f32x4*>(xs + (size_t)(t * CP + c) * 4);
        *xp = (*xp - *reinterpret_cast<const f32x4*>(STl + g * 8)) * *reinterpret_cast<const f32x4*>(STl + g * 8 + 4);
    }
    // ---- 1: r (zero without an embedding gradient: W-TAE)
    const int sc_ = tid % C, sth = tid / C;
#pragma unroll
    for (int k = 0; k < HPT; ++k) {
        const int h = sth * HPT + k;
        f32x4 acc = zero4;
        if (has_ge) {
#pragma unroll
            for (int j = 0; j < DV; ++j)
                acc += *reinterpret_cast<const f32x4*>(GEl + (h * DV + j) * 4) * p.Wc[(size_t)(h * DV + j) * C + sc_];
        }
        *reinterpret_cast<f32x4*>(Rl + (size_t)(h * C + sc_) * 4) = acc;
    }
    __syncthreads();
    // ---- 2: dots
    {
        const int tl = tid & 31, hg = tid >> 5;
        for (int t = tl; t < T; t += 32) {
            f32x4 d0 = zero4, d1 = zero4;
            if (has_ge) {
                const float* xr = xs + (size_t)t * CP * 4;
                const float* r0 = Rl + (size_t)(2 * hg) * C * 4;
#pragma unroll 8
                for (int c = 0; c < C; ++c) {
                    const f32x4 xh = *reinterpret_cast<const f32x4*>(xr + c * 4) * GBl[2 * c] + GBl[2 * c + 1];
                    d0 += *reinterpret_cast<const f32x4*>(r0 + c * 4) * xh;
                    d1 += *reinterpret_cast<const f32x4*>(r0 + (C + c) * 4) * xh;
                }
            }
            *reinterpret_cast<f32x4*>(Dl + t * 64 + (2 * hg) * 4) = d0;
            *reinterpret_cast<f32x4*>(Dl + t * 64 + (2 * hg + 1) * 4) = d1;
        }
    }
    __syncthreads();
    // ---- 3: softmax / dropout backward.  thread = (px, head, tq)
    {
        const int hp = tid & 63, tq = tid >> 6, h = hp >> 2, px = hp & 3;
        const long pidx = (long)b * HW + pix0 + px;
        float ge[DV];
        float gebc = 0.f;
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            ge[j] = GEl[(h * DV + j) * 4 + px];
            gebc = fmaf(ge[j], p.bc[h * DV + j], gebc);
        }
        float dsum = 0.f, asum = 0.f;
        for (int t = tq; t < T; t += 4) {
            float gap = Dl[t * 64 + hp] + gebc + GAl[t * 64 + hp];
            const f32x4* pe4 = reinterpret_cast<const f32x4*>(p.pe + (size_t)(b * T + t) * DV);
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                const f32x4 pv = pe4[jq];
                gap = fmaf(ge[4 * jq], pv.x, gap); gap = fmaf(ge[4 * jq + 1], pv.y, gap);
                gap = fmaf(ge[4 * jq + 2], pv.z, gap); gap = fmaf(ge[4 * jq + 3], pv.w, gap);
            }
            const float ga = gap * keep_scale(p, h, Ptot, pidx, t);
            dsum = fmaf(APl[t * 64 + hp], ga, dsum);
            Dl[t * 64 + hp] = ga;
            asum += Al[t * 64 + hp];
        }
        red[tq * 64 + hp] = dsum;
        red[256 + tq * 64 + hp] = asum;
        __syncthreads();
        dsum = (red[hp] + red[64 + hp]) + (red[128 + hp] + red[192 + hp]);
        asum = (red[256 + hp] + red[320 + hp]) + (red[384 + hp] + red[448 + hp]);
        float gssum = 0.f;
        for (int t = tq; t < T; t += 4) {
            const float gs = APl[t * 64 + hp] * (Dl[t * 64 + hp] - dsum);
            Dl[t * 64 + hp] = gs;
            gssum += gs;
            float r = gs;                          // d s0[b,t,h]: sum over the pixels of the tile
            r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64);
            if (px == 0) p.part_s0[((size_t)tile * T + t) * NH + h] = r;
        }
        red[512 + tq * 64 + hp] = gssum;
        __syncthreads();
        if (tq == 0) {
            SUMl[hp] = asum;
            SUMl[64 + hp] = (red[512 + hp] + red[576 + hp]) + (red[640 + hp] + red[704 + hp]);
#pragma unroll
            for (int j = 0; j < DV; ++j) {
                float r = ge[j] * asum;
                r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64);
                if (px == 0) p.part_bc[(size_t)tile * NH * DV + h * DV + j] = r;
            }
        }
        __syncthreads();
    }
    // ---- 4: V = sum_t gs xhat (d U partial of the tile), Z = sum_t attn xhat (global: d Wc)
    const float gm = GBl[2 * sc_], bt = GBl[2 * sc_ + 1];
    {
        f32x4 vr[HPT], zr[HPT];
#pragma unroll
        for (int k = 0; k < HPT; ++k) { vr[k] = zero4; zr[k] = zero4; }
#pragma unroll 2
        for (int t = 0; t < T; ++t) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + (size_t)(t * CP + sc_) * 4);
#pragma unroll
            for (int k = 0; k < HPT; ++k) {
                zr[k] += *reinterpret_cast<const f32x4*>(Al + t * 64 + (sth * HPT + k) * 4) * xv;
                vr[k] += *reinterpret_cast<const f32x4*>(Dl + t * 64 + (sth * HPT + k) * 4) * xv;
            }
        }
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int h = sth * HPT + k;
            const f32x4 zf = zr[k] * gm + *reinterpret_cast<const f32x4*>(SUMl + h * 4) * bt;
            const f32x4 vf = vr[k] * gm + *reinterpret_cast<const f32x4*>(SUMl + 64 + h * 4) * bt;
            if (has_ge) *reinterpret_cast<f32x4*>(p.Z + (((size_t)b * NH + h) * C + sc_) * HW + pix0) = zf;
            sb.part_U[((size_t)tile * NH + h) * C + sc_] = (vf.x + vf.y) + (vf.z + vf.w);
        }
    }
    // ---- 5: d x.  thread = (channel, time slice sth: steps sth, sth + NTH, ...)
    {
        auto group_total = [&](f32x4 v, float* scratch) -> f32x4 {
#pragma unroll
            for (int o = 1; o < CPG; o <<= 1) {
                v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
            }
            if constexpr (NTH == 1) return v;
            if ((sc_ % CPG) == 0) *reinterpret_cast<f32x4*>(scratch + (sth * NH + sc_ / CPG) * 4) = v;
            __syncthreads();
            f32x4 tot = zero4;
#pragma unroll
            for (int k = 0; k < NTH; ++k) tot += *reinterpret_cast<const f32x4*>(scratch + (k * NH + sc_ / CPG) * 4);
            return tot;
        };
        const int g = sc_ / CPG;
        f32x4 r[NH];
        float u[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            r[h] = *reinterpret_cast<const f32x4*>(Rl + (size_t)(h * C + sc_) * 4);
            u[h] = p.U[h * C + sc_];
        }
        float* gxc = p.gx + (size_t)b * T * C * HW + (size_t)sc_ * HW + pix0;
        f32x4 dg = zero4, db = zero4, m1 = zero4, m2 = zero4;
        for (int t = sth; t < T; t += NTH) {
            f32x4 gxh = zero4;
#pragma unroll
            for (int h = 0; h < NH; ++h)
                gxh += *reinterpret_cast<const f32x4*>(Al + t * 64 + h * 4) * r[h] + *reinterpret_cast<const f32x4*>(Dl + t * 64 + h * 4) * u[h];
            const f32x4 xn = *reinterpret_cast<const f32x4*>(xs + (size_t)(t * CP + sc_) * 4);
            dg += gxh * xn;
            db += gxh;
            const f32x4 dxn = gxh * gm;
            m1 += dxn;
            m2 += dxn * xn;
            *reinterpret_cast<f32x4*>(gxc + (size_t)t * C * HW) = dxn;
        }
        const float inv_n = 1.f / (float)(CPG * T);
        __syncthreads();                           // `red` free again (phase 3 read it before its last barrier)
        m1 = group_total(m1, red) * inv_n;
        m2 = group_total(m2, red + 256) * inv_n;
        // d gamma / d beta partial of the tile: sum over its pixels and the time slices
        const float dgs = (dg.x + dg.y) + (dg.z + dg.w), dbs = (db.x + db.y) + (db.z + db.w);
        if constexpr (NTH > 1) {
            red[512 + (sth * C + sc_) * 2] = dgs;
            red[512 + (sth * C + sc_) * 2 + 1] = dbs;
            __syncthreads();
            if (sth == 0) {
                float a = 0.f, bsum = 0.f;
#pragma unroll
                for (int k = 0; k < NTH; ++k) { a += red[512 + (k * C + sc_) * 2]; bsum += red[512 + (k * C + sc_) * 2 + 1]; }
                p.part_gb[((size_t)tile * C + sc_) * 2] = a;
                p.part_gb[((size_t)tile * C + sc_) * 2 + 1] = bsum;
            }
        } else {
            p.part_gb[((size_t)tile * C + sc_) * 2] = dgs;
            p.part_gb[((size_t)tile * C + sc_) * 2 + 1] = dbs;
        }
        const f32x4 rstd = *reinterpret_cast<const f32x4*>(STl + g * 8 + 4);
        for (int t = sth; t < T; t += NTH) {
            f32x4* gp = reinterpret_cast<f32x4*>(gxc + (size_t)t * C * HW);
            const f32x4 xn = *reinterpret_cast<const f32x4*>(xs + (size_t)(t * CP + sc_) * 4);
            *gp = rstd * (*gp - m1 - xn * m2);
        }
    }
}

// ------------------------------------------------------------------------------------------ reductions
// out[grp][k] = sum_{i<count} part[(grp*count + i)*K + k] with lane = k: a wave reads 256-byte pieces of the rows (rounds 1-3: one
// wave per output element, one float per row and lane -- 19x the partials' bytes fetched at the TimeUNet shape: 0.61 GB, 0.22 ms
// in three launches).  The 16 waves of a workgroup take
// the rows i = w, w + 16, ... with 16 requests in flight each and are added in wave order through LDS; more than 512 rows go
// through RR_SLICES double partials per group first.  Fixed order: bitwise reproducible.
constexpr int RR_SLICES = 32;
template <typename TI, typename TO>
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const TI* __restrict__ part, TO* __restrict__ out, TO* __restrict__ out1,
                                                           int count, int K, int slices, int rows_per_slice) {
    __shared__ double sh[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const int grp = blockIdx.y / slices, sl = blockIdx.y - grp * slices;
    const int r0 = sl * rows_per_slice;
    const int n = count - r0 < rows_per_slice ? count - r0 : rows_per_slice;
    const bool live = k < K;
    const TI* src = part + ((size_t)grp * count + r0) * K + (live ? k : 0);
    double s = 0.0;
    for (int i0 = w; i0 < n; i0 += 256) {
        TI v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + 16 * u;
            v[u] = (live && i < n) ? src[(size_t)i * K] : TI(0);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    sh[w][lane] = s;
    __syncthreads();
    if (w == 0 && live) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < 16; ++ww) t += sh[ww][lane];
        if (out1 != nullptr) ((k & 1) ? out1 : out)[k >> 1] = (TO)t;      // interleaved pairs to two arrays (one group)
        else out[(size_t)blockIdx.y * K + k] = (TO)t;
    }
}

// out[grp][k] = sum_{i < count} part[(grp * count + i) * K + k]; tmp: groups * RR_SLICES * K doubles when count > 512
void reduce_rows(const float* part, float* out, int groups, int count, int K, double* tmp, hipStream_t st, float* out1 = nullptr) {
    const int kb = (K + 63) / 64;
    if (count <= 512 || tmp == nullptr) {
        hipLaunchKernelGGL((reduce_rows_kernel<float, float>), dim3(kb, groups), dim3(1024), 0, st, part, out, out1, count, K, 1, count);
    } else {
        const int rps_min = (count + RR_SLICES - 1) / RR_SLICES;
        const int rps = rps_min > 256 ? rps_min : 256;          // 16 rows per wave and batch of requests
        const int slices = (count + rps - 1) / rps;
        hipLaunchKernelGGL((reduce_rows_kernel<float, double>), dim3(kb, groups * slices), dim3(1024), 0, st, part, tmp, (double*)nullptr,
                           count, K, slices, rps);
        hipLaunchKernelGGL((reduce_rows_kernel<double, float>), dim3(kb, groups), dim3(1024), 0, st, (const double*)tmp, out, out1,
                           slices, K, 1, slices);
    }
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

// Large pixel counts: d Wc as 16 small GEMMs on the f32 MFMA (v_mfma_f32_16x16x4_f32), split over pixel slices.
//   part[slice][16h+j][c] = sum_{pixels of the slice} ge[b,16h+j,pix] * Z[b,h,c,pix]        (C = 64)
// Workgroup = (head, slice), 4 waves = the four 16-channel column tiles; the 16 ge rows and 64 Z rows of a 64-pixel
// tile are staged row-major in LDS (row pitch 68 floats: A/B operand reads of the 64 lanes hit 64 distinct banks).
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int GW_PITCH = 68;
__global__ __launch_bounds__(256) void gwc_mfma_kernel(const float* __restrict__ ge, const float* __restrict__ Z,
                                                       float* __restrict__ part, int B, int HW, int tiles_per_slice) {
    constexpr int C = 64;
    __shared__ float tl[(DV + C) * GW_PITCH];
    const int h = blockIdx.x, slice = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tiles_per_b = HW / 64;
    const int ntiles = B * tiles_per_b;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int tile = slice * tiles_per_slice; tile < (slice + 1) * tiles_per_slice && tile < ntiles; ++tile) {
        const int b = tile / tiles_per_b, pix0 = (tile % tiles_per_b) * 64;
        __syncthreads();
        // 80 rows x 16 float4
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int e = tid + i * 256;
            const int row = e >> 4, q = e & 15;
            const float* src = row < DV ? ge + ((size_t)b * NH * DV + h * DV + row) * HW
                                        : Z + (((size_t)b * NH + h) * C + (row - DV)) * HW;
            *reinterpret_cast<f32x4v*>(tl + row * GW_PITCH + 4 * q) = *reinterpret_cast<const f32x4v*>(src + pix0 + 4 * q);
        }
        __syncthreads();
        const float* ap = tl + (lane & 15) * GW_PITCH + (lane >> 4);
        const float* bp = tl + (DV + w * 16 + (lane & 15)) * GW_PITCH + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks], bp[4 * ks], acc, 0, 0, 0);
    }
    // D[row = 4*(lane>>4) + r][col = lane & 15]
#pragma unroll
    for (int r = 0; r < 4; ++r)
        part[((size_t)slice * NH * DV + h * DV + 4 * (lane >> 4) + r) * C + w * 16 + (lane & 15)] = acc[r];
}

// ------------------------------------------------------------------------------------------ parameter fold
// Parameter-only part of the re-association, d_model = 256, n_head = 16, d_k = 4:
//   qWk[h][m] = (1/2) sum_d Q[h][d] Wk[4h+d][m]                    (1/sqrt(d_k) = 1/2)
//   U[h][c]   = sum_m qWk[h][m] Wc[m][c]
//   s0[b,t,h] = sum_j (sum_{m = j mod 16} qWk[h][m]) pe[b,t,j] + sum_m qWk[h][m] bc[m] + (1/2) sum_d Q[h][d] bk[4h+d]
// One workgroup per head for the forward, one workgroup per adjoint stage (a few MFLOP); fixed summation orders.
// Replaces positional_encoding.py:16-33 and the key/query algebra of tae.py:790-830 on the parameter side.
constexpr int DM = NH * DV;      // 256
constexpr int DK = 4;

__global__ void positional_table_kernel(const long long* __restrict__ dates, float* __restrict__ pe, long n, float period) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= n * DV) return;
    const int j = (int)(e % DV);
    const float denom = powf(period, (float)(2 * (j / 2)) / (float)DV);
    const float a = (float)dates[e / DV] / denom;
    pe[e] = (j & 1) ? cosf(a) : sinf(a);
}

// one workgroup per head (a single workgroup spent 47 us in 512 dependent loads per thread)
__global__ __launch_bounds__(1024) void ltae_fold_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Wk,
                                                             const float* __restrict__ bk, const float* __restrict__ Wc,
                                                             const float* __restrict__ bc, const float* __restrict__ pe,
                                                             float* __restrict__ U, float* __restrict__ s0,
                                                             float* __restrict__ qwk, int BT, int C) {
    __shared__ float q[DM];
    __shared__ float qs[DV];
    __shared__ float red[1024];
    __shared__ float k0;
    const int tid = threadIdx.x, h = blockIdx.x;
    if (tid < DM) {
        float v = 0.f;
#pragma unroll
        for (int d = 0; d < DK; ++d) v = fmaf(Q[h * DK + d], Wk[(size_t)(h * DK + d) * DM + tid], v);
        v *= 0.5f;
        q[tid] = v;
        qwk[h * DM + tid] = v;
    }
    __syncthreads();
    if (tid < DV) {
        float v = 0.f;
        for (int i = 0; i < DM / DV; ++i) v += q[i * DV + tid];
        qs[tid] = v;
    }
    if (tid == 64) {
        float v = 0.f;
        for (int m = 0; m < DM; ++m) v = fmaf(q[m], bc[m], v);
        float qb = 0.f;
        for (int d = 0; d < DK; ++d) qb = fmaf(Q[h * DK + d], bk[h * DK + d], qb);
        k0 = v + 0.5f * qb;
    }
    // U[h][c] = sum_m q[m] Wc[m][c]: thread = (c, part); the parts (contiguous m ranges) are summed in order through LDS
    const int parts = 1024 / C, mlen = (DM + parts - 1) / parts;
    {
        const int c = tid % C, part = tid / C;
        float v = 0.f;
        if (part < parts) {
            const int m0 = part * mlen, m1 = min(DM, m0 + mlen);
#pragma unroll 8
            for (int m = m0; m < m1; ++m) v = fmaf(q[m], Wc[(size_t)m * C + c], v);
        }
        red[tid] = v;
    }
    __syncthreads();
    if (tid < C) {
        float v = 0.f;
        for (int part = 0; part < parts; ++part) v += red[part * C + tid];
        U[h * C + tid] = v;
    }
    for (int bt = tid; bt < BT; bt += 1024) {
        float v = k0;
#pragma unroll
        for (int j = 0; j < DV; ++j) v = fmaf(qs[j], pe[(size_t)bt * DV + j], v);
        s0[bt * NH + h] = v;
    }
}

// adjoint: given gU [16][C], gs0 [BT][16] and the attention kernel's direct d Wc / d bc (or NULL), writes the final
// gradients of Q, fc1_k.weight, fc1_k.bias, inconv.weight, inconv.bias (accumulating where acc_* is set).
// Three small launches (a single workgroup doing everything took 0.30 ms: serial, uncoalesced reads of Wc):
//   1  S[h] = sum_bt gs0, P[h][j] = sum_bt gs0 pe         one wave per output, lanes over bt
//   2  per m (one wave each): gq[h][m] = gU[h].Wc[m] + P[h][m%16] + S[h] bc[m]  (lanes over c: coalesced),
//      d inconv.weight[m][:], d inconv.bias[m]
//   3  per (h,d) (one wave each): d fc1_k.weight row, d fc1_k.bias, d Q (lanes over m)
__global__ __launch_bounds__(64) void ltae_fold_bwd1_kernel(const float* __restrict__ gs0, const float* __restrict__ pe,
                                                            float* __restrict__ SP, int BT) {
    const int o = blockIdx.x, lane = threadIdx.x;      // o < 16: S[h]; else P[h][j]
    float v = 0.f;
    if (o < NH) {
        for (int bt = lane; bt < BT; bt += 64) v += gs0[(size_t)bt * NH + o];
    } else {
        const int h = (o - NH) / DV, j = (o - NH) % DV;
        for (int bt = lane; bt < BT; bt += 64) v = fmaf(gs0[(size_t)bt * NH + h], pe[(size_t)bt * DV + j], v);
    }
    v = wave_sum(v);
    if (lane == 0) SP[o] = v;
}

__global__ __launch_bounds__(64) void ltae_fold_bwd2_kernel(const float* __restrict__ Wc, const float* __restrict__ bc,
                                                            const float* __restrict__ qwk, const float* __restrict__ gU,
                                                            const float* __restrict__ SP, const float* __restrict__ gWc_attn,
                                                            const float* __restrict__ gbc_attn, float* __restrict__ gq,
                                                            float* __restrict__ gWc, float* __restrict__ gbc, int C,
                                                            int acc_wc, int acc_bc) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const float* S = SP;
    const float* P = SP + NH;
    float q[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) q[h] = qwk[h * DM + m];
    float dot[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) dot[h] = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float w = Wc[(size_t)m * C + c];
        float v = gWc_attn != nullptr ? gWc_attn[(size_t)m * C + c] : 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float g = gU[h * C + c];
            dot[h] = fmaf(g, w, dot[h]);
            v = fmaf(q[h], g, v);
        }
        gWc[(size_t)m * C + c] = acc_wc ? gWc[(size_t)m * C + c] + v : v;
    }
    float vb = gbc_attn != nullptr ? gbc_attn[m] : 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float d = wave_sum(dot[h]);
        if (lane == 0) gq[h * DM + m] = d + fmaf(S[h], bc[m], P[h * DV + m % DV]);
        vb = fmaf(q[h], S[h], vb);
    }
    if (lane == 0) gbc[m] = acc_bc ? gbc[m] + vb : vb;
}

__global__ __launch_bounds__(64) void ltae_fold_bwd3_kernel(const float* __restrict__ Q, const float* __restrict__ Wk,
                                                            const float* __restrict__ bk, const float* __restrict__ gq,
                                                            const float* __restrict__ SP, float* __restrict__ gQ,
                                                            float* __restrict__ gWk, float* __restrict__ gbk, int acc_q,
                                                            int acc_wk, int acc_bk) {
    const int hd = blockIdx.x, h = hd / DK, lane = threadIdx.x;
    const float qv = Q[hd], S = SP[h];
    float acc = 0.f;
    for (int m = lane; m < DM; m += 64) {
        const float g = gq[h * DM + m];
        const float v = 0.5f * qv * g;
        gWk[(size_t)hd * DM + m] = acc_wk ? gWk[(size_t)hd * DM + m] + v : v;
        acc = fmaf(g, Wk[(size_t)hd * DM + m], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        const float vb = 0.5f * qv * S;
        gbk[hd] = acc_bk ? gbk[hd] + vb : vb;
        const float vq = 0.5f * (acc + bk[hd] * S);
        gQ[hd] = acc_q ? gQ[hd] + vq : vq;
    }
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

// ------------------------------------------------------------------------------------------ streaming forward
// Large pixel counts (TimeUNet runs the L-TAE at full resolution: P = B*128*128, 15.6 KB of x per pixel).
// lane = pixel: every global access is one 256-byte row segment of 64 adjacent pixels; the per-pixel math is
// lane-local and the small weight operands (U^T gamma, Wc) come through the scalar cache as SGPR operands.
// Workgroup = 16 waves on one 64-pixel tile; the role of wave w changes per phase so that nothing needs a
// cross-lane reduction:
//   P1 statistics : wave = GroupNorm group   (its C/16 channels, all t)        -> rstd, -mean*rstd via LDS
//   P2 scores     : wave = time step mod 16  (all channels; 16 head accumulators) -> attn_pre buffer (scratch, L2)
//   P3 softmax    : wave = head              (all t; dropout; sum_t a, sum_t a*pe stay in registers)
//   P4 z          : wave = group             (z_raw[h][c] = sum_t a[h,t] x[t,c], 16 x C/16 accumulators)
//   P5 embedding  : per head through LDS; wave = output row j of the head
// x is read three times (P1, P2, P4); the second and third pass find part of the tile in the Infinity Cache.
// Ut [C][16] = U^T * gamma and cU[16] = U beta are prepared by ltae_prep_kernel.
__global__ void ltae_prep_kernel(const float* __restrict__ U, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 float* __restrict__ Ut, float* __restrict__ cU, int C) {
    const int tid = threadIdx.x;
    for (int e = tid; e < C * NH; e += blockDim.x) {
        const int c = e / NH, h = e % NH;
        Ut[e] = U[h * C + c] * gamma[c];
    }
    if (tid < NH) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(U[tid * C + c], beta[c], s);
        cU[tid] = s;
    }
}


template <int CPG>
__global__ __launch_bounds__(1024) void ltae_stream_fwd_kernel(LtaeParams p, const float* __restrict__ Ut,
                                                                const float* __restrict__ cU) {
    constexpr int C = CPG * NH;
    __shared__ float st[NH][2][64];          // rstd, -mean*rstd per (group, pixel)
    __shared__ float asl[NH][64];            // sum_t attn per (head, pixel)
    __shared__ float apl[NH][DV][64];        // sum_t attn * pe per (head, j, pixel)
    constexpr int ZP = 65;                   // pitch of a z row: the MFMA B-operand reads (16 pixels x channels 16k+ks) hit 64 banks
    __shared__ float zu[4 * C * ZP];         // P4: staged attention chunk [4 t][4][64][4]; P5: z of four heads [4][C][ZP]
    float* ach = zu;
    const int T = p.T, HW = p.HW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tiles_per_b = (HW + 63) / 64;
    const int b = blockIdx.x / tiles_per_b;
    const int pix0 = (blockIdx.x % tiles_per_b) * 64;
    const bool act = pix0 + lane < HW;
    const int pix = act ? pix0 + lane : HW - 1;
    const long pidx = (long)b * HW + pix, Ptot = (long)p.B * HW;
    const float* xb = p.x + (size_t)b * T * C * HW + pix;          // + (t*C + c)*HW

    LT_STAMP(0);
    // ---- P1: GroupNorm statistics of group w (padded frames included, tae.py:461).
    // Exact two-pass moments of every register batch (8 time steps x CPG channels), batches merged with Chan's update
    // (n, mean, M2): the error does not depend on the data.  (A one-pass sum of squares shifted by the first sample lost
    // 1e-5 of the variance wherever that sample sat 4 sigma from the group mean -- with half of the frames zero padding
    // that showed up as 4e-5 on the attention weights.)
    {
        const float* xg = xb + (size_t)(w * CPG) * HW;
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
        // 8 time steps x CPG channels per batch: 32 independent 256-byte loads in flight per wave
        for (int t0 = 0; t0 < T; t0 += 8) {
            float v[8][CPG];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u < T ? t0 + u : T - 1;
#pragma unroll
                for (int cc = 0; cc < CPG; ++cc) v[u][cc] = xg[(size_t)(t * C + cc) * HW];
            }
            const int nt = T - t0 < 8 ? T - t0 : 8;
            const float nb = (float)(nt * CPG);
            float sb = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < nt) {
#pragma unroll
                    for (int cc = 0; cc < CPG; ++cc) sb += v[u][cc];
                }
            const float mb = sb / nb;
            float qb = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < nt) {
#pragma unroll
                    for (int cc = 0; cc < CPG; ++cc) {
                        const float d = v[u][cc] - mb;
                        qb = fmaf(d, d, qb);
                    }
                }
            const float tot = cnt + nb, delta = mb - mean;
            mean = fmaf(delta, nb / tot, mean);
            m2 += qb + delta * delta * (cnt * nb / tot);
            cnt = tot;
        }
        const float var = fmaxf(m2 / cnt, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        if (act) {
            p.stats[(pidx * NH + w) * 2] = mean;
            p.stats[(pidx * NH + w) * 2 + 1] = rstd;
        }
        st[w][0][lane] = rstd;
        st[w][1][lane] = -mean * rstd;
    }
    __syncthreads();

    LT_STAMP(1);
    // ---- P2: scores of time steps t = w, w+16, ...   score[h] = s0 + cU[h] + sum_c Ut[c][h] * (x*rstd - mean*rstd)
    for (int t = w; t < T; t += NH) {
        float sc[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) sc[h] = p.s0[(b * T + t) * NH + h] + cU[h];
        const float* xt = xb + (size_t)t * C * HW;
#pragma unroll 1
        for (int g0 = 0; g0 < NH; g0 += 8) {       // 8 groups = 32 loads in flight per wave (all 64 would not fit 128 registers)
            float d[8][CPG];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int cc = 0; cc < CPG; ++cc) d[u][cc] = xt[(size_t)((g0 + u) * CPG + cc) * HW];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float rs = st[g0 + u][0][lane], nm = st[g0 + u][1][lane];
#pragma unroll
                for (int cc = 0; cc < CPG; ++cc) {
                    const float dn = fmaf(d[u][cc], rs, nm);
#pragma unroll
                    for (int h = 0; h < NH; ++h) sc[h] = fmaf(Ut[((g0 + u) * CPG + cc) * NH + h], dn, sc[h]);
                    // one channel's 16 scalar operands at a time: hoisted together the 32 s_load_dwordx16 of the batch
                    // overflow the SGPR file and come back through v_readlane (3 of them per FMA)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const bool padded = p.valid != nullptr && p.valid[b * T + t] == 0;
        if (act) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
                p.attn_pre[((size_t)(h * p.B + b) * T + t) * HW + pix] = padded ? -1e6f : sc[h];      // tae.py:831
        }
    }
    __threadfence_block();
    __syncthreads();

    LT_STAMP(2);
    // ---- P3: softmax over T for head w, dropout; sum_t a and sum_t a*pe go to LDS for P5
    {
        float asum = 0.f, ape[DV];
#pragma unroll
        for (int j = 0; j < DV; ++j) ape[j] = 0.f;
        float* sp = p.attn_pre + (size_t)(w * p.B + b) * T * HW + pix;
        float* ap = p.attn + (size_t)(w * p.B + b) * T * HW + pix;
        // the T <= 64 scores of this (pixel, head) stay in registers: one pass of loads, 16 in flight at a time
        float sv[64];
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            sv[t] = t < T ? sp[(size_t)t * HW] : -3.0e38f;
        }
#pragma unroll
        for (int t = 0; t < 64; ++t) mx = fmaxf(mx, sv[t]);
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            sv[t] = t < T ? __expf(sv[t] - mx) : 0.f;
            den += sv[t];
        }
        const float inv_den = 1.f / den;
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            if (t < T) {
                const float a = sv[t] * inv_den;
                const float ad = a * keep_scale(p, w, Ptot, pidx, t);
                if (act) {
                    sp[(size_t)t * HW] = a;
                    ap[(size_t)t * HW] = ad;
                }
                asum += ad;
#pragma unroll
                for (int j = 0; j < DV; ++j) ape[j] = fmaf(ad, p.pe[(b * T + t) * DV + j], ape[j]);
            }
        }
        asl[w][lane] = asum;
#pragma unroll
        for (int j = 0; j < DV; ++j) apl[w][j][lane] = ape[j];
    }
    if (p.emb == nullptr) return;            // W-TAE: attention masks only (tae.py:619)
    __threadfence_block();
    __syncthreads();

    LT_STAMP(3);
    // ---- P4: z_raw[h][c] = sum_t attn[h,t] x[t,c] for the channels of group w (padded frames have attn == 0 exactly)
    float z[NH][CPG];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int cc = 0; cc < CPG; ++cc) z[h][cc] = 0.f;
    {
        // attention of 4 time steps at a time through LDS ([t][h/4][pixel][4]: wave w stages head w, every wave reads all
        // heads with ds_read_b128); with the 16 x loads of the chunk that is 20 requests in flight per wave
        constexpr int ZCH = 2;
        const float* xg = xb + (size_t)(w * CPG) * HW;
        const float* aw = p.attn + (size_t)(w * p.B + b) * T * HW + pix;
        float an[ZCH], xn[ZCH][CPG];
        auto issue = [&](int t0) {                 // global loads of one chunk -> registers
#pragma unroll
            for (int tt = 0; tt < ZCH; ++tt) {
                const int t = t0 + tt < T ? t0 + tt : T - 1;
                an[tt] = aw[(size_t)t * HW];
#pragma unroll
                for (int cc = 0; cc < CPG; ++cc) xn[tt][cc] = xg[(size_t)(t * C + cc) * HW];
            }
        };
        issue(0);
        for (int t0 = 0; t0 < T; t0 += ZCH) {
            __syncthreads();
            float xv[ZCH][CPG];
#pragma unroll
            for (int tt = 0; tt < ZCH; ++tt) {
                ach[((tt * 4 + (w >> 2)) * 64 + lane) * 4 + (w & 3)] = t0 + tt < T ? an[tt] : 0.f;
#pragma unroll
                for (int cc = 0; cc < CPG; ++cc) xv[tt][cc] = xn[tt][cc];
            }
            __syncthreads();
            if (t0 + ZCH < T) issue(t0 + ZCH);     // in flight during this chunk's arithmetic (after the barrier: it drains loads)
#pragma unroll
            for (int tt = 0; tt < ZCH; ++tt) {
#pragma unroll
                for (int hq = 0; hq < 4; ++hq) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(ach + ((tt * 4 + hq) * 64 + lane) * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int cc = 0; cc < CPG; ++cc) z[hq * 4 + k][cc] = fmaf(a[k], xv[tt][cc], z[hq * 4 + k][cc]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // GroupNorm affine of group w applied after the t-sum: z = A_c * z_raw + B_c * sum_t a
    float Ac[CPG], Bc[CPG];
    {
        const float rstd = st[w][0][lane], nmr = st[w][1][lane];
#pragma unroll
        for (int cc = 0; cc < CPG; ++cc) {
            const float gm = p.gamma[w * CPG + cc];
            Ac[cc] = gm * rstd;
            Bc[cc] = fmaf(nmr, gm, p.beta[w * CPG + cc]);
        }
    }

    LT_STAMP(4);
    // ---- P5: embedding on the MFMA (v_mfma_f32_16x16x4_f32), four heads per round:
    //   emb[16h+j][px] = sum_c Wc[16h+j][c] z[h][c][px] + (sum_t a) bc[16h+j] + sum_t a pe_t[j]
    // z of the round goes to LDS as [head][c][pixel]; wave w multiplies head w>>2 of the round by the 16-pixel tile w&3:
    // A = Wc rows (i = j, k = channel), B = z (k = channel, n = pixel), 16 k-steps.
    const int mi = lane & 15, mk = lane >> 4;
#pragma unroll
    for (int h0 = 0; h0 < NH; h0 += 4) {
        __syncthreads();                           // previous round (or the last P4 chunk) consumed
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const float ah = asl[h0 + hh][lane];
#pragma unroll
            for (int cc = 0; cc < CPG; ++cc) zu[(hh * C + w * CPG + cc) * ZP + lane] = fmaf(Ac[cc], z[h0 + hh][cc], Bc[cc] * ah);
        }
        __syncthreads();
        if (h0 == 0) LT_STAMP(6);
        const int h = h0 + (w >> 2), nt = w & 3;
        // MFMA k index <-> channel c = 16*mk + ks: lane (j, mk) needs 16 consecutive weights = 4 float4 loads
        const f32x4* wa = reinterpret_cast<const f32x4*>(p.Wc + (size_t)(h * DV + mi) * C + 16 * mk);
        const float* zb = zu + ((w >> 2) * C + 16 * mk) * ZP + nt * 16 + mi;
        f32x4 wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = wa[i];
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks >> 2][ks & 3], zb[ks * ZP], d, 0, 0, 0);
        if (h0 == 0) LT_STAMP(7);
        // D[row j = 4*mk + r][col = pixel nt*16 + mi]
        const int pl = nt * 16 + mi;
        if (pix0 + pl < HW) {
            const float ah = asl[h][pl];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 4 * mk + r;
                p.emb[((size_t)b * NH * DV + h * DV + j) * HW + pix0 + pl] = d[r] + fmaf(ah, p.bc[h * DV + j], apl[h][j][pl]);
            }
        }
    }
    LT_STAMP(5);
}

// ------------------------------------------------------------------------------------------ register-resident forward
// The streaming kernel above reads x three times (statistics, scores, z): a 64-pixel tile is 1 MB and fits nowhere on the
// CU.  A 16-pixel tile is 250 KB -- half of the CU's register file -- so here x is read from HBM ONCE and stays in
// registers; everything else (attention weights, partial sums) moves through LDS.  HBM traffic is the algorithmic
// minimum: x in, attn / attn_pre / emb out.
//
//   workgroup = 8 waves on one tile of 16 adjacent pixels; lane = (px = lane & 15, q = lane >> 4);
//   wave w owns time steps 8w..8w+7; lane q owns channels 16q..16q+15 (GroupNorm groups 4q..4q+3, complete in the lane):
//   x[8 t][16 c] = 128 registers per lane.  A 64-byte row segment per (t, c) row and quarter-wave.
//
//   F1 load        128 independent dword loads per lane
//   F2 statistics  exact two-pass (mean, then squared deviations) with two 8-wave sums through LDS; normalise in place
//   F3 scores      v_mfma_f32_16x16x4: A = U [16 heads x 4 channels], B = xhat [4 channels x 16 pixels] -- the k index of the
//                  B operand is the lane quarter, so step s multiplies channels {s, 16+s, 32+s, 48+s}: 16 MFMAs per time
//                  step, the sum over the four quarters happens inside the MFMA.  D: lane (px, q) gets heads 4q..4q+3.
//   F4 softmax     over T: wave-local over its 8 steps, then max / sum across the 8 waves through LDS; dropout;
//                  attn, attn_pre to HBM; the post-dropout weights also to LDS  adL[t][h][px]
//   F5 pe / sum    ape[h][j][px] = sum_t a pe_t[j],  asum = sum_t a   (thread = (px, h, half of j); pe through the scalar cache)
//   F6 z, emb      8 rounds of 2 heads: every lane accumulates z[h][its 16 channels] over its 8 time steps (VALU), the 8
//                  wave-partials are summed through LDS, and one wave per head multiplies by Wc on the MFMA
//                  (A = Wc rows, B = z [4 channels x 16 pixels]) and stores emb.
constexpr int RPX = 16, RZP = 17;
constexpr int R_TP = 272;                          // adL pitch per time step: 16 px x 16 heads + 16 (the MFMA A reads of 4 steps hit 64 banks)
constexpr int R_XT = 80;                           // xs pitch per time step: 64 c + 16 (the MFMA B reads of 4 steps hit 64 banks)
constexpr int R_XP = 16 * R_XT + 4;                // xs pitch per pixel (= 4 mod 64: the 16-byte stores of 16 lanes hit 64 banks)
constexpr int R_XB = 16 * R_XP;                    // the xs buffer: 16 pixels x 16 time steps
constexpr int R_ZH = 64 * RZP + 4;                 // zT pitch per head (= 4 mod 16: the transposing stores of the 4 lane quarters spread)
constexpr int R_AH = 16 * RZP + 4;                 // apeT pitch per head
constexpr int R_AD = 0;                            // adL [64 t][16 px][16 h] (pitch R_TP);  later zT [16 h][64 c][RZP] (pitch R_ZH)
constexpr int R_XS = R_AD + 16 * R_ZH;             // xs [16 px][16 t][64 c]: a quarter of the time steps; before: red [2][8][16][16]; later apeT
constexpr int R_AS = R_XS + R_XB;                  // asum partials [8 w][16 h][16 px]
constexpr int R_KB = R_AS + 8 * 16 * 16;           // keep flags of the tile as bits [16 h][16 px][2 x 32]
constexpr int R_FLOATS = R_KB + 512;               // 40,576 floats = 162,304 bytes
static_assert(16 * R_ZH >= 64 * R_TP && R_XB >= 16 * R_AH && R_XB >= 2 * 8 * 16 * 16, "LDS map of the register-resident forward");

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global stores (s_waitcnt vmcnt(0)):
// in this kernel no wave reads what another one stored to global memory, so the attention rows written in F4 may stay in
// flight while the next phases run (3.4k cycles of every tile went into that drain).
__device__ __forceinline__ void lds_barrier() {
#ifdef C2S_REG_SYNCTHREADS
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// H32: every dropout counter (row * ceil(T/2) + pair) fits in 32 bits (host check): the hash then needs no 64-bit index arithmetic
// and no multiply of the (zero) high word -- the same bits as the general form (196 quarter-rate integer instructions per wave
// were 11 % of this kernel's issue time).
template <bool H32>
__global__ __launch_bounds__(512) void ltae_reg_fwd_kernel(LtaeParams p) {
    extern __shared__ float lds[];
    float* adL = lds + R_AD;
    float* xs = lds + R_XS;
    float* asp = lds + R_AS;
    float* red = xs;                               // the small cross-wave reductions are over before xs is used
    constexpr int C = 64;
    const int T = p.T, HW = p.HW;
    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave index as a scalar: row addresses become SGPR bases
    // XCD-aware tile order: workgroups go to the 8 XCDs round-robin, so give each XCD a contiguous eighth of the tiles --
    // the two 16-pixel tiles that share every 128-byte line of x then sit behind the same L2.
    // (A persistent tile loop was tried: the loop-carried state pushed the kernel from 234 to 256 VGPRs + 65 spills, 1.39 -> 2.07 ms.)
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
    const int tiles_per_b = HW / RPX;
    const int b = (int)(tile / tiles_per_b);
    const int pix = (int)(tile % tiles_per_b) * RPX + px;
    const long pidx = (long)b * HW + pix, Ptot = (long)p.B * HW;
    const float* xb = p.x + (size_t)b * T * C * HW;                       // uniform base; 32-bit offsets below (< 2^30 floats)
    const unsigned xoff = (unsigned)(16 * q) * (unsigned)HW + (unsigned)pix;
    const int t0 = 8 * w;
    const int nt = T - t0 < 0 ? 0 : (T - t0 < 8 ? T - t0 : 8);          // time steps of this wave that exist
    unsigned* kbL = reinterpret_cast<unsigned*>(lds + R_KB);
    kbL[tid] = 0u;                                 // (512 words; the first workgroup barrier comes long before F4 ORs into them)
    LT_STAMP(0);
    // ---- F1
    float x[8][16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int tc = t0 + i < T ? t0 + i : T - 1;
#pragma unroll
#ifdef C2S_LT_COMPUTEONLY
        for (int s = 0; s < 16; ++s) x[i][s] = __builtin_bit_cast(float, (xoff + (unsigned)(tc * C + s) * 2654435761u) & 0x3fffffffu | 0x30000000u);
#else
        for (int s = 0; s < 16; ++s) x[i][s] = (xb + (size_t)(tc * C + s) * HW)[xoff];        // uniform row pointer + per-lane offset
#endif
    }
#ifdef C2S_LT_LOADONLY
    {   // diagnostic build: the load phase alone
        float a_ = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int s = 0; s < 16; ++s) a_ += x[i][s];
        if (a_ == 12345.678f) p.stats[0] = a_;
        return;
    }
#endif
    // ---- F2: GroupNorm statistics over (4 channels x T), padded frames included (tae.py:461): exact two-pass moments of the
    // wave's own 4 nt values per group, then ONE exchange: every wave merges the 8 (count, mean, M2) partials with Chan's update
    // in the same order (identical results in all waves, independent of the data).
    {
        float* red2 = red + 8 * 16 * 16;
        float mean[4], rs[4];
        const float inv_cw = nt > 0 ? __frcp_rn((float)(4 * nt)) : 0.f;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < nt) s1 += (x[i][4 * gg] + x[i][4 * gg + 1]) + (x[i][4 * gg + 2] + x[i][4 * gg + 3]);
            const float mw = s1 * inv_cw;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < nt) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const float d = x[i][4 * gg + cc] - mw;
                        s2 = fmaf(d, d, s2);
                    }
                }
            red[(w * 16 + 4 * q + gg) * 16 + px] = mw;
            red2[(w * 16 + 4 * q + gg) * 16 + px] = s2;
        }
        lds_barrier();
        // merge weights of partial ww (functions of T only): r1 = n_b / (n + n_b), r2 = n n_b / (n + n_b)
        float r1[8], r2[8];
        {
            float cnt = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                const int ntw = T - 8 * ww < 0 ? 0 : (T - 8 * ww < 8 ? T - 8 * ww : 8);
                const float nb = (float)(4 * ntw), tot = cnt + nb;
                const float inv = tot > 0.f ? __frcp_rn(tot) : 0.f;
                r1[ww] = nb * inv;
                r2[ww] = cnt * nb * inv;
                cnt = tot;
            }
        }
        const float inv_n = __frcp_rn((float)(4 * T));
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            float mean_ = 0.f, m2 = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                const float mb = red[(ww * 16 + 4 * q + gg) * 16 + px], qb = red2[(ww * 16 + 4 * q + gg) * 16 + px];
                const float delta = mb - mean_;          // an empty partial (T <= 8 ww) has r1 = r2 = 0 and qb = 0
                mean_ = fmaf(delta, r1[ww], mean_);
                m2 += fmaf(delta * delta, r2[ww], qb);
            }
            mean[gg] = mean_;
            rs[gg] = rsqrtf(fmaf(m2, inv_n, p.eps));
            if (w == 0) {
                p.stats[(pidx * NH + 4 * q + gg) * 2] = mean[gg];
                p.stats[(pidx * NH + 4 * q + gg) * 2 + 1] = rs[gg];
            }
        }
        // normalise in place: xhat = (x - mean) * (rstd * gamma) + beta
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = rs[s >> 2] * p.gamma[16 * q + s], bt = p.beta[16 * q + s], m = mean[s >> 2];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i][s] = fmaf(x[i][s] - m, a, bt);
        }
    }
    LT_STAMP(1);
    // ---- F3: scores on the MFMA
    // operands of the softmax phase, requested here so that their latency hides behind the MFMAs (earlier they would hold 40
    // registers through the statistics): s0[b][t][4q..4q+3] (one 16-byte load per time step) and the per-frame padding flags
    f32x4 s0v[8];
    int vflag[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int tc = t0 + i < T ? t0 + i : T - 1;
        s0v[i] = *reinterpret_cast<const f32x4*>(p.s0 + (size_t)(b * T + tc) * NH + 4 * q);
        vflag[i] = p.valid != nullptr ? p.valid[b * T + tc] : 1;
    }
    float sc[8][4];
    {
        float Ua[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) Ua[s] = p.U[px * C + 16 * q + s];      // A[i = head px][k = q] of step s
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ua[s], x[i][s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[i][r] = acc[r];                   // head 4q + r, pixel px
        }
    }
    LT_STAMP(2);
    // ---- F4: softmax over T (masked, tae.py:831), dropout, outputs
    {
        float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = vflag[i] == 0 ? -1e6f : sc[i][r] + s0v[i][r];
                    sc[i][r] = v;
                    mx[r] = fmaxf(mx[r], v);
                }
            }
        // local softmax pieces of the wave's 8 steps, then ONE exchange: M = max_w m_w, S = sum_w s_w exp(m_w - M)
        float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(sc[i][r] - mx[r]);
                    sc[i][r] = e;
                    sm[r] += e;
                }
            }
        float* red2 = red + 8 * 16 * 16;
        lds_barrier();                             // the statistics partials in `red` have been read by every wave
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[(w * 16 + 4 * q + r) * 16 + px] = mx[r];
            red2[(w * 16 + 4 * q + r) * 16 + px] = sm[r];
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mw[8], m = -3.0e38f, d = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                mw[ww] = red[(ww * 16 + 4 * q + r) * 16 + px];
                m = fmaxf(m, mw[ww]);
            }
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) d = fmaf(red2[(ww * 16 + 4 * q + r) * 16 + px], __expf(mw[ww] - m), d);
            sm[r] = __expf(mx[r] - m) / d;         // a = exp(v - m_w) * exp(m_w - M) / S
        }
        // output rows: uniform base (b, t) + per-lane offset (head, pixel); 32-bit offsets (the host checks 16*B*T*HW < 2^30)
        unsigned hoff[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hoff[r] = (unsigned)(4 * q + r) * (unsigned)(p.B * T) * (unsigned)HW + (unsigned)pix;
        const bool rng = p.drop_p > 0.f && p.keep == nullptr;
        DropCtx dc = {};
        if (rng) dc = drop_ctx(p);
        // H32: counter of (head 4q + r, pixel, pair u) = ((4q) Ptot + pidx) half_t + t0 / 2  +  r (Ptot half_t)  +  u
        const uint32_t k32 = (uint32_t)Ptot * (uint32_t)dc.half_t;
        const uint32_t rb32 = ((uint32_t)(4 * q) * (uint32_t)Ptot + (uint32_t)pidx) * (uint32_t)dc.half_t + (uint32_t)(t0 >> 1);
        // head by head (4 hashes live at a time): t0 is even, so steps (2u, 2u+1) of a row share one hash
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = 4 * q + r;
            uint32_t bits[4] = {0u, 0u, 0u, 0u};
            if (rng) {
                if constexpr (H32) {
                    const uint32_t rb = rb32 + (uint32_t)r * k32;
#pragma unroll
                    for (int u = 0; u < 4; ++u) bits[u] = c2s_hash32((rb + (uint32_t)u) ^ dc.key);
                } else {
                    const uint64_t rb = (uint64_t)((long)h * Ptot + pidx) * (uint64_t)dc.half_t + (uint64_t)(t0 >> 1);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint64_t i2 = rb + (uint64_t)u;
                        bits[u] = c2s_hash32((uint32_t)i2 ^ dc.key ^ (uint32_t)(i2 >> 32) * 0x85EBCA6Bu);
                    }
                }
            }
            float as = 0.f;
            unsigned kept = 0u;                    // bit i: step t0 + i was kept
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float ad = 0.f;                    // steps T..63 carry zero weight
                if (i < nt) {
                    const int t = t0 + i;
                    const float a = sc[i][r] * sm[r];
                    const float ks = rng ? drop_pick(dc, bits[i >> 1], t) : keep_scale(p, h, Ptot, pidx, t);
                    kept |= (ks != 0.f ? 1u : 0u) << i;
                    ad = a * ks;
                    if (p.attn_pre != nullptr) (p.attn_pre + (size_t)(b * T + t) * HW)[hoff[r]] = a;
                    if (p.attn != nullptr) (p.attn + (size_t)(b * T + t) * HW)[hoff[r]] = ad;    // NULL: nobody reads the post-dropout
                                                                                                 // weights (TimeUNet without return_att)
                    as += ad;
                }
                sc[i][r] = ad;
            }
            asp[(w * 16 + h) * 16 + px] = as;      // sum_t a, per wave (summed in the epilogue)
            atomicOr(&kbL[(h * 16 + px) * 2 + (w >> 2)], kept << (8 * (w & 3)));     // (unconditional: no branch inside the phase)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<f32x4*>(adL + (t0 + i) * R_TP + px * 16 + 4 * q) = (f32x4){sc[i][0], sc[i][1], sc[i][2], sc[i][3]};
    }
    if (p.emb == nullptr) return;                  // attention masks only (tae.py:619)
    lds_barrier();
    LT_STAMP(3);
    if (p.keepbits != nullptr && tid < 256) {      // thread = (px, h): the 16 words of a pixel are 128 contiguous bytes
        const int kh = tid & 15, kp = tid >> 4;
        const unsigned long long word = (unsigned long long)kbL[(kh * 16 + kp) * 2] | ((unsigned long long)kbL[(kh * 16 + kp) * 2 + 1] << 32);
        p.keepbits[((size_t)b * HW + (size_t)(pix - px + kp)) * NH + kh] = word;
    }

    // ---- F5: z[h][c] = sum_t a[h][t] xhat[t][c] PER PIXEL on the MFMA:  A = a_px [16 h x 4 t] from adL,  B = xhat_px [4 t x 16 c].
    // The B operand of one pixel is spread over lanes and waves, so xhat goes through LDS, 16 time steps (two waves' registers,
    // full-wave 16-byte stores) at a time; every wave multiplies two pixels x four channel blocks per quarter and keeps the
    // eight accumulators.  D: lane (n = c - 16 cb, rows h = 4q + r).
    // (Tried: staging by channel quarter with 16 active lanes per store, 24.8k cycles per tile; one wave staging its 8 steps into a
    // second buffer while the others multiply, one barrier per chunk, 19.8k; this form 16.3k.)
    const int hl = px;                             // the lane's row index of an A operand / column index of a B operand
    f32x4 zacc[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) { zacc[cb][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; zacc[cb][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float pb[16];                                  // B operand of the pe product and the operands of the embedding GEMM: requested
    f32x4 wa[2][4], bcv[2];                        // after the last quarter is staged (x is dead then), used after the loop
    const float* ap0 = adL + q * R_TP + (2 * w) * 16 + hl;               // A operand of pixel 2w: + t R_TP; pixel 2w+1: + 16
    const float* bp0 = xs + (2 * w) * R_XP + q * R_XT + hl;              // B operand of pixel 2w: + 4 s R_XT + 16 cb; pixel 2w+1: + R_XP
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
        if ((w >> 1) == tq) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    *reinterpret_cast<f32x4*>(xs + px * R_XP + (8 * (w & 1) + i) * R_XT + 16 * q + 4 * u) =
                        (f32x4){x[i][4 * u], x[i][4 * u + 1], x[i][4 * u + 2], x[i][4 * u + 3]};
        }
        if (tq == 3) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int t = 4 * s + q;
                pb[s] = t < T ? p.pe[(size_t)(b * T + t) * DV + hl] : 0.f;
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4* wr = reinterpret_cast<const f32x4*>(p.Wc + (size_t)((2 * w + hh) * DV + px) * C + 16 * q);
#pragma unroll
                for (int u = 0; u < 4; ++u) wa[hh][u] = wr[u];
                bcv[hh] = *reinterpret_cast<const f32x4*>(p.bc + (2 * w + hh) * DV + 4 * q);
            }
        }
        lds_barrier();
        // operands first (8 A + 32 B values), then the 32 MFMAs: 8 independent chains of 4
        float av[2][4], bv[2][4][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                av[u][s] = ap0[(16 * tq + 4 * s) * R_TP + 16 * u];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) bv[u][cb][s] = bp0[4 * s * R_XT + 16 * cb + R_XP * u];
            }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    zacc[cb][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s], bv[u][cb][s], zacc[cb][u], 0, 0, 0);
        lds_barrier();
    }
    LT_STAMP(4);
    // the same with B = pe[b][t][j] (shared by all pixels):  ape[h][j] = sum_t a[h][t] pe_t[j];   D: lane (n = j, rows h = 4q + r)
    f32x4 pacc[2];
    {
        float av[2][16];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 16; ++s) av[u][s] = ap0[4 * s * R_TP + 16 * u];        // a[h][t = 4s + q] of pixel 2w + u
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s], pb[s], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s + 1], pb[s + 1], d1, 0, 0, 0);
            }
            pacc[u] = d0 + d1;
        }
    }
    lds_barrier();                               // every MFMA above has read adL: zT takes its place, apeT that of xs
    float* zT = lds + R_AD;
    float* apeT = lds + R_XS;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) zT[(4 * q + r) * R_ZH + (16 * ch + hl) * RZP + 2 * w + u] = zacc[ch][u][r];
            apeT[(4 * q + r) * R_AH + hl * RZP + 2 * w + u] = pacc[u][r];
        }
    }
    lds_barrier();
    LT_STAMP(5);
    // ---- F6: emb[16h+j][px] = sum_c Wc[16h+j][c] z[h][c][px] + (sum_t a) bc[16h+j] + ape[h][j][px];  step s multiplies channels
    // 16k + s (k = lane quarter):  A[i = j][k] = Wc[16h + j][16k + s],  B[k][n = px] = zT[h][16k + s][px]
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int hm = 2 * w + hh;
        const float* zb = zT + hm * R_ZH + (16 * q) * RZP + px;
        f32x4 dd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dd[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) dd[s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[hh][s >> 2][s & 3], zb[s * RZP], dd[s & 3], 0, 0, 0);
        float as = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) as += asp[(ww * 16 + hm) * 16 + px];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dsum = (dd[0][r] + dd[1][r]) + (dd[2][r] + dd[3][r]);
            p.emb[((size_t)b * NH * DV + hm * DV + 4 * q + r) * HW + pix] = dsum + fmaf(as, bcv[hh][r], apeT[hm * R_AH + (4 * q + r) * RZP + px]);
        }
    }
    LT_STAMP(6);
}

// ------------------------------------------------------------------------------------------ streaming backward
// Same idea as the streaming forward, 32-pixel tiles: lane = (pixel, half), 16 waves per workgroup, the role of a wave
// changes per phase.  Two kernels (x is read three times in total):
//  heads kernel
//   S1 wave = head      r[h][c] = sum_j ge[16h+j] Wc[16h+j][c] -> LDS (128 KB, float4 of 4 channels per pixel);
//                       c0[h][t] = ge.(bc + pe_t) + g_attn -> GS buffer; sum_t attn; d bc partials
//   A  wave = t mod 16  half = channel half: dot[h] = sum_c r[h][c] xhat[t][c] (r: ds_read_b128), halves added with one
//                       cross-lane move; ga = (dot + c0) * keep -> GS buffer
//   B  wave = head      half = t parity: softmax backward gs = a' (ga - sum_t a' ga) -> GS buffer; d s0 partials
//   C  wave = group     half = channel pair: V_raw = sum_t gs x, Z_raw = sum_t attn x (attn / gs of 4 time steps at a
//                       time staged in LDS as [t][h/4][pixel][4]); GroupNorm-backward means m1, m2 per (pixel, group)
//                       in closed form from r, U, V, Z (no pass over t); Z -> global (d Wc), d U / d gamma / d beta partials
//  gx kernel
//   X  wave = group     half = channel pair: d xhat[t][c] = sum_h attn[h,t] r[h][c] + gs[h,t] U[h][c];
//                       gx = rstd (gamma d xhat - m1 - xn m2)
constexpr int SPT = 32;      // pixels per streaming backward tile
constexpr int SCH = 4;       // time steps per staged attn / gs chunk

typedef float f32x2s __attribute__((ext_vector_type(2)));


__device__ __forceinline__ float half_sum32(float v) {      // sum over the 32 lanes of this half of the wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// attn and gs of time steps [t0, t0+SCH) for the tile's pixels, staged as LDS [arr 2][SCH][h/4][32 px][4]:
// 2 arrays x SCH x 16 heads = 128 rows of 32 pixels; wave w owns rows [8w, 8w+8), 4 per half.  Split into the global
// loads (issued a chunk ahead, under the previous chunk's arithmetic) and the LDS writes.
__device__ __forceinline__ void chunk_load(const LtaeParams& p, float (&v)[4], int b, int pix, int t0, int w, int hf,
                                           const float* gs_src) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = w * 8 + hf * 4 + i;
        const int arr = row >> 6, rem = row & 63;
        const int tt = rem >> 4, h = rem & 15;
        const int t = t0 + tt < p.T ? t0 + tt : p.T - 1;
        const size_t o = ((size_t)(h * p.B + b) * p.T + t) * p.HW + pix;
        v[i] = arr == 0 ? p.attn_in[o] : gs_src[o];
    }
}
__device__ __forceinline__ void chunk_store(float* buf, const float (&v)[4], int w, int px, int hf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = w * 8 + hf * 4 + i;
        const int arr = row >> 6, rem = row & 63;
        const int tt = rem >> 4, h = rem & 15;
        buf[(((arr * SCH + tt) * 4 + (h >> 2)) * SPT + px) * 4 + (h & 3)] = v[i];
    }
}

template <int CPG>
__global__ __launch_bounds__(1024) void ltae_stream_bwd_heads_kernel(LtaeParams p, StreamBwd sb) {
    constexpr int C = CPG * NH;
    static_assert(CPG == 4, "lane halves own channel pairs of a 4-channel group");
    extern __shared__ float lds[];
    float* rl = lds;                               // [16 h][C/4][32 px][4]        128 KB
    float* stl = rl + NH * C * SPT;                // [16 g][2][32]  rstd, -mean*rstd
    float* asl = stl + NH * 2 * SPT;               // [16 h][32]     sum_t attn
    float* gsl = asl + NH * SPT;                   // [16 h][32]     sum_t gs
    float* chk = gsl + NH * SPT;                   // [2][SCH][4][32][4]  staged attn / gs chunk   16 KB
    float* pel = chk + 2 * SCH * 4 * SPT * 4;      // [T][16]  positional table of this batch element
    const int T = p.T, HW = p.HW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = lane & 31, hf = lane >> 5;
    const int tiles_per_b = (HW + SPT - 1) / SPT;
    const int b = blockIdx.x / tiles_per_b;
    const int pix0 = (blockIdx.x % tiles_per_b) * SPT;
    const bool act = pix0 + px < HW;
    const int pix = act ? pix0 + px : HW - 1;
    const long pidx = (long)b * HW + pix, Ptot = (long)p.B * HW;
    const float* xb = p.x + (size_t)b * T * C * HW + pix;

    LT_STAMP_B(0);
    // ---- S1: wave = head
    for (int e = threadIdx.x; e < T * DV; e += 1024) pel[e] = p.pe[(size_t)b * T * DV + e];
    __syncthreads();
    {
        const int h = w;
        if (hf == 0) {
            stl[(h * 2 + 0) * SPT + px] = p.stats_in[(pidx * NH + h) * 2 + 1];
            stl[(h * 2 + 1) * SPT + px] = -p.stats_in[(pidx * NH + h) * 2] * p.stats_in[(pidx * NH + h) * 2 + 1];
        }
        float ge[DV];
#pragma unroll
        for (int j = 0; j < DV; ++j) ge[j] = p.g_emb[((size_t)b * NH * DV + h * DV + j) * HW + pix];
        // r[h][c][px] = sum_j Wc[16h+j][c] ge[16h+j][px] on the MFMA (v_mfma_f32_16x16x4_f32): per head a
        // [64 c x 16 j] x [16 j x 32 px] product = 4 channel tiles x 2 pixel tiles x 4 k-steps.  D[c = 16mt + 4(l>>4) + r]
        // [px = 16nt + (l&15)] is exactly one float4 of the LDS layout [h][c/4][px][4].  (The VALU version with scalar
        // weight operands spent 57k cycles here, mostly waiting for 256 s_load_dwordx4 per wave.)
        {
            const int mi = lane & 15, mk = lane >> 4;
            float gb_[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int pl = pix0 + nt * 16 + mi;
                    gb_[ks][nt] = p.g_emb[((size_t)b * NH * DV + h * DV + 4 * ks + mk) * HW + (pl < HW ? pl : HW - 1)];
                }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float wa_[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wa_[ks] = p.Wc[(size_t)(h * DV + 4 * ks + mk) * C + mt * 16 + mi];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_[ks], gb_[ks][nt], d, 0, 0, 0);
                    *reinterpret_cast<f32x4*>(rl + ((size_t)(h * (C / 4) + mt * 4 + mk) * SPT + nt * 16 + mi) * 4) = d;
                }
            }
        }
        float gebc = 0.f;
#pragma unroll
        for (int j = 0; j < DV; ++j) gebc = fmaf(ge[j], p.bc[h * DV + j], gebc);
        float asum = 0.f;
        for (int i0 = 0; i0 < 32; i0 += 8) {           // t = 2i + hf, 8 time steps per batch of loads
            float av[8], gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = 2 * (i0 + u) + hf < T ? 2 * (i0 + u) + hf : T - 1;
                const size_t o = ((size_t)(h * p.B + b) * T + t) * HW + pix;
                av[u] = p.attn_in[o];
                gv[u] = p.g_attn != nullptr ? p.g_attn[o] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = 2 * (i0 + u) + hf;
                if (t < T) {
                    float c0 = gebc + gv[u];
#pragma unroll
                    for (int jq = 0; jq < DV / 4; ++jq) {           // wave-uniform address: LDS broadcast reads
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(pel + t * DV + 4 * jq);
                        c0 = fmaf(ge[4 * jq + 0], pv[0], c0); c0 = fmaf(ge[4 * jq + 1], pv[1], c0);
                        c0 = fmaf(ge[4 * jq + 2], pv[2], c0); c0 = fmaf(ge[4 * jq + 3], pv[3], c0);
                    }
                    asum += av[u];
                    if (act) p.GS[((size_t)(h * p.B + b) * T + t) * HW + pix] = c0;
                }
            }
        }
        asum += __shfl_xor(asum, 32, 64);
        if (hf == 0) asl[h * SPT + px] = asum;
        // d bc[16h+j] = sum_px ge[j] * sum_t attn      (half 0 reduces j < 8, half 1 the rest)
#pragma unroll
        for (int jj = 0; jj < DV / 2; ++jj) {
            const int j = hf * (DV / 2) + jj;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < DV; ++k) v = k == j ? ge[k] : v;
            v = half_sum32(act ? v * asum : 0.f);
            if (px == 0) p.part_bc[(size_t)blockIdx.x * NH * DV + h * DV + j] = v;
        }
    }
    __threadfence_block();
    __syncthreads();

    LT_STAMP_B(1);
    // ---- A: wave = time steps t = w, w+16, ...; half = channels [32 hf, 32 hf + 32)
    for (int t = w; t < T; t += NH) {
        const float* xt = xb + (size_t)(t * C + hf * (C / 2)) * HW;
        float xh[C / 2];
#pragma unroll
        for (int i = 0; i < C / 2; ++i) xh[i] = xt[(size_t)i * HW];
#pragma unroll
        for (int i = 0; i < C / 2; ++i) {
            const int c = hf * (C / 2) + i, g = c / CPG;
            const float dn = fmaf(xh[i], stl[(g * 2 + 0) * SPT + px], stl[(g * 2 + 1) * SPT + px]);
            const float gm = hf ? p.gamma[C / 2 + i] : p.gamma[i], bt = hf ? p.beta[C / 2 + i] : p.beta[i];   // scalar loads + select
            xh[i] = fmaf(dn, gm, bt);
        }
#pragma unroll 1
        for (int h = 0; h < NH; ++h) {
            const float* rr = rl + ((size_t)(h * (C / 4) + hf * (C / 8)) * SPT + px) * 4;
            float dot = 0.f;
#pragma unroll
            for (int i4 = 0; i4 < C / 8; ++i4) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(rr + (size_t)i4 * SPT * 4);
                dot = fmaf(r[0], xh[4 * i4 + 0], dot); dot = fmaf(r[1], xh[4 * i4 + 1], dot);
                dot = fmaf(r[2], xh[4 * i4 + 2], dot); dot = fmaf(r[3], xh[4 * i4 + 3], dot);
            }
            dot += __shfl_xor(dot, 32, 64);
            const size_t o = ((size_t)(h * p.B + b) * T + t) * HW + pix;
            const float ga = (dot + p.GS[o]) * keep_scale(p, h, Ptot, pidx, t);
            if (act && hf == 0) p.GS[o] = ga;
        }
    }
    __threadfence_block();
    __syncthreads();

    LT_STAMP_B(2);
    // ---- B: wave = head; half = t parity
    {
        const int h = w;
        const size_t o0 = (size_t)(h * p.B + b) * T * HW + pix;
        // all a' and ga of this (pixel, head, t parity) in registers: one round of loads instead of 31 dependent ones
        float ap[32], ga[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int t = 2 * i + hf < T ? 2 * i + hf : T - 1;
            ap[i] = p.attn_pre_in[o0 + (size_t)t * HW];
            ga[i] = p.GS[o0 + (size_t)t * HW];
        }
        float dsum = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) dsum = 2 * i + hf < T ? fmaf(ap[i], ga[i], dsum) : dsum;
        dsum += __shfl_xor(dsum, 32, 64);
        float gssum = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int t = 2 * i + hf;
            if (t < T) {
                const float gs = ap[i] * (ga[i] - dsum);
                if (act) p.GS[o0 + (size_t)t * HW] = gs;
                gssum += gs;
                const float r = half_sum32(act ? gs : 0.f);            // d s0[b,t,h]: sum over the pixels of the tile
                if (px == 0) p.part_s0[((size_t)blockIdx.x * T + t) * NH + h] = r;
            }
        }
        gssum += __shfl_xor(gssum, 32, 64);
        if (hf == 0) gsl[h * SPT + px] = gssum;
    }
    __threadfence_block();
    __syncthreads();

    LT_STAMP_B(3);
    // ---- C: wave = group g; half = channel pair (c0, c0+1) = 4g + 2hf
    {
        const int g = w, c0 = g * CPG + 2 * hf;
        float Vr[NH][2], Zr[NH][2];
#pragma unroll
        for (int h = 0; h < NH; ++h) { Vr[h][0] = Vr[h][1] = Zr[h][0] = Zr[h][1] = 0.f; }
        const float* xg = xb + (size_t)c0 * HW;
        float sv[4], xn[SCH][2];
        auto issue = [&](int t0) {                               // global loads of one chunk -> registers
            chunk_load(p, sv, b, pix, t0, w, hf, p.GS);
#pragma unroll
            for (int tt = 0; tt < SCH; ++tt) {
                const int t = t0 + tt < T ? t0 + tt : T - 1;
                xn[tt][0] = xg[(size_t)(t * C) * HW];
                xn[tt][1] = xg[(size_t)(t * C + 1) * HW];
            }
        };
        issue(0);
        for (int t0 = 0; t0 < T; t0 += SCH) {
            __syncthreads();                                     // previous chunk consumed
            chunk_store(chk, sv, w, px, hf);
            float xv[SCH][2];
#pragma unroll
            for (int tt = 0; tt < SCH; ++tt) { xv[tt][0] = xn[tt][0]; xv[tt][1] = xn[tt][1]; }
            __syncthreads();
            if (t0 + SCH < T) issue(t0 + SCH);                   // in flight during this chunk's arithmetic (after the barrier:
                                                                 // __syncthreads() waits for outstanding loads)
#pragma unroll
            for (int tt = 0; tt < SCH; ++tt) {
                if (t0 + tt < T) {
#pragma unroll
                    for (int hq = 0; hq < 4; ++hq) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(chk + (((0 * SCH + tt) * 4 + hq) * SPT + px) * 4);
                        const f32x4 gs = *reinterpret_cast<const f32x4*>(chk + (((1 * SCH + tt) * 4 + hq) * SPT + px) * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            Zr[hq * 4 + k][0] = fmaf(a[k], xv[tt][0], Zr[hq * 4 + k][0]);
                            Zr[hq * 4 + k][1] = fmaf(a[k], xv[tt][1], Zr[hq * 4 + k][1]);
                            Vr[hq * 4 + k][0] = fmaf(gs[k], xv[tt][0], Vr[hq * 4 + k][0]);
                            Vr[hq * 4 + k][1] = fmaf(gs[k], xv[tt][1], Vr[hq * 4 + k][1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);   // two ds_read_b128 at a time (all 32 hoisted = 128 registers)
                    }
                }
            }
        }
        LT_STAMP_B(6);
        // normalised sums: sum_t w xn = rstd * W_raw - mean rstd * sum_t w
        const float rs = stl[(g * 2 + 0) * SPT + px], nm = stl[(g * 2 + 1) * SPT + px];
        float gmk[2], btk[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            gmk[k] = hf ? p.gamma[g * CPG + 2 + k] : p.gamma[g * CPG + k];      // scalar loads + select
            btk[k] = hf ? p.beta[g * CPG + 2 + k] : p.beta[g * CPG + k];
        }
        float dg[2] = {0.f, 0.f}, db[2] = {0.f, 0.f};
        // Z part first (consumes Z_raw and r), then the V part (consumes V_raw, writes V over r in LDS): two short loops keep
        // fewer values live than one, and the LDS stores of the second do not interleave with the r loads of the first
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float as = asl[h * SPT + px];
            const f32x2s r2 = *reinterpret_cast<const f32x2s*>(rl + ((size_t)(h * (C / 4) + g) * SPT + px) * 4 + 2 * hf);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float zt = fmaf(rs, Zr[h][k], nm * as);
                dg[k] = fmaf(r2[k], zt, dg[k]);
                db[k] = fmaf(r2[k], as, db[k]);
                if (act) p.Z[(((size_t)b * NH + h) * C + c0 + k) * HW + pix] = fmaf(gmk[k], zt, btk[k] * as);   // for d Wc
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float gss = gsl[h * SPT + px];
            f32x2s v2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float vt = fmaf(rs, Vr[h][k], nm * gss);
                const float u = hf ? p.U[h * C + g * CPG + 2 + k] : p.U[h * C + g * CPG + k];
                dg[k] = fmaf(u, vt, dg[k]);
                db[k] = fmaf(u, gss, db[k]);
                v2[k] = act ? fmaf(gmk[k], vt, btk[k] * gss) : 0.f;           // V (xhat-based) for d U
            }
            *reinterpret_cast<f32x2s*>(rl + ((size_t)(h * (C / 4) + g) * SPT + px) * 4 + 2 * hf) = v2;   // only this lane reads the slot
            __builtin_amdgcn_sched_barrier(0);
        }
        LT_STAMP_B(7);
        // d gamma / d beta partials of the tile; GroupNorm-backward means of the group
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = c0 + k;
            const float gm = hf ? p.gamma[g * CPG + 2 + k] : p.gamma[g * CPG + k];
            m1 = fmaf(gm, db[k], m1);
            m2 = fmaf(gm, dg[k], m2);
            const float sg = half_sum32(act ? dg[k] : 0.f), sbt = half_sum32(act ? db[k] : 0.f);
            if (px == 0) {
                p.part_gb[((size_t)blockIdx.x * C + c) * 2 + 0] = sg;
                p.part_gb[((size_t)blockIdx.x * C + c) * 2 + 1] = sbt;
            }
        }
        m1 += __shfl_xor(m1, 32, 64);
        m2 += __shfl_xor(m2, 32, 64);
        const float inv_n = 1.f / (float)(CPG * T);
        if (act && hf == 0) {
            sb.M[(pidx * NH + g) * 2 + 0] = m1 * inv_n;
            sb.M[(pidx * NH + g) * 2 + 1] = m2 * inv_n;
        }
    }
    LT_STAMP_B(4);
    __syncthreads();
    // d U partial of the tile: thread = (head, channel), fixed-order sum over the 32 pixels (rotated start: no bank conflicts)
    {
        const int h = threadIdx.x >> 6, c = threadIdx.x & 63;
        const float* vp = rl + (size_t)(h * (C / 4) + (c >> 2)) * SPT * 4 + (c & 3);
        float sum = 0.f;
#pragma unroll 8
        for (int i = 0; i < SPT; ++i) sum += vp[((i + (c >> 2)) & (SPT - 1)) * 4];
        sb.part_U[((size_t)blockIdx.x * NH + h) * C + c] = sum;
    }
    LT_STAMP_B(5);
}

template <int CPG>
__global__ __launch_bounds__(1024) void ltae_stream_bwd_gx_kernel(LtaeParams p, StreamBwd sb) {
    constexpr int C = CPG * NH;
    extern __shared__ float lds[];
    float* rl = lds;                                          // [16 h][C/4][32 px][4]  r, as in the heads kernel (128 KB)
    float* chk = rl + NH * C * SPT;                           // staged attn / gs chunk
    const int T = p.T, HW = p.HW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = lane & 31, hf = lane >> 5;
    const int tiles_per_b = (HW + SPT - 1) / SPT;
    const int b = blockIdx.x / tiles_per_b;
    const int pix0 = (blockIdx.x % tiles_per_b) * SPT;
    const bool act = pix0 + px < HW;
    const int pix = act ? pix0 + px : HW - 1;
    const long pidx = (long)b * HW + pix;
    const int g = w, c0 = g * CPG + 2 * hf;
    // r[h][c][px] = sum_j Wc[16h+j][c] ge[16h+j][px] on the MFMA, wave = head (same block as in the heads kernel; the
    // VALU form with scalar weight operands waits on ~1000 scalar loads per wave)
    {
        const int h = w, mi = lane & 15, mk = lane >> 4;
        float gb_[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int pl = pix0 + nt * 16 + mi;
                gb_[ks][nt] = p.g_emb[((size_t)b * NH * DV + h * DV + 4 * ks + mk) * HW + (pl < HW ? pl : HW - 1)];
            }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float wa_[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wa_[ks] = p.Wc[(size_t)(h * DV + 4 * ks + mk) * C + mt * 16 + mi];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_[ks], gb_[ks][nt], d, 0, 0, 0);
                *reinterpret_cast<f32x4*>(rl + ((size_t)(h * (C / 4) + mt * 4 + mk) * SPT + nt * 16 + mi) * 4) = d;
            }
        }
    }
    __syncthreads();
    // r and U of the two channels of this lane (wave = group from here on)
    float r[NH][2], u[NH][2];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const f32x2s r2 = *reinterpret_cast<const f32x2s*>(rl + ((size_t)(h * (C / 4) + g) * SPT + px) * 4 + 2 * hf);
        r[h][0] = r2[0];
        r[h][1] = r2[1];
        u[h][0] = hf ? p.U[h * C + g * CPG + 2] : p.U[h * C + g * CPG + 0];
        u[h][1] = hf ? p.U[h * C + g * CPG + 3] : p.U[h * C + g * CPG + 1];
    }
    const float mean = p.stats_in[(pidx * NH + g) * 2], rstd = p.stats_in[(pidx * NH + g) * 2 + 1];
    const float m1 = sb.M[(pidx * NH + g) * 2 + 0], m2 = sb.M[(pidx * NH + g) * 2 + 1];
    const float gm0 = (hf ? p.gamma[g * CPG + 2] : p.gamma[g * CPG]) * rstd, gm1 = (hf ? p.gamma[g * CPG + 3] : p.gamma[g * CPG + 1]) * rstd;
    const float* xg = p.x + (size_t)b * T * C * HW + (size_t)c0 * HW + pix;
    float* gxg = p.gx + (size_t)b * T * C * HW + (size_t)c0 * HW + pix;
    float sv[4], xn[SCH][2];
    auto issue = [&](int t0) {
        chunk_load(p, sv, b, pix, t0, w, hf, p.GS);
#pragma unroll
        for (int tt = 0; tt < SCH; ++tt) {
            const int t = t0 + tt < T ? t0 + tt : T - 1;
            xn[tt][0] = xg[(size_t)(t * C) * HW];
            xn[tt][1] = xg[(size_t)(t * C + 1) * HW];
        }
    };
    issue(0);
    for (int t0 = 0; t0 < T; t0 += SCH) {
        __syncthreads();
        chunk_store(chk, sv, w, px, hf);
        float xv[SCH][2];
#pragma unroll
        for (int tt = 0; tt < SCH; ++tt) { xv[tt][0] = xn[tt][0]; xv[tt][1] = xn[tt][1]; }
        __syncthreads();
        if (t0 + SCH < T) issue(t0 + SCH);
#pragma unroll
        for (int tt = 0; tt < SCH; ++tt) {
            const int t = t0 + tt;
            if (t < T) {
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int hq = 0; hq < 4; ++hq) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(chk + (((0 * SCH + tt) * 4 + hq) * SPT + px) * 4);
                    const f32x4 gs = *reinterpret_cast<const f32x4*>(chk + (((1 * SCH + tt) * 4 + hq) * SPT + px) * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int h = hq * 4 + k;
                        d0 = fmaf(a[k], r[h][0], d0); d0 = fmaf(gs[k], u[h][0], d0);
                        d1 = fmaf(a[k], r[h][1], d1); d1 = fmaf(gs[k], u[h][1], d1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (act) {
                    const float xn0 = (xv[tt][0] - mean) * rstd, xn1 = (xv[tt][1] - mean) * rstd;
                    gxg[(size_t)(t * C) * HW] = gm0 * d0 - rstd * fmaf(xn0, m2, m1);
                    gxg[(size_t)(t * C + 1) * HW] = gm1 * d1 - rstd * fmaf(xn1, m2, m1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ register-resident backward, heads part
// Backward of the attention block up to the dscores, in the layout of the register-resident forward (16-pixel tiles, lane =
// (px, q), wave w owns time steps 8w..8w+7, xhat[8 t][16 c] = the GroupNorm output in registers).  Every tensor is read ONCE;
// the streaming heads kernel above reads x twice and re-reads r (128 KB of LDS) for every time step (its dots phase is
// LDS-bound: 137k of its 414k cycles per 32-pixel tile).  Phases, in program order:
//
//   P0  small operands: tile of g_emb, pe, bc -> LDS; g_attn -> ga; Wc operands; the first half of x is requested here
//   H2  r[h][c] = sum_j ge[16h+j] Wc[16h+j][c]           MFMA, two heads per wave; rows permuted so that lane (px, q) receives
//                                                         its own 16 channels -> rL[h][px][c] (LDS)
//   c0  c0[h][t] = ge_h . (bc_h + pe_t) + g_attn          all operands from LDS, under the x stream; second half of x requested
//   H1  statistics of the forward -> xhat = gamma xn + beta, in place
//   H3  dot[h][t] = sum_c r[h][c] xhat[t][c]              packed FMAs; reduce-scatter over the four lane quarters
//   H4  ga = (dot + c0) * keep;  gs = a (ga - sum_t a ga) keep read off attn / attn_pre; one cross-wave exchange; attn -> aL
//                                                         (LDS operand of the Z pass), gs -> GS (global, for the dx kernel)
//   H5z Zt[h][c] = sum_t attn xhat                        per pixel on the MFMA, xhat staged through LDS as in the forward (F5);
//                                                         sum_t attn per (h, px) from aL on the way
//   H6a r again; m1 and the Z part of m2 (GroupNorm-backward means, closed forms from r, Z, asum); Z -> global, transposed
//   H5v Vt[h][c] = sum_t gs xhat                          second MFMA pass (the accumulators of both do not fit beside x);
//                                                         d s0 partial from aL on the way
//   H6b d U partial of the tile, the V part of m2, d bc partial
constexpr int RB_RP = 68;                          // rL pitch per (head, pixel): 64 c + 4
constexpr int RB_A = 0;                            // aL [64 t][16 px][16 h] (pitch R_TP); before: exchange scratch + geL; between the passes: pU; later zT
constexpr int RB_GE = RB_A + 4096;                 // geL [16 h][16 j][16 px] (head pitch RB_GH): the tile of g_emb, until aL is filled
constexpr int RB_GH = 16 * 16 + 4;                 // = 4 mod 16: the four lane quarters (heads 4q + r) read 64 banks
constexpr int RB_XS = RB_A + 16 * R_ZH;            // xs [16 px][16 t][64 c] (pitch R_XT / R_XP); before and after: rL [16 h][16 px][RB_RP], peL [T][16]
constexpr int RB_AS = RB_XS + R_XB;                // asL [16 h][16 px]  sum_t attn;  before: bcL [256]
constexpr int RB_FLOATS = RB_AS + 256;             // 38,272 floats = 153,088 bytes
static_assert(R_XB >= 16 * 16 * RB_RP + 2048, "rL and peL / the H4 exchange fit the staging area");
static_assert(RB_GE + 16 * RB_GH <= RB_XS && 8 * 1024 <= 16 * R_ZH, "geL / pU fit the aL area");

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bstore(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}

// REKEEP: the forward did not store the post-dropout weights (attn_in == NULL) but the keep flags as bits (p.keepbits [P][16]
// words, bit t): one float tensor less to read.
template <bool REKEEP>
__global__ __launch_bounds__(512) void ltae_reg_bwd_heads_kernel(LtaeParams p, StreamBwd sb) {
    extern __shared__ float lds[];
    float* aL = lds + RB_A;
    float* red = lds + RB_XS + 16 * 16 * RB_RP;    // cross-wave exchange of H4 [8 w][16 h][16 px] (peL is dead by then)
    float* geL = lds + RB_GE;
    float* xs = lds + RB_XS;
    float* rL = lds + RB_XS;
    float* peL = lds + RB_XS + 16 * 16 * RB_RP;
    float* asL = lds + RB_AS;
    float* bcL = lds + RB_AS;
    constexpr int C = 64;
    const int T = p.T, HW = p.HW;
    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);                              // XCD-aware order (partials: by tile)
    const int tiles_per_b = HW / RPX;
    const int b = (int)(tile / tiles_per_b);
    const int pix0 = (int)(tile % tiles_per_b) * RPX, pix = pix0 + px;
    const long pidx = (long)b * HW + pix;
    const int t0 = 8 * w;
    const int nt = T - t0 < 0 ? 0 : (T - t0 < 8 ? T - t0 : 8);
    // Global addressing through buffer descriptors: a wave-uniform row offset in an SGPR + ONE per-lane byte offset per tensor.
    // (Flat 64-bit per-lane addresses for the ~400 unrolled accesses of this kernel cost two registers per access in flight;
    // the first version of this kernel carried 2.6 KB of scratch per lane and ran at 10.6 ms.)  The host checks the sizes < 2^31.
    const unsigned rowb = (unsigned)HW * 4u;                                              // bytes per [HW] row
    const unsigned abytes = 16u * (unsigned)(p.B * T) * rowb;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * T * C * HW), 0, (int)((unsigned)(T * C) * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rap = __builtin_amdgcn_make_buffer_rsrc((void*)p.attn_pre_in, 0, (int)abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rad = __builtin_amdgcn_make_buffer_rsrc((void*)(REKEEP ? p.attn_pre_in : p.attn_in), 0, (int)abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rga = __builtin_amdgcn_make_buffer_rsrc((void*)(p.g_attn != nullptr ? p.g_attn : p.attn_pre_in), 0, (int)abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rgs = __builtin_amdgcn_make_buffer_rsrc((void*)p.GS, 0, (int)abytes, 0x00020000);
    const float gat_w = p.g_attn != nullptr ? 1.f : 0.f;   // no upstream gradient of the attention output: weight 0
    const unsigned xvo = ((unsigned)(16 * q) * (unsigned)HW + (unsigned)pix) * 4u;
    // attention tensors [16][B][T][HW]: per-lane offset of head 4q + pixel; head 4q + r and the row (b, t) go into the SGPR offset
    const unsigned headb = (unsigned)(p.B * T) * rowb;
    const unsigned hv0 = (unsigned)(4 * q) * headb + (unsigned)pix * 4u;
    auto trow = [&](int i, int r) {                // byte offset of head r, row (b, t0 + i) clamped to the last time step: an SGPR
        const int tc = t0 + i < T ? t0 + i : T - 1;
        return (unsigned)(b * T + tc) * rowb + (unsigned)r * headb;
    };
    const float* geb = p.g_emb + (size_t)b * NH * DV * HW + pix0;                          // [256][HW] rows of the tile
    const int chan = 16 * (px >> 2) + (px & 3);    // + 4 cb: the permuted MFMA rows of H2

    // dropout scale (branch-free and up here: a basic-block boundary in the middle of the kernel lets LLVM sink the c0 sums of P0
    // below the dots, with every operand spilled on the way)
    const bool dropping = p.drop_p > 0.f;
    const float thr16 = (float)(uint32_t)(p.drop_p * 65536.f + 0.5f);
    const float kscale = !dropping ? 1.f : (p.keep != nullptr ? 1.f / (1.f - p.drop_p) : 65536.f / (65536.f - thr16));

    LT_STAMP_B(0);
    // ---- P0 .. H1, ordered for the load queue (a wave has 64 loads in flight at most and issues in order): the small operands
    // first, then half of x, then -- while x streams -- the LDS fill, H2 and half of c0, the other half of x, the rest of c0.
    // Small operands: the tile of g_emb, pe and bc (for LDS); upstream gradient of the attention output; Wc operands of H2.
    float gav[8][4];
    f32x4 gt[2];
    float pev[2], wA[2][4][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + 512 * k;             // (row, quarter of the 16 pixels)
        gt[k] = *reinterpret_cast<const f32x4*>(geb + (size_t)(idx >> 2) * HW + 4 * (idx & 3));
        pev[k] = idx < T * DV ? p.pe[(size_t)b * T * DV + idx] : 0.f;
    }
    const float bcv = tid < NH * DV ? p.bc[tid] : 0.f;
    // H2 computes r of heads 2w, 2w+1.  MFMA rows are permuted: row i of channel block cb is channel 16 (i >> 2) + 4 cb + (i & 3), so
    // that D (lane (px, q): rows 4q..4q+3) holds channels 16q + 4cb + r -- the lane's own.  A[i][k] = Wc[16h + 4s + k][chan(i)],
    // B[k][n = px] = ge[16h + 4s + k][px].
    auto load_wA = [&]() {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) wA[hh][cb][s] = p.Wc[((2 * w + hh) * DV + 4 * s + q) * C + 4 * cb + chan];
    };
    auto r_mma = [&](const float (&gb)[2][4]) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[hh][cb][s], gb[hh][s], acc, 0, 0, 0);
                *reinterpret_cast<f32x4*>(rL + ((2 * w + hh) * 16 + px) * RB_RP + 16 * q + 4 * cb) = acc;
            }
    };
    load_wA();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) gav[i][r] = gat_w * bload(rga, hv0, trow(i, r));
    __builtin_amdgcn_sched_barrier(0);
    f32x2 x[8][8];
    auto load_x = [&](int i) {
        const int tc = t0 + i < T ? t0 + i : T - 1;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#ifdef C2S_LT_COMPUTEONLY
            x[i][s][0] = __builtin_bit_cast(float, (xvo + (unsigned)(tc * C + 2 * s) * 2654435761u) & 0x3fffffffu | 0x30000000u);
            x[i][s][1] = __builtin_bit_cast(float, (xvo + (unsigned)(tc * C + 2 * s + 1) * 2654435761u) & 0x3fffffffu | 0x30000000u);
#else
            x[i][s][0] = bload(rx, xvo, (unsigned)(tc * C + 2 * s) * rowb);
            x[i][s][1] = bload(rx, xvo, (unsigned)(tc * C + 2 * s + 1) * rowb);
#endif
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) load_x(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + 512 * k, row = idx >> 2;
        *reinterpret_cast<f32x4*>(geL + (row >> 4) * RB_GH + (row & 15) * 16 + 4 * (idx & 3)) = gt[k];
        if (idx < 64 * DV) peL[idx] = pev[k];
    }
    if (tid < NH * DV) bcL[tid] = bcv;
    lds_barrier();                                 // geL, peL, bcL written
    LT_STAMP_B(1);
    {
        float gb[2][4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int s = 0; s < 4; ++s) gb[hh][s] = geL[(2 * w + hh) * RB_GH + (4 * s + q) * 16 + px];
        r_mma(gb);
    }
    __builtin_amdgcn_sched_barrier(0);
    LT_STAMP_B(2);
    // c0[h][t] = ge_h . (bc_h + pe_t) + g_attn[h][t] for the lane's heads 4q..4q+3, all operands from LDS (no global load
    // behind the x stream); the dots are added in H3
    auto c0_head = [&](int r) {
        const int h = 4 * q + r;
        float gev[16];
        float gebc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) gev[j] = geL[h * RB_GH + j * 16 + px];
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            const f32x4 bc4 = *reinterpret_cast<const f32x4*>(bcL + h * DV + 4 * jq);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) gebc = fmaf(gev[4 * jq + jj], bc4[jj], gebc);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tc = t0 + i < T ? t0 + i : T - 1;
            float c0 = gav[i][r] + gebc;
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(peL + tc * DV + 4 * jq);
                c0 = fmaf(gev[4 * jq + 0], pv[0], c0); c0 = fmaf(gev[4 * jq + 1], pv[1], c0);
                c0 = fmaf(gev[4 * jq + 2], pv[2], c0); c0 = fmaf(gev[4 * jq + 3], pv[3], c0);
            }
            asm volatile("" : "+v"(c0));           // computed HERE (see the note on code sinking at kscale)
            gav[i][r] = c0;
        }
    };
    c0_head(0);
    __builtin_amdgcn_sched_barrier(0);
    c0_head(1);
    __builtin_amdgcn_sched_barrier(0);
    // the other half of x
#pragma unroll
    for (int i = 4; i < 8; ++i) load_x(i);
    __builtin_amdgcn_sched_barrier(0);
    c0_head(2);
    __builtin_amdgcn_sched_barrier(0);
    c0_head(3);
    __builtin_amdgcn_sched_barrier(0);
    float na[16], nc[16];                          // xhat = x * na + nc  (na = rstd gamma, nc = beta - mean rstd gamma)
    {
        const f32x4 st0 = *reinterpret_cast<const f32x4*>(p.stats_in + (pidx * NH + 4 * q) * 2);
        const f32x4 st1 = *reinterpret_cast<const f32x4*>(p.stats_in + (pidx * NH + 4 * q) * 2 + 4);
        const float mean[4] = {st0[0], st0[2], st1[0], st1[2]}, rs[4] = {st0[1], st0[3], st1[1], st1[3]};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + 16 * q + 4 * u);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.beta + 16 * q + 4 * u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                na[4 * u + e] = rs[u] * g4[e];
                nc[4 * u + e] = fmaf(-mean[u], na[4 * u + e], b4[e]);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    LT_STAMP_B(3);
    // xhat = gamma xn + beta in registers: the dots, Z = sum_t attn xhat (for d Wc) and V = sum_t gs xhat (= d U) need no further
    // gamma / beta; the GroupNorm-backward means follow from Z, V, r, U in closed form (H6); d gamma / d beta come from the dx kernel
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x[i][s][0] = i < nt ? fmaf(x[i][s][0], na[2 * s], nc[2 * s]) : 0.f;            // steps T..63: zero
            x[i][s][1] = i < nt ? fmaf(x[i][s][1], na[2 * s + 1], nc[2 * s + 1]) : 0.f;
        }
    lds_barrier();                                 // rL complete; geL, peL, bcL consumed
    LT_STAMP_B(4);
    // ---- H3: dots dot[h][t] = sum_c r[h][c] xhat[t][c].  Every lane sums its 16 channels for all 16 heads (packed FMAs); the sum
    // over the four channel quarters is a reduce-scatter: head h = 4Q + j belongs to lane quarter Q, so for a given j the four
    // partials of a lane go to the four quarters -- one exchange across the wave halves (Q >> 1), one across the quarter pairs.
    {
        const bool Hh = (q >> 1) != 0, Pp = (q & 1) != 0;
        auto head_dot = [&](int h, float (&out)[8]) {
            f32x2 rv[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rL + (h * 16 + px) * RB_RP + 16 * q + 4 * u);
                rv[2 * u] = (f32x2){v[0], v[1]}; rv[2 * u + 1] = (f32x2){v[2], v[3]};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x2 acc = {0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = __builtin_elementwise_fma(rv[s], x[i][s], acc);
                out[i] = acc[0] + acc[1];
            }
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float keepA[8], keepB[8];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float dlo[8], dhi[8];
                __builtin_amdgcn_sched_barrier(0);
                head_dot(4 * pr + j, dlo);         // owner quarter Q = pr      (lower half of the wave)
                __builtin_amdgcn_sched_barrier(0);
                head_dot(4 * (pr + 2) + j, dhi);   // owner quarter Q = pr + 2  (upper half)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float send = Hh ? dlo[i] : dhi[i], mine = Hh ? dhi[i] : dlo[i];
                    float v = mine + __shfl_xor(send, 32, 64);
                    asm volatile("" : "+v"(v));    // the exchange happens HERE: left alone, instruction selection defers all 96
                                                   // of them to the end of H3, with every partial sum kept (and spilled)
                    if (pr == 0) keepA[i] = v; else keepB[i] = v;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {          // keepA: quarter 2H, keepB: quarter 2H + 1
                const float send = Pp ? keepA[i] : keepB[i], mine = Pp ? keepB[i] : keepA[i];
                float v = gav[i][j] + (mine + __shfl_xor(send, 16, 64));
                asm volatile("" : "+v"(v));
                gav[i][j] = v;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    LT_STAMP_B(5);
    // ---- H4: ga = (dot + c0) * keep; softmax backward gs = a (ga - sum_t a ga): one exchange over the 8 waves.
    // keep is read off the stored weights (attn = attn_pre * keep: no RNG here); where attn_pre = 0 the value of ga is irrelevant.
    // attn_pre stays in registers until gs is formed, attn goes straight into aL for the Z pass: every tensor is read once.
    {
        float apk[8][4], sm[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned long long kw[4] = {0ull, 0ull, 0ull, 0ull};   // REKEEP: keep flags of (pixel, heads 4q .. 4q+3)
        if constexpr (REKEEP) {
            const ulonglong2* kp2 = reinterpret_cast<const ulonglong2*>(p.keepbits + (size_t)pidx * NH + 4 * q);
            const ulonglong2 k01 = kp2[0], k23 = kp2[1];
            kw[0] = k01.x; kw[1] = k01.y; kw[2] = k23.x; kw[3] = k23.y;
        }
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            __builtin_amdgcn_sched_barrier(0);
            float adv[4][4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    apk[4 * ih + ii][r] = bload(rap, hv0, trow(4 * ih + ii, r));
                    if constexpr (!REKEEP) adv[ii][r] = bload(rad, hv0, trow(4 * ih + ii, r));
                }
            if constexpr (REKEEP) {
                // attn = attn_pre * keep-scale, the keep flags from the forward's bit words (pixel, head): bits t0 .. t0+7
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned m8 = (unsigned)(kw[r] >> (t0 + 4 * ih)) & 0xfu;
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) adv[ii][r] = apk[4 * ih + ii][r] * (((m8 >> ii) & 1u) ? kscale : 0.f);
                }
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = 4 * ih + ii;
                const float live = i < nt ? 1.f : 0.f;             // steps T..63 carry no weight (their loads repeat row T-1)
                f32x4 at;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (arithmetic masks, not selects of the loaded values: a load that is only needed when i < nt becomes a
                    // branch, and a basic-block boundary here lets LLVM sink the c0 sums below the dots)
                    const float ks = ((dropping && adv[ii][r] == 0.f) ? 0.f : kscale) * live;
                    const float ga = gav[i][r] * ks;
                    gav[i][r] = ga;
                    sm[r] = fmaf(apk[i][r], ga, sm[r]);
                    at[r] = adv[ii][r] * live;
                }
                *reinterpret_cast<f32x4*>(aL + (t0 + i) * R_TP + px * 16 + 4 * q) = at;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * 16 + 4 * q + r) * 16 + px] = sm[r];
        lds_barrier();                             // also: aL (attn) complete
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) s += red[(ww * 16 + 4 * q + r) * 16 + px];
            sm[r] = s;
        }
        lds_barrier();                             // the exchange area is part of xs: read by all waves before the first chunk is staged
        // gs (in place of ga), GS rows to global (for the dx kernel)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gs = i < nt ? apk[i][r] * (gav[i][r] - sm[r]) : 0.f;
                gav[i][r] = gs;
                bstore(gs, rgs, i < nt ? hv0 : 0x80000000u, trow(i, r));        // steps T..63: out of range, dropped by the range check
            }
    }
    __builtin_amdgcn_sched_barrier(0);

    LT_STAMP_B(6);
    // ---- H5: per-pixel MFMA products with xhat staged through LDS (see F5 of the forward): D[h][c] = sum_t w[h][t] xhat[t][c],
    // lane (n = hl, q): head 4q + r, channel 16 cb + n, pixels 2w + u.  Z pass (weights attn, already in aL) first, then the V pass
    // (weights gs): the accumulators of both do not fit beside x, so everything that needs Z is done between the passes.
    const int hl = px;
    const float* ap0 = aL + q * R_TP + (2 * w) * 16 + hl;
    const float* bp0 = xs + (2 * w) * R_XP + q * R_XT + hl;
    f32x4 acc[4][2];
    auto weighted_sums = [&](bool vpass) {
        if (vpass) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(aL + (t0 + i) * R_TP + px * 16 + 4 * q) = (f32x4){gav[i][0], gav[i][1], gav[i][2], gav[i][3]};
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) { acc[cb][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[cb][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            if ((w >> 1) == tq) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        *reinterpret_cast<f32x4*>(xs + px * R_XP + (8 * (w & 1) + i) * R_XT + 16 * q + 4 * u) =
                            (f32x4){x[i][2 * u][0], x[i][2 * u][1], x[i][2 * u + 1][0], x[i][2 * u + 1][1]};
            }
            lds_barrier();
            float av[2][4], bv[2][4][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    av[u][s] = ap0[(16 * tq + 4 * s) * R_TP + 16 * u];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) bv[u][cb][s] = bp0[4 * s * R_XT + 16 * cb + R_XP * u];
                }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[cb][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s], bv[u][cb][s], acc[cb][u], 0, 0, 0);
            if (tq == 3) {
                if (vpass) {
                    // d s0[t][h] partial = sum over the 16 pixels of gs: thread = (t, h), two each
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int o = tid + 512 * k, t = o >> 4;
                        float v = 0.f;
#pragma unroll
                        for (int e = 0; e < 16; ++e) v += aL[t * R_TP + e * 16 + (o & 15)];
                        if (t < T) p.part_s0[(size_t)tile * T * NH + o] = v;
                    }
                } else if (tid < 256) {
                    // asum[h][px] = sum_t attn: thread = (px, h)
                    float v = 0.f;
#pragma unroll 16
                    for (int t = 0; t < 64; ++t) v += aL[t * R_TP + tid];
                    asL[(tid & 15) * 16 + (tid >> 4)] = v;
                }
            }
            lds_barrier();
        }
    };
    weighted_sums(false);                          // Zt
    // ---- H6a: r again (the staging area is free), then everything that needs Z, in the D layout of H5:
    // lane (n = hl, q): head h = 4q + r, channel c = 16 cb + n, pixel pu = 2w + u;  Z, V are the xhat-based sums.
    //   m1_g = 1/n sum_{c in g} gamma_c sum_h asum_h r[h][c]
    //   m2_g = 1/n sum_{c in g} sum_h (r[h][c] (Z[h][c] - beta_c asum_h) + U[h][c] V[h][c])      (sum_t gs = 0)
    {
        float gb[2][4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int s = 0; s < 4; ++s) gb[hh][s] = geb[(size_t)((2 * w + hh) * DV + 4 * s + q) * HW + px];
        load_wA();
        r_mma(gb);
    }
    lds_barrier();
    const float inv_n = 1.f / (float)(4 * T);
    float g2z[4][2];
    {
        float as_[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < 2; ++u) as_[r][u] = asL[(4 * q + r) * 16 + 2 * w + u];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int c = 16 * cb + hl;
            const float gmc = p.gamma[c], btc = p.beta[c];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pu = 2 * w + u;
                float g2 = 0.f, g1 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float rr = rL[((4 * q + r) * 16 + pu) * RB_RP + c];
                    g2 = fmaf(rr, acc[cb][u][r] - btc * as_[r][u], g2);
                    g1 = fmaf(rr, as_[r][u], g1);
                }
                asm volatile("" : "+v"(g2));
                g2z[cb][u] = g2;
                float m1 = gmc * g1;
                // all 16 heads (the four lane quarters) and the 4 channels of group 4 cb + (hl >> 2) (4 adjacent lanes)
                m1 += __shfl_xor(m1, 16, 64); m1 += __shfl_xor(m1, 32, 64); m1 += __shfl_xor(m1, 1, 64); m1 += __shfl_xor(m1, 2, 64);
                if (q == 0 && (hl & 3) == 0) sb.M[(((long)b * HW + pix0 + pu) * NH + 4 * cb + (hl >> 2)) * 2] = m1 * inv_n;
            }
        }
        // Z[h][c][px] transposed through LDS for 64-byte row stores (aL is dead since the last barrier of the Z pass)
        float* zT = lds + RB_A;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) zT[(4 * q + r) * R_ZH + (16 * cb + hl) * RZP + 2 * w + u] = acc[cb][u][r];
        lds_barrier();
        float* zrow = p.Z + (size_t)b * NH * C * HW + pix0 + (tid & 15);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int row = (tid >> 4) + 32 * kk;  // (h, c)
            zrow[(size_t)row * HW] = zT[(row >> 6) * R_ZH + (row & 63) * RZP + (tid & 15)];
        }
        lds_barrier();                             // zT consumed: the area becomes aL again; rL is dead (xs takes its place)
    }
    weighted_sums(true);                           // Vt
    // x is dead from here on
    // ---- H6b: d U partial of the tile through LDS, the V part of m2, d bc partial
    {
        float* pU = lds + RB_A;                    // [8 w][16 (cb, r)][64 lanes]
        f32x4 gd[4];
        if (tid < NH * DV) {
#pragma unroll
            for (int k = 0; k < 4; ++k) gd[k] = *reinterpret_cast<const f32x4*>(geb + (size_t)tid * HW + 4 * k);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            float uv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) uv[r] = p.U[(4 * q + r) * C + 16 * cb + hl];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float m2 = g2z[cb][u];
#pragma unroll
                for (int r = 0; r < 4; ++r) m2 = fmaf(uv[r], acc[cb][u][r], m2);
                m2 += __shfl_xor(m2, 16, 64); m2 += __shfl_xor(m2, 32, 64); m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64);
                if (q == 0 && (hl & 3) == 0) sb.M[(((long)b * HW + pix0 + 2 * w + u) * NH + 4 * cb + (hl >> 2)) * 2 + 1] = m2 * inv_n;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) pU[(w * 16 + cb * 4 + r) * 64 + lane] = acc[cb][0][r] + acc[cb][1][r];
        }
        // d bc[16h+j] partial = sum_px ge[16h+j][px] sum_t attn[h][px]: thread = (h, j), its row of the tile is 64 contiguous bytes
        if (tid < NH * DV) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(asL + (tid >> 4) * 16 + 4 * k);
                v = fmaf(gd[k][0], a4[0], v); v = fmaf(gd[k][1], a4[1], v); v = fmaf(gd[k][2], a4[2], v); v = fmaf(gd[k][3], a4[3], v);
            }
            p.part_bc[(size_t)tile * NH * DV + tid] = v;
        }
        lds_barrier();
        // tile partial of d U [16 h][64 c] (2 per thread), fixed order over the waves
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int o = tid + 512 * kk;          // o = (cb * 4 + r) * 64 + lane'  with lane' = q' * 16 + n
            float sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) sum += pU[ww * 1024 + o];
            const int cbr = o >> 6, ln = o & 63;
            const int h = 4 * (ln >> 4) + (cbr & 3), c = 16 * (cbr >> 2) + (ln & 15);
            sb.part_U[((size_t)tile * NH + h) * C + c] = sum;
        }
    }
    LT_STAMP_B(7);
}

// ------------------------------------------------------------------------------------------ gx kernel, 64-pixel tiles
// d x of the streaming backward with the lane = pixel / wave = GroupNorm group layout of the streaming forward:
//   gx[t][c] = rstd (gamma_c sum_h (attn[h,t] r[h][c] + gs[h,t] U[h][c]) - m1 - xn[t][c] m2)
// The 32-pixel kernel above keeps r AND U of (16 heads x 2 channels) in 64 VGPRs per lane and spills at the 128-register
// limit of a 1024-thread workgroup (25 scratch reloads in its inner loops).  Here a wave owns ONE group, so U[h][4g..4g+3] is
// wave-uniform (SGPR operands), only r[16][4] lives in VGPRs, rows are 256 bytes, and the attention / dscore rows of 16 time
// steps are staged at once (128 KB, the area r passed through): two barriers per 16 steps instead of per 4.
constexpr int GXT = 8;                             // time steps per staged chunk
constexpr int GX_BUF = 2 * GXT * 16 * 64;          // one chunk: [arr 2][GXT][16 h][64 px] = 16,384 floats
constexpr int GX_FLOATS = 2 * GX_BUF;              // two chunks in flight = 128 KB

__global__ __launch_bounds__(1024) void ltae_stream_bwd_gx64_kernel(LtaeParams p, StreamBwd sb) {
    constexpr int C = 64, CPG = 4;
    extern __shared__ float lds[];                 // r of 8 heads [8][16 c4][64 px][4], then the attn / gs chunks
    const int T = p.T, HW = p.HW;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);                              // XCD-aware order
    const int tiles_per_b = HW / 64;
    const int b = (int)(tile / tiles_per_b);
    const int pix = (int)(tile % tiles_per_b) * 64 + lane;
    const long pidx = (long)b * HW + pix;
    const bool rekeep = p.attn_in == nullptr;      // the forward stored the pre-dropout weights only + the keep flags as bits
    const float kscale = p.drop_p <= 0.f ? 1.f : (p.keep != nullptr ? 1.f / (1.f - p.drop_p)
                                                                    : 65536.f / (65536.f - (float)(uint32_t)(p.drop_p * 65536.f + 0.5f)));
    unsigned long long* kbl = reinterpret_cast<unsigned long long*>(lds + GX_FLOATS);        // [16 h][64 px] keep words of the tile
    if (rekeep) kbl[(threadIdx.x & 15) * 64 + (threadIdx.x >> 4)] = p.keepbits[((size_t)pidx - lane) * NH + threadIdx.x];   // 8 KB, contiguous
    const unsigned* kb32 = reinterpret_cast<const unsigned*>(kbl);                           // (read after the barriers of the r phase below)
    const int g = w;

    // ---- r[h][c][px] = sum_j Wc[16h+j][c] ge[16h+j][px] on the MFMA, wave = head:  4 channel tiles x 4 pixel tiles x 4 k-steps
    f32x4 d[4][4];
    {
        const int h = w, mi = lane & 15, mk = lane >> 4;
        float gb_[4][4], wa_[4][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                gb_[ks][nt] = p.g_emb[((size_t)b * NH * DV + h * DV + 4 * ks + mk) * HW + (pix - lane) + nt * 16 + mi];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) wa_[ks][mt] = p.Wc[(size_t)(h * DV + 4 * ks + mk) * C + mt * 16 + mi];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_[ks][mt], gb_[ks][nt], acc, 0, 0, 0);
                d[mt][nt] = acc;                   // rows c = 16 mt + 4 mk + r, column px = 16 nt + mi
            }
    }
    // through LDS in two halves of 8 heads: [h & 7][c / 4][px][c & 3]; wave g then keeps r[16][4] of its group
    float r[NH][CPG];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half == 1) __syncthreads();            // first half read
        if ((w >> 3) == half) {
            const int mi = lane & 15, mk = lane >> 4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    *reinterpret_cast<f32x4*>(lds + ((size_t)((w & 7) * 16 + mt * 4 + mk) * 64 + nt * 16 + mi) * 4) = d[mt][nt];
        }
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((size_t)(hh * 16 + g) * 64 + lane) * 4);
#pragma unroll
            for (int cc = 0; cc < CPG; ++cc) r[half * 8 + hh][cc] = v[cc];
        }
    }
    const float mean = p.stats_in[(pidx * NH + g) * 2], rstd = p.stats_in[(pidx * NH + g) * 2 + 1];
    const float m1 = sb.M[(pidx * NH + g) * 2 + 0], m2 = sb.M[(pidx * NH + g) * 2 + 1];
    float gm[CPG];
#pragma unroll
    for (int cc = 0; cc < CPG; ++cc) gm[cc] = p.gamma[g * CPG + cc] * rstd;
    // U[h][4g + cc] is wave-uniform: 64 scalar registers (readfirstlane pins them to SGPRs; left to the compiler they were
    // re-loaded with vector loads inside the time loop because the gx stores might alias U)
    float ug[NH][CPG];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int cc = 0; cc < CPG; ++cc)
            ug[h][cc] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.U[h * C + g * CPG + cc])));
    const float* xg = p.x + (size_t)b * T * C * HW + (size_t)(g * CPG) * HW + pix;
    float* gxg = p.gx + (size_t)b * T * C * HW + (size_t)(g * CPG) * HW + pix;
    const size_t hstride = (size_t)p.B * T * HW;
    float dgam[CPG] = {0.f, 0.f, 0.f, 0.f}, dbet[CPG] = {0.f, 0.f, 0.f, 0.f};     // sum_t d xhat * xn, sum_t d xhat (this pixel)

    // attn / gs rows of GXT steps per chunk by LDS-DMA, two buffers: wave w fetches step (w & 7) of array (w >> 3), 4 heads x 64
    // pixels per instruction; chunk k + 1 is in flight while chunk k is consumed
    const float* asrc = rekeep ? p.attn_pre_in : p.attn_in;
    auto issue_dma = [&](int tc0, int kbuf) {
        const int tt = w & 7, t = tc0 + tt < T ? tc0 + tt : T - 1;
        const float* src = (w >> 3) == 0 ? asrc : p.GS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int h = 4 * j + (lane >> 4);
            const size_t o = (size_t)h * hstride + ((size_t)b * T + t) * HW + (pix - lane) + (lane & 15) * 4;
            __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(src + o), (C2S_AS3 void*)(lds + kbuf * GX_BUF + (w * 16 + j * 4) * 64), 16, 0, 0);
        }
    };
    float xn[2][CPG];
    auto issue = [&](int t0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = t0 + u < T ? t0 + u : T - 1;
#pragma unroll
            for (int cc = 0; cc < CPG; ++cc) xn[u][cc] = xg[(size_t)(t * C + cc) * HW];
        }
    };
    __syncthreads();                               // r read by every wave: the area becomes the two chunk buffers
    issue_dma(0, 0);
    issue(0);
    for (int tc0 = 0, kb = 0; tc0 < T; tc0 += GXT, kb ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // chunk kb landed (every wave's part); chunk kb ^ 1 consumed
        if (tc0 + GXT < T) issue_dma(tc0 + GXT, kb ^ 1);
        float* bufk = lds + kb * GX_BUF;
        if (rekeep) {                              // thread (head w, pixel lane): the 8 steps of the chunk lie in one byte of the keep word
            const unsigned kbyte = kb32[(w * 64 + lane) * 2 + (tc0 >> 5)] >> (tc0 & 31);
#pragma unroll
            for (int tt = 0; tt < GXT; ++tt) bufk[(tt * NH + w) * 64 + lane] *= ((kbyte >> tt) & 1u) ? kscale : 0.f;
            lds_barrier();
        }
        const int tn = T - tc0 < GXT ? T - tc0 : GXT;
        for (int tt0 = 0; tt0 < tn; tt0 += 2) {
            float xv[2][CPG];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int cc = 0; cc < CPG; ++cc) xv[u][cc] = xn[u][cc];
            if (tc0 + tt0 + 2 < T) issue(tc0 + tt0 + 2);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tt = tt0 + u, t = tc0 + tt;
                if (tt < tn) {
                    float acc[CPG] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const float a = bufk[(tt * NH + h) * 64 + lane];
                        const float gs = bufk[((GXT + tt) * NH + h) * 64 + lane];
#pragma unroll
                        for (int cc = 0; cc < CPG; ++cc) {
                            acc[cc] = fmaf(a, r[h][cc], acc[cc]);
                            acc[cc] = fmaf(gs, ug[h][cc], acc[cc]);
                        }
                    }
#pragma unroll
                    for (int cc = 0; cc < CPG; ++cc) {
                        const float xnn = (xv[u][cc] - mean) * rstd;
                        gxg[(size_t)(t * C + cc) * HW] = gm[cc] * acc[cc] - rstd * fmaf(xnn, m2, m1);
                        dgam[cc] = fmaf(acc[cc], xnn, dgam[cc]);
                        dbet[cc] += acc[cc];
                    }
                }
            }
        }
    }
    // d gamma / d beta partials of the tile (sum over its 64 pixels) when the heads kernel leaves them to this one
    if (sb.gb64 != nullptr) {
#pragma unroll
        for (int cc = 0; cc < CPG; ++cc) {
            const float sg = wave_sum(dgam[cc]), sbt = wave_sum(dbet[cc]);
            if (lane == 0) {
                sb.gb64[((size_t)tile * C + g * CPG + cc) * 2 + 0] = sg;
                sb.gb64[((size_t)tile * C + g * CPG + cc) * 2 + 1] = sbt;
            }
        }
    }
}

// LDS-resident 4-pixel forward (ltae_lds_fwd_kernel): C in {64, 128, 256}, whole pixel quads, the series of four pixels in LDS
size_t lds_fwd_bytes(const c2s_ltae_desc* d) {
    const size_t rows = (size_t)d->T * (d->C + 1) > (size_t)NH * d->C ? (size_t)d->T * (d->C + 1) : (size_t)NH * d->C;
    return (rows * 4 + (size_t)d->T * 64 + (size_t)d->C * 24 + 1024 + 64) * sizeof(float);
}
bool use_lds_fwd(const c2s_ltae_desc* d) {
    static const bool enabled = [] { const char* e = getenv("C2S_LTAE_LDS"); return !(e && e[0] == '0'); }();
    return enabled && (d->C == 64 || d->C == 128 || d->C == 256) && d->HW % 4 == 0 && lds_fwd_bytes(d) <= 160 * 1024;
}
size_t lds_bwd_bytes(const c2s_ltae_desc* d) {
    return ((size_t)d->T * (d->C + 1) * 4 + (size_t)NH * d->C * 4 + 4 * (size_t)d->T * 64 + 1024 + 2 * (size_t)d->C + 128 + 1024 + 128) *
           sizeof(float);
}
bool use_lds_bwd(const c2s_ltae_desc* d) {
    static const bool enabled = [] { const char* e = getenv("C2S_LTAE_LDS_BWD"); return !(e && e[0] == '0'); }();
    return enabled && (d->C == 64 || d->C == 128 || d->C == 256) && d->HW % 4 == 0 && lds_bwd_bytes(d) <= 160 * 1024;
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
    p.keepbits = (unsigned long long*)d->keep_bits;
}

void init_hook() {
    C2S_RAISE_LDS(ltae_fwd_kernel);
    C2S_RAISE_LDS(ltae_lds_fwd_kernel<64>);
    C2S_RAISE_LDS(ltae_lds_fwd_kernel<128>);
    C2S_RAISE_LDS(ltae_lds_fwd_kernel<256>);
    C2S_RAISE_LDS(ltae_lds_bwd_kernel<64>);
    C2S_RAISE_LDS(ltae_lds_bwd_kernel<128>);
    C2S_RAISE_LDS(ltae_lds_bwd_kernel<256>);
    C2S_RAISE_LDS(ltae_bwd_heads_kernel);
    C2S_RAISE_LDS(ltae_bwd_gx_kernel);
    C2S_RAISE_LDS(ltae_stream_bwd_heads_kernel<4>);
    C2S_RAISE_LDS(ltae_stream_bwd_gx_kernel<4>);
    C2S_RAISE_LDS(ltae_stream_bwd_gx64_kernel);
    C2S_RAISE_LDS(ltae_reg_bwd_heads_kernel<false>);
    C2S_RAISE_LDS(ltae_reg_bwd_heads_kernel<true>);
    C2S_RAISE_LDS(ltae_reg_fwd_kernel<false>);
    C2S_RAISE_LDS(ltae_reg_fwd_kernel<true>);
}
C2sInitRegistrar registrar(init_hook);

}  // namespace

#ifdef C2S_LT_STAMP
extern "C" int c2s_debug_ltae_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lt_stamps), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : 1;
}
extern "C" int c2s_debug_ltae_stamps_bwd(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lt_stamps_bwd), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t c2s_ltae_fwd_workspace_floats(const c2s_ltae_desc* d) {
    if (!d) return 0;
    return (size_t)d->C * NH + NH;          // Ut [C][16], cU [16] of the streaming path
}

// The streaming kernels pay off once the 64-pixel tiles fill the chip; below that the 16-pixel LDS kernel is used.
static bool use_stream(const c2s_ltae_desc* d) {
    const int cus = c2s_cus();
    return d->C == 64 && (long)d->B * ((d->HW + 63) / 64) >= 2L * cus;
}

// The register-resident forward (16-pixel tiles, x read once) needs C == 64 and whole tiles; it pays off as soon as the tiles
// fill the chip a few times over.  C2S_LTAE_REG=0 keeps the three-pass streaming kernel (A/B runs).
static bool use_reg_fwd(const c2s_ltae_desc* d) {
    static const bool enabled = [] { const char* e = getenv("C2S_LTAE_REG"); return !(e && e[0] == '0'); }();
    return enabled && d->C == 64 && d->HW % RPX == 0 && (long)d->B * (d->HW / RPX) >= 4L * c2s_cus() &&
           (long)NH * d->B * d->T * d->HW < (1L << 30);             // 32-bit element offsets into attn
}

extern "C" int c2s_ltae_uses_streaming(const c2s_ltae_desc* d) {
    return d && check(d) == C2S_OK && (use_stream(d) || use_reg_fwd(d)) ? 1 : 0;
}

extern "C" int c2s_ltae_fwd_path(const c2s_ltae_desc* d) {
    if (!d || check(d) != C2S_OK) return -1;
    return use_reg_fwd(d) ? 2 : (use_stream(d) ? 1 : 0);
}

static bool reg_bwd_enabled() {
    static const bool on = [] { const char* e = getenv("C2S_LTAE_REG_BWD"); return !(e && e[0] == '0'); }();
    return on;
}
// register-resident heads kernel + 64-px dx kernel; its buffer descriptors address < 2^31 bytes per tensor
static bool use_reg_bwd(const c2s_ltae_desc* d, bool with_emb) {
    return with_emb && use_stream(d) && reg_bwd_enabled() && use_reg_fwd(d) && d->HW % 64 == 0 &&
           (size_t)16 * d->B * d->T * d->HW < ((size_t)1 << 29) && (size_t)d->T * d->C * d->HW < ((size_t)1 << 29);
}

// 1 when a caller that never reads the post-dropout weights may pass attn == NULL to c2s_ltae_attn_fwd_ws AND to
// c2s_ltae_attn_bwd (both take the register-resident kernels for this shape; RNG mask, embedding output)
extern "C" int c2s_ltae_attn_optional(const c2s_ltae_desc* d) {
    return d && check(d) == C2S_OK && d->keep == nullptr && use_reg_bwd(d, true) ? 1 : 0;
}

extern "C" int c2s_ltae_attn_fwd_ws(const c2s_ltae_desc* d, const float* x, const float* gamma, const float* beta,
                                    const float* U, const float* s0, const float* Wc, const float* bc, const float* pe,
                                    const int* valid, float* attn, float* attn_pre, float* emb, float* stats,
                                    float* workspace, size_t ws_floats, void* stream) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(x && gamma && beta && U && s0 && stats, "ltae_fwd: null pointer");
    // attn == NULL: the caller never reads the post-dropout weights (TimeUNet without return_att); only the register-resident
    // forward skips the store; a later backward then needs attn_pre and the RNG mask (c2s_ltae_attn_optional)
    C2S_REQUIRE(attn != nullptr || (use_reg_fwd(d) && emb != nullptr),
                "ltae_fwd: attn may only be NULL on the register-resident path (c2s_ltae_fwd_path == 2) with an embedding output");
    C2S_REQUIRE(emb == nullptr || (Wc && bc && pe), "ltae_fwd: embedding output needs Wc, bc, pe");
    LtaeParams p = {};
    fill(p, d);
    p.x = x; p.gamma = gamma; p.beta = beta; p.U = U; p.s0 = s0; p.Wc = Wc; p.bc = bc; p.pe = pe; p.valid = valid;
    p.attn = attn; p.attn_pre = attn_pre; p.emb = emb; p.stats = stats;
    hipStream_t st = (hipStream_t)stream;
    if (use_reg_fwd(d)) {
        const bool h32 = (unsigned long long)NH * d->B * d->HW * ((d->T + 1) / 2) <= 0xFFFFFFFFull;
        if (h32) hipLaunchKernelGGL(ltae_reg_fwd_kernel<true>, dim3(d->B * (d->HW / RPX)), dim3(512), R_FLOATS * sizeof(float), st, p);
        else hipLaunchKernelGGL(ltae_reg_fwd_kernel<false>, dim3(d->B * (d->HW / RPX)), dim3(512), R_FLOATS * sizeof(float), st, p);
        C2S_CHECK_LAUNCH("ltae_reg_fwd");
        return C2S_OK;
    }
    if (workspace != nullptr && attn_pre != nullptr && use_stream(d)) {
        C2S_REQUIRE(ws_floats >= c2s_ltae_fwd_workspace_floats(d), "ltae_fwd: workspace too small");
        float* Ut = workspace;
        float* cU = workspace + (size_t)d->C * NH;
        hipLaunchKernelGGL(ltae_prep_kernel, dim3(1), dim3(256), 0, st, U, gamma, beta, Ut, cU, d->C);
        C2S_CHECK_LAUNCH("ltae_prep");
        hipLaunchKernelGGL(ltae_stream_fwd_kernel<4>, dim3(d->B * ((d->HW + 63) / 64)), dim3(1024), 0, st, p, Ut, cU);
        C2S_CHECK_LAUNCH("ltae_stream_fwd");
        return C2S_OK;
    }
    c2s_ensure_init();
    if (use_lds_fwd(d)) {
        const dim3 grid(d->B * (d->HW / 4));
        const size_t lb = lds_fwd_bytes(d);
        if (d->C == 64) hipLaunchKernelGGL(ltae_lds_fwd_kernel<64>, grid, dim3(256), lb, st, p);
        else if (d->C == 128) hipLaunchKernelGGL(ltae_lds_fwd_kernel<128>, grid, dim3(256), lb, st, p);
        else hipLaunchKernelGGL(ltae_lds_fwd_kernel<256>, grid, dim3(256), lb, st, p);
        C2S_CHECK_LAUNCH("ltae_lds_fwd");
        return C2S_OK;
    }
    hipLaunchKernelGGL(ltae_fwd_kernel, dim3(d->B * ((d->HW + 15) / 16)), dim3(256), fwd_lds(d), st, p);
    C2S_CHECK_LAUNCH("ltae_fwd");
    return C2S_OK;
}

extern "C" int c2s_ltae_attn_fwd(const c2s_ltae_desc* d, const float* x, const float* gamma, const float* beta,
                                 const float* U, const float* s0, const float* Wc, const float* bc, const float* pe,
                                 const int* valid, float* attn, float* attn_pre, float* emb, float* stats,
                                 void* stream) {
    return c2s_ltae_attn_fwd_ws(d, x, gamma, beta, U, s0, Wc, bc, pe, valid, attn, attn_pre, emb, stats, nullptr, 0, stream);
}

// workspace: GS [16,B,T,HW] | V [B,16,C,HW] | Z [B,16,C,HW] | part_s0 [tiles][T][16] | part_bc [tiles][256]
//            | part_gb [tiles][C][2] | the slice sums of reduce_rows (doubles)
static size_t reduce_tmp_floats(const c2s_ltae_desc* d) {      // groups * RR_SLICES * K doubles for the widest of the three sums
    const size_t kmax = (size_t)d->B * d->T * NH > (size_t)NH * d->C ? (size_t)d->B * d->T * NH : (size_t)NH * d->C;
    return 2 * (size_t)RR_SLICES * (kmax > 256 ? kmax : 256);
}
extern "C" size_t c2s_ltae_bwd_workspace_floats(const c2s_ltae_desc* d) {
    if (!d) return 0;
    const size_t tiles = (size_t)d->B * ((d->HW + 3) / 4);   // upper bound (4-pixel tiles)
    return (size_t)NH * d->B * d->T * d->HW + 2 * (size_t)d->B * NH * d->C * d->HW + tiles * d->T * NH + tiles * 256 +
           tiles * d->C * 2 + 2 + reduce_tmp_floats(d);
}

extern "C" int c2s_ltae_attn_bwd(const c2s_ltae_desc* d, const float* x, const float* gamma, const float* beta,
                                 const float* U, const float* s0, const float* Wc, const float* bc, const float* pe,
                                 const int* valid, const float* attn, const float* attn_pre, const float* stats,
                                 const float* g_emb, const float* g_attn, float* gx, float* gU, float* gs0, float* gWc,
                                 float* gbc, float* ggamma, float* gbeta, float* workspace, size_t ws_floats,
                                 void* stream) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(x && gamma && beta && U && Wc && bc && pe && attn_pre && stats && gx && gU && gs0 && gWc && gbc &&
                    ggamma && gbeta && workspace,
                "ltae_bwd: null pointer");
    C2S_REQUIRE(ws_floats >= c2s_ltae_bwd_workspace_floats(d), "ltae_bwd: workspace too small");
    (void)s0; (void)valid;
    const bool stream_path = g_emb != nullptr && use_stream(d);
    const bool reg_heads = use_reg_bwd(d, g_emb != nullptr);
    C2S_REQUIRE(attn != nullptr || (reg_heads && d->keep_bits != nullptr),
                "ltae_bwd: attn may only be NULL where the forward could omit it (register-resident path) and left the keep flags in d->keep_bits");
    const bool lds_path = !stream_path && use_lds_bwd(d);      // fused LDS-resident kernel on 4-pixel tiles (small maps)
    const int PT = reg_heads ? RPX : (stream_path ? SPT : (lds_path ? 4 : bwd_pt(d)));
    const size_t tiles = (size_t)d->B * ((d->HW + PT - 1) / PT);
    const size_t tiles_ws = (size_t)d->B * ((d->HW + 3) / 4);
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
    float* rt_ = p.part_gb + tiles_ws * d->C * 2;
    double* rtmp = reinterpret_cast<double*>(rt_ + (((uintptr_t)rt_ >> 2) & 1));      // 8-byte aligned
    hipStream_t st = (hipStream_t)stream;
    c2s_ensure_init();
    StreamBwd sb = {};
    if (stream_path) {
        // the V area of the workspace is not used by the streaming kernels: it holds M [P][16][2] and part_U [tiles][16][C]
        sb.M = p.V;
        sb.part_U = p.V + (size_t)d->B * d->HW * NH * 2;
        const size_t lds1 = ((size_t)NH * d->C * SPT + NH * 2 * SPT + 2 * NH * SPT + 2 * SCH * 4 * SPT * 4 + (size_t)d->T * DV) * sizeof(float);
        if (reg_heads) {
            sb.gb64 = p.part_gb;                   // d gamma / d beta partials come from the dx kernel, one per 64-pixel tile
            if (attn == nullptr) hipLaunchKernelGGL(ltae_reg_bwd_heads_kernel<true>, dim3(tiles), dim3(512), RB_FLOATS * sizeof(float), st, p, sb);
            else hipLaunchKernelGGL(ltae_reg_bwd_heads_kernel<false>, dim3(tiles), dim3(512), RB_FLOATS * sizeof(float), st, p, sb);
        } else {
            hipLaunchKernelGGL(ltae_stream_bwd_heads_kernel<4>, dim3(tiles), dim3(1024), lds1, st, p, sb);
        }
        C2S_CHECK_LAUNCH("ltae_stream_bwd_heads");
        const size_t lds2 = ((size_t)NH * d->C * SPT + 2 * SCH * 4 * SPT * 4) * sizeof(float);
        static const bool gx64 = [] { const char* e = getenv("C2S_LTAE_GX64"); return !(e && e[0] == '0'); }();
        if ((gx64 || reg_heads) && d->C == 64 && d->HW % 64 == 0) {
            hipLaunchKernelGGL(ltae_stream_bwd_gx64_kernel, dim3(d->B * (d->HW / 64)), dim3(1024), GX_FLOATS * sizeof(float) + 8192, st, p, sb);
        } else {
            hipLaunchKernelGGL(ltae_stream_bwd_gx_kernel<4>, dim3(tiles), dim3(1024), lds2, st, p, sb);
        }
        C2S_CHECK_LAUNCH("ltae_stream_bwd_gx");
    } else if (lds_path) {
        sb.part_U = p.V;                           // [tiles][16][C] <= the V area [B][16][C][HW]
        const size_t lb = lds_bwd_bytes(d);
        if (d->C == 64) hipLaunchKernelGGL(ltae_lds_bwd_kernel<64>, dim3(tiles), dim3(256), lb, st, p, sb);
        else if (d->C == 128) hipLaunchKernelGGL(ltae_lds_bwd_kernel<128>, dim3(tiles), dim3(256), lb, st, p, sb);
        else hipLaunchKernelGGL(ltae_lds_bwd_kernel<256>, dim3(tiles), dim3(256), lb, st, p, sb);
        C2S_CHECK_LAUNCH("ltae_lds_bwd");
    } else {
        hipLaunchKernelGGL(ltae_bwd_heads_kernel, dim3(tiles), dim3(256), bwd1_lds(d), st, p);
        C2S_CHECK_LAUNCH("ltae_bwd_heads");
        hipLaunchKernelGGL(ltae_bwd_gx_kernel, dim3(tiles), dim3(256), bwd2_lds(d), st, p);
        C2S_CHECK_LAUNCH("ltae_bwd_gx");
    }
    // reductions
    const int tpb = (d->HW + PT - 1) / PT;
    {   // gs0[b][t][h] = sum over the tiles of b
        reduce_rows(p.part_s0, gs0, d->B, tpb, d->T * NH, rtmp, st);
        C2S_CHECK_LAUNCH("ltae_reduce_s0");
    }
    reduce_rows(p.part_bc, gbc, 1, (int)tiles, 256, rtmp, st);
    C2S_CHECK_LAUNCH("ltae_reduce_bc");
    {   // interleaved (dgamma, dbeta) partials -> the two outputs
        const int gb_tiles = reg_heads ? d->B * (d->HW / 64) : (int)tiles;
        reduce_rows(p.part_gb, ggamma, 1, gb_tiles, 2 * d->C, rtmp, st, gbeta);
        C2S_CHECK_LAUNCH("ltae_reduce_gb");
    }
    if (stream_path || lds_path) {
        reduce_rows(sb.part_U, gU, 1, (int)tiles, NH * d->C, rtmp, st);
    } else {
        hipLaunchKernelGGL(sum_over_pixels_kernel, dim3(NH * d->C), dim3(64), 0, st, p.V, gU, d->B, NH * d->C, d->HW);
    }
    C2S_CHECK_LAUNCH("ltae_gU");
    if (g_emb != nullptr && stream_path && d->HW % 64 == 0) {
        // 32 pixel slices per head; the partials live behind part_U in the (unused) V area of the workspace
        const int ntiles = d->B * (d->HW / 64);
        const int slices = ntiles >= 32 ? 32 : ntiles;
        const int tps = (ntiles + slices - 1) / slices;
        float* part_wc = sb.part_U + tiles * NH * d->C;
        hipLaunchKernelGGL(gwc_mfma_kernel, dim3(NH, slices), dim3(256), 0, st, g_emb, p.Z, part_wc, d->B, d->HW, tps);
        C2S_CHECK_LAUNCH("ltae_gWc_mfma");
        reduce_rows(part_wc, gWc, 1, slices, NH * DV * d->C, nullptr, st);
        C2S_CHECK_LAUNCH("ltae_gWc_reduce");
    } else if (g_emb != nullptr) {
        hipLaunchKernelGGL(gwc_kernel, dim3(NH * d->C), dim3(256), 0, st, g_emb, p.Z, gWc, d->B, d->C, d->HW);
        C2S_CHECK_LAUNCH("ltae_gWc");
    } else {
        hipMemsetAsync(gWc, 0, (size_t)256 * d->C * sizeof(float), st);
    }
    return C2S_OK;
}

extern "C" int c2s_positional_table(const long long* dates, float* pe, long n, float period, void* stream) {
    C2S_REQUIRE(dates && pe && n > 0 && period > 0.f, "positional_table: bad args");
    hipLaunchKernelGGL(positional_table_kernel, dim3(cdiv(n * DV, 256)), dim3(256), 0, (hipStream_t)stream, dates, pe, n, period);
    C2S_CHECK_LAUNCH("positional_table");
    return C2S_OK;
}

extern "C" int c2s_ltae_fold_fwd(const float* Q, const float* Wk, const float* bk, const float* Wc, const float* bc,
                                 const float* pe, float* U, float* s0, float* qwk, int BT, int C, void* stream) {
    C2S_REQUIRE(Q && Wk && bk && Wc && bc && pe && U && s0 && qwk && BT > 0 && C > 0, "ltae_fold_fwd: bad args");
    C2S_REQUIRE(C <= 1024, "ltae_fold_fwd: C > 1024");
    hipLaunchKernelGGL(ltae_fold_fwd_kernel, dim3(NH), dim3(1024), 0, (hipStream_t)stream, Q, Wk, bk, Wc, bc, pe, U, s0, qwk, BT, C);
    C2S_CHECK_LAUNCH("ltae_fold_fwd");
    return C2S_OK;
}

extern "C" size_t c2s_ltae_fold_bwd_workspace_floats(void) { return (size_t)NH + NH * DV + NH * DM; }

extern "C" int c2s_ltae_fold_bwd(const float* Q, const float* Wk, const float* bk, const float* Wc, const float* bc,
                                 const float* pe, const float* qwk, const float* gU, const float* gs0,
                                 const float* gWc_attn, const float* gbc_attn, float* gQ, float* gWk, float* gbk,
                                 float* gWc, float* gbc, int BT, int C, int acc_mask, float* workspace, size_t ws_floats,
                                 void* stream) {
    C2S_REQUIRE(Q && Wk && bk && Wc && bc && pe && qwk && gU && gs0 && gQ && gWk && gbk && gWc && gbc && workspace && BT > 0 &&
                    C > 0, "ltae_fold_bwd: bad args");
    C2S_REQUIRE(ws_floats >= c2s_ltae_fold_bwd_workspace_floats(), "ltae_fold_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* SP = workspace;                    // S [16] | P [16][16]
    float* gq = workspace + NH + NH * DV;     // [16][256]
    hipLaunchKernelGGL(ltae_fold_bwd1_kernel, dim3(NH + NH * DV), dim3(64), 0, st, gs0, pe, SP, BT);
    C2S_CHECK_LAUNCH("ltae_fold_bwd1");
    hipLaunchKernelGGL(ltae_fold_bwd2_kernel, dim3(DM), dim3(64), 0, st, Wc, bc, qwk, gU, SP, gWc_attn, gbc_attn, gq, gWc, gbc, C,
                       (acc_mask >> 3) & 1, (acc_mask >> 4) & 1);
    C2S_CHECK_LAUNCH("ltae_fold_bwd2");
    hipLaunchKernelGGL(ltae_fold_bwd3_kernel, dim3(NH * DK), dim3(64), 0, st, Q, Wk, bk, gq, SP, gQ, gWk, gbk, acc_mask & 1,
                       (acc_mask >> 1) & 1, (acc_mask >> 2) & 1);
    C2S_CHECK_LAUNCH("ltae_fold_bwd3");
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
    reduce_rows(workspace, dgamma, 1, B * chunks, 2 * C, nullptr, st, dbeta);
    C2S_CHECK_LAUNCH("pixel_gn_reduce");
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
