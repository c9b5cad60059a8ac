// Learnable positional encoders of the L-TAE (off-default constructor flags use_doy, use_abs_rel_enc, add_linear).
//
// Reference: src/backbones/positional_encoding.py:7-43 (PositionalEncoder with add_linear: Linear(256,256) on the tiled
// sinusoid), :46-73 (AbsolutePositionalEncoder: one_hot(day of year, 365) -> Linear(365,16), tiled over the 16 heads),
// src/backbones/tae.py:404-430 (which encoder a flag combination builds), :467-479 (out = out + pe (+ pe_abs)).
//
// The attention kernels of ltae.hip are built around the default table pe[b,t,16] (the same 16 values for every head).  A
// learnable encoder is a general table pe[b,t,256] with parameters behind it; since the positional term enters the block
// linearly, it is handled NEXT TO the attention kernels, which run with a zero table:
//   forward   s0[bt,h]        += sum_m qwk[h,m] pe[bt,m]                     (scores: keys see e = Wc xhat + bc + pe)
//             emb[b,16h+j,p]  += sum_t attn[h,b,t,p] pe[b,t,16h+j]           (values are the unprojected e)
//   backward  g_attn[h,b,t,p] += sum_j g_emb[b,16h+j,p] pe[b,t,16h+j]        (before the attention backward)
//             g_pe[b,t,16h+j]  = sum_p attn[h,b,t,p] g_emb[b,16h+j,p] + sum_h' gs0[bt,h'] qwk[h',16h+j]
//             d fc1_k.weight, d Q: the pe part of the key fold (ltae_fold_bwd sees a zero table)
//             d encoder parameters from g_pe (scatter over the day of year / a 256 x 256 outer-product sum)
// These are a few MFLOP and a few MB on the 16 x 16 maps of U-TAE / W-TAE; at TimeUNet's full resolution the two pixel sums
// read the attention masks once more (0.5 GB each): an off-default option, not tuned further.  Fixed summation orders.
#include "common.h"

namespace {

constexpr int NH = 16, DV = 16, DM = 256, DK = 4, NDAY = 365;

__device__ __forceinline__ float sinus(long long date, int j, float period) {
    const float denom = powf(period, (float)(2 * (j / 2)) / (float)DV);
    const float a = (float)date / denom;
    return (j & 1) ? cosf(a) : sinf(a);
}

// mode 1: doy (dates0 = day of year); 2: abs_rel (dates0 relative, dates1 day of year); 3: linear (dates0 relative)
__global__ __launch_bounds__(256) void pe_table_kernel(int mode, const long long* __restrict__ d0, const long long* __restrict__ d1,
                                                       float period, const float* __restrict__ W, const float* __restrict__ b,
                                                       float* __restrict__ pe, float* __restrict__ sin256, int* __restrict__ bad) {
    __shared__ float s16[DV];
    const int bt = blockIdx.x, o = threadIdx.x, j = o & (DV - 1);
    if (mode == 3) {
        if (o < DV) s16[o] = sinus(d0[bt], o, period);
        __syncthreads();
        float v = b[o];
        for (int i = 0; i < DM; ++i) v = fmaf(W[(size_t)o * DM + i], s16[i & (DV - 1)], v);
        pe[(size_t)bt * DM + o] = v;
        sin256[(size_t)bt * DM + o] = s16[j];
        return;
    }
    long long day = mode == 1 ? d0[bt] : d1[bt];
    if (day < 0 || day >= NDAY) {                       // F.one_hot raises on these: counted, clamped
        if (o == 0) atomicAdd(bad, 1);
        day = day < 0 ? 0 : NDAY - 1;
    }
    float v = W[(size_t)j * NDAY + day] + b[j];
    if (mode == 2) v += sinus(d0[bt], j, period);
    pe[(size_t)bt * DM + o] = v;
}

// second encoder of use_abs_rel_enc next to a learnable first one (tae.py:419-422,473): pe[bt,16h+j] += W2[j,day] + b2[j]
__global__ __launch_bounds__(256) void pe_abs_add_kernel(const long long* __restrict__ d1, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ pe, int* __restrict__ bad) {
    const int bt = blockIdx.x, o = threadIdx.x, j = o & (DV - 1);
    long long day = d1[bt];
    if (day < 0 || day >= NDAY) {
        if (o == 0) atomicAdd(bad, 1);
        day = day < 0 ? 0 : NDAY - 1;
    }
    pe[(size_t)bt * DM + o] += W[(size_t)j * NDAY + day] + b[j];
}

__global__ __launch_bounds__(256) void pe_s0_add_kernel(const float* __restrict__ qwk, const float* __restrict__ pe,
                                                        float* __restrict__ s0, int BT) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= BT * NH) return;
    const int bt = e / NH, h = e % NH;
    float v = 0.f;
    for (int m = 0; m < DM; ++m) v = fmaf(qwk[h * DM + m], pe[(size_t)bt * DM + m], v);
    s0[e] += v;
}

__global__ __launch_bounds__(256) void pe_emb_add_kernel(const float* __restrict__ attn, const float* __restrict__ pe,
                                                         float* __restrict__ emb, int B, int T, int HW) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int bc = blockIdx.y, b = bc / DM, c = bc % DM, h = c / DV;
    if (p >= HW) return;
    const float* a = attn + ((size_t)(h * B + b) * T) * HW + p;
    const float* pr = pe + (size_t)b * T * DM + c;
    float v = 0.f;
    for (int t = 0; t < T; ++t) v = fmaf(a[(size_t)t * HW], pr[(size_t)t * DM], v);
    emb[(size_t)bc * HW + p] += v;
}

__global__ __launch_bounds__(256) void pe_gattn_kernel(const float* __restrict__ gemb, const float* __restrict__ pe,
                                                       const float* __restrict__ gin, float* __restrict__ gout, int B, int T,
                                                       int HW) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int hbt = blockIdx.y, t = hbt % T, hb = hbt / T, b = hb % B, h = hb / B;
    if (p >= HW) return;
    const float* ge = gemb + ((size_t)b * DM + h * DV) * HW + p;
    const float* pr = pe + ((size_t)b * T + t) * DM + h * DV;
    float v = gin != nullptr ? gin[(size_t)hbt * HW + p] : 0.f;
#pragma unroll
    for (int j = 0; j < DV; ++j) v = fmaf(ge[(size_t)j * HW], pr[j], v);
    gout[(size_t)hbt * HW + p] = v;
}

// block per (b, t, h): g_pe[b,t,16h+j] for the 16 j
__global__ __launch_bounds__(256) void pe_grad_kernel(const float* __restrict__ attn, const float* __restrict__ gemb,
                                                      const float* __restrict__ gs0, const float* __restrict__ qwk,
                                                      float* __restrict__ gpe, int B, int T, int HW) {
    __shared__ float red[4][DV];
    const int h = blockIdx.x % NH, bt = blockIdx.x / NH, b = bt / T, t = bt % T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[DV];
#pragma unroll
    for (int j = 0; j < DV; ++j) acc[j] = 0.f;
    if (gemb != nullptr) {
        const float* a = attn + ((size_t)(h * B + b) * T + t) * HW;
        const float* ge = gemb + ((size_t)b * DM + h * DV) * HW;
        for (int p = tid; p < HW; p += 256) {
            const float av = a[p];
#pragma unroll
            for (int j = 0; j < DV; ++j) acc[j] = fmaf(av, ge[(size_t)j * HW + p], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < DV; ++j) {
        const float v = wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (tid < DV) {
        float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        for (int hh = 0; hh < NH; ++hh) v = fmaf(gs0[(size_t)bt * NH + hh], qwk[hh * DM + h * DV + tid], v);
        gpe[(size_t)bt * DM + h * DV + tid] = v;
    }
}

// pe part of the key fold's adjoint: P[h][m] = sum_bt gs0[bt,h] pe[bt,m];  gWk[4h+d][m] += Q[4h+d] P[m] / 2,
// gQ[4h+d] += sum_m P[m] Wk[4h+d][m] / 2.   One block per head, thread = m.
__global__ __launch_bounds__(256) void pe_fold_adj_kernel(const float* __restrict__ Q, const float* __restrict__ Wk,
                                                          const float* __restrict__ gs0, const float* __restrict__ pe,
                                                          float* __restrict__ gWk, float* __restrict__ gQ, int BT) {
    __shared__ float red[4];
    const int h = blockIdx.x, m = threadIdx.x, lane = m & 63, wave = m >> 6;
    float P = 0.f;
    for (int bt = 0; bt < BT; ++bt) P = fmaf(gs0[(size_t)bt * NH + h], pe[(size_t)bt * DM + m], P);
    for (int d = 0; d < DK; ++d) {
        const int hd = h * DK + d;
        gWk[(size_t)hd * DM + m] += 0.5f * Q[hd] * P;
        const float v = wave_sum(P * Wk[(size_t)hd * DM + m]);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (m == 0) gQ[hd] += 0.5f * ((red[0] + red[1]) + (red[2] + red[3]));
        __syncthreads();
    }
}

// AbsolutePositionalEncoder adjoint: gW[j][day] = sum over (b,t) with that day of sum_h g_pe[bt,16h+j]; gb[j] = sum of all.
// One block per j, thread = day (384 threads): fixed order, no atomics.
__global__ __launch_bounds__(384) void pe_abs_bwd_kernel(const long long* __restrict__ days, const float* __restrict__ gpe,
                                                         float* __restrict__ gW, float* __restrict__ gb, int BT) {
    __shared__ float red[6];
    const int j = blockIdx.x, d = threadIdx.x, lane = d & 63, wave = d >> 6;
    float acc = 0.f;
    for (int bt = 0; bt < BT; ++bt) {
        long long day = days[bt];
        day = day < 0 ? 0 : (day >= NDAY ? NDAY - 1 : day);
        if (day == d) {
            float s = 0.f;
            for (int h = 0; h < NH; ++h) s += gpe[(size_t)bt * DM + h * DV + j];
            acc += s;
        }
    }
    if (d < NDAY) gW[(size_t)j * NDAY + d] = acc;
    const float v = wave_sum(d < NDAY ? acc : 0.f);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (d == 0) gb[j] = ((red[0] + red[1]) + (red[2] + red[3])) + (red[4] + red[5]);
}

// Linear(256,256) adjoint: gW[o][i] = sum_bt g_pe[bt,o] sin256[bt,i]; gb[o] = sum_bt g_pe[bt,o].  Block per o, thread = i.
__global__ __launch_bounds__(256) void pe_lin_bwd_kernel(const float* __restrict__ gpe, const float* __restrict__ sin256,
                                                         float* __restrict__ gW, float* __restrict__ gb, int BT) {
    const int o = blockIdx.x, i = threadIdx.x;
    float acc = 0.f, sb = 0.f;
    for (int bt = 0; bt < BT; ++bt) {
        const float g = gpe[(size_t)bt * DM + o];
        acc = fmaf(g, sin256[(size_t)bt * DM + i], acc);
        sb += g;
    }
    gW[(size_t)o * DM + i] = acc;
    if (i == 0) gb[o] = sb;
}

}  // namespace

extern "C" int c2s_ltae_pe_table(int mode, const long long* dates0, const long long* dates1, float period, const float* W,
                                 const float* b, float* pe, float* sin256, int* bad_days, int BT, void* stream) {
    C2S_REQUIRE(mode >= 1 && mode <= 3 && dates0 && W && b && pe && bad_days && BT > 0, "ltae_pe_table: bad args");
    C2S_REQUIRE(mode != 2 || dates1, "ltae_pe_table: abs_rel needs the day-of-year dates");
    C2S_REQUIRE(mode != 3 || sin256, "ltae_pe_table: linear mode saves the tiled sinusoid for the adjoint");
    hipLaunchKernelGGL(pe_table_kernel, dim3(BT), dim3(256), 0, (hipStream_t)stream, mode, dates0, dates1, period, W, b, pe,
                       sin256, bad_days);
    C2S_CHECK_LAUNCH("ltae_pe_table");
    return C2S_OK;
}

extern "C" int c2s_ltae_pe_abs_add(const long long* dates1, const float* W2, const float* b2, float* pe, int* bad_days, int BT,
                                   void* stream) {
    C2S_REQUIRE(dates1 && W2 && b2 && pe && bad_days && BT > 0, "ltae_pe_abs_add: bad args");
    hipLaunchKernelGGL(pe_abs_add_kernel, dim3(BT), dim3(256), 0, (hipStream_t)stream, dates1, W2, b2, pe, bad_days);
    C2S_CHECK_LAUNCH("ltae_pe_abs_add");
    return C2S_OK;
}

extern "C" int c2s_ltae_pe_abs_bwd(const long long* dates1, const float* g_pe, float* gW2, float* gb2, int BT, void* stream) {
    C2S_REQUIRE(dates1 && g_pe && gW2 && gb2 && BT > 0, "ltae_pe_abs_bwd: bad args");
    hipLaunchKernelGGL(pe_abs_bwd_kernel, dim3(DV), dim3(384), 0, (hipStream_t)stream, dates1, g_pe, gW2, gb2, BT);
    C2S_CHECK_LAUNCH("ltae_pe_abs_bwd");
    return C2S_OK;
}

extern "C" int c2s_ltae_pe_fwd(const float* qwk, const float* pe, const float* attn, float* s0, float* emb, int B, int T,
                               int HW, int phase, void* stream) {
    C2S_REQUIRE(pe && B > 0 && T > 0 && HW > 0, "ltae_pe_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (phase == 0) {           // before the attention kernel: scores
        C2S_REQUIRE(qwk && s0, "ltae_pe_fwd: null pointer");
        hipLaunchKernelGGL(pe_s0_add_kernel, dim3(cdiv((long)B * T * NH, 256)), dim3(256), 0, st, qwk, pe, s0, B * T);
        C2S_CHECK_LAUNCH("ltae_pe_s0_add");
    } else if (emb != nullptr) {  // after it: the embedding
        C2S_REQUIRE(attn, "ltae_pe_fwd: null pointer");
        hipLaunchKernelGGL(pe_emb_add_kernel, dim3(cdiv(HW, 256), B * DM), dim3(256), 0, st, attn, pe, emb, B, T, HW);
        C2S_CHECK_LAUNCH("ltae_pe_emb_add");
    }
    return C2S_OK;
}

extern "C" int c2s_ltae_pe_gattn(const float* g_emb, const float* pe, const float* g_attn_in, float* g_attn_out, int B, int T,
                                 int HW, void* stream) {
    C2S_REQUIRE(g_emb && pe && g_attn_out && B > 0 && T > 0 && HW > 0, "ltae_pe_gattn: bad args");
    C2S_REQUIRE((long)NH * B * T <= 65535, "ltae_pe_gattn: 16*B*T > 65535");
    hipLaunchKernelGGL(pe_gattn_kernel, dim3(cdiv(HW, 256), NH * B * T), dim3(256), 0, (hipStream_t)stream, g_emb, pe, g_attn_in,
                       g_attn_out, B, T, HW);
    C2S_CHECK_LAUNCH("ltae_pe_gattn");
    return C2S_OK;
}

extern "C" int c2s_ltae_pe_bwd(int mode, const long long* dates0, const long long* dates1, const float* Q, const float* Wk,
                               const float* qwk, const float* pe, const float* sin256, const float* attn, const float* g_emb,
                               const float* gs0, float* g_pe, float* gWk, float* gQ, float* gW, float* gb, int B, int T, int HW,
                               void* stream) {
    C2S_REQUIRE(mode >= 1 && mode <= 3 && dates0 && Q && Wk && qwk && pe && attn && gs0 && g_pe && gWk && gQ && gW && gb,
                "ltae_pe_bwd: null pointer");
    C2S_REQUIRE(mode != 2 || dates1, "ltae_pe_bwd: abs_rel needs the day-of-year dates");
    C2S_REQUIRE(mode != 3 || sin256, "ltae_pe_bwd: linear mode needs the saved sinusoid");
    hipStream_t st = (hipStream_t)stream;
    const int BT = B * T;
    hipLaunchKernelGGL(pe_grad_kernel, dim3(BT * NH), dim3(256), 0, st, attn, g_emb, gs0, qwk, g_pe, B, T, HW);
    C2S_CHECK_LAUNCH("ltae_pe_grad");
    hipLaunchKernelGGL(pe_fold_adj_kernel, dim3(NH), dim3(256), 0, st, Q, Wk, gs0, pe, gWk, gQ, BT);
    C2S_CHECK_LAUNCH("ltae_pe_fold_adj");
    if (mode == 3) hipLaunchKernelGGL(pe_lin_bwd_kernel, dim3(DM), dim3(256), 0, st, g_pe, sin256, gW, gb, BT);
    else hipLaunchKernelGGL(pe_abs_bwd_kernel, dim3(DV), dim3(384), 0, st, mode == 1 ? dates0 : dates1, g_pe, gW, gb, BT);
    C2S_CHECK_LAUNCH("ltae_pe_param_bwd");
    return C2S_OK;
}
