// 3x3 stride-1 pad-1 convolution (forward and data gradient) as Winograd F(2x2, 3x3) on the exact-f32 MFMA:
// 2.25x fewer multiplies than the direct implicit GEMM of conv_igemm.hip, all arithmetic in fp32.
//
//   Y(2x2 block) = At [ (G g Gt) (.) (Bt d B) ] A        d = 4x4 input patch, g = 3x3 filter
//   M[xi][nu][o][t] = sum_c U[xi][nu][c][o] * V[xi][nu][c][t]      16 independent GEMMs over the input channels
//
// MFMA mapping as in conv_igemm.hip: A operand = (transformed) weights, rows = output channel; B operand =
// (transformed) input, columns = 2x2 output blocks.  Workgroup = 4 waves; wave xi owns row xi of the 4x4
// transform domain: 4 nu x 2 channel fragments = 8 accumulators for 64 output channels x 32 blocks (fragment of
// BR x BC blocks = 4x32 output pixels at W >= 32).  The raw input tile (with halo) and the U chunk are staged in LDS
// (double-buffered, one barrier per chunk of 8 input channels); each lane transforms its own patch row pair on the
// fly (8 VALU ops per k-step next to 8 MFMAs).  After the K loop the waves exchange the nu-reduced sums through LDS
// and each lane stores 2x2 outputs as float2 pairs (128-byte row segments).
//
// Data gradient of a reflect-padded convolution: the adjoint of the reflection (gx[1] += gxp[-1], gx[H-2] += gxp[H],
// same for columns) equals a modification of the raw patch of the border blocks (top block: d3 += d1, bottom block:
// d0 += d2; left/right likewise on columns) because those patch rows enter exactly the outputs that receive the
// folded halo.  It is applied inside the transform with per-lane coefficients.
//
// Reference call sites replaced: nn.Conv2d 3x3 (src/backbones/conv.py:70-80,378-382) and its
// convolution_backward-input, including reflection_pad2d_backward (conv.py:72-79).
#include "common.h"

namespace {

struct WinoParams {
    const float* src0;
    const float* src1;
    const float* upk;      // [16][Cin][CoutP]
    const float* bias;
    float* out;
    const int* valid;
    int C0, C1, H, W, Cout, CoutP;
    int pad_mode, accumulate;
    int log2bc, tiles_x;
};

constexpr int WN_CK = 8;
constexpr int wn_plane(int l2) { return (2 * (32 >> l2) + 2) * (2 * (1 << l2) + 2); }
constexpr int wn_max(int a, int b) { return a > b ? a : b; }
constexpr int WN_MAXPLANE = wn_max(wn_max(wn_plane(2), wn_plane(3)), wn_plane(4));
constexpr int WN_MAXE = (WN_CK * WN_MAXPLANE + 255) / 256;
constexpr int WN_USLAB = 16 * WN_CK * 64;          // floats of one U chunk
constexpr int WN_EXCH = 4 * 2 * 32 * 64;           // floats of the epilogue exchange [xi][x][m*16+r][lane]

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool ADJ>
__global__ __launch_bounds__(256, 2) void conv_winograd_kernel(WinoParams p) {
    constexpr int CK = WN_CK, MAXE = WN_MAXE;
    constexpr int NWV = WN_USLAB / 4;             // float4 items of the U chunk
    constexpr int WPT = NWV / 256;
    constexpr int NS = CK / 2;                    // k-steps (channel pairs) per chunk
    extern __shared__ float lds[];

    const int n = blockIdx.z;
    if (p.valid != nullptr && p.valid[n] == 0) return;

    const int BC = 1 << p.log2bc, BR = 32 >> p.log2bc;
    const int RR = 2 * BR + 2, RC = 2 * BC + 2;
    const int plane = RR * RC;                    // even
    const int xsz = CK * plane;
    float* Xl = lds;                              // [2][ [CK][plane] | [16][CK][64] ]
    float* Wl = lds + xsz;
    const int BUF = xsz + WN_USLAB;

    const int tyi = blockIdx.x / p.tiles_x, txi = blockIdx.x % p.tiles_x;
    const int oy0 = tyi * 2 * BR, ox0 = txi * 2 * BC;
    const int co0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HW = p.H * p.W;

    int goff[MAXE];
    const int total = CK * plane;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = tid + i * 256;
        int off = -1;
        if (e < total) {
            const int c = e / plane;
            const int rem = e - c * plane;
            const int r = rem / RC;
            const int cc = rem - r * RC;
            int gy = oy0 - 1 + r, gx = ox0 - 1 + cc;
            bool ok;
            if (p.pad_mode == C2S_PAD_REFLECT) {
                ok = gy >= -1 && gy <= p.H && gx >= -1 && gx <= p.W;
                gy = reflect_idx(gy, p.H);
                gx = reflect_idx(gx, p.W);
            } else {
                ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            }
            if (ok) off = (c * HW + gy * p.W + gx) * 4;
        }
        goff[i] = off;
    }

    const int lane = tid & 63, xi = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int by = li >> p.log2bc, bx = li & (BC - 1);
    // patch rows combined by this wave: t = ca * d[ra] + cb * d[rb]   (rows of Bt)
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    float ca = 1.f, cb = xi == 1 ? 1.f : -1.f, e0 = 1.f, e3 = 1.f;
    if constexpr (ADJ) {
        const int gby = (oy0 >> 1) + by, gbx = (ox0 >> 1) + bx;      // global block coordinates
        const bool top = gby == 0, bottom = gby == (p.H >> 1) - 1;
        const bool left = gbx == 0, right = gbx == (p.W >> 1) - 1;
        if (xi == 0 && bottom) cb = 0.f;        // d0 += d2  ->  (d0 + d2) - d2
        if (xi == 3 && top) ca = 0.f;           // d3 += d1  ->  d1 - (d3 + d1)
        if (right) e0 = 0.f;                    // col0 += col2
        if (left) e3 = 0.f;                     // col3 += col1
    }
    const int boffa = lk * plane + (2 * by + ra) * RC + 2 * bx;
    const int boffb = lk * plane + (2 * by + rb) * RC + 2 * bx;
    const int aoff = (xi * 4 * CK + lk) * 64 + li;

    f32x16 acc[4][2];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][m][r] = 0.f;

    const float* s0n = p.src0 + (size_t)n * p.C0 * HW;
    const float* s1n = p.src1 != nullptr ? p.src1 + (size_t)n * p.C1 * HW : nullptr;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)s0n, 0, p.C0 * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(s1n != nullptr ? s1n : s0n), 0,
                                                                         p.C1 * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.upk, 0, 16 * Cin * p.CoutP * 4, 0x00020000);

    float xr[MAXE];
    f32x4 wr[WPT];
    auto prefetch = [&](int cb_) {
        const bool first = cb_ < p.C0;
        const int chan0 = (first ? cb_ : cb_ - p.C0) * HW * 4;
        if (first) {
#pragma unroll
            for (int i = 0; i < MAXE; ++i)
                xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, goff[i] >= 0 ? goff[i] + chan0 : -1, 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < MAXE; ++i)
                xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, goff[i] >= 0 ? goff[i] + chan0 : -1, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;             // float4 item: [xn 16][c CK][o4 16]
            const int o4 = e & 15, c = (e >> 4) & (CK - 1), xn = e >> 7;
            static_assert(CK == 8, "item decomposition assumes 8 channels per chunk");
            const bool ok = cb_ + c < Cin;
            const int off = ok ? (((xn * Cin + cb_ + c) * p.CoutP + co0 + o4 * 4) * 4) : -1;
            wr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
        }
    };
    auto commit = [&](int buf) {
        float* Xd = Xl + buf * BUF;
        float* Wd = Wl + buf * BUF;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int e = tid + i * 256;
            if (e < total) Xd[e] = xr[i];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<f32x4*>(Wd + (size_t)e * 4) = wr[i];
        }
    };

    // raw patch rows + U operands of k-step s (channels 2s, 2s+1)
    auto load_ops = [&](const float* Xc, const float* Wc, int s_, f32x2 (&d)[4], float (&a)[4][2]) {
        const float* pa = Xc + 2 * s_ * plane + boffa;
        const float* pb = Xc + 2 * s_ * plane + boffb;
        d[0] = *reinterpret_cast<const f32x2*>(pa);
        d[1] = *reinterpret_cast<const f32x2*>(pa + 2);
        d[2] = *reinterpret_cast<const f32x2*>(pb);
        d[3] = *reinterpret_cast<const f32x2*>(pb + 2);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int m = 0; m < 2; ++m) a[v][m] = Wc[aoff + (v * CK + 2 * s_) * 64 + m * 32];
    };
    auto transform = [&](const f32x2 (&d)[4], float (&V)[4]) {
        float t0, t1, t2, t3;
        if constexpr (ADJ) {
            t0 = fmaf(cb, d[2].x, ca * d[0].x);
            t1 = fmaf(cb, d[2].y, ca * d[0].y);
            t2 = fmaf(cb, d[3].x, ca * d[1].x);
            t3 = fmaf(cb, d[3].y, ca * d[1].y);
            V[0] = fmaf(-e0, t2, t0);
            V[3] = fmaf(e3, t1, -t3);
        } else {
            t0 = fmaf(cb, d[2].x, d[0].x);
            t1 = fmaf(cb, d[2].y, d[0].y);
            t2 = fmaf(cb, d[3].x, d[1].x);
            t3 = fmaf(cb, d[3].y, d[1].y);
            V[0] = t0 - t2;
            V[3] = t1 - t3;
        }
        V[1] = t1 + t2;
        V[2] = t2 - t1;
    };

    constexpr int COMMIT_AT = NS / 2;
    prefetch(0);
    commit(0);
    if (CK < Cin) prefetch(CK);
    __syncthreads();
    int cur = 0;
    for (int cb_ = 0; cb_ < Cin; cb_ += CK, cur ^= 1) {
        const float* Xc = Xl + cur * BUF;
        const float* Wc = Wl + cur * BUF;
        f32x2 d[2][4];
        float a[2][4][2];
        load_ops(Xc, Wc, 0, d[0], a[0]);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            float V[4];
            transform(d[s_ & 1], V);
            if (s_ + 1 < NS) load_ops(Xc, Wc, s_ + 1, d[(s_ + 1) & 1], a[(s_ + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[v][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][v][m], V[v], acc[v][m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s_ == COMMIT_AT && cb_ + CK < Cin) {
                commit(cur ^ 1);
                if (cb_ + 2 * CK < Cin) prefetch(cb_ + 2 * CK);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  nu side of the output transform in registers: P[x] = sum_nu M[nu] A[nu][x]
    float* ex = lds;                                 // [xi 4][x 2][m*16+r 32][lane 64]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r];
            ex[((xi * 2 + 0) * 32 + m * 16 + r) * 64 + lane] = m0 + m1 + m2;
            ex[((xi * 2 + 1) * 32 + m * 16 + r) * 64 + lane] = m1 - m2 - m3;
        }
    __syncthreads();
    // xi side: Y[0][x] = P0 + P1 + P2, Y[1][x] = P1 - P2 - P3; wave w finishes the accumulator rows r with r>>2 == w
    const int oy = oy0 + 2 * by, ox = ox0 + 2 * bx;
    if (oy >= p.H || ox >= p.W) return;
    float* on = p.out + (size_t)n * p.Cout * HW + (size_t)oy * p.W + ox;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = xi * 4 + rr;
            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (co >= p.Cout) continue;
            float P[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int x = 0; x < 2; ++x) P[q][x] = ex[((q * 2 + x) * 32 + m * 16 + r) * 64 + lane];
            f32x2 y0 = {P[0][0] + P[1][0] + P[2][0], P[0][1] + P[1][1] + P[2][1]};
            f32x2 y1 = {P[1][0] - P[2][0] - P[3][0], P[1][1] - P[2][1] - P[3][1]};
            if (p.bias != nullptr) { const float bv = p.bias[co]; y0 += bv; y1 += bv; }
            f32x2* d0 = reinterpret_cast<f32x2*>(on + (size_t)co * HW);
            f32x2* d1 = reinterpret_cast<f32x2*>(on + (size_t)co * HW + p.W);
            if (p.accumulate) { y0 += *d0; y1 += *d1; }
            *d0 = y0;
            *d1 = y1;
        }
}

struct TapTable9 {
    int off[9];
};

// U[xi][nu][c][o] = (G g Gt)[xi][nu] of the 3x3 filter g[k] = src[o*so + c*sc + tap[k]]
__global__ void pack_winograd_kernel(const float* __restrict__ src, float* __restrict__ upk, int cin, int cout, int coutP,
                                     long so, long sc, TapTable9 tt) {
    const long total = (long)cin * coutP;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int o = (int)(e % coutP), c = (int)(e / coutP);
    float g[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = o < cout ? src[o * so + c * sc + tt.off[k]] : 0.f;
    float t[4][3];                                  // G g
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = t[i][0];
        const float u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        const float u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        const float u3 = t[i][2];
        upk[((size_t)(i * 4 + 0) * cin + c) * coutP + o] = u0;
        upk[((size_t)(i * 4 + 1) * cin + c) * coutP + o] = u1;
        upk[((size_t)(i * 4 + 2) * cin + c) * coutP + o] = u2;
        upk[((size_t)(i * 4 + 3) * cin + c) * coutP + o] = u3;
    }
}

}  // namespace

extern "C" int c2s_pack_weights_winograd(const float* src, float* upk, int cin, int cout, int coutP, long stride_o,
                                         long stride_c, const int* host_tap_off, void* stream) {
    C2S_REQUIRE(src && upk && host_tap_off, "pack_weights_winograd: null pointer");
    C2S_REQUIRE(coutP % 64 == 0 && coutP >= cout && cin > 0, "pack_weights_winograd: CoutP must be a multiple of 64");
    TapTable9 tt;
    for (int i = 0; i < 9; ++i) tt.off[i] = host_tap_off[i];
    const long total = (long)cin * coutP;
    hipLaunchKernelGGL(pack_winograd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, upk, cin, cout,
                       coutP, stride_o, stride_c, tt);
    C2S_CHECK_LAUNCH("pack_weights_winograd");
    return C2S_OK;
}

extern "C" int c2s_conv3x3_winograd(const c2s_conv_desc* d, const float* src0, const float* src1, const float* upk,
                                    const float* bias, float* out, const int* valid, void* stream) {
    C2S_REQUIRE(d && src0 && upk && out, "conv3x3_winograd: null pointer");
    C2S_REQUIRE(d->N > 0 && d->C0 > 0 && d->C1 >= 0 && (d->C1 == 0 || src1), "conv3x3_winograd: bad channels");
    C2S_REQUIRE(d->KH == 3 && d->KW == 3 && d->S == 1 && d->pad_y == 1 && d->pad_x == 1, "conv3x3_winograd: 3x3 stride 1 pad 1 only");
    C2S_REQUIRE(d->Hout == d->Hin && d->Wout == d->Win && d->OutH == d->Hout && d->OutW == d->Wout && d->osy == 1 &&
                d->osx == 1 && d->ooy == 0 && d->oox == 0, "conv3x3_winograd: dense same-size output only");
    C2S_REQUIRE(d->Hin % 2 == 0 && d->Win % 2 == 0 && d->Hin >= 2 && d->Win >= 8, "conv3x3_winograd: even planes, W >= 8");
    C2S_REQUIRE(d->CoutP % 64 == 0 && d->CoutP >= d->Cout && d->Cout > 0, "conv3x3_winograd: CoutP must be a multiple of 64");
    C2S_REQUIRE(d->C1 == 0 || d->C0 % WN_CK == 0, "conv3x3_winograd: with two sources C0 must be a multiple of 8");
    C2S_REQUIRE((long)(d->C0 > d->C1 ? d->C0 : d->C1) * d->Hin * d->Win * 4 < (1L << 31), "conv3x3_winograd: frame too large");
    if (d->reflect_adjoint) C2S_REQUIRE(d->pad_mode == C2S_PAD_ZEROS, "conv3x3_winograd: the reflect adjoint is a zero-padded launch");
    WinoParams p;
    p.src0 = src0; p.src1 = src1; p.upk = upk; p.bias = bias; p.out = out; p.valid = valid;
    p.C0 = d->C0; p.C1 = d->C1; p.H = d->Hin; p.W = d->Win; p.Cout = d->Cout; p.CoutP = d->CoutP;
    p.pad_mode = d->pad_mode; p.accumulate = d->accumulate;
    int l2 = 4;
    while (l2 > 2 && (2 << l2) > d->Win) --l2;
    p.log2bc = l2;
    const int BC = 1 << l2, BR = 32 >> l2;
    p.tiles_x = cdiv(d->Win, 2 * BC);
    const int tiles = p.tiles_x * cdiv(d->Hin, 2 * BR);
    const int plane = (2 * BR + 2) * (2 * BC + 2);
    size_t fl = 2 * ((size_t)WN_CK * plane + WN_USLAB);
    if (fl < (size_t)WN_EXCH) fl = WN_EXCH;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_winograd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_winograd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(tiles, d->CoutP / 64, d->N);
    hipStream_t st = (hipStream_t)stream;
    if (d->reflect_adjoint) hipLaunchKernelGGL(conv_winograd_kernel<true>, grid, dim3(256), fl * sizeof(float), st, p);
    else hipLaunchKernelGGL(conv_winograd_kernel<false>, grid, dim3(256), fl * sizeof(float), st, p);
    C2S_CHECK_LAUNCH("conv3x3_winograd");
    return C2S_OK;
}
