// 3x3 stride-1 convolution of a few input channels (the first layer: 10 Sentinel-2 bands -> 64 features at full
// resolution) on the exact-f32 MFMA.
//
// With 10 input channels the generic implicit-GEMM kernel spends its time staging 4-channel chunks (three LDS
// round trips + barriers per tile for 90 k-values).  Here the whole reduction (9 taps x NP channel pairs) is unrolled:
//   * the weights of the wave's 64 output channels stay in registers for the life of the (persistent) workgroup:
//     9*NP*2 VGPRs, read once from wpk[tap][cin][coutP] (the layout of c2s_pack_weights);
//   * one tile = 8 output rows x 32 columns of one frame; its input patch (2NP channels x 10 rows x 34 columns,
//     padding applied while staging) is written to LDS once; the next tile's patch is prefetched into registers while
//     the MFMAs of the current tile run;
//   * wave w owns output rows 2w, 2w+1: per (tap, channel pair) one ds_read_b32 per row feeds two MFMAs (both halves
//     of the 64 channels); A = weights (rows = output channel), B = input (columns = pixel), as in conv_igemm.hip, so a
//     half-wave stores 128 contiguous bytes of an output row.
// Channel stride of the patch is padded to == 32 (mod 64) words so the two k-halves of a B read hit disjoint banks.
//
// Reference call site replaced: the first nn.Conv2d of ConvLayer / ConvBlock in_conv (src/backbones/conv.py:70-80
// through utae.py:64-71, timeunet_v1.py, wtae.py) with reflect or zero padding.
#include "common.h"

namespace {

struct FirstParams {
    const float* src;
    const float* wpk;      // [tap 9][Cin][CoutP]
    const float* bias;
    float* out;
    const int* valid;
    int N, Cin, H, W, Cout, CoutP;
    int tiles_x, tiles_y, ntiles;
};

constexpr int FT_ROWS = 8, FT_COLS = 32;
constexpr int FT_PR = FT_ROWS + 2, FT_PC = FT_COLS + 2;          // patch rows / columns
constexpr int ft_chs() {                                            // channel stride: >= PR*PC and == 32 (mod 64)
    int v = FT_PR * FT_PC;
    while (v % 64 != 32) ++v;
    return v;
}
constexpr int FT_CHS = ft_chs();

template <int NP, bool REFLECT>
__global__ __launch_bounds__(256, 2) void conv_first_kernel(FirstParams p) {
    constexpr int NCH = 2 * NP;
    constexpr int NE = NCH * FT_PR * FT_PC;                         // patch elements
    constexpr int EPT = (NE + 255) / 256;
    __shared__ float patch[NCH * FT_CHS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, kk = lane >> 5;
    const int cb = blockIdx.y * 64;

    // weights of this wave's 64 output channels: wreg[tap][s][half] = A[i = cout][k = kk] of that k-step
    float wreg[9][NP][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const int ch = 2 * s + kk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float v = p.wpk[((size_t)t * p.Cin + (ch < p.Cin ? ch : 0)) * p.CoutP + cb + h * 32 + px];
                wreg[t][s][h] = ch < p.Cin ? v : 0.f;
            }
        }
    __shared__ float bias_s[64];
    if (tid < 64) bias_s[tid] = (p.bias != nullptr && cb + tid < p.Cout) ? p.bias[cb + tid] : 0.f;

    auto load_patch = [&](int tile, float (&v)[EPT]) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                               // keep the per-element index math out of the registers
                                                                    // that live across tiles (it is recomputed per tile)
        const int n = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - n * (p.tiles_x * p.tiles_y);
        const int y0 = (rem / p.tiles_x) * FT_ROWS - 1, x0 = (rem % p.tiles_x) * FT_COLS - 1;
        const float* base = p.src + (size_t)n * p.Cin * p.H * p.W;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int e = tid + 256 * j;
            const int ch = e / (FT_PR * FT_PC), r2 = e - ch * (FT_PR * FT_PC);
            const int row = r2 / FT_PC, col = r2 - row * FT_PC;
            int gy = y0 + row, gx = x0 + col;
            bool ok = e < NE && ch < p.Cin;
            if (REFLECT) {
                gy = reflect_idx(gy, p.H); gx = reflect_idx(gx, p.W);
            } else {
                ok = ok && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                gy = min(max(gy, 0), p.H - 1); gx = min(max(gx, 0), p.W - 1);
            }
            const int chc = min(ch, p.Cin - 1);                     // always a legal address; the value is masked below
            const float t = base[((size_t)chc * p.H + gy) * p.W + gx];
            v[j] = ok ? t : 0.f;
        }
    };
    auto store_patch = [&](const float (&v)[EPT]) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int e = tid + 256 * j;
            const int ch = e / (FT_PR * FT_PC), r2 = e - ch * (FT_PR * FT_PC);
            if (e < NE) patch[ch * FT_CHS + r2] = v[j];
        }
    };
    auto frame_ok = [&](int tile) {
        return p.valid == nullptr || p.valid[tile / (p.tiles_x * p.tiles_y)] != 0;
    };

    int tile = blockIdx.x;
    float pre[EPT];
    if (tile < p.ntiles && frame_ok(tile)) load_patch(tile, pre);
    for (; tile < p.ntiles; tile += gridDim.x) {
        const bool ok = frame_ok(tile);                             // uniform over the workgroup
        const int nxt = tile + gridDim.x;
        if (ok) {
            __syncthreads();                                        // the previous tile's reads are done
            store_patch(pre);
            __syncthreads();
        }
        if (nxt < p.ntiles && frame_ok(nxt)) load_patch(nxt, pre);  // in flight during the MFMAs below
        if (!ok) continue;                                          // padded frame: the norm pass fills it
        f32x16 acc[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][h][i] = 0.f;
        const float* pb = patch + kk * FT_CHS + (2 * wave) * FT_PC + px;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int s = 0; s < NP; ++s) {
                    const float b0 = pb[2 * s * FT_CHS + ky * FT_PC + kx];
                    const float b1 = pb[2 * s * FT_CHS + (ky + 1) * FT_PC + kx];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[ky * 3 + kx][s][0], b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[ky * 3 + kx][s][1], b0, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[ky * 3 + kx][s][0], b1, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[ky * 3 + kx][s][1], b1, acc[1][1], 0, 0, 0);
                }
        // stores: wave-uniform 64-bit row pointers + one 32-bit per-lane offset (channel 4*kk of the group, column px)
        const int n = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - n * (p.tiles_x * p.tiles_y);
        const int y = (rem / p.tiles_x) * FT_ROWS + 2 * wave, x0 = (rem % p.tiles_x) * FT_COLS;
        const size_t plane = (size_t)p.H * p.W;
        float* ub = p.out + ((size_t)n * p.Cout + cb) * plane + (size_t)y * p.W + x0;
        const unsigned lo = (unsigned)(4 * kk) * (unsigned)plane + (unsigned)px;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int og = h * 32 + (i & 3) + 8 * (i >> 2);                 // Cout % 64 == 0: every channel exists
                float* op = ub + (size_t)og * plane;
                const float b = bias_s[og + 4 * kk];
                op[lo] = acc[0][h][i] + b;
                op[lo + p.W] = acc[1][h][i] + b;
            }
    }
}

}  // namespace

extern "C" int c2s_conv3x3_smallcin_supported(const c2s_conv_desc* d) {
    return d && d->KH == 3 && d->KW == 3 && d->S == 1 && d->pad_y == 1 && d->pad_x == 1 && d->C1 == 0 && d->C0 >= 1 &&
           d->C0 <= 10 && d->Cout % 64 == 0 && d->CoutP == d->Cout && d->Hin % FT_ROWS == 0 && d->Win % FT_COLS == 0 && d->Hout == d->Hin &&
           d->Wout == d->Win && d->OutH == d->Hin && d->OutW == d->Win && d->osy == 1 && d->osx == 1 && d->ooy == 0 &&
           d->oox == 0 && !d->accumulate && !d->reflect_adjoint;
}

extern "C" int c2s_conv3x3_smallcin(const c2s_conv_desc* d, const float* src, const float* wpk, const float* bias,
                                    float* out, const int* valid, void* stream) {
    C2S_REQUIRE(d && src && wpk && out, "conv3x3_smallcin: null pointer");
    C2S_REQUIRE(c2s_conv3x3_smallcin_supported(d),
                "conv3x3_smallcin: needs 3x3/s1/pad1, one source of <= 10 channels, Cout %% 64 == 0, H %% 8 == 0, W %% 32 == 0");
    if (d->pad_mode == C2S_PAD_REFLECT) C2S_REQUIRE(d->Hin >= 2 && d->Win >= 2, "conv3x3_smallcin: reflect needs planes >= 2x2");
    FirstParams p;
    p.src = src; p.wpk = wpk; p.bias = bias; p.out = out; p.valid = valid;
    p.N = d->N; p.Cin = d->C0; p.H = d->Hin; p.W = d->Win; p.Cout = d->Cout; p.CoutP = d->CoutP;
    p.tiles_x = d->Win / FT_COLS; p.tiles_y = d->Hin / FT_ROWS;
    const long nt = (long)d->N * p.tiles_x * p.tiles_y;
    C2S_REQUIRE(nt < (1L << 30), "conv3x3_smallcin: too many tiles");
    p.ntiles = (int)nt;
    const int cus = c2s_cus();
    const int wgs = (int)(nt < 2L * cus ? nt : 2L * cus);
    const dim3 grid(wgs, d->CoutP / 64);
    hipStream_t st = (hipStream_t)stream;
    const bool reflect = d->pad_mode == C2S_PAD_REFLECT;
    if (d->C0 <= 4) {
        if (reflect) hipLaunchKernelGGL((conv_first_kernel<2, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_first_kernel<2, false>), grid, dim3(256), 0, st, p);
    } else {
        if (reflect) hipLaunchKernelGGL((conv_first_kernel<5, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_first_kernel<5, false>), grid, dim3(256), 0, st, p);
    }
    C2S_CHECK_LAUNCH("conv_first");
    return C2S_OK;
}
