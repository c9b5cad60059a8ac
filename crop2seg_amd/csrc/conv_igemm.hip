// Implicit-GEMM convolution for gfx950 on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// GEMM view (per frame n):   D[cout][pos] = sum_{tap, cin} Wpk[tap][cin][cout] * In[cin][gather(pos, tap)]
//   A operand (MFMA rows i) = output channel  -> accumulator rows are channels,
//   B operand (MFMA cols j) = output position -> lanes 0..31 of a store hit 32 consecutive x of one
//                                               NCHW row: 128-byte coalesced stores.
// Workgroup = 256 threads = 4 waves.  Tile = 8 position fragments of 32 positions (FR rows x FC cols,
// FC = 2^log2fc adapts to narrow planes: 128 -> 1x32, 16 -> 2x16) x COT = 32*MF output channels.
// Wave w owns position fragments 2w, 2w+1 and all MF channel fragments: MF*2 accumulators of 16 VGPRs.
// K loop: input channels in chunks of CK staged in LDS together with the [tap][CK][COT] weight slab;
// the gather offsets of a thread's staging elements are computed once per tile (reflection included)
// and kept in registers, so the per-chunk staging is load + ds_write only.
//
// Reference call sites replaced: nn.Conv2d / nn.ConvTranspose2d in src/backbones/conv.py:70-80,
// 263-271,378-390 and their convolution_backward-input.
#include "common.h"

namespace {

struct ConvParams {
    const float* src0;
    const float* src1;
    const float* wpk;
    const float* bias;
    float* out;
    const int* valid;
    int C0, C1, Hin, Win, Cout, CoutP, Hout, Wout, OutH, OutW;
    int pad_y, pad_x, pad_mode, osy, osx, ooy, oox, accumulate;
    int log2fc, tiles_x;
    // reflect-adjoint rules (data gradient of a reflect-padded convolution): for each dimension up to two rules
    // (output position, tap, extra input index): the B operand of (position, tap) additionally reads `extra`.
    int adj;            // 0 = off
    int ay_pos[2], ay_tap[2], ay_src[2];
    int ax_pos[2], ax_tap[2], ax_src[2];
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int plane_for(int K, int S, int log2fc) {
    return ((8 * (32 >> log2fc) - 1) * S + K) * (((1 << log2fc) - 1) * S + K);
}
constexpr int max_plane(int K, int S) {
    return cmax(cmax(plane_for(K, S, 2), plane_for(K, S, 3)), cmax(plane_for(K, S, 4), plane_for(K, S, 5)));
}

template <int K, int S>
struct Cfg {
    static constexpr int CK = (S == 2) ? 8 : 16;
    static constexpr int NT = K * K;
    static constexpr int MAXE = (CK * max_plane(K, S) + 255) / 256;
};

template <int K, int S, int MF, bool ADJ>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
    constexpr int CK = Cfg<K, S>::CK;
    constexpr int NT = Cfg<K, S>::NT;
    constexpr int MAXE = Cfg<K, S>::MAXE;
    constexpr int COT = 32 * MF;
    extern __shared__ float lds[];

    const int n = blockIdx.z;
    if (p.valid != nullptr && p.valid[n] == 0) return;

    const int FC = 1 << p.log2fc, FR = 32 >> p.log2fc;
    const int tile_h = 8 * FR, tile_w = FC;
    const int rows = (tile_h - 1) * S + K, cols = (tile_w - 1) * S + K;
    const int plane = rows * cols;
    float* Xl = lds;                 // [CK][plane]
    float* Wl = lds + CK * plane;    // [NT][CK][COT]

    const int tyi = blockIdx.x / p.tiles_x, txi = blockIdx.x % p.tiles_x;
    const int oy0 = tyi * tile_h, ox0 = txi * tile_w;
    const int co0 = blockIdx.y * COT;
    const int tid = threadIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HWin = p.Hin * p.Win;

    // ---- gather offsets of this thread's staging elements (bits 0..19 spatial, 20..25 channel, 31 = zero)
    int goff[MAXE];
    const int total = CK * plane;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = tid + i * 256;
        int pk = (int)0x80000000;
        if (e < total) {
            const int c = e / plane;
            const int rem = e - c * plane;
            const int r = rem / cols;
            const int cc = rem - r * cols;
            int gy = oy0 * S - p.pad_y + r;
            int gx = ox0 * S - p.pad_x + cc;
            bool ok;
            if (p.pad_mode == C2S_PAD_REFLECT) {
                ok = gy >= -p.pad_y && gy < p.Hin + p.pad_y && gx >= -p.pad_x && gx < p.Win + p.pad_x;
                gy = reflect_idx(gy, p.Hin);
                gx = reflect_idx(gx, p.Win);
            } else {
                ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
            }
            if (ok) pk = (c << 20) | (gy * p.Win + gx);
            else pk = (int)0x80000000 | (c << 20);
        }
        goff[i] = pk;
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int fy = li >> p.log2fc, fx = li & (FC - 1);
    int boff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = 2 * wave + q;
        boff[q] = lk * plane + ((f * FR + fy) * S) * cols + fx * S;
    }
    const int aoff = lk * COT + li;

    // reflect adjoint: LDS row / column of the extra input of each (fragment, rule), or -1
    int ady[2][2], adx[2][2];
    bool wave_adj = false;
    if constexpr (ADJ) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = 2 * wave + q;
            const int oy = oy0 + f * FR + fy, ox = ox0 + fx;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                ady[q][r] = (oy == p.ay_pos[r]) ? (p.ay_src[r] - oy0 * S + p.pad_y) : -1;
                adx[q][r] = (ox == p.ax_pos[r]) ? (p.ax_src[r] - ox0 * S + p.pad_x) : -1;
                any = any || ady[q][r] >= 0 || adx[q][r] >= 0;
            }
        }
        wave_adj = __any(any);
    }

    f32x16 acc[MF][2];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    const float* s0n = p.src0 + (size_t)n * p.C0 * HWin;
    const float* s1n = p.src1 != nullptr ? p.src1 + (size_t)n * p.C1 * HWin : nullptr;

    for (int cb = 0; cb < Cin; cb += CK) {
        // ---- stage input chunk
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int e = tid + i * 256;
            if (e < total) {
                const int pk = goff[i];
                const int cg = cb + ((pk >> 20) & 63);
                float v = 0.f;
                if (pk >= 0 && cg < Cin) {
                    const float* s = cg < p.C0 ? s0n + (size_t)cg * HWin : s1n + (size_t)(cg - p.C0) * HWin;
                    v = s[pk & 0xFFFFF];
                }
                Xl[e] = v;
            }
        }
        // ---- stage weight slab [NT][CK][COT] (float4 along cout)
        {
            constexpr int V = COT / 4;
            constexpr int NV = NT * CK * V;
            for (int e = tid; e < NV; e += 256) {
                const int o4 = e % V;
                const int tc = e / V;
                const int c = tc % CK, t = tc / CK;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (cb + c < Cin)
                    v = *reinterpret_cast<const f32x4*>(p.wpk + ((size_t)t * Cin + cb + c) * p.CoutP + co0 + o4 * 4);
                *reinterpret_cast<f32x4*>(Wl + (size_t)tc * COT + o4 * 4) = v;
            }
        }
        __syncthreads();
        // ---- MFMA
        if (!ADJ || !wave_adj) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int ky = t / K, kx = t % K;
                const int bt = ky * cols + kx;
#pragma unroll
                for (int cp = 0; cp < CK / 2; ++cp) {
                    float a[MF], b[2];
#pragma unroll
                    for (int m = 0; m < MF; ++m) a[m] = Wl[(t * CK + 2 * cp) * COT + aoff + m * 32];
#pragma unroll
                    for (int q = 0; q < 2; ++q) b[q] = Xl[boff[q] + 2 * cp * plane + bt];
#pragma unroll
                    for (int m = 0; m < MF; ++m)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[q], acc[m][q], 0, 0, 0);
                }
            }
        } else {
            // waves that own border positions: B(position, tap) = sum over {normal, extra row} x {normal, extra col};
            // branch-free (unused extras read the normal address with weight 0); tap loop kept rolled to bound registers
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                const int ky = t / K, kx = t % K;
                const int bt = ky * cols + kx;
                int o_r[2], o_c[2], o_rc[2];
                float m_r[2], m_c[2], m_rc[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f = 2 * wave + q;
                    int ey = -1, ex = -1;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (p.ay_tap[r] == ky && ady[q][r] >= 0) ey = ady[q][r];
                        if (p.ax_tap[r] == kx && adx[q][r] >= 0) ex = adx[q][r];
                    }
                    const int nrow = (f * FR + fy) * S + ky, ncol = fx * S + kx;
                    const int base = lk * plane;
                    o_r[q] = base + (ey >= 0 ? ey : nrow) * cols + ncol;
                    o_c[q] = base + nrow * cols + (ex >= 0 ? ex : ncol);
                    o_rc[q] = base + (ey >= 0 ? ey : nrow) * cols + (ex >= 0 ? ex : ncol);
                    m_r[q] = ey >= 0 ? 1.f : 0.f;
                    m_c[q] = ex >= 0 ? 1.f : 0.f;
                    m_rc[q] = (ey >= 0 && ex >= 0) ? 1.f : 0.f;
                }
#pragma unroll
                for (int cp = 0; cp < CK / 2; ++cp) {
                    float a[MF], b[2];
#pragma unroll
                    for (int m = 0; m < MF; ++m) a[m] = Wl[(t * CK + 2 * cp) * COT + aoff + m * 32];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float v = Xl[boff[q] + 2 * cp * plane + bt];
                        v = fmaf(m_r[q], Xl[o_r[q] + 2 * cp * plane], v);
                        v = fmaf(m_c[q], Xl[o_c[q] + 2 * cp * plane], v);
                        v = fmaf(m_rc[q], Xl[o_rc[q] + 2 * cp * plane], v);
                        b[q] = v;
                    }
#pragma unroll
                    for (int m = 0; m < MF; ++m)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[q], acc[m][q], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: row (channel) = (r&3) + 8*(r>>2) + 4*lk ; col (position) = li
    const size_t outHW = (size_t)p.OutH * p.OutW;
    float* on = p.out + (size_t)n * p.Cout * outHW;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = 2 * wave + q;
        const int oy = oy0 + f * FR + fy, ox = ox0 + fx;
        if (oy >= p.Hout || ox >= p.Wout) continue;
        const size_t sp = (size_t)(oy * p.osy + p.ooy) * p.OutW + (ox * p.osx + p.oox);
#pragma unroll
        for (int m = 0; m < MF; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < p.Cout) {
                    float v = acc[m][q][r];
                    if (p.bias != nullptr) v += p.bias[co];
                    float* dst = on + (size_t)co * outHW + sp;
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

struct TapTable {
    int off[16];
};

__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ wpk, int cin, int cout,
                                    int coutP, int ntaps, long so, long sc, TapTable tt) {
    const long total = (long)ntaps * cin * coutP;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int o = (int)(e % coutP);
        const long tc = e / coutP;
        const int c = (int)(tc % cin), t = (int)(tc / cin);
        wpk[e] = o < cout ? src[o * so + c * sc + tt.off[t]] : 0.f;
    }
}

template <int K, int S, int MF, bool ADJ>
int launch_conv(const ConvParams& p, int N, int tiles, hipStream_t st) {
    constexpr int CK = Cfg<K, S>::CK;
    constexpr int NT = Cfg<K, S>::NT;
    const int FC = 1 << p.log2fc, FR = 32 >> p.log2fc;
    const int rows = (8 * FR - 1) * S + K, cols = (FC - 1) * S + K;
    const size_t lds = ((size_t)CK * rows * cols + (size_t)NT * CK * 32 * MF) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<K, S, MF, ADJ>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(tiles, p.CoutP / (32 * MF), N);
    hipLaunchKernelGGL((conv_igemm_kernel<K, S, MF, ADJ>), grid, dim3(256), lds, st, p);
    C2S_CHECK_LAUNCH("conv_igemm");
    return C2S_OK;
}

}  // namespace

extern "C" int c2s_pack_weights(const float* src, float* wpk, int cin, int cout, int coutP, int ntaps, long stride_o,
                                long stride_c, const int* host_tap_off, void* stream) {
    C2S_REQUIRE(src && wpk && host_tap_off, "pack_weights: null pointer");
    C2S_REQUIRE(ntaps >= 1 && ntaps <= 16 && coutP % 32 == 0 && coutP >= cout, "pack_weights: bad sizes");
    TapTable tt;
    for (int i = 0; i < 16; ++i) tt.off[i] = i < ntaps ? host_tap_off[i] : 0;
    const long total = (long)ntaps * cin * coutP;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, wpk, cin, cout,
                       coutP, ntaps, stride_o, stride_c, tt);
    C2S_CHECK_LAUNCH("pack_weights");
    return C2S_OK;
}

extern "C" int c2s_conv_igemm(const c2s_conv_desc* d, const float* src0, const float* src1, const float* wpk,
                              const float* bias, float* out, const int* valid, void* stream) {
    C2S_REQUIRE(d && src0 && wpk && out, "conv_igemm: null pointer");
    C2S_REQUIRE(d->N > 0 && d->C0 > 0 && d->C1 >= 0 && (d->C1 == 0 || src1), "conv_igemm: bad channels");
    C2S_REQUIRE(d->CoutP % 32 == 0 && d->CoutP >= d->Cout && d->Cout > 0, "conv_igemm: CoutP must be a multiple of 32");
    C2S_REQUIRE(d->KH == d->KW, "conv_igemm: square kernels only");
    C2S_REQUIRE((long)d->Hin * d->Win < (1 << 20), "conv_igemm: input plane too large");
    C2S_REQUIRE(d->Hout > 0 && d->Wout > 0, "conv_igemm: empty output");
    C2S_REQUIRE((d->Hout - 1) * d->osy + d->ooy < d->OutH && (d->Wout - 1) * d->osx + d->oox < d->OutW,
                "conv_igemm: output placement exceeds the output plane");
    if (d->pad_mode == C2S_PAD_REFLECT)
        C2S_REQUIRE(d->pad_y < d->Hin && d->pad_x < d->Win, "conv_igemm: reflect padding needs pad < size");
    ConvParams p;
    p.src0 = src0; p.src1 = src1; p.wpk = wpk; p.bias = bias; p.out = out; p.valid = valid;
    p.C0 = d->C0; p.C1 = d->C1; p.Hin = d->Hin; p.Win = d->Win; p.Cout = d->Cout; p.CoutP = d->CoutP;
    p.Hout = d->Hout; p.Wout = d->Wout; p.OutH = d->OutH; p.OutW = d->OutW;
    p.pad_y = d->pad_y; p.pad_x = d->pad_x; p.pad_mode = d->pad_mode;
    p.osy = d->osy; p.osx = d->osx; p.ooy = d->ooy; p.oox = d->oox; p.accumulate = d->accumulate;
    p.adj = d->reflect_adjoint;
    for (int r = 0; r < 2; ++r) { p.ay_pos[r] = p.ax_pos[r] = -1; p.ay_tap[r] = p.ax_tap[r] = -1; p.ay_src[r] = p.ax_src[r] = 0; }
    if (d->reflect_adjoint) {
        C2S_REQUIRE(d->pad_mode == C2S_PAD_ZEROS && d->S == 1 && (d->KH == 3 || d->KH == 2),
                    "conv_igemm: reflect_adjoint applies to the zero-padded 3x3 / 2x2-parity data-gradient launches");
        C2S_REQUIRE(d->Hin >= 2 && d->Win >= 2 && d->Hin != 3 && d->Win != 3, "conv_igemm: reflect_adjoint needs planes of 2 or >= 4");
        if (d->KH == 3) {
            // dgrad of conv3x3(reflect pad 1): g_x[1] += w[0] g_y[0] ; g_x[H-2] += w[2] g_y[H-1]  (flipped taps 2 / 0)
            C2S_REQUIRE(d->pad_y == 1 && d->pad_x == 1 && d->Hin == d->Hout && d->Win == d->Wout, "conv_igemm: bad 3x3 adjoint geometry");
            p.ay_pos[0] = 1; p.ay_tap[0] = 2; p.ay_src[0] = 0;
            p.ay_pos[1] = d->Hout - 2; p.ay_tap[1] = 0; p.ay_src[1] = d->Hin - 1;
            p.ax_pos[0] = 1; p.ax_tap[0] = 2; p.ax_src[0] = 0;
            p.ax_pos[1] = d->Wout - 2; p.ax_tap[1] = 0; p.ax_src[1] = d->Win - 1;
        } else {
            // parity sub-kernel of the dgrad of conv4x4s2(reflect pad 1); parity = 1 - pad
            if (d->pad_y == 0) { p.ay_pos[0] = 0; p.ay_tap[0] = 1; p.ay_src[0] = 0; }
            else { p.ay_pos[0] = d->Hout - 1; p.ay_tap[0] = 0; p.ay_src[0] = d->Hin - 1; }
            if (d->pad_x == 0) { p.ax_pos[0] = 0; p.ax_tap[0] = 1; p.ax_src[0] = 0; }
            else { p.ax_pos[0] = d->Wout - 1; p.ax_tap[0] = 0; p.ax_src[0] = d->Win - 1; }
        }
    }
    int l2 = 5;
    while (l2 > 2 && (1 << l2) > d->Wout) --l2;
    p.log2fc = l2;
    const int FC = 1 << l2, FR = 32 >> l2;
    p.tiles_x = cdiv(d->Wout, FC);
    const int tiles = p.tiles_x * cdiv(d->Hout, 8 * FR);
    hipStream_t st = (hipStream_t)stream;
    const bool wide = d->CoutP % 64 == 0;
#define C2S_DISPATCH(K_, S_, A_)                                        \
    if (d->KH == K_ && d->S == S_ && (p.adj != 0) == A_)                \
        return wide ? launch_conv<K_, S_, 2, A_>(p, d->N, tiles, st) : launch_conv<K_, S_, 1, A_>(p, d->N, tiles, st);
    C2S_DISPATCH(3, 1, false)
    C2S_DISPATCH(3, 1, true)
    C2S_DISPATCH(1, 1, false)
    C2S_DISPATCH(2, 1, false)
    C2S_DISPATCH(2, 1, true)
    C2S_DISPATCH(4, 2, false)
#undef C2S_DISPATCH
    c2s_set_error("conv_igemm: unsupported (K=%d,S=%d)", d->KH, d->S);
    return C2S_EINVAL;
}
