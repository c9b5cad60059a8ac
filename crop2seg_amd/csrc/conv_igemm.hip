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
    // reflect adjoint (data gradient of a reflect-padded convolution).  Per dimension the fold of the halo gradient
    // is:  operand(pos == lo, tap K-1) += input[tap position - (K-1)]   and
    //      operand(pos == hi, tap 0)   += input[tap position + (K-1)]      (-1 = rule absent)
    int adj, ay_lo, ay_hi, ax_lo, ax_hi;
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
    // channels per LDS chunk: enough MFMA k-steps per barrier pair (>= 16) without growing the prefetch registers
    static constexpr int CK = K == 1 ? 16 : (K == 2 ? 8 : (K == 4 ? 2 : 4));
    static constexpr int NT = K * K;
    static constexpr int MAXE = (CK * max_plane(K, S) + 255) / 256;
};

template <int K, int S, int MF, bool ADJ>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
    constexpr int CK = Cfg<K, S>::CK;
    constexpr int NT = Cfg<K, S>::NT;
    constexpr int MAXE = Cfg<K, S>::MAXE;
    constexpr int COT = 32 * MF;
    constexpr int WV = COT / 4;                  // float4 per (tap, channel) weight row
    constexpr int NWV = NT * CK * WV;            // float4 items of the weight slab
    constexpr int WPT = (NWV + 255) / 256;
    constexpr int NS = NT * (CK / 2);            // MFMA k-steps per chunk
    extern __shared__ float lds[];

    // XCD-aware placement (workgroups go to the 8 XCDs round-robin in linear order): each XCD takes a contiguous eighth of the
    // (frame, tile) range, so tiles that share halo rows are processed behind the same L2
    int bxi = blockIdx.x, n = blockIdx.z;
    {
        const unsigned tot = gridDim.x * gridDim.z;
        if (gridDim.y == 1 && (tot & 7) == 0) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z;
            const unsigned l2_ = (lin & 7) * (tot >> 3) + (lin >> 3);
            n = (int)(l2_ / gridDim.x);
            bxi = (int)(l2_ - (unsigned)n * gridDim.x);
        }
    }
    if (p.valid != nullptr && p.valid[n] == 0) return;

    const int FC = 1 << p.log2fc, FR = 32 >> p.log2fc;
    const int tile_h = 8 * FR, tile_w = FC;
    const int rows = (tile_h - 1) * S + K, cols = (tile_w - 1) * S + K;
    const int plane = rows * cols;
    float* Xl = lds;                 // [2][ [CK][plane] | [NT][CK][COT] ]
    const int xsz = (CK * plane + 3) & ~3;
    float* Wl = lds + xsz;
    const int BUF = xsz + NT * CK * COT;

    const int tyi = bxi / p.tiles_x, txi = bxi % p.tiles_x;
    const int oy0 = tyi * tile_h, ox0 = txi * tile_w;
    const int co0 = blockIdx.y * COT;
    const int tid = threadIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HWin = p.Hin * p.Win;

    // ---- byte offsets (within a frame, relative to the chunk's first channel) of this thread's staging elements;
    //      negative = zero padding / outside the tile.  Reflection is resolved here, once per tile.
    int goff[MAXE];
    const int total = CK * plane;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = tid + i * 256;
        int off = -1;
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
            if (ok) off = (c * HWin + gy * p.Win + gx) * 4;
        }
        goff[i] = off;
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

    // reflect adjoint: per-lane masks of the border positions this lane owns
    float mxlo[2] = {0.f, 0.f}, mxhi[2] = {0.f, 0.f}, mylo[2] = {0.f, 0.f}, myhi[2] = {0.f, 0.f};
    if constexpr (ADJ) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = 2 * wave + q;
            const int oy = oy0 + f * FR + fy, ox = ox0 + fx;
            mxlo[q] = ox == p.ax_lo ? 1.f : 0.f;
            mxhi[q] = ox == p.ax_hi ? 1.f : 0.f;
            mylo[q] = oy == p.ay_lo ? 1.f : 0.f;
            myhi[q] = oy == p.ay_hi ? 1.f : 0.f;
        }
    }

    f32x16 acc[MF][2];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    // Global -> register prefetch through buffer loads: out-of-range offsets (padding, channels beyond Cin, idle
    // lanes) return 0 in hardware, so no select follows the load and the wait can sink to the LDS commit.
    // A chunk never straddles the two concatenated sources (host checks C0 % CK == 0 when C1 > 0).
    const float* s0n = p.src0 + (size_t)n * p.C0 * HWin;
    const float* s1n = p.src1 != nullptr ? p.src1 + (size_t)n * p.C1 * HWin : nullptr;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)s0n, 0, p.C0 * HWin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(s1n != nullptr ? s1n : s0n), 0,
                                                                         p.C1 * HWin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, NT * Cin * p.CoutP * 4, 0x00020000);

    float xr[MAXE];
    f32x4 wr[WPT];
    auto prefetch = [&](int cb) {
        const bool first = cb < p.C0;
        const int chan0 = (first ? cb : cb - p.C0) * HWin * 4;
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
            const int e = tid + i * 256;
            const int o4 = e % WV, tc = e / WV;
            const int c = tc % CK, t = tc / CK;
            const bool ok = e < NWV && cb + c < Cin;
            const int off = ok ? (((t * Cin + cb + c) * p.CoutP + co0 + o4 * 4) * 4) : -1;
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
            if (e < NWV) *reinterpret_cast<f32x4*>(Wd + (size_t)e * 4) = wr[i];
        }
    };

    // operands of k-step s = (tap t, channel pair cp)
    auto load_ops = [&](const float* Xl, const float* Wl, int s_, float (&a)[MF], float (&b)[2]) {
        const int t = s_ / (CK / 2), cp = s_ % (CK / 2);
        const int ky = t / K, kx = t % K;
#pragma unroll
        for (int m = 0; m < MF; ++m) a[m] = Wl[(t * CK + 2 * cp) * COT + aoff + m * 32];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ad = boff[q] + 2 * cp * plane + ky * cols + kx;
            float v = Xl[ad];
            if constexpr (ADJ) {
                // fold of the reflected halo: fixed offsets +-(K-1), always inside the staged tile
                constexpr int D = K - 1;
                const bool xl = kx == K - 1, xh = kx == 0, yl = ky == K - 1, yh = ky == 0;
                if (xl || xh) {                      // branch-free (0 / 1 lane masks): see conv_xpair.hip
                    const float mx = xl ? mxlo[q] : mxhi[q];
                    v = fmaf(mx, Xl[ad + (xl ? -D : D)], v);
                }
                if (yl || yh) {
                    const float my = yl ? mylo[q] : myhi[q];
                    const int dy = (yl ? -D : D) * cols;
                    v = fmaf(my, Xl[ad + dy], v);
                    if (xl || xh) {
                        const float mx = xl ? mxlo[q] : mxhi[q];
                        v = fmaf(my * mx, Xl[ad + dy + (xl ? -D : D)], v);
                    }
                }
            }
            b[q] = v;
        }
    };

    // K loop, LDS double-buffered: one barrier per chunk.  While chunk k is multiplied out of buffer k&1, the
    // registers holding chunk k+1 (requested one chunk earlier) are committed to the other buffer and the request
    // for chunk k+2 is issued, both in the middle of the MFMA sequence so that neither sits next to the barrier.
    constexpr int COMMIT_AT = NS / 2;
    prefetch(0);
    commit(0);
    if (CK < Cin) prefetch(CK);
    __syncthreads();
    int cur = 0;
    for (int cb = 0; cb < Cin; cb += CK, cur ^= 1) {
        const float* Xc = Xl + cur * BUF;
        const float* Wc = Wl + cur * BUF;
        float a[2][MF], b[2][2];
        load_ops(Xc, Wc, 0, a[0], b[0]);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            // operands one k-step ahead; the scheduling barriers keep hipcc from sinking the ds_reads below the
            // MFMAs that hide them (it would then wait on them right away) or hoisting a whole chunk's reads
            if (s_ + 1 < NS) load_ops(Xc, Wc, s_ + 1, a[(s_ + 1) & 1], b[(s_ + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][m], b[s_ & 1][q], acc[m][q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s_ == COMMIT_AT && cb + CK < Cin) {
                commit(cur ^ 1);
                if (cb + 2 * CK < Cin) prefetch(cb + 2 * CK);
                __builtin_amdgcn_sched_barrier(0);
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
        // (accumulation: eight reads in flight, then their adds and stores -- read-add-store per element serialised 32 memory
        // round trips, the compiler cannot move a read above the store before it)
#pragma unroll
        for (int m = 0; m < MF; ++m) {
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float old[8];
                if (p.accumulate) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = r0 + u;
                        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        old[u] = co < p.Cout ? on[(size_t)co * outHW + sp] : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + u;
                    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (co < p.Cout) {
                        float v = acc[m][q][r];
                        if (p.bias != nullptr) v += p.bias[co];
                        if (p.accumulate) v += old[u];
                        on[(size_t)co * outHW + sp] = v;
                    }
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
    const size_t lds = 2 * ((((size_t)CK * rows * cols + 3) & ~(size_t)3) + (size_t)NT * CK * 32 * MF) * sizeof(float);
    dim3 grid(tiles, p.CoutP / (32 * MF), N);
    hipLaunchKernelGGL((conv_igemm_kernel<K, S, MF, ADJ>), grid, dim3(256), lds, st, p);
    C2S_CHECK_LAUNCH("conv_igemm");
    return C2S_OK;
}

void init_hook() {
#define C2S_RAISE4(K_, S_)                               \
    C2S_RAISE_LDS((conv_igemm_kernel<K_, S_, 1, false>)); \
    C2S_RAISE_LDS((conv_igemm_kernel<K_, S_, 2, false>));
    C2S_RAISE4(3, 1) C2S_RAISE4(1, 1) C2S_RAISE4(2, 1) C2S_RAISE4(4, 2)
#undef C2S_RAISE4
    C2S_RAISE_LDS((conv_igemm_kernel<3, 1, 1, true>));
    C2S_RAISE_LDS((conv_igemm_kernel<3, 1, 2, true>));
    C2S_RAISE_LDS((conv_igemm_kernel<2, 1, 1, true>));
    C2S_RAISE_LDS((conv_igemm_kernel<2, 1, 2, true>));
}
C2sInitRegistrar registrar(init_hook);

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
    C2S_REQUIRE(d->C1 == 0 || d->C0 % 16 == 0, "conv_igemm: with two sources C0 must be a multiple of 16");
    C2S_REQUIRE((long)(d->C0 > d->C1 ? d->C0 : d->C1) * d->Hin * d->Win * 4 < (1L << 31), "conv_igemm: frame too large");
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
    p.ay_lo = p.ay_hi = p.ax_lo = p.ax_hi = -1;
    if (d->reflect_adjoint) {
        C2S_REQUIRE(d->pad_mode == C2S_PAD_ZEROS && d->S == 1 && (d->KH == 3 || d->KH == 2),
                    "conv_igemm: reflect_adjoint applies to the zero-padded 3x3 / 2x2-parity data-gradient launches");
        C2S_REQUIRE(d->Hin >= 2 && d->Win >= 2 && d->Hin != 3 && d->Win != 3, "conv_igemm: reflect_adjoint needs planes of 2 or >= 4");
        if (d->KH == 3) {
            // dgrad of conv3x3(reflect pad 1): g_x[1] += w[0] g_y[0] ; g_x[H-2] += w[2] g_y[H-1]
            // (flipped taps: position 1 / tap 2 reads input 0 = tap position - 2; position H-2 / tap 0 reads H-1 = +2)
            C2S_REQUIRE(d->pad_y == 1 && d->pad_x == 1 && d->Hin == d->Hout && d->Win == d->Wout, "conv_igemm: bad 3x3 adjoint geometry");
            p.ay_lo = 1; p.ay_hi = d->Hout - 2;
            p.ax_lo = 1; p.ax_hi = d->Wout - 2;
        } else {
            // parity sub-kernel of the dgrad of conv4x4s2(reflect pad 1); parity = 1 - pad:
            // parity 1 (pad 0): position 0 / tap 1 additionally reads input 0; parity 0 (pad 1): position Ho-1 / tap 0 reads Ho-1
            if (d->pad_y == 0) p.ay_lo = 0; else p.ay_hi = d->Hout - 1;
            if (d->pad_x == 0) p.ax_lo = 0; else p.ax_hi = d->Wout - 1;
        }
    }
    int l2 = 5;
    while (l2 > 2 && (1 << l2) > d->Wout) --l2;
    p.log2fc = l2;
    const int FC = 1 << l2, FR = 32 >> l2;
    p.tiles_x = cdiv(d->Wout, FC);
    const int tiles = p.tiles_x * cdiv(d->Hout, 8 * FR);
    hipStream_t st = (hipStream_t)stream;
    // 64-channel tiles halve the LDS reads per MFMA, but on small planes (16x16 and below) they leave the grid short of
    // two workgroups per CU: fall back to 32-channel tiles there
    const int cus = c2s_cus();
    const bool wide = d->CoutP % 64 == 0 && (long)tiles * d->N * (d->CoutP / 64) >= 2L * cus;
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
