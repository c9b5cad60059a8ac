// 3x3 stride-1 implicit-GEMM convolution on the bf16 MFMA with split-precision ("bf16x3") operands.
//
// Every fp32 operand v is split into hi = bf16(v) and lo = bf16(v - hi) (16 significant bits together); a product
// a*b is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with three v_mfma_f32_32x32x16_bf16 and fp32 accumulation.
// The dropped a_lo*b_lo term and the split residual are <= 2^-16 relative per product (measured end to end:
// ~1e-5 relative on a 64-channel 3x3 layer, i.e. ~10x the rounding noise of the exact-fp32 MFMA path and 100x
// below the 1e-3 parity bar).  The bf16 MFMA runs at 16x the rate of v_mfma_f32_32x32x2_f32, so three of them are
// still 5.3x faster than the exact path.  This is an opt-in mode (engine.CONV_MODE): the exact-fp32 kernels in
// conv_igemm.hip remain the default.
//
// GEMM view as in conv_igemm.hip (A = weights, rows = output channels; B = gathered input, columns = positions).
// K is consumed 16 at a time: 8 input channels x 2 taps (lane half h = lane >> 5 selects the tap of a pair), so an
// 8-channel chunk of a 3x3 kernel is 5 MFMA k-steps (the 10th tap is zero weights).  LDS holds the input tile as
// [position][8 channels] bf16 (one ds_read_b128 per operand, consecutive lanes = consecutive 16-byte slots:
// conflict-free) and the weight slab as [tap][cout][8 channels], both in a hi and a lo copy.
//
// Reflect adjoint (ADJ): the fold of the halo gradient is realised as extra MFMA k-steps whose B operand is the
// shifted input for the border lanes and a zero slot for all others (no bf16 arithmetic needed); only waves that
// own border positions execute them.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct BxParams {
    const float* src0;
    const float* src1;
    const __bf16* whi;
    const __bf16* wlo;
    const float* bias;
    float* out;
    const int* valid;
    int C0, C1, Hin, Win, Cout, CoutP, Hout, Wout;
    int pad_mode, accumulate;
    int log2fc, tiles_x;
    int adj, ay_lo, ay_hi, ax_lo, ax_hi;
    int single;            // measurement mode: activations rounded to bf16 (no lo part) -- with zeroed wlo a plain bf16 x bf16 product
};

constexpr int BX_CK = 8, BX_NTP = 10, BX_NSTEP = 5;

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo, bool single) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = single ? (__bf16)0.f : (__bf16)(x[j] - (float)h);
    }
}

template <int MF, bool ADJ>
__global__ __launch_bounds__(256) void conv3x3_bf16x3_kernel(BxParams p) {
    constexpr int COT = 32 * MF;
    constexpr int NW = BX_NTP * COT;                 // 16-byte weight rows of a chunk (per hi / lo copy)
    constexpr int WPT = (NW + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int n = blockIdx.z;
    if (p.valid != nullptr && p.valid[n] == 0) return;
    const int FC = 1 << p.log2fc, FR = 32 >> p.log2fc;
    const int tile_h = 8 * FR;
    const int rows = tile_h + 2, cols = FC + 2;
    const int plane = rows * cols;
    __bf16* Xh = reinterpret_cast<__bf16*>(lds_raw);      // [plane][8]
    __bf16* Xl = Xh + plane * 8;
    __bf16* Wh = Xl + plane * 8;                          // [NTP][COT][8]
    __bf16* Wl = Wh + NW * 8;
    __bf16* Zs = Wl + NW * 8;                             // 8 zeros (masked operand of the adjoint steps)

    const int tyi = blockIdx.x / p.tiles_x, txi = blockIdx.x % p.tiles_x;
    const int oy0 = tyi * tile_h, ox0 = txi * FC;
    const int co0 = blockIdx.y * COT;
    const int tid = threadIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HWin = p.Hin * p.Win;
    if (tid < 4) reinterpret_cast<float*>(Zs)[tid] = 0.f;

    // ---- spatial byte offsets of the (up to 2) tile positions this thread stages; negative = zero padding
    int poff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * 256;
        int off = -1;
        if (e < plane) {
            const int r = e / cols, cc = e - r * cols;
            int gy = oy0 - 1 + r, gx = ox0 - 1 + cc;
            bool ok;
            if (p.pad_mode == C2S_PAD_REFLECT) {
                ok = gy >= -1 && gy <= p.Hin && gx >= -1 && gx <= p.Win;
                gy = reflect_idx(gy, p.Hin);
                gx = reflect_idx(gx, p.Win);
            } else {
                ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
            }
            if (ok) off = (gy * p.Win + gx) * 4;
        }
        poff[i] = off;
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int fy = li >> p.log2fc, fx = li & (FC - 1);
    int pbase[2];                     // LDS position index (tap 0,0) of the two position fragments
    int oyq[2], oxq;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = 2 * wave + q;
        pbase[q] = (f * FR + fy) * cols + fx;
        oyq[q] = oy0 + f * FR + fy;
    }
    oxq = ox0 + fx;
    // per k-step tap of this lane half
    int tapoff[BX_NSTEP], tapw[BX_NSTEP];
#pragma unroll
    for (int s = 0; s < BX_NSTEP; ++s) {
        const int t = 2 * s + lh;
        const int ta = t < 9 ? t : 8;                    // the 10th tap has zero weights: read any valid position
        tapoff[s] = (ta / 3) * cols + (ta % 3);
        tapw[s] = t * COT;
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
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)s0n, 0, p.C0 * HWin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(s1n != nullptr ? s1n : s0n), 0,
                                                                         p.C1 * HWin * 4, 0x00020000);
    const int nchunks = (Cin + BX_CK - 1) / BX_CK;
    const int wbytes = nchunks * BX_NTP * p.CoutP * 16;
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)p.whi, 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)p.wlo, 0, wbytes, 0x00020000);

    float xr[2][8];
    f32x4 wrh[WPT], wrl[WPT];
    auto prefetch = [&](int cb) {
        const bool first = cb < p.C0;
        const int chan0 = (first ? cb : cb - p.C0) * HWin * 4;
        // channels past the end of the source fall outside the descriptor and read as 0
        if (first) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    xr[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        r0, poff[i] >= 0 ? poff[i] + chan0 + k * HWin * 4 : -1, 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    xr[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        r1, poff[i] >= 0 ? poff[i] + chan0 + k * HWin * 4 : -1, 0, 0));
        }
        const int ci = cb / BX_CK;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;
            const int t = e / COT, o = e - t * COT;
            const int off = e < NW ? (((ci * BX_NTP + t) * p.CoutP + co0 + o) * 16) : -1;
            wrh[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rwh, off, 0, 0));
            wrl[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rwl, off, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;
            if (e < plane) {
                bf16x8 hi, lo;
                split8(xr[i], hi, lo, p.single != 0);
                *reinterpret_cast<bf16x8*>(Xh + e * 8) = hi;
                *reinterpret_cast<bf16x8*>(Xl + e * 8) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;
            if (e < NW) {
                *reinterpret_cast<f32x4*>(Wh + e * 8) = wrh[i];
                *reinterpret_cast<f32x4*>(Wl + e * 8) = wrl[i];
            }
        }
    };

    // adjoint: border flags of this lane / wave
    bool bxlo = false, bxhi = false, bylo[2] = {false, false}, byhi[2] = {false, false};
    bool wave_x = false, wave_y = false;
    if constexpr (ADJ) {
        bxlo = oxq == p.ax_lo;
        bxhi = oxq == p.ax_hi;
        bool anyy = false;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bylo[q] = oyq[q] == p.ay_lo;
            byhi[q] = oyq[q] == p.ay_hi;
            anyy = anyy || bylo[q] || byhi[q];
        }
        wave_x = __any(bxlo || bxhi);
        wave_y = __any(anyy);
    }

    auto mma3 = [&](const bf16x8 (&ah)[MF], const bf16x8 (&al)[MF], const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[q], acc[m][q], 0, 0, 0);
                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[q], acc[m][q], 0, 0, 0);
                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[q], acc[m][q], 0, 0, 0);
            }
    };

    prefetch(0);
    for (int cb = 0; cb < Cin; cb += BX_CK) {
        commit();
        __syncthreads();
        if (cb + BX_CK < Cin) prefetch(cb + BX_CK);
#pragma unroll
        for (int s = 0; s < BX_NSTEP; ++s) {
            bf16x8 ah[MF], al[MF], bh[2], bl[2];
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(Wh + (tapw[s] + m * 32 + li) * 8);
                al[m] = *reinterpret_cast<const bf16x8*>(Wl + (tapw[s] + m * 32 + li) * 8);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bh[q] = *reinterpret_cast<const bf16x8*>(Xh + (pbase[q] + tapoff[s]) * 8);
                bl[q] = *reinterpret_cast<const bf16x8*>(Xl + (pbase[q] + tapoff[s]) * 8);
            }
            mma3(ah, al, bh, bl);
        }
        if constexpr (ADJ) {
            // extra k-steps: lane half 0 applies the "low" rule (tap index 2, input shifted by -2), half 1 the "high"
            // rule (tap index 0, shifted by +2); lanes that do not own the border position read the zero slot
            if (wave_x) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int t = ky * 3 + (lh == 0 ? 2 : 0);
                    const int sh = lh == 0 ? -2 : 2;
                    const bool on = lh == 0 ? bxlo : bxhi;
                    bf16x8 ah[MF], al[MF], bh[2], bl[2];
#pragma unroll
                    for (int m = 0; m < MF; ++m) {
                        ah[m] = *reinterpret_cast<const bf16x8*>(Wh + (t * COT + m * 32 + li) * 8);
                        al[m] = *reinterpret_cast<const bf16x8*>(Wl + (t * COT + m * 32 + li) * 8);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int pos = pbase[q] + ky * cols + (lh == 0 ? 2 : 0) + sh;
                        bh[q] = *reinterpret_cast<const bf16x8*>(on ? Xh + pos * 8 : Zs);
                        bl[q] = *reinterpret_cast<const bf16x8*>(on ? Xl + pos * 8 : Zs);
                    }
                    mma3(ah, al, bh, bl);
                }
            }
            if (wave_y) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int t = (lh == 0 ? 2 : 0) * 3 + kx;
                    const int sh = (lh == 0 ? -2 : 2) * cols;
                    bf16x8 ah[MF], al[MF], bh[2], bl[2];
#pragma unroll
                    for (int m = 0; m < MF; ++m) {
                        ah[m] = *reinterpret_cast<const bf16x8*>(Wh + (t * COT + m * 32 + li) * 8);
                        al[m] = *reinterpret_cast<const bf16x8*>(Wl + (t * COT + m * 32 + li) * 8);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const bool on = lh == 0 ? bylo[q] : byhi[q];
                        const int pos = pbase[q] + (lh == 0 ? 2 : 0) * cols + kx + sh;
                        bh[q] = *reinterpret_cast<const bf16x8*>(on ? Xh + pos * 8 : Zs);
                        bl[q] = *reinterpret_cast<const bf16x8*>(on ? Xl + pos * 8 : Zs);
                    }
                    mma3(ah, al, bh, bl);
                }
                if (wave_x) {
                    // corners: step 0 -> (row rule low) x (col low | col high), step 1 -> (row rule high) x (...)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int ky = c == 0 ? 2 : 0, kx = lh == 0 ? 2 : 0;
                        const int t = ky * 3 + kx;
                        const int sh = (c == 0 ? -2 : 2) * cols + (lh == 0 ? -2 : 2);
                        bf16x8 ah[MF], al[MF], bh[2], bl[2];
#pragma unroll
                        for (int m = 0; m < MF; ++m) {
                            ah[m] = *reinterpret_cast<const bf16x8*>(Wh + (t * COT + m * 32 + li) * 8);
                            al[m] = *reinterpret_cast<const bf16x8*>(Wl + (t * COT + m * 32 + li) * 8);
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const bool on = (c == 0 ? bylo[q] : byhi[q]) && (lh == 0 ? bxlo : bxhi);
                            const int pos = pbase[q] + ky * cols + kx + sh;
                            bh[q] = *reinterpret_cast<const bf16x8*>(on ? Xh + pos * 8 : Zs);
                            bl[q] = *reinterpret_cast<const bf16x8*>(on ? Xl + pos * 8 : Zs);
                        }
                        mma3(ah, al, bh, bl);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue (identical to the exact-fp32 kernel): row = channel, column = position
    const size_t outHW = (size_t)p.Hout * p.Wout;
    float* on = p.out + (size_t)n * p.Cout * outHW;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int oy = oyq[q], ox = oxq;
        if (oy >= p.Hout || ox >= p.Wout) continue;
        const size_t sp = (size_t)oy * p.Wout + ox;
#pragma unroll
        for (int m = 0; m < MF; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
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

// whi/wlo[chunk][tap (NTP)][coutP][8 cin] <- split of src[o*so + c*sc + tap_off[t]]
__global__ void pack_bf16x3_kernel(const float* __restrict__ src, __bf16* __restrict__ whi, __bf16* __restrict__ wlo, int cin,
                                   int cout, int coutP, int ntaps, long so, long sc, TapTable tt, int single) {
    const int nchunks = (cin + BX_CK - 1) / BX_CK;
    const long total = (long)nchunks * BX_NTP * coutP * 8;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e & 7);
        long r = e >> 3;
        const int o = (int)(r % coutP); r /= coutP;
        const int t = (int)(r % BX_NTP);
        const int ci = (int)(r / BX_NTP);
        const int c = ci * BX_CK + k;
        float v = 0.f;
        if (o < cout && t < ntaps && c < cin) v = src[o * so + c * sc + tt.off[t]];
        const __bf16 h = (__bf16)v;
        whi[e] = h;
        wlo[e] = single ? (__bf16)0.f : (__bf16)(v - (float)h);
    }
}

template <int MF, bool ADJ>
int launch_bx(const BxParams& p, int N, int tiles, hipStream_t st) {
    const int FC = 1 << p.log2fc, FR = 32 >> p.log2fc;
    const int plane = (8 * FR + 2) * (FC + 2);
    const size_t lds = ((size_t)plane * 8 * 2 + (size_t)BX_NTP * 32 * MF * 8 * 2 + 8) * sizeof(__bf16);
    dim3 grid(tiles, p.CoutP / (32 * MF), N);
    hipLaunchKernelGGL((conv3x3_bf16x3_kernel<MF, ADJ>), grid, dim3(256), lds, st, p);
    C2S_CHECK_LAUNCH("conv3x3_bf16x3");
    return C2S_OK;
}

}  // namespace

extern "C" size_t c2s_bf16x3_packed_elems(int cin, int coutP) {
    return (size_t)((cin + BX_CK - 1) / BX_CK) * BX_NTP * coutP * 8;
}

// Measurement switch ("report the bf16 deviation from the fp32 oracle"): != 0 drops the lo parts of weights and
// activations, i.e. the 3x3 stride-1 convolutions run as plain bf16 x bf16 products with fp32 accumulation.  Process-wide,
// read by the pack and convolution launchers; never set on the product path.
static int g_single_product = 0;
extern "C" void c2s_bf16x3_set_single_product(int on) { g_single_product = on != 0; }

extern "C" int c2s_pack_weights_bf16x3(const float* src, void* whi, void* wlo, int cin, int cout, int coutP, int ntaps,
                                       long stride_o, long stride_c, const int* host_tap_off, void* stream) {
    C2S_REQUIRE(src && whi && wlo && host_tap_off, "pack_bf16x3: null pointer");
    C2S_REQUIRE(ntaps >= 1 && ntaps <= 9 && coutP % 32 == 0 && coutP >= cout, "pack_bf16x3: bad sizes");
    TapTable tt;
    for (int i = 0; i < 16; ++i) tt.off[i] = i < ntaps ? host_tap_off[i] : 0;
    const long total = (long)c2s_bf16x3_packed_elems(cin, coutP);
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)whi, (__bf16*)wlo,
                       cin, cout, coutP, ntaps, stride_o, stride_c, tt, g_single_product);
    C2S_CHECK_LAUNCH("pack_bf16x3");
    return C2S_OK;
}

extern "C" int c2s_conv3x3_bf16x3(const c2s_conv_desc* d, const float* src0, const float* src1, const void* whi,
                                  const void* wlo, const float* bias, float* out, const int* valid, void* stream) {
    C2S_REQUIRE(d && src0 && whi && wlo && out, "conv3x3_bf16x3: null pointer");
    C2S_REQUIRE(d->KH == 3 && d->KW == 3 && d->S == 1 && d->pad_y == 1 && d->pad_x == 1, "conv3x3_bf16x3: 3x3 stride 1 pad 1 only");
    C2S_REQUIRE(d->Hin == d->Hout && d->Win == d->Wout && d->OutH == d->Hout && d->OutW == d->Wout && d->osy == 1 &&
                    d->osx == 1 && d->ooy == 0 && d->oox == 0, "conv3x3_bf16x3: dense same-size output only");
    C2S_REQUIRE(d->C0 % 8 == 0 && d->C1 % 8 == 0 && (d->C1 == 0 || src1), "conv3x3_bf16x3: channel counts must be multiples of 8");
    C2S_REQUIRE(d->CoutP % 32 == 0 && d->CoutP >= d->Cout, "conv3x3_bf16x3: CoutP must be a multiple of 32");
    C2S_REQUIRE((long)(d->C0 > d->C1 ? d->C0 : d->C1) * d->Hin * d->Win * 4 < (1L << 31), "conv3x3_bf16x3: frame too large");
    if (d->pad_mode == C2S_PAD_REFLECT) C2S_REQUIRE(d->Hin >= 2 && d->Win >= 2, "conv3x3_bf16x3: reflect needs planes >= 2");
    BxParams p;
    p.src0 = src0; p.src1 = src1; p.whi = (const __bf16*)whi; p.wlo = (const __bf16*)wlo; p.bias = bias; p.out = out;
    p.valid = valid; p.C0 = d->C0; p.C1 = d->C1; p.Hin = d->Hin; p.Win = d->Win; p.Cout = d->Cout; p.CoutP = d->CoutP;
    p.Hout = d->Hout; p.Wout = d->Wout; p.pad_mode = d->pad_mode; p.accumulate = d->accumulate;
    p.adj = d->reflect_adjoint;
    p.single = g_single_product;
    p.ay_lo = p.ay_hi = p.ax_lo = p.ax_hi = -1;
    if (d->reflect_adjoint) {
        C2S_REQUIRE(d->pad_mode == C2S_PAD_ZEROS && d->Hin != 3 && d->Win != 3, "conv3x3_bf16x3: bad adjoint geometry");
        p.ay_lo = 1; p.ay_hi = d->Hout - 2; p.ax_lo = 1; p.ax_hi = d->Wout - 2;
    }
    int l2 = 5;
    while (l2 > 2 && (1 << l2) > d->Wout) --l2;
    p.log2fc = l2;
    p.tiles_x = cdiv(d->Wout, 1 << l2);
    const int tiles = p.tiles_x * cdiv(d->Hout, 8 * (32 >> l2));
    hipStream_t st = (hipStream_t)stream;
    const bool wide = d->CoutP % 64 == 0;
    if (p.adj) return wide ? launch_bx<2, true>(p, d->N, tiles, st) : launch_bx<1, true>(p, d->N, tiles, st);
    return wide ? launch_bx<2, false>(p, d->N, tiles, st) : launch_bx<1, false>(p, d->N, tiles, st);
}
