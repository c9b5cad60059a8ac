// Transposed 4x4 stride-2 pad-1 convolution rows on the exact-f32 MFMA: both x-parities of one output row
// parity in a single launch.
//
// out[n, o, 2i+py, 2j+px] = sum_{c,ty,tx} in[n, c, i-(1-py)+ty, j-(1-px)+tx] * wpk[px][ty*2+tx][c][o]
//
// One launch covers one y-parity py and BOTH x-parities: a lane owns sub-grid position (i, j) and accumulates the
// two outputs x = 2j and x = 2j+1, which it stores as one float2 -> 32 lanes write 256 contiguous bytes of an
// output row.  (Four single-parity launches of the generic kernel each wrote every other float of the row:
// half-used 32-byte sectors and 4 passes over the output; this kernel is 2 launches with full-line stores.)
// GEMM mapping as in conv_igemm.hip (A = weights, rows = output channel; B = gathered input, columns =
// position); workgroup = 4 waves, wave w owns position fragment w (32 sub-grid positions) x 32*MF channels x 2
// parities = 2*MF accumulators; K loop over input channels in LDS-double-buffered chunks of 4 (21 KB of LDS per workgroup: more resident workgroups hide more than the extra barriers cost; 8 and 2 were both slower).
//
// Reference call sites replaced: nn.ConvTranspose2d(k=4,s=2,p=1) forward (src/backbones/conv.py:384-390) and the
// convolution_backward-input of the 4x4 stride-2 down convolutions (conv.py:263-271), including the adjoint of
// their reflection padding (conv.py:72-79).
#include "common.h"

namespace {

struct XpParams {
    const float* src;
    const float* wpk;      // [px 2][tap 4][Cin][CoutP]
    const float* bias;
    float* out;
    const int* valid;
    int Cin, Hin, Win, Cout, CoutP, Hout, Wout, OutH, OutW;
    int pad_y, ooy, accumulate;
    int log2fc, tiles_x;
    int ay_lo, ay_hi;      // reflect adjoint rows of this y-parity (-1 = rule absent); the x rules are fixed:
                           // px=0: position Wout-1 / tap tx=0 also reads column j ; px=1: position 0 / tap tx=1 too
};

constexpr int XP_CK = 4;
constexpr int xp_plane(int l2) { return (4 * (32 >> l2) + 1) * ((1 << l2) + 2); }
constexpr int xp_cmax(int a, int b) { return a > b ? a : b; }
constexpr int XP_MAXPLANE = xp_cmax(xp_cmax(xp_plane(2), xp_plane(3)), xp_cmax(xp_plane(4), xp_plane(5)));
constexpr int XP_MAXE = (XP_CK * XP_MAXPLANE + 255) / 256;

template <int MF, bool ADJ>
__global__ __launch_bounds__(256) void conv_xpair_kernel(XpParams p) {
    constexpr int CK = XP_CK, MAXE = XP_MAXE;
    constexpr int COT = 32 * MF;
    constexpr int WV = COT / 4;
    constexpr int NWV = 2 * 4 * CK * WV;         // float4 items of the weight slab [px][tap][CK][COT]
    constexpr int WPT = (NWV + 255) / 256;
    constexpr int NS = 4 * (CK / 2);             // k-steps per chunk: (tap, channel pair)
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
    const int tile_h = 4 * FR;
    const int rows = tile_h + 1, cols = FC + 2;
    const int plane = rows * cols;
    const int xsz = (CK * plane + 3) & ~3;
    float* Xl = lds;                 // [2][ [CK][plane] | [2][4][CK][COT] ]
    float* Wl = lds + xsz;
    const int BUF = xsz + 2 * 4 * CK * COT;

    const int tyi = bxi / p.tiles_x, txi = bxi % p.tiles_x;
    const int oy0 = tyi * tile_h, ox0 = txi * FC;
    const int co0 = blockIdx.y * COT;
    const int tid = threadIdx.x;
    const int HWin = p.Hin * p.Win;

    // byte offsets of this thread's staging elements within a chunk (zero padding -> -1 -> the buffer load returns 0)
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
            const int gy = oy0 - p.pad_y + r, gx = ox0 - 1 + cc;
            if (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) off = (c * HWin + gy * p.Win + gx) * 4;
        }
        goff[i] = off;
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int fy = li >> p.log2fc, fx = li & (FC - 1);
    const int oy = oy0 + wave * FR + fy, ox = ox0 + fx;
    const int boff = lk * plane + (wave * FR + fy) * cols + fx;
    const int aoff = lk * COT + li;

    float mylo = 0.f, myhi = 0.f, mx[2] = {0.f, 0.f};
    if constexpr (ADJ) {
        mylo = oy == p.ay_lo ? 1.f : 0.f;
        myhi = oy == p.ay_hi ? 1.f : 0.f;
        mx[0] = ox == p.Wout - 1 ? 1.f : 0.f;
        mx[1] = ox == 0 ? 1.f : 0.f;
    }

    f32x16 acc[MF][2];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    const float* sn = p.src + (size_t)n * p.Cin * HWin;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)sn, 0, p.Cin * HWin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 8 * p.Cin * p.CoutP * 4, 0x00020000);

    float xr[MAXE];
    f32x4 wr[WPT];
    auto prefetch = [&](int cb) {
        const int chan0 = cb * HWin * 4;
#pragma unroll
        for (int i = 0; i < MAXE; ++i)
            xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, goff[i] >= 0 ? goff[i] + chan0 : -1, 0, 0));
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;
            const int o4 = e % WV, tc = e / WV;
            const int c = tc % CK, pt = tc / CK;           // pt = px*4 + tap
            const bool ok = e < NWV && cb + c < p.Cin;
            const int off = ok ? (((pt * p.Cin + cb + c) * p.CoutP + co0 + o4 * 4) * 4) : -1;
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

    // operands of k-step s = (tap t = ty*2+tx, channel pair cp): a[px][m], b[px]
    auto load_ops = [&](const float* Xc, const float* Wc, int s_, float (&a)[2][MF], float (&b)[2]) {
        const int t = s_ / (CK / 2), cp = s_ % (CK / 2);
        const int ty = t >> 1, tx = t & 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int m = 0; m < MF; ++m) a[q][m] = Wc[((q * 4 + t) * CK + 2 * cp) * COT + aoff + m * 32];
            const int ad = boff + 2 * cp * plane + ty * cols + tx + q;       // tile column 0 = sub-grid column j-1
            float v = Xc[ad];
            if constexpr (ADJ) {
                // adjoint of the reflection: the halo tap folds back onto the neighbouring input (offset +-1).  Branch-free:
                // the 0 / 1 lane masks select, every extra element lies inside the staged tile (finite data).  With
                // wave-uniform branches here (round 1-2: `if (wave_x)`, `if (wave_y)`) the basic-block boundaries kept the
                // operand reads of the next k-step from being scheduled under the MFMAs: +110 us on the 64 -> 64 @128x128
                // data gradient (tools/xpair_bench.py).
                const bool xr_ = (q == 0 && tx == 0) || (q == 1 && tx == 1);
                const int dx = q == 0 ? 1 : -1;
                const float my = ty == 1 ? mylo : myhi;
                const int dy = (ty == 1 ? -1 : 1) * cols;
                if (xr_) v = fmaf(mx[q], Xc[ad + dx], v);
                v = fmaf(my, Xc[ad + dy], v);
                if (xr_) v = fmaf(my * mx[q], Xc[ad + dy + dx], v);
            }
            b[q] = v;
        }
    };

    constexpr int COMMIT_AT = NS / 2;
    prefetch(0);
    commit(0);
    if (CK < p.Cin) prefetch(CK);
    __syncthreads();
    int cur = 0;
    for (int cb = 0; cb < p.Cin; cb += CK, cur ^= 1) {
        const float* Xc = Xl + cur * BUF;
        const float* Wc = Wl + cur * BUF;
        float a[2][2][MF], b[2][2];
        load_ops(Xc, Wc, 0, a[0], b[0]);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            if (s_ + 1 < NS) load_ops(Xc, Wc, s_ + 1, a[(s_ + 1) & 1], b[(s_ + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][q][m], b[s_ & 1][q], acc[m][q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s_ == COMMIT_AT && cb + CK < p.Cin) {
                commit(cur ^ 1);
                if (cb + 2 * CK < p.Cin) prefetch(cb + 2 * CK);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: row (channel) = (r&3) + 8*(r>>2) + 4*lk ; one float2 (x = 2*ox, 2*ox+1) per channel
    if (oy >= p.Hout || ox >= p.Wout) return;
    const size_t outHW = (size_t)p.OutH * p.OutW;
    float* on = p.out + (size_t)n * p.Cout * outHW + (size_t)(oy * 2 + p.ooy) * p.OutW + 2 * ox;
    // accumulation into an existing gradient: the reads of EIGHT elements first, then their adds and stores (read-add-store per
    // element made every read wait for the store before it -- the compiler cannot reorder them: 32 memory round trips in a
    // row; eight at a time keeps the kernel at three workgroups per CU)
#pragma unroll
    for (int m = 0; m < MF; ++m) {
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
            float2 old[8];
            if (p.accumulate) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + u;
                    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    old[u] = co < p.Cout ? *reinterpret_cast<const float2*>(on + (size_t)co * outHW) : make_float2(0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u;
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < p.Cout) {
                    float2 v = make_float2(acc[m][0][r], acc[m][1][r]);
                    if (p.bias != nullptr) { v.x += p.bias[co]; v.y += p.bias[co]; }
                    if (p.accumulate) { v.x += old[u].x; v.y += old[u].y; }
                    *reinterpret_cast<float2*>(on + (size_t)co * outHW) = v;
                }
            }
        }
    }
}

template <int MF, bool ADJ>
int launch_xpair(const XpParams& p, int N, int tiles, hipStream_t st) {
    const int FC = 1 << p.log2fc, FR = 32 >> p.log2fc;
    const int plane = (4 * FR + 1) * (FC + 2);
    const size_t lds = 2 * ((((size_t)XP_CK * plane + 3) & ~(size_t)3) + (size_t)2 * 4 * XP_CK * 32 * MF) * sizeof(float);
    dim3 grid(tiles, p.CoutP / (32 * MF), N);
    hipLaunchKernelGGL((conv_xpair_kernel<MF, ADJ>), grid, dim3(256), lds, st, p);
    C2S_CHECK_LAUNCH("conv_xpair");
    return C2S_OK;
}

}  // namespace

extern "C" int c2s_conv_xpair(const c2s_conv_desc* d, const float* src, const float* wpk, const float* bias, float* out,
                              const int* valid, void* stream) {
    C2S_REQUIRE(d && src && wpk && out, "conv_xpair: null pointer");
    C2S_REQUIRE(d->N > 0 && d->C0 > 0 && d->C1 == 0 && d->Cout > 0, "conv_xpair: bad channels (single source only)");
    C2S_REQUIRE(d->CoutP % 32 == 0 && d->CoutP >= d->Cout, "conv_xpair: CoutP must be a multiple of 32");
    C2S_REQUIRE(d->KH == 2 && d->KW == 2 && d->S == 1 && d->pad_mode == C2S_PAD_ZEROS, "conv_xpair: 2x2 zero-padded sub-kernels only");
    C2S_REQUIRE(d->pad_y == 0 || d->pad_y == 1, "conv_xpair: pad_y = 1 - py");
    C2S_REQUIRE(d->osy == 2 && d->osx == 2 && d->ooy == 1 - d->pad_y, "conv_xpair: output rows 2*i + py, columns 2*j + {0,1}");
    C2S_REQUIRE(d->Hout == d->Hin && d->Wout == d->Win && d->OutH == 2 * d->Hout && d->OutW == 2 * d->Wout,
                "conv_xpair: output plane must be twice the input plane");
    C2S_REQUIRE((long)d->C0 * d->Hin * d->Win * 4 < (1L << 31), "conv_xpair: frame too large");
    if (d->reflect_adjoint) C2S_REQUIRE(d->Hin >= 2 && d->Win >= 2, "conv_xpair: reflect_adjoint needs planes >= 2");
    XpParams p;
    p.src = src; p.wpk = wpk; p.bias = bias; p.out = out; p.valid = valid;
    p.Cin = d->C0; p.Hin = d->Hin; p.Win = d->Win; p.Cout = d->Cout; p.CoutP = d->CoutP;
    p.Hout = d->Hout; p.Wout = d->Wout; p.OutH = d->OutH; p.OutW = d->OutW;
    p.pad_y = d->pad_y; p.ooy = d->ooy; p.accumulate = d->accumulate;
    p.ay_lo = p.ay_hi = -1;
    if (d->reflect_adjoint) {
        // parity 1 (pad 0): position 0 / tap 1 additionally reads input 0; parity 0 (pad 1): position H-1 / tap 0 reads H-1
        if (d->pad_y == 0) p.ay_lo = 0; else p.ay_hi = d->Hout - 1;
    }
    int l2 = 5;
    while (l2 > 2 && (1 << l2) > d->Wout) --l2;
    p.log2fc = l2;
    const int FC = 1 << l2, FR = 32 >> l2;
    p.tiles_x = cdiv(d->Wout, FC);
    const int tiles = p.tiles_x * cdiv(d->Hout, 4 * FR);
    hipStream_t st = (hipStream_t)stream;
    const int cus = c2s_cus();
    const bool wide = d->CoutP % 64 == 0 && (long)tiles * d->N * (d->CoutP / 64) >= 2L * cus;
    if (d->reflect_adjoint)
        return wide ? launch_xpair<2, true>(p, d->N, tiles, st) : launch_xpair<1, true>(p, d->N, tiles, st);
    return wide ? launch_xpair<2, false>(p, d->N, tiles, st) : launch_xpair<1, false>(p, d->N, tiles, st);
}
