// Data gradient of the 4x4 stride-2 pad-1 convolution (= a transposed convolution) as Winograd F(2x2, 2x2) per output parity,
// on the structure of conv_s2wino.hip / conv_winograd16.hip.
//
// With u = 2m + e (e = parity of the input-resolution row u) the gradient is a 2-tap stride-1 filter over gy per parity:
//   gx[2m]   = gy[m-1] w3 + gy[m] w1        (e = 0: window gy[m-1 .. ], taps (w3, w1))
//   gx[2m+1] = gy[m]   w2 + gy[m+1] w0      (e = 1: window gy[m   .. ], taps (w2, w0))
// so a pair of outputs (m, m+1) of one parity needs three gy values and F(2,2) takes 3 multiplies instead of 4 per dimension:
// 9 instead of 16 per (gy channel, parity, 2x2 block of the parity plane).  A workgroup takes ONE row parity ey (grid.y), its
// lanes BOTH column parities ex: the two 3-wide column windows of a block overlap in two columns (one 4-float read), and the
// four outputs (n, ex) of a row are contiguous in gx: one 16-byte store.  The MFMA's output rows are the "virtual channels"
// (ex, cin): a wave owns 32 input channels x 2 column parities x 16 blocks = acc[9][2][2] of 16x16.
//
// Reflect padding of the forward pass: the gradient of the halo rows/columns (u = -1 and u = 2 Ho) is added to u = 1 and
// u = 2 Ho - 2.  The halo term of u = 1 is gy[0] w0 = g1 d0 of the first block of parity 1 -- not expressible by changing the
// data of the standard F(2,2) points, but another 3-multiply algorithm computes both outputs INCLUDING it:
//   first output folds ("top"):   V = (d1, d0 + d1, d2),   y0 = -m0 + m1,  y1 = m0 + m2     (y0 = g0 d0 + g1 (d0 + d1))
//   last output folds ("bottom"): V = (d0, d1 + d2, d1),   y0 =  m0 + m2,  y1 = m1 - m2     (y1 = g0 (d1 + d2) + g1 d2)
// with the same weights U = (g0, g0 + g1, g1).  Rows: per wave (uniform branch); columns: per lane (0/1 masks), only in tiles
// that touch the left / right edge.  No border kernel, no second pass.
//
// Reference call site replaced: convolution_backward-input of nn.Conv2d(4, stride 2, padding 1, reflect) in DownConvBlock
// (src/backbones/conv.py:263-271), for layers with a multiple of 8 (>= 24) output channels on planes at least 32 wide
// (C2S_S2WINO=0 keeps conv_xpair_kernel).
#include "common.h"
#include <stdlib.h>

namespace {

struct S2dParams {
    const float* src;      // gy [N][Kc][Ho][Wo]
    const float* upk;      // [ey 2][cin block][chunk][2 k-steps][4 k][128 = ex * 64 + cin][12]
    float* out;            // gx [N][Cs][2 Ho][2 Wo]
    const int* valid;
    int Kc, Ho, Wo, Cs, CsP;
    int fold, accumulate;
    int N, tiles, tiles_x, nchunks;
};

constexpr int D2_UP = 12;
constexpr int D2_UHALF = 4 * 128 * D2_UP;            // one k-step (4 gy channels x 128 virtual channels): 6,144 floats = 24 KB
constexpr int D2_USLAB = 2 * D2_UHALF;               // a chunk of 8 gy channels: 48 KB
constexpr int D2_URING = 5 * D2_UHALF;               // 120 KB
constexpr int D2_RR = 9, D2_RC = 34;                 // raw gy tile: 9 rows x 34 columns per channel
constexpr int D2_PLANE = D2_RR * D2_RC;              // 306
constexpr int D2_XP = 352;                           // LDS pitch per channel (= 32 mod 64 banks)
constexpr int D2_MAXE = 6;
constexpr int D2_XS = D2_MAXE * 512;                 // 3,072 floats per raw slot
constexpr int D2_XSLOTS = 3;
constexpr int D2_LDS_FLOATS = D2_URING + D2_XSLOTS * D2_XS;      // 159,744 B (+ frame flag bits)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define C2S_AS1 __attribute__((address_space(1)))
#define C2S_AS3 __attribute__((address_space(3)))

__global__ __launch_bounds__(512, 1) void conv_s2dgrad_kernel(S2dParams p) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, kq = lane >> 4;        // block column / k index (B; D: channel quad) ; A: virtual channel / k index
    const int ch = w & 1, brow = w >> 1;            // input-channel half (32) and block row of this wave
    const int ey = blockIdx.y & 1, cblk = blockIdx.y >> 1;
    const int HWo = p.Ho * p.Wo, Hin = 2 * p.Ho, Win = 2 * p.Wo, HWin = Hin * Win;
    const int ntotal = p.N * p.tiles;
    const int K = p.nchunks;

    unsigned* lvalid = reinterpret_cast<unsigned*>(lds + D2_LDS_FLOATS);
    for (int wi = tid; wi < (p.N + 31) / 32; wi += 512) {
        unsigned m = 0;
        for (int b = 0; b < 32; ++b) {
            const int f = wi * 32 + b;
            if (f < p.N && (p.valid == nullptr || p.valid[f] != 0)) m |= 1u << b;
        }
        lvalid[wi] = m;
    }
    __syncthreads();
    auto next_valid = [&](int tt) {
        while (tt < ntotal) {
            const int f = tt / p.tiles;
            if ((lvalid[f >> 5] >> (f & 31)) & 1u) break;
            tt += gridDim.x;
        }
        return tt;
    };
    // tile (ty, tx): parity-plane rows 8 ty .. 8 ty + 7 (4 block rows), columns 32 tx .. 32 tx + 31 (16 block columns)
    auto tile_origin = [&](int tt, int& n, int& m0, int& n0) {
        n = tt / p.tiles;
        const int ti = tt - n * p.tiles;
        const int tyi = ti / p.tiles_x, txi = ti - tyi * p.tiles_x;
        m0 = tyi * 8; n0 = txi * 32;
    };

    // ---- staging side: raw tile element (c, r, q) = gy[8 k + c][m0 - 1 + ey + r][n0 - 1 + q], zero outside the plane
    int goff[D2_MAXE];
    __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, 0, 0x00020000);
    auto begin_staging = [&](int tt) {
        int n, m0, n0;
        tile_origin(tt, n, m0, n0);
#pragma unroll
        for (int i = 0; i < D2_MAXE; ++i) {
            const int e = tid + i * 512;
            const int c = e / D2_XP, rem = e - c * D2_XP;
            const int r = rem / D2_RC, q = rem - r * D2_RC;
            const int gy = m0 - 1 + ey + r, gx = n0 - 1 + q;
            const bool ok = rem < D2_PLANE && c < 8 && gy >= 0 && gy < p.Ho && gx >= 0 && gx < p.Wo;
            goff[i] = ok ? ((c * HWo + gy * p.Wo + gx) * 4) : 0x7FFF0000;
        }
        r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + (size_t)n * p.Kc * HWo), 0, p.Kc * HWo * 4, 0x00020000);
    };
    // one chunk's requests: U (48 KB: six 16-byte pieces per thread, three per half slab), then the raw tile
    auto stage = [&](int k, int h0, int slot) {
        const int h1 = h0 == 4 ? 0 : h0 + 1;
        const C2S_AS1 char* g = (const C2S_AS1 char*)p.upk + ((size_t)(ey * (p.CsP / 64) + cblk) * K + k) * (D2_USLAB * 4);
        float* d0 = lds + h0 * D2_UHALF + w * 256;
        float* d1 = lds + h1 * D2_UHALF + w * 256;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float* dst = (i < 3 ? d0 : d1) + (i % 3) * 2048;
            __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(g + i * 8192 + (unsigned)(tid * 16)), (C2S_AS3 void*)dst, 16, 0, 0);
        }
        const int chan0 = 8 * k * HWo * 4;
        float* Xd = lds + D2_URING + slot * D2_XS + w * 64;
#pragma unroll
        for (int i = 0; i < D2_MAXE; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (C2S_AS3 void*)(Xd + i * 512), 4, goff[i], chan0, 0, 0);
    };
    const int wg0 = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    int tile = next_valid(wg0);
    if (tile >= ntotal) return;
    int stile = tile, sk = 0;                          // next chunk to stage: chunk sk of tile stile (stile >= ntotal: none left)
    begin_staging(stile);
    auto stage_next = [&](int h0, int slot) {          // (uniform across the workgroup)
        if (stile >= ntotal) return;
        stage(sk, h0, slot);
        if (++sk == K) {                               // once per multiplied tile (K >= 3), at its chunk K - 3
            sk = 0;
            stile = next_valid(stile + gridDim.x);
            if (stile < ntotal) begin_staging(stile);
        }
    };
    int u0 = 0, rc = 0;                                // ring positions of the chunk being multiplied
    stage_next(0, 0);
    stage_next(2, 1);

    // LDS offsets of this lane's operands
    const int aoff = (kq * 128 + 32 * ch + t) * D2_UP;                          // in a half slab; + (ex * 64 + mt * 16) * UP
    const int boff = D2_URING + kq * D2_XP + (2 * brow) * D2_RC + 2 * t;        // in a raw slot; + s * 4 * XP + r * RC
    auto load_a = [&](const float* ab, int ex, int mt, float (&a)[12]) {
#pragma unroll
        for (int q4 = 0; q4 < 3; ++q4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ab + (ex * 64 + mt * 16) * D2_UP + 4 * q4);
            a[4 * q4] = v[0]; a[4 * q4 + 1] = v[1]; a[4 * q4 + 2] = v[2]; a[4 * q4 + 3] = v[3];
        }
    };
    int boff0 = boff, boff1 = boff + 4 * D2_XP;
    asm volatile("" : "+v"(boff0), "+v"(boff1));
    auto load_d = [&](const float* bufp, int s, f32x2 (&dl)[3], f32x2 (&dh)[3]) {      // patch rows as (cols 0,1), (cols 2,3)
        const float* bb = bufp + (s ? boff1 : boff0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dl[r] = *reinterpret_cast<const f32x2*>(bb + r * D2_RC);
            dh[r] = *reinterpret_cast<const f32x2*>(bb + r * D2_RC + 2);
        }
    };
    // fold state of the tile being multiplied: rows per wave (rt: 0 standard, 1 first output folds, 2 last output folds),
    // columns per lane (fl: column parity 1 of block column 0; fr: column parity 0 of the last block column)
    int rt = 0;
    float fl = 0.f, fr = 0.f;
    bool colfold = false;
    // row stage of a patch (in place: rows become the three row points), then the column stage of one column parity:
    // V[3 xi + nu]
    auto transform_rows = [&](f32x2 (&dl)[3], f32x2 (&dh)[3]) {
        const f32x2 l0 = dl[0], l1 = dl[1], l2 = dl[2], h0 = dh[0], h1 = dh[1], h2 = dh[2];
        if (rt == 0) {
            dl[0] = l0 - l1; dl[2] = l2 - l1;
            dh[0] = h0 - h1; dh[2] = h2 - h1;
        } else if (rt == 1) {
            dl[0] = l1; dl[1] = l0 + l1;
            dh[0] = h1; dh[1] = h0 + h1;
        } else {
            dl[1] = l1 + l2; dl[2] = l1;
            dh[1] = h1 + h2; dh[2] = h1;
        }
    };
    auto transform_cols = [&](const f32x2 (&tl)[3], const f32x2 (&th)[3], int ex, float (&V)[9]) {
#pragma unroll
        for (int xi = 0; xi < 3; ++xi) {
            const float c0 = tl[xi][0], c1 = tl[xi][1], c2 = th[xi][0], c3 = th[xi][1];
            if (ex == 0) {                       // window (c0, c1, c2): the last output may fold (fr)
                float a0 = c0 - c1, a1 = c1, a2 = c2 - c1;
                if (colfold) { a0 = fmaf(fr, c1, a0); a1 = fmaf(fr, c2, a1); a2 = fmaf(fr, 2.f * c1 - c2, a2); }
                V[3 * xi] = a0; V[3 * xi + 1] = a1; V[3 * xi + 2] = a2;
            } else {                             // window (c1, c2, c3): the first output may fold (fl)
                float b0 = c1 - c2, b1 = c2, b2 = c3 - c2;
                if (colfold) { b0 = fmaf(fl, 2.f * c2 - c1, b0); b1 = fmaf(fl, c1, b1); b2 = fmaf(fl, c2, b2); }
                V[3 * xi] = b0; V[3 * xi + 1] = b1; V[3 * xi + 2] = b2;
            }
        }
    };
    f32x4 acc[9][2][2];
    auto mma = [&](const float (&a)[12], const float (&V)[9], int ex, int mt, bool first) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (first) acc[q][ex][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], V[q], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[q][ex][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], V[q], acc[q][ex][mt], 0, 0, 0);
        }
    };

    float a0[12], a1[12], V[9];
    f32x2 dl[3], dh[3];
    const int cof = cblk * 64 + 32 * ch + 4 * kq;      // this lane's input channels: cof + 16 mt + r (D rows 4 kq + r)
    __syncthreads();                                  // (drains the first requests: vmcnt(0))
    load_a(lds + aoff, 0, 0, a0);
    load_d(lds, 0, dl, dh);
    while (true) {
        int n, m0, n0;
        tile_origin(tile, n, m0, n0);
        {
            const int gbr = (m0 >> 1) + brow, gbc = (n0 >> 1) + t;          // global block row / column
            const int lastr = (p.Ho >> 1) - 1, lastc = (p.Wo >> 1) - 1;
            rt = !p.fold ? 0 : (ey == 1 && gbr == 0 ? 1 : (ey == 0 && gbr == lastr ? 2 : 0));
            fl = (p.fold && gbc == 0) ? 1.f : 0.f;
            fr = (p.fold && gbc == lastc) ? 1.f : 0.f;
            colfold = p.fold && (n0 == 0 || (n0 >> 1) + 16 > lastc);
        }
        // one chunk = two k-steps (4 gy channels each) x (ex, mt) in {0,1}^2: eight half-steps of 9 MFMAs; the A operand of a
        // half-step is read one half-step ahead; the patch of the next k-step is read once the column stage of ex = 1 has
        // consumed the current one (same registers)
        auto kstep = [&](const float* A, const float* Anext, const float* Xnext, int snext, bool first) {
            load_a(A, 0, 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            transform_rows(dl, dh);
            transform_cols(dl, dh, 0, V);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, V, 0, 0, first);
            __builtin_amdgcn_sched_barrier(0);
            load_a(A, 1, 0, a0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, V, 0, 1, first);
            __builtin_amdgcn_sched_barrier(0);
            load_a(A, 1, 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            transform_cols(dl, dh, 1, V);
            __builtin_amdgcn_sched_barrier(0);
            load_d(Xnext, snext, dl, dh);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, V, 1, 0, first);
            __builtin_amdgcn_sched_barrier(0);
            load_a(Anext, 0, 0, a0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, V, 1, 1, first);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto chunk = [&](bool first) {
            const int u1 = u0 == 4 ? 0 : u0 + 1, u2 = u1 == 4 ? 0 : u1 + 1;
            const int rn = rc == 2 ? 0 : rc + 1;
            // the first k-step's tail reads this chunk's second half slab and raw slot: landed a chunk ago
            kstep(lds + u0 * D2_UHALF + aoff, lds + u1 * D2_UHALF + aoff, lds + rc * D2_XS, 1, first);
            // everyone's requests for the NEXT chunk (issued one chunk ago) have landed; everyone has left the previous chunk's
            // raw slot and this chunk's first half slab
            __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stage_next(u0 == 0 ? 4 : u0 - 1, rc == 0 ? 2 : rc - 1);      // chunk after next: half slabs u0 + 4, u0 (mod 5); raw slot rc + 2 (mod 3)
            // (after the last chunk of the last tile the tail reads are stale and unused)
            kstep(lds + u1 * D2_UHALF + aoff, lds + u2 * D2_UHALF + aoff, lds + rn * D2_XS, 0, false);
            u0 = u2;
            rc = rn;
        };
        chunk(true);
        for (int k = 1; k < K; ++k) chunk(false);
        // ---- epilogue: per column parity the output transform on channel pairs (columns with the lane's fold masks, rows with
        // the wave's variant), then one 16-byte store per (channel, output row): (n, ex) = (2C,0) (2C,1) (2C+1,0) (2C+1,1)
        const int mrow = m0 + 2 * brow;                      // parity-plane row of this lane's first output
        const int ncol = n0 + 2 * t;
        const __amdgpu_buffer_rsrc_t ro =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)n * p.Cs * HWin), 0, p.Cs * HWin * 4, 0x00020000);
        const bool in0 = ncol < p.Wo && mrow < p.Ho, in1 = in0 && mrow + 1 < p.Ho;
        const int vo0 = in0 ? (cof * HWin + (2 * mrow + ey) * Win + 2 * ncol) * 4 : 0x7FFF0000;
        const int vo1 = in1 ? vo0 + 2 * Win * 4 : 0x7FFF0000;
        const f32x2 fl2 = {fl, fl}, fr2 = {fr, fr};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 Y[2][2][2];                                         // [ex][output row i][output col j] on the channel pair
#pragma unroll
                for (int ex = 0; ex < 2; ++ex) {
                    f32x2 P0[3], P1[3];
#pragma unroll
                    for (int xi = 0; xi < 3; ++xi) {
                        f32x2 m[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) m[j] = (f32x2){acc[3 * xi + j][ex][mt][2 * h], acc[3 * xi + j][ex][mt][2 * h + 1]};
                        if (ex == 0) {                                    // the last output of the last block column may fold
                            P0[xi] = m[0] + m[1] + fr2 * (m[2] - m[1]);
                            P1[xi] = m[1] + m[2] - 2.f * fr2 * m[2];
                        } else {                                          // the first output of block column 0 may fold
                            P0[xi] = m[0] + m[1] - 2.f * fl2 * m[0];
                            P1[xi] = m[1] + m[2] + fl2 * (m[0] - m[1]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x2* P = j == 0 ? P0 : P1;
                        if (rt == 0) { Y[ex][0][j] = P[0] + P[1]; Y[ex][1][j] = P[1] + P[2]; }
                        else if (rt == 1) { Y[ex][0][j] = P[1] - P[0]; Y[ex][1][j] = P[0] + P[2]; }
                        else { Y[ex][0][j] = P[0] + P[2]; Y[ex][1][j] = P[1] - P[2]; }
                    }
                }
                const int so = (16 * mt + 2 * h) * HWin * 4;
                f32x4 v[2][2];                                              // [channel of the pair][output row]
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) v[j][i] = (f32x4){Y[0][i][0][j], Y[1][i][0][j], Y[0][i][1][j], Y[1][i][1][j]};
                if (p.accumulate) {                                         // the four reads in flight before the first add
                    u32x4 o[2][2];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) o[j][i] = __builtin_amdgcn_raw_buffer_load_b128(ro, i == 0 ? vo0 : vo1, so + j * HWin * 4, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) v[j][i] += __builtin_bit_cast(f32x4, o[j][i]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j][i]), ro, i == 0 ? vo0 : vo1, so + j * HWin * 4, 0);
            }
        }
        if (stile >= ntotal) break;
        tile = stile;
    }
}

struct TapTable16d {
    int off[16];
};

// U = G g Gt of the 2x2 sub-filter of output parity (ey, ex): rows ky = (3, 1) for ey = 0, (2, 0) for ey = 1 (columns likewise);
// element w(k = gy channel, c = input channel, ky, kx) = src[c * so + k * sc + tap[ky * 4 + kx]];
// stored [ey][cin block][chunk][2 k-steps][4 k][128 = ex * 64 + (c & 63)][12]
__global__ void pack_s2dgrad_kernel(const float* __restrict__ src, float* __restrict__ upk, int kc, int cs, int csP, long so,
                                    long sc, TapTable16d tt) {
    const int nchunks = kc / 8;
    const long total = (long)4 * kc * csP;               // (ey, ex, k, c)
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % csP);
    const int k = (int)((e / csP) % kc);
    const int par = (int)(e / csP / kc);
    const int ey = par >> 1, ex = par & 1;
    const bool real = c < cs;
    float g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ky = (ey == 0 ? 3 : 2) - 2 * a, kx = (ex == 0 ? 3 : 2) - 2 * b;
            g[a][b] = real ? src[c * so + k * sc + tt.off[ky * 4 + kx]] : 0.f;
        }
    const int cblocks = csP / 64;
    float* base = upk + ((((size_t)(ey * cblocks + (c >> 6)) * nchunks + (k >> 3)) * 2 + ((k >> 2) & 1)) * 4 + (k & 3)) * 128 * D2_UP +
                  (size_t)(ex * 64 + (c & 63)) * D2_UP;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float u0 = i == 0 ? g[0][0] : (i == 1 ? g[0][0] + g[1][0] : g[1][0]);
        const float u1 = i == 0 ? g[0][1] : (i == 1 ? g[0][1] + g[1][1] : g[1][1]);
        base[i * 3 + 0] = u0;
        base[i * 3 + 1] = u0 + u1;
        base[i * 3 + 2] = u1;
    }
#pragma unroll
    for (int i = 9; i < D2_UP; ++i) base[i] = 0.f;
}

void init_hook() {
    C2S_RAISE_LDS(conv_s2dgrad_kernel);
}
C2sInitRegistrar registrar(init_hook);

}  // namespace

extern "C" size_t c2s_s2dgrad_packed_floats(int kc, int csP) {
    return (size_t)4 * kc * csP * D2_UP;
}

extern "C" int c2s_pack_weights_s2dgrad(const float* src, float* upk, int kc, int cs, int csP, long stride_c, long stride_k,
                                        const int* host_tap_off, void* stream) {
    C2S_REQUIRE(src && upk && host_tap_off && kc > 0 && kc % 8 == 0 && cs > 0 && csP % 64 == 0 && csP >= cs, "pack_s2dgrad: bad args");
    TapTable16d tt;
    for (int i = 0; i < 16; ++i) tt.off[i] = host_tap_off[i];
    const long total = (long)4 * kc * csP;
    hipLaunchKernelGGL(pack_s2dgrad_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, upk, kc, cs, csP,
                       stride_c, stride_k, tt);
    C2S_CHECK_LAUNCH("pack_s2dgrad");
    return C2S_OK;
}

// d: the FORWARD convolution's geometry seen from the gradient: C0 = gy channels (the forward Cout), Hin x Win = the gy plane
// (forward output), Cout / CoutP = the forward input channels of this source, Hout x Wout = 2 Hin x 2 Win = the gx plane
extern "C" int c2s_conv4x4s2_dgrad_winograd_supported(const c2s_conv_desc* d) {
    return d && d->KH == 4 && d->KW == 4 && d->S == 2 && d->pad_y == 1 && d->pad_x == 1 && d->C1 == 0 && d->C0 % 8 == 0 &&
           d->C0 >= 24 && d->Hout == 2 * d->Hin && d->Wout == 2 * d->Win && d->Hin % 2 == 0 && d->Win % 2 == 0 &&
           d->Win >= 32 && d->Hin >= 8 && d->CoutP % 64 == 0 && d->N <= 32768;
}

extern "C" int c2s_conv4x4s2_dgrad_winograd(const c2s_conv_desc* d, const float* gy, const float* upk, float* gx,
                                            const int* valid, void* stream) {
    C2S_REQUIRE(d && gy && upk && gx, "conv4x4s2_dgrad_winograd: null pointer");
    C2S_REQUIRE(c2s_conv4x4s2_dgrad_winograd_supported(d), "conv4x4s2_dgrad_winograd: data gradient of a 4x4 stride 2 pad 1 convolution, gy channels a multiple of 8 (>= 24), even gy planes at least 32 wide and 8 high, CoutP %% 64");
    C2S_REQUIRE(d->N > 0 && d->N <= 32768 && d->Cout > 0 && d->CoutP >= d->Cout,
                "conv4x4s2_dgrad_winograd: bad N / channels (at most 32768 frames: their flag bits share the last 4 KB of LDS)");
    C2S_REQUIRE(d->OutH == d->Hout && d->OutW == d->Wout && d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0,
                "conv4x4s2_dgrad_winograd: dense output only");
    C2S_REQUIRE((long)d->C0 * d->Hin * d->Win * 4 < 0x7FFF0000L && (long)d->CoutP * d->Hout * d->Wout * 4 < 0x7FFF0000L,
                "conv4x4s2_dgrad_winograd: frame too large");
    S2dParams p;
    p.src = gy; p.upk = upk; p.out = gx; p.valid = valid;
    p.Kc = d->C0; p.Ho = d->Hin; p.Wo = d->Win; p.Cs = d->Cout; p.CsP = d->CoutP;
    p.fold = d->reflect_adjoint; p.accumulate = d->accumulate;
    p.tiles_x = cdiv(d->Win, 32);
    p.tiles = p.tiles_x * cdiv(d->Hin, 8);
    p.N = d->N;
    p.nchunks = d->C0 / 8;
    const int cus = c2s_cus();
    const int yblocks = 2 * (d->CoutP / 64);
    const long ntotal = (long)d->N * p.tiles;
    long gx_ = ((long)cus + yblocks - 1) / yblocks;  // persistent: one 8-wave workgroup per CU
    if (gx_ > ntotal) gx_ = ntotal;
    dim3 grid((unsigned)gx_, yblocks, 1);
    const size_t ldsb = (size_t)(D2_LDS_FLOATS + (d->N + 31) / 32) * sizeof(float);
    hipLaunchKernelGGL(conv_s2dgrad_kernel, grid, dim3(512), ldsb, (hipStream_t)stream, p);
    C2S_CHECK_LAUNCH("conv4x4s2_dgrad_winograd");
    return C2S_OK;
}
