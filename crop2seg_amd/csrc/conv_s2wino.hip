// 4x4 stride-2 pad-1 convolution (forward) as Winograd F(2x2, 2x2) over the four input parities, on the structure of
// conv_winograd16.hip (8 waves, v_mfma_f32_16x16x4_f32, all transform points of a (channel, block) in one lane, operands by
// LDS-DMA into rings, requests two / three chunks ahead of the MFMAs).
//
// With x0[i] = x[2i], x1[i] = x[2i+1] (the two parities of a row) the 4-tap stride-2 filter is two 2-tap stride-1 filters with
// shifted windows:   y[i] = x1[i-1] w0 + x0[i] w1 + x1[i] w2 + x0[i+1] w3
//   parity 0: window (x0[i], x0[i+1]),   taps (w1, w3);      parity 1: window (x1[i-1], x1[i]),   taps (w0, w2)
// so a 2x2 output block needs a 3x3 patch of each of the four (row parity, column parity) planes, and F(2x2, 2x2)
//   Y = At [ sum_{c, parity} (G g Gt) (.) (Bt d B) ] A,   Bt = [[1,-1,0],[0,1,0],[0,-1,1]],  G = [[1,0],[1,1],[0,1]],  At = [[1,1,0],[0,1,1]]
// takes 9 multiplies per (input channel, parity) and block instead of 16: 36 instead of 64 per (cin, cout) pair and 2x2 block.
// The transform matrices are integer: the arithmetic is as exact as the direct form's.
// MFMA mapping: k index of the 16x16x4 MFMA = the parity (one k-step = one input channel), lane (t = block column, kq = parity)
// transforms the 3x3 patch of its parity plane; A operand = U[p][(c, parity)][o], 9 points padded to 12 floats (three
// conflict-free ds_read_b128).  A chunk is two input channels: 24 KB of U (two half slabs of a ring of five) and the eight
// parity planes of the 8 x 32 output tile (9 x 33 each, gathered by buffer_load_dword ... lds with reflected / stride-2 source
// addresses; ring of four).  Wave w: output channels 32 (w & 1) .., block rows 2 (w >> 1), 2 (w >> 1) + 1 (an A operand
// serves two block rows: half the U traffic per MFMA of a one-row wave); accumulators acc[9][2][2] of 16x16.
//
// Reference call sites replaced: nn.Conv2d(4, stride 2, padding 1, reflect) of DownConvBlock (src/backbones/conv.py:263-271)
// for layers with an even number (>= 8) of input channels on output planes at least 32 wide (the engine keeps
// conv_igemm_kernel<4,2,*> for the rest; C2S_S2WINO=0 keeps it everywhere).
#include "common.h"
#include <stdlib.h>

namespace {

struct S2wParams {
    const float* src;
    const float* upk;      // [cout block][chunk][2 c][4 parities][64 o][12] (9 points + 3 pad)
    const float* bias;
    float* out;
    const int* valid;
    int Cin, Hin, Win, H, W, Cout, CoutP;     // H, W: the output plane
    int pad_mode, accumulate;
    int N, tiles, tiles_x, nchunks;
};

constexpr int S2_UP = 12;                            // floats per (c, parity, o): 9 points + 3
constexpr int S2_UHALF = 4 * 64 * S2_UP;             // one k-step (one input channel, four parities): 3,072 floats = 12 KB
constexpr int S2_USLAB = 2 * S2_UHALF;               // a chunk of two input channels: 24 KB
constexpr int S2_URING = 5 * S2_UHALF;               // 60 KB
constexpr int S2_BR = 8, S2_BC = 16;                 // blocks per tile: 8 rows x 16 columns = 16 x 32 output pixels
constexpr int S2_RR = 2 * S2_BR + 1, S2_RC = 34;     // a parity plane of the tile: 17 rows x 33 columns, row pitch 34
constexpr int S2_PLANE = S2_RR * S2_RC;              // 578
constexpr int S2_XP = 608;                           // LDS pitch per plane (= 32 mod 64 banks)
constexpr int S2_MAXE = 10;                          // raw-tile LDS-DMA pieces per thread (one float each)
constexpr int S2_XS = S2_MAXE * 512;                 // 5,120 floats: 8 planes x 608 and a zero-filled tail
constexpr int S2_XSLOTS = 4;
constexpr int S2_LDS_FLOATS = S2_URING + S2_XSLOTS * S2_XS + 64;     // + the bias of the 64 channels (+ frame flag bits)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define C2S_AS1 __attribute__((address_space(1)))
#define C2S_AS3 __attribute__((address_space(3)))

__global__ __launch_bounds__(512, 1) void conv_s2wino_kernel(S2wParams p) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, kq = lane >> 4;        // block column / parity (B, D: channel quad) ; A: output channel / parity
    const int ch = w & 1, brow = w >> 1;            // channel half (32) and block row of this wave
    const int co0 = blockIdx.y * 64;
    const int HWin = p.Hin * p.Win, HWo = p.H * p.W;
    const bool reflect = p.pad_mode == C2S_PAD_REFLECT;
    const int ntotal = p.N * p.tiles;
    const int K = p.nchunks;

    // frame flags as a bit mask in LDS (the tile walk must not touch the vector-memory counter)
    unsigned* lvalid = reinterpret_cast<unsigned*>(lds + S2_URING + S2_XSLOTS * S2_XS + 64);
    for (int wi = tid; wi < (p.N + 31) / 32; wi += 512) {
        unsigned m = 0;
        for (int b = 0; b < 32; ++b) {
            const int f = wi * 32 + b;
            if (f < p.N && (p.valid == nullptr || p.valid[f] != 0)) m |= 1u << b;
        }
        lvalid[wi] = m;
    }
    float* lbias = lds + S2_URING + S2_XSLOTS * S2_XS;
    if (tid < 64) lbias[tid] = (p.bias != nullptr && co0 + tid < p.Cout) ? p.bias[co0 + tid] : 0.f;
    __syncthreads();
    auto next_valid = [&](int tt) __attribute__((always_inline)) {
        while (tt < ntotal) {
            const int f = tt / p.tiles;
            if ((lvalid[f >> 5] >> (f & 31)) & 1u) break;
            tt += gridDim.x;
        }
        return tt;
    };
    auto tile_origin = [&](int tt, int& n, int& oy0, int& ox0) __attribute__((always_inline)) {
        n = tt / p.tiles;
        const int ti = tt - n * p.tiles;
        const int tyi = ti / p.tiles_x, txi = ti - tyi * p.tiles_x;
        oy0 = tyi * 2 * S2_BR; ox0 = txi * 2 * S2_BC;
    };

    // ---- the staging side (its own tile state): plane pc = 4 c + parity of the chunk, parity = 2 py + px; plane element
    // (r, q) is the input pixel (2 (oy0 + r) - py, 2 (ox0 + q) - px)
    int goff[S2_MAXE];
    __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, 0, 0x00020000);
    auto begin_staging = [&](int tt) __attribute__((always_inline)) {
        int n, oy0, ox0;
        tile_origin(tt, n, oy0, ox0);
#pragma clang loop unroll(full)
        for (int i = 0; i < S2_MAXE; ++i) {
            const int e = tid + i * 512;
            const int pc = e / S2_XP, rem = e - pc * S2_XP;
            const int c = pc >> 2, py = (pc >> 1) & 1, px = pc & 1;
            const int r = rem / S2_RC, q = rem - r * S2_RC;
            int gy = 2 * (oy0 + r) - py, gx = 2 * (ox0 + q) - px;
            const bool ok = rem < S2_PLANE && q < 2 * S2_BC + 1 && pc < 8 &&
                            (reflect ? (gy <= p.Hin && gx <= p.Win) : (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win));
            gy = reflect_idx(gy, p.Hin);
            gx = reflect_idx(gx, p.Win);
            goff[i] = ok ? ((c * HWin + gy * p.Win + gx) * 4) : 0x7FFF0000;
        }
        r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + (size_t)n * p.Cin * HWin), 0, p.Cin * HWin * 4, 0x00020000);
    };
    // U chunk k -> half slabs h0, h0 + 1 (mod 5): 24 KB contiguous in global memory, three 16-byte pieces per thread; piece 1
    // straddles the halves at a wave boundary (waves 0-3 | 4-7)
    auto stage_u = [&](int k, int h0) __attribute__((always_inline)) {
        const int h1 = h0 == 4 ? 0 : h0 + 1;
        const C2S_AS1 char* g = (const C2S_AS1 char*)p.upk + ((size_t)blockIdx.y * K + k) * (S2_USLAB * 4);
        float* d0 = lds + h0 * S2_UHALF + w * 256;
        float* d1 = lds + h1 * S2_UHALF + w * 256 - S2_UHALF;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float* dst = (i == 0 || (i == 1 && w < 4)) ? d0 : d1;
            __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(g + i * 8192 + (unsigned)(tid * 16)), (C2S_AS3 void*)(dst + i * 2048), 16, 0, 0);
        }
    };
    auto stage_raw = [&](int k, int slot) __attribute__((always_inline)) {
        const int chan0 = 2 * k * HWin * 4;                      // scalar offset of the request
        float* Xd = lds + S2_URING + slot * S2_XS + w * 64;
#pragma clang loop unroll(full)
        for (int i = 0; i < S2_MAXE; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (C2S_AS3 void*)(Xd + i * 512), 4, goff[i], chan0, 0, 0);
    };
    // XCD-aware start: each XCD walks a contiguous eighth of the tiles in flight
    const int wg0 = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    int tile = next_valid(wg0);
    if (tile >= ntotal) return;
    int stile = tile, sk = 0;                          // next chunk to stage: chunk sk of tile stile (stile >= ntotal: none left)
    begin_staging(stile);
    // requests (uniform across the workgroup): raw planes three chunks ahead, U two ahead (see conv_winograd16.hip)
    bool u_ok = false, young_raw = false;
    int u_k = 0;
#ifndef C2S_S2W_DIAG
#define C2S_S2W_DIAG 0          // diagnostic builds: 1 = no U requests after the first chunks, 2 = no raw requests, 3 = neither
#endif
    int diag_n = 0;
    auto stage_next_u = [&](int h0) __attribute__((always_inline)) {
        if (u_ok && (!(C2S_S2W_DIAG & 1) || diag_n < 3)) stage_u(u_k, h0);
    };
    auto stage_next_raw = [&](int slot) __attribute__((always_inline)) {
        u_ok = stile < ntotal;
        u_k = sk;
        young_raw = u_ok;
        if (!u_ok) return;
        if (!(C2S_S2W_DIAG & 2) || diag_n < 3) stage_raw(sk, slot);
        ++diag_n;
        if (++sk == K) {                               // once per multiplied tile (K >= 4), at its chunk K - 4
            sk = 0;
            stile = next_valid(stile + gridDim.x);
            if (stile < ntotal) begin_staging(stile);
        }
    };
    int u0 = 0, rc = 0;                                // ring positions of the chunk being multiplied
    stage_next_raw(0); stage_next_u(0);
    stage_next_raw(1); stage_next_u(2);
    stage_next_raw(2);

    // LDS offsets of this lane's operands
    const int aoff = (kq * 64 + 32 * ch + t) * S2_UP;                          // in a half slab; + mt * 16 * UP
    const int boff = S2_URING + kq * S2_XP + (4 * brow) * S2_RC + 2 * t;       // in a raw slot (block row 2 brow); + s * 4 * XP + (2 br + r) * RC
    auto load_a = [&](const float* ab, int mt, float (&a)[12]) {                // (9 used)
#pragma unroll
        for (int q4 = 0; q4 < 3; ++q4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ab + mt * 16 * S2_UP + 4 * q4);
            a[4 * q4] = v[0]; a[4 * q4 + 1] = v[1]; a[4 * q4 + 2] = v[2]; a[4 * q4 + 3] = v[3];
        }
    };
    int boff0 = boff, boff1 = boff + 4 * S2_XP;        // (the two k-steps; opaque: the row offsets stay ds_read2 immediates)
    asm volatile("" : "+v"(boff0), "+v"(boff1));
    // the patches of the wave's two block rows share their middle row: five rows as (cols 0,1), (cols 2,3)
    auto load_d = [&](const float* bufp, int s, f32x2 (&dl)[5], f32x2 (&dh)[5]) {
        const float* bb = bufp + (s ? boff1 : boff0);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            dl[r] = *reinterpret_cast<const f32x2*>(bb + r * S2_RC);
            dh[r] = *reinterpret_cast<const f32x2*>(bb + r * S2_RC + 2);
        }
    };
    // V = Bt d B of the 3x3 patch (point p = 3 xi + nu)
    auto transform = [&](f32x2 l0, f32x2 l1, f32x2 l2, f32x2 h0, f32x2 h1, f32x2 h2, float (&V)[9]) {
        const f32x2 tl[3] = {l0 - l1, l1, l2 - l1};
        const float th[3] = {h0[0] - h1[0], h1[0], h2[0] - h1[0]};
#pragma unroll
        for (int xi = 0; xi < 3; ++xi) {
            V[3 * xi] = tl[xi][0] - tl[xi][1];
            V[3 * xi + 1] = tl[xi][1];
            V[3 * xi + 2] = th[xi] - tl[xi][1];
        }
    };
    f32x4 acc[9][2][2];                                // [point][channel group][block row]
    // first = the first k-step of a tile: accumulators start at 0 (inline constant), those of point (1,1) at the bias
    // (At e11 A = [[1,1],[1,1]])
    auto mma = [&](const float (&a)[12], const float (&V)[9], int mt, int br, bool first) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (first) {
                const f32x4 c0 = q == 4 ? *reinterpret_cast<const f32x4*>(lbias + 32 * ch + 16 * mt + 4 * kq) : (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[q][mt][br] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], V[q], c0, 0, 0, 0);
            } else {
                acc[q][mt][br] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], V[q], acc[q][mt][br], 0, 0, 0);
            }
        }
    };

    float a0[12], a1[12], V0[9], V1[9];
    f32x2 dl[5], dh[5];
    const int cof = co0 + 32 * ch + 4 * kq;            // this lane's output channels: cof + 16 mt + r (D rows 4 kq + r)
    __syncthreads();                                  // (drains the first requests: vmcnt(0))
    load_a(lds + aoff, 0, a0);
    load_d(lds, 0, dl, dh);
    while (true) {
        int n, oy0, ox0;
        tile_origin(tile, n, oy0, ox0);
        // one k-step (one input channel): four half-steps (channel group, block row) of 9 MFMAs; the A operand of a channel
        // group is read one group ahead; the next k-step's patch rows are read once both transforms have consumed the current
        auto kstep = [&](const float* A, const float* Anext, const float* Xnext, int snext, bool first) __attribute__((always_inline)) {
            load_a(A, 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            transform(dl[0], dl[1], dl[2], dh[0], dh[1], dh[2], V0);
            transform(dl[2], dl[3], dl[4], dh[2], dh[3], dh[4], V1);
            __builtin_amdgcn_sched_barrier(0);
            load_d(Xnext, snext, dl, dh);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, V0, 0, 0, first);
            mma(a0, V1, 0, 1, first);
            __builtin_amdgcn_sched_barrier(0);
            load_a(Anext, 0, a0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, V0, 1, 0, first);
            mma(a1, V1, 1, 1, first);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto chunk = [&](bool first) __attribute__((always_inline)) {
            const int u1 = u0 == 4 ? 0 : u0 + 1, u2 = u1 == 4 ? 0 : u1 + 1;
            kstep(lds + u0 * S2_UHALF + aoff, lds + u1 * S2_UHALF + aoff, lds + rc * S2_XS, 1, first);
            // everyone's requests for the NEXT chunk have landed; the youngest raw pieces (chunk after next) may stay in flight
            if (young_raw) __builtin_amdgcn_s_waitcnt(0x0F7A);      // vmcnt(10)
            else __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stage_next_u(u0 == 0 ? 4 : u0 - 1);
            stage_next_raw((rc + 3) & 3);
            // (after the last chunk of the last tile the tail reads are stale and unused)
            kstep(lds + u1 * S2_UHALF + aoff, lds + u2 * S2_UHALF + aoff, lds + ((rc + 1) & 3) * S2_XS, 0, false);
            u0 = u2;
            rc = (rc + 1) & 3;
        };
        chunk(true);
        for (int k = 1; k < K; ++k) chunk(false);
        // ---- epilogue: At M A on channel pairs; P[xi][0] = M[xi][0] + M[xi][1], P[xi][1] = M[xi][1] + M[xi][2];
        // Y[0][j] = P[0][j] + P[1][j], Y[1][j] = P[1][j] + P[2][j]; buffer stores as in conv_winograd16.hip
        const int ox = ox0 + 2 * t;
        const __amdgpu_buffer_rsrc_t ro =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)n * p.Cout * HWo), 0, p.Cout * HWo * 4, 0x00020000);
#pragma unroll
        for (int br = 0; br < 2; ++br) {
        const int oy = oy0 + 2 * (2 * brow + br);
        const bool in0 = ox < p.W && oy < p.H, in1 = in0 && oy + 1 < p.H;
        const int vo0 = in0 ? (cof * HWo + oy * p.W + ox) * 4 : 0x7FFF0000;
        const int vo1 = in1 ? vo0 + p.W * 4 : 0x7FFF0000;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 P0[3], P1[3];
#pragma unroll
                for (int xi = 0; xi < 3; ++xi) {
                    f32x2 m[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) m[j] = (f32x2){acc[3 * xi + j][mt][br][2 * h], acc[3 * xi + j][mt][br][2 * h + 1]};
                    P0[xi] = m[0] + m[1];
                    P1[xi] = m[1] + m[2];
                }
                const f32x2 Y00 = P0[0] + P0[1], Y01 = P1[0] + P1[1], Y10 = P0[1] + P0[2], Y11 = P1[1] + P1[2];
                f32x2 y[2][2] = {{(f32x2){Y00[0], Y01[0]}, (f32x2){Y10[0], Y11[0]}},        // [channel of the pair][output row]
                                 {(f32x2){Y00[1], Y01[1]}, (f32x2){Y10[1], Y11[1]}}};
                const int so = (16 * mt + 2 * h) * HWo * 4;
                if (p.accumulate) {
                    u32x2 o[2][2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        o[j][0] = __builtin_amdgcn_raw_buffer_load_b64(ro, vo0, so + j * HWo * 4, 0);
                        o[j][1] = __builtin_amdgcn_raw_buffer_load_b64(ro, vo1, so + j * HWo * 4, 0);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        y[j][0] += __builtin_bit_cast(f32x2, o[j][0]);
                        y[j][1] += __builtin_bit_cast(f32x2, o[j][1]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, y[j][0]), ro, vo0, so + j * HWo * 4, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, y[j][1]), ro, vo1, so + j * HWo * 4, 0);
                }
            }
        }
        }
        if (stile >= ntotal) break;
        tile = stile;
    }
}

struct TapTable16s {
    int off[16];
};

// U = G g Gt of the four 2x2 parity sub-filters of the 4x4 filter w[ky][kx] = src[o*so + c*sc + tap[ky*4+kx]]:
// parity 0 of a dimension uses taps (1, 3), parity 1 taps (0, 2); stored [cout block][chunk][2 c][4 parities][64 o][12]
__global__ void pack_s2wino_kernel(const float* __restrict__ src, float* __restrict__ upk, int cin, int cout, int coutP,
                                   long so, long sc, TapTable16s tt) {
    const int nchunks = (cin + 1) / 2;
    const long total = (long)nchunks * 2 * 4 * coutP;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int o = (int)(e % coutP);
    const int par = (int)((e / coutP) & 3), c = (int)(e / coutP / 4);
    const int py = par >> 1, px = par & 1;
    const bool real = o < cout && c < cin;
    float g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ky = (py == 0 ? 1 : 0) + 2 * a, kx = (px == 0 ? 1 : 0) + 2 * b;
            g[a][b] = real ? src[o * so + c * sc + tt.off[ky * 4 + kx]] : 0.f;
        }
    float u[3][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) { u[0][b] = g[0][b]; u[1][b] = g[0][b] + g[1][b]; u[2][b] = g[1][b]; }
    float* base = upk + ((((size_t)(o >> 6) * nchunks + (c >> 1)) * 2 + (c & 1)) * 4 + par) * 64 * S2_UP + (size_t)(o & 63) * S2_UP;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        base[i * 3 + 0] = u[i][0];
        base[i * 3 + 1] = u[i][0] + u[i][1];
        base[i * 3 + 2] = u[i][1];
    }
#pragma unroll
    for (int i = 9; i < S2_UP; ++i) base[i] = 0.f;
}

void init_hook() {
    C2S_RAISE_LDS(conv_s2wino_kernel);
}
C2sInitRegistrar registrar(init_hook);

}  // namespace

extern "C" size_t c2s_s2wino_packed_floats(int cin, int coutP) {
    return (size_t)((cin + 1) / 2) * 2 * 4 * coutP * S2_UP;
}

extern "C" int c2s_pack_weights_s2wino(const float* src, float* upk, int cin, int cout, int coutP, long stride_o,
                                       long stride_c, const int* host_tap_off, void* stream) {
    C2S_REQUIRE(src && upk && host_tap_off && cin > 0 && cout > 0 && coutP % 64 == 0 && coutP >= cout, "pack_s2wino: bad args");
    TapTable16s tt;
    for (int i = 0; i < 16; ++i) tt.off[i] = host_tap_off[i];
    const long total = (long)((cin + 1) / 2) * 2 * 4 * coutP;
    hipLaunchKernelGGL(pack_s2wino_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, upk, cin, cout,
                       coutP, stride_o, stride_c, tt);
    C2S_CHECK_LAUNCH("pack_s2wino");
    return C2S_OK;
}

extern "C" int c2s_conv4x4s2_winograd_supported(const c2s_conv_desc* d) {
    return d && d->KH == 4 && d->KW == 4 && d->S == 2 && d->pad_y == 1 && d->pad_x == 1 && d->C1 == 0 && d->C0 % 2 == 0 &&
           d->C0 >= 8 && d->Hin == 2 * d->Hout && d->Win == 2 * d->Wout && d->Hout % 2 == 0 && d->Wout % 2 == 0 &&
           d->Wout >= 32 && d->Hout >= 8 && d->CoutP % 64 == 0 && d->reflect_adjoint == 0 && d->N <= 65536;
}

extern "C" int c2s_conv4x4s2_winograd(const c2s_conv_desc* d, const float* src, const float* upk, const float* bias,
                                      float* out, const int* valid, void* stream) {
    C2S_REQUIRE(d && src && upk && out, "conv4x4s2_winograd: null pointer");
    C2S_REQUIRE(c2s_conv4x4s2_winograd_supported(d), "conv4x4s2_winograd: 4x4 stride 2 pad 1, one source with an even number (>= 8) of channels, even output planes at least 32 wide and 8 high, CoutP %% 64");
    C2S_REQUIRE(d->N > 0 && d->N <= 65536 && d->Cout > 0 && d->CoutP >= d->Cout, "conv4x4s2_winograd: bad N / Cout");
    C2S_REQUIRE(d->OutH == d->Hout && d->OutW == d->Wout && d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0,
                "conv4x4s2_winograd: dense output only");
    C2S_REQUIRE((long)d->C0 * d->Hin * d->Win * 4 < 0x7FFF0000L && (long)d->CoutP * d->Hout * d->Wout * 4 < 0x7FFF0000L,
                "conv4x4s2_winograd: frame too large");
    S2wParams p;
    p.src = src; p.upk = upk; p.bias = bias; p.out = out; p.valid = valid;
    p.Cin = d->C0; p.Hin = d->Hin; p.Win = d->Win; p.H = d->Hout; p.W = d->Wout; p.Cout = d->Cout; p.CoutP = d->CoutP;
    p.pad_mode = d->pad_mode; p.accumulate = d->accumulate;
    p.tiles_x = cdiv(d->Wout, 2 * S2_BC);
    p.tiles = p.tiles_x * cdiv(d->Hout, 2 * S2_BR);
    p.N = d->N;
    p.nchunks = d->C0 / 2;
    const int cus = c2s_cus();
    const int cblocks = d->CoutP / 64;
    const long ntotal = (long)d->N * p.tiles;
    long gx = ((long)cus + cblocks - 1) / cblocks;  // persistent: one 8-wave workgroup per CU
    if (gx > ntotal) gx = ntotal;
    dim3 grid((unsigned)gx, cblocks, 1);
    const size_t ldsb = (size_t)(S2_LDS_FLOATS + (d->N + 31) / 32) * sizeof(float);
    hipLaunchKernelGGL(conv_s2wino_kernel, grid, dim3(512), ldsb, (hipStream_t)stream, p);
    C2S_CHECK_LAUNCH("conv4x4s2_winograd");
    return C2S_OK;
}
