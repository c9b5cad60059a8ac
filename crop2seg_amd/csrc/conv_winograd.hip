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
#include <stdlib.h>

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
    int N, tiles, tiles_x, nchunks;
};

constexpr int WN_CK = 8;
constexpr int WN_USLAB = 16 * WN_CK * 64;          // floats of one U chunk
constexpr int WN_EXCH = 4 * 2 * 16 * 64;           // floats of the epilogue exchange [xi][x][row group][lane][4 rows] (one 32-channel half)

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef C2S_WN_STAMP
// diagnostic build only: per-workgroup cycles spent in each phase, summed over the tiles the workgroup processed
// (slot k accumulates the time between stamp k-1 and stamp k; slot 0 counts tiles)
__device__ unsigned long long wn_stamps[8192 * 8];
#define WN_STAMP_DECL unsigned long long wn_t_ = __builtin_amdgcn_s_memtime(), wn_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WN_STAMP(k)                                                     \
    do {                                                                \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();  \
        wn_acc_[k] += (k) == 0 ? 1 : now__ - wn_t_;                     \
        wn_t_ = now__;                                                  \
    } while (0)
#define WN_STAMP_FLUSH                                                                                    \
    do {                                                                                                  \
        const int wg__ = blockIdx.y * gridDim.x + blockIdx.x;                                             \
        if (threadIdx.x == 0 && wg__ < 8192)                                                              \
            for (int k__ = 0; k__ < 8; ++k__) wn_stamps[wg__ * 8 + k__] = wn_acc_[k__];                   \
    } while (0)
#else
#define WN_STAMP_DECL
#define WN_STAMP(k)
#define WN_STAMP_FLUSH
#endif

// Persistent workgroups: each walks tiles t = blockIdx.x, +gridDim.x, ... of (frame, tile) order; the requests for the
// next tile's first chunk are issued before the epilogue of the current tile, so their latency hides behind the
// exchange and the stores instead of sitting in front of the next K loop.
// Staging: the U chunk goes global -> LDS directly (global_load_lds_dwordx4: its LDS image [xn][c][o] is lane-linear),
// which keeps 32 staging registers free next to the 128 accumulators; the gathered raw tile is register-staged.
#define C2S_AS1 __attribute__((address_space(1)))
#define C2S_AS3 __attribute__((address_space(3)))

template <int LOG2BC, bool ADJ>
__global__ __launch_bounds__(256, 2) void conv_winograd_kernel(WinoParams p) {
    constexpr int CK = WN_CK;
    constexpr int BC = 1 << LOG2BC, BR = 32 >> LOG2BC;
    constexpr int RR = 2 * BR + 2, RC = 2 * BC + 2;
    constexpr int plane = RR * RC;                // even
    constexpr int total = CK * plane;
    constexpr int MAXE = (total + 255) / 256;
    constexpr int NWV = WN_USLAB / 4;             // float4 items of the U chunk
    constexpr int WPT = NWV / 256;
    constexpr int NS = CK / 2;                    // k-steps (channel pairs) per chunk
    constexpr int xsz = total;
    constexpr int BUF = xsz + WN_USLAB;
    static_assert(CK == 8 && WN_EXCH <= BUF, "chunk decomposition / exchange area");
    extern __shared__ float lds[];
    float* Xl = lds;                              // [2][ [CK][plane] | [16][CK][64] ]
    float* Wl = lds + xsz;

    const int co0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HW = p.H * p.W;
    const int lane = tid & 63, xi = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int by = li >> LOG2BC, bx = li & (BC - 1);
    // patch rows combined by this wave: t = ca * d[ra] + cb * d[rb]   (rows of Bt)
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const int boffa = lk * plane + (2 * by + ra) * RC + 2 * bx;
    const int boffb = lk * plane + (2 * by + rb) * RC + 2 * bx;
    const int aoff = (xi * 4 * CK + lk) * 64 + li;
    const bool reflect = p.pad_mode == C2S_PAD_REFLECT;

    // staging element of this thread packed as (channel << 20 | tile row << 10 | tile col); -1 = none
    int epk[MAXE];
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = tid + i * 256;
        const int c = e / plane, rem = e % plane;     // compile-time divisors
        epk[i] = e < total ? ((c << 20) | ((rem / RC) << 10) | (rem % RC)) : -1;
    }

    // ---- per-tile state
    int goff[MAXE];
    int n = 0, oy0 = 0, ox0 = 0;
    float ca = 1.f, cb = xi == 1 ? 1.f : -1.f, e0 = 1.f, e3 = 1.f;
    __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src0, 0, 0, 0x00020000), r1 = r0;
    const int ntotal = p.N * p.tiles;

    auto next_valid = [&](int t) {
        while (t < ntotal && p.valid != nullptr && p.valid[t / p.tiles] == 0) t += gridDim.x;
        return t;
    };
    auto begin_tile = [&](int t) {
        n = t / p.tiles;
        const int tt = t - n * p.tiles;
        const int tyi = tt / p.tiles_x, txi = tt - tyi * p.tiles_x;
        oy0 = tyi * 2 * BR; ox0 = txi * 2 * BC;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int pk = epk[i];
            int gy = oy0 - 1 + ((pk >> 10) & 1023), gx = ox0 - 1 + (pk & 1023);
            bool ok = pk >= 0;
            if (reflect) {
                ok = ok && gy <= p.H && gx <= p.W;       // gy, gx >= -1 by construction
                gy = reflect_idx(gy, p.H);
                gx = reflect_idx(gx, p.W);
            } else {
                ok = ok && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            }
            goff[i] = ok ? (((pk >> 20) * HW + gy * p.W + gx) * 4) : -1;
        }
        if constexpr (ADJ) {
            const int gby = (oy0 >> 1) + by, gbx = (ox0 >> 1) + bx;      // global block coordinates
            const bool top = gby == 0, bottom = gby == (p.H >> 1) - 1;
            const bool left = gbx == 0, right = gbx == (p.W >> 1) - 1;
            ca = (xi == 3 && top) ? 0.f : 1.f;                      // d3 += d1  ->  d1 - (d3 + d1)
            cb = xi == 1 ? 1.f : ((xi == 0 && bottom) ? 0.f : -1.f); // d0 += d2  ->  (d0 + d2) - d2
            e0 = right ? 0.f : 1.f;                                 // col0 += col2
            e3 = left ? 0.f : 1.f;                                  // col3 += col1
        }
        const float* s0n = p.src0 + (size_t)n * p.C0 * HW;
        const float* s1n = p.src1 != nullptr ? p.src1 + (size_t)n * p.C1 * HW : nullptr;
        r0 = __builtin_amdgcn_make_buffer_rsrc((void*)s0n, 0, p.C0 * HW * 4, 0x00020000);
        r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(s1n != nullptr ? s1n : s0n), 0, p.C1 * HW * 4, 0x00020000);
    };

    float xr[MAXE];
    // raw tile of chunk cb_: global -> registers (out-of-range offsets return 0: padding, channels past Cin);
    // one request per call so that the K loop can place them between MFMAs
    auto prefetch_raw_piece = [&](int cb_, int i) {
        const bool first = cb_ < p.C0;
        const int chan0 = (first ? cb_ : cb_ - p.C0) * HW * 4;
        const int off = goff[i] >= 0 ? goff[i] + chan0 : -1;
        xr[i] = __builtin_bit_cast(float, first ? __builtin_amdgcn_raw_buffer_load_b32(r0, off, 0, 0)
                                                : __builtin_amdgcn_raw_buffer_load_b32(r1, off, 0, 0));
    };
    auto prefetch_raw = [&](int cb_) {
#pragma unroll
        for (int i = 0; i < MAXE; ++i) prefetch_raw_piece(cb_, i);
    };
    auto commit_raw = [&](int buf) {
        float* Xd = Xl + buf * BUF;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int e = tid + i * 256;
            if (e < total) Xd[e] = xr[i];
        }
    };
    // U chunk cb_: global -> LDS.  The packed layout [cout block][chunk][xn 16][c 8][o 64] makes a chunk one contiguous
    // 32 KB run that is copied verbatim (float4 item e lands at byte 16*e of the slab: lane-linear, as LDS-DMA needs),
    // and spreads consecutive chunks over all L2 channels (a [xn][cin][cout] layout put the 16 pieces of a chunk
    // 16 KB apart = on 4 of the 16 channels of the XCD's L2).
    const float* ublock = p.upk + (size_t)blockIdx.y * p.nchunks * WN_USLAB + tid * 4;
    auto stage_u_piece = [&](int cb_, int buf, int i) {
        float* Wd = Wl + buf * BUF + tid * 4;
        const float* g = ublock + (size_t)(cb_ / CK) * WN_USLAB;
        __builtin_amdgcn_global_load_lds((const C2S_AS1 void*)(g + i * 1024), (C2S_AS3 void*)(Wd + i * 1024), 16, 0, 0);
    };
    auto stage_u = [&](int cb_, int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) stage_u_piece(cb_, buf, i);
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

    // XCD-aware start: workgroups are dealt to the 8 XCDs round-robin, so workgroup b and b+1 have different L2s.  Give each
    // XCD a contiguous eighth of the 2*CUs tiles in flight (half a frame at 128x128): vertically adjacent tiles, which share
    // two of their six input rows, then meet in the same L2 instead of fetching the halo from HBM twice.
    const int wg0 = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    int t = next_valid(wg0);
    if (t >= ntotal) return;
    WN_STAMP_DECL
    begin_tile(t);
    WN_STAMP(1);
    int cur = 0;
    stage_u(0, cur);
    prefetch_raw(0);
    while (true) {
        f32x16 acc[4][2];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[v][m][r] = 0.f;
        WN_STAMP(0);
        commit_raw(cur);
        __syncthreads();                              // (also drains the U chunk's LDS-DMA: vmcnt(0))
        WN_STAMP(2);
        for (int cb_ = 0; cb_ < Cin; cb_ += CK, cur ^= 1) {
            const float* Xc = Xl + cur * BUF;
            const float* Wc = Wl + cur * BUF;
            const bool nextc = cb_ + CK < Cin;
            f32x2 d[2][4];
            float a[2][4][2];
            load_ops(Xc, Wc, 0, d[0], a[0]);
            __builtin_amdgcn_sched_barrier(0);
            static_assert(WPT + MAXE <= 2 * 8, "the next chunk's requests are spread over the MFMAs of two k-steps");
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                float V[4];
                transform(d[s_ & 1], V);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int v = j >> 1, m = j & 1;
                    acc[v][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][v][m], V[v], acc[v][m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // behind the MFMA just issued (each runs 64 cycles): the LDS reads of the next step (j == 0) and one
                    // request of the next chunk per MFMA -- an LDS-DMA piece costs ~60-100 issue cycles, eight of them in
                    // a row at the top of the chunk left the MFMA pipe idle for most of a k-step
                    if (j == 0) {
                        if (s_ + 1 < NS) load_ops(Xc, Wc, s_ + 1, d[(s_ + 1) & 1], a[(s_ + 1) & 1]);
                        if (s_ == NS - 1 && nextc) commit_raw(cur ^ 1);
                    }
                    const int q = s_ * 8 + j;
                    if (nextc) {                      // the other buffer was last read one chunk ago (barrier since)
#ifndef C2S_WN_NO_U
                        if (q < WPT) stage_u_piece(cb_ + CK, cur ^ 1, q);
#endif
#ifndef C2S_WN_NO_RAW
                        if (q >= WPT && q < WPT + MAXE) prefetch_raw_piece(cb_ + CK, q - WPT);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#ifndef C2S_WN_NO_BAR
            __syncthreads();
#endif
        }
        WN_STAMP(3);
        // `cur` now names the buffer the NEXT chunk would use: it was last read two barriers ago and is free; the
        // exchange area of the epilogue lives in the other one.

        // ---- finished tile: output position of this lane's 2x2 block; then start the next tile's first requests
        const int oy = oy0 + 2 * by, ox = ox0 + 2 * bx;
        const bool inb = oy < p.H && ox < p.W;
        float* on = p.out + (size_t)n * p.Cout * HW + (size_t)oy * p.W + ox;
        t = next_valid(t + gridDim.x);
        const bool more = t < ntotal;
        if (more) {
            begin_tile(t);
            stage_u(0, cur);                          // in flight during the epilogue
            prefetch_raw(0);
        }

        // ---- epilogue, one 32-channel half at a time.  nu side of the output transform in registers:
        //      P[x] = sum_nu M[nu] A[nu][x]; exchange [xi 4][x 2][r 16][lane 64]
        WN_STAMP(4);
        // Barriers of the epilogue order LDS traffic only: a raw s_barrier behind lgkmcnt(0) -- __syncthreads() would
        // also wait for vmcnt(0), i.e. for the next tile's requests that are meant to stay in flight here.
        auto lds_barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        float* ex = lds + (cur ^ 1) * BUF;
        const bool odd = (lane & 1) != 0;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m == 1) lds_barrier();                // first half fully read
            // exchange layout [xi 4][x 2][row group 4][lane 64][4 rows]: one 16-byte write per (x, row group) and one 16-byte
            // read per (xi, x) of the row group a wave finishes (a quarter of the LDS instructions of a [..][r][lane] layout)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 p0, p1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rg + e;
                    const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r];
                    p0[e] = m0 + m1 + m2;
                    p1[e] = m1 - m2 - m3;
                }
                *reinterpret_cast<f32x4*>(ex + ((xi * 2 + 0) * 4 + rg) * 256 + lane * 4) = p0;
                *reinterpret_cast<f32x4*>(ex + ((xi * 2 + 1) * 4 + rg) * 256 + lane * 4) = p1;
            }
            lds_barrier();
            if (m == 0) WN_STAMP(5); else WN_STAMP(7);
            // xi side: Y[0][x] = P0 + P1 + P2, Y[1][x] = P1 - P2 - P3; wave w finishes accumulator rows r with r>>2 == w.
            // Neighbouring lanes (blocks bx, bx+1) swap one row each so that every lane stores ONE float4 (4 pixels of
            // a row): even lanes write row 0 of both blocks, odd lanes row 1.
            f32x4 Pq[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int x = 0; x < 2; ++x) Pq[q][x] = *reinterpret_cast<const f32x4*>(ex + ((q * 2 + x) * 4 + xi) * 256 + lane * 4);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = xi * 4 + rr;
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                float P[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int x = 0; x < 2; ++x) P[q][x] = Pq[q][x][rr];
                const float y00 = P[0][0] + P[1][0] + P[2][0], y01 = P[0][1] + P[1][1] + P[2][1];
                const float y10 = P[1][0] - P[2][0] - P[3][0], y11 = P[1][1] - P[2][1] - P[3][1];
                // send the row the partner stores, receive the partner's share of the row this lane stores
                const float s0 = odd ? y00 : y10, s1 = odd ? y01 : y11;
                const float g0 = __shfl_xor(s0, 1, 64), g1 = __shfl_xor(s1, 1, 64);
                f32x4 v = odd ? f32x4{g0, g1, y10, y11} : f32x4{y00, y01, g0, g1};
                if (inb && co < p.Cout) {
                    if (p.bias != nullptr) v += p.bias[co];
                    // even lane: row oy, columns ox..ox+3 ; odd lane: row oy+1, columns ox-2..ox+1
                    f32x4* dst = reinterpret_cast<f32x4*>(on + (size_t)co * HW + (odd ? p.W - 2 : 0));
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
        WN_STAMP(6);
        if (!more) break;
        // no barrier needed here: the next commit writes the X area of buffer `cur`, the exchange lived in the other
        // buffer, and the barrier after that commit orders every later write to it.
    }
    WN_STAMP_FLUSH;
}

struct TapTable9 {
    int off[9];
};

// U = G g Gt of the 3x3 filter g[k] = src[o*so + c*sc + tap[k]], stored [cout block][chunk][xn 16][c 8][o 64]
// (zero for channels / outputs past the real counts)
__global__ void pack_winograd_kernel(const float* __restrict__ src, float* __restrict__ upk, int cin, int cout, int coutP,
                                     long so, long sc, TapTable9 tt) {
    const int nchunks = (cin + WN_CK - 1) / WN_CK;
    const long total = (long)nchunks * WN_CK * coutP;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int o = (int)(e % coutP), c = (int)(e / coutP);
    const bool real = o < cout && c < cin;
    float g[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = real ? src[o * so + c * sc + tt.off[k]] : 0.f;
    float t[4][3];                                  // G g
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
    float* base = upk + ((size_t)(o >> 6) * nchunks + (c >> 3)) * WN_USLAB + (c & 7) * 64 + (o & 63);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        base[(i * 4 + 0) * WN_CK * 64] = t[i][0];
        base[(i * 4 + 1) * WN_CK * 64] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        base[(i * 4 + 2) * WN_CK * 64] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        base[(i * 4 + 3) * WN_CK * 64] = t[i][2];
    }
}

// ---- all per-step weight packs in one launch: a device table of jobs (plain tap-major packs for the direct / transposed
// kernels and Winograd transforms); block -> job by the table's block offsets
struct PackJob {
    const float* src;
    float* dst;
    long so, sc;
    int cin, cout, coutP, ntaps;
    int kind;            // 0: wpk[tap][cin][coutP]   1: Winograd U   2: Winograd U in the 8-wave kernel's layout   3: conv_s2wino.hip's   4: conv_s2dgrad.hip's
    int block_start;     // first block of this job
    int taps[16];
};

__global__ void pack_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].block_start) ++j;      // <= ~64 jobs: linear scan
    const PackJob& jb = jobs[j];
    const long e = (long)(blockIdx.x - jb.block_start) * blockDim.x + threadIdx.x;
    if (jb.kind == 0) {
        const long total = (long)jb.ntaps * jb.cin * jb.coutP;
        if (e >= total) return;
        const int o = (int)(e % jb.coutP);
        const long tc = e / jb.coutP;
        const int c = (int)(tc % jb.cin), t = (int)(tc / jb.cin);
        jb.dst[e] = o < jb.cout ? jb.src[o * jb.so + c * jb.sc + jb.taps[t]] : 0.f;
        return;
    }
    if (jb.kind == 3) {          // F(2x2,2x2) parity sub-filters of a 4x4 stride-2 filter (conv_s2wino.hip): [cout block][chunk][2 c][4][64 o][12]
        const int nch2 = (jb.cin + 1) / 2;
        const long total3 = (long)nch2 * 2 * 4 * jb.coutP;
        if (e >= total3) return;
        const int o = (int)(e % jb.coutP);
        const int par = (int)((e / jb.coutP) & 3), c = (int)(e / jb.coutP / 4);
        const int py = par >> 1, px = par & 1;
        const bool real = o < jb.cout && c < jb.cin;
        float g2[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ky = (py == 0 ? 1 : 0) + 2 * a, kx = (px == 0 ? 1 : 0) + 2 * b;
                g2[a][b] = real ? jb.src[o * jb.so + c * jb.sc + jb.taps[ky * 4 + kx]] : 0.f;
            }
        float* base = jb.dst + ((((size_t)(o >> 6) * nch2 + (c >> 1)) * 2 + (c & 1)) * 4 + par) * 64 * 12 + (size_t)(o & 63) * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float u0 = i == 0 ? g2[0][0] : (i == 1 ? g2[0][0] + g2[1][0] : g2[1][0]);
            const float u1 = i == 0 ? g2[0][1] : (i == 1 ? g2[0][1] + g2[1][1] : g2[1][1]);
            base[i * 3 + 0] = u0;
            base[i * 3 + 1] = u0 + u1;
            base[i * 3 + 2] = u1;
        }
        return;
    }
    if (jb.kind == 4) {          // per-parity sub-filters of the 4x4 stride-2 data gradient (conv_s2dgrad.hip); here cin = gy
                                 // channels k, cout / coutP = input channels c: element at src[c * so + k * sc + tap]
        const long total4 = (long)4 * jb.cin * jb.coutP;
        if (e >= total4) return;
        const int c = (int)(e % jb.coutP);
        const int k = (int)((e / jb.coutP) % jb.cin);
        const int par = (int)(e / jb.coutP / jb.cin);
        const int ey = par >> 1, ex = par & 1;
        const bool real = c < jb.cout;
        float g2[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ky = (ey == 0 ? 3 : 2) - 2 * a, kx = (ex == 0 ? 3 : 2) - 2 * b;
                g2[a][b] = real ? jb.src[c * jb.so + k * jb.sc + jb.taps[ky * 4 + kx]] : 0.f;
            }
        const int cblocks = jb.coutP / 64, nch8 = jb.cin / 8;
        float* base = jb.dst + ((((size_t)(ey * cblocks + (c >> 6)) * nch8 + (k >> 3)) * 2 + ((k >> 2) & 1)) * 4 + (k & 3)) * 128 * 12 +
                      (size_t)(ex * 64 + (c & 63)) * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float u0 = i == 0 ? g2[0][0] : (i == 1 ? g2[0][0] + g2[1][0] : g2[1][0]);
            const float u1 = i == 0 ? g2[0][1] : (i == 1 ? g2[0][1] + g2[1][1] : g2[1][1]);
            base[i * 3 + 0] = u0;
            base[i * 3 + 1] = u0 + u1;
            base[i * 3 + 2] = u1;
        }
        return;
    }
    const int nchunks = (jb.cin + WN_CK - 1) / WN_CK;
    const long total = (long)nchunks * WN_CK * jb.coutP;
    if (e >= total) return;
    const int o = (int)(e % jb.coutP), c = (int)(e / jb.coutP);
    const bool real = o < jb.cout && c < jb.cin;
    float g[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = real ? jb.src[o * jb.so + c * jb.sc + jb.taps[k]] : 0.f;
    float t[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        t[0][q] = g[0][q];
        t[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
        t[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
        t[3][q] = g[2][q];
    }
    if (jb.kind == 2) {          // the 8-wave kernel's layout (conv_winograd16.hip): [cout block][chunk][8 c][4 xi][64 o][4 nu]
        float* wide = jb.dst + (((size_t)(o >> 6) * nchunks + (c >> 3)) * WN_CK + (c & 7)) * 64 * 16 + (size_t)(o & 63) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* w4 = wide + i * 256;
            w4[0] = t[i][0];
            w4[1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
            w4[2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
            w4[3] = t[i][2];
        }
        return;
    }
    float* base = jb.dst + ((size_t)(o >> 6) * nchunks + (c >> 3)) * WN_USLAB + (c & 7) * 64 + (o & 63);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        base[(i * 4 + 0) * WN_CK * 64] = t[i][0];
        base[(i * 4 + 1) * WN_CK * 64] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        base[(i * 4 + 2) * WN_CK * 64] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        base[(i * 4 + 3) * WN_CK * 64] = t[i][2];
    }
}

void init_hook() {
    C2S_RAISE_LDS((conv_winograd_kernel<4, false>));
    C2S_RAISE_LDS((conv_winograd_kernel<4, true>));
    C2S_RAISE_LDS((conv_winograd_kernel<3, false>));
    C2S_RAISE_LDS((conv_winograd_kernel<3, true>));
    C2S_RAISE_LDS((conv_winograd_kernel<2, false>));
    C2S_RAISE_LDS((conv_winograd_kernel<2, true>));
}
C2sInitRegistrar registrar(init_hook);

}  // namespace

extern "C" size_t c2s_winograd_packed_floats(int cin, int coutP) {
    return (size_t)(coutP / 64) * cdiv(cin, WN_CK) * WN_USLAB;
}

#ifdef C2S_WN_STAMP
extern "C" int c2s_debug_winograd_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wn_stamps), sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t c2s_pack_job_bytes(void) { return sizeof(PackJob); }

// Fill one job record of a host-side table (the caller uploads the table once and reuses it every step)
extern "C" int c2s_pack_job_fill(void* host_record, const float* src, float* dst, int cin, int cout, int coutP, int ntaps,
                                 long stride_o, long stride_c, int winograd, const int* host_tap_off, int block_start) {
    C2S_REQUIRE(host_record && src && dst && host_tap_off && ntaps >= 1 && ntaps <= 16 && winograd >= 0 && winograd <= 4,
                "pack_job_fill: bad args");
    PackJob* j = reinterpret_cast<PackJob*>(host_record);
    j->src = src; j->dst = dst; j->so = stride_o; j->sc = stride_c;
    j->cin = cin; j->cout = cout; j->coutP = coutP; j->ntaps = ntaps; j->kind = winograd; j->block_start = block_start;
    for (int i = 0; i < 16; ++i) j->taps[i] = i < ntaps ? host_tap_off[i] : 0;
    return C2S_OK;
}

// blocks (of 256 threads) a job needs
extern "C" int c2s_pack_job_blocks(int cin, int coutP, int ntaps, int winograd) {
    const long total = winograd == 4 ? (long)4 * cin * coutP : winograd == 3 ? (long)((cin + 1) / 2) * 2 * 4 * coutP
                       : (winograd ? (long)cdiv(cin, WN_CK) * WN_CK * coutP : (long)ntaps * cin * coutP);
    return cdiv(total, 256);
}

extern "C" int c2s_pack_batch(const void* device_table, int njobs, int total_blocks, void* stream) {
    C2S_REQUIRE(device_table && njobs > 0 && total_blocks > 0, "pack_batch: bad args");
    hipLaunchKernelGGL(pack_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PackJob*>(device_table), njobs);
    C2S_CHECK_LAUNCH("pack_batch");
    return C2S_OK;
}

extern "C" int c2s_pack_weights_winograd(const float* src, float* upk, int cin, int cout, int coutP, long stride_o,
                                         long stride_c, const int* host_tap_off, void* stream) {
    C2S_REQUIRE(src && upk && host_tap_off, "pack_weights_winograd: null pointer");
    C2S_REQUIRE(coutP % 64 == 0 && coutP >= cout && cin > 0, "pack_weights_winograd: CoutP must be a multiple of 64");
    TapTable9 tt;
    for (int i = 0; i < 9; ++i) tt.off[i] = host_tap_off[i];
    const long total = (long)cdiv(cin, WN_CK) * WN_CK * coutP;
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
    C2S_REQUIRE(d->Hin % 2 == 0 && d->Win % 4 == 0 && d->Hin >= 2 && d->Win >= 8, "conv3x3_winograd: even H, W a multiple of 4 and >= 8");
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
    const int BC = 1 << l2, BR = 32 >> l2;
    p.tiles_x = cdiv(d->Win, 2 * BC);
    p.tiles = p.tiles_x * cdiv(d->Hin, 2 * BR);
    p.N = d->N;
    p.nchunks = cdiv(d->C0 + d->C1, WN_CK);
    const int plane = (2 * BR + 2) * (2 * BC + 2);
    size_t fl = 2 * ((size_t)WN_CK * plane + WN_USLAB);
    if (fl < (size_t)WN_EXCH) fl = WN_EXCH;
    const int cus = c2s_cus();
    // persistent workgroups: two per CU (register / LDS limit), split over the output-channel blocks
    const int cblocks = d->CoutP / 64;
    const long ntotal = (long)d->N * p.tiles;
    int wpc = 2;
#ifdef C2S_WN_STAMP
    if (const char* e = getenv("C2S_WN_WGS_PER_CU")) wpc = atoi(e);     // diagnostic build: solo-workgroup timing
#endif
    long gx = ((long)wpc * cus + cblocks - 1) / cblocks;
    if (gx > ntotal) gx = ntotal;
    dim3 grid((unsigned)gx, cblocks, 1);
    hipStream_t st = (hipStream_t)stream;
    const size_t ldsb = fl * sizeof(float);
#define C2S_WN_LAUNCH(L2_)                                                                                          \
    if (l2 == L2_) {                                                                                                \
        if (d->reflect_adjoint) hipLaunchKernelGGL((conv_winograd_kernel<L2_, true>), grid, dim3(256), ldsb, st, p); \
        else hipLaunchKernelGGL((conv_winograd_kernel<L2_, false>), grid, dim3(256), ldsb, st, p);                   \
    }
    C2S_WN_LAUNCH(4)
    C2S_WN_LAUNCH(3)
    C2S_WN_LAUNCH(2)
#undef C2S_WN_LAUNCH
    C2S_CHECK_LAUNCH("conv3x3_winograd");
    return C2S_OK;
}
