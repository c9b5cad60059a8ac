// Weight gradient of the convolutions as a split-K implicit GEMM on the exact-f32 MFMA.
//
//   slab[s][tap][cin][cout] = sum_{(n,oy,ox) in slice s} In[n][cin][gather(oy,ox,tap)] * Gout[n][cout][oy][ox]
//
// MFMA view: A[i = cin][k = position], B[k = position][j = cout]  ->  D[cin][cout] per tap.
// Workgroup = 256 threads: one 32-channel cin block x one 64-channel cout block x all taps; wave w owns
// cout fragment (w & 1) and the taps t with t % 2 == (w >> 1)  (<= 8 accumulators of 16 VGPRs).
// The K dimension (N*Hout*Wout positions, up to millions) is split into `nslices` strided sets of
// position tiles; each workgroup keeps its accumulators in registers over its whole slice and writes one
// slab.  c2s_wgrad_reduce sums the slabs in a fixed order (bitwise reproducible, no float atomics) and
// scatters into the torch weight layout.
// LDS: input tile [32][plane | odd stride] and gout tile [64][TP+1]: lanes that differ in channel hit
// different banks because both strides are odd.
//
// Reference call sites replaced: convolution_backward-weight of src/backbones/conv.py:70-80,263-271,
// 378-390.
#include "common.h"
#include <stdlib.h>

namespace {

struct WgradParams {
    const float* src0;
    const float* src1;
    const float* gout;
    float* slabs;
    const int* valid;
    int N, C0, C1, Hin, Win, Cout, Hout, Wout;
    int pad_y, pad_x, pad_mode;
    int nslices, ntiles, tiles_x, tiles_y, log2pc;
    int CinP, CoutB;
};

constexpr int wg_cmax(int a, int b) { return a > b ? a : b; }
template <int K, int S>
struct WCfg {
    static constexpr int TP = (S == 2) ? 64 : 128;     // positions per tile
    static constexpr int NT = K * K;
    static constexpr int TH = (NT + 1) / 2;            // taps per wave half
    static constexpr int plane_for(int l2) { return (((TP >> l2) - 1) * S + K) * (((1 << l2) - 1) * S + K); }
    static constexpr int MAXPLANE = wg_cmax(wg_cmax(plane_for(2), plane_for(3)), wg_cmax(plane_for(4), plane_for(5))) | 1;
    static constexpr int MAXE = (32 * MAXPLANE + 255) / 256;
};

template <int K, int S>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    using C = WCfg<K, S>;
    constexpr int TP = C::TP, NT = C::NT, TH = C::TH, MAXE = C::MAXE;
    extern __shared__ float lds[];

    const int PC = 1 << p.log2pc, PR = TP >> p.log2pc;
    const int rows = (PR - 1) * S + K, cols = (PC - 1) * S + K;
    const int plane = rows * cols;
    const int planeP = plane | 1;
    float* Xl = lds;                   // [32][planeP]
    float* Gl = lds + 32 * planeP;     // [64][TP+1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int ofrag = wave & 1, thalf = wave >> 1;
    const int cb = blockIdx.y * 32, ob = blockIdx.z * 64;
    const int slice = blockIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HWin = p.Hin * p.Win, HWo = p.Hout * p.Wout;

    // element decomposition of the staging loop (fixed for the whole kernel)
    int epk[MAXE];
    const int total = 32 * plane;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = tid + i * 256;
        int pk = -1;
        if (e < total) {
            const int c = e / plane;
            const int rem = e - c * plane;
            const int r = rem / cols;
            pk = (c << 20) | (r << 10) | (rem - r * cols);
        }
        epk[i] = pk;
    }

    f32x16 acc[TH];
#pragma unroll
    for (int i = 0; i < TH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int tile = slice; tile < p.ntiles; tile += p.nslices) {
        const int n = tile / (p.tiles_x * p.tiles_y);
        const int trem = tile - n * (p.tiles_x * p.tiles_y);
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        if (p.valid != nullptr && p.valid[n] == 0) continue;
        const int oy0 = tyi * PR, ox0 = txi * PC;
        // ---- stage input tile
        const float* s0n = p.src0 + (size_t)n * p.C0 * HWin;
        const float* s1n = p.src1 != nullptr ? p.src1 + (size_t)n * p.C1 * HWin : nullptr;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int pk = epk[i];
            if (pk >= 0) {
                const int c = pk >> 20, r = (pk >> 10) & 1023, cc = pk & 1023;
                int gy = oy0 * S - p.pad_y + r, gx = ox0 * S - p.pad_x + cc;
                bool ok;
                if (p.pad_mode == C2S_PAD_REFLECT) {
                    ok = gy >= -p.pad_y && gy < p.Hin + p.pad_y && gx >= -p.pad_x && gx < p.Win + p.pad_x;
                    gy = reflect_idx(gy, p.Hin);
                    gx = reflect_idx(gx, p.Win);
                } else {
                    ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
                }
                const int cg = cb + c;
                float v = 0.f;
                if (ok && cg < Cin) {
                    const float* s = cg < p.C0 ? s0n + (size_t)cg * HWin : s1n + (size_t)(cg - p.C0) * HWin;
                    v = s[gy * p.Win + gx];
                }
                Xl[c * planeP + r * cols + cc] = v;
            }
        }
        // ---- stage gout tile [64][TP]
        {
            const float* gn = p.gout + (size_t)n * p.Cout * HWo;
            for (int e = tid; e < 64 * TP; e += 256) {
                const int o = e / TP, q = e % TP;
                const int oy = oy0 + (q >> p.log2pc), ox = ox0 + (q & (PC - 1));
                float v = 0.f;
                if (ob + o < p.Cout && oy < p.Hout && ox < p.Wout) v = gn[(size_t)(ob + o) * HWo + oy * p.Wout + ox];
                Gl[o * (TP + 1) + q] = v;
            }
        }
        __syncthreads();
        // ---- MFMA over position pairs
        for (int kk = 0; kk < TP / 2; ++kk) {
            const int q = 2 * kk + lk;
            const int qy = q >> p.log2pc, qx = q & (PC - 1);
            const float b = Gl[(ofrag * 32 + li) * (TP + 1) + q];
            const int abase = li * planeP + (qy * S) * cols + qx * S;
#pragma unroll
            for (int i = 0; i < TH; ++i) {
                const int t = 2 * i + thalf;
                if (t < NT) {
                    const float a = Xl[abase + (t / K) * cols + (t % K)];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- write slab [slice][tap][CinP][CoutB]
#pragma unroll
    for (int i = 0; i < TH; ++i) {
        const int t = 2 * i + thalf;
        if (t < NT) {
            float* sl = p.slabs + (((size_t)slice * NT + t) * p.CinP + cb) * p.CoutB + ob + ofrag * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * lk;
                sl[(size_t)ci * p.CoutB] = acc[i][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Tiled path (planes whose width is 16 or a multiple of 32: every layer of the default models at 16..128 px).
// The tile is TW = 32 or 16 positions wide so that every row segment of the input tile (TW*S floats + 1 halo column
// each side) and of the gout tile (TW floats) is 16-byte aligned: global traffic is float4, and the loads of
// tile i+1 are issued into registers BEFORE the MFMA loop of tile i (register-staged software pipeline,
// cdna_hip_programming.md T14): HBM/L2 latency hides under ~20k cycles of MFMA.
// Work split over the 4 waves (wave = (ofrag, half)):
//   MODE 0 (4x4):  half = tap parity      -> 8 accumulators per wave, every wave walks all positions
//   MODE 1 (3x3, 1x1): half = position half of the tile -> NT accumulators per wave (every wave the same number of
//                  MFMAs per k-step; the tap-parity split of 9 taps is 5:4 and left one SIMD pair 20 % idle);
//                  the two halves are added through LDS once, after the last tile
//   MODE 2 (3x3, Cin <= 10, the first layer): as MODE 1 but the MFMA rows are (tap, cin) pairs: 9*Cin <= 96 rows
//                  = 3 fragments instead of 9 fragments of 32 mostly-empty channel rows.
template <int K, int S>
struct TCfg {
    // positions per tile: 64 keeps (accumulators + the register-staged next tile) within 256 registers = 2 waves/SIMD
    static constexpr int TP = (K == 1) ? 128 : 64;
};

template <int K, int S, int LW, int MODE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_tile_kernel(WgradParams p) {
    constexpr int TP = TCfg<K, S>::TP;
    constexpr int TW = 1 << LW;
    constexpr int PR = TP / TW;                    // tile rows
    constexpr int NT = K * K;
    constexpr int CB = (MODE == 2) ? 16 : 32;      // input channels staged per workgroup
    constexpr int NA = (MODE == 0) ? (NT + 1) / 2 : (MODE == 1 ? NT : 3);   // accumulators per wave
    constexpr int PAD = (K == 1) ? 0 : 1;
    constexpr int XR = (PR - 1) * S + K;           // input tile rows
    constexpr int XC = (TW - 1) * S + K;           // input tile cols
    constexpr int HALO_R = XC - TW * S - PAD;      // right halo columns (== 1 for K=3,S=1 and K=4,S=2; 0 for K=1)
    constexpr int PLANE = XR * XC, PLANEP = PLANE | 1;
    constexpr int V4 = TW * S / 4;                 // float4 per interior row
    constexpr int GV4 = TW / 4;                    // float4 per gout row
    constexpr int NXI = CB * XR * V4;              // interior float4 items
    constexpr int NXH = (PAD + HALO_R) * CB * XR;  // halo scalars
    constexpr int NG = 64 * PR * GV4;              // gout float4 items
    constexpr int XI_PT = (NXI + 255) / 256, XH_PT = (NXH + 255) / 256, G_PT = NG / 256;
    constexpr int KS = (MODE == 0) ? TP / 2 : TP / 4;   // k-steps (position pairs) per wave per tile
    static_assert(HALO_R == PAD, "halo is symmetric for the supported kernels");
    static_assert(NG % 256 == 0 && KS % 2 == 0, "tile shape");
    extern __shared__ float lds[];
    float* Xl = lds;                    // [CB][PLANEP]
    float* Gl = lds + CB * PLANEP;      // [64][TP+1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int ofrag = wave & 1, half = wave >> 1;
    const int cb = blockIdx.y * 32, ob = blockIdx.z * 64;
    const int slice = blockIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HWin = p.Hin * p.Win, HWo = p.Hout * p.Wout;
    const bool reflect = p.pad_mode == C2S_PAD_REFLECT;

    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    f32x4 xi[XI_PT];
    float xh[XH_PT > 0 ? XH_PT : 1];
    f32x4 gv[G_PT];

    // issue the global loads of one tile into registers.  Item order (float4 j, channel, row): the LDS commits of
    // 32 neighbouring lanes then fall into distinct banks (channel stride PLANEP / TP+1 is odd)
    auto prefetch = [&](int tile) {
        const int n = tile / (p.tiles_x * p.tiles_y);
        const int trem = tile - n * (p.tiles_x * p.tiles_y);
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int oy0 = tyi * PR, ox0 = txi * TW;
        const float* s0n = p.src0 + (size_t)n * p.C0 * HWin;
        const float* s1n = p.src1 != nullptr ? p.src1 + (size_t)n * p.C1 * HWin : nullptr;
#pragma unroll
        for (int i = 0; i < XI_PT; ++i) {
            const int e = tid + i * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < NXI) {
                const int j = e % V4, rr = e / V4;
                const int c = rr % CB, r = rr / CB;
                int gy = oy0 * S - PAD + r;
                bool ok = true;
                if (reflect) gy = reflect_idx(gy, p.Hin); else ok = gy >= 0 && gy < p.Hin;
                const int cg = cb + c;
                if (ok && cg < Cin) {
                    const float* sp = cg < p.C0 ? s0n + (size_t)cg * HWin : s1n + (size_t)(cg - p.C0) * HWin;
                    v = *reinterpret_cast<const f32x4*>(sp + (size_t)gy * p.Win + ox0 * S + 4 * j);
                }
            }
            xi[i] = v;
        }
#pragma unroll
        for (int i = 0; i < XH_PT; ++i) {
            const int e = tid + i * 256;
            float v = 0.f;
            if (e < NXH) {
                const int side = e & 1, rr = e >> 1;          // PAD + HALO_R == 2
                const int c = rr % CB, r = rr / CB;
                int gy = oy0 * S - PAD + r;
                int gx = side == 0 ? ox0 * S - 1 : ox0 * S + TW * S;
                bool ok = true;
                if (reflect) { gy = reflect_idx(gy, p.Hin); gx = reflect_idx(gx, p.Win); }
                else ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
                const int cg = cb + c;
                if (ok && cg < Cin) {
                    const float* sp = cg < p.C0 ? s0n + (size_t)cg * HWin : s1n + (size_t)(cg - p.C0) * HWin;
                    v = sp[(size_t)gy * p.Win + gx];
                }
            }
            xh[i] = v;
        }
        const float* gn = p.gout + (size_t)n * p.Cout * HWo;
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            const int e = tid + i * 256;
            const int j = e % GV4, rr = e / GV4;
            const int o = rr & 63, r = rr >> 6;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ob + o < p.Cout)
                v = *reinterpret_cast<const f32x4*>(gn + (size_t)(ob + o) * HWo + (size_t)(oy0 + r) * p.Wout + ox0 + 4 * j);
            gv[i] = v;
        }
    };
    // registers -> LDS (odd plane strides: channel-varying lanes of the MFMA reads hit distinct banks)
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < XI_PT; ++i) {
            const int e = tid + i * 256;
            if (e < NXI) {
                const int j = e % V4, rr = e / V4;
                const int c = rr % CB, r = rr / CB;
                float* d = Xl + c * PLANEP + r * XC + PAD + 4 * j;
                d[0] = xi[i].x; d[1] = xi[i].y; d[2] = xi[i].z; d[3] = xi[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < XH_PT; ++i) {
            const int e = tid + i * 256;
            if (e < NXH) {
                const int side = e & 1, rr = e >> 1;
                const int c = rr % CB, r = rr / CB;
                Xl[c * PLANEP + r * XC + (side == 0 ? 0 : XC - 1)] = xh[i];
            }
        }
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            const int e = tid + i * 256;
            const int j = e % GV4, rr = e / GV4;
            const int o = rr & 63, r = rr >> 6;
            float* d = Gl + o * (TP + 1) + r * TW + 4 * j;
            d[0] = gv[i].x; d[1] = gv[i].y; d[2] = gv[i].z; d[3] = gv[i].w;
        }
    };

    // LDS offset of the A operand of accumulator i for this lane: channel row + tap displacement.  MODE 0/1: the tap
    // displacement is a compile-time constant folded into the ds_read offset field (one base register).
    int aoff[MODE == 2 ? NA : 1];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = i * 32 + li;                 // (tap, cin) pair; rows past NT*Cin are never written out
            const int t = row < NT * Cin ? row / Cin : 0, c = row < NT * Cin ? row - t * Cin : 0;
            aoff[i] = c * PLANEP + (t / K) * XC + (t % K);
        }
    } else {
        aoff[0] = li * PLANEP + (MODE == 0 ? (half / K) * XC + (half % K) : 0);
    }
    const int kk0 = (MODE == 0) ? 0 : half * KS;

    // next valid tile of this slice (padded frames are skipped)
    auto next_tile = [&](int tile) {
        while (tile < p.ntiles) {
            const int n = tile / (p.tiles_x * p.tiles_y);
            if (p.valid == nullptr || p.valid[n] != 0) break;
            tile += p.nslices;
        }
        return tile;
    };

    // XCD-aware start (see conv_winograd.hip): each XCD walks a contiguous eighth of the tiles in flight, so vertically
    // adjacent tiles (shared halo rows) meet in one L2; the slab index stays the workgroup index
    int tile = next_tile((p.nslices & 7) == 0 ? (slice & 7) * (p.nslices >> 3) + (slice >> 3) : slice);
    if (tile < p.ntiles) prefetch(tile);
    while (tile < p.ntiles) {
        // the next tile's frame flag is fetched under the LDS commit and the barrier (read right after the barrier it stalled
        // every wave of the workgroup for a memory round trip per tile)
        const int cand = tile + p.nslices;
        const int cflag = (p.valid != nullptr && cand < p.ntiles) ? p.valid[cand / (p.tiles_x * p.tiles_y)] : 1;
        commit();
        __syncthreads();
        const int nxt = cflag != 0 ? cand : next_tile(cand + p.nslices);
        if (nxt < p.ntiles) prefetch(nxt);          // in flight during the MFMA loop below
        // operand reads software-pipelined one position-pair ahead of the MFMAs (order pinned with sched barriers)
        auto load_ops = [&](int kk, float (&a)[NA], float& b) {
            const int q = 2 * (kk0 + kk) + lk;
            const int qy = q >> LW, qx = q & (TW - 1);
            b = Gl[(ofrag * 32 + li) * (TP + 1) + q];
            const int pos = (qy * S) * XC + qx * S;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if constexpr (MODE == 2) {
                    a[i] = Xl[aoff[i] + pos];
                } else if constexpr (MODE == 1) {
                    a[i] = Xl[aoff[0] + pos + (i / K) * XC + (i % K)];
                } else {
                    // tap 2i + half = tap `half` displaced by 2i taps; K is even, so the row/column split is static
                    a[i] = Xl[aoff[0] + pos + ((2 * i) / K) * XC + ((2 * i) % K)];
                }
            }
        };
        auto mfmas = [&](const float (&a)[NA], float b) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (MODE != 0 || 2 * i + half < NT) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b, acc[i], 0, 0, 0);
        };
        float a0[NA], a1[NA], b0, b1;
        load_ops(0, a0, b0);
#pragma unroll 2
        for (int kk = 0; kk < KS; kk += 2) {
            load_ops(kk + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_ops(kk + 2 < KS ? kk + 2 : kk, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        tile = nxt;
    }

    // ---- slab [slice][tap][CinP][CoutB]; position halves are combined through LDS first (fixed order: half 0 + half 1)
    float* xch = lds;                                  // [ofrag][16][64] floats, reuses the tile area
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        if constexpr (MODE != 0) {
            __syncthreads();
            if (half == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(ofrag * 16 + r) * 64 + lane] = acc[i][r];
            }
            __syncthreads();
            if (half == 1) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] += xch[(ofrag * 16 + r) * 64 + lane];
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < NT * Cin) {
                    const int t = row / Cin, c = row - t * Cin;
                    p.slabs[(((size_t)slice * NT + t) * p.CinP + c) * p.CoutB + ob + ofrag * 32 + li] = acc[i][r];
                }
            }
        } else {
            const int t = (MODE == 0) ? 2 * i + half : i;
            if (t < NT) {
                float* sl = p.slabs + (((size_t)slice * NT + t) * p.CinP + cb) * p.CoutB + ob + ofrag * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = (r & 3) + 8 * (r >> 2) + 4 * lk;
                    sl[(size_t)ci * p.CoutB] = acc[i][r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 pad-1 weight gradient as Winograd F(2x2,3x3):  d U[xi][nu][cin][cout] = sum_{2x2 blocks} V * Mt  with
// V = Bt d B (4x4 input patch) and Mt = A gy At (2x2 output-gradient block -> 4x4), then d W = Gt (d U) G in the
// reduce kernel: 16 multiplies per (block, channel pair) instead of 36.
// MFMA: D[cin][cout] per transform point, A operand = V (rows = cin, k = block), B operand = Mt (cols = cout).
// Workgroup = 4 waves, wave xi owns row xi of the transform domain: 4 nu x 2 cout fragments = 8 accumulators for a
// (32 cin x 64 cout) block.  A spatial tile is 2 x 16 blocks (4 x 32 output pixels): 16 k-steps of 2 blocks.  Both
// operands are transformed on the fly by the lane that feeds them (its channel's patch rows / gradient block come as
// ds_read_b64 from channel-major LDS tiles whose pitch is 2*odd mod 64 banks: conflict-free).  The next tile is
// prefetched into registers under the MFMA loop (single LDS buffer, two barriers per tile), as in the direct kernel.
#ifdef C2S_WW_STAMP
// diagnostic build only (tools/wgrad_stamps.py): per workgroup, wave 0's cycles in commit + barrier, in the MFMA loop, at the
// barrier behind the loop, and in all
__device__ unsigned long long ww_stamps[1024 * 4];
#endif
constexpr int WW_XP = 206;     // raw input tile pitch per channel: 6 x 34 = 204 -> 206 (= 2*7 mod 64)
constexpr int WW_GP = 130;     // gout tile pitch per channel:      4 x 32 = 128 -> 130 (= 2*1 mod 64)

typedef float f32x2w __attribute__((ext_vector_type(2)));

// CB = input-channel blocks of 32 per workgroup.  CB = 2 (layers with a multiple of 64 input channels): one 8-wave workgroup
// per CU instead of two 4-wave ones, waves 0-3 / 4-7 on the two blocks, ONE gradient tile in LDS for both -- every operand is
// read once (with CB = 1 the gradient tile is fetched once per input block: 1.58 GB instead of 1.07 GB on 64 -> 64 @128x128).
template <int CB>
__global__ __launch_bounds__(256 * CB, CB == 1 ? 2 : 1) void conv_wgrad_winograd_kernel(WgradParams p) {
    constexpr int XR = 6, XC = 34;
    constexpr int NTH = 256 * CB, CW = 32 * CB;    // threads, input channels of the workgroup
    constexpr int NXI = CW * XR * 8;               // interior float4 items of the raw tile
    constexpr int NXH = 2 * CW * XR;               // halo scalars
    constexpr int NG = 64 * 4 * 8;                 // gout float4 items
    constexpr int XI_PT = NXI / NTH, XH_PT = (NXH + NTH - 1) / NTH, G_PT = NG / NTH;
    extern __shared__ float lds[];
    float* Xl = lds;                               // [CW cin][WW_XP]
    float* Gl = lds + CW * WW_XP;                  // [64 cout][WW_GP]

    const int tid = threadIdx.x, lane = tid & 63, xi = (tid >> 6) & 3, cbh = tid >> 8;
    const int li = lane & 31, lk = lane >> 5;
    const int cb0 = blockIdx.y * CW, cb = cb0 + cbh * 32, ob = blockIdx.z * 64;
    const int slice = blockIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HW = p.Hin * p.Win;
    const bool reflect = p.pad_mode == C2S_PAD_REFLECT;
    // rows of Bt (input) and of A (output gradient) combined by this wave
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sb_ = xi == 1 ? 1.f : -1.f;                        // t = d[ra] + sb * d[rb]
    const float ga = xi == 3 ? 0.f : 1.f, gb = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);   // r = ga * gy[0] + gb * gy[1]

    f32x2w sb2 = {sb_, sb_}, ga2 = {ga, ga}, gb2 = {gb, gb};
    asm volatile("" : "+v"(sb2), "+v"(ga2), "+v"(gb2));       // (in vector registers: packed operands of v_pk_fma_f32 / v_pk_mul_f32)

    f32x16 acc[4][2];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][m][r] = 0.f;

    f32x4 xi4[XI_PT];
    float xh[XH_PT];
    f32x4 gv[G_PT];
    // Tile-invariant byte offsets of this thread's float4 items (input rows, gradient rows) inside a frame: on tiles that
    // touch neither the top nor the bottom of the plane the requests are buffer loads at (offset, scalar tile offset) with no
    // per-tile address arithmetic in the vector ALU (it was 287 VALU instructions per tile next to 128 MFMAs; a VALU
    // instruction costs ~8 cycles of MFMA issue).  Channels past the end of a source are out-of-range offsets (read 0).
    int xoff[XI_PT], gof[G_PT];
    unsigned xs1 = 0;                                   // bit i: item i comes from the second source (uniform per wave)
    const bool fast_ok = CB == 2 && (p.C1 == 0 || p.C0 % 8 == 0) &&      // (CB == 1 has no registers to spare for the offsets)
                         (long)(p.C0 > p.C1 ? p.C0 : p.C1) * HW * 4 < 0x7FFF0000L &&
                         (long)p.Cout * HW * 4 < 0x7FFF0000L;
#pragma unroll
    for (int i = 0; i < XI_PT; ++i) {
        const int e = tid + i * NTH;
        const int j = e & 7, rr = e >> 3;
        const int c = rr % CW, r = rr / CW;
        const int cg = cb0 + c;
        const bool second = cg >= p.C0;
        xoff[i] = (((second ? cg - p.C0 : cg) * HW) + r * p.Win + 4 * j) * 4;
        xs1 |= (unsigned)__builtin_amdgcn_readfirstlane(second ? 1 : 0) << i;
    }
#pragma unroll
    for (int i = 0; i < G_PT; ++i) {
        const int e = tid + i * NTH;
        const int j = e & 7, rr = e >> 3;
        const int o = rr & 63, r = rr >> 6;
        gof[i] = (o * HW + r * p.Win + 4 * j) * 4;
    }
    typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
    auto prefetch = [&](int tile) {
        const int n = tile / (p.tiles_x * p.tiles_y);
        const int trem = tile - n * (p.tiles_x * p.tiles_y);
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int oy0 = tyi * 4, ox0 = txi * 32;
        const float* s0n = p.src0 + (size_t)n * p.C0 * HW;
        const float* s1n = p.src1 != nullptr ? p.src1 + (size_t)n * p.C1 * HW : nullptr;
        const float* gn = p.gout + (size_t)n * p.Cout * HW;
        const bool fast = fast_ok && oy0 > 0 && oy0 + 4 < p.Hin;
        if (fast) {
            const __amdgpu_buffer_rsrc_t q0 = __builtin_amdgcn_make_buffer_rsrc((void*)s0n, 0, p.C0 * HW * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t q1 =
                __builtin_amdgcn_make_buffer_rsrc((void*)(s1n != nullptr ? s1n : s0n), 0, p.C1 * HW * 4, 0x00020000);
            const int so = ((oy0 - 1) * p.Win + ox0) * 4;
#pragma unroll
            for (int i = 0; i < XI_PT; ++i) {
                const u32x4w v = ((xs1 >> i) & 1u) ? __builtin_amdgcn_raw_buffer_load_b128(q1, xoff[i], so, 0)
                                                   : __builtin_amdgcn_raw_buffer_load_b128(q0, xoff[i], so, 0);
                xi4[i] = __builtin_bit_cast(f32x4, v);
            }
            const __amdgpu_buffer_rsrc_t qg =
                __builtin_amdgcn_make_buffer_rsrc((void*)(gn + (size_t)ob * HW), 0, (p.Cout - ob) * HW * 4, 0x00020000);
            const int sg = (oy0 * p.Win + ox0) * 4;
#pragma unroll
            for (int i = 0; i < G_PT; ++i) gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qg, gof[i], sg, 0));
        } else {
#pragma unroll
            for (int i = 0; i < XI_PT; ++i) {
                const int e = tid + i * NTH;
                const int j = e & 7, rr = e >> 3;
                const int c = rr % CW, r = rr / CW;
                int gy = oy0 - 1 + r;
                bool ok = true;
                if (reflect) gy = reflect_idx(gy, p.Hin); else ok = gy >= 0 && gy < p.Hin;
                const int cg = cb0 + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok && cg < Cin) {
                    const float* sp = cg < p.C0 ? s0n + (size_t)cg * HW : s1n + (size_t)(cg - p.C0) * HW;
                    v = *reinterpret_cast<const f32x4*>(sp + (size_t)gy * p.Win + ox0 + 4 * j);
                }
                xi4[i] = v;
            }
#pragma unroll
            for (int i = 0; i < G_PT; ++i) {
                const int e = tid + i * NTH;
                const int j = e & 7, rr = e >> 3;
                const int o = rr & 63, r = rr >> 6;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ob + o < p.Cout) v = *reinterpret_cast<const f32x4*>(gn + (size_t)(ob + o) * HW + (size_t)(oy0 + r) * p.Win + ox0 + 4 * j);
                gv[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < XH_PT; ++i) {
            const int e = tid + i * NTH;
            float v = 0.f;
            if (e < NXH) {
                const int side = e & 1, rr = e >> 1;
                const int c = rr % CW, r = rr / CW;
                int gy = oy0 - 1 + r;
                int gx = side == 0 ? ox0 - 1 : ox0 + 32;
                bool ok = true;
                if (reflect) { gy = reflect_idx(gy, p.Hin); gx = reflect_idx(gx, p.Win); }
                else ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
                const int cg = cb0 + c;
                if (ok && cg < Cin) {
                    const float* sp = cg < p.C0 ? s0n + (size_t)cg * HW : s1n + (size_t)(cg - p.C0) * HW;
                    v = sp[(size_t)gy * p.Win + gx];
                }
            }
            xh[i] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < XI_PT; ++i) {
            const int e = tid + i * NTH;
            const int j = e & 7, rr = e >> 3;
            const int c = rr % CW, r = rr / CW;
            float* d = Xl + c * WW_XP + r * XC + 1 + 4 * j;
            d[0] = xi4[i].x; d[1] = xi4[i].y; d[2] = xi4[i].z; d[3] = xi4[i].w;
        }
#pragma unroll
        for (int i = 0; i < XH_PT; ++i) {
            const int e = tid + i * NTH;
            if (e < NXH) {
                const int side = e & 1, rr = e >> 1;
                const int c = rr % CW, r = rr / CW;
                Xl[c * WW_XP + r * XC + (side == 0 ? 0 : XC - 1)] = xh[i];
            }
        }
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            const int e = tid + i * NTH;
            const int j = e & 7, rr = e >> 3;
            const int o = rr & 63, r = rr >> 6;
            // the pitch keeps rows 8-byte (not 16-byte) aligned: two ds_write_b64
            f32x2w* gd = reinterpret_cast<f32x2w*>(Gl + o * WW_GP + r * 32 + 4 * j);
            gd[0] = f32x2w{gv[i].x, gv[i].y};
            gd[1] = f32x2w{gv[i].z, gv[i].w};
        }
    };
    auto next_tile = [&](int tile) {
        while (tile < p.ntiles) {
            const int n = tile / (p.tiles_x * p.tiles_y);
            if (p.valid == nullptr || p.valid[n] != 0) break;
            tile += p.nslices;
        }
        return tile;
    };

    // operands of k-step kk: blocks t = 2*kk + lk of the 2 x 16 block tile
    auto load_ops = [&](int kk, f32x2w (&d)[4], f32x2w (&g)[2][2]) {
        const int t = 2 * kk + lk;
        const int by = t >> 4, bx = t & 15;
        const float* pa = Xl + (cbh * 32 + li) * WW_XP + (2 * by + ra) * XC + 2 * bx;
        const float* pb = Xl + (cbh * 32 + li) * WW_XP + (2 * by + rb) * XC + 2 * bx;
        d[0] = *reinterpret_cast<const f32x2w*>(pa);
        d[1] = *reinterpret_cast<const f32x2w*>(pa + 2);
        d[2] = *reinterpret_cast<const f32x2w*>(pb);
        d[3] = *reinterpret_cast<const f32x2w*>(pb + 2);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float* pg = Gl + (m * 32 + li) * WW_GP + (2 * by) * 32 + 2 * bx;
            g[m][0] = *reinterpret_cast<const f32x2w*>(pg);
            g[m][1] = *reinterpret_cast<const f32x2w*>(pg + 32);
        }
    };

    // XCD-aware start (see conv_winograd.hip): each XCD walks a contiguous eighth of the tiles in flight, so vertically
    // adjacent tiles (shared halo rows) meet in one L2; the slab index stays the workgroup index
    int tile = next_tile((p.nslices & 7) == 0 ? (slice & 7) * (p.nslices >> 3) + (slice >> 3) : slice);
    if (tile < p.ntiles) prefetch(tile);
#ifdef C2S_WW_STAMP
    unsigned long long st_commit = 0, st_loop = 0, st_tail = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    while (tile < p.ntiles) {
        // the next tile's frame flag is fetched under the LDS commit and the barrier (read right after the barrier it stalled
        // every wave of the workgroup for a memory round trip per tile)
        const int cand = tile + p.nslices;
        const int cflag = (p.valid != nullptr && cand < p.ntiles) ? p.valid[cand / (p.tiles_x * p.tiles_y)] : 1;
#ifdef C2S_WW_STAMP
        const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
        commit();
        __syncthreads();
#ifdef C2S_WW_STAMP
        const unsigned long long st_b = __builtin_amdgcn_s_memtime();
        st_commit += st_b - st_a;
#endif
        const int nxt = cflag != 0 ? cand : next_tile(cand + p.nslices);
        if (nxt < p.ntiles) prefetch(nxt);          // in flight during the MFMA loop below
        f32x2w d[2][4], g[2][2][2];
        load_ops(0, d[0], g[0]);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const f32x2w(&dd)[4] = d[kk & 1];
            const f32x2w(&gg)[2][2] = g[kk & 1];
            // input transform (row pair of Bt, then the four columns) and output-gradient transform (row of A, then the four
            // columns of At) on register pairs: 10 packed VALU instructions per k-step instead of 22 scalar ones (each VALU
            // instruction costs the SIMD ~8 cycles of MFMA issue; tools/_diag/mfma_rate.hip).  The last column pair carries a
            // flipped sign on both operands (V3' = t3 - t1, Mt3' = r1: the product is unchanged, bit for bit).  The asm holds
            // the three operand-select forms the compiler does not emit; its s_nop covers the VALU -> MFMA wait states.
            f32x2w V03, V12, M0, M1, R0, R1, T01, T23;
            asm("v_pk_fma_f32 %6, %16, %10, %8\n\t"                                            // t01 = d[ra] + sb * d[rb]
                "v_pk_fma_f32 %7, %16, %11, %9\n\t"                                            // t23
                "v_pk_mul_f32 %4, %17, %12\n\t"
                "v_pk_mul_f32 %5, %17, %14\n\t"
                "v_pk_fma_f32 %4, %18, %13, %4\n\t"                                            // r = ga * gy[0] + gb * gy[1]
                "v_pk_fma_f32 %5, %18, %15, %5\n\t"
                "v_pk_add_f32 %0, %6, %7 neg_lo:[0,1] neg_hi:[1,0]\n\t"                        // (t0 - t2, t3 - t1)
                "v_pk_add_f32 %1, %6, %7 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"       // (t1 + t2, t2 - t1)
                "v_pk_add_f32 %2, %4, %4 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"       // (r0 + r1, r0 - r1)
                "v_pk_add_f32 %3, %5, %5 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
                "s_nop 1"
                : "=&v"(V03), "=&v"(V12), "=&v"(M0), "=&v"(M1), "=&v"(R0), "=&v"(R1), "=&v"(T01), "=&v"(T23)
                : "v"(dd[0]), "v"(dd[1]), "v"(dd[2]), "v"(dd[3]), "v"(gg[0][0]), "v"(gg[0][1]), "v"(gg[1][0]), "v"(gg[1][1]),
                  "v"(sb2), "v"(ga2), "v"(gb2));
            const float V[4] = {V03.x, V12.x, V12.y, V03.y};
            const float Mt[2][4] = {{R0.x, M0.x, M0.y, R0.y}, {R1.x, M1.x, M1.y, R1.y}};
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[0], Mt[0][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < 16) load_ops(kk + 1, d[(kk + 1) & 1], g[(kk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (v + m > 0) acc[v][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[v], Mt[m][v], acc[v][m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef C2S_WW_STAMP
        const unsigned long long st_c = __builtin_amdgcn_s_memtime();
        st_loop += st_c - st_b;
#endif
        __syncthreads();
#ifdef C2S_WW_STAMP
        st_tail += __builtin_amdgcn_s_memtime() - st_c;
#endif
        tile = nxt;
    }
#ifdef C2S_WW_STAMP
    if (tid == 0 && blockIdx.x < 1024 && blockIdx.y == 0 && blockIdx.z == 0) {
        ww_stamps[blockIdx.x * 4 + 0] = st_commit;
        ww_stamps[blockIdx.x * 4 + 1] = st_loop;
        ww_stamps[blockIdx.x * 4 + 2] = st_tail;
        ww_stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - st_begin;
    }
#endif

    // ---- dW = Gt dU G inside the workgroup (16 -> 9 values per (cin, cout): 44 % less slab traffic and no separate
    // transform launch): the nu direction is lane-local, the xi direction goes through LDS, one kx column per round.
    // Slab layout [slice][ky*3+kx][CinP][CoutB] = the generic one, summed over slices by wgrad_reduce_kernel.
    float* ex = lds + cbh * (4 * 2 * 16 * 64);         // [xi 4][m 2][r 16][lane 64] per input block; the tiles are done (barrier above)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float u1 = acc[1][m][r], u2 = acc[2][m][r];
                const float t = kx == 0 ? acc[0][m][r] + 0.5f * (u1 + u2)
                              : (kx == 1 ? 0.5f * (u1 - u2) : 0.5f * (u1 + u2) + acc[3][m][r]);
                ex[((xi * 2 + m) * 16 + r) * 64 + lane] = t;
            }
        __syncthreads();
        if (xi < 3) {                                  // wave xi produces filter row ky = xi
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float* sl = p.slabs + (((size_t)slice * 9 + xi * 3 + kx) * p.CinP + cb) * p.CoutB + ob + m * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t0 = ex[((0 * 2 + m) * 16 + r) * 64 + lane], t1 = ex[((1 * 2 + m) * 16 + r) * 64 + lane];
                    const float t2 = ex[((2 * 2 + m) * 16 + r) * 64 + lane], t3 = ex[((3 * 2 + m) * 16 + r) * 64 + lane];
                    const float w = xi == 0 ? t0 + 0.5f * (t1 + t2) : (xi == 1 ? 0.5f * (t1 - t2) : 0.5f * (t1 + t2) + t3);
                    const int ci = (r & 3) + 8 * (r >> 2) + 4 * lk;
                    sl[(size_t)ci * p.CoutB] = w;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// 4x4 stride-2 pad-1 weight gradient (DownConvBlock, conv.py:263-271) as Winograd F(2x2,2x2) over the four input parities.
// Over the parities of its input a 4-tap stride-2 filter is two 2-tap stride-1 filters (conv_s2wino.hip):
//   y[i] = x1[i-1] w0 + x0[i] w1 + x1[i] w2 + x0[i+1] w3,   x0[i] = x[2i], x1[i] = x[2i+1]
// so per parity pair (py, px) the layer is a 2x2-tap correlation of a parity plane with g = (w[1-py + 2a][1-px + 2b])_{a,b}, and
//   d U_pp = sum_{2x2 output blocks} (Bt D B) (.) (A dY At),   d g_pp = Gt (d U_pp) G
// with D the 3x3 patch of the parity plane, dY the 2x2 block of the output gradient and the INTEGER matrices
// Bt = [[1,-1,0],[0,1,0],[0,-1,1]], G = A = [[1,0],[1,1],[0,1]]: 9 multiplies per (block, channel pair, parity pair) = 36 instead
// of the direct form's 64, as exact as the direct form.
// MFMA (32x32x2): D[cin][cout] per transform point, A operand = V (rows = cin, k = block), B operand = Mt (cols = cout).
// Workgroup = 8 waves on a (32 cin x 64 cout) block: wave (pp, m) owns the 9 points of parity pair pp for cout fragment m
// (9 accumulators = 144 registers), so d g = Gt dU G is formed in registers and the slabs are the 16-tap slabs of the direct
// kernel (same reduce kernel).  A spatial tile is 2 x 16 blocks (4 x 32 output pixels, 10 x 66 input pixels, stored as four
// parity planes of 5 x 33 per channel); both operands are transformed on the fly by the lane that feeds them; the next tile is
// prefetched into registers under the MFMA loop.
constexpr int S2W_PL = 5 * 34;                      // one parity plane [5][33 -> 34]
constexpr int S2W_XP = 4 * S2W_PL + 2;              // input pitch per channel: 682 = 2 * 341 (odd): conflict-free ds_read_b64 across channels

__global__ __launch_bounds__(512) void conv_wgrad_s2wino_kernel(WgradParams p) {
    constexpr int NTH = 512;
    constexpr int NXI = 32 * 10 * 16;              // interior float4 items of the raw tile: [10 rows][32 cin][16]
    constexpr int NXH = 2 * 32 * 10;               // halo scalars
    constexpr int NG = 64 * 4 * 8;                 // gout float4 items
    constexpr int XI_PT = NXI / NTH, XH_PT = (NXH + NTH - 1) / NTH, G_PT = NG / NTH;
    extern __shared__ float lds[];
    float* Xl = lds;                               // [32 cin][S2W_XP]
    float* Gl = lds + 32 * S2W_XP;                 // [64 cout][WW_GP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pp = w >> 1, m = w & 1, py = pp >> 1, px = pp & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int cb = blockIdx.y * 32, ob = blockIdx.z * 64;
    const int slice = blockIdx.x;
    const int Cin = p.C0 + p.C1;
    const int HWi = p.Hin * p.Win, HWo = p.Hout * p.Wout;
    const bool reflect = p.pad_mode == C2S_PAD_REFLECT;

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    f32x4 xi4[XI_PT];
    float xh[XH_PT];
    f32x4 gv[G_PT];
    auto prefetch = [&](int tile) {
        const int n = tile / (p.tiles_x * p.tiles_y);
        const int trem = tile - n * (p.tiles_x * p.tiles_y);
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int oy0 = tyi * 4, ox0 = txi * 32;
        const float* s0n = p.src0 + (size_t)n * p.C0 * HWi;
        const float* s1n = p.src1 != nullptr ? p.src1 + (size_t)n * p.C1 * HWi : nullptr;
        const float* gn = p.gout + (size_t)n * p.Cout * HWo;
#pragma unroll
        for (int i = 0; i < XI_PT; ++i) {
            const int e = tid + i * NTH;
            const int j = e & 15, rr = e >> 4;
            const int c = rr & 31, r = rr >> 5;
            int gy = 2 * oy0 - 1 + r;
            bool ok = true;
            if (reflect) gy = reflect_idx(gy, p.Hin); else ok = gy >= 0 && gy < p.Hin;
            const int cg = cb + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok && cg < Cin) {
                const float* sp = cg < p.C0 ? s0n + (size_t)cg * HWi : s1n + (size_t)(cg - p.C0) * HWi;
                v = *reinterpret_cast<const f32x4*>(sp + (size_t)gy * p.Win + 2 * ox0 + 4 * j);
            }
            xi4[i] = v;
        }
#pragma unroll
        for (int i = 0; i < XH_PT; ++i) {
            const int e = tid + i * NTH;
            float v = 0.f;
            if (e < NXH) {
                const int side = e & 1, rr = e >> 1;
                const int c = rr & 31, r = rr >> 5;
                int gy = 2 * oy0 - 1 + r;
                int gx = side == 0 ? 2 * ox0 - 1 : 2 * ox0 + 64;
                bool ok = true;
                if (reflect) { gy = reflect_idx(gy, p.Hin); gx = reflect_idx(gx, p.Win); }
                else ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
                const int cg = cb + c;
                if (ok && cg < Cin) {
                    const float* sp = cg < p.C0 ? s0n + (size_t)cg * HWi : s1n + (size_t)(cg - p.C0) * HWi;
                    v = sp[(size_t)gy * p.Win + gx];
                }
            }
            xh[i] = v;
        }
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            const int e = tid + i * NTH;
            const int j = e & 7, rr = e >> 3;
            const int o = rr & 63, r = rr >> 6;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ob + o < p.Cout) v = *reinterpret_cast<const f32x4*>(gn + (size_t)(ob + o) * HWo + (size_t)(oy0 + r) * p.Wout + ox0 + 4 * j);
            gv[i] = v;
        }
    };
    // raw row r of the tile = input row 2 oy0 - 1 + r: r even -> odd input rows = plane py 1, row r / 2; r odd -> plane py 0, row
    // (r - 1) / 2.  Columns alike: input column 2 ox0 - 1 + q: q even -> px 1, q / 2; q odd -> px 0, (q - 1) / 2.
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < XI_PT; ++i) {
            const int e = tid + i * NTH;
            const int j = e & 15, rr = e >> 4;
            const int c = rr & 31, r = rr >> 5;
            const int ppy = (r & 1) ? 0 : 1, prow = (r & 1) ? (r - 1) >> 1 : r >> 1;
            float* d0 = Xl + c * S2W_XP + (ppy * 2 + 0) * S2W_PL + prow * 34;       // px 0: q = 4j+1, 4j+3 -> columns 2j, 2j+1
            float* d1 = Xl + c * S2W_XP + (ppy * 2 + 1) * S2W_PL + prow * 34;       // px 1: q = 4j+2, 4j+4 -> columns 2j+1, 2j+2
            d0[2 * j] = xi4[i].x; d0[2 * j + 1] = xi4[i].z;
            d1[2 * j + 1] = xi4[i].y; d1[2 * j + 2] = xi4[i].w;
        }
#pragma unroll
        for (int i = 0; i < XH_PT; ++i) {
            const int e = tid + i * NTH;
            if (e < NXH) {
                const int side = e & 1, rr = e >> 1;
                const int c = rr & 31, r = rr >> 5;
                const int ppy = (r & 1) ? 0 : 1, prow = (r & 1) ? (r - 1) >> 1 : r >> 1;
                // left halo: q = 0 -> px 1, column 0;  right halo: q = 65 -> px 0, column 32
                Xl[c * S2W_XP + (ppy * 2 + (side == 0 ? 1 : 0)) * S2W_PL + prow * 34 + (side == 0 ? 0 : 32)] = xh[i];
            }
        }
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            const int e = tid + i * NTH;
            const int j = e & 7, rr = e >> 3;
            const int o = rr & 63, r = rr >> 6;
            f32x2w* gd = reinterpret_cast<f32x2w*>(Gl + o * WW_GP + r * 32 + 4 * j);
            gd[0] = f32x2w{gv[i].x, gv[i].y};
            gd[1] = f32x2w{gv[i].z, gv[i].w};
        }
    };
    auto next_tile = [&](int tile) {
        while (tile < p.ntiles) {
            const int n = tile / (p.tiles_x * p.tiles_y);
            if (p.valid == nullptr || p.valid[n] != 0) break;
            tile += p.nslices;
        }
        return tile;
    };
    // operands of k-step kk: blocks t = 2 kk + lk of the 2 x 16 block tile: the 3 x 3 patch of the wave's parity plane for
    // channel li, the 2 x 2 block of the output gradient for output channel m * 32 + li
    const float* xbase = Xl + li * S2W_XP + pp * S2W_PL;
    const float* gbase = Gl + (m * 32 + li) * WW_GP;
    auto load_ops = [&](int kk, float (&d)[3][3], f32x2w (&g)[2]) {
        const int t = 2 * kk + lk;
        const int by = t >> 4, bx = t & 15;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* pr = xbase + (2 * by + r) * 34 + 2 * bx;
            const f32x2w v = *reinterpret_cast<const f32x2w*>(pr);
            d[r][0] = v.x; d[r][1] = v.y; d[r][2] = pr[2];
        }
        g[0] = *reinterpret_cast<const f32x2w*>(gbase + (2 * by) * 32 + 2 * bx);
        g[1] = *reinterpret_cast<const f32x2w*>(gbase + (2 * by + 1) * 32 + 2 * bx);
    };

    int tile = next_tile((p.nslices & 7) == 0 ? (slice & 7) * (p.nslices >> 3) + (slice >> 3) : slice);
    if (tile < p.ntiles) prefetch(tile);
    while (tile < p.ntiles) {
        const int cand = tile + p.nslices;
        const int cflag = (p.valid != nullptr && cand < p.ntiles) ? p.valid[cand / (p.tiles_x * p.tiles_y)] : 1;
        commit();
        __syncthreads();
        const int nxt = cflag != 0 ? cand : next_tile(cand + p.nslices);
        if (nxt < p.ntiles) prefetch(nxt);          // in flight during the MFMA loop below
        float d[2][3][3];
        f32x2w g[2][2];
        load_ops(0, d[0], g[0]);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float(&dd)[3][3] = d[kk & 1];
            const f32x2w(&gg)[2] = g[kk & 1];
            // V = Bt D B: rows (d0 - d1, d1, d2 - d1), then the same on the columns;  Mt = A dY At: rows (y0, y0 + y1, y1), columns alike
            float V[3][3], Mt[3][3];
            {
                float t[3][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { t[0][c] = dd[0][c] - dd[1][c]; t[1][c] = dd[1][c]; t[2][c] = dd[2][c] - dd[1][c]; }
#pragma unroll
                for (int r = 0; r < 3; ++r) { V[r][0] = t[r][0] - t[r][1]; V[r][1] = t[r][1]; V[r][2] = t[r][2] - t[r][1]; }
                const f32x2w u1 = gg[0] + gg[1];
                const f32x2w u[3] = {gg[0], u1, gg[1]};
#pragma unroll
                for (int r = 0; r < 3; ++r) { Mt[r][0] = u[r].x; Mt[r][1] = u[r].x + u[r].y; Mt[r][2] = u[r].y; }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[0][0], Mt[0][0], acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < 16) load_ops(kk + 1, d[(kk + 1) & 1], g[(kk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 1; i < 9; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[i / 3][i % 3], Mt[i / 3][i % 3], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        tile = nxt;
    }
    // ---- d g = Gt dU G in registers: g[a][b] = sum of the 2 x 2 window of dU at (a, b); tap (ky, kx) = (2a + 1 - py, 2b + 1 - px)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int tap = (2 * a + 1 - py) * 4 + (2 * b + 1 - px);
            float* sl = p.slabs + (((size_t)slice * 16 + tap) * p.CinP + cb) * p.CoutB + ob + m * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = (acc[a * 3 + b][r] + acc[a * 3 + b + 1][r]) + (acc[(a + 1) * 3 + b][r] + acc[(a + 1) * 3 + b + 1][r]);
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * lk;
                sl[(size_t)ci * p.CoutB] = v;
            }
        }
}

int g_wgrad_f23 = -1, g_wgrad_f22 = -1;     // -1: from the environment; 0 / 1: set by c2s_wgrad_algorithms (A/B tests)
// F(2x2,2x2) path of the 4x4 stride-2 layers on planes that tile into 4 x 32 output pixels (C2S_S2WINO=0 keeps the direct kernel)
bool s2wino_wgrad(const c2s_wgrad_desc* d) {
    static const bool env_on = [] { const char* e = getenv("C2S_S2WINO"); const char* e2 = getenv("C2S_S2WINO_WGRAD");
                                    return !(e && e[0] == '0') && !(e2 && e2[0] == '0'); }();
    const bool enabled = g_wgrad_f22 >= 0 ? g_wgrad_f22 != 0 : env_on;
    return enabled && d->KH == 4 && d->KW == 4 && d->S == 2 && d->pad_y == 1 && d->pad_x == 1 && d->Hin == 2 * d->Hout &&
           d->Win == 2 * d->Wout && d->Wout % 32 == 0 && d->Hout % 4 == 0 && d->C0 + d->C1 >= 32 && d->Cout >= 32 &&
           d->C0 % 4 == 0;
}

// Winograd path: wide 3x3 layers on planes that tile into 4 x 32 pixel pieces (C2S_WINOGRAD=0 disables it)
bool wino_wgrad(const c2s_wgrad_desc* d) {
    static int env_on = -1;
    if (env_on < 0) {
        const char* e = getenv("C2S_WINOGRAD");
        env_on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    const int enabled = g_wgrad_f23 >= 0 ? g_wgrad_f23 : env_on;
    return enabled && d->KH == 3 && d->KW == 3 && d->S == 1 && d->pad_y == 1 && d->pad_x == 1 && d->C0 + d->C1 >= 32 &&
           d->Cout >= 32 && d->Win == d->Wout && d->Hin == d->Hout && d->Wout % 32 == 0 && d->Hout % 4 == 0;
}

struct TapTable {
    int off[16];
};

// Slice sum of one output element, shared by the per-layer and the batched kernel.  FOUR lanes per element (a launch has only
// tens of thousands of elements and a thread-per-element loop over hundreds of slices 100+ KB apart is a chain of memory round
// trips): lane q of a quad sums the slices k = q (mod 4) with 4 independent partial sums (loads in flight), the quad
// combines in a fixed order -- bitwise reproducible.
__device__ __forceinline__ float slice_sum4(const float* __restrict__ s, size_t stride, int nslices, int q) {
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    int k = q;
    for (; k + 12 < nslices; k += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) part[u] += s[(size_t)(k + 4 * u) * stride];
    }
    for (int u = 0; k < nslices; k += 4, ++u) part[u] += s[(size_t)k * stride];
    float acc = (part[0] + part[1]) + (part[2] + part[3]);
    acc += __shfl_xor(acc, 1, 64);          // (q, q^1), then (pair, pair^2): the same tree in every lane of the quad
    acc += __shfl_xor(acc, 2, 64);
    return acc;
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dst, int nslices, int NT,
                                    int Cin, int Cout, int CinP, int CoutB, long so, long sc, TapTable tt,
                                    int accumulate) {
    const long total = (long)NT * Cin * Cout;
    const long t4 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long e = t4 >> 2;
    const int q = (int)(t4 & 3);
    const bool live = e < total;
    const long ee = live ? e : total - 1;             // (whole quads stay converged for the shuffles)
    const int o = (int)(ee % Cout);
    const long tc = ee / Cout;
    const int c = (int)(tc % Cin), t = (int)(tc / Cin);
    const size_t stride = (size_t)NT * CinP * CoutB;
    const float acc = slice_sum4(slabs + ((size_t)t * CinP + c) * CoutB + o, stride, nslices, q);
    if (live && q == 0) {
        float* d = dst + o * so + c * sc + tt.off[t];
        *d = accumulate ? *d + acc : acc;
    }
}

void geometry(const c2s_wgrad_desc* d, int TP, int* log2pc, int* tiles_x, int* tiles_y) {
    int l2 = 5;
    while (l2 > 2 && (1 << l2) > d->Wout) --l2;
    *log2pc = l2;
    *tiles_x = cdiv(d->Wout, 1 << l2);
    *tiles_y = cdiv(d->Hout, TP >> l2);
}

template <int K, int S, int LW, int MODE>
int launch_wgrad_tile(const c2s_wgrad_desc* d, WgradParams& p, hipStream_t st) {
    constexpr int TP = TCfg<K, S>::TP, TW = 1 << LW, PR = TP / TW;
    constexpr int XR = (PR - 1) * S + K, XC = (TW - 1) * S + K;
    constexpr int CB = (MODE == 2) ? 16 : 32;
    p.log2pc = LW;
    p.tiles_x = d->Wout / TW;
    p.tiles_y = d->Hout / PR;
    p.ntiles = d->N * p.tiles_x * p.tiles_y;
    size_t fl = (size_t)CB * ((XR * XC) | 1) + (size_t)64 * (TP + 1);
    if (fl < 2 * 16 * 64) fl = 2 * 16 * 64;                       // exchange area of the position halves
    c2s_ensure_init();
    dim3 grid(p.nslices, p.CinP / 32, p.CoutB / 64);
    hipLaunchKernelGGL((conv_wgrad_tile_kernel<K, S, LW, MODE>), grid, dim3(256), fl * sizeof(float), st, p);
    C2S_CHECK_LAUNCH("conv_wgrad_tile");
    return C2S_OK;
}

template <int K, int S>
int launch_wgrad(const c2s_wgrad_desc* d, WgradParams& p, hipStream_t st) {
    using C = WCfg<K, S>;
    {   // tiled path: 32- or 16-wide aligned tiles
        constexpr int TP = TCfg<K, S>::TP;
        constexpr int MODE = (K == 4) ? 0 : 1;
        const int pad = (K == 1) ? 0 : 1;
        const bool geom = d->Win == d->Wout * S && d->Hin == d->Hout * S && d->pad_y == pad && d->pad_x == pad && d->Win % 4 == 0;
        if (geom && d->Wout % 32 == 0 && d->Hout % (TP / 32) == 0) {
            if constexpr (K == 3)
                if (d->C0 + d->C1 <= 10) return launch_wgrad_tile<K, S, 5, 2>(d, p, st);
            return launch_wgrad_tile<K, S, 5, MODE>(d, p, st);
        }
        if (geom && d->Wout == 16 && d->Hout % (TP / 16) == 0) return launch_wgrad_tile<K, S, 4, MODE>(d, p, st);
    }
    geometry(d, C::TP, &p.log2pc, &p.tiles_x, &p.tiles_y);
    p.ntiles = d->N * p.tiles_x * p.tiles_y;
    const int PC = 1 << p.log2pc, PR = C::TP >> p.log2pc;
    const int plane = ((PR - 1) * S + K) * ((PC - 1) * S + K);
    const size_t lds = ((size_t)32 * (plane | 1) + (size_t)64 * (C::TP + 1)) * sizeof(float);
    c2s_ensure_init();
    dim3 grid(p.nslices, p.CinP / 32, p.CoutB / 64);
    hipLaunchKernelGGL((conv_wgrad_kernel<K, S>), grid, dim3(256), lds, st, p);
    C2S_CHECK_LAUNCH("conv_wgrad");
    return C2S_OK;
}

int check(const c2s_wgrad_desc* d) {
    C2S_REQUIRE(d && d->N > 0 && d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "wgrad: bad channels");
    C2S_REQUIRE(d->KH == d->KW, "wgrad: square kernels only");
    C2S_REQUIRE(d->nslices > 0, "wgrad: nslices must be positive");
    C2S_REQUIRE(d->Hin < 1024 && d->Win < 1024, "wgrad: plane too large");
    return C2S_OK;
}

void init_hook() {
    C2S_RAISE_LDS((conv_wgrad_tile_kernel<3, 1, 5, 2>));
    C2S_RAISE_LDS((conv_wgrad_tile_kernel<3, 1, 5, 1>));
    C2S_RAISE_LDS((conv_wgrad_tile_kernel<3, 1, 4, 1>));
    C2S_RAISE_LDS((conv_wgrad_tile_kernel<1, 1, 5, 1>));
    C2S_RAISE_LDS((conv_wgrad_tile_kernel<1, 1, 4, 1>));
    C2S_RAISE_LDS((conv_wgrad_tile_kernel<4, 2, 5, 0>));
    C2S_RAISE_LDS((conv_wgrad_tile_kernel<4, 2, 4, 0>));
    C2S_RAISE_LDS((conv_wgrad_kernel<3, 1>));
    C2S_RAISE_LDS((conv_wgrad_kernel<1, 1>));
    C2S_RAISE_LDS((conv_wgrad_kernel<4, 2>));
    C2S_RAISE_LDS(conv_wgrad_winograd_kernel<2>);
    C2S_RAISE_LDS(conv_wgrad_s2wino_kernel);
}
C2sInitRegistrar registrar(init_hook);

}  // namespace

#ifdef C2S_WW_STAMP
extern "C" int c2s_debug_ww_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ww_stamps), sizeof(unsigned long long) * 1024 * 4) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int c2s_wgrad_algorithms(int winograd_3x3, int winograd_4x4s2) {
    g_wgrad_f23 = winograd_3x3 < 0 ? -1 : (winograd_3x3 != 0);
    g_wgrad_f22 = winograd_4x4s2 < 0 ? -1 : (winograd_4x4s2 != 0);
    return C2S_OK;
}

extern "C" size_t c2s_wgrad_workspace_floats(const c2s_wgrad_desc* d) {
    if (!d) return 0;
    const size_t CinP = (size_t)cdiv(d->C0 + d->C1, 32) * 32, CoutB = (size_t)cdiv(d->Cout, 64) * 64;
    return (size_t)d->nslices * d->KH * d->KW * CinP * CoutB;
}

extern "C" int c2s_conv_wgrad(const c2s_wgrad_desc* d, const float* src0, const float* src1, const float* gout,
                              float* slabs, size_t slab_floats, const int* valid, void* stream) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(src0 && gout && slabs && (d->C1 == 0 || src1), "wgrad: null pointer");
    C2S_REQUIRE(slab_floats >= c2s_wgrad_workspace_floats(d), "wgrad: slab workspace too small");
    WgradParams p;
    p.src0 = src0; p.src1 = src1; p.gout = gout; p.slabs = slabs; p.valid = valid;
    p.N = d->N; p.C0 = d->C0; p.C1 = d->C1; p.Hin = d->Hin; p.Win = d->Win; p.Cout = d->Cout;
    p.Hout = d->Hout; p.Wout = d->Wout; p.pad_y = d->pad_y; p.pad_x = d->pad_x; p.pad_mode = d->pad_mode;
    p.nslices = d->nslices;
    p.CinP = cdiv(d->C0 + d->C1, 32) * 32;
    p.CoutB = cdiv(d->Cout, 64) * 64;
    hipStream_t st = (hipStream_t)stream;
    if (wino_wgrad(d)) {
        c2s_ensure_init();      // conv_wgrad_winograd_kernel<2> needs 84 KB of dynamic LDS: the raise lives in the init hook
        p.tiles_x = d->Wout / 32;
        p.tiles_y = d->Hout / 4;
        p.ntiles = d->N * p.tiles_x * p.tiles_y;
        p.log2pc = 5;
        static const bool wide_ok = [] { const char* e = getenv("C2S_WGRAD_WINO_CB2"); return !(e && e[0] == '0'); }();
        if (wide_ok && p.CinP % 64 == 0) {
            const size_t ldsb = ((size_t)64 * WW_XP + (size_t)64 * WW_GP) * sizeof(float);
            dim3 grid(p.nslices, p.CinP / 64, p.CoutB / 64);
            hipLaunchKernelGGL(conv_wgrad_winograd_kernel<2>, grid, dim3(512), ldsb, st, p);
        } else {
            const size_t ldsb = ((size_t)32 * WW_XP + (size_t)64 * WW_GP) * sizeof(float);
            dim3 grid(p.nslices, p.CinP / 32, p.CoutB / 64);
            hipLaunchKernelGGL(conv_wgrad_winograd_kernel<1>, grid, dim3(256), ldsb, st, p);
        }
        C2S_CHECK_LAUNCH("conv_wgrad_winograd");
        return C2S_OK;
    }
    if (s2wino_wgrad(d)) {
        c2s_ensure_init();
        p.tiles_x = d->Wout / 32;
        p.tiles_y = d->Hout / 4;
        p.ntiles = d->N * p.tiles_x * p.tiles_y;
        p.log2pc = 5;
        const size_t ldsb = ((size_t)32 * S2W_XP + (size_t)64 * WW_GP) * sizeof(float);
        dim3 grid(p.nslices, p.CinP / 32, p.CoutB / 64);
        hipLaunchKernelGGL(conv_wgrad_s2wino_kernel, grid, dim3(512), ldsb, st, p);
        C2S_CHECK_LAUNCH("conv_wgrad_s2wino");
        return C2S_OK;
    }
    if (d->KH == 3 && d->S == 1) return launch_wgrad<3, 1>(d, p, st);
    if (d->KH == 1 && d->S == 1) return launch_wgrad<1, 1>(d, p, st);
    if (d->KH == 4 && d->S == 2) return launch_wgrad<4, 2>(d, p, st);
    c2s_set_error("wgrad: unsupported (K=%d,S=%d)", d->KH, d->S);
    return C2S_EINVAL;
}

// ---- all slice sums of a backward pass in ONE launch (26 launches of a few dozen workgroups each in a U-TAE step): the
// caller keeps one slab buffer per layer, builds a table of job records once and reuses it every step (as c2s_pack_batch)
namespace {
struct ReduceJob {
    const float* slabs;
    float* dst;
    long so, sc;
    int nslices, NT, Cin, Cout, CinP, CoutB, accumulate, block_start;
    int taps[16];
};

__global__ void wgrad_reduce_batch_kernel(const ReduceJob* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].block_start) ++j;      // a few dozen jobs: linear scan
    const ReduceJob& jb = jobs[j];
    const long total = (long)jb.NT * jb.Cin * jb.Cout;
    const long t4 = (long)(blockIdx.x - jb.block_start) * blockDim.x + threadIdx.x;
    const long e = t4 >> 2;
    const int q = (int)(t4 & 3);
    const bool live = e < total;
    const long ee = live ? e : total - 1;
    const int o = (int)(ee % jb.Cout);
    const long tc = ee / jb.Cout;
    const int c = (int)(tc % jb.Cin), t = (int)(tc / jb.Cin);
    const size_t stride = (size_t)jb.NT * jb.CinP * jb.CoutB;
    const float acc = slice_sum4(jb.slabs + ((size_t)t * jb.CinP + c) * jb.CoutB + o, stride, jb.nslices, q);   // (= wgrad_reduce_kernel)
    if (live && q == 0) {
        float* d = jb.dst + o * jb.so + c * jb.sc + jb.taps[t];
        *d = jb.accumulate ? *d + acc : acc;
    }
}
}  // namespace

extern "C" size_t c2s_wgrad_reduce_job_bytes(void) { return sizeof(ReduceJob); }

extern "C" int c2s_wgrad_reduce_job_blocks(const c2s_wgrad_desc* d) {
    if (int rc = check(d)) return rc;
    return cdiv((long)4 * d->KH * d->KW * (d->C0 + d->C1) * d->Cout, 256);      // four lanes per element
}

extern "C" int c2s_wgrad_reduce_job_fill(void* host_record, const c2s_wgrad_desc* d, const float* slabs, float* dst,
                                         long stride_o, long stride_c, const int* host_tap_off, int accumulate, int block_start) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(host_record && slabs && dst && host_tap_off && block_start >= 0, "wgrad_reduce_job_fill: bad args");
    ReduceJob* j = reinterpret_cast<ReduceJob*>(host_record);
    const int NT = d->KH * d->KW, Cin = d->C0 + d->C1;
    j->slabs = slabs; j->dst = dst; j->so = stride_o; j->sc = stride_c;
    j->nslices = d->nslices; j->NT = NT; j->Cin = Cin; j->Cout = d->Cout;
    j->CinP = cdiv(Cin, 32) * 32; j->CoutB = cdiv(d->Cout, 64) * 64;
    j->accumulate = accumulate; j->block_start = block_start;
    for (int i = 0; i < 16; ++i) j->taps[i] = i < NT ? host_tap_off[i] : 0;
    return C2S_OK;
}

extern "C" int c2s_wgrad_reduce_batch(const void* device_table, int njobs, int total_blocks, void* stream) {
    C2S_REQUIRE(device_table && njobs > 0 && total_blocks > 0, "wgrad_reduce_batch: bad args");
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const ReduceJob*>(device_table), njobs);
    C2S_CHECK_LAUNCH("wgrad_reduce_batch");
    return C2S_OK;
}

extern "C" int c2s_wgrad_reduce(const c2s_wgrad_desc* d, const float* slabs, float* dst, long stride_o, long stride_c,
                                const int* host_tap_off, int accumulate, void* stream) {
    if (int rc = check(d)) return rc;
    C2S_REQUIRE(slabs && dst && host_tap_off, "wgrad_reduce: null pointer");
    const int NT = d->KH * d->KW;
    const int Cin = d->C0 + d->C1;
    TapTable tt;
    for (int i = 0; i < 16; ++i) tt.off[i] = i < NT ? host_tap_off[i] : 0;
    const long total = (long)NT * Cin * d->Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(4 * total, 256)), dim3(256), 0, (hipStream_t)stream, slabs, dst,
                       d->nslices, NT, Cin, d->Cout, cdiv(Cin, 32) * 32, cdiv(d->Cout, 64) * 64, stride_o, stride_c, tt,
                       accumulate);
    C2S_CHECK_LAUNCH("wgrad_reduce");
    return C2S_OK;
}
