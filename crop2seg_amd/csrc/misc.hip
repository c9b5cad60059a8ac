// Small kernels of the hot path: frame flags, depthwise convolution (W-TAE), cross-entropy, flat Adam,
// fill / add.  All HBM- or latency-bound; see DESIGN.md for the per-kernel roofline notes.
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void c2s_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int c2s_abi_version(void) { return 4; }
extern "C" const char* c2s_last_error(void) { return g_err; }

// ---------------------------------------------------------------- per-device one-time set-up
namespace {
constexpr int kMaxDevices = 64;
struct InitState {
    std::mutex mu;
    std::vector<c2s_init_hook> hooks;
    std::atomic<uint64_t> done{0};          // bit d: device d is set up
    int cus[kMaxDevices] = {};
};
InitState& init_state() {
    static InitState* s = new InitState();   // construct on first use (registrars run during static initialisation)
    return *s;
}
int init_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        (void)hipGetLastError();
        return -1;
    }
    InitState& s = init_state();
    if (s.done.load(std::memory_order_acquire) >> dev & 1) return dev;
    std::lock_guard<std::mutex> lock(s.mu);
    if (s.done.load(std::memory_order_relaxed) >> dev & 1) return dev;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    s.cus[dev] = cus;
    for (c2s_init_hook h : s.hooks) h();
    (void)hipGetLastError();
    s.done.fetch_or(1ull << dev, std::memory_order_release);
    return dev;
}
}  // namespace

C2sInitRegistrar::C2sInitRegistrar(c2s_init_hook hook) { init_state().hooks.push_back(hook); }
void c2s_ensure_init() { (void)init_current_device(); }
int c2s_cus() {
    const int dev = init_current_device();
    return dev < 0 ? 256 : init_state().cus[dev];
}

extern "C" int c2s_init(int device) {
    int cur = -1;
    C2S_REQUIRE(hipGetDevice(&cur) == hipSuccess, "c2s_init: no HIP device");
    C2S_REQUIRE(device < 0 || device == cur, "c2s_init: device %d is not the current device (%d); make it current first", device, cur);
    C2S_REQUIRE(init_current_device() >= 0, "c2s_init: device index out of range");
    return C2S_OK;
}
extern "C" int c2s_device_cus(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return c2s_cus();
}

namespace {

// ---------------------------------------------------------------- frame flags (utae.py:201-203)
__global__ __launch_bounds__(256) void frame_flags_kernel(const float* __restrict__ x, int* __restrict__ valid,
                                                          long frame_elems, float pad_value, int chunks) {
    const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const long len = (frame_elems + chunks - 1) / chunks;
    const long beg = chunk * len, end = (beg + len) < frame_elems ? (beg + len) : frame_elems;
    const float* xf = x + (size_t)n * frame_elems;
    // a real frame shows itself in the first 256 values of a chunk (any value != pad_value): only padded frames are read in full
    // (before: every frame in full -- 84 MB per U-TAE step, 320 MB at TimeUNet's B = 8, T = 61)
    long i = beg + threadIdx.x;
    int any = (i < end) ? (xf[i] != pad_value) : 0;
    if (__syncthreads_or(any)) {
        if (threadIdx.x == 0) atomicOr(valid + n, 1);
        return;
    }
    for (i += 256; i < end; i += 256) any |= (xf[i] != pad_value);
    any = __any(any);
    if ((threadIdx.x & 63) == 0 && any) atomicOr(valid + n, 1);
}

// ---------------------------------------------------------------- depthwise conv (conv.py:18-24)
// HBM-bound.  grid = (pixel blocks, N*C planes); (K,S) are template parameters so the tap loops unroll and the
// K*K weights of the plane sit in registers.  A thread owns PX adjacent outputs of a row: per input row they read one
// span of (PX-1)*S + K columns starting at ox0*S - pad, which (pad == 1, PX == 4) is one scalar, one or two aligned
// float4s and one scalar instead of PX*K dwords.
template <int K, int S, int PX>
__device__ __forceinline__ void dw_load_span(const float* __restrict__ row, int xs, int Win, bool reflect, bool fast,
                                             float (&v)[(PX - 1) * S + K]) {
    constexpr int SP = (PX - 1) * S + K;
    if constexpr (PX == 4) {
        // pad == 1, Win % 4 == 0: xs + 1 is a multiple of 4, the middle of the span is always in range and only its two
        // end columns can fall on the padding (-1 -> 1, Win -> Win - 2 when reflecting, else zero) -- no divergent path
        (void)fast;
        const int xl = xs >= 0 ? xs : 1, xr = xs + SP - 1 < Win ? xs + SP - 1 : Win - 2;
        const float tl = row[xl], tr = row[xr];
        v[0] = (xs >= 0 || reflect) ? tl : 0.f;
#pragma unroll
        for (int q = 0; q < (SP - 2) / 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(row + xs + 1 + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[1 + 4 * q + i] = t[i];
        }
        v[SP - 1] = (xs + SP - 1 < Win || reflect) ? tr : 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            int gx = xs + i;
            bool ok = true;
            if (reflect) gx = reflect_idx(gx, Win); else ok = gx >= 0 && gx < Win;
            const float t = row[ok ? gx : 0];
            v[i] = ok ? t : 0.f;
        }
    }
}

template <int K, int S, int PX>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     float* __restrict__ out, const int* __restrict__ valid, int C,
                                                     int Hin, int Win, int pad, int reflect) {
    constexpr int SP = (PX - 1) * S + K;
    const int plane = blockIdx.y, c = plane % C, n = plane / C;
    if (valid != nullptr && valid[n] == 0) return;
    const int Ho = (Hin + 2 * pad - K) / S + 1, Wo = (Win + 2 * pad - K) / S + 1;
    const int W4 = Wo / PX;                                          // PX = 4 needs Wo % 4 == 0 (host dispatch)
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Ho * W4) return;
    const int oy = e / W4, ox0 = (e - oy * W4) * PX;
    float wk[K * K];
#pragma unroll
    for (int k = 0; k < K * K; ++k) wk[k] = w[(size_t)c * K * K + k];
    const float* ip = in + (size_t)plane * Hin * Win;
    const int xs = ox0 * S - pad;
    const bool fast = false;
    float acc[PX];
#pragma unroll
    for (int u = 0; u < PX; ++u) acc[u] = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        int gy = oy * S - pad + ky;
        bool oky = true;
        if (reflect) gy = reflect_idx(gy, Hin); else oky = gy >= 0 && gy < Hin;
        if (!oky) continue;                                          // a zero-padded row adds nothing
        float v[SP];
        dw_load_span<K, S, PX>(ip + (size_t)gy * Win, xs, Win, reflect != 0, fast, v);
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int u = 0; u < PX; ++u) acc[u] = fmaf(wk[ky * K + kx], v[u * S + kx], acc[u]);
    }
    float* op = out + (size_t)plane * Ho * Wo + (size_t)oy * Wo + ox0;
    if constexpr (PX == 4) {
        const f32x4 r = {acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(op) = r;
    } else {
        op[0] = acc[0];
    }
}

// data gradient, gather form: input pixel j receives from the padded positions that reflect onto it
// (j itself; -1 if j == 1; n if j == n-2) through every tap whose output index is integral and in range.
// Per dimension that is a short list of (output index, tap) pairs -- at most 3 for 3x3/s1 plus one mirrored, at most
// 2 + 2 for 4x4/s2 -- built once per thread without branches in the accumulation loop (invalid slots point at output 0
// with weight 0); the 2-D gradient is the product of the two lists.  Four adjacent pixels of a row per thread (float4
// store; the row list is shared).
template <int K, int S>
__device__ __forceinline__ void dw_pairs(int j, int n_in, int n_out, int pad, bool rf, int (&o)[4], int (&k)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = 0; k[i] = -1; }
    int cnt = 0;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        const int t = j + pad - kk;
        const bool ok = t >= 0 && (t % S) == 0 && t / S < n_out;
        if (ok && cnt < 4) { o[cnt] = t / S; k[cnt] = kk; ++cnt; }
    }
    const int qm = (rf && j == 1) ? -1 : ((rf && j == n_in - 2) ? n_in : -2);
    if (qm != -2) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int t = qm + pad - kk;
            const bool ok = t >= 0 && (t % S) == 0 && t / S < n_out;
            if (ok && cnt < 4) { o[cnt] = t / S; k[cnt] = kk; ++cnt; }
        }
    }
}

template <int K, int S, int PX>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const float* __restrict__ gout, const float* __restrict__ w,
                                                       float* __restrict__ gin, const int* __restrict__ valid, int C,
                                                       int Hin, int Win, int pad, int reflect, int accumulate) {
    __shared__ float ws[K * K + 1];
    const int plane = blockIdx.y, c = plane % C, n = plane / C;
    if (threadIdx.x < K * K) ws[threadIdx.x] = w[(size_t)c * K * K + threadIdx.x];
    if (threadIdx.x == K * K) ws[K * K] = 0.f;                      // weight of an empty slot
    __syncthreads();
    const int W4 = Win / PX;                                          // PX = 4 needs Win % 4 == 0 (host dispatch)
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Hin * W4) return;
    const int jy = e / W4, jx0 = (e - jy * W4) * PX;
    float* gp_in = gin + (size_t)plane * Hin * Win + (size_t)jy * Win + jx0;
    if (valid != nullptr && valid[n] == 0) {
        if (accumulate) return;                                      // gin += 0
#pragma unroll
        for (int u = 0; u < PX; ++u) gp_in[u] = 0.f;
        return;
    }
    const int Ho = (Hin + 2 * pad - K) / S + 1, Wo = (Win + 2 * pad - K) / S + 1;
    const float* gp = gout + (size_t)plane * Ho * Wo;
    const bool rf = reflect && pad >= 1;
    // PX == 4 (host: pad == 1, Win % 4 == 0, Win >= 8, Hin >= 4, Hin even for 4x4/s2): a plain stencil over gout with the
    // border handled in place -- columns/rows off the plane are dropped and the reflection fold adds one tap at input
    // columns 1 and Win - 2 and one (row, tap) slot at input rows 1 and Hin - 2; no divergent general path.
    if constexpr (PX == 4) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        const bool fl = rf && jx0 == 0, fr = rf && jx0 + 4 == Win;
        int rows[3], kys[3];
        if constexpr (K == 3 && S == 1) {
            rows[0] = jy + 1; kys[0] = rows[0] < Ho ? 0 : -1;
            rows[1] = jy;     kys[1] = 1;
            rows[2] = jy - 1; kys[2] = rows[2] >= 0 ? 2 : -1;
        } else {
            const int py = (jy + 1) & 1;                       // taps ky = py (row oyb), py + 2 (row oyb - 1)
            const int oyb = (jy + 1 - py) >> 1;
            rows[0] = oyb;     kys[0] = oyb < Ho ? py : -1;
            rows[1] = oyb - 1; kys[1] = oyb >= 1 ? py + 2 : -1;
            rows[2] = 0;       kys[2] = -1;
        }
        int xrow = -1, xky = 0;                                // the reflected slot: padded row -1 folds onto 1, Hin onto Hin-2
        if (rf && jy == 1) { xrow = 0; xky = 0; }
        if (rf && jy == Hin - 2) { xrow = Ho - 1; xky = K - 1; }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int rw, kyv;
            if (a < 3) { rw = rows[a]; kyv = kys[a]; } else { rw = xrow; kyv = xrow >= 0 ? xky : -1; }
            if (K == 4 && a == 2) continue;
            if (kyv < 0) continue;
            const float* wr = ws + kyv * K;
            if constexpr (K == 3 && S == 1) {
                const float* row = gp + (size_t)rw * Wo + jx0;                      // columns jx0-1 .. jx0+4
                float g[6];
                const float t0 = row[jx0 > 0 ? -1 : 0], t5 = row[jx0 + 4 < Wo ? 4 : 3];
                g[0] = jx0 > 0 ? t0 : 0.f;
                const f32x4 m = *reinterpret_cast<const f32x4*>(row);
                g[1] = m[0]; g[2] = m[1]; g[3] = m[2]; g[4] = m[3];
                g[5] = jx0 + 4 < Wo ? t5 : 0.f;
#pragma unroll
                for (int kxx = 0; kxx < 3; ++kxx)
#pragma unroll
                    for (int u = 0; u < 4; ++u) r[u] = fmaf(wr[kxx], g[u + 2 - kxx], r[u]);   // column x + 1 - kx
                if (fl) r[1] = fmaf(wr[0], g[1], r[1]);                             // padded column -1 = column 1, tap 0 of output 0
                if (fr) r[2] = fmaf(wr[2], g[4], r[2]);                             // padded column Win = column Win-2, tap 2
            } else {
                const int oxb = jx0 >> 1;                      // gout columns oxb-1 .. oxb+2 serve the four pixels
                const float* row = gp + (size_t)rw * Wo + oxb;
                const float t0 = row[oxb > 0 ? -1 : 0], t3 = row[oxb + 2 < Wo ? 2 : 1];
                const float g0 = oxb > 0 ? t0 : 0.f, g1 = row[0], g2 = row[1], g3 = oxb + 2 < Wo ? t3 : 0.f;
                // even pixel x: taps kx = 1 (column x/2), 3 (x/2 - 1); odd pixel: taps 0 (column (x+1)/2), 2 ((x-1)/2)
                r[0] = fmaf(wr[1], g1, fmaf(wr[3], g0, r[0]));
                r[1] = fmaf(wr[0], g2, fmaf(wr[2], g1, r[1]));
                r[2] = fmaf(wr[1], g2, fmaf(wr[3], g1, r[2]));
                r[3] = fmaf(wr[0], g3, fmaf(wr[2], g2, r[3]));
                if (fl) r[1] = fmaf(wr[0], g1, r[1]);                               // padded column -1: tap 0 of output 0
                if (fr) r[2] = fmaf(wr[3], g2, r[2]);                               // padded column Win: tap 3 of output Wo-1
            }
        }
        if (accumulate) {
            const f32x4 cur = *reinterpret_cast<const f32x4*>(gp_in);
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] += cur[u];
        }
        *reinterpret_cast<f32x4*>(gp_in) = r;
        return;
    }
    int oy[4], ky[4];
    dw_pairs<K, S>(jy, Hin, Ho, pad, rf, oy, ky);
    float res[PX];
#pragma unroll
    for (int u = 0; u < PX; ++u) {
        int ox[4], kx[4];
        dw_pairs<K, S>(jx0 + u, Win, Wo, pad, rf, ox, kx);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int wi = (ky[a] >= 0 && kx[bb] >= 0) ? ky[a] * K + kx[bb] : K * K;
                acc = fmaf(ws[wi], gp[oy[a] * Wo + ox[bb]], acc);
            }
        }
        res[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < PX; ++u) gp_in[u] = accumulate ? gp_in[u] + res[u] : res[u];
}

// partial[n][c][k] : one workgroup per (n, c)
template <int K, int S, int PX>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                       float* __restrict__ partial, const int* __restrict__ valid, int C,
                                                       int Hin, int Win, int pad, int reflect) {
    constexpr int SP = (PX - 1) * S + K;
    __shared__ float red[4][K * K];
    const int n = blockIdx.x / C;
    const int Ho = (Hin + 2 * pad - K) / S + 1, Wo = (Win + 2 * pad - K) / S + 1;
    const int W4 = Wo / PX;                                          // PX = 4 needs Wo % 4 == 0 (host dispatch)
    float acc[K * K];
#pragma unroll
    for (int k = 0; k < K * K; ++k) acc[k] = 0.f;
    if (valid == nullptr || valid[n] != 0) {
        const float* ip = in + (size_t)blockIdx.x * Hin * Win;
        const float* gp = gout + (size_t)blockIdx.x * Ho * Wo;
        for (int e = threadIdx.x; e < Ho * W4; e += 256) {
            const int oy = e / W4, ox0 = (e - oy * W4) * PX;
            float g[PX];
            if constexpr (PX == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(gp + (size_t)oy * Wo + ox0);
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = t[u];
            } else {
                g[0] = gp[(size_t)oy * Wo + ox0];
            }
            const int xs = ox0 * S - pad;
            const bool fast = false;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                int gy = oy * S - pad + ky;
                bool oky = true;
                if (reflect) gy = reflect_idx(gy, Hin); else oky = gy >= 0 && gy < Hin;
                if (!oky) continue;
                float v[SP];
                dw_load_span<K, S, PX>(ip + (size_t)gy * Win, xs, Win, reflect != 0, fast, v);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int u = 0; u < PX; ++u) acc[ky * K + kx] = fmaf(g[u], v[u * S + kx], acc[ky * K + kx]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K * K; ++k) {
        const float s = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K * K)
        partial[(size_t)blockIdx.x * K * K + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// gw[i] = sum_n partial[n][i]: one wave per output, lanes stride the frames (fixed lane assignment + fixed shuffle tree:
// reproducible); a serial loop per output took 33 us for 128 frames
__global__ __launch_bounds__(64) void dw_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw, int N,
                                                             int CKK) {
    const int i = blockIdx.x;
    double s = 0.0;
    for (int n = threadIdx.x; n < N; n += 64) s += partial[(size_t)n * CKK + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) gw[i] = (float)s;
}

// ---------------------------------------------------------------- cross entropy (train.py:463-468)
// nn.CrossEntropyLoss(weight=w, label_smoothing=eps), reduction "mean" (torch semantics with class weights):
//   loss_i = (1-eps) w[y_i] (-log p_i[y_i]) + (eps/K) sum_k w[k] (-log p_i[k]);  loss = sum_i loss_i / sum_i w[y_i]
// Pixels whose target equals ignore_index (torch default -100) contribute to neither sum.  A target outside [0,K) that is
// not ignore_index makes torch raise (device assert); here it is treated like ignore_index -- no out-of-bounds read.
// per block partial (sum loss_i, sum w) -> ws[2*blocks]; finalize -> ws_tot[2]; grad kernel uses ws_tot[1]
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ cw, float* __restrict__ part, int B, int K,
                                                     int HW, float eps, long long ignore_index) {
    __shared__ float red[4][3];
    const long total = (long)B * HW;
    float num = 0.f, den = 0.f, bad = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long long yy = target[e];
        if (yy == ignore_index) continue;
        if (yy < 0 || yy >= K) { bad += 1.f; continue; }     // torch raises (device assert); here: skipped and counted
        const int y = (int)yy;
        const int pix = (int)(e % HW), b = (int)(e / HW);
        const float* lp = logits + (size_t)b * K * HW + pix;
        float mx = lp[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[(size_t)k * HW]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += __expf(lp[(size_t)k * HW] - mx);
        const float w = cw[y];
        const float lse = mx + __logf(s);
        float li = w * (lse - lp[(size_t)y * HW]);
        if (eps > 0.f) {
            float sm = 0.f;
            for (int k = 0; k < K; ++k) sm += cw[k] * (lse - lp[(size_t)k * HW]);
            li = (1.f - eps) * li + (eps / (float)K) * sm;
        }
        num += li;
        den += w;
    }
    num = wave_sum(num); den = wave_sum(den); bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = num; red[threadIdx.x >> 6][1] = den; red[threadIdx.x >> 6][2] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        part[blockIdx.x * 3 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        part[blockIdx.x * 3 + 2] = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    }
}

__global__ void ce_finalize_kernel(const float* __restrict__ part, float* __restrict__ tot, float* __restrict__ loss,
                                   int blocks) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double num = 0, den = 0, bad = 0;
    for (int i = 0; i < blocks; ++i) { num += part[i * 3]; den += part[i * 3 + 1]; bad += part[i * 3 + 2]; }
    tot[0] = (float)num; tot[1] = (float)den; tot[2] = (float)bad;
    *loss = (float)(num / den);
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ cw, const float* __restrict__ tot,
                                                     float* __restrict__ glogits, int B, int K, int HW, float eps,
                                                     long long ignore_index) {
    const long total = (long)B * HW;
    const float inv_den = 1.f / tot[1];
    float wsum = 0.f;
    if (eps > 0.f)
        for (int k = 0; k < K; ++k) wsum += cw[k];
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(e % HW), b = (int)(e / HW);
        const float* lp = logits + (size_t)b * K * HW + pix;
        float* gp = glogits + (size_t)b * K * HW + pix;
        const long long yy = target[e];
        if (yy == ignore_index || yy < 0 || yy >= K) {
            for (int k = 0; k < K; ++k) gp[(size_t)k * HW] = 0.f;
            continue;
        }
        const int y = (int)yy;
        float mx = lp[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[(size_t)k * HW]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += __expf(lp[(size_t)k * HW] - mx);
        const float w = cw[y] * inv_den, inv_s = 1.f / s;
        if (eps > 0.f) {
            // (1-eps) w_y (p_k - delta_ky) + (eps/K) (W p_k - w_k), all over sum_i w[y_i]
            const float a = (1.f - eps) * w, c = eps / (float)K * inv_den;
            for (int k = 0; k < K; ++k) {
                const float pk = __expf(lp[(size_t)k * HW] - mx) * inv_s;
                gp[(size_t)k * HW] = a * (pk - (k == y ? 1.f : 0.f)) + c * (wsum * pk - cw[k]);
            }
        } else {
            for (int k = 0; k < K; ++k)
                gp[(size_t)k * HW] = w * (__expf(lp[(size_t)k * HW] - mx) * inv_s - (k == y ? 1.f : 0.f));
        }
    }
}

// ---------------------------------------------------------------- Adam (train.py:454, torch defaults)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, int step,
                            const int* __restrict__ step_dev, float gscale) {
    // the step count may live on the device so that a captured graph advances it on replay
    const int st = step_dev != nullptr ? *step_dev : step;
    const float bc1 = 1.f - powf(b1, (float)st);
    const float bc2s = sqrtf(1.f - powf(b2, (float)st));
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gg = g[i] * gscale;
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        const float denom = sqrtf(vv) / bc2s + eps;
        p[i] -= (lr / bc1) * (mm / denom);
    }
}

__global__ void fill_kernel(float* p, long n, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void add_kernel(float* __restrict__ d, const float* __restrict__ s, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] += s[i];
}

inline int grid_for(long n, int cap = 4096) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
constexpr int CE_BLOCKS = 512;

}  // namespace

extern "C" int c2s_frame_flags(const float* x, int* valid, int N, long frame_elems, float pad_value, void* stream) {
    C2S_REQUIRE(x && valid && N > 0 && frame_elems > 0, "frame_flags: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(valid, 0, (size_t)N * sizeof(int), st);
    int chunks = (int)((frame_elems + 16383) / 16384);
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(frame_flags_kernel, dim3(N * chunks), dim3(256), 0, st, x, valid, frame_elems, pad_value, chunks);
    C2S_CHECK_LAUNCH("frame_flags");
    return C2S_OK;
}

static int dw_check(int N, int C, int Hin, int Win, int K, int S, int pad, int pad_mode) {
    C2S_REQUIRE(N > 0 && C > 0 && Hin > 0 && Win > 0, "dwconv: bad shape");
    C2S_REQUIRE(((K == 3 && S == 1) || (K == 4 && S == 2)) && pad == 1, "dwconv: built for (K,S,pad) = (3,1,1) and (4,2,1)");
    if (pad_mode == C2S_PAD_REFLECT) C2S_REQUIRE(Hin >= 2 && Win >= 2, "dwconv: reflect needs planes >= 2x2");
    return C2S_OK;
}

extern "C" int c2s_dwconv_fwd(const float* in, const float* w, float* out, const int* valid, int N, int C, int Hin,
                              int Win, int K, int S, int pad, int pad_mode, void* stream) {
    if (int rc = dw_check(N, C, Hin, Win, K, S, pad, pad_mode)) return rc;
    C2S_REQUIRE(in && w && out, "dwconv_fwd: null pointer");
    const int Ho = (Hin + 2 * pad - K) / S + 1, Wo = (Win + 2 * pad - K) / S + 1;
    const bool reflect = pad_mode == C2S_PAD_REFLECT;
    hipStream_t st = (hipStream_t)stream;
    if (Wo % 4 == 0) {
        const dim3 grid(cdiv(Ho * (Wo / 4), 256), N * C);
        if (K == 3) hipLaunchKernelGGL((dw_fwd_kernel<3, 1, 4>), grid, dim3(256), 0, st, in, w, out, valid, C, Hin, Win, pad, reflect);
        else hipLaunchKernelGGL((dw_fwd_kernel<4, 2, 4>), grid, dim3(256), 0, st, in, w, out, valid, C, Hin, Win, pad, reflect);
    } else {
        const dim3 grid(cdiv(Ho * Wo, 256), N * C);
        if (K == 3) hipLaunchKernelGGL((dw_fwd_kernel<3, 1, 1>), grid, dim3(256), 0, st, in, w, out, valid, C, Hin, Win, pad, reflect);
        else hipLaunchKernelGGL((dw_fwd_kernel<4, 2, 1>), grid, dim3(256), 0, st, in, w, out, valid, C, Hin, Win, pad, reflect);
    }
    C2S_CHECK_LAUNCH("dwconv_fwd");
    return C2S_OK;
}

extern "C" int c2s_dwconv_dgrad(const float* gout, const float* w, float* gin, const int* valid, int N, int C, int Hin,
                                int Win, int K, int S, int pad, int pad_mode, int accumulate, void* stream) {
    if (int rc = dw_check(N, C, Hin, Win, K, S, pad, pad_mode)) return rc;
    C2S_REQUIRE(gout && w && gin, "dwconv_dgrad: null pointer");
    const bool reflect = pad_mode == C2S_PAD_REFLECT;
    hipStream_t st = (hipStream_t)stream;
    if (Win % 4 == 0 && Win >= 8 && Hin >= 4 && (K == 3 || Hin % 2 == 0)) {
        const dim3 grid(cdiv(Hin * (Win / 4), 256), N * C);
        if (K == 3) hipLaunchKernelGGL((dw_dgrad_kernel<3, 1, 4>), grid, dim3(256), 0, st, gout, w, gin, valid, C, Hin, Win, pad, reflect, accumulate);
        else hipLaunchKernelGGL((dw_dgrad_kernel<4, 2, 4>), grid, dim3(256), 0, st, gout, w, gin, valid, C, Hin, Win, pad, reflect, accumulate);
    } else {
        const dim3 grid(cdiv(Hin * Win, 256), N * C);
        if (K == 3) hipLaunchKernelGGL((dw_dgrad_kernel<3, 1, 1>), grid, dim3(256), 0, st, gout, w, gin, valid, C, Hin, Win, pad, reflect, accumulate);
        else hipLaunchKernelGGL((dw_dgrad_kernel<4, 2, 1>), grid, dim3(256), 0, st, gout, w, gin, valid, C, Hin, Win, pad, reflect, accumulate);
    }
    C2S_CHECK_LAUNCH("dwconv_dgrad");
    return C2S_OK;
}

extern "C" int c2s_dwconv_wgrad(const float* in, const float* gout, float* partial, float* gw, const int* valid, int N,
                                int C, int Hin, int Win, int K, int S, int pad, int pad_mode, void* stream) {
    if (int rc = dw_check(N, C, Hin, Win, K, S, pad, pad_mode)) return rc;
    C2S_REQUIRE(in && gout && partial && gw, "dwconv_wgrad: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool reflect = pad_mode == C2S_PAD_REFLECT;
    const int Wo = (Win + 2 * pad - K) / S + 1;
    const dim3 grid(N * C);
    if (Wo % 4 == 0) {
        if (K == 3) hipLaunchKernelGGL((dw_wgrad_kernel<3, 1, 4>), grid, dim3(256), 0, st, in, gout, partial, valid, C, Hin, Win, pad, reflect);
        else hipLaunchKernelGGL((dw_wgrad_kernel<4, 2, 4>), grid, dim3(256), 0, st, in, gout, partial, valid, C, Hin, Win, pad, reflect);
    } else {
        if (K == 3) hipLaunchKernelGGL((dw_wgrad_kernel<3, 1, 1>), grid, dim3(256), 0, st, in, gout, partial, valid, C, Hin, Win, pad, reflect);
        else hipLaunchKernelGGL((dw_wgrad_kernel<4, 2, 1>), grid, dim3(256), 0, st, in, gout, partial, valid, C, Hin, Win, pad, reflect);
    }
    C2S_CHECK_LAUNCH("dwconv_wgrad");
    hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(C * K * K), dim3(64), 0, st, partial, gw, N, C * K * K);
    C2S_CHECK_LAUNCH("dwconv_wgrad_reduce");
    return C2S_OK;
}

extern "C" size_t c2s_cross_entropy_workspace_floats(int B, int HW) {
    (void)B; (void)HW;
    return 3 * CE_BLOCKS + 3;
}

extern "C" int c2s_cross_entropy(const float* logits, const int64_t* target, const float* class_w, float* loss,
                                 float* glogits, int B, int K, int HW, float label_smoothing, long long ignore_index,
                                 float* workspace, size_t ws_floats, void* stream) {
    C2S_REQUIRE(logits && target && class_w && loss && workspace, "cross_entropy: null pointer");
    C2S_REQUIRE(ws_floats >= 3 * CE_BLOCKS + 3 && B > 0 && K > 0 && HW > 0, "cross_entropy: bad args");
    C2S_REQUIRE(label_smoothing >= 0.f && label_smoothing <= 1.f, "cross_entropy: label_smoothing must be in [0, 1]");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = grid_for((long)B * HW, CE_BLOCKS);
    float* tot = workspace + 3 * CE_BLOCKS;      // (sum, weight sum, targets outside [0,K) other than ignore_index)
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(blocks), dim3(256), 0, st, logits, target, class_w, workspace, B, K, HW,
                       label_smoothing, ignore_index);
    C2S_CHECK_LAUNCH("ce_fwd");
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, tot, loss, blocks);
    C2S_CHECK_LAUNCH("ce_finalize");
    if (glogits != nullptr) {
        hipLaunchKernelGGL(ce_bwd_kernel, dim3(blocks), dim3(256), 0, st, logits, target, class_w, tot, glogits, B, K, HW,
                           label_smoothing, ignore_index);
        C2S_CHECK_LAUNCH("ce_bwd");
    }
    return C2S_OK;
}

extern "C" int c2s_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                             float eps, int step, const int* step_dev, float grad_scale, void* stream) {
    C2S_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_dev != nullptr), "adam: bad args");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2,
                       eps, step, step_dev, grad_scale);
    C2S_CHECK_LAUNCH("adam");
    return C2S_OK;
}

extern "C" int c2s_fill(float* p, long n, float v, void* stream) {
    C2S_REQUIRE(p && n >= 0, "fill: bad args");
    if (n == 0) return C2S_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, v);
    C2S_CHECK_LAUNCH("fill");
    return C2S_OK;
}

extern "C" int c2s_add_inplace(float* dst, const float* src, long n, void* stream) {
    C2S_REQUIRE(dst && src && n >= 0, "add: bad args");
    if (n == 0) return C2S_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    C2S_CHECK_LAUNCH("add");
    return C2S_OK;
}
