// GroupNorm / BatchNorm (+ReLU, +residual) forward and backward for NCHW fp32 activations.
// HBM-bound: every kernel that touches activations is "one wave per (row, segment)" with float4 loads,
// a row being the HW contiguous floats of one (frame, channel); per-row scale/shift and backward
// coefficients are wave-uniform scalars.  The tiny per-group reductions run in double precision.
//
// Reference call sites replaced: nn.GroupNorm(4)+ReLU (src/backbones/conv.py:56-60,85-88),
// nn.BatchNorm2d+ReLU (conv.py:52-53,380,388), the residual adds (conv.py:292,410) and their backward.
#include "common.h"

namespace {

constexpr int SEG = 2048;  // floats per wave-segment (32 per lane = 8 float4 in flight)

__host__ __device__ inline int seg_len(int HW) { return HW < SEG ? HW : SEG; }
__host__ __device__ inline int n_segs(int HW) { return (HW + seg_len(HW) - 1) / seg_len(HW); }

// ---------------------------------------------------------------- forward statistics
// part[row][seg] = (mean, M2) of the segment (two passes over the segment; the second hits L1/L2)
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                        const int* __restrict__ valid, int C, int HW, int segs,
                                                        long nitems) {
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63;
    const long row = item / segs;
    const int seg = (int)(item % segs);
    if (valid != nullptr && valid[row / C] == 0) return;
    const int L = seg_len(HW);
    const int beg = seg * L;
    const int len = (HW - beg) < L ? (HW - beg) : L;
    const float* xr = x + row * HW + beg;
    float s = 0.f;
    if ((len & 3) == 0) {
        for (int i = lane * 4; i < len; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = lane; i < len; i += 64) s += xr[i];
    }
    s = wave_sum(s);
    const float mean = s / (float)len;
    float m2 = 0.f;
    if ((len & 3) == 0) {
        for (int i = lane * 4; i < len; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            m2 += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (int i = lane; i < len; i += 64) {
            const float a = xr[i] - mean;
            m2 += a * a;
        }
    }
    m2 = wave_sum(m2);
    if (lane == 0) {
        part[item * 2 + 0] = mean;
        part[item * 2 + 1] = m2;
    }
}

// one wave per group; combines (mean, M2) partials of equal-or-known length with Chan's formula in double.
// GROUP: group = (n, g): rows (n, g*cpg .. g*cpg+cpg-1).   BATCH: group = c: rows (n, c) for all valid n.
__global__ __launch_bounds__(64) void norm_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean,
                                                           float* __restrict__ rvar, float* __restrict__ gstats,
                                                           float* __restrict__ row_ab, const int* __restrict__ valid,
                                                           c2s_norm_desc d, int segs) {
    const int grp = blockIdx.x;
    const int lane = threadIdx.x;
    const int L = seg_len(d.HW);
    const bool batch = d.kind == C2S_NORM_BATCH;
    const int cpg = batch ? 1 : d.C / d.groups;
    const int n_rows = batch ? d.N : cpg;         // rows in this group
    const int items = n_rows * segs;
    float mean_f, rstd_f;
    if (batch && !d.training) {
        mean_f = rmean[grp];
        rstd_f = rsqrtf(rvar[grp] + d.eps);
    } else {
        if (!batch && valid != nullptr && valid[grp / d.groups] == 0) {
            for (int r = lane; r < cpg; r += 64) {
                const long row = (long)(grp / d.groups) * d.C + (grp % d.groups) * cpg + r;
                row_ab[row * 3] = 0.f;
                row_ab[row * 3 + 1] = 0.f;
                row_ab[row * 3 + 2] = 0.f;
            }
            if (lane == 0) { gstats[grp * 2] = 0.f; gstats[grp * 2 + 1] = 0.f; }
            return;
        }
        double cnt = 0.0, sum = 0.0;
        for (int it = lane; it < items; it += 64) {
            const int r = it / segs, sg = it % segs;
            const long row = batch ? (long)r * d.C + grp : (long)(grp / d.groups) * d.C + (grp % d.groups) * cpg + r;
            if (batch && valid != nullptr && valid[r] == 0) continue;
            const int len = (d.HW - sg * L) < L ? (d.HW - sg * L) : L;
            cnt += len;
            sum += (double)part[(row * segs + sg) * 2] * len;
        }
        for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); sum += __shfl_xor(sum, o, 64); }
        const double mean = cnt > 0 ? sum / cnt : 0.0;
        double m2 = 0.0;
        for (int it = lane; it < items; it += 64) {
            const int r = it / segs, sg = it % segs;
            const long row = batch ? (long)r * d.C + grp : (long)(grp / d.groups) * d.C + (grp % d.groups) * cpg + r;
            if (batch && valid != nullptr && valid[r] == 0) continue;
            const int len = (d.HW - sg * L) < L ? (d.HW - sg * L) : L;
            const double dm = (double)part[(row * segs + sg) * 2] - mean;
            m2 += (double)part[(row * segs + sg) * 2 + 1] + dm * dm * len;
        }
        for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
        const double var = cnt > 0 ? m2 / cnt : 0.0;
        mean_f = (float)mean;
        rstd_f = (float)(1.0 / sqrt(var + (double)d.eps));
        if (batch && lane == 0 && rmean != nullptr) {
            const double unb = cnt > 1 ? m2 / (cnt - 1.0) : var;
            rmean[grp] = (1.f - d.momentum) * rmean[grp] + d.momentum * mean_f;
            rvar[grp] = (1.f - d.momentum) * rvar[grp] + d.momentum * (float)unb;
        }
    }
    if (lane == 0) { gstats[grp * 2] = mean_f; gstats[grp * 2 + 1] = rstd_f; }
    for (int r = lane; r < n_rows; r += 64) {
        const long row = batch ? (long)r * d.C + grp : (long)(grp / d.groups) * d.C + (grp % d.groups) * cpg + r;
        const int c = (int)(row % d.C);
        row_ab[row * 3] = gamma[c] * rstd_f;
        row_ab[row * 3 + 1] = beta[c];
        row_ab[row * 3 + 2] = mean_f;
    }
}

// ---------------------------------------------------------------- forward apply
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ row_ab,
                                                         const float* __restrict__ res, float* __restrict__ y,
                                                         const int* __restrict__ valid, int C, int HW, int segs,
                                                         long nitems, int relu, float pad_value) {
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63;
    const long row = item / segs;
    const int seg = (int)(item % segs);
    const int L = seg_len(HW);
    const int beg = seg * L;
    const int len = (HW - beg) < L ? (HW - beg) : L;
    const size_t base = (size_t)row * HW + beg;
    const bool ok = valid == nullptr || valid[row / C] != 0;
    if (!ok) {
        for (int i = lane; i < len; i += 64) y[base + i] = pad_value;
        return;
    }
    const float a = row_ab[row * 3], b = row_ab[row * 3 + 1], mu = row_ab[row * 3 + 2];
    if ((len & 3) == 0) {
        for (int i = lane * 4; i < len; i += 256) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + base + i);
            v.x = (v.x - mu) * a + b; v.y = (v.y - mu) * a + b; v.z = (v.z - mu) * a + b; v.w = (v.w - mu) * a + b;
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (res != nullptr) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(res + base + i);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            *reinterpret_cast<f32x4*>(y + base + i) = v;
        }
    } else {
        for (int i = lane; i < len; i += 64) {
            float v = (x[base + i] - mu) * a + b;
            if (relu) v = fmaxf(v, 0.f);
            if (res != nullptr) v += res[base + i];
            y[base + i] = v;
        }
    }
}

// ---------------------------------------------------------------- backward
// part[item] = (sum g', sum g'*xhat, sum x) with g' = g * [a*x+b > 0] (relu) ; xhat = (x-mean)*rstd
__global__ __launch_bounds__(256) void norm_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ row_ab,
                                                            const float* __restrict__ gstats, float* __restrict__ part,
                                                            const int* __restrict__ valid, c2s_norm_desc d, int segs,
                                                            long nitems, int relu) {
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63;
    const long row = item / segs;
    const int seg = (int)(item % segs);
    const int n = (int)(row / d.C), c = (int)(row % d.C);
    if (valid != nullptr && valid[n] == 0) {
        if (lane == 0) { part[item * 3] = 0.f; part[item * 3 + 1] = 0.f; part[item * 3 + 2] = 0.f; }
        return;
    }
    const int L = seg_len(d.HW);
    const int beg = seg * L;
    const int len = (d.HW - beg) < L ? (d.HW - beg) : L;
    const size_t base = (size_t)row * d.HW + beg;
    const int grp = d.kind == C2S_NORM_BATCH ? c : n * d.groups + c / (d.C / d.groups);
    const float mean = gstats[grp * 2], rstd = gstats[grp * 2 + 1];
    const float a = row_ab[row * 3], b = row_ab[row * 3 + 1];
    float s1 = 0.f, s2 = 0.f, sx = 0.f;
    if ((len & 3) == 0) {
        for (int i = lane * 4; i < len; i += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + base + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xx = xv[k];
                const float gg = (!relu || (xx - mean) * a + b > 0.f) ? gv[k] : 0.f;
                s1 += gg;
                s2 += gg * ((xx - mean) * rstd);
                sx += xx;
            }
        }
    } else {
        for (int i = lane; i < len; i += 64) {
            const float xx = x[base + i];
            const float gg = (!relu || (xx - mean) * a + b > 0.f) ? g[base + i] : 0.f;
            s1 += gg;
            s2 += gg * ((xx - mean) * rstd);
            sx += xx;
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2); sx = wave_sum(sx);
    if (lane == 0) { part[item * 3] = s1; part[item * 3 + 1] = s2; part[item * 3 + 2] = sx; }
}

// One launch for the two small reductions between the sums pass and the apply pass (wave per block):
//   blocks [0, ngroups):      coefficients k1,k2,k3 per row of the group ( dx = k1*g' + k2*(x-mean) + k3 )
//   blocks [ngroups, +C):     dgamma, dbeta of one channel (sum over frames)
// Both read the per-(row, segment) partials directly; sums run in double in a fixed order.
__device__ __forceinline__ void row_sums(const float* __restrict__ part, long row, int segs, double& s1, double& s2) {
    s1 = 0.0; s2 = 0.0;
    for (int s = 0; s < segs; ++s) {
        s1 += part[(row * segs + s) * 3];
        s2 += part[(row * segs + s) * 3 + 1];
    }
}

__global__ __launch_bounds__(64) void norm_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                             const float* __restrict__ gstats, float* __restrict__ rowk,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             const int* __restrict__ valid, c2s_norm_desc d, int segs,
                                                             int ngroups) {
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= ngroups) {
        const int c = blockIdx.x - ngroups;
        double dg = 0, db = 0;
        for (int n = lane; n < d.N; n += 64) {
            if (valid != nullptr && valid[n] == 0) continue;
            double s1, s2;
            row_sums(part, (long)n * d.C + c, segs, s1, s2);
            dg += (double)(float)s2;
            db += (double)(float)s1;
        }
        for (int o = 32; o > 0; o >>= 1) { dg += __shfl_xor(dg, o, 64); db += __shfl_xor(db, o, 64); }
        if (lane == 0) {
            if (dgamma != nullptr) dgamma[c] = (float)dg;
            if (dbeta != nullptr) dbeta[c] = (float)db;
        }
        return;
    }
    const int grp = blockIdx.x;
    const bool batch = d.kind == C2S_NORM_BATCH;
    const int cpg = batch ? 1 : d.C / d.groups;
    const int n_rows = batch ? d.N : cpg;
    const float rstd = gstats[grp * 2 + 1];
    double A = 0.0, Bv = 0.0, cnt = 0.0;
    for (int r = lane; r < n_rows; r += 64) {
        const long row = batch ? (long)r * d.C + grp : (long)(grp / d.groups) * d.C + (grp % d.groups) * cpg + r;
        const int n = (int)(row / d.C);
        if (valid != nullptr && valid[n] == 0) continue;
        const float gm = gamma[row % d.C];
        double s1, s2;
        row_sums(part, row, segs, s1, s2);
        A += (double)gm * (double)(float)s1;
        Bv += (double)gm * (double)(float)s2;
        cnt += d.HW;
    }
    for (int o = 32; o > 0; o >>= 1) { A += __shfl_xor(A, o, 64); Bv += __shfl_xor(Bv, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    const bool frozen = batch && !d.training;   // eval-mode BatchNorm: statistics are constants
    const double m = cnt > 0 ? cnt : 1.0;
    for (int r = lane; r < n_rows; r += 64) {
        const long row = batch ? (long)r * d.C + grp : (long)(grp / d.groups) * d.C + (grp % d.groups) * cpg + r;
        const int n = (int)(row / d.C);
        float k1 = 0.f, k2 = 0.f, k3 = 0.f;
        if (valid == nullptr || valid[n] != 0) {
            k1 = rstd * gamma[row % d.C];
            if (!frozen) {
                k2 = (float)(-(double)rstd * rstd * Bv / m);
                k3 = (float)(-(double)rstd * A / m);
            }
        }
        rowk[row * 3] = k1; rowk[row * 3 + 1] = k2; rowk[row * 3 + 2] = k3;
    }
}

// gradient of the producing convolution's bias: per-channel sum of dx, from the per-(row,segment) partials the
// apply pass leaves in part[item*3]; one wave per channel, lanes over (frame, segment)
__global__ __launch_bounds__(64) void norm_bwd_dbias_kernel(const float* __restrict__ part, float* __restrict__ dbias,
                                                            const int* __restrict__ valid, int N, int C, int segs) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0;
    for (int it = lane; it < N * segs; it += 64) {
        const int n = it / segs, k = it - n * segs;
        if (valid != nullptr && valid[n] == 0) continue;
        s += part[(((long)n * C + c) * segs + k) * 3];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) dbias[c] = (float)s;
}

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ row_ab,
                                                             const float* __restrict__ rowk, float* __restrict__ gx,
                                                             float* __restrict__ part,
                                                             const int* __restrict__ valid, int C, int HW, int segs,
                                                             long nitems, int relu) {
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63;
    const long row = item / segs;
    const int seg = (int)(item % segs);
    const int L = seg_len(HW);
    const int beg = seg * L;
    const int len = (HW - beg) < L ? (HW - beg) : L;
    const size_t base = (size_t)row * HW + beg;
    if (valid != nullptr && valid[row / C] == 0) {
        for (int i = lane; i < len; i += 64) gx[base + i] = 0.f;
        if (lane == 0) part[item * 3] = 0.f;
        return;
    }
    const float a = row_ab[row * 3], b = row_ab[row * 3 + 1], mu = row_ab[row * 3 + 2];
    const float k1 = rowk[row * 3], k2 = rowk[row * 3 + 1], k3 = rowk[row * 3 + 2];
    float sdx = 0.f;
    if ((len & 3) == 0) {
        for (int i = lane * 4; i < len; i += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + base + i);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xc = xv[k] - mu;
                const float gg = (!relu || xc * a + b > 0.f) ? gv[k] : 0.f;
                o[k] = k1 * gg + k2 * xc + k3;
                sdx += o[k];
            }
            *reinterpret_cast<f32x4*>(gx + base + i) = o;
        }
    } else {
        for (int i = lane; i < len; i += 64) {
            const float xc = x[base + i] - mu;
            const float gg = (!relu || xc * a + b > 0.f) ? g[base + i] : 0.f;
            const float o = k1 * gg + k2 * xc + k3;
            gx[base + i] = o;
            sdx += o;
        }
    }
    sdx = wave_sum(sdx);
    if (lane == 0) part[item * 3] = sdx;
}

int check_desc(const c2s_norm_desc* d) {
    C2S_REQUIRE(d && d->N > 0 && d->C > 0 && d->HW > 0, "norm: bad shape");
    C2S_REQUIRE(d->kind == C2S_NORM_GROUP || d->kind == C2S_NORM_BATCH, "norm: bad kind");
    if (d->kind == C2S_NORM_GROUP) C2S_REQUIRE(d->groups > 0 && d->C % d->groups == 0, "norm: C %% groups != 0");
    return C2S_OK;
}

}  // namespace

// workspace: max(fwd: rows*segs*2, bwd: rows*segs*3 + rows*3 + rows*3)
extern "C" size_t c2s_norm_workspace_floats(const c2s_norm_desc* d) {
    if (!d) return 0;
    const size_t rows = (size_t)d->N * d->C;
    const size_t segs = n_segs(d->HW);
    return rows * segs * 3 + rows * 6 + 64;
}

extern "C" int c2s_norm_stats(const c2s_norm_desc* d, const float* x, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float* group_stats, float* row_ab,
                              float* workspace, size_t ws_floats, const int* valid, void* stream) {
    if (int rc = check_desc(d)) return rc;
    C2S_REQUIRE(x && gamma && beta && group_stats && row_ab, "norm_stats: null pointer");
    const int segs = n_segs(d->HW);
    const long rows = (long)d->N * d->C;
    const long nitems = rows * segs;
    hipStream_t st = (hipStream_t)stream;
    const bool eval_bn = d->kind == C2S_NORM_BATCH && !d->training;
    if (eval_bn) C2S_REQUIRE(running_mean && running_var, "norm_stats: eval BatchNorm needs running stats");
    if (!eval_bn) {
        C2S_REQUIRE(workspace && ws_floats >= (size_t)nitems * 2, "norm_stats: workspace too small");
        hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, workspace, valid, d->C, d->HW,
                           segs, nitems);
        C2S_CHECK_LAUNCH("row_stats");
    }
    const int ngroups = d->kind == C2S_NORM_BATCH ? d->C : d->N * d->groups;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(ngroups), dim3(64), 0, st, workspace, gamma, beta, running_mean,
                       running_var, group_stats, row_ab, valid, *d, segs);
    C2S_CHECK_LAUNCH("norm_finalize");
    return C2S_OK;
}

extern "C" int c2s_norm_apply(const c2s_norm_desc* d, const float* x, const float* row_ab, const float* residual,
                              float* y, int relu, const int* valid, float pad_value, void* stream) {
    if (int rc = check_desc(d)) return rc;
    C2S_REQUIRE(x && row_ab && y, "norm_apply: null pointer");
    const int segs = n_segs(d->HW);
    const long nitems = (long)d->N * d->C * segs;
    hipLaunchKernelGGL(norm_apply_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, (hipStream_t)stream, x, row_ab, residual,
                       y, valid, d->C, d->HW, segs, nitems, relu, pad_value);
    C2S_CHECK_LAUNCH("norm_apply");
    return C2S_OK;
}

extern "C" int c2s_norm_bwd(const c2s_norm_desc* d, const float* x, const float* g, const float* gamma,
                            const float* group_stats, const float* row_ab, int relu, float* gx, float* dgamma,
                            float* dbeta, float* dbias, float* workspace, size_t ws_floats, const int* valid,
                            void* stream) {
    if (int rc = check_desc(d)) return rc;
    C2S_REQUIRE(x && g && gamma && group_stats && row_ab && gx && workspace, "norm_bwd: null pointer");
    C2S_REQUIRE(ws_floats >= c2s_norm_workspace_floats(d), "norm_bwd: workspace too small");
    const int segs = n_segs(d->HW);
    const long rows = (long)d->N * d->C;
    const long nitems = rows * segs;
    float* part = workspace;
    float* rowsum = part + nitems * 3;
    float* rowk = rowsum + rows * 3;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(norm_bwd_sums_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, g, row_ab, group_stats, part,
                       valid, *d, segs, nitems, relu);
    C2S_CHECK_LAUNCH("norm_bwd_sums");
    const int ngroups = d->kind == C2S_NORM_BATCH ? d->C : d->N * d->groups;
    hipLaunchKernelGGL(norm_bwd_reduce_kernel, dim3(ngroups + d->C), dim3(64), 0, st, part, gamma, group_stats, rowk, dgamma,
                       dbeta, valid, *d, segs, ngroups);
    C2S_CHECK_LAUNCH("norm_bwd_reduce");
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, g, row_ab, rowk, gx, part, valid,
                       d->C, d->HW, segs, nitems, relu);
    C2S_CHECK_LAUNCH("norm_bwd_apply");
    if (dbias != nullptr) {
        hipLaunchKernelGGL(norm_bwd_dbias_kernel, dim3(d->C), dim3(64), 0, st, part, dbias, valid, d->N, d->C,
                           segs);
        C2S_CHECK_LAUNCH("norm_bwd_dbias");
    }
    return C2S_OK;
}
