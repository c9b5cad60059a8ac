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
    float s = 0.f, m2 = 0.f, mean;
    if (len == SEG) {
        // full segment: all eight float4 of the lane are requested at once (the rolled loop below keeps one load per wave in
        // flight: 3.2 TB/s on the 537 MB layers) and stay in registers for the second pass; same summation order
        f32x4 v[SEG / 256];
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k) v[k] = *reinterpret_cast<const f32x4*>(xr + lane * 4 + 256 * k);
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        s = wave_sum(s);
        mean = s / (float)len;
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k) {
            const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            m2 += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        if ((len & 3) == 0) {
            for (int i = lane * 4; i < len; i += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
                s += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int i = lane; i < len; i += 64) s += xr[i];
        }
        s = wave_sum(s);
        mean = s / (float)len;
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
    }
    m2 = wave_sum(m2);
    if (lane == 0) {
        part[item * 2 + 0] = mean;
        part[item * 2 + 1] = m2;
    }
}

// Group statistics from the (mean, M2) partials of equal-or-known length, Chan's formula in double, one wave.
// GROUP: group = (n, g): rows (n, g*cpg .. g*cpg+cpg-1).   BATCH: group = c: rows (n, c) for all valid n.
// Every wave of the apply pass recomputes the statistics of its own group (a few hundred partials, L2-resident) -- that
// is cheaper than a separate one-wave-per-group launch between the two activation passes.  `unbiased` returns the
// variance for the running-statistics update.
__device__ __forceinline__ void group_mean_rstd(const float* __restrict__ part, const int* __restrict__ valid,
                                                const c2s_norm_desc& d, int segs, int grp, int lane, float& mean_f,
                                                float& rstd_f, float& unbiased) {
    const int L = seg_len(d.HW);
    const bool batch = d.kind == C2S_NORM_BATCH;
    const int cpg = batch ? 1 : d.C / d.groups;
    const int n_rows = batch ? d.N : cpg;         // rows in this group
    const int items = n_rows * segs;
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
    unbiased = (float)(cnt > 1 ? m2 / (cnt - 1.0) : var);
}

// ---------------------------------------------------------------- forward apply
// y = relu?((x - mean) * rstd * gamma + beta) (+ residual); also leaves what the backward needs: row_ab[row] =
// (gamma*rstd, beta, mean) from the wave of segment 0, group_stats[grp] = (mean, rstd) and the BatchNorm running
// statistics from the wave of the group's first row.
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ rmean, float* __restrict__ rvar,
                                                         long long* __restrict__ nbt,
                                                         float* __restrict__ gstats, float* __restrict__ row_ab,
                                                         const float* __restrict__ res, float* __restrict__ y,
                                                         const int* __restrict__ valid, c2s_norm_desc d, int segs,
                                                         long nitems, int relu, float pad_value) {
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63;
    const int C = d.C, HW = d.HW;
    const long row = item / segs;
    const int seg = (int)(item % segs);
    const int n = (int)(row / C), c = (int)(row % C);
    const int L = seg_len(HW);
    const int beg = seg * L;
    const int len = (HW - beg) < L ? (HW - beg) : L;
    const size_t base = (size_t)row * HW + beg;
    const bool ok = valid == nullptr || valid[n] != 0;
    const bool batch = d.kind == C2S_NORM_BATCH;
    const int cpg = batch ? 1 : C / d.groups;
    const int grp = batch ? c : n * d.groups + c / cpg;
    const bool first_row = seg == 0 && (batch ? n == 0 : c % cpg == 0);
    float mu = 0.f, rstd = 0.f;
    if (batch && !d.training) {
        mu = rmean[grp];
        rstd = rsqrtf(rvar[grp] + d.eps);
    } else if (batch || ok) {
        float unb;
        group_mean_rstd(part, valid, d, segs, grp, lane, mu, rstd, unb);
        if (batch && first_row && lane == 0 && rmean != nullptr) {
            rmean[grp] = (1.f - d.momentum) * rmean[grp] + d.momentum * mu;
            rvar[grp] = (1.f - d.momentum) * rvar[grp] + d.momentum * unb;
        }
        if (batch && item == 0 && lane == 0 && nbt != nullptr) *nbt += 1;     // num_batches_tracked
    }
    if (first_row && lane == 0) { gstats[grp * 2] = mu; gstats[grp * 2 + 1] = rstd; }   // zeros for a padded GroupNorm frame
    if (!ok) {
        if (seg == 0 && lane == 0) { row_ab[row * 3] = 0.f; row_ab[row * 3 + 1] = 0.f; row_ab[row * 3 + 2] = 0.f; }
        for (int i = lane; i < len; i += 64) y[base + i] = pad_value;
        return;
    }
    const float a = gamma[c] * rstd, b = beta[c];
    if (seg == 0 && lane == 0) { row_ab[row * 3] = a; row_ab[row * 3 + 1] = b; row_ab[row * 3 + 2] = mu; }
    if (len == SEG) {
        // full segment: every load of the lane is requested before the first use (the rolled loop below keeps one round per
        // wave in flight)
        f32x4 v[SEG / 256], r[SEG / 256];
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k) v[k] = *reinterpret_cast<const f32x4*>(x + base + lane * 4 + 256 * k);
        if (res != nullptr) {
#pragma unroll
            for (int k = 0; k < SEG / 256; ++k) r[k] = *reinterpret_cast<const f32x4*>(res + base + lane * 4 + 256 * k);
        }
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[k][e] - mu) * a + b;
                if (relu) t = fmaxf(t, 0.f);
                if (res != nullptr) t += r[k][e];
                o[e] = t;
            }
            *reinterpret_cast<f32x4*>(y + base + lane * 4 + 256 * k) = o;
        }
    } else if ((len & 3) == 0) {
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
// part[item] = (sum g', sum g'*xhat, -) with g' = g * [a*x+b > 0] (relu) ; xhat = (x-mean)*rstd; the third slot is
// filled by the apply pass (sum of dx, for the producing convolution's bias gradient)
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
        if (lane == 0) { part[item * 3] = 0.f; part[item * 3 + 1] = 0.f; }
        return;
    }
    const int L = seg_len(d.HW);
    const int beg = seg * L;
    const int len = (d.HW - beg) < L ? (d.HW - beg) : L;
    const size_t base = (size_t)row * d.HW + beg;
    const int grp = d.kind == C2S_NORM_BATCH ? c : n * d.groups + c / (d.C / d.groups);
    const float mean = gstats[grp * 2], rstd = gstats[grp * 2 + 1];
    const float a = row_ab[row * 3], b = row_ab[row * 3 + 1];
    float s1 = 0.f, s2 = 0.f;
    if (len == SEG) {
        f32x4 xa[SEG / 256], ga[SEG / 256];                       // all 16 loads of the lane in flight at once
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k) {
            xa[k] = *reinterpret_cast<const f32x4*>(x + base + lane * 4 + 256 * k);
            ga[k] = *reinterpret_cast<const f32x4*>(g + base + lane * 4 + 256 * k);
        }
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xx = xa[k][e];
                const float gg = (!relu || (xx - mean) * a + b > 0.f) ? ga[k][e] : 0.f;
                s1 += gg;
                s2 += gg * ((xx - mean) * rstd);
            }
    } else if ((len & 3) == 0) {
        for (int i = lane * 4; i < len; i += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + base + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xx = xv[k];
                const float gg = (!relu || (xx - mean) * a + b > 0.f) ? gv[k] : 0.f;
                s1 += gg;
                s2 += gg * ((xx - mean) * rstd);
            }
        }
    } else {
        for (int i = lane; i < len; i += 64) {
            const float xx = x[base + i];
            const float gg = (!relu || (xx - mean) * a + b > 0.f) ? g[base + i] : 0.f;
            s1 += gg;
            s2 += gg * ((xx - mean) * rstd);
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { part[item * 3] = s1; part[item * 3 + 1] = s2; }
}

// The per-(row, segment) partials are reduced where they are needed, in double and in a fixed order:
//   * the apply pass: every wave derives the coefficients of its own row, dx = k1*g' + k2*(x-mean) + k3, from the sums of
//     its group (a few hundred L2-resident partials; cheaper than a launch between the two activation passes);
//   * norm_bwd_params_kernel, after the apply pass: dgamma, dbeta and the bias gradient of one channel per wave.
__device__ __forceinline__ void row_sums(const float* __restrict__ part, long row, int segs, double& s1, double& s2) {
    s1 = 0.0; s2 = 0.0;
    for (int s = 0; s < segs; ++s) {
        s1 += part[(row * segs + s) * 3];
        s2 += part[(row * segs + s) * 3 + 1];
    }
}

__global__ __launch_bounds__(64) void norm_bwd_params_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dbias,
                                                             const int* __restrict__ valid, int N, int C, int segs) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double dg = 0, db = 0;
    for (int n = lane; n < N; n += 64) {
        if (valid != nullptr && valid[n] == 0) continue;
        double s1, s2;
        row_sums(part, (long)n * C + c, segs, s1, s2);
        dg += (double)(float)s2;
        db += (double)(float)s1;
    }
    for (int o = 32; o > 0; o >>= 1) { dg += __shfl_xor(dg, o, 64); db += __shfl_xor(db, o, 64); }
    if (lane == 0) {
        if (dgamma != nullptr) dgamma[c] = (float)dg;
        if (dbeta != nullptr) dbeta[c] = (float)db;
    }
    if (dbias == nullptr) return;
    // gradient of the producing convolution's bias: per-channel sum of dx, lanes over (frame, segment)
    double s = 0;
    for (int it = lane; it < N * segs; it += 64) {
        const int n = it / segs, k = it - n * segs;
        if (valid != nullptr && valid[n] == 0) continue;
        s += part[(((long)n * C + c) * segs + k) * 3 + 2];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) dbias[c] = (float)s;
}

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ row_ab,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ gstats, float* __restrict__ gx,
                                                             float* __restrict__ part,
                                                             const int* __restrict__ valid, c2s_norm_desc d, int segs,
                                                             long nitems, int relu) {
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63;
    const int C = d.C, HW = d.HW;
    const long row = item / segs;
    const int seg = (int)(item % segs);
    const int L = seg_len(HW);
    const int beg = seg * L;
    const int len = (HW - beg) < L ? (HW - beg) : L;
    const size_t base = (size_t)row * HW + beg;
    if (valid != nullptr && valid[row / C] == 0) {
        for (int i = lane; i < len; i += 64) gx[base + i] = 0.f;
        if (lane == 0) part[item * 3 + 2] = 0.f;
        return;
    }
    // coefficients of this row from the sums of its group
    const bool batch = d.kind == C2S_NORM_BATCH;
    const int cpg = batch ? 1 : C / d.groups;
    const int n_rows = batch ? d.N : cpg;
    const int cc = (int)(row % C), nn = (int)(row / C);
    const int grp = batch ? cc : nn * d.groups + cc / cpg;
    const float rstd = gstats[grp * 2 + 1];
    float k1 = rstd * gamma[cc], k2 = 0.f, k3 = 0.f;
    if (!(batch && !d.training)) {                      // eval-mode BatchNorm: statistics are constants
        double A = 0.0, Bv = 0.0, cnt = 0.0;
        for (int r = lane; r < n_rows; r += 64) {
            const long rr = batch ? (long)r * C + grp : (long)nn * C + (grp % d.groups) * cpg + r;
            if (valid != nullptr && valid[rr / C] == 0) continue;
            const float gm = gamma[rr % C];
            double s1, s2;
            row_sums(part, rr, segs, s1, s2);
            A += (double)gm * (double)(float)s1;
            Bv += (double)gm * (double)(float)s2;
            cnt += HW;
        }
        for (int o = 32; o > 0; o >>= 1) { A += __shfl_xor(A, o, 64); Bv += __shfl_xor(Bv, o, 64); cnt += __shfl_xor(cnt, o, 64); }
        const double m = cnt > 0 ? cnt : 1.0;
        k2 = (float)(-(double)rstd * rstd * Bv / m);
        k3 = (float)(-(double)rstd * A / m);
    }
    const float a = row_ab[row * 3], b = row_ab[row * 3 + 1], mu = row_ab[row * 3 + 2];
    float sdx = 0.f;
    if (len == SEG) {
        f32x4 xa[SEG / 256], ga[SEG / 256];                       // all 16 loads of the lane in flight at once
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k) {
            xa[k] = *reinterpret_cast<const f32x4*>(x + base + lane * 4 + 256 * k);
            ga[k] = *reinterpret_cast<const f32x4*>(g + base + lane * 4 + 256 * k);
        }
#pragma unroll
        for (int k = 0; k < SEG / 256; ++k) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xc = xa[k][e] - mu;
                const float gg = (!relu || xc * a + b > 0.f) ? ga[k][e] : 0.f;
                o[e] = k1 * gg + k2 * xc + k3;
                sdx += o[e];
            }
            *reinterpret_cast<f32x4*>(gx + base + lane * 4 + 256 * k) = o;
        }
    } else if ((len & 3) == 0) {
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
    if (lane == 0) part[item * 3 + 2] = sdx;
}

int check_desc(const c2s_norm_desc* d) {
    C2S_REQUIRE(d && d->N > 0 && d->C > 0 && d->HW > 0, "norm: bad shape");
    C2S_REQUIRE(d->kind == C2S_NORM_GROUP || d->kind == C2S_NORM_BATCH, "norm: bad kind");
    if (d->kind == C2S_NORM_GROUP) C2S_REQUIRE(d->groups > 0 && d->C % d->groups == 0, "norm: C %% groups != 0");
    return C2S_OK;
}

}  // namespace

// workspace: max(fwd: rows*segs*2, bwd: rows*segs*3)
extern "C" size_t c2s_norm_workspace_floats(const c2s_norm_desc* d) {
    if (!d) return 0;
    const size_t rows = (size_t)d->N * d->C;
    const size_t segs = n_segs(d->HW);
    return rows * segs * 3 + rows * 6 + 64;
}

extern "C" int c2s_norm_fwd(const c2s_norm_desc* d, const float* x, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, long long* num_batches_tracked,
                            float* group_stats, float* row_ab,
                            const float* residual, float* y, int relu, float* workspace, size_t ws_floats,
                            const int* valid, float pad_value, void* stream) {
    if (int rc = check_desc(d)) return rc;
    C2S_REQUIRE(x && gamma && beta && group_stats && row_ab && y, "norm_fwd: null pointer");
    const int segs = n_segs(d->HW);
    const long rows = (long)d->N * d->C;
    const long nitems = rows * segs;
    hipStream_t st = (hipStream_t)stream;
    const bool eval_bn = d->kind == C2S_NORM_BATCH && !d->training;
    if (eval_bn) C2S_REQUIRE(running_mean && running_var, "norm_fwd: eval BatchNorm needs running stats");
    if (!eval_bn) {
        C2S_REQUIRE(workspace && ws_floats >= (size_t)nitems * 2, "norm_fwd: workspace too small");
        hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, workspace, valid, d->C, d->HW,
                           segs, nitems);
        C2S_CHECK_LAUNCH("row_stats");
    }
    hipLaunchKernelGGL(norm_apply_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, workspace, gamma, beta, running_mean,
                       running_var, num_batches_tracked, group_stats, row_ab, residual, y, valid, *d, segs, nitems, relu, pad_value);
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
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(norm_bwd_sums_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, g, row_ab, group_stats, part,
                       valid, *d, segs, nitems, relu);
    C2S_CHECK_LAUNCH("norm_bwd_sums");
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(cdiv(nitems, 4)), dim3(256), 0, st, x, g, row_ab, gamma, group_stats, gx,
                       part, valid, *d, segs, nitems, relu);
    C2S_CHECK_LAUNCH("norm_bwd_apply");
    if (dgamma != nullptr || dbeta != nullptr || dbias != nullptr) {
        hipLaunchKernelGGL(norm_bwd_params_kernel, dim3(d->C), dim3(64), 0, st, part, dgamma, dbeta, dbias, valid, d->N,
                           d->C, segs);
        C2S_CHECK_LAUNCH("norm_bwd_params");
    }
    return C2S_OK;
}

extern "C" int c2s_norm_bwd_params(const c2s_norm_desc* d, const float* workspace, float* dgamma, float* dbeta,
                                   float* dbias, const int* valid, void* stream) {
    if (int rc = check_desc(d)) return rc;
    C2S_REQUIRE(workspace && (dgamma || dbeta || dbias), "norm_bwd_params: null pointer");
    hipLaunchKernelGGL(norm_bwd_params_kernel, dim3(d->C), dim3(64), 0, (hipStream_t)stream, workspace, dgamma, dbeta, dbias,
                       valid, d->N, d->C, n_segs(d->HW));
    C2S_CHECK_LAUNCH("norm_bwd_params");
    return C2S_OK;
}
