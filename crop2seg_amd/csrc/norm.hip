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

// ================================================================= one-pass forms
// The two-pass forms above read every activation twice per direction (statistics / sums, then apply): 8 tensor passes per
// layer and step, the largest HBM consumer of a train step (profiles/r02_step_traffic.csv: 12.1 of 30.4 GB).  The one-pass
// forms read once: a wave keeps its (row, segment) in registers (32 floats per lane), publishes its partial sums, and the
// waves of a normalisation group meet through memory before they apply from registers -- 5 passes instead of 8.
//
// Hand-off = data-tagged granules (MI355X_MICROARCH.md, inter-workgroup visibility: 8-byte {data, tag} words written by ONE
// sc1 store and read by sc1 loads need no ordering, no counter and no fence): wave w of a group stores its two partial sums
// as two granules {value, epoch} at gran[first granule of the group + w]; wave 0 of every workgroup sweeps the granules of
// its group (lane = granule) until all carry this launch's epoch, merges them in the fixed order of the two-pass form
// (double: bit-identical results), and hands the two group values to the other waves through LDS.  The epoch word only
// grows (the last workgroup of a launch advances it), so granules of earlier launches -- of any shape, the area is shared
// by all layers of a stream -- never match and nothing is ever cleared: the area only has to be all zero before its first
// use.  The grid is persistent (at most the workgroups the chip holds at once, positions dealt round-robin), so every
// workgroup a sweep waits for is resident; a sweep gives up after 2^17 rounds (~0.2 s), raises the error word (header
// word 3) instead of hanging the queue and makes the group's outputs NaN; launches on an area whose error word is set do not
// wait at all (incomplete groups come out NaN).  The host side (Workspace.check_sync, called by TrainStep wherever it
// synchronises anyway) raises on the error word, re-zeroes the area and switches the process to the two-pass kernels.
struct SyncView {
    unsigned* hdr;                 // [0] finished workgroups, [2] epoch, [3] error
    unsigned long long* gran;      // [nitems][2] {float bits | epoch << 32}
};
__host__ __device__ inline size_t sync_bytes(long nitems) { return 64 + (size_t)nitems * 16; }
__device__ __forceinline__ SyncView sync_view(void* base) {
    SyncView v;
    v.hdr = reinterpret_cast<unsigned*>(base);
    v.gran = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(base) + 64);
    return v;
}
constexpr int OP_MAXG = 8;         // granule pairs per lane of the sweeping wave: groups of up to 512 waves

__device__ __forceinline__ void onepass_begin(const SyncView& sv, unsigned* s_hdr, unsigned& epoch, bool& poisoned) {
    if (threadIdx.x == 0) {
        s_hdr[0] = __hip_atomic_load(&sv.hdr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_hdr[1] = __hip_atomic_load(&sv.hdr[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    epoch = s_hdr[0] + 1u;
    poisoned = s_hdr[1] != 0u;    // an earlier sweep on this area gave up: its state is unknown, nothing waits on it any more
}
// the last workgroup to finish advances the epoch (every workgroup has read it by then)
__device__ __forceinline__ void onepass_end(const SyncView& sv, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned f = __hip_atomic_fetch_add(&sv.hdr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f + 1u == gridDim.x) {
            __hip_atomic_store(&sv.hdr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sv.hdr[2], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ void onepass_publish(const SyncView& sv, long v, unsigned epoch, float p0, float p1, int lane) {
    if (lane == 0) {
        const unsigned long long e = (unsigned long long)epoch << 32;
        __hip_atomic_store(&sv.gran[v * 2 + 0], e | __float_as_uint(p0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sv.gran[v * 2 + 1], e | __float_as_uint(p1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// sweep of the wpg granule pairs of the group that starts at granule g0 (one wave): on return pa[w], pb[w] (LDS) hold the two
// partial sums of wave w of the group.  Rolled over 64-granule rounds (two rounds of loads in flight): a sweep needs a
// handful of registers next to the activation the wave keeps.
__device__ __forceinline__ void onepass_sweep(const SyncView& sv, long g0, int wpg, unsigned epoch, int lane, bool poisoned,
                                              float* __restrict__ pa, float* __restrict__ pb) {
    for (int polls = poisoned ? (1 << 17) : 0;; ++polls) {
        bool hit = true;
#pragma unroll 2
        for (int w = lane; w < wpg; w += 64) {
            const unsigned long long ga = __hip_atomic_load(&sv.gran[(g0 + w) * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long gb = __hip_atomic_load(&sv.gran[(g0 + w) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hit = hit && (unsigned)(ga >> 32) == epoch && (unsigned)(gb >> 32) == epoch;
            pa[w] = __uint_as_float((unsigned)ga);
            pb[w] = __uint_as_float((unsigned)gb);
        }
        if (__all(hit)) break;
        if (polls > (1 << 17)) {
            // gave up (or the area is poisoned and this group is incomplete): raise the error word and hand NaN partials to the
            // merge, so that the group's outputs -- and with them the loss -- are NaN instead of plausible numbers from stale sums
            if (lane == 0) __hip_atomic_store(&sv.hdr[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int w = lane; w < wpg; w += 64) { pa[w] = __uint_as_float(0x7fc00000u); pb[w] = __uint_as_float(0x7fc00000u); }
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave reads its own LDS writes next
}

// Position -> (group, position in the group, row, segment): the waves of a group are consecutive positions (GROUP: rows of a
// group are consecutive anyway; BATCH: the rows (n, c) of channel c for all n).
struct OnepassItem { int grp, w, seg; long row; };
__device__ __forceinline__ OnepassItem onepass_item(long v, const c2s_norm_desc& d, int segs) {
    const bool batch = d.kind == C2S_NORM_BATCH;
    const int cpg = batch ? 1 : d.C / d.groups;
    const int n_rows = batch ? d.N : cpg;
    const int ipg = n_rows * segs;
    OnepassItem it;
    it.grp = (int)(v / ipg);
    it.w = (int)(v - (long)it.grp * ipg);
    const int r = it.w / segs;
    it.seg = it.w - r * segs;
    it.row = batch ? (long)r * d.C + it.grp : (long)it.grp * cpg + r;
    return it;
}

// NK = float4 per lane: every segment is exactly NK*256 floats; a group is one wave or a whole number of workgroups
// (the host checks both), so the four waves of a workgroup always work on one group.
// Software pipeline: the loads of the workgroup's NEXT position are issued before the group of the current one meets, so a
// wave always has a segment in flight while it waits (two register sets A / B, loop unrolled by two).
template <int NK>
struct OpSlot {
    f32x4 xv[NK];
    long v, row;
    size_t base;
    int grp, w, seg;
    bool live, ok;
};

template <int NK, bool RES>
__global__ __launch_bounds__(256) void norm_onepass_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ rmean,
                                                               float* __restrict__ rvar, long long* __restrict__ nbt,
                                                               float* __restrict__ gstats, float* __restrict__ row_ab,
                                                               const float* __restrict__ res, float* __restrict__ y,
                                                               const int* __restrict__ valid, c2s_norm_desc d, int segs,
                                                               long nitems, int relu, float pad_value, void* sync) {
    __shared__ unsigned s_hdr[2];
    __shared__ float s_stat[2][2];
    __shared__ float s_part[2][OP_MAXG * 64];
    const SyncView sv = sync_view(sync);
    unsigned epoch;
    bool poisoned;
    onepass_begin(sv, s_hdr, epoch, poisoned);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = d.C, HW = d.HW;
    const bool batch = d.kind == C2S_NORM_BATCH;
    const int cpg = batch ? 1 : C / d.groups;
    const int wpg = (batch ? d.N : cpg) * segs;
    constexpr int L = NK * 256;
    const long quads = (nitems + 3) >> 2;

    // position q of this workgroup -> slot: item, flags, loads of the segment in flight
    auto fetch = [&](OpSlot<NK>& s, long q) {
        s.v = q * 4 + wave;
        s.live = s.v < nitems;                                   // only a single-wave-group launch has a ragged last quad
        const OnepassItem it = onepass_item(s.live ? s.v : nitems - 1, d, segs);
        s.row = it.row; s.grp = it.grp; s.w = it.w; s.seg = it.seg;
        s.base = (size_t)it.row * HW + (size_t)it.seg * L;
        s.ok = valid == nullptr || valid[(int)(it.row / C)] != 0;       // BATCH launches come without flags (host check)
        if (s.live && s.ok) {
#pragma unroll
            for (int k = 0; k < NK; ++k) s.xv[k] = *reinterpret_cast<const f32x4*>(x + s.base + lane * 4 + 256 * k);
        }
    };
    // partial sums of the slot's segment, published for the group
    auto reduce_publish = [&](OpSlot<NK>& s, float& pmean, float& m2) {
        pmean = 0.f; m2 = 0.f;
        if (!(s.live && s.ok)) return;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) t += (s.xv[k].x + s.xv[k].y) + (s.xv[k].z + s.xv[k].w);
        t = wave_sum(t);
        pmean = t / (float)L;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float a = s.xv[k].x - pmean, b = s.xv[k].y - pmean, cc = s.xv[k].z - pmean, dd = s.xv[k].w - pmean;
            m2 += (a * a + b * b) + (cc * cc + dd * dd);
        }
        m2 = wave_sum(m2);
        if (wpg > 1) onepass_publish(sv, s.v, epoch, pmean, m2, lane);
    };
    // the group meets (wave 0 sweeps), then the slot's segment is normalised from registers and stored
    auto finish = [&](OpSlot<NK>& s, float pmean, float m2, int par) {
        const int c = (int)(s.row % C);
        if (!s.ok) {                                             // uniform over the workgroup when it meets (wpg % 4 == 0)
            if (s.live) {
                if (s.seg == 0 && lane == 0) {
                    row_ab[s.row * 3] = 0.f; row_ab[s.row * 3 + 1] = 0.f; row_ab[s.row * 3 + 2] = 0.f;
                    if (c % cpg == 0) { gstats[s.grp * 2] = 0.f; gstats[s.grp * 2 + 1] = 0.f; }
                }
                f32x4 pv = {pad_value, pad_value, pad_value, pad_value};
#pragma unroll
                for (int k = 0; k < NK; ++k) *reinterpret_cast<f32x4*>(y + s.base + lane * 4 + 256 * k) = pv;
            }
            return;
        }
        f32x4 rv[RES ? NK : 1];
        if constexpr (RES) {
            if (s.live) {
#pragma unroll
                for (int k = 0; k < NK; ++k) rv[k] = *reinterpret_cast<const f32x4*>(res + s.base + lane * 4 + 256 * k);
            }
        }
        float mu, rstd, unb;
        bool leader;                                             // the wave that leaves the group's statistics behind
        if (wpg > 1) {
            if (wave == 0) {
                onepass_sweep(sv, s.v - s.w, wpg, epoch, lane, poisoned, s_part[0], s_part[1]);
                // group_mean_rstd of the two-pass form on the swept partials (same order, same arithmetic)
                double cnt = 0.0, sum = 0.0;
                for (int w = lane; w < wpg; w += 64) { cnt += L; sum += (double)s_part[0][w] * L; }
                for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); sum += __shfl_xor(sum, o, 64); }
                const double mean = cnt > 0 ? sum / cnt : 0.0;
                double mm = 0.0;
                for (int w = lane; w < wpg; w += 64) { const double dm = (double)s_part[0][w] - mean; mm += (double)s_part[1][w] + dm * dm * L; }
                for (int o = 32; o > 0; o >>= 1) mm += __shfl_xor(mm, o, 64);
                const double var = cnt > 0 ? mm / cnt : 0.0;
                mu = (float)mean;
                rstd = (float)(1.0 / sqrt(var + (double)d.eps));
                unb = (float)(cnt > 1 ? mm / (cnt - 1.0) : var);
                if (lane == 0) { s_stat[par][0] = mu; s_stat[par][1] = rstd; }
            }
            __syncthreads();
            if (wave != 0) { mu = s_stat[par][0]; rstd = s_stat[par][1]; unb = 0.f; }
            leader = s.w == 0;                                   // wave 0 of the group's first workgroup
        } else {
            const double var = (double)m2 / L;
            mu = pmean;
            rstd = (float)(1.0 / sqrt(var + (double)d.eps));
            unb = (float)(L > 1 ? (double)m2 / (L - 1.0) : var);
            leader = s.live;
        }
        if (!s.live) return;
        if (leader && lane == 0) {
            gstats[s.grp * 2] = mu; gstats[s.grp * 2 + 1] = rstd;
            if (batch && rmean != nullptr) {
                rmean[s.grp] = (1.f - d.momentum) * rmean[s.grp] + d.momentum * mu;
                rvar[s.grp] = (1.f - d.momentum) * rvar[s.grp] + d.momentum * unb;
            }
            if (batch && s.grp == 0 && nbt != nullptr) *nbt += 1;
        }
        const float a = gamma[c] * rstd, b = beta[c];
        if (s.seg == 0 && lane == 0) { row_ab[s.row * 3] = a; row_ab[s.row * 3 + 1] = b; row_ab[s.row * 3 + 2] = mu; }
        const bool failed = mu != mu;                            // the sweep gave up (NaN partials): fmaxf would turn NaN into 0
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (s.xv[k][e] - mu) * a + b;
                if (relu) t = fmaxf(t, 0.f);
                if constexpr (RES) t += rv[k][e];
                o[e] = failed ? mu : t;
            }
            *reinterpret_cast<f32x4*>(y + s.base + lane * 4 + 256 * k) = o;
        }
    };

    OpSlot<NK> A, B;
    long q = blockIdx.x;
    if (q < quads) fetch(A, q);
    while (q < quads) {
        float pm, m2;
        reduce_publish(A, pm, m2);
        long qn = q + gridDim.x;
        if (qn < quads) fetch(B, qn);
        finish(A, pm, m2, 0);
        q = qn;
        if (q >= quads) break;
        reduce_publish(B, pm, m2);
        qn = q + gridDim.x;
        if (qn < quads) fetch(A, qn);
        finish(B, pm, m2, 1);
        q = qn;
    }
    onepass_end(sv, epoch);
}

template <int NK>
struct OpSlotB {
    f32x4 xa[NK], ga[NK];
    long v, row;
    size_t base;
    int grp, w, seg;
    bool live, ok;
};

// backward: same pipeline (the next position's x and g are in flight while the group of the current one meets)
template <int NK>
__global__ __launch_bounds__(256) void norm_onepass_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                               const float* __restrict__ row_ab,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ gstats, float* __restrict__ gx,
                                                               float* __restrict__ part, const int* __restrict__ valid,
                                                               c2s_norm_desc d, int segs, long nitems, int relu,
                                                               void* sync) {
    __shared__ unsigned s_hdr[2];
    __shared__ float s_stat[2][2];
    __shared__ float s_part[2][OP_MAXG * 64];
    const SyncView sv = sync_view(sync);
    unsigned epoch;
    bool poisoned;
    onepass_begin(sv, s_hdr, epoch, poisoned);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = d.C, HW = d.HW;
    const bool batch = d.kind == C2S_NORM_BATCH;
    const int cpg = batch ? 1 : C / d.groups;
    const int n_rows = batch ? d.N : cpg;
    const int wpg = n_rows * segs;
    constexpr int L = NK * 256;
    const long quads = (nitems + 3) >> 2;

    auto fetch = [&](OpSlotB<NK>& s, long q) {
        s.v = q * 4 + wave;
        s.live = s.v < nitems;
        const OnepassItem it = onepass_item(s.live ? s.v : nitems - 1, d, segs);
        s.row = it.row; s.grp = it.grp; s.w = it.w; s.seg = it.seg;
        s.base = (size_t)it.row * HW + (size_t)it.seg * L;
        s.ok = valid == nullptr || valid[(int)(it.row / C)] != 0;
        if (s.live && s.ok) {
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                s.xa[k] = *reinterpret_cast<const f32x4*>(x + s.base + lane * 4 + 256 * k);
                s.ga[k] = *reinterpret_cast<const f32x4*>(g + s.base + lane * 4 + 256 * k);
            }
        }
    };
    // masked gradient (kept in ga), partial sums s1 = sum g', s2 = sum g' xhat, published for the group
    auto reduce_publish = [&](OpSlotB<NK>& s, float& s1, float& s2) {
        s1 = 0.f; s2 = 0.f;
        if (!(s.live && s.ok)) return;
        const float mean = gstats[s.grp * 2], rstd = gstats[s.grp * 2 + 1];
        const float a = row_ab[s.row * 3], b = row_ab[s.row * 3 + 1];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xx = s.xa[k][e];
                const float gg = (!relu || (xx - mean) * a + b > 0.f) ? s.ga[k][e] : 0.f;
                s.ga[k][e] = gg;
                s1 += gg;
                s2 += gg * ((xx - mean) * rstd);
            }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (wpg > 1) onepass_publish(sv, s.v, epoch, s1, s2, lane);
        const long item = s.row * segs + s.seg;
        if (lane == 0) { part[item * 3] = s1; part[item * 3 + 1] = s2; }       // for norm_bwd_params_kernel (a later launch)
    };
    auto finish = [&](OpSlotB<NK>& s, float s1, float s2, int par) {
        const int cc = (int)(s.row % C);
        const long item = s.row * segs + s.seg;
        if (!s.ok) {                                             // uniform over the workgroup when it meets
            if (s.live) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NK; ++k) *reinterpret_cast<f32x4*>(gx + s.base + lane * 4 + 256 * k) = z;
                if (lane == 0) { part[item * 3] = 0.f; part[item * 3 + 1] = 0.f; part[item * 3 + 2] = 0.f; }
            }
            return;
        }
        const float rstd = gstats[s.grp * 2 + 1];
        float k2, k3;
        if (wpg > 1) {
            if (wave == 0) {
                onepass_sweep(sv, s.v - s.w, wpg, epoch, lane, poisoned, s_part[0], s_part[1]);
                // the per-row sums over the segments run in the order of row_sums(): lanes over rows
                double A = 0.0, Bv = 0.0, cnt = 0.0;
                for (int r = lane; r < n_rows; r += 64) {
                    const long rr = batch ? (long)r * C + s.grp : (long)s.grp * cpg + r;
                    const float gm = gamma[rr % C];
                    double t1 = 0.0, t2 = 0.0;
                    for (int sg = 0; sg < segs; ++sg) { t1 += s_part[0][r * segs + sg]; t2 += s_part[1][r * segs + sg]; }
                    A += (double)gm * (double)(float)t1;
                    Bv += (double)gm * (double)(float)t2;
                    cnt += HW;
                }
                for (int o = 32; o > 0; o >>= 1) { A += __shfl_xor(A, o, 64); Bv += __shfl_xor(Bv, o, 64); cnt += __shfl_xor(cnt, o, 64); }
                const double m = cnt > 0 ? cnt : 1.0;
                k2 = (float)(-(double)rstd * rstd * Bv / m);
                k3 = (float)(-(double)rstd * A / m);
                if (lane == 0) { s_stat[par][0] = k2; s_stat[par][1] = k3; }
            }
            __syncthreads();
            if (wave != 0) { k2 = s_stat[par][0]; k3 = s_stat[par][1]; }
        } else {
            const double gm = (double)gamma[cc];
            k2 = (float)(-(double)rstd * rstd * (gm * (double)s2) / (double)HW);
            k3 = (float)(-(double)rstd * (gm * (double)s1) / (double)HW);
        }
        if (!s.live) return;
        const float mu = row_ab[s.row * 3 + 2];
        const float k1 = rstd * gamma[cc];
        float sdx = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = k1 * s.ga[k][e] + k2 * (s.xa[k][e] - mu) + k3;
                sdx += o[e];
            }
            *reinterpret_cast<f32x4*>(gx + s.base + lane * 4 + 256 * k) = o;
        }
        sdx = wave_sum(sdx);
        if (lane == 0) part[item * 3 + 2] = sdx;
    };

    OpSlotB<NK> A, B;
    long q = blockIdx.x;
    if (q < quads) fetch(A, q);
    while (q < quads) {
        float s1, s2;
        reduce_publish(A, s1, s2);
        long qn = q + gridDim.x;
        if (qn < quads) fetch(B, qn);
        finish(A, s1, s2, 0);
        q = qn;
        if (q >= quads) break;
        reduce_publish(B, s1, s2);
        qn = q + gridDim.x;
        if (qn < quads) fetch(A, qn);
        finish(B, s1, s2, 1);
        q = qn;
    }
    onepass_end(sv, epoch);
}

// shapes the one-pass forms take: every segment full and a power-of-two number of float4 per lane; a group is one wave or a
// whole number of workgroups, at most 512 waves; BatchNorm only with batch statistics and without frame flags (its groups
// span all frames)
__host__ inline int onepass_nk(const c2s_norm_desc* d, const int* valid) {
    const int L = seg_len(d->HW);
    if (d->HW % L != 0 || L % 256 != 0) return 0;
    const int nk = L / 256;
    if (nk != 1 && nk != 2 && nk != 4 && nk != 8) return 0;
    const bool batch = d->kind == C2S_NORM_BATCH;
    if (batch && (!d->training || valid != nullptr)) return 0;
    const long wpg = (long)(batch ? d->N : d->C / d->groups) * n_segs(d->HW);
    if (wpg > 64 * OP_MAXG || (wpg != 1 && wpg % 4 != 0)) return 0;
    return nk;
}
__host__ inline long norm_items(const c2s_norm_desc* d) { return (long)d->N * d->C * n_segs(d->HW); }

// persistent grid: at most the workgroups the chip holds at once (every workgroup a sweep waits for must be resident)
template <typename K>
int onepass_grid(K kernel, long quads) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    const long cap = (long)c2s_cus() * per_cu;
    return (int)(quads < cap ? quads : cap);
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

// ---------------------------------------------------------------- one-pass entry points
extern "C" size_t c2s_norm_onepass_sync_bytes(const c2s_norm_desc* d, int has_valid) {
    if (!d || check_desc(d) != C2S_OK) return 0;
    static const int dummy = 0;
    if (onepass_nk(d, has_valid ? &dummy : nullptr) == 0) return 0;
    return sync_bytes(norm_items(d));
}

extern "C" int c2s_norm_fwd_onepass(const c2s_norm_desc* d, const float* x, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, long long* num_batches_tracked,
                                    float* group_stats, float* row_ab, const float* residual, float* y, int relu,
                                    const int* valid, float pad_value, void* sync, size_t sync_nbytes, void* stream) {
    if (int rc = check_desc(d)) return rc;
    C2S_REQUIRE(x && gamma && beta && group_stats && row_ab && y && sync, "norm_fwd_onepass: null pointer");
    const int nk = onepass_nk(d, valid);
    C2S_REQUIRE(nk != 0, "norm_fwd_onepass: shape not taken by the one-pass form (c2s_norm_onepass_sync_bytes() == 0): use c2s_norm_fwd");
    const int segs = n_segs(d->HW);
    const long nitems = norm_items(d);
    C2S_REQUIRE(sync_nbytes >= sync_bytes(nitems), "norm_fwd_onepass: sync area too small");
    hipStream_t st = (hipStream_t)stream;
    const long quads = (nitems + 3) / 4;
#define C2S_OP_FWD(NK_, R_)                                                                                              \
    if (nk == NK_ && (residual != nullptr) == R_) {                                                                    \
        static thread_local int grid_cap = 0;                                                                          \
        if (grid_cap == 0) grid_cap = onepass_grid((norm_onepass_fwd_kernel<NK_, R_>), 1L << 40);                      \
        hipLaunchKernelGGL((norm_onepass_fwd_kernel<NK_, R_>), dim3(quads < grid_cap ? quads : grid_cap), dim3(256), 0, st, x, \
                           gamma, beta, running_mean, running_var, num_batches_tracked, group_stats, row_ab, residual, y, \
                           valid, *d, segs, nitems, relu, pad_value, sync);                                             \
    }
    C2S_OP_FWD(8, false) C2S_OP_FWD(4, false) C2S_OP_FWD(2, false) C2S_OP_FWD(1, false)
    C2S_OP_FWD(8, true) C2S_OP_FWD(4, true) C2S_OP_FWD(2, true) C2S_OP_FWD(1, true)
#undef C2S_OP_FWD
    C2S_CHECK_LAUNCH("norm_onepass_fwd");
    return C2S_OK;
}

extern "C" int c2s_norm_bwd_onepass(const c2s_norm_desc* d, const float* x, const float* g, const float* gamma,
                                    const float* group_stats, const float* row_ab, int relu, float* gx, float* dgamma,
                                    float* dbeta, float* dbias, float* workspace, size_t ws_floats, const int* valid,
                                    void* sync, size_t sync_nbytes, void* stream) {
    if (int rc = check_desc(d)) return rc;
    C2S_REQUIRE(x && g && gamma && group_stats && row_ab && gx && workspace && sync, "norm_bwd_onepass: null pointer");
    C2S_REQUIRE(ws_floats >= c2s_norm_workspace_floats(d), "norm_bwd_onepass: workspace too small");
    const int nk = onepass_nk(d, valid);
    C2S_REQUIRE(nk != 0, "norm_bwd_onepass: shape not taken by the one-pass form (c2s_norm_onepass_sync_bytes() == 0): use c2s_norm_bwd");
    const int segs = n_segs(d->HW);
    const long nitems = norm_items(d);
    C2S_REQUIRE(sync_nbytes >= sync_bytes(nitems), "norm_bwd_onepass: sync area too small");
    hipStream_t st = (hipStream_t)stream;
    const long quads = (nitems + 3) / 4;
#define C2S_OP_BWD(NK_)                                                                                                 \
    if (nk == NK_) {                                                                                                    \
        static thread_local int grid_cap = 0;                                                                           \
        if (grid_cap == 0) grid_cap = onepass_grid(norm_onepass_bwd_kernel<NK_>, 1L << 40);                             \
        hipLaunchKernelGGL(norm_onepass_bwd_kernel<NK_>, dim3(quads < grid_cap ? quads : grid_cap), dim3(256), 0, st, x, \
                           g, row_ab, gamma, group_stats, gx, workspace, valid, *d, segs, nitems, relu, sync);           \
    }
    C2S_OP_BWD(8) C2S_OP_BWD(4) C2S_OP_BWD(2) C2S_OP_BWD(1)
#undef C2S_OP_BWD
    C2S_CHECK_LAUNCH("norm_onepass_bwd");
    if (dgamma != nullptr || dbeta != nullptr || dbias != nullptr) {
        hipLaunchKernelGGL(norm_bwd_params_kernel, dim3(d->C), dim3(64), 0, st, workspace, dgamma, dbeta, dbias, valid, d->N,
                           d->C, segs);
        C2S_CHECK_LAUNCH("norm_bwd_params");
    }
    return C2S_OK;
}
