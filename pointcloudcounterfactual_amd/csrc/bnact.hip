// BatchNorm1d + ReLU (+ channel-repeated residual) over [B,C,N] for gfx950 (MI355X): the tail of the reference's
// PointsConv block (src/module/layers.py:159-166: conv -> BatchNorm1d -> activation -> `+ x.repeat_interleave(r, 1)`),
// fused so that the 268 MB activation tensors of the PCGen decoder are read and written as few times as the
// arithmetic allows: forward = one statistics pass + one apply pass (PyTorch: BN 3 passes, ReLU 2, add 3);
// backward = one reduction pass + one apply pass (PyTorch: threshold 3, BN backward ~5).  Everything here is
// HBM-bound streaming with float4 accesses; statistics are accumulated in double and combined in a fixed order.
//   y[b,c,i] = max(0, (z[b,c,i] - mean[c]) * invstd[c] * gamma[c] + beta[c]) + res[b, c / r, i]
#include "pcc_common.hpp"

#include "pcc_neighbour.h"

namespace {

// Channel reductions are cut into `splits` sample ranges (grid = channels x splits, so that even a 16-channel layer
// fills the chip); every workgroup leaves two double partials and a small second kernel adds them in index order.
// per channel: mean and biased variance over the B*N samples.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(int b, int c, int n, int splits,
                                                                const float *__restrict__ z, double *__restrict__ part) {
    __shared__ double red[2][256];
    const int ch = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x;
    const int s0 = (int)((long long)b * sp / splits), s1 = (int)((long long)b * (sp + 1) / splits);
    double s = 0.0, ss = 0.0;
    const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(z) & 15) == 0);
    for (int smp = s0; smp < s1; smp++) {
        const float *row = z + ((size_t)smp * c + ch) * n;
        if (vec) {
            const float4 *r4 = reinterpret_cast<const float4 *>(row);
            for (int i = tid; i < n / 4; i += 256) {
                const float4 v = r4[i];
                s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
                ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
            }
        } else {
            for (int i = tid; i < n; i += 256) {
                const double v = row[i];
                s += v;
                ss += v * v;
            }
        }
    }
    red[0][tid] = s;
    red[1][tid] = ss;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            red[0][tid] += red[0][tid + off];
            red[1][tid] += red[1][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        part[((size_t)ch * splits + sp) * 2 + 0] = red[0][0];
        part[((size_t)ch * splits + sp) * 2 + 1] = red[1][0];
    }
}

// MODE 0: (sum, sum of squares) -> mean, biased variance;  MODE 1: (sum_g, sum_gx) -> as floats
template <int MODE>
__global__ __launch_bounds__(256) void bn_finalize_kernel(int c, int splits, double count, const double *__restrict__ part,
                                                           float *__restrict__ out0, float *__restrict__ out1) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    double a = 0.0, q = 0.0;
    for (int sp = 0; sp < splits; sp++) {
        a += part[((size_t)ch * splits + sp) * 2 + 0];
        q += part[((size_t)ch * splits + sp) * 2 + 1];
    }
    if (MODE == 0) {
        const double m = a / count;
        out0[ch] = (float)m;
        out1[ch] = (float)fmax(q / count - m * m, 0.0);
    } else {
        out0[ch] = (float)a;
        out1[ch] = (float)q;
    }
}

// y = relu(bn(z)) + res[b, ch / r, :]
__global__ __launch_bounds__(256) void bn_relu_res_fwd_kernel(int c, int n, const float *__restrict__ z,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ var, float eps,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ res, int res_c, int r,
                                                               float *__restrict__ y) {
    const int row = blockIdx.y;  // b * c + ch
    const int ch = row % c, smp = row / c;
    const float sc = gamma[ch] * __builtin_amdgcn_rsqf(var[ch] + eps);
    const float sh = beta[ch] - mean[ch] * sc;
    const float *zr = z + (size_t)row * n;
    float *yr = y + (size_t)row * n;
    const float *rr = res ? res + ((size_t)smp * res_c + ch / r) * n : nullptr;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n && (n % 4 == 0)) {
        const float4 v = *reinterpret_cast<const float4 *>(zr + i);
        float4 o = make_float4(fmaxf(__builtin_fmaf(v.x, sc, sh), 0.f), fmaxf(__builtin_fmaf(v.y, sc, sh), 0.f),
                               fmaxf(__builtin_fmaf(v.z, sc, sh), 0.f), fmaxf(__builtin_fmaf(v.w, sc, sh), 0.f));
        if (rr) {
            const float4 q = *reinterpret_cast<const float4 *>(rr + i);
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        *reinterpret_cast<float4 *>(yr + i) = o;
    } else {
        for (int j = i; j < min(i + 4, n); j++) yr[j] = fmaxf(__builtin_fmaf(zr[j], sc, sh), 0.f) + (rr ? rr[j] : 0.f);
    }
}

// per channel and sample range: sum_g = sum g*[pre > 0], sum_gx = sum g*[pre > 0]*xhat   (-> dbeta, dgamma)
__global__ __launch_bounds__(256) void bn_relu_bwd_partial_kernel(int b, int c, int n, int splits,
                                                                   const float *__restrict__ z,
                                                                   const float *__restrict__ mean,
                                                                   const float *__restrict__ var, float eps,
                                                                   const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta,
                                                                   const float *__restrict__ gy, double *__restrict__ part) {
    __shared__ double red[2][256];
    const int ch = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x;
    const int s0 = (int)((long long)b * sp / splits), s1 = (int)((long long)b * (sp + 1) / splits);
    const float mu = mean[ch], inv = __builtin_amdgcn_rsqf(var[ch] + eps);
    const float sc = gamma[ch] * inv, sh = beta[ch] - mu * sc;
    double s = 0.0, sx = 0.0;
    const bool vec = (n % 4 == 0) && (((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0);
    for (int smp = s0; smp < s1; smp++) {
        const float *zr = z + ((size_t)smp * c + ch) * n;
        const float *gr = gy + ((size_t)smp * c + ch) * n;
        if (vec) {
            for (int i = tid; i < n / 4; i += 256) {
                const float4 v = reinterpret_cast<const float4 *>(zr)[i];
                const float4 g4 = reinterpret_cast<const float4 *>(gr)[i];
                const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
                float ps = 0.f, px = 0.f;  // four terms in float, the running sums in double
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float g = __builtin_fmaf(vv[q], sc, sh) > 0.f ? gg[q] : 0.f;
                    ps += g;
                    px = __builtin_fmaf(g, (vv[q] - mu) * inv, px);
                }
                s += ps;
                sx += px;
            }
        } else {
            for (int i = tid; i < n; i += 256) {
                const float v = zr[i];
                const float g = __builtin_fmaf(v, sc, sh) > 0.f ? gr[i] : 0.f;
                s += g;
                sx += (double)g * ((v - mu) * inv);
            }
        }
    }
    red[0][tid] = s;
    red[1][tid] = sx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            red[0][tid] += red[0][tid + off];
            red[1][tid] += red[1][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        part[((size_t)ch * splits + sp) * 2 + 0] = red[0][0];
        part[((size_t)ch * splits + sp) * 2 + 1] = red[1][0];
    }
}

// dz = gamma * invstd * (g_act - sum_g / M - xhat * sum_gx / M)      (training mode)
// dz = gamma * invstd * g_act                                        (eval mode: statistics are constants)
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(int c, int n, float inv_count,
                                                                 const float *__restrict__ z,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ var, float eps,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta,
                                                                 const float *__restrict__ gy,
                                                                 const float *__restrict__ sum_g,
                                                                 const float *__restrict__ sum_gx,
                                                                 float *__restrict__ dz) {
    const int row = blockIdx.y;
    const int ch = row % c;
    const float mu = mean[ch], inv = __builtin_amdgcn_rsqf(var[ch] + eps);
    const float gsc = gamma[ch] * inv, sh = beta[ch] - mu * gsc;
    const float a = sum_g ? sum_g[ch] * inv_count : 0.f, bq = sum_gx ? sum_gx[ch] * inv_count : 0.f;
    const float *zr = z + (size_t)row * n;
    const float *gr = gy + (size_t)row * n;
    float *dr = dz + (size_t)row * n;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (i0 + 3 < n && n % 4 == 0) {
        const float4 v4 = *reinterpret_cast<const float4 *>(zr + i0);
        const float4 g4 = *reinterpret_cast<const float4 *>(gr + i0);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float g = __builtin_fmaf(vv[q], gsc, sh) > 0.f ? gg[q] : 0.f;
            o[q] = gsc * (g - a - ((vv[q] - mu) * inv) * bq);
        }
        *reinterpret_cast<float4 *>(dr + i0) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        for (int j = i0; j < min(i0 + 4, n); j++) {
            const float v = zr[j];
            const float g = __builtin_fmaf(v, gsc, sh) > 0.f ? gr[j] : 0.f;
            dr[j] = gsc * (g - a - ((v - mu) * inv) * bq);
        }
    }
}

// sample ranges per channel: ~2048 workgroups on the chip, at most one per sample
int pick_splits(int b, int c) {
    int s = (2048 + c - 1) / c;
    s = s < 1 ? 1 : s;
    return s > b ? b : s;
}

int chk(const char *who, int b, int c, int n) {
    if (b < 0 || c < 0 || n < 0 || (long long)b * c > 0x7fffffffLL || c > 65535 * 256) return pcc::invalid(who);
    return PCC_OK;
}

}  // namespace

extern "C" {

int pcc_bn_stats(int b, int c, int n, const float *z, float *mean, float *var, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = chk("bn_stats: bad size", b, c, n)) return rc;
    if (c == 0) return PCC_OK;
    if (b == 0 || n == 0) return pcc::invalid("bn_stats: no samples");
    if (!z || !mean || !var) return pcc::invalid("bn_stats: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int splits = pick_splits(b, c);
    double *part = nullptr;
    if (pcc::ws_malloc(reinterpret_cast<void **>(&part), (size_t)c * splits * 2 * sizeof(double), st) != hipSuccess) {
        (void)hipGetLastError();
        pcc::set_error(PCC_ENOMEM, "bn_stats: workspace allocation failed");
        return PCC_ENOMEM;
    }
    {
        pcc::ProfScope prof("bn_stats_partial_kernel", st);
        hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(c, splits), dim3(256), 0, st, b, c, n, splits, z, part);
    }
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(pcc::ceil_div(c, 256)), dim3(256), 0, st, c, splits, (double)b * n, part,
                       mean, var);
    (void)pcc::ws_free(part, st);
    return pcc::check_launch("bn_stats");
}

int pcc_bn_relu_res_fwd(int b, int c, int n, const float *z, const float *mean, const float *var, float eps,
                        const float *gamma, const float *beta, const float *res, int res_c, int r, float *y,
                        pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = chk("bn_relu_res_fwd: bad size", b, c, n)) return rc;
    if (b == 0 || c == 0 || n == 0) return PCC_OK;
    if (!z || !mean || !var || !gamma || !beta || !y) return pcc::invalid("bn_relu_res_fwd: null pointer");
    if (res && (r < 1 || (c - 1) / r >= res_c)) return pcc::invalid("bn_relu_res_fwd: residual has too few channels");
    if ((long long)b * c > 65535LL) return pcc::invalid("bn_relu_res_fwd: more than 65535 (sample, channel) rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    pcc::ProfScope prof("bn_relu_res_fwd_kernel", st);
    hipLaunchKernelGGL(bn_relu_res_fwd_kernel, dim3(pcc::ceil_div(n, 1024), b * c), dim3(256), 0, st, c, n, z, mean, var, eps,
                       gamma, beta, res, res_c, r, y);
    return pcc::check_launch("bn_relu_res_fwd");
}

int pcc_bn_relu_bwd(int b, int c, int n, const float *z, const float *mean, const float *var, float eps,
                    const float *gamma, const float *beta, const float *grad_y, int training, float *grad_z,
                    float *grad_gamma, float *grad_beta, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = chk("bn_relu_bwd: bad size", b, c, n)) return rc;
    if (b == 0 || c == 0 || n == 0) return PCC_OK;
    if (!z || !mean || !var || !gamma || !beta || !grad_y || !grad_z || !grad_gamma || !grad_beta)
        return pcc::invalid("bn_relu_bwd: null pointer");
    if ((long long)b * c > 65535LL) return pcc::invalid("bn_relu_bwd: more than 65535 (sample, channel) rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int splits = pick_splits(b, c);
    double *part = nullptr;
    if (pcc::ws_malloc(reinterpret_cast<void **>(&part), (size_t)c * splits * 2 * sizeof(double), st) != hipSuccess) {
        (void)hipGetLastError();
        pcc::set_error(PCC_ENOMEM, "bn_relu_bwd: workspace allocation failed");
        return PCC_ENOMEM;
    }
    {
        pcc::ProfScope prof("bn_relu_bwd_partial_kernel", st);
        hipLaunchKernelGGL(bn_relu_bwd_partial_kernel, dim3(c, splits), dim3(256), 0, st, b, c, n, splits, z, mean, var, eps,
                           gamma, beta, grad_y, part);
    }
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(pcc::ceil_div(c, 256)), dim3(256), 0, st, c, splits, 1.0, part, grad_beta,
                       grad_gamma);
    (void)pcc::ws_free(part, st);
    {
        pcc::ProfScope prof("bn_relu_bwd_apply_kernel", st);
        const float inv_count = 1.0f / ((float)b * (float)n);
        hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(pcc::ceil_div(n, 1024), b * c), dim3(256), 0, st, c, n, inv_count, z,
                           mean, var, eps, gamma, beta, grad_y, training ? grad_beta : nullptr,
                           training ? grad_gamma : nullptr, grad_z);
    }
    return pcc::check_launch("bn_relu_bwd");
}

}  // extern "C"
