// Generic-dimension pairwise reductions for gfx950 (MI355X): the three KeOps reductions the reference takes of
// the lazy matrix D[b,i,j] = sum_c (x[b,i,c] - y[b,j,c])^2 (src/utils/neighbour_ops.py:35-40) outside the 3-D
// Chamfer / kNN paths, which have their own kernels:
//   * argmin over j / over i   -- vector-quantiser nearest code (src/module/quantize.py:26-28) and the
//                                 pykeops_chamfer indices when the clouds are not 3-D
//   * sum over j / over i      -- quantize.py:31 (`dist.sum(1)`, differentiable: the w-autoencoder loss reads it)
// These are small problems (8192 batches of 1 x 16 codes x 4 channels at the reference's vqvae.yaml sizes):
// one thread per output row, the reduced axis streamed from L2, difference form accumulated with explicit fma in
// channel order (d = fma(t, t, d)), lowest index on ties.
#include "pcc_common.hpp"

#include "pcc_neighbour.h"

namespace {

// rows: P[b][np][d] (one thread per row), reduced axis: Q[b][nq][d]
template <bool ARGMIN>
__global__ __launch_bounds__(256) void pair_reduce_kernel(int b, int np, int nq, int d, const float *__restrict__ P,
                                                           const float *__restrict__ Q, int64_t *__restrict__ out_idx,
                                                           float *__restrict__ out_val) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= (long long)b * np) return;
    const int smp = (int)(row / np);
    const float *p = P + (size_t)row * d;
    const float *q = Q + (size_t)smp * nq * d;
    float best = __builtin_inff(), sum = 0.f;
    int best_j = 0;
    for (int j = 0; j < nq; j++) {
        float acc = 0.f;
        for (int c = 0; c < d; c++) {
            const float t = p[c] - q[(size_t)j * d + c];
            acc = __builtin_fmaf(t, t, acc);
        }
        if (ARGMIN) {
            const bool lt = acc < best;  // strict: the lowest index wins ties
            best = lt ? acc : best;
            best_j = lt ? j : best_j;
        } else {
            sum += acc;
        }
    }
    if (ARGMIN) {
        out_idx[row] = best_j;
        if (out_val) out_val[row] = best;
    } else {
        out_val[row] = sum;
    }
}

// out[b][i] = sum_j D[b,i,j] with upstream gradient g[b][i]:
//   grad_P[b][i][:] = 2 g[b][i] sum_j (p_i - q_j)          (one thread per row i)
//   grad_Q[b][j][:] = -2 sum_i g[b][i] (p_i - q_j)         (one thread per reduced point j)
// Both loops run in index order: deterministic.
__global__ __launch_bounds__(256) void pair_sum_bwd_rows_kernel(int b, int np, int nq, int d, const float *__restrict__ P,
                                                                 const float *__restrict__ Q, const float *__restrict__ g,
                                                                 float *__restrict__ gP) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (row, channel)
    if (e >= (long long)b * np * d) return;
    const long long row = e / d;
    const int c = (int)(e - row * d);
    const int smp = (int)(row / np);
    const float pc = P[e];
    const float *q = Q + (size_t)smp * nq * d + c;
    float acc = 0.f;
    for (int j = 0; j < nq; j++) acc += pc - q[(size_t)j * d];
    gP[e] = 2.f * g[row] * acc;
}

__global__ __launch_bounds__(256) void pair_sum_bwd_cols_kernel(int b, int np, int nq, int d, const float *__restrict__ P,
                                                                 const float *__restrict__ Q, const float *__restrict__ g,
                                                                 float *__restrict__ gQ) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (reduced point, channel)
    if (e >= (long long)b * nq * d) return;
    const long long col = e / d;
    const int c = (int)(e - col * d);
    const int smp = (int)(col / nq);
    const float qc = Q[e];
    const float *p = P + (size_t)smp * np * d + c;
    const float *gr = g + (size_t)smp * np;
    float acc = 0.f;
    for (int i = 0; i < np; i++) acc += gr[i] * (p[(size_t)i * d] - qc);
    gQ[e] = -2.f * acc;
}

int check(const char *who, int b, int np, int nq, int d) {
    if (b < 0 || np < 0 || nq < 0 || d < 0) return pcc::invalid(who);
    if ((long long)b * np * (long long)std::max(d, 1) > 0x7fffffffLL * 256LL) return pcc::invalid(who);
    return PCC_OK;
}

unsigned blocks(long long items) { return (unsigned)((items + 255) / 256); }

}  // namespace

extern "C" {

int pcc_pair_argmin(int b, int np, int nq, int d, const float *p, const float *q, int64_t *idx, float *dist,
                    pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("pair_argmin: bad size", b, np, nq, d)) return rc;
    if (b == 0 || np == 0) return PCC_OK;
    if (nq == 0) return pcc::invalid("pair_argmin: nothing to reduce over");
    if (!p || !q || !idx) return pcc::invalid("pair_argmin: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        pcc::ProfScope prof("pair_reduce_kernel<argmin>", st);
        hipLaunchKernelGGL((pair_reduce_kernel<true>), dim3(blocks((long long)b * np)), dim3(256), 0, st, b, np, nq, d, p, q,
                           idx, dist);
    }
    return pcc::check_launch("pair_argmin");
}

int pcc_pair_sqdist_sum(int b, int np, int nq, int d, const float *p, const float *q, float *out, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("pair_sqdist_sum: bad size", b, np, nq, d)) return rc;
    if (b == 0 || np == 0) return PCC_OK;
    if (!p || !out || (nq && !q)) return pcc::invalid("pair_sqdist_sum: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        pcc::ProfScope prof("pair_reduce_kernel<sum>", st);
        hipLaunchKernelGGL((pair_reduce_kernel<false>), dim3(blocks((long long)b * np)), dim3(256), 0, st, b, np, nq, d, p, q,
                           nullptr, out);
    }
    return pcc::check_launch("pair_sqdist_sum");
}

int pcc_pair_sqdist_sum_bwd(int b, int np, int nq, int d, const float *p, const float *q, const float *grad_out,
                            float *grad_p, float *grad_q, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("pair_sqdist_sum_bwd: bad size", b, np, nq, d)) return rc;
    if (b == 0 || d == 0) return PCC_OK;
    if (!p || !q || !grad_out) return pcc::invalid("pair_sqdist_sum_bwd: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (grad_p && np) {
        pcc::ProfScope prof("pair_sum_bwd_rows_kernel", st);
        hipLaunchKernelGGL(pair_sum_bwd_rows_kernel, dim3(blocks((long long)b * np * d)), dim3(256), 0, st, b, np, nq, d, p, q,
                           grad_out, grad_p);
    }
    if (grad_q && nq) {
        pcc::ProfScope prof("pair_sum_bwd_cols_kernel", st);
        hipLaunchKernelGGL(pair_sum_bwd_cols_kernel, dim3(blocks((long long)b * nq * d)), dim3(256), 0, st, b, np, nq, d, p, q,
                           grad_out, grad_q);
    }
    return pcc::check_launch("pair_sqdist_sum_bwd");
}

}  // extern "C"
