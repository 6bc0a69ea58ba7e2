/*
 * pcc_neighbour.h -- C ABI of the kNN-graph / neighbour-gather / max-pool primitives in
 * libpcc_structural.so (MI355X / gfx950).
 *
 * The reference has no native boundary for these: they are Python functions in
 * src/utils/neighbour_ops.py that call PyKeOps (GPU) or torch (CPU).  Each entry cites the function it
 * replaces.  Tensors are contiguous row-major device arrays in the reference's layouts:
 *   x[b, c, n] float32 (channels-major), indices[b, n, k] int64.
 * All entries enqueue on `stream` (hipStream_t as void*), never synchronise, and return 0 or an error code
 * (message via pcc_last_error(), include/pcc_structural.h).
 */
#ifndef PCC_NEIGHBOUR_H
#define PCC_NEIGHBOUR_H

#include <stddef.h>
#include <stdint.h>

#include "pcc_structural.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* knn / pykeops_knn (neighbour_ops.py:63-82): for every point the k nearest points of the same cloud in
 * feature space, ascending distance (the point itself first), ties by ascending index.
 *   c <= 3 : exact difference form sum_c (x_j - x_i)^2 on the f32 VALU (the GPU reference's formula, :35-40)
 *   c >= 4 : expanded form |x_i|^2 + |x_j|^2 - 2 x_i.x_j with the inner product on the f32 MFMA pipe
 *            (the reference CPU path's formula, self_square_distance :53-60)
 * Requires 1 <= k <= min(n, 32). */
int pcc_knn(int b, int c, int n, int k, const float *x, int64_t *indices, pcc_stream_t stream);

/* get_neighbours (neighbour_ops.py:85-94): out[b,c,n,j] = x[b,c,indices[b,n,j]]. */
int pcc_gather_neighbours(int b, int c, int n, int k, const float *x, const int64_t *indices, float *out,
                          pcc_stream_t stream);
/* backward of the gather: grad_x[b,c,t] = sum over (n,j) with indices[b,n,j]==t of grad_out[b,c,n,j].
 * grad_x is overwritten.  Accumulated in per-workgroup LDS bins (ds_add_f32): like the torch scatter_add the
 * reference's gather backward runs, the float summation order is not fixed. */
int pcc_gather_neighbours_bwd(int b, int c, int n, int k, const int64_t *indices, const float *grad_out,
                              float *grad_x, pcc_stream_t stream);

/* get_graph_features (neighbour_ops.py:113-119): out[b, 0:c, n, j] = x[b,:,indices[b,n,j]] - x[b,:,n],
 * out[b, c:2c, n, j] = x[b,:,n]. */
int pcc_graph_features(int b, int c, int n, int k, const float *x, const int64_t *indices, float *out,
                       pcc_stream_t stream);
/* backward: grad_x[b,c,t] = sum_{(n,j): idx=t} g[b,c,n,j] + sum_j (g[b,c+C,t,j] - g[b,c,t,j]). */
int pcc_graph_features_bwd(int b, int c, int n, int k, const int64_t *indices, const float *grad_out,
                           float *grad_x, pcc_stream_t stream);

/* graph_max_pooling (neighbour_ops.py:106-110): out[b,c,n] = max_j x[b,c,indices[b,n,j]];
 * argmax[b,c,n] (int32, the winning j, first on ties as torch.max) is written when non-null. */
int pcc_graph_max_pool(int b, int c, int n, int k, const float *x, const int64_t *indices, float *out,
                       int32_t *argmax, pcc_stream_t stream);
int pcc_graph_max_pool_bwd(int b, int c, int n, int k, const int64_t *indices, const int32_t *argmax,
                           const float *grad_out, float *grad_x, pcc_stream_t stream);

/* Building blocks of the fused EdgeConv front-end (pointcloudcounterfactual_amd/edgeconv.py): because
 * the 1x1 convolution is linear, W.[x_j - x_i ; x_i] = Wa.x_j + (Wb - Wa).x_i, so everything the EdgeConv block
 * (get_graph_features -> conv2d -> BatchNorm2d -> LeakyReLU -> max over k; src/module/encoders.py:50-53,
 * layers.py:159-203) needs from the [B,2C,N,k] edge tensor can be had from [B,C',N] tensors:
 *   pcc_neighbour_sum            out[b,c,n] = sum_j y[b,c,indices[b,n,j]]          (BatchNorm statistics)
 *   pcc_neighbour_sum_bwd        grad_y[b,c,t] = sum_{(n,j): idx=t} grad_out[b,c,n]
 *   pcc_neighbour_minmax_target  tsel[b,0,c,n] / tsel[b,1,c,n] = the neighbour (point index) with the largest /
 *                                smallest y among the k neighbours of n (first on ties) -- the edge that survives
 *                                max-over-k for a positive / negative BatchNorm scale. */
int pcc_neighbour_sum(int b, int c, int n, int k, const float *y, const int64_t *indices, float *out,
                      pcc_stream_t stream);
int pcc_neighbour_sum_bwd(int b, int c, int n, int k, const int64_t *indices, const float *grad_out,
                          float *grad_y, pcc_stream_t stream);
int pcc_neighbour_minmax_target(int b, int c, int n, int k, const float *y, const int64_t *indices, int64_t *tsel,
                                pcc_stream_t stream);

/* Encoder / classifier global pooling (src/module/encoders.py:58,90; classifier.py:63-64):
 * out_max[b,c] = max_n x[b,c,n] with argmax[b,c] (int32, first maximum), out_mean[b,c] = mean_n (either
 * output pointer may be null). */
int pcc_global_pool(int b, int c, int n, const float *x, float *out_max, int32_t *argmax, float *out_mean,
                    pcc_stream_t stream);

/* ---- generic-dimension pairwise reductions (the KeOps reductions of the reference outside the 3-D paths) ----------
 * D[b,i,j] = sum_c (p[b,i,c] - q[b,j,c])^2 over p[b,np,d], q[b,nq,d] (pykeops_square_distance, neighbour_ops.py:35-40),
 * difference form, channels accumulated in order with fma; never materialised.
 * pcc_pair_argmin: idx[b,i] = argmin_j D[b,i,j] (lowest index on ties), dist[b,i] = the minimum (may be NULL) --
 *   VectorQuantizer.quantize's `dist.argmin(axis=2)` (src/module/quantize.py:26-28); swap p and q for axis=1.
 * pcc_pair_sqdist_sum: out[b,i] = sum_j D[b,i,j] -- `dist.sum(1)` of quantize.py:31 (with p = codebook rows, q = the
 *   single query); pcc_pair_sqdist_sum_bwd: its gradients, grad_p[b,i,:] = 2 g[b,i] sum_j (p_i - q_j),
 *   grad_q[b,j,:] = -2 sum_i g[b,i] (p_i - q_j); either output may be NULL. */
int pcc_pair_argmin(int b, int np, int nq, int d, const float *p, const float *q, int64_t *idx, float *dist,
                    pcc_stream_t stream);
int pcc_pair_sqdist_sum(int b, int np, int nq, int d, const float *p, const float *q, float *out, pcc_stream_t stream);
int pcc_pair_sqdist_sum_bwd(int b, int np, int nq, int d, const float *p, const float *q, const float *grad_out,
                            float *grad_p, float *grad_q, pcc_stream_t stream);

/* ---- BatchNorm1d + ReLU (+ channel-repeated residual) over [b,c,n] --------------------------------------------------
 * The tail of the reference's PointsConv block (src/module/layers.py:159-166: conv -> BatchNorm1d -> activation ->
 * `+ x.repeat_interleave(r, 1)[:, :c]`), fused into streaming passes: pcc_bn_stats (training: per-channel mean and
 * biased variance over b*n, accumulated in double), pcc_bn_relu_res_fwd
 *   y[b,ch,i] = max(0, (z - mean[ch]) * rsqrt(var[ch] + eps) * gamma[ch] + beta[ch]) + res[b, ch / r, i]   (res may be NULL)
 * and pcc_bn_relu_bwd (grad_z, grad_gamma[c], grad_beta[c]; `training` = the statistics depend on z).  The gradient of
 * the residual operand is grad_y summed over each group of r channels (left to the caller).  b*c <= 65535. */
int pcc_bn_stats(int b, int c, int n, const float *z, float *mean, float *var, pcc_stream_t stream);
int pcc_bn_relu_res_fwd(int b, int c, int n, const float *z, const float *mean, const float *var, float eps,
                        const float *gamma, const float *beta, const float *res, int res_c, int r, float *y,
                        pcc_stream_t stream);
int pcc_bn_relu_bwd(int b, int c, int n, const float *z, const float *mean, const float *var, float eps,
                    const float *gamma, const float *beta, const float *grad_y, int training, float *grad_z,
                    float *grad_gamma, float *grad_beta, pcc_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PCC_NEIGHBOUR_H */
