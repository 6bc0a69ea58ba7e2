/*
 * pcc_structural.h -- C ABI of libpcc_structural.so (MI355X / gfx950 structural-loss kernels).
 *
 * Drop-in boundary for the native half of the reference's `structural_losses` package: the five
 * launchers declared at external/pytorch_structural_losses/src/structural_loss.cpp:10-14 and
 * defined in nndistance.cu:125-128,149-154 and approxmatch.cu:299-326 of the reference.  Same names,
 * same argument order and meaning; the only change is `cudaStream_t` -> `hipStream_t` (passed as
 * `void *` so that the header needs no HIP include).  All pointers are device pointers to
 * contiguous row-major float32 / int32 arrays; inputs are borrowed and never written; outputs are
 * fully overwritten.  Calls enqueue work on `stream` and return without synchronising.
 *
 * Error behaviour: the reference's approxmatch/matchcost/matchcostgrad launchers throw
 * std::runtime_error("CUDA kernel failed : <code>") (approxmatch.cu:303-306) and its nndistance
 * launchers check nothing.  A C ABI cannot throw, so every `pcc_*` entry returns an int
 * (0 = success, otherwise the hipError_t of the failed launch / a PCC_E* code) and records a message
 * retrievable with pcc_last_error(); the reference-named void launchers record the same message and
 * the host-side binding raises RuntimeError("HIP kernel failed : <code>") from it.
 */
#ifndef PCC_STRUCTURAL_H
#define PCC_STRUCTURAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built -fvisibility=hidden; only this ABI is exported */

typedef void *pcc_stream_t; /* hipStream_t */

#define PCC_OK 0
#define PCC_EINVAL (-22)  /* bad sizes / null pointers */
#define PCC_ENOMEM (-12)  /* workspace allocation failed */

/* ---- library info ------------------------------------------------------------------------- */
const char *pcc_version(void);
/* Message of the last failed call on this thread ("" if none). */
const char *pcc_last_error(void);
/* Status of the last reference-named (void) launcher call on this thread; reset to 0 by each call. */
int pcc_last_status(void);

/* ---- per-kernel timing (measurement aid, off by default) ----------------------------------------
 * pcc_profile_enable(1): every kernel launch of this library is bracketed by two hipEvents recorded on the
 * launch stream.  pcc_profile_enable(2): only whole launch SEQUENCES are bracketed (one event before the first
 * launch, one after the last; name "am_phase_sequence" = the 19 back-to-back am_phase_kernel launches of one
 * approxmatch) -- no event sits between the kernels, so sequence / launches is the per-launch duration a kernel
 * trace reports.  pcc_profile_enable(0): off.  pcc_profile_read synchronises on the recorded events and returns the
 * average duration (microseconds) and the number of recorded scopes whose name starts with `kernel_prefix` since the
 * last enable / pcc_profile_reset().  Returns 0, or PCC_EINVAL if nothing matched.  Not thread-safe against
 * concurrent launches; meant for bench.py. */
void pcc_profile_enable(int on);
void pcc_profile_reset(void);
int pcc_profile_read(const char *kernel_prefix, double *avg_us, int *launches);

/* ---- Chamfer nearest neighbour ---------------------------------------------------------------
 * Replaces `nndistance` (reference nndistance.cu:125-128; declared structural_loss.cpp:13).
 *   xyz[b,n,3], xyz2[b,m,3] -> result[b,n] = min_k |xyz_j - xyz2_k|^2, result_i[b,n] = argmin
 *   (lowest index on ties), and the same with roles swapped -> result2[b,m], result2_i[b,m].
 * Distances are evaluated as fmaf(dz,dz, fmaf(dx,dx, dy*dy)) on differences (bit-exact vs oracle).
 * Non-finite coordinates follow the reference's loop literally (nndistance.cu:26-28,116: `k == 0 || d < best` inside
 * 512-candidate chunks, chunks merged with `result > best`): a NaN query or a NaN candidate 0 gives NaN / index 0; a NaN
 * candidate at index 512 c (c >= 1) hides candidates 512 c .. 512 c + 511 from every query; a NaN candidate elsewhere never
 * wins.  (pcc_chamfer_emd, like the approximate EMD itself, is specified for finite coordinates.) */
void nndistance(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                float *result2, int *result2_i, pcc_stream_t stream);
int pcc_nndistance(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                   float *result2, int *result2_i, pcc_stream_t stream);

/* Replaces `nndistancegrad` (reference nndistance.cu:149-154; declared structural_loss.cpp:14).
 *   grad_xyz1[b,n,3] = 2 g1_j (p1_j - p2_{idx1_j}) + sum_{k: idx2_k = j} 2 g2_k (p1_j - p2_k), and
 *   symmetrically grad_xyz2[b,m,3].  Outputs are overwritten (no prior memset needed). */
void nndistancegrad(int b, int n, const float *xyz1, int m, const float *xyz2, const float *grad_dist1,
                    const int *idx1, const float *grad_dist2, const int *idx2, float *grad_xyz1,
                    float *grad_xyz2, pcc_stream_t stream);
int pcc_nndistancegrad(int b, int n, const float *xyz1, int m, const float *xyz2, const float *grad_dist1,
                       const int *idx1, const float *grad_dist2, const int *idx2, float *grad_xyz1,
                       float *grad_xyz2, pcc_stream_t stream);

/* ---- Chamfer loss with the reduction fused (extension) ----------------------------------------------
 * What the reference's training path computes around the nearest-neighbour search
 * (src/train/metrics_and_losses.py:21-47): loss[b] = mean_j dist1[b,j] + mean_k dist2[b,k] (`mean` != 0,
 * pykeops_chamfer) or the plain sums (`mean` == 0, torch_chamfer's scale).  pcc_chamfer_loss = pcc_nndistance + one
 * fixed-order reduction; pcc_chamfer_loss_grad = pcc_nndistancegrad with grad_dist1[b,:] = grad_loss[b] (/ n),
 * grad_dist2[b,:] = grad_loss[b] (/ m) formed inside the kernel; grad_loss_stride is 1, or 0 when the upstream
 * gradient is one scalar expanded over the batch (what `loss.sum().backward()` hands down). */
int pcc_chamfer_loss(int b, int n, const float *xyz1, int m, const float *xyz2, int mean, float *loss, float *dist1,
                     int *idx1, float *dist2, int *idx2, pcc_stream_t stream);
int pcc_chamfer_loss_grad(int b, int n, const float *xyz1, int m, const float *xyz2, const int *idx1, const int *idx2,
                          const float *grad_loss, int grad_loss_stride, int mean, float *grad_xyz1, float *grad_xyz2,
                          pcc_stream_t stream);

/* Forward of the reference's ChamferEMD reconstruction loss (src/train/metrics_and_losses.py:70-79: Chamfer and
 * match_cost on the same pair of clouds) in one call: pcc_chamfer_loss + pcc_match_cost, same outputs, same bits.
 * emd_grad1 / emd_grad2: both or neither (NULL: cost only). */
int pcc_chamfer_emd(int b, int n, const float *xyz1, int m, const float *xyz2, int mean, float *chamfer_loss,
                    float *dist1, int *idx1, float *dist2, int *idx2, float *emd_cost, float *emd_grad1,
                    float *emd_grad2, pcc_stream_t stream);

/* Backward of the reference's ChamferEMD reconstruction loss (src/train/metrics_and_losses.py:70-79: Chamfer and
 * match_cost on the same pair of clouds) in one launch: pcc_chamfer_loss_grad of grad_chamfer[b] plus the unscaled
 * match_cost gradients emd_grad1[b,n,3] / emd_grad2[b,m,3] (as pcc_match_cost returns them) times grad_emd[b]
 * (NULL = 1).  Strides as above (0 = one scalar for the batch).  Bit-identical to the two backward passes followed by
 * autograd's gradient accumulation. */
int pcc_chamfer_emd_grad(int b, int n, const float *xyz1, int m, const float *xyz2, const int *idx1, const int *idx2,
                         const float *grad_chamfer, int grad_chamfer_stride, int mean, const float *emd_grad1,
                         const float *emd_grad2, const float *grad_emd, int grad_emd_stride, float *grad_xyz1,
                         float *grad_xyz2, pcc_stream_t stream);

/* ---- approximate EMD ---------------------------------------------------------------------------
 * Replaces `approxmatch` (reference approxmatch.cu:299-307; declared structural_loss.cpp:10).
 *   xyz1[b,n,3], xyz2[b,m,3] -> match[b,m,n] (query-major), temp[b,2(n+m)] =
 *   [remainL(n) | remainR(m) | ratioL(n) | ratioR(m)] after the last level.
 * The reference-named form allocates its per-level workspace with hipMallocAsync on `stream`;
 * pcc_approxmatch_ws takes a caller-provided workspace of pcc_approxmatch_workspace_bytes(b,n,m)
 * bytes instead (graph-capture friendly, no allocation in the call). */
void approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                 pcc_stream_t stream);
int pcc_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                    pcc_stream_t stream);
size_t pcc_approxmatch_workspace_bytes(int b, int n, int m);
int pcc_approxmatch_ws(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                       void *workspace, size_t workspace_bytes, pcc_stream_t stream);

/* approxmatch + matchcost in one call (what the Python-level match_cost forward needs,
 * reference structural_losses/match_cost.py:25-27): the pass that materialises `match` also
 * accumulates cost[b], so `match` is not re-read.  Same results as pcc_approxmatch + pcc_matchcost up to
 * float summation order. */
int pcc_approxmatch_cost(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                         float *cost, pcc_stream_t stream);

/* Replaces `matchcost` (reference approxmatch.cu:309-316; declared structural_loss.cpp:11).
 *   out[b] = sum_{k<m} sum_{j<n} match[b,k,j] * sqrt(|xyz1_j - xyz2_k|^2).  `match` is read-only
 *   (the reference declares it non-const but never writes it). */
void matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *out,
               pcc_stream_t stream);
int pcc_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out,
                  pcc_stream_t stream);

/* Replaces `matchcostgrad` (reference approxmatch.cu:318-326; declared structural_loss.cpp:12).
 *   grad1[b,l,:] = sum_k match[b,k,l] (p1_l - p2_k) rsqrt(max(d2,1e-20));
 *   grad2[b,k,:] = sum_j match[b,k,j] (p2_k - p1_j) rsqrt(max(d2,1e-20)). */
void matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                   float *grad1, float *grad2, pcc_stream_t stream);
int pcc_matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                      float *grad1, float *grad2, pcc_stream_t stream);
/* The same with the upstream gradient folded in: grad1[b] *= grad_cost[b], grad2[b] *= grad_cost[b] -- what the
 * Python wrapper does with two extra elementwise passes (match_cost.py:41-42).  grad_cost == NULL means 1. */
int pcc_matchcostgrad_scaled(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                             const float *grad_cost, float *grad1, float *grad2, pcc_stream_t stream);

/* ---- match_cost without the match tensor (extension) ---------------------------------------------
 * The Python-level match_cost (reference structural_losses/match_cost.py:11-50) needs cost[b] in forward and
 * grad * grad_cost[b] in backward; `match` only travels from ApproxMatch to MatchCost / MatchCostGrad and the
 * gradient treats it as a constant (approxmatch.cu:229-291).  pcc_match_cost evaluates every match element in
 * registers (same level order and rounding as pcc_approxmatch) and feeds it straight into the cost and gradient
 * sums: no 4*b*n*m-byte tensor is written or read.
 *   cost[b]                     = what pcc_approxmatch + pcc_matchcost return (float summation order differs);
 *   grad1[b,n,3], grad2[b,m,3]  = what pcc_matchcostgrad_scaled returns; pass both or neither (NULL: cost only);
 *   grad_cost[b] or NULL (= 1).
 * Non-finite coordinates: a NaN propagates through the distances as in the reference; a sample with an infinite
 * coordinate returns NaN cost and NaN gradients (the reference's 0 * sqrt(inf), approxmatch.cu:207,247-248), the other
 * samples of the batch are unaffected. */
int pcc_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *grad_cost, float *cost,
                   float *grad1, float *grad2, pcc_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PCC_STRUCTURAL_H */
