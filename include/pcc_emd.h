/*
 * pcc_emd.h -- C ABI of the auction EMD in libpcc_structural.so (MI355X / gfx950).
 *
 * Replaces the reference's `emd_backend` pybind module (external/emd/src/emd.cpp:14-30 ->
 * emd_cuda_forward / emd_cuda_backward, external/emd/src/emd_cuda.cu:227-315).  The reference passes twelve
 * caller-allocated work tensors that its seven kernels per iteration communicate through; here a cluster of
 * persistent workgroups per sample (one when the batch already fills the chip) runs all iterations with the target
 * cloud and prices in LDS and workgroup / sample-local barriers instead of launches, so only the inputs and the two
 * outputs cross the boundary (shared state and scratch come from the stream-ordered pool).
 *
 *   xyz1[b,n,3], xyz2[b,n,3] float32 in [0,1]^3 ; dist[b,n] float32 ; assignment[b,n] int32.
 * Deterministic where the reference races (GetMax / Assign, emd_cuda.cu:189,205): bidders of an iteration =
 * points unassigned when it starts; lowest bidder index wins a tie within the 1e-6 window.
 */
#ifndef PCC_EMD_H
#define PCC_EMD_H

#include "pcc_structural.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* emd_cuda_forward (emd_cuda.cu:227-281).  Returns 0, or PCC_EINVAL for iters < 1, n < 1 or n > 8192
 * (the reference's own n % 1024 == 0 and b <= 512 limits, emd_module.py:23-30, are enforced by the Python
 * wrapper, not needed by the kernel). */
int pcc_auction_forward(int b, int n, const float *xyz1, const float *xyz2, float eps, int iters, float *dist,
                        int *assignment, pcc_stream_t stream);

/* Failure reporting of the cluster schedule (several workgroups per sample that meet at sample-local barriers): if a
 * barrier times out -- a workgroup of the sample was never scheduled -- the kernel poisons its outputs (dist = NaN,
 * unassigned points keep assignment -1) and raises a sticky per-device word.  The NEXT pcc_auction_forward /
 * pcc_auction_backward on that device returns PCC_EINVAL with a message instead of starting, and
 * pcc_auction_status() returns 1 (each of them clears the word).  Launches are asynchronous: synchronise the stream
 * before asking.  Cluster launches issued on different streams are ordered one after the other by an event, so two of
 * them never compete for residency.  (The reporting path is exercised through include/pcc_test_hooks.h.) */
int pcc_auction_status(void);

/* emd_cuda_backward (emd_cuda.cu:283-315): grad_xyz1[b,n,3] = 2 grad_dist (xyz1 - xyz2[assignment]); overwritten.
 * An assignment outside [0, n) contributes a zero gradient. */
int pcc_auction_backward(int b, int n, const float *xyz1, const float *xyz2, const float *grad_dist,
                         const int *assignment, float *grad_xyz1, pcc_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PCC_EMD_H */
