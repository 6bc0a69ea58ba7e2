// Auction EMD for gfx950 (MI355X), wave64.
//
// Replaces external/emd/src/emd_cuda.cu (7 kernels x iters launches, all state in global memory, racy GetMax /
// Assign).  One persistent 1024-thread workgroup per sample runs every iteration with the auction state in
// LDS (targets SoA, prices, max increments; for n <= 4096 also the bidder-side arrays): phases are separated
// by workgroup barriers instead of kernel boundaries, compaction/winner selection use LDS integer atomics,
// and every tie is resolved deterministically (lowest bidder index), so two runs agree bit for bit.
// Bid values follow emd_cuda.cu:145 literally: `3.0 - sqrtf(d2) - price` is DOUBLE arithmetic rounded once
// to float; sqrtf is correctly rounded (-fno-fast-math).
#include "pcc_common.hpp"
#include "pcc_emd.h"

namespace {

using pcc::sq3;

struct Cand {
    float best, better;
    int best_i;
};

// first-maximum rule (emd_cuda.cu:146-153 / :166-171) as an order-independent merge
__device__ __forceinline__ Cand merge(const Cand &a, const Cand &b) {
    Cand r;
    const bool take_b = (b.best > a.best) || (b.best == a.best && (unsigned)b.best_i < (unsigned)a.best_i);
    r.best = take_b ? b.best : a.best;
    r.best_i = take_b ? b.best_i : a.best_i;
    const float loser = take_b ? a.best : b.best;
    r.better = fmaxf(loser, fmaxf(a.better, b.better));
    return r;
}

__global__ __launch_bounds__(1024) void auction_kernel(int n, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, float eps, int iters,
                                                        float *__restrict__ dist, int *__restrict__ assignment,
                                                        int *__restrict__ scratch, int state_in_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *sx = reinterpret_cast<float *>(smem);
    float *sy = sx + n, *sz = sy + n, *price = sz + n;
    int *max_inc = reinterpret_cast<int *>(price + n);  // float bits; values written are > 0 or -1e9
    int *cnt = max_inc + n;                              // [4]
    int *st = state_in_lds ? cnt + 4 : scratch + (size_t)blockIdx.x * 5 * n;
    int *unass = st, *inv = st + n, *bid = st + 2 * n, *win = st + 3 * n;
    float *inc = reinterpret_cast<float *>(st + 4 * n);

    const int tid = threadIdx.x, T = 1024;
    const int smp = blockIdx.x;
    const float *p1 = xyz1 + (size_t)smp * n * 3, *p2 = xyz2 + (size_t)smp * n * 3;
    int *ass = assignment + (size_t)smp * n;

    for (int k = tid; k < n; k += T) {
        sx[k] = p2[k * 3 + 0];
        sy[k] = p2[k * 3 + 1];
        sz[k] = p2[k * 3 + 2];
        price[k] = 0.f;
        max_inc[k] = 0;  // emd_module.py:41 zeros
        ass[k] = -1;
        inv[k] = -1;
        bid[k] = 0;
        inc[k] = 0.f;
    }
    __syncthreads();

    for (int it = 0; it < iters; it++) {
        const bool last = it == iters - 1;
        if (tid == 0) cnt[0] = 0;
        __syncthreads();
        for (int j = tid; j < n; j += T)
            if (ass[j] == -1) unass[atomicAdd(&cnt[0], 1)] = j;  // order irrelevant to the result
        __syncthreads();
        const int nu = cnt[0];
        if (nu == 0) break;  // everything assigned: the remaining iterations (and the forced one) are no-ops
        // ---- Bid (emd_cuda.cu:94-178): TPB lanes share one bidder and split the targets ----
        int tpb = 1;
        while (tpb < 64 && tpb * 2 * nu <= T) tpb *= 2;
        const int per_round = T / tpb;
        const int sub = tid & (tpb - 1);
        for (int u0 = 0; u0 < nu; u0 += per_round) {
            const int u = u0 + tid / tpb;
            const bool live = u < nu;
            const int j = live ? unass[u] : unass[0];
            const float x1 = p1[j * 3 + 0], y1 = p1[j * 3 + 1], z1 = p1[j * 3 + 2];
            Cand c{-1e9f, -1e9f, -1};
            for (int k = sub; k < n; k += tpb) {
                const float s = __builtin_sqrtf(sq3(sx[k] - x1, sy[k] - y1, sz[k] - z1));
                const float d = (float)(3.0 - (double)s - (double)price[k]);
                if (d > c.best) {
                    c.better = c.best;
                    c.best = d;
                    c.best_i = k;
                } else if (d > c.better) {
                    c.better = d;
                }
            }
            for (int off = 1; off < tpb; off <<= 1) {
                Cand o;
                o.best = __shfl_xor(c.best, off, 64);
                o.better = __shfl_xor(c.better, off, 64);
                o.best_i = __shfl_xor(c.best_i, off, 64);
                c = merge(c, o);
            }
            if (live && sub == 0) {
                const float bi = c.best - c.better + eps;  // :174
                bid[j] = c.best_i;
                inc[j] = bi;
                atomicMax(&max_inc[c.best_i], __float_as_int(bi));  // :175 (bi > 0: int order == float order)
                win[c.best_i] = 0x7fffffff;
            }
        }
        __syncthreads();
        // ---- GetMax (:180-193): lowest qualifying bidder wins ----
        for (int u = tid; u < nu; u += T) {
            const int j = unass[u], t = bid[j];
            const double bi = inc[j], mi = __int_as_float(max_inc[t]);
            if (bi - 1e-6 <= mi && mi <= bi + 1e-6) atomicMin(&win[t], j);
        }
        __syncthreads();
        // ---- Assign (:195-214) ----
        for (int u = tid; u < nu; u += T) {
            const int j = unass[u], t = bid[j];
            if (last || win[t] == j) {
                const int owner = inv[t];
                if (!last && owner != -1) ass[owner] = -1;
                inv[t] = j;
                ass[j] = t;
                if (last) atomicAdd(&price[t], inc[j]);
                else price[t] += inc[j];
                max_inc[t] = __float_as_int(-1e9f);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    for (int j = tid; j < n; j += T) {  // CalcDist :216-225
        const int k = ass[j];
        dist[(size_t)smp * n + j] = sq3(p1[j * 3 + 0] - sx[k], p1[j * 3 + 1] - sy[k], p1[j * 3 + 2] - sz[k]);
    }
}

__global__ __launch_bounds__(256) void auction_bwd_kernel(size_t total, int n, const float *__restrict__ xyz1,
                                                           const float *__restrict__ xyz2,
                                                           const float *__restrict__ grad_dist,
                                                           const int *__restrict__ idx, float *__restrict__ grad1) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const size_t smp = t / n;
    const int j2 = idx[t];
    const float g = grad_dist[t] * 2;
#pragma unroll
    for (int c = 0; c < 3; c++) grad1[t * 3 + c] = g * (xyz1[t * 3 + c] - xyz2[(smp * n + j2) * 3 + c]);
}

}  // namespace

extern "C" {

int pcc_auction_forward(int b, int n, const float *xyz1, const float *xyz2, float eps, int iters, float *dist,
                        int *assignment, pcc_stream_t stream) {
    pcc::clear_error();
    if (b < 0 || n < 1 || iters < 1) return pcc::invalid("auction: bad size (n >= 1, iters >= 1)");
    if (n > 8192) return pcc::invalid("auction: n > 8192 does not fit the LDS-resident state");
    if (b == 0) return PCC_OK;
    if (!xyz1 || !xyz2 || !dist || !assignment) return pcc::invalid("auction: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t hot = (size_t)5 * n * 4 + 16, state = (size_t)5 * n * 4;
    const int in_lds = hot + state <= 160 * 1024;
    const size_t lds = in_lds ? hot + state : hot;
    static bool attr_done = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(auction_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    }();
    (void)attr_done;
    int *scratch = nullptr;
    if (!in_lds && hipMallocAsync(reinterpret_cast<void **>(&scratch), (size_t)b * state, st) != hipSuccess) {
        (void)hipGetLastError();
        pcc::set_error(PCC_ENOMEM, "auction: workspace hipMallocAsync failed");
        return PCC_ENOMEM;
    }
    {
        pcc::ProfScope prof("auction_kernel", st);
        hipLaunchKernelGGL(auction_kernel, dim3(b), dim3(1024), lds, st, n, xyz1, xyz2, eps, iters, dist, assignment,
                           scratch, in_lds);
    }
    if (scratch) (void)hipFreeAsync(scratch, st);
    return pcc::check_launch("auction_forward");
}

int pcc_auction_backward(int b, int n, const float *xyz1, const float *xyz2, const float *grad_dist,
                         const int *assignment, float *grad_xyz1, pcc_stream_t stream) {
    pcc::clear_error();
    if (b < 0 || n < 0) return pcc::invalid("auction_backward: bad size");
    if (b == 0 || n == 0) return PCC_OK;
    if (!xyz1 || !xyz2 || !grad_dist || !assignment || !grad_xyz1) return pcc::invalid("auction_backward: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t total = (size_t)b * n;
    hipLaunchKernelGGL(auction_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, n, xyz1, xyz2,
                       grad_dist, assignment, grad_xyz1);
    return pcc::check_launch("auction_backward");
}

}  // extern "C"
