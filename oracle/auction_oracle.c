/*
 * auction_oracle.c -- CPU restatement of the reference's auction EMD (TEST INFRASTRUCTURE ONLY).
 *
 * Reference: external/emd/src/emd_cuda.cu (kernels :22-225, host loop :227-281, backward :283-315) and
 * external/emd/emd/emd_module.py:16-100 (buffer initialisation).  The CUDA version is racy by construction:
 *   - GetMax (:180-193): several bidders whose increment equals the maximum within 1e-6 write max_idx[target]
 *     concurrently; the survivor is unspecified;
 *   - Assign (:195-214): assignment[ass_inv] = -1 (:205) is written while other threads of the same launch read
 *     assignment[j] == -1 (:198); whether an owner evicted in this launch also acts in it is unspecified;
 *   - on the forced last iteration several bidders may take the same target (assignment_inv / price races).
 * This oracle fixes one deterministic interpretation, the one the HIP kernel implements:
 *   - the set of bidders of an iteration is the set unassigned when the iteration starts;
 *   - among qualifying bidders of a target the LOWEST index wins;
 *   - forced assignments of the last iteration are applied in ascending bidder order.
 * Everything else (bid values in double as written `3.0 - sqrtf(..) - price` :145, first-maximum tie rule
 * :146-153, increment best-better+eps :174, price update :209, the 1e-6 window in double :187, dist :220-223)
 * follows the file.  Parity unpinned: the reference has no tests or vectors for this module and it cannot run
 * here; because of the races even two runs of the reference need not agree.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

extern int oracle_get_threads(void);

static inline float sq3c(float x, float y, float z) { return fmaf(z, z, fmaf(x, x, y * y)); }

/* returns 1 on success, -1 on the reference's input errors (:235-248) */
int oracle_auction_forward(int b, int n, const float *xyz1, const float *xyz2, float eps, int iters, float *dist,
                           int *assignment, float *price_out) {
    if (n % 1024 != 0 || b > 512 || iters < 1) return -1;
    int threads = oracle_get_threads();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int s = 0; s < b; s++) {
        const float *p1 = xyz1 + (size_t)s * n * 3, *p2 = xyz2 + (size_t)s * n * 3;
        int *ass = assignment + (size_t)s * n;
        int *inv = (int *)malloc(sizeof(int) * n), *bid = (int *)calloc(n, sizeof(int));
        int *max_idx = (int *)calloc(n, sizeof(int)), *un = (int *)malloc(sizeof(int) * n);
        float *price = (float *)calloc(n, sizeof(float)), *inc = (float *)calloc(n, sizeof(float));
        float *max_inc = (float *)calloc(n, sizeof(float)); /* emd_module.py:41: zeros */
        for (int j = 0; j < n; j++) { ass[j] = -1; inv[j] = -1; }
        for (int it = 0; it < iters; it++) {
            int last = it == iters - 1, nu = 0;
            for (int j = 0; j < n; j++) if (ass[j] == -1) un[nu++] = j;              /* :29-92 */
            for (int u = 0; u < nu; u++) {                                            /* Bid :94-178 */
                int j = un[u];
                float x1 = p1[j * 3], y1 = p1[j * 3 + 1], z1 = p1[j * 3 + 2];
                float best = -1e9f, better = -1e9f;
                int best_i = -1;
                for (int k = 0; k < n; k++) {
                    float x2 = p2[k * 3] - x1, y2 = p2[k * 3 + 1] - y1, z2 = p2[k * 3 + 2] - z1;
                    float d = (float)(3.0 - (double)sqrtf(sq3c(x2, y2, z2)) - (double)price[k]); /* :145 */
                    if (d > best) { better = best; best = d; best_i = k; }              /* :146-150 */
                    else if (d > better) better = d;                                    /* :151-153 */
                }
                bid[j] = best_i;                                                        /* :173 */
                inc[j] = best - better + eps;                                           /* :174 */
                if (inc[j] > max_inc[best_i]) max_inc[best_i] = inc[j];                 /* :175 atomicMax */
            }
            for (int u = nu - 1; u >= 0; u--) {                                       /* GetMax :180-193, lowest j last */
                int j = un[u], t = bid[j];
                double bi = inc[j], mi = max_inc[t];
                if (bi - 1e-6 <= mi && mi <= bi + 1e-6) max_idx[t] = j;               /* :187-189 */
            }
            for (int u = 0; u < nu; u++) {                                            /* Assign :195-214 */
                int j = un[u], t = bid[j];
                if (last || max_idx[t] == j) {
                    int owner = inv[t];
                    if (!last && owner != -1) ass[owner] = -1;                          /* :204-206 */
                    inv[t] = j;
                    ass[j] = t;
                    price[t] += inc[j];                                                 /* :209 */
                    max_inc[t] = -1e9f;                                                 /* :210 */
                }
            }
        }
        for (int j = 0; j < n; j++) {                                                 /* CalcDist :216-225 */
            int k = ass[j];
            dist[(size_t)s * n + j] = sq3c(p1[j * 3] - p2[k * 3], p1[j * 3 + 1] - p2[k * 3 + 1], p1[j * 3 + 2] - p2[k * 3 + 2]);
        }
        if (price_out) memcpy(price_out + (size_t)s * n, price, sizeof(float) * n);
        free(inv); free(bid); free(max_idx); free(un); free(price); free(inc); free(max_inc);
    }
    return 1;
}

/* emd_cuda_backward :283-315: grad_xyz1 = 2 g (p1 - p2[idx]); grad_xyz2 untouched (zeros, emd_module.py:76-79) */
void oracle_auction_backward(int b, int n, const float *xyz1, const float *xyz2, const float *grad_dist,
                             const int *idx, float *grad_xyz1) {
    for (size_t t = 0; t < (size_t)b * n; t++) {
        size_t s = t / n;
        int j2 = idx[t];
        float g = grad_dist[t] * 2;
        for (int c = 0; c < 3; c++)
            grad_xyz1[t * 3 + c] = g * (xyz1[t * 3 + c] - xyz2[(s * n + j2) * 3 + c]);
    }
}
