/*
 * neighbour_oracle.c -- CPU restatement of the reference's kNN (TEST INFRASTRUCTURE ONLY, see
 * structural_oracle.c for the rules).
 *
 * Reference: src/utils/neighbour_ops.py
 *   pykeops_knn  (:77-82)  GPU path: argKmin over ((x_i - x_j)**2).sum(-1)      -> oracle_knn_diff
 *   torch_knn    (:71-74)  CPU path: topk(largest=False) of -2 x^T x + |x|^2 + |x|^2^T (:53-60) -> oracle_knn_expanded
 * PyKeOps is a third-party dependency that is not installed here (pyproject.toml:15, >=2.3, no lockfile) and the
 * reference holds no vectors for it: the difference-form variant is "parity unpinned"; the expanded-form
 * variant is pinned by tests/golden/ref_neighbour_ops.npz (outputs of the reference's own torch_knn).
 * Both return, per query, the k candidate indices sorted by (distance, index) ascending.  The f32 rounding
 * orders are the ones the HIP kernels use (sequential fma chains over the channel index), so the kernels
 * are compared bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

extern int oracle_get_threads(void);

typedef struct { float d; int i; } cand_t;

static int cmp_cand(const void *a, const void *b) {
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* x[b][c][n]; idx[b][n][k]; only the queries q % qstride == 0 are computed (qstride 1: all; larger strides let the
 * tests spot-check big clouds without sorting every row) */
void oracle_knn_diff_strided(int b, int c, int n, int k, const float *x, int64_t *idx, int qstride) {
    int threads = oracle_get_threads();
#pragma omp parallel for num_threads(threads) collapse(2) schedule(static)
    for (int s = 0; s < b; s++)
        for (int q = 0; q < n; q++) {
            if (q % qstride) continue;
            const float *xb = x + (size_t)s * c * n;
            cand_t *cd = (cand_t *)malloc(sizeof(cand_t) * n);
            for (int j = 0; j < n; j++) {
                float acc = 0.f;
                for (int ch = 0; ch < c; ch++) {
                    float df = xb[(size_t)ch * n + j] - xb[(size_t)ch * n + q];
                    acc = ch == 0 ? df * df : fmaf(df, df, acc);
                }
                cd[j].d = acc;
                cd[j].i = j;
            }
            qsort(cd, n, sizeof(cand_t), cmp_cand);
            for (int o = 0; o < k; o++) idx[((size_t)s * n + q) * k + o] = cd[o].i;
            free(cd);
        }
}

void oracle_knn_diff(int b, int c, int n, int k, const float *x, int64_t *idx) {
    oracle_knn_diff_strided(b, c, n, k, x, idx, 1);
}

void oracle_knn_expanded(int b, int c, int n, int k, const float *x, int64_t *idx, float *dist_out) {
    int threads = oracle_get_threads();
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int s = 0; s < b; s++) {
        const float *xb = x + (size_t)s * c * n;
        float *sq = (float *)malloc(sizeof(float) * n);
        for (int j = 0; j < n; j++) {
            float a = 0.f;
            for (int ch = 0; ch < c; ch++) a = fmaf(xb[(size_t)ch * n + j], xb[(size_t)ch * n + j], a);
            sq[j] = a;
        }
        cand_t *cd = (cand_t *)malloc(sizeof(cand_t) * n);
        for (int q = 0; q < n; q++) {
            for (int j = 0; j < n; j++) {
                float dot = 0.f;
                for (int ch = 0; ch < c; ch++) dot = fmaf(xb[(size_t)ch * n + j], xb[(size_t)ch * n + q], dot);
                cd[j].d = (-2.0f * dot + sq[j]) + sq[q];
                cd[j].i = j;
                if (dist_out) dist_out[((size_t)s * n + q) * n + j] = cd[j].d;
            }
            qsort(cd, n, sizeof(cand_t), cmp_cand);
            for (int o = 0; o < k; o++) idx[((size_t)s * n + q) * k + o] = cd[o].i;
        }
        free(cd);
        free(sq);
    }
}
