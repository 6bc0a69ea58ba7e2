// Micro-benchmark: could an f32-MFMA lower-bound FILTER speed up the exhaustive Chamfer search?
// Build: hipcc -O3 --offload-arch=gfx950 -w -DQG=4 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_filter_bench.hip -o tools/mfma_filter_bench
// (-amdgpu-mfma-vgpr-form: MFMA results straight into VGPRs; with AGPR results every value costs a v_accvgpr_read on the VALU)
//
// The exact search (csrc/chamfer.hip: nn_fwd_kernel) evaluates every pair on the VALU in difference form -- 6.9 lane
// operations per pair, 42 us for both directions at B=32, N=M=2048.  The filter idea: the expanded form
// |q|^2 - 2 p.q = [-2qx, -2qy, -2qz, |q|^2] . [px, py, pz, 1] is a K=4 contraction, i.e. ONE v_mfma_f32_16x16x4_f32 per
// (16 candidates x 16 queries) tile on the matrix pipe, which issues next to the VALU.  Its values carry ~1e-7 (|p|^2+|q|^2)
// of cancellation error, so they cannot name the winner (indices must be bit-exact); they can only BOUND it:
//   pass 1: approximate minimum m~ per query (MFMA + v_min3 on the VALU);
//   pass 2: the same tiles again, every value compared with m~ + eps; only pairs below it are re-evaluated exactly
//           (difference form, 64-bit (distance, index) key) -- a handful per query.
// This program measures exactly those two passes (the exact re-evaluation of the survivors included) on the bench's
// shapes, checks the result against a CPU brute force, and prints the times next to the MFMA pipe's own floor
// (2 x 268 M pairs / (256 pairs per 32 cycles x 1024 SIMDs x 2.4 GHz) = 2 x 13.7 us).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef QG
#define QG 2
#endif
constexpr int kQG = QG;         // 16-query groups per wave (B operands kept in registers)
constexpr int kWaves = 4;       // waves per workgroup
constexpr int kCH = 2048;       // candidates staged in LDS per chunk

__device__ __forceinline__ float sq3(float x, float y, float z) { return __builtin_fmaf(z, z, __builtin_fmaf(x, x, y * y)); }

// MODE 1: pass 1 (approximate minima -> approx[]);  MODE 2: pass 2 (survivors of approx[] + eps re-evaluated exactly)
template <int MODE>
__global__ __launch_bounds__(64 * kWaves) void nn_mfma_kernel(int b, int n, const float *__restrict__ xyz, int m,
                                                               const float *__restrict__ xyz2, float *__restrict__ approx1,
                                                               float *__restrict__ approx2, float eps_rel,
                                                               float *__restrict__ d1, int *__restrict__ i1,
                                                               float *__restrict__ d2, int *__restrict__ i2,
                                                               unsigned long long *__restrict__ survivors) {
    __shared__ __attribute__((aligned(16))) float lds[4 * kCH];  // -2x | -2y | -2z | |q|^2 of the candidates
    constexpr int TQ = 16 * kQG * kWaves;                         // queries per workgroup
    const int tiles_n = (n + TQ - 1) / TQ, tiles_m = (m + TQ - 1) / TQ;
    int bid = blockIdx.x;
    const int dir0 = b * tiles_n;
    const float *Q, *C;
    float *approx, *out_d;
    int *out_i;
    int nq, nc, tiles;
    if (bid < dir0) { Q = xyz; C = xyz2; nq = n; nc = m; tiles = tiles_n; approx = approx1; out_d = d1; out_i = i1; }
    else { bid -= dir0; Q = xyz2; C = xyz; nq = m; nc = n; tiles = tiles_m; approx = approx2; out_d = d2; out_i = i2; }
    const int smp = bid / tiles, tile = bid - smp * tiles;
    Q += (size_t)smp * nq * 3;
    C += (size_t)smp * nc * 3;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;  // B operand: component kq of query `col`; D: candidate rows 4 kq + r, query col

    // this lane's B operands (component kq of its query in each of the kQG groups) and, for pass 2, the query itself
    float bq[kQG], qx[kQG], qy[kQG], qz[kQG], thr[kQG], qn[kQG];
    unsigned long long best[kQG];
#pragma unroll
    for (int g = 0; g < kQG; g++) {
        int q = tile * TQ + (w * kQG + g) * 16 + col;
        q = q < nq ? q : nq - 1;
        qx[g] = Q[q * 3 + 0]; qy[g] = Q[q * 3 + 1]; qz[g] = Q[q * 3 + 2];
        bq[g] = kq == 0 ? qx[g] : kq == 1 ? qy[g] : kq == 2 ? qz[g] : 1.0f;
        qn[g] = sq3(qx[g], qy[g], qz[g]);
        best[g] = ~0ull;
        // pass 2 threshold in the MFMA's own units (|q'|^2 - 2 p.q'): approximate minimum + a bound of both roundings
        thr[g] = MODE == 2 ? approx[(size_t)smp * nq + q] - qn[g] + eps_rel * (qn[g] + 4.0f) : 0.f;
    }
    float mn[kQG];
#pragma unroll
    for (int g = 0; g < kQG; g++) mn[g] = __builtin_inff();
    unsigned long long nsurv = 0;

    for (int c0 = 0; c0 < nc; c0 += kCH) {
        const int cnt = min(kCH, nc - c0);
        __syncthreads();
        for (int i = tid; i < kCH; i += 64 * kWaves) {
            float x = 0.f, y = 0.f, z = 0.f, s = __builtin_inff();  // a padded candidate can never win
            if (i < cnt) { x = C[(c0 + i) * 3 + 0]; y = C[(c0 + i) * 3 + 1]; z = C[(c0 + i) * 3 + 2]; s = sq3(x, y, z); }
            lds[i] = -2.f * x; lds[kCH + i] = -2.f * y; lds[2 * kCH + i] = -2.f * z; lds[3 * kCH + i] = s;
        }
        __syncthreads();
        const int ntiles = (cnt + 15) / 16;  // (kCH is a multiple of 64: the last 4-tile step may read +inf padding)
        constexpr int U = 4;                 // tiles per step: the A operands of a step are fetched before its MFMAs
        for (int t0 = 0; t0 < ntiles; t0 += U) {
            float a[U];
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = lds[kq * kCH + (t0 + u) * 16 + col];  // component kq of candidate (t0+u)*16 + col
            v4f d[U][kQG];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int g = 0; g < kQG; g++) {
                    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
                    d[u][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bq[g], z4, 0, 0, 0);  // D[cand 4 kq + r][query col]
                }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int g = 0; g < kQG; g++) {
                    const v4f dv = d[u][g];
                    if (MODE == 1) {
                        mn[g] = __builtin_fminf(__builtin_fminf(mn[g], dv.x), dv.y);
                        mn[g] = __builtin_fminf(__builtin_fminf(mn[g], dv.z), dv.w);
                    } else {
                        const bool hit = (dv.x <= thr[g]) | (dv.y <= thr[g]) | (dv.z <= thr[g]) | (dv.w <= thr[g]);
                        if (__builtin_amdgcn_ballot_w64(hit)) {  // rare: exact difference form for this lane's four candidates
                            const float dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                if (dd[r] <= thr[g]) {
                                    const int ci = (t0 + u) * 16 + 4 * kq + r;
                                    // (-2 x) * -0.5 is x exactly: the staged operands serve the exact re-evaluation too
                                    const float e = sq3(-0.5f * lds[ci] - qx[g], -0.5f * lds[kCH + ci] - qy[g], -0.5f * lds[2 * kCH + ci] - qz[g]);
                                    const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | (unsigned)(c0 + ci);
                                    best[g] = key < best[g] ? key : best[g];
                                    nsurv++;
                                }
                            }
                        }
                    }
                }
        }
    }
    // the four lanes of a query (kq = 0..3) meet
#pragma unroll
    for (int g = 0; g < kQG; g++) {
        const int q = tile * TQ + (w * kQG + g) * 16 + col;
        if (MODE == 1) {
            float v = mn[g];
            v = fminf(v, __shfl_xor(v, 16, 64));
            v = fminf(v, __shfl_xor(v, 32, 64));
            if (kq == 0 && q < nq) approx[(size_t)smp * nq + q] = v + qn[g];
        } else {
            unsigned long long k = best[g];
            for (int off = 16; off < 64; off <<= 1) {
                const unsigned hi = (unsigned)__shfl_xor((int)(k >> 32), off, 64), lo = (unsigned)__shfl_xor((int)(k & 0xffffffffu), off, 64);
                const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                k = o < k ? o : k;
            }
            if (kq == 0 && q < nq) {
                out_d[(size_t)smp * nq + q] = __uint_as_float((unsigned)(k >> 32));
                out_i[(size_t)smp * nq + q] = (int)(k & 0xffffffffu);
            }
        }
    }
    if (MODE == 2) {
        for (int off = 32; off > 0; off >>= 1) nsurv += __shfl_down(nsurv, off, 64);
        if (lane == 0 && nsurv) atomicAdd(survivors, nsurv);
    }
}

// The same pass 1 on v_mfma_f32_32x32x2_f32: two chained K=2 steps per (32 candidates x 32 queries) tile.
typedef float v16f __attribute__((ext_vector_type(16)));
template <int G32>
__global__ __launch_bounds__(64 * kWaves) void nn_mfma32_pass1_kernel(int b, int n, const float *__restrict__ xyz, int m,
                                                                       const float *__restrict__ xyz2, float *__restrict__ approx1,
                                                                       float *__restrict__ approx2) {
    __shared__ __attribute__((aligned(16))) float lds[4 * kCH];
    constexpr int TQ = 32 * G32 * kWaves;
    const int tiles_n = (n + TQ - 1) / TQ, tiles_m = (m + TQ - 1) / TQ;
    int bid = blockIdx.x;
    const int dir0 = b * tiles_n;
    const float *Q, *C;
    float *approx;
    int nq, nc, tiles;
    if (bid < dir0) { Q = xyz; C = xyz2; nq = n; nc = m; tiles = tiles_n; approx = approx1; }
    else { bid -= dir0; Q = xyz2; C = xyz; nq = m; nc = n; tiles = tiles_m; approx = approx2; }
    const int smp = bid / tiles, tile = bid - smp * tiles;
    Q += (size_t)smp * nq * 3;
    C += (size_t)smp * nc * 3;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;  // operand component kh (step 0: x|y, step 1: z|1) of point `col`
    float b0[G32], b1[G32], qn[G32], mn[G32];
#pragma unroll
    for (int g = 0; g < G32; g++) {
        int q = tile * TQ + (w * G32 + g) * 32 + col;
        q = q < nq ? q : nq - 1;
        const float x = Q[q * 3], y = Q[q * 3 + 1], z = Q[q * 3 + 2];
        b0[g] = kh ? y : x;
        b1[g] = kh ? 1.0f : z;
        qn[g] = sq3(x, y, z);
        mn[g] = __builtin_inff();
    }
    for (int c0 = 0; c0 < nc; c0 += kCH) {
        const int cnt = min(kCH, nc - c0);
        __syncthreads();
        for (int i = tid; i < kCH; i += 64 * kWaves) {
            float x = 0.f, y = 0.f, z = 0.f, s = __builtin_inff();
            if (i < cnt) { x = C[(c0 + i) * 3 + 0]; y = C[(c0 + i) * 3 + 1]; z = C[(c0 + i) * 3 + 2]; s = sq3(x, y, z); }
            lds[i] = -2.f * x; lds[kCH + i] = -2.f * y; lds[2 * kCH + i] = -2.f * z; lds[3 * kCH + i] = s;
        }
        __syncthreads();
        const int ntiles = (cnt + 31) / 32;
        for (int t = 0; t < ntiles; t += 2) {
            float a0[2], a1[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                a0[u] = lds[kh * kCH + (t + u) * 32 + col];        // -2x | -2y
                a1[u] = lds[(2 + kh) * kCH + (t + u) * 32 + col];  // -2z | |q|^2
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int g = 0; g < G32; g++) {
                    v16f d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[g], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[g], d, 0, 0, 0);
                    float v = mn[g];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) v = __builtin_fminf(__builtin_fminf(v, d[r]), d[r + 1]);
                    mn[g] = v;
                }
        }
    }
#pragma unroll
    for (int g = 0; g < G32; g++) {
        const int q = tile * TQ + (w * G32 + g) * 32 + col;
        float v = fminf(mn[g], __shfl_xor(mn[g], 32, 64));
        if (kh == 0 && q < nq) approx[(size_t)smp * nq + q] = v + qn[g];
    }
}

int main() {
    const int B = 32, N = 2048, M = 2048;
    std::vector<float> h1((size_t)B * N * 3), h2((size_t)B * M * 3);
    srand(1236);
    // the bench's "recon" pair: a surface-like cloud in the unit ball and a noisy permutation of it
    for (int s = 0; s < B; s++) {
        for (int i = 0; i < M; i++) {
            float v[3], nr = 0;
            for (int c = 0; c < 3; c++) { v[c] = (float)rand() / RAND_MAX * 2 - 1; nr += v[c] * v[c]; }
            nr = std::sqrt(nr) + 1e-9f;
            const float r = std::cbrt(0.3f + 0.7f * (float)rand() / RAND_MAX);
            for (int c = 0; c < 3; c++) h2[((size_t)s * M + i) * 3 + c] = v[c] / nr * r;
        }
        for (int i = 0; i < N; i++) {
            const int j = rand() % M;
            for (int c = 0; c < 3; c++) h1[((size_t)s * N + i) * 3 + c] = h2[((size_t)s * M + j) * 3 + c] + 0.02f * ((float)rand() / RAND_MAX - 0.5f) * 3.4f;
        }
    }
    float *x1, *x2, *a1, *a2, *d1, *d2;
    int *i1, *i2;
    unsigned long long *surv;
    hipMalloc(&x1, h1.size() * 4); hipMalloc(&x2, h2.size() * 4);
    hipMalloc(&a1, (size_t)B * N * 4); hipMalloc(&a2, (size_t)B * M * 4);
    hipMalloc(&d1, (size_t)B * N * 4); hipMalloc(&d2, (size_t)B * M * 4);
    hipMalloc(&i1, (size_t)B * N * 4); hipMalloc(&i2, (size_t)B * M * 4);
    hipMalloc(&surv, 8);
    hipMemcpy(x1, h1.data(), h1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x2, h2.data(), h2.size() * 4, hipMemcpyHostToDevice);
    constexpr int TQ = 16 * kQG * kWaves;
    const int grid = B * ((N + TQ - 1) / TQ) + B * ((M + TQ - 1) / TQ);
    const float eps_rel = 8 * 1.1920929e-7f;  // 8 ulp of (|p|^2 + |q|^2 <= |p|^2 + 4): covers the MFMA's fma chain and the two norms
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    float t1 = 0, t2 = 0;
    const int reps = 20;
    for (int it = 0; it < reps + 3; it++) {
        hipMemset(surv, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(nn_mfma_kernel<1>, dim3(grid), dim3(64 * kWaves), 0, 0, B, N, x1, M, x2, a1, a2, eps_rel, d1, i1, d2, i2, surv);
        hipEventRecord(e1);
        hipLaunchKernelGGL(nn_mfma_kernel<2>, dim3(grid), dim3(64 * kWaves), 0, 0, B, N, x1, M, x2, a1, a2, eps_rel, d1, i1, d2, i2, surv);
        hipEventRecord(e2);
        hipEventSynchronize(e2);
        float u1, u2;
        hipEventElapsedTime(&u1, e0, e1); hipEventElapsedTime(&u2, e1, e2);
        if (it >= 3) { t1 += u1; t2 += u2; }
    }
    unsigned long long hs = 0;
    hipMemcpy(&hs, surv, 8, hipMemcpyDeviceToHost);
    // the 32x32x2 form of pass 1 (two chained K=2 steps per tile)
    float t32 = 0;
    {
        constexpr int G32 = 2;
        constexpr int TQ32 = 32 * G32 * kWaves;
        const int grid32 = B * ((N + TQ32 - 1) / TQ32) + B * ((M + TQ32 - 1) / TQ32);
        for (int it = 0; it < reps + 3; it++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(nn_mfma32_pass1_kernel<G32>, dim3(grid32), dim3(64 * kWaves), 0, 0, B, N, x1, M, x2, d1, d2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float u1;
            hipEventElapsedTime(&u1, e0, e1);
            if (it >= 3) t32 += u1;
        }
        // (d1 / d2 hold that pass's approximate minima now; pass 2 below is re-run to restore the exact results)
        std::vector<float> c16((size_t)N), c32((size_t)N);
        hipMemcpy(c16.data(), a1, N * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c32.data(), d1, N * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int q = 0; q < N; q++) worst = std::max(worst, (double)std::fabs(c16[q] - c32[q]));
        printf("  pass 1 on v_mfma_f32_32x32x2_f32       %7.1f us   (largest difference to the 16x16x4 minima: %.2e)\n", t32 / reps * 1e3, worst);
        hipLaunchKernelGGL(nn_mfma_kernel<2>, dim3(grid), dim3(64 * kWaves), 0, 0, B, N, x1, M, x2, a1, a2, eps_rel, d1, i1, d2, i2, surv);
        hipDeviceSynchronize();
    }
    // exactness check of the filtered result on sample 0, direction 1, against the CPU difference form (lowest index on ties)
    std::vector<float> gd((size_t)N);
    std::vector<int> gi((size_t)N);
    hipMemcpy(gd.data(), d1, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(gi.data(), i1, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int q = 0; q < N; q++) {
        float bd = INFINITY; int bi = 0;
        for (int k = 0; k < M; k++) {
            const float dx = h2[k * 3] - h1[q * 3], dy = h2[k * 3 + 1] - h1[q * 3 + 1], dz = h2[k * 3 + 2] - h1[q * 3 + 2];
            const float d = std::fmaf(dz, dz, std::fmaf(dx, dx, dy * dy));
            if (d < bd) { bd = d; bi = k; }
        }
        if (bi != gi[q] || bd != gd[q]) bad++;
    }
    const double pairs = 2.0 * B * N * M;
    printf("MFMA filter, B=%d N=M=%d both directions (%d workgroups of %d waves, %d queries per wave):\n", B, N, grid, kWaves, 16 * kQG);
    printf("  pass 1 (approximate minima)            %7.1f us   (matrix-pipe floor 13.7 us)\n", t1 / reps * 1e3);
    printf("  pass 2 (threshold + exact survivors)   %7.1f us   survivors per query %.2f\n", t2 / reps * 1e3, (double)hs / (B * (N + M)));
    printf("  filter total                           %7.1f us   = %.1f TFLOP/s at 8 flop/pair (nn_fwd_kernel: 42 us = 50.9)\n",
           (t1 + t2) / reps * 1e3, pairs * 8 / ((t1 + t2) / reps * 1e-3) / 1e12);
    printf("  exactness on sample 0 (index and distance bits vs CPU difference form): %d of %d queries differ\n", bad, N);
    return 0;
}
