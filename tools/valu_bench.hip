// Micro-benchmark: f32 VALU issue rates on gfx950 (scalar vs packed f32, v_exp_f32, v_min3_f32).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_t p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float b = 1.0001f, c = 0.5f;
    const float2_t pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {  // 8 independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 1) {  // 4 independent v_pk_fma_f32 (same flops as 8 v_fma)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            } else if (MODE == 2) {  // 8 v_exp_f32
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 3) {  // 8 v_min3_f32
                asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                             "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 4) {  // 4 v_pk_add_f32 + 4 v_pk_mul_f32
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %5\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            } else if (MODE == 5) {  // mix: 6 fma + 2 exp (approxmatch-like)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %3, %3\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char *name, int wg_per_cu, double ops_per_thread_iter) {
    const int iters = 4096;
    const int grid = 256 * wg_per_cu;
    float *out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    k<MODE><<<grid, 256>>>(out, 16, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<MODE><<<grid, 256>>>(out, iters, 0.5f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    double lane_ops = (double)grid * 256 * iters * 8 * ops_per_thread_iter;
    printf("%-28s wg/cu=%d  %.3f ms  %.1f T lane-instr-results/s\n", name, wg_per_cu, ms, lane_ops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 x8", w, 8);
        run<1>("v_pk_fma_f32 x4 (8 results)", w, 8);
        run<2>("v_exp_f32 x8", w, 8);
        run<3>("v_min3_f32 x8", w, 8);
        run<4>("v_pk_add/mul x8 (16 results)", w, 16);
        run<5>("6 fma + 2 exp", w, 8);
    }
    return 0;
}
