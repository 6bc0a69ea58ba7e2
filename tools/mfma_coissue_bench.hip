// Micro-benchmark: does a wave streaming VALU instructions slow down a wave streaming v_mfma_f32_32x32x2_f32 on the same
// SIMD (gfx950)?  8-wave workgroups, one per CU: waves 0-3 run CH independent MFMA chains, waves 4-7 run v_fma (or idle).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_coissue_bench.hip -o tools/mfma_coissue_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, int VALU /* 0: waves 4-7 exit, 1: v_fma stream, 2: v_fma stream and waves 0-3 exit */>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed) {
    const int w = threadIdx.x >> 6;
    if (w < 4) {
        if (VALU == 2) return;
        f32x16 acc[CH];
        for (int c = 0; c < CH; c++)
            for (int r = 0; r < 16; r++) acc[c][r] = seed;
        const float a = seed + threadIdx.x, b = seed * 0.5f;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
        float s = 0;
        for (int c = 0; c < CH; c++)
            for (int r = 0; r < 16; r++) s += acc[c][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (VALU == 0) return;
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const float b = 1.0001f, c = 0.5f;
        // the same wall time as the MFMA waves when alone: 8 * CH MFMAs of 64 cycles against 64 * CH v_fma of >= 8 cycles
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8 * CH; u++)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
}

template <int CH, int VALU>
void run(const char *name) {
    const int iters = 4096, grid = 256;
    float *out;
    (void)hipMalloc(&out, (size_t)grid * 512 * 4);
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    k<CH, VALU><<<grid, 512>>>(out, 16, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    k<CH, VALU><<<grid, 512>>>(out, iters, 1.f);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-44s chains=%d  %7.3f ms  %6.1f cycles per MFMA of a wave, %5.1f per v_fma of a wave\n", name, CH, ms,
           cyc / ((double)iters * 8 * CH), cyc / ((double)iters * 64 * CH * 8));
    (void)hipFree(out);
}

int main() {
    run<1, 0>("MFMA waves alone");
    run<2, 0>("MFMA waves alone");
    run<4, 0>("MFMA waves alone");
    run<1, 2>("v_fma waves alone");
    run<2, 1>("MFMA waves + v_fma waves on the same SIMDs");
    run<4, 1>("MFMA waves + v_fma waves on the same SIMDs");
    return 0;
}
