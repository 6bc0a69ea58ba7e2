// Micro-benchmark: issue cost of v_cmp_lt_u64 against v_cmp_lt_u32 (+ the v_cndmask that consumes the mask) on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/cmp_bench.hip -o tools/cmp_bench
// Measured (MI355X): the two compares cost the same (5.6 cycles per SIMD with 4 waves, 11.6 with one); compare + s_nop +
// two v_cndmask on its mask take 40-50 cycles per group whatever the compare width: the insertion chains of the k-NN
// kernels pay for the mask hand-off, not for 64-bit keys.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, b0 = a0 ^ 0x55, b1 = a1 ^ 0x33, r0 = 0, r1 = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {  // 64-bit compare + 2 cndmask
                asm volatile("v_cmp_lt_u64 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n"
                             : "+v"(r0), "+v"(r1) : "v"((unsigned long long)a0 | ((unsigned long long)a1 << 32)),
                               "v"((unsigned long long)b0 | ((unsigned long long)b1 << 32)), "v"(a0), "v"(b0) : "vcc");
            } else if (MODE == 1) {  // 32-bit compare + 2 cndmask
                asm volatile("v_cmp_lt_u32 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n"
                             : "+v"(r0), "+v"(r1) : "v"(a1), "v"(b1), "v"(a0), "v"(b0) : "vcc");
            } else if (MODE == 2) {  // 64-bit compare alone
                asm volatile("v_cmp_lt_u64 vcc, %0, %1\n" :: "v"((unsigned long long)a0 | ((unsigned long long)a1 << 32)),
                             "v"((unsigned long long)b0 | ((unsigned long long)b1 << 32)) : "vcc");
            } else {  // 32-bit compare alone
                asm volatile("v_cmp_lt_u32 vcc, %0, %1\n" :: "v"(a1), "v"(b1) : "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1;
}

template <int MODE>
void run(const char *name, int wg_per_cu, int insts) {
    const int iters = 4096, grid = 256 * wg_per_cu;
    unsigned *out;
    (void)hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    k<MODE><<<grid, 256>>>(out, 16, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    k<MODE><<<grid, 256>>>(out, iters, 1);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    // cycles per wave-instruction-group on one SIMD: waves per SIMD = wg_per_cu (4 waves per WG over 4 SIMDs)
    const double groups = (double)iters * 8 * wg_per_cu;  // per SIMD
    printf("%-34s wg/cu=%d  %.3f ms  %.2f cycles per group of %d instruction(s) at 2.4 GHz\n", name, wg_per_cu, ms, ms * 1e-3 * 2.4e9 / groups, insts);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("v_cmp_lt_u64 + nop + 2 cndmask", w, 3);
        run<1>("v_cmp_lt_u32 + nop + 2 cndmask", w, 3);
        run<2>("v_cmp_lt_u64", w, 1);
        run<3>("v_cmp_lt_u32", w, 1);
    }
    return 0;
}
