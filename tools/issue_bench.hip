// Micro-benchmark: issue cost of the instruction patterns of the k-NN selection (gfx950), one or two waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_bench.hip -o tools/issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    int n0 = threadIdx.x, n1 = n0 * 3;
    __shared__ float2 lds[17 * 256];
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == 1) {  // 8 dependent v_fma_f32
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
        } else if (MODE == 2) {  // 8 independent v_med3_f32 reading neighbours (the list update)
            asm volatile("v_med3_f32 %0, %1, %8, %0\n v_med3_f32 %1, %2, %8, %1\n v_med3_f32 %2, %3, %8, %2\n v_med3_f32 %3, %4, %8, %3\n"
                         "v_med3_f32 %4, %5, %8, %4\n v_med3_f32 %5, %6, %8, %5\n v_med3_f32 %6, %7, %8, %6\n v_med3_f32 %7, %0, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (MODE == 3) {  // 8 v_cmp_lt_f32_e64 into distinct scalar pairs
            asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cmp_lt_f32_e64 s[22:23], %1, %8\n v_cmp_lt_f32_e64 s[24:25], %2, %8\n"
                         "v_cmp_lt_f32_e64 s[26:27], %3, %8\n v_cmp_lt_f32_e64 s[28:29], %4, %8\n v_cmp_lt_f32_e64 s[30:31], %5, %8\n"
                         "v_cmp_lt_f32_e64 s[32:33], %6, %8\n v_cmp_lt_f32_e64 s[34:35], %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
        } else if (MODE == 4) {  // 8 v_cmp_lt_f32_e32 (all into vcc)
            asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %1\n") : "+v"(a0) : "v"(b) : "vcc");
        } else if (MODE == 5) {  // 8 compares into distinct pairs, then 8 selects on them (pipelined hand-off)
            asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cmp_lt_f32_e64 s[22:23], %1, %8\n v_cmp_lt_f32_e64 s[24:25], %2, %8\n"
                         "v_cmp_lt_f32_e64 s[26:27], %3, %8\n v_cmp_lt_f32_e64 s[28:29], %4, %8\n v_cmp_lt_f32_e64 s[30:31], %5, %8\n"
                         "v_cmp_lt_f32_e64 s[32:33], %6, %8\n v_cmp_lt_f32_e64 s[34:35], %7, %8\n s_nop 1\n"
                         "v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cndmask_b32_e64 %1, %1, %9, s[22:23]\n v_cndmask_b32_e64 %2, %2, %9, s[24:25]\n"
                         "v_cndmask_b32_e64 %3, %3, %9, s[26:27]\n v_cndmask_b32_e64 %4, %4, %9, s[28:29]\n v_cndmask_b32_e64 %5, %5, %9, s[30:31]\n"
                         "v_cndmask_b32_e64 %6, %6, %9, s[32:33]\n v_cndmask_b32_e64 %7, %7, %9, s[34:35]\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
        } else if (MODE == 6) {  // compare -> select through vcc, 8 times (what the compiler emits)
            asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %2, vcc\n") : "+v"(a0) : "v"(b), "v"(c) : "vcc");
        } else if (MODE == 7) {  // 8 selects on masks made long ago
            asm volatile("v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n"
                         "v_cndmask_b32_e64 %3, %3, %9, s[20:21]\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n"
                         "v_cndmask_b32_e64 %6, %6, %9, s[20:21]\n v_cndmask_b32_e64 %7, %7, %9, s[20:21]\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21");
        } else if (MODE == 8) {  // offer tail: address, ds_write_b64, carry-in add (8 times, masks made long ago)
            float2 *base = lds + threadIdx.x;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                base[(n0 & 15) * 256] = make_float2(a0, a1);
                asm volatile("v_addc_co_u32_e64 %0, vcc, 0, %0, s[20:21]" : "+v"(n0) : : "vcc", "s20", "s21");
            }
        } else if (MODE == 9) {  // compare + s_and_saveexec + restore, 8 times (the branchy offer)
            asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n")
                         : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20", "s21");
        } else if (MODE == 10) {  // 8 v_mov_b32 (copies the compiler adds around the chain)
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n"
                         "v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + n0 + n1 + lds[threadIdx.x].x;
}

template <int MODE>
void run(const char *name, int wg_per_cu, int insts) {
    const int iters = 1 << 15, grid = 256 * wg_per_cu;
    float *out;
    (void)hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    k<MODE><<<grid, 256>>>(out, 16, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    k<MODE><<<grid, 256>>>(out, iters, 1.f);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    // cycles per instruction as one wave sees them (wall time of the kernel / instructions of ONE wave)
    printf("%-52s waves/SIMD=%d  %7.3f ms  %6.2f cycles per instruction of a wave (2.4 GHz)\n", name, wg_per_cu, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * insts));
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2}) {
        run<0>("8 independent v_fma_f32", w, 8);
        run<1>("8 dependent v_fma_f32", w, 8);
        run<2>("8 v_med3_f32 on neighbours", w, 8);
        run<3>("8 v_cmp_lt_f32_e64 -> distinct scalar pairs", w, 8);
        run<4>("8 v_cmp_lt_f32_e32 -> vcc", w, 8);
        run<5>("8 v_cmp_e64 then 8 v_cndmask on them", w, 16);
        run<6>("8 x (v_cmp -> vcc -> v_cndmask)", w, 16);
        run<7>("8 v_cndmask_b32_e64 on an old mask", w, 8);
        run<8>("8 x (address, ds_write_b64, v_addc carry-in)", w, 24);
        run<9>("8 x (v_cmp, s_and_saveexec, v_add, s_or exec)", w, 32);
        run<10>("8 v_mov_b32", w, 8);
    }
    return 0;
}
