// Micro-benchmark: what does one dependent kernel launch cost on this runtime?  A chain of K launches on one stream, each
// doing (almost) nothing, timed with hipEvents: empty grid of 1 workgroup, 512 workgroups x 512 threads that return at once,
// and 512 workgroups that read one float4 per thread written by the previous launch (cold first access after the
// kernel boundary) and write one.
// Build: hipcc -O3 --offload-arch=gfx950 -w tools/launch_gap.hip -o tools/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ __launch_bounds__(512) void k_touch(const float4 *in, float4 *out) {
    const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    float4 v = in[i];
    v.x += 1.f;
    out[i] = v;
}
__global__ __launch_bounds__(512) void k_touch_sync(const float4 *in, float4 *out) {
    __shared__ float s[512];
    const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    float4 v = in[i];
    s[threadIdx.x] = v.x;
    __syncthreads();
    v.x = s[511 - threadIdx.x] + 1.f;
    out[i] = v;
}
__global__ __launch_bounds__(512) void k_touch_n(const float *in, float *out, int n) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.f;
}
int main() {
    const int K = 400;
    float4 *a, *b;
    hipMalloc(&a, 512 * 512 * 16);
    hipMalloc(&b, 512 * 512 * 16);
    hipMemset(a, 0, 512 * 512 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; i++) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < K; i++) launch(i);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %6.2f us per launch\n", name, ms / K * 1e3);
    };
    run("empty kernel, 1 workgroup", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); });
    run("empty kernel, 512 workgroups x 512 threads", [&](int) { hipLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, 0); });
    run("512 x 512 threads: one float4 in, one out (ping-pong)", [&](int i) {
        hipLaunchKernelGGL(k_touch, dim3(512), dim3(512), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b);
    });
    run("the same through LDS and a workgroup barrier", [&](int i) {
        hipLaunchKernelGGL(k_touch_sync, dim3(512), dim3(512), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b);
    });
    for (int n : {262144, 65536, 16384, 4096, 512}) {
        char name[96];
        snprintf(name, sizeof name, "512 x 512 threads, the first %d read and write one float", n);
        run(name, [&](int i) {
            hipLaunchKernelGGL(k_touch_n, dim3(512), dim3(512), 0, 0, reinterpret_cast<const float *>((i & 1) ? b : a),
                               reinterpret_cast<float *>((i & 1) ? a : b), n);
        });
    }
    return 0;
}
