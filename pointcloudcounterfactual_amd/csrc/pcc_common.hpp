// Shared host/device helpers for libpcc_structural.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "pcc_structural.h"

namespace pcc {

constexpr int kWave = 64;  // CDNA wavefront

// Per-thread last-error record behind pcc_last_error() / pcc_last_status().
void set_error(int status, const char *what);
void clear_error();

inline int check_launch(const char *what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "HIP kernel failed : %d (%s) in %s", (int)err, hipGetErrorString(err), what);
        set_error((int)err, buf);
        return (int)err;
    }
    return PCC_OK;
}

inline int invalid(const char *what) {
    set_error(PCC_EINVAL, what);
    return PCC_EINVAL;
}

// Squared norm of a difference vector, in the one rounding order shared with the CPU oracle
// (oracle/structural_oracle.c sqsum3 mode 0).  The translation unit is built with
// -ffp-contract=off, so `y * y` stays a rounded multiply and nothing else is fused.
__device__ __forceinline__ float sq3(float x, float y, float z) {
    return __builtin_fmaf(z, z, __builtin_fmaf(x, x, y * y));
}

// A/B switches for measurements inside ONE process (include/pcc_test_hooks.h: pcc_test_set_tuning; inert without
// PCC_TEST_HOOKS=1): the value of switch `key` (0 = the product's behaviour).
int tuning(int key);

// Compute units of the CURRENT device (cached per device: a process may drive several); 0 if the query fails.
int device_cus();

// Optional hipEvent bracket around one kernel launch (pcc_profile_* in the C ABI).
bool profiling();
struct ProfScope {
    hipEvent_t start = nullptr;
    hipStream_t st;
    const char *name;
    // coarse scopes bracket a whole launch sequence and are recorded only in mode 2 (no per-launch events inside)
    ProfScope(const char *kernel, hipStream_t s, bool coarse = false, bool enabled = true);
    ~ProfScope();
};

__host__ __device__ constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Stream-ordered workspace of the library: one PRIVATE memory pool per device (hipMemPoolCreate), so that nothing is
// configured on the device's default pool, which belongs to the application (PyTorch, RCCL ...).  Freed blocks stay
// cached in the pool up to kPoolKeepBytes between calls; more than that is handed back at the next synchronisation.
constexpr unsigned long long kPoolKeepBytes = 1ull << 30;
hipError_t ws_malloc(void **p, size_t bytes, hipStream_t st);
hipError_t ws_free(void *p, hipStream_t st);

// The Chamfer half of pcc_chamfer_emd: computed inside the approximate-EMD call, on the clouds that call sorts
// (approxmatch.hip); the loss reduction rides in that call's finish launch.
struct ChamferOut {
    int mean;
    float *loss, *dist1, *dist2;
    int *idx1, *idx2;
};
int match_cost_with_chamfer(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1,
                            float *grad2, hipStream_t st, const ChamferOut &chamfer);

// Logical block id such that the blocks of one XCD (equal blockIdx % 8: the dispatcher deals workgroups round-robin over the
// 8 XCDs) hold a contiguous run of logical ids.  Bijective on [0, nwg); consecutive logical ids -- the workgroups of one
// sample -- then share an L2.  Placement is a speed matter only: nothing may depend on it.
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
    if (nwg <= 8) return bid;
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}

// Hilbert sort of one channels-major cloud per sample (approxmatch.hip's sort kernel) for the k-NN graph: aos [b][n]
// (x, y, z, original index as bits), box16 [b][ceil(n/16)][8] (lo xyz, pad, hi xyz, pad), perm [b][n] sorted -> original.
constexpr int kSortBox = 16;
int sort_cloud_cmajor(int b, int c, int n, const float *x, float4 *aos, float *box16, int *perm, hipStream_t st);

}  // namespace pcc
