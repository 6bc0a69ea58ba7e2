// Per-thread error record and library info for libpcc_structural.so.
#include "pcc_common.hpp"

#include <atomic>

#include <mutex>
#include <string>
#include <vector>

namespace {
thread_local int t_status = 0;
thread_local char t_msg[320] = "";
}  // namespace

namespace {
int g_tuning[16] = {};
}

namespace pcc {
// compute units of the CURRENT device (cached per device: a process may drive several)
int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
            (void)hipGetLastError();
            v = 0;
        }
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

int tuning(int key) { return key >= 0 && key < 16 ? __atomic_load_n(&g_tuning[key], __ATOMIC_RELAXED) : 0; }
void set_error(int status, const char *what) {
    t_status = status;
    std::strncpy(t_msg, what ? what : "", sizeof t_msg - 1);
    t_msg[sizeof t_msg - 1] = '\0';
}
void clear_error() {
    t_status = 0;
    t_msg[0] = '\0';
}

namespace {
struct ProfRec {
    const char *name;
    hipEvent_t a, b;
};
int g_prof_mode = 0;  // 0 off, 1 every kernel launch, 2 only the coarse scopes (launch sequences)
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
void prof_clear() {
    for (auto &r : g_prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
}
}  // namespace

namespace {
std::mutex g_pool_mu;
hipMemPool_t g_pools[64] = {};
bool g_pool_failed[64] = {};
hipMemPool_t device_pool() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pools[dev] && !g_pool_failed[dev]) {
        hipMemPoolProps props{};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess) {
            uint64_t keep = kPoolKeepBytes;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            g_pools[dev] = pool;
        } else {
            (void)hipGetLastError();
            g_pool_failed[dev] = true;  // fall back to the default pool, untouched
        }
    }
    return g_pools[dev];
}
}  // namespace

hipError_t ws_malloc(void **p, size_t bytes, hipStream_t st) {
    hipMemPool_t pool = device_pool();
    hipError_t e = pool ? hipMallocFromPoolAsync(p, bytes, pool, st) : hipMallocAsync(p, bytes, st);
    if (e != hipSuccess) {
        *p = nullptr;
        (void)hipGetLastError();
        return e;
    }
    // PCC_WS_POISON=1 (test runs): every workspace starts as 0xFF bytes -- NaN as a float, -1 as an index -- so a kernel
    // that reads scratch nobody wrote shows up as a NaN / a fault instead of passing on whatever the pool held before
    static const bool poison = [] {
        const char *v = std::getenv("PCC_WS_POISON");
        return v && v[0] == '1';
    }();
    if (poison) (void)hipMemsetAsync(*p, 0xFF, bytes, st);
    return e;
}
hipError_t ws_free(void *p, hipStream_t st) { return p ? hipFreeAsync(p, st) : hipSuccess; }

bool profiling() { return g_prof_mode != 0; }
ProfScope::ProfScope(const char *kernel, hipStream_t s, bool coarse, bool enabled) : st(s), name(kernel) {
    if (!enabled || g_prof_mode != (coarse ? 2 : 1)) return;
    if (hipEventCreate(&start) != hipSuccess) {
        start = nullptr;
        return;
    }
    (void)hipEventRecord(start, st);
}
ProfScope::~ProfScope() {
    if (!start) return;
    hipEvent_t stop;
    if (hipEventCreate(&stop) != hipSuccess) {
        (void)hipEventDestroy(start);
        return;
    }
    (void)hipEventRecord(stop, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({name, start, stop});
}
}  // namespace pcc

extern "C" {
void pcc_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(pcc::g_prof_mu);
    pcc::g_prof_mode = on;
    if (on) pcc::prof_clear();
}
void pcc_profile_reset(void) {
    std::lock_guard<std::mutex> lk(pcc::g_prof_mu);
    pcc::prof_clear();
}
int pcc_profile_read(const char *kernel_prefix, double *avg_us, int *launches) {
    std::lock_guard<std::mutex> lk(pcc::g_prof_mu);
    const std::string pre = kernel_prefix ? kernel_prefix : "";
    double total_ms = 0;
    int n = 0;
    for (auto &r : pcc::g_prof) {
        if (std::string(r.name).compare(0, pre.size(), pre) != 0) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        total_ms += ms;
        n++;
    }
    if (avg_us) *avg_us = n ? total_ms * 1e3 / n : 0.0;
    if (launches) *launches = n;
    return n ? PCC_OK : PCC_EINVAL;
}

const char *pcc_version(void) { return "pcc_structural 0.1 (gfx950)"; }
const char *pcc_last_error(void) { return t_msg; }
int pcc_last_status(void) { return t_status; }
}

#include <cstdlib>
#include "pcc_test_hooks.h"
extern "C" int pcc_test_set_tuning(int key, int value) {
    static const bool armed = [] {
        const char *e = std::getenv("PCC_TEST_HOOKS");
        return e && e[0] == '1';
    }();
    if (!armed || key < 0 || key >= 16) return 0;
    __atomic_store_n(&g_tuning[key], value, __ATOMIC_RELAXED);
    return 1;
}
