// Per-lane running top-K (smallest distances) for the kNN kernels, wave64.
//
// A lane owns one query.  Keeping a sorted K-list per lane and inserting whenever a candidate beats the
// K-th distance wastes the wave: with 64 lanes some lane almost always inserts, so the 5K-instruction
// insertion chain would run for ~90 % of the candidates although each lane truly inserts only
// ~K(1+ln(N/K)) times.  Instead qualifying candidates are appended to a small per-lane FIFO in LDS
// (2 ds_write) against a threshold that is only refreshed at flush time; the chain runs when some lane's
// FIFO is nearly full, once per buffered entry, for all lanes together.
//
// Ordering contract: ascending distance, and among equal distances ascending candidate index
// (candidates are appended in ascending index order, the FIFO is drained in order and the chain uses a
// strict '<').
#pragma once
#include <hip/hip_runtime.h>

namespace pcc {

template <int K>
struct TopK {
    float d[K];
    int i[K];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < K; s++) {
            d[s] = __builtin_inff();
            i[s] = 0x7fffffff;
        }
    }
    // Insert (x,id) keeping the list sorted; the displaced tail falls off.  Strict '<' for the NEW element (an equal
    // distance that arrived earlier stays in front); an entry behind a displaced one shifts whatever its own value, so
    // displaced entries keep their relative order too (equal distances inside the list).
    // Every slot is computed from the OLD list, independently of the other slots -- no value is carried from slot to
    // slot, so the K compares and selects overlap instead of forming one dependent chain (a compare's mask takes tens
    // of cycles to reach the select that consumes it: tools/cmp_bench.hip):
    //   d'[s] = median(d[s-1], x, d[s])            (one v_med3_f32: the list is sorted)
    //   i'[s] = x < d[s-1] ? i[s-1] : x < d[s] ? id : i[s]
    // Distances in the list are never NaN (offers compare '<' against the threshold first).
    __device__ __forceinline__ void insert(float x, int id) {
        bool lt[K];
#pragma unroll
        for (int s = 0; s < K; s++) lt[s] = x < d[s];
#pragma unroll
        for (int s = K - 1; s > 0; s--) {  // downwards: slot s - 1 still holds its old value
            i[s] = lt[s - 1] ? i[s - 1] : (lt[s] ? id : i[s]);
            d[s] = __builtin_amdgcn_fmed3f(d[s - 1], x, d[s]);
        }
        i[0] = lt[0] ? id : i[0];
        d[0] = lt[0] ? x : d[0];
    }
    __device__ __forceinline__ float worst() const { return d[K - 1]; }
};

// Per-lane FIFO in LDS, laid out [slot][thread] (conflict-free: a wave touches 64 consecutive words).
template <int K, int CAP, int T>
struct BufferedTopK {
    TopK<K> top;
    float thr;   // threshold used while buffering (K-th distance at the last flush)
    int cnt;     // entries in this lane's FIFO
    float *bd;   // [CAP][T]
    int *bi;     // [CAP][T]
    int tid;

    __device__ __forceinline__ void init(float *buf_d, int *buf_i, int thread) {
        top.init();
        thr = __builtin_inff();
        cnt = 0;
        bd = buf_d;
        bi = buf_i;
        tid = thread;
    }
    __device__ __forceinline__ void offer(float x, int id) {
        if (x < thr) {
            bd[cnt * T + tid] = x;
            bi[cnt * T + tid] = id;
            cnt++;
        }
    }
    // true when another block of `next` offers could overflow some lane's FIFO (wave-uniform)
    __device__ __forceinline__ bool must_flush(int next) const { return __any(cnt > CAP - next); }

    __device__ __forceinline__ void flush() {
        for (int t = 0; t < CAP; t++) {
            if (!__any(t < cnt)) break;
            const bool live = t < cnt;
            const float x = live ? bd[t * T + tid] : __builtin_inff();
            const int id = live ? bi[t * T + tid] : 0x7fffffff;
            if (x < top.worst()) top.insert(x, id);
        }
        cnt = 0;
        thr = top.worst();
    }
};

}  // namespace pcc
