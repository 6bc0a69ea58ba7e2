// k-nearest-neighbour graph for gfx950 (MI355X), wave64.
//
// Replaces knn / pykeops_knn (reference src/utils/neighbour_ops.py:63-82: a PyKeOps argKmin over a lazy
// (B,N,N) squared-distance tensor) -- PyKeOps has no ROCm backend.  Four kernels:
//   * knn_sorted_kernel (c <= 3, n <= 16384): exact difference-form distances on the f32 VALU, the formula the GPU
//     reference evaluates (pykeops_square_distance, :35-40), on the Hilbert-sorted cloud: only the candidate boxes
//     that can still hold one of a query's k nearest are visited (see the kernel).
//   * knn_small_kernel (c <= 3, larger clouds): the same distances, exhaustive.  A lane owns one query; the candidate cloud sits
//     in LDS as SoA rows (x is already channels-major, so staging is a straight coalesced copy); S waves
//     scan disjoint candidate ranges; per-lane buffered top-K (topk.hpp); the S sorted lists are merged
//     with strict '<' in range order, so equal distances come out in ascending index order.
//   * knn_mfma_kernel (c >= 4): expanded form |xi|^2 + |xj|^2 - 2 xi.xj with the inner product on
//     v_mfma_f32_32x32x2_f32 (exact f32 FMA chain per output) -- the formula of the reference's CPU path
//     (self_square_distance, :53-60) and a genuine dense contraction (C = 64..128 in the DGCNN encoder).
//     The 32x32 accumulator tile is oriented with the QUERY on the lane (column) and 16 candidates in
//     the lane's accumulator registers, so the same per-lane top-K consumes distances straight from
//     registers: the (B,N,N) distance matrix never exists in memory.
//   * knn_mfma_split_kernel (c >= 4, launches that fill the chip): the same arithmetic with the matrix waves and the
//     selection waves of a workgroup apart, one query per selection lane (see the kernel).
#include "pcc_common.hpp"
#include "pcc_neighbour.h"
#include "pcc_test_hooks.h"
#include "topk.hpp"

namespace {

constexpr int kCap = 16;     // FIFO slots per lane
constexpr int kCH = 2048;    // candidates staged per chunk (small-c kernel)
constexpr int kSortBoxK = pcc::kSortBox;
constexpr int kTT128 = 1;
constexpr int kSortedMaxN = 16384;  // the sort kernel orders up to 16384 points per cloud    // tiles per stage of the 128-channel MFMA instantiation

template <int K, int S>
struct SmallLayout {
    static constexpr int T = 64 * S;
    static constexpr int cand_bytes = 3 * kCH * 4;
    static constexpr int buf_bytes = 2 * kCap * T * 4;
    static constexpr int merge_bytes = 2 * S * K * 64 * 4;
    static constexpr int bytes = (cand_bytes + buf_bytes) > merge_bytes ? (cand_bytes + buf_bytes) : merge_bytes;
};

// Merge S sorted K-lists per lane (LDS layout [s][slot][lane]) and write the first k indices as int64.
template <int K, int S>
__device__ __forceinline__ void merge_and_store(const float *md, const int *mi, int lane, int k, int64_t *dst, int n) {
    int p[S];
    float h[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        p[s] = 0;
        h[s] = md[(s * K) * 64 + lane];
    }
    for (int o = 0; o < k; o++) {
        int best = 0;
        float bv = h[0];
#pragma unroll
        for (int s = 1; s < S; s++) {
            const bool lt = h[s] < bv;  // strict: the earlier candidate range wins ties
            bv = lt ? h[s] : bv;
            best = lt ? s : best;
        }
        int pos = 0;
#pragma unroll
        for (int s = 0; s < S; s++) pos = (best == s) ? p[s] : pos;
        // (a list can run out only when distances are NaN: never emit an index outside the cloud)
        dst[o] = (int64_t)min(mi[(best * K + pos) * 64 + lane], n - 1);
        const int np = pos + 1;
        const float nh = np < K ? md[(best * K + np) * 64 + lane] : __builtin_inff();
#pragma unroll
        for (int s = 0; s < S; s++) {
            const bool sel = best == s;
            p[s] = sel ? np : p[s];
            h[s] = sel ? nh : h[s];
        }
    }
}

template <int K, int S>
__global__ __launch_bounds__(64 * S) void knn_small_kernel(int c, int n, int k, const float *__restrict__ x,
                                                            int64_t *__restrict__ indices) {
    using L = SmallLayout<K, S>;
    constexpr int T = L::T;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::bytes];
    float *lds_c = reinterpret_cast<float *>(smem);
    float *buf_d = reinterpret_cast<float *>(smem + L::cand_bytes);
    int *buf_i = reinterpret_cast<int *>(smem + L::cand_bytes + kCap * T * 4);
    float *mrg_d = reinterpret_cast<float *>(smem);
    int *mrg_i = reinterpret_cast<int *>(smem + S * K * 64 * 4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int smp = blockIdx.y;
    const float *xb = x + (size_t)smp * c * n;
    int q = blockIdx.x * 64 + lane;
    const bool q_ok = q < n;
    q = q_ok ? q : n - 1;
    const float qx = xb[q];
    const float qy = c > 1 ? xb[(size_t)n + q] : 0.f;
    const float qz = c > 2 ? xb[(size_t)2 * n + q] : 0.f;

    pcc::BufferedTopK<K, kCap, T> tk;
    tk.init(buf_d, buf_i, tid);

    const float4 *X4 = reinterpret_cast<const float4 *>(lds_c);
    const float4 *Y4 = X4 + kCH / 4;
    const float4 *Z4 = Y4 + kCH / 4;

    for (int c0 = 0; c0 < n; c0 += kCH) {
        const int cnt = min(kCH, n - c0);
        const int ngroups = (cnt + 7) / 8;
        if (c0) __syncthreads();
        for (int ch = 0; ch < 3; ch++) {
            for (int i = tid; i < ngroups * 8; i += T)
                lds_c[ch * kCH + i] = (i < cnt) ? (ch < c ? xb[(size_t)ch * n + c0 + i] : 0.f) : __builtin_inff();
        }
        __syncthreads();
        const int gs = (ngroups + S - 1) / S;
        const int g_begin = w * gs;
        const int g_end = min(g_begin + gs, ngroups);
        for (int g = g_begin; g < g_end; g++) {
            const float4 xa = X4[2 * g], xb4 = X4[2 * g + 1];
            const float4 ya = Y4[2 * g], yb4 = Y4[2 * g + 1];
            const float4 za = Z4[2 * g], zb4 = Z4[2 * g + 1];
            const float cx[8] = {xa.x, xa.y, xa.z, xa.w, xb4.x, xb4.y, xb4.z, xb4.w};
            const float cy[8] = {ya.x, ya.y, ya.z, ya.w, yb4.x, yb4.y, yb4.z, yb4.w};
            const float cz[8] = {za.x, za.y, za.z, za.w, zb4.x, zb4.y, zb4.z, zb4.w};
            if (tk.must_flush(8)) tk.flush();
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float dx = cx[j] - qx, dy = cy[j] - qy, dz = cz[j] - qz;
                // sum over channels in channel order: ((dx^2 + dy^2) + dz^2) as an fma chain
                const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                tk.offer(d, c0 + g * 8 + j);
            }
        }
    }
    tk.flush();
    __syncthreads();  // every wave is done with the candidate / FIFO regions: reuse them for the merge
#pragma unroll
    for (int s = 0; s < K; s++) {
        mrg_d[(w * K + s) * 64 + lane] = tk.top.d[s];
        mrg_i[(w * K + s) * 64 + lane] = tk.top.i[s];
    }
    __syncthreads();
    if (w == 0 && q_ok) merge_and_store<K, S>(mrg_d, mrg_i, lane, k, indices + ((size_t)smp * n + q) * k, n);
}

// ---------------------------------------------------------------------------------------------------
// c <= 3, n <= 16384: search on the Hilbert-sorted cloud.
// The exhaustive kernel above spends > 90 % of its time in the top-K insertion chains: candidates arrive in index
// order, so a lane's K-th distance keeps improving all through the scan (K(1 + ln(N/K)) insertions per list, four lists
// per query).  Here the cloud is first put in Hilbert order with one bounding box per 16 consecutive points (the sort
// kernel of the approximate EMD, approxmatch.hip).  A WAVE owns one box of 16 consecutive sorted queries and works
// alone: lane = (query, candidate slice), the four slices of a query take four candidates each of every 16-candidate
// block and keep their own sorted K-list.
//   * the candidate blocks of a window are ordered by the distance between their box and the queries' box and visited
//     nearest first, so the lists fill with near points at once;
//   * a query's bound: if each of its four slice lists holds at least ceil(k/4) entries <= t, at least k candidates are
//     <= t, so its k-th distance is <= the largest of the four slices' ceil(k/4)-th entries.  Candidates beyond the bound
//     are not even buffered; the first block whose box is farther than the largest bound of the 16 queries ends the
//     walk (exact: nothing that could enter a list, or tie with a lower index, is skipped);
//   * list entries are 64-bit keys (distance bits : ORIGINAL index): squared distances are non-negative floats, which
//     order like unsigned integers, so one 64-bit compare is "ascending distance, ties ascending index" although
//     candidates no longer arrive in index order;
//   * the four slice lists of a query are merged at the end (LDS, one lane per query).
// ---------------------------------------------------------------------------------------------------
constexpr int kSW = 4;     // independent waves per workgroup (no barrier; the workgroup only shares the LDS allocation)
constexpr int kSQ = 16;    // queries per wave = one box of the sort
constexpr int kSlices = 4; // candidate slices per query
constexpr unsigned long long kKeyInf = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffull;

// One step of the insertion chain: (slot, carry) <- (min, max) of the two 64-bit keys.  Keys are unique, so once the
// carry displaces an entry everything behind shifts.  One v_cmp_lt_u64 and four v_cndmask_b32 on ITS mask (written as
// asm: the compiler turns the two selects into separate unsigned min / max, i.e. two of the slow 64-bit compares).
__device__ __forceinline__ void ce_step(unsigned long long &slot, unsigned long long &carry) {
    const unsigned long long m = __ballot(carry < slot);
    const unsigned sl = (unsigned)slot, sh = (unsigned)(slot >> 32), cl = (unsigned)carry, ch = (unsigned)(carry >> 32);
    unsigned nsl, nsh, ncl, nch;
    // (s_nop: a VALU-written SGPR pair needs two wait states before a VALU reads it as a mask)
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %4, %6, %8\n\tv_cndmask_b32_e64 %1, %5, %7, %8\n\t"
        "v_cndmask_b32_e64 %2, %6, %4, %8\n\tv_cndmask_b32_e64 %3, %7, %5, %8"
        : "=&v"(nsl), "=&v"(nsh), "=&v"(ncl), "=&v"(nch)
        : "v"(sl), "v"(sh), "v"(cl), "v"(ch), "s"(m));
    slot = ((unsigned long long)nsh << 32) | nsl;
    carry = ((unsigned long long)nch << 32) | ncl;
}

struct KnnSortedArgs {
    int n, nb, batch, k;
    const float4 *aos;   // [b][n] (x, y, z, original index) per sorted point (+ padding, see pcc_knn)
    const float *box;    // [b][nb][8]
    const int *perm;     // [b][n]
    int64_t *out;        // [b][n][k] in the caller's point order
};

// K = list slots (>= k), KPREV = the next smaller instantiation (k > KPREV)
template <int K, int KPREV>
__global__ __launch_bounds__(64 * kSW, (K == 16 ? 3 : K <= 25 ? 4 : 1)) void knn_sorted_kernel(KnnSortedArgs a) {  // (<= 128 VGPRs up to K = 25: four waves per SIMD; the carried chain of K = 16 needs 3 to stay out of scratch)
    // per wave: FIFO [kCap][64] x (distance, index), reused as the merge area [slice][K][16] x (distance 4 B | index 2 B:
    // n <= 16384) -- 6 bytes per entry keep K = 25 under 10 KB per wave, i.e. four waves per SIMD
    constexpr int kEntries = kSlices * K * kSQ;
    constexpr int kMergeWords = kEntries + (kEntries + 1) / 2;
    constexpr int kWaveWords = (2 * kCap * 64) > kMergeWords ? (2 * kCap * 64) : kMergeWords;
    __shared__ __attribute__((aligned(8))) unsigned smem[kSW * kWaveWords];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = (int)blockIdx.x * kSW + w;  // global wave = (sample, box)
    const int smp = gw / a.nb, grp = gw - smp * a.nb;
    if (smp >= a.batch) return;  // (whole wave; the kernel has no barrier)
    unsigned *wbase = smem + w * kWaveWords;
    float *buf_d = reinterpret_cast<float *>(wbase);
    int *buf_i = reinterpret_cast<int *>(wbase + kCap * 64);
    const int n = a.n, k = a.k;
    const int ql = lane & (kSQ - 1), cs = lane >> 4;
    const float4 *C = a.aos + (size_t)smp * n;
    const int qs = min(grp * kSQ + ql, n - 1);
    const float4 me = C[qs];
    const float4 *gb = reinterpret_cast<const float4 *>(a.box + ((size_t)smp * a.nb + grp) * 8);
    const float4 glo = gb[0], ghi = gb[1];

    // ascending; the list lives in the LAST k slots (the first K - k hold key 0, which nothing displaces), so that the
    // k-th entry is the static register pair key[K - 1]
    unsigned long long key[K];
#pragma unroll
    for (int s = 0; s < K; s++) key[s] = s < K - k ? 0ull : kKeyInf;
    // the slot whose entry bounds the query's k-th distance (see above): rank ceil(k/4) of the slice when k == K,
    // otherwise a static slot that has at least that rank for every k in (KPREV, K]
    // The insertion pass in its carry-free form (every slot from the old list) needs ~30 fewer VGPRs than the chain that
    // carries the displaced key from slot to slot, which decides the occupancy at K = 20 and 25 (four waves per SIMD
    // together with the 6-byte merge entries; surface clouds: 118 -> 104 us and 139 -> 133 us, Gaussian 231 -> 197 us at
    // K = 25) and is worth a few per cent at K <= 8; at K = 16 and 32 the carried chain measured faster (90 / 164 us
    // against 100 / 193) -- there the launch bound alone (128 VGPRs up to K = 25) is what helps (K = 16: 101 -> 90 us).
    constexpr bool kCarryFree = K <= 8 || K == 20 || K == 25;
    constexpr int kTight = K - 1 - (3 * K) / 4, kLoose = K - 1 - (3 * (KPREV + 1)) / 4;
    unsigned long long thr = kKeyInf;  // buffering threshold: min(own k-th key, the query's bound) at the last flush
    int cnt = 0;
    float r = __builtin_inff();        // the largest bound of the 16 queries

    auto flush = [&]() {
        for (int t = 0; t < kCap; t++) {
            if (!__any(t < cnt)) break;
            unsigned long long x = kKeyInf;
            if (t < cnt) x = ((unsigned long long)__float_as_uint(buf_d[t * 64 + lane]) << 32) | (unsigned)buf_i[t * 64 + lane];
            if (x < key[K - 1]) {
                if (kCarryFree) {
                    // every slot from the OLD list: key'[s] = x < key[s-1] ? key[s-1] : x < key[s] ? x : key[s]
                    bool lt[K];
#pragma unroll
                    for (int s = 0; s < K; s++) lt[s] = x < key[s];
#pragma unroll
                    for (int s = K - 1; s > 0; s--) key[s] = lt[s - 1] ? key[s - 1] : (lt[s] ? x : key[s]);
                    key[0] = lt[0] ? x : key[0];
                } else {
#pragma unroll
                    for (int s = 0; s < K; s++) ce_step(key[s], x);
                }
            }
        }
        cnt = 0;
        // the query's bound over its four slices, then the largest over the 16 queries (distance bits order like ints)
        int qb = (int)((k == K ? key[kTight] : key[kLoose]) >> 32);
        qb = max(qb, __shfl_xor(qb, 16, 64));
        qb = max(qb, __shfl_xor(qb, 32, 64));
        const unsigned long long bound = ((unsigned long long)(unsigned)qb << 32) | 0x7fffffffull;
        thr = key[K - 1] < bound ? key[K - 1] : bound;
        int m = grp * kSQ + ql < n ? qb : 0;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
        r = __int_as_float(__builtin_amdgcn_readfirstlane(m));
    };

    for (int b0 = 0; b0 < a.nb; b0 += 128) {  // windows of 128 candidate blocks
        unsigned bkey[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int blk = b0 + lane + 64 * h;
            bkey[h] = 0xffffffffu;
            if (blk < a.nb) {
                const float4 *cb = reinterpret_cast<const float4 *>(a.box + ((size_t)smp * a.nb + blk) * 8);
                const float4 lo = cb[0], hi = cb[1];
                const float dx = fmaxf(fmaxf(glo.x - hi.x, lo.x - ghi.x), 0.f);
                const float dy = fmaxf(fmaxf(glo.y - hi.y, lo.y - ghi.y), 0.f);
                const float dz = fmaxf(fmaxf(glo.z - hi.z, lo.z - ghi.z), 0.f);
                // the candidates' own fma chain on the box gaps (every step is monotone: a true lower bound in f32), its 7
                // lowest mantissa bits replaced by the slot: truncation only lowers the bound
                const float lb = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                bkey[h] = (__float_as_uint(lb) & ~127u) | (unsigned)(lane + 64 * h);
            }
        }
#pragma unroll
        for (int kk = 2; kk <= 128; kk <<= 1) {  // ascending bitonic sort of the wave's 128 keys (element lane + 64 h)
#pragma unroll
            for (int j = kk >> 1; j > 0; j >>= 1) {
                if (j == 64) {
                    const unsigned mn = min(bkey[0], bkey[1]), mx = max(bkey[0], bkey[1]);
                    bkey[0] = mn;
                    bkey[1] = mx;
                } else {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int i = lane + 64 * h;
                        const unsigned other = (unsigned)__shfl_xor((int)bkey[h], j, 64);
                        const bool take_min = ((i & j) == 0) == ((i & kk) == 0);
                        bkey[h] = take_min ? min(bkey[h], other) : max(bkey[h], other);
                    }
                }
            }
        }
        const int nwin = min(128, a.nb - b0);
        auto key_at = [&](int p) -> unsigned {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)bkey[0], p & 63);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)bkey[1], p & 63);
            return p < 64 ? lo : hi;
        };
        // a lane's four candidates of a block are 64 contiguous bytes; the next block's are in flight while this one is
        // consumed.  Rows past the cloud's end are loaded (the workspace is padded) and never offered.
        auto load4 = [&](float4 (&v)[4], int c0) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = C[c0 + cs * 4 + j];
        };
        unsigned bk = key_at(0);
        float4 cur[4], nxt[4];
        load4(cur, (b0 + (int)(bk & 127u)) * kSortBoxK);
        for (int p = 0; p < nwin; p++) {
            if (__uint_as_float(bk & ~127u) > r) break;  // everything behind is farther still
            const int c0 = (b0 + (int)(bk & 127u)) * kSortBoxK;
            const unsigned bk_next = key_at(min(p + 1, nwin - 1));
            load4(nxt, (b0 + (int)(bk_next & 127u)) * kSortBoxK);
            if (__any(cnt > kCap - 4)) flush();
            const int left = n - c0 - cs * 4;  // real candidates from cur[0] on
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float dx = cur[j].x - me.x, dy = cur[j].y - me.y, dz = cur[j].z - me.z;
                const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                const unsigned long long x = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(cur[j].w);
                if (j < left && x < thr) {  // (false for NaN distances: their bits sort above +inf)
                    buf_d[cnt * 64 + lane] = d;
                    buf_i[cnt * 64 + lane] = __float_as_int(cur[j].w);
                    cnt++;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) cur[j] = nxt[j];
            if (__any(cnt > 0) && ((p & 7) == 7 || p < 8)) flush();  // fresh bounds: every block at first, then every 8th (measured)
            bk = bk_next;
        }
    }
    flush();

    // merge the four slice lists of every query: [slice][slot][query] keys in the wave's LDS region (the FIFO is drained)
    unsigned *md = wbase;                                                        // distance bits
    unsigned short *mi = reinterpret_cast<unsigned short *>(wbase + kEntries);  // original index (0xffff: the empty-slot sentinel)
    auto merged_key = [&](int e) -> unsigned long long {
        const unsigned i16 = mi[e];
        return ((unsigned long long)md[e] << 32) | (i16 == 0xffffu ? 0x7fffffffu : i16);
    };
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < K; s++) {
        const unsigned lo = (unsigned)key[s];
        md[(cs * K + s) * kSQ + ql] = (unsigned)(key[s] >> 32);
        mi[(cs * K + s) * kSQ + ql] = (unsigned short)(lo == 0x7fffffffu ? 0xffffu : lo);
    }
    __builtin_amdgcn_wave_barrier();
    if (cs == 0 && grp * kSQ + ql < n) {
        int64_t *dst = a.out + ((size_t)smp * n + a.perm[(size_t)smp * n + qs]) * k;
        int pos[kSlices];
        unsigned long long h[kSlices];
#pragma unroll
        for (int s = 0; s < kSlices; s++) {
            pos[s] = K - k;
            h[s] = merged_key((s * K + (K - k)) * kSQ + ql);
        }
        for (int o = 0; o < k; o++) {
            int best = 0;
            unsigned long long bv = h[0];
#pragma unroll
            for (int s = 1; s < kSlices; s++) {
                const bool lt = h[s] < bv;
                bv = lt ? h[s] : bv;
                best = lt ? s : best;
            }
            // (the lists can run short only when distances are NaN: never emit an index outside the cloud)
            dst[o] = (int64_t)min((int)(bv & 0xffffffffull), n - 1);
            int np = 0;
#pragma unroll
            for (int s = 0; s < kSlices; s++) np = best == s ? pos[s] + 1 : np;
            const unsigned long long nh = np < K ? merged_key((best * K + np) * kSQ + ql) : kKeyInf;
#pragma unroll
            for (int s = 0; s < kSlices; s++) {
                const bool sel = best == s;
                pos[s] = sel ? np : pos[s];
                h[s] = sel ? nh : h[s];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// c >= 4: MFMA kernel.
// Workgroup = 4 waves, each wave 32 queries (columns of the 32x32 accumulator tile = lane & 31); the two
// half-waves hold different candidate rows of the tile (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so each
// query has two partial top-K lists which are merged at the end.  Candidate tiles [c][32] are staged in
// LDS once per workgroup and shared by the 4 waves.  B operand (queries) lives in c/2 VGPRs per lane.
// ---------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void sqnorm_kernel(int c, int n, const float *__restrict__ x, float *__restrict__ sq) {
    // sq[b][i] = sum_c x[b,c,i]^2 in channel order
    const int smp = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *xb = x + (size_t)smp * c * n;
    float s = 0.f;
    for (int ch = 0; ch < c; ch++) {
        const float v = xb[(size_t)ch * n + i];
        s = __builtin_fmaf(v, v, s);
    }
    sq[(size_t)smp * n + i] = s;
}

template <int K, int CP /* padded channels, multiple of 2, <= 128 */, int TT /* 32-candidate tiles per stage */>
__global__ __launch_bounds__(256, CP >= 128 ? 2 : 1) void knn_mfma_kernel(int c, int n, int k, const float *__restrict__ x,
                                                        const float *__restrict__ sq,
                                                        int64_t *__restrict__ indices) {
    constexpr int T = 256;
    constexpr int KS = CP / 2;  // MFMA k-steps (32x32x2)
    constexpr int TW = 32 * TT;  // candidates per stage
    constexpr int tile_bytes = CP * TW * 4;
    constexpr int buf_bytes = 2 * kCap * T * 4;
    constexpr int merge_bytes = 2 * 8 * K * 32 * 4;  // [wave(4)][half(2)][K][32 queries]
    constexpr int main_bytes = 2 * tile_bytes + 2 * TW * 4 + buf_bytes;
    constexpr int bytes = main_bytes > merge_bytes ? main_bytes : merge_bytes;
    __shared__ __attribute__((aligned(16))) unsigned char smem[bytes];
    float *tile = reinterpret_cast<float *>(smem);                       // [2][CP][TW]
    float *tsq = reinterpret_cast<float *>(smem + 2 * tile_bytes);      // [2][TW]
    float *buf_d = reinterpret_cast<float *>(smem + 2 * tile_bytes + 2 * TW * 4);
    int *buf_i = reinterpret_cast<int *>(smem + 2 * tile_bytes + 2 * TW * 4 + kCap * T * 4);
    float *mrg_d = reinterpret_cast<float *>(smem);
    int *mrg_i = reinterpret_cast<int *>(smem + 8 * K * 32 * 4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int smp = blockIdx.y;
    const float *xb = x + (size_t)smp * c * n;
    const float *sqb = sq + (size_t)smp * n;
    int q = blockIdx.x * 128 + w * 32 + col;
    const bool q_ok = q < n;
    q = q_ok ? q : n - 1;
    // B operand: query[col][k = 2*ks + half]
    float bq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const int ch = 2 * ks + half;
        bq[ks] = ch < c ? xb[(size_t)ch * n + q] : 0.f;
    }
    const float sq_q = sqb[q];

    pcc::BufferedTopK<K, kCap, T> tk;
    tk.init(buf_d, buf_i, tid);

    const int nstages = (n + TW - 1) / TW;
    // Staging is software-pipelined by hand: the global loads of stage t+1 are all issued (unconditional, clamped
    // addresses; the select happens on the value) BEFORE the MFMAs of stage t and land in LDS after them.  Written as
    // a plain conditional copy loop the compiler emitted load -> s_waitcnt vmcnt(0) -> ds_write per element, i.e. one
    // exposed memory round trip per element (the kernel spent most of its time there).
    constexpr int E = CP * TW / T;  // elements per thread per stage
    static_assert(CP * TW % T == 0, "stage size must be a multiple of the workgroup");
    float pre[E], pre_sq = 0.f;
    // fetch only issues the loads (clamped addresses, nothing consumes the values); the out-of-range select happens at
    // commit time, behind the MFMAs -- a select next to the load makes the compiler wait for every load where it is issued.
    auto fetch = [&](int t) {
        const int j0 = t * TW;
#pragma unroll
        for (int i = 0; i < E; i++) {
            const int e = tid + i * T;
            const int ch = e / TW, j = e - ch * TW;
            pre[i] = xb[(size_t)min(ch, c - 1) * n + min(j0 + j, n - 1)];
        }
        pre_sq = sqb[min(j0 + (tid % TW), n - 1)];
    };
    auto commit = [&](int t) {
        const int j0 = t * TW;
        float *dst = tile + (t & 1) * CP * TW;
#pragma unroll
        for (int i = 0; i < E; i++) {
            const int e = tid + i * T;
            const int ch = e / TW, j = e - ch * TW;
            dst[e] = (ch < c && j0 + j < n) ? pre[i] : 0.f;
        }
        if (tid < TW) tsq[(t & 1) * TW + tid] = (j0 + tid < n) ? pre_sq : __builtin_inff();
    };
    // The two half-waves keep separate lists for the same query (lane and lane ^ 32).  Each list alone would keep
    // buffering until ITS K-th distance is beaten; but once both lists hold ceil(K/2) entries <= t, at least K candidates
    // are <= t, so nothing above t can reach the query's k nearest: after every drain the buffering threshold drops to
    // the larger of the two lists' ceil(K/2)-th entries (ties at t still pass: the threshold is the next float above t).
    auto flush_shared = [&]() {
        tk.flush();
        const float mine = tk.top.d[(K + 1) / 2 - 1];
        const float t = fmaxf(mine, __shfl_xor(mine, 32, 64));
        float up = t;  // next float above t (t is never NaN; +inf stays)
        if (t < __builtin_inff()) {
            const int bits = __float_as_int(t);
            up = t == 0.f ? __int_as_float(1) : __int_as_float(t > 0.f ? bits + 1 : bits - 1);
        }
        tk.thr = fminf(tk.thr, up);
    };
    fetch(0);
    commit(0);
    __syncthreads();
    for (int t = 0; t < nstages; t++) {
        const int slot = t & 1;
        if (t + 1 < nstages) fetch(t + 1);
        const float *cur = tile + slot * CP * TW;
        // TT independent accumulator chains: a dependent MFMA cannot issue before the previous one has left the
        // matrix pipe, so one chain per wave leaves the pipe idle half of the time (measured 166 cycles per
        // v_mfma_f32_32x32x2_f32 against the 64 it occupies)
        f32x16 acc[TT];
#pragma unroll
        for (int u = 0; u < TT; u++)
            acc[u] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // A operands are read from LDS eight k-steps ahead of the MFMAs that consume them (one ds_read + full wait
        // per MFMA left the matrix pipe idle for the LDS latency every step)
        constexpr int KB = KS < 4 ? KS : 4;
#pragma unroll
        for (int ks0 = 0; ks0 < KS; ks0 += KB) {
            float av[KB][TT];
#pragma unroll
            for (int kk = 0; kk < KB; kk++)
#pragma unroll
                for (int u = 0; u < TT; u++)
                    av[kk][u] = cur[(2 * (ks0 + kk) + half) * TW + u * 32 + col];  // candidate[row = lane&31][k] of sub-tile u
#pragma unroll
            for (int kk = 0; kk < KB; kk++)
#pragma unroll
                for (int u = 0; u < TT; u++)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][u], bq[ks0 + kk], acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < TT; u++) {
            if (tk.must_flush(16)) flush_shared();
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // candidate inside the stage
                // reference CPU path: dist = -2*dot ; dist += |xj|^2 (column term) ; dist += |xi|^2 (row term)
                const float d = (-2.0f * acc[u][r] + tsq[slot * TW + row]) + sq_q;
                tk.offer(d, t * TW + row);
            }
        }
        if (t + 1 < nstages) commit(t + 1);
        __syncthreads();
    }
    tk.flush();
    __syncthreads();
#pragma unroll
    for (int s = 0; s < K; s++) {
        mrg_d[((w * 2 + half) * K + s) * 32 + col] = tk.top.d[s];
        mrg_i[((w * 2 + half) * K + s) * 32 + col] = tk.top.i[s];
    }
    __syncthreads();
    if (half == 0 && q_ok) {
        // two-way merge of the half-wave lists; ties: lower candidate index first (the lists cover interleaved
        // row groups, so compare indices explicitly)
        const float *d0 = mrg_d + ((w * 2 + 0) * K) * 32 + col, *d1 = mrg_d + ((w * 2 + 1) * K) * 32 + col;
        const int *i0 = mrg_i + ((w * 2 + 0) * K) * 32 + col, *i1 = mrg_i + ((w * 2 + 1) * K) * 32 + col;
        int p0 = 0, p1 = 0;
        int64_t *dst = indices + ((size_t)smp * n + q) * k;
        for (int o = 0; o < k; o++) {
            const float a = p0 < K ? d0[p0 * 32] : __builtin_inff();
            const float bb = p1 < K ? d1[p1 * 32] : __builtin_inff();
            const int ia = p0 < K ? i0[p0 * 32] : 0x7fffffff;
            const int ib = p1 < K ? i1[p1 * 32] : 0x7fffffff;
            const bool take0 = (a < bb) || (a == bb && ia < ib);
            dst[o] = (int64_t)min(take0 ? ia : ib, n - 1);
            p0 += take0 ? 1 : 0;
            p1 += take0 ? 0 : 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// c >= 4, role-split form of the kernel above for launches that fill the chip with 256-query workgroups.
//
// What the selection costs is instructions (tools/issue_bench.hip, tools/mfma_coissue_bench.hip: a wave issues a VALU
// instruction every 5-8 cycles, a branch costs tens, and MFMAs of one wave and VALU work of another on the same SIMD
// add up rather than overlap): knn_mfma_kernel spends ~50 k of them per wave of 32 queries, 4/5 in the 5-instructions-
// per-slot insertion chains of (distance, index) lists split over two half-waves.  Here:
//   * waves 0-3 ("matrix waves") only run MFMAs: each owns 64 queries = two 32-query accumulator tiles against the
//     staged 32-candidate tile, and stores the raw inner products of the stage to LDS as [query][candidate] rows --
//     the transposition the selection needs comes with the store;
//   * waves 4-7 ("selection waves") own ONE query per lane.  They stage the candidate tiles, and per stage
//       - test the 32 candidates of their query against a conservative bound of the K-th distance: fma, compare, and
//         the compare's carry shifted into a 32-bit mask (3 instructions per candidate, no branch, no LDS write);
//       - visit the set bits: the exact distance in the reference's order from the inner product still in LDS, and
//         where it beats the K-th distance, ONE v_med3_f32 per slot into a sorted list of distances WITHOUT indices,
//         plus an 8-byte (distance, index) record appended to the lane's log in global memory (stream-ordered
//         workspace, [slot][lane]: coalesced);
//     after the scan the k-th distance tau is final: a log record belongs to the result iff its distance is below
//     tau, or equals tau and it is among the first (k - #below) such records -- records are in candidate order, which
//     is the order equal distances are listed in.  The <= k selected records are ranked against the sorted distances
//     (equal distances: next free slot, in record order) and written out.  A log that nears its capacity is compacted
//     to the records not above the current K-th distance (fewer than 2K: a record is only written when it enters the
//     list).
// One barrier per stage; inner products and candidate tiles are double-buffered.  Same MFMA instruction, same k order
// and the same distance expression as knn_mfma_kernel: identical results, ties included.
// ---------------------------------------------------------------------------------------------------
constexpr int kSplitQ = 256;      // queries per workgroup
constexpr int kSplitPitch = 36;   // floats per query row of one stage (32 + 4: the rows' ds_read_b128 spread over all banks)
constexpr int kLogCap = 256;      // log records per query (compacted when fewer than 32 are free)

template <int K, int CP>
constexpr int split_lds_bytes() {
    constexpr int main_bytes = 2 * CP * 32 * 4 + 3 * 32 * 4 + 2 * kSplitQ * kSplitPitch * 4;
    constexpr int final_bytes = K * 256 * (8 + 4);  // selected records | output slots
    return main_bytes > final_bytes ? main_bytes : final_bytes;
}
inline size_t split_log_bytes(int b, int n) { return (size_t)b * pcc::ceil_div(n, kSplitQ) * kLogCap * 256 * sizeof(float2); }

// Largest value of a non-negative int over the wave (wave-uniform result).
__device__ __forceinline__ int wave_max_nonneg(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

// An 8-byte log record, read past the L1 (written by this lane earlier, read once).
__device__ __forceinline__ float2 log_load(const float2 *p) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(p));
    return float2{v.x, v.y};
}

template <int K, int CP /* padded channels, multiple of 2, <= 128 */>
__global__ __launch_bounds__(512) void knn_mfma_split_kernel(int c, int n, int k, const float *__restrict__ x,
                                                             const float *__restrict__ sq, float2 *__restrict__ logs,
                                                             int64_t *__restrict__ indices) {
    constexpr int KS = CP / 2;  // MFMA k-steps (32x32x2)
    constexpr int tile_floats = CP * 32;
    constexpr int dist_floats = kSplitQ * kSplitPitch;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tile = reinterpret_cast<float *>(smem);                 // [2][CP][32]
    float *tsq = tile + 2 * tile_floats;                           // [3][32] (read one stage later than the tile: see commit)
    float *dist = tsq + 3 * 32;                                    // [2][256][36]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int smp = blockIdx.y;
    const float *xb = x + (size_t)smp * c * n;
    const float *sqb = sq + (size_t)smp * n;
    const int nstages = (n + 31) / 32;

    if (w < 4) {
        // ---- matrix wave: queries blockIdx.x * 256 + w * 64 + [0, 64)
        const int half = lane >> 5, col = lane & 31;
        float bq[2][KS];  // B operand: query[col][k = 2 * ks + half] of the two query tiles
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int q = min(blockIdx.x * kSplitQ + w * 64 + u * 32 + col, n - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const int ch = 2 * ks + half;
                bq[u][ks] = ch < c ? xb[(size_t)ch * n + q] : 0.f;
            }
        }
        __syncthreads();  // stage 0 is in LDS
        for (int t = 0; t < nstages; t++) {
            const float *cur = tile + (t & 1) * tile_floats;
            f32x16 acc[2];
#pragma unroll
            for (int u = 0; u < 2; u++)
                acc[u] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            constexpr int KB = KS < 8 ? KS : 8;  // A operands are read this many k-steps ahead of their MFMAs
#pragma unroll
            for (int ks0 = 0; ks0 < KS; ks0 += KB) {
                float av[KB];
#pragma unroll
                for (int kk = 0; kk < KB; kk++) av[kk] = cur[(2 * (ks0 + kk) + half) * 32 + col];  // candidate[row = col][k]
#pragma unroll
                for (int kk = 0; kk < KB; kk++)
#pragma unroll
                    for (int u = 0; u < 2; u++)
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bq[u][ks0 + kk], acc[u], 0, 0, 0);
            }
            // accumulator register r of lane (half, col) = candidate row (r & 3) + 8 * (r >> 2) + 4 * half of query col
            float *drow = dist + (t & 1) * dist_floats;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                float *qrow = drow + (w * 64 + u * 32 + col) * kSplitPitch + 4 * half;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float4 *>(qrow + 8 * j) =
                        float4{acc[u][4 * j], acc[u][4 * j + 1], acc[u][4 * j + 2], acc[u][4 * j + 3]};
            }
            __syncthreads();
        }
        __syncthreads();  // (the selection waves reuse the stage buffers after this one)
        return;
    }

    // ---- selection wave: one query per lane
    const int ct = tid - 256;
    int q = blockIdx.x * kSplitQ + ct;
    const bool q_ok = q < n;
    q = q_ok ? q : n - 1;
    float sq_q = sqb[q];
    asm volatile("" : "+v"(sq_q));  // (consumed here: left pending, its wait lands in the rounds and drains the tile prefetch with it)
    float2 *logp = logs + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kLogCap * 256 + ct;  // record s at logp[s * 256]
    float ld[K];  // the K smallest distances so far, ascending
#pragma unroll
    for (int s2 = 0; s2 < K; s2++) ld[s2] = __builtin_inff();
    int lcnt = 0;  // records in this lane's log
    // Pre-test bound on a = |xj|^2 - 2 xi.xj: every a whose distance fl(a + |xi|^2) is below the K-th distance W is
    // below it (W - |xi|^2 plus 64 times the rounding the sum and this expression can carry; +inf while the list is open).
    float bound = __builtin_inff();
    auto refresh_bound = [&]() {
        const float W = ld[K - 1];
        bound = (W - sq_q) + ((fabsf(W) + fabsf(sq_q)) * 0x1p-18f + 1e-30f);
    };
    // keep the records that can still belong to the result (fewer than 2K).  Rare; its memory accesses are asm for the
    // reason given at the log store in drain (a visible load or store here costs every stage of every call a drain of the
    // tile prefetch), one round trip per record.
    auto compact = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float W = ld[K - 1];
        const int m = wave_max_nonneg(lcnt);
        int kept = 0;
        for (int i = 0; i < m; i++) {
            if (i < lcnt) {
                unsigned long long rec;
                asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(rec) : "v"(logp + i * 256) : "memory");
                if (__uint_as_float((unsigned)rec) <= W) {
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(logp + kept * 256), "v"(rec) : "memory");
                    kept++;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lcnt = kept;
    };

    // Tile staging: global -> registers TWO stages ahead -> LDS one stage ahead.  (One stage ahead, every stage waited
    // out a global-memory round trip -- several microseconds, more than its arithmetic -- before its barrier.)
    constexpr int E = CP * 32 / 256;  // tile elements per selection thread per stage
    float pre_a[E], pre_b[E], pre_sq_a = 0.f, pre_sq_b = 0.f;
    // fetch only issues the loads (clamped addresses, nothing consumes the values); the out-of-range select happens at
    // commit time, a stage later -- a select next to the load makes the compiler wait for every load where it is issued.
    auto fetch = [&](int t, float (&pre)[E], float &pre_sq) {
        const int j = min(t * 32 + (ct & 31), n - 1);
#pragma unroll
        for (int i = 0; i < E; i++) pre[i] = xb[(size_t)min((ct >> 5) + i * 8, c - 1) * n + j];
        pre_sq = sqb[j];
    };
    // (the norms of stage t are read by the selection of stage t one iteration after the matrix waves read its tile, while
    // another selection wave may already commit stage t + 2: three norm buffers, two tiles)
    auto commit = [&](int t, const float (&pre)[E], float pre_sq) {
        float *dst = tile + (t & 1) * tile_floats;
        const bool in = t * 32 + (ct & 31) < n;
        // (bit masks, not selects: the compiler turns the selects into a branch per element)
#pragma unroll
        for (int i = 0; i < E; i++)
            dst[ct + i * 256] = __int_as_float(__float_as_int(pre[i]) & -(int)(in && (ct >> 5) + i * 8 < c));
        if (ct < 32) tsq[(t % 3) * 32 + ct] = in ? pre_sq : __builtin_inff();
    };
    // screen: the candidates of stage t that may beat the K-th distance, as a bit mask
    auto screen = [&](int t) -> unsigned {
        const float *drow = dist + (t & 1) * dist_floats + ct * kSplitPitch;
        const float *ts = tsq + (t % 3) * 32;
        unsigned mask = 0;  // candidate e of the stage ends up in bit 31 - e
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 a0 = *reinterpret_cast<const float4 *>(drow + 8 * g);
            const float4 a1 = *reinterpret_cast<const float4 *>(drow + 8 * g + 4);
            const float4 n0 = *reinterpret_cast<const float4 *>(ts + 8 * g);
            const float4 n1 = *reinterpret_cast<const float4 *>(ts + 8 * g + 4);
            const float dot[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const float nj[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float a = __builtin_fmaf(-2.0f, dot[e], nj[e]);  // == (-2*dot) + |xj|^2: the product is exact
                // mask = 2 * mask + (a < bound): the compare's carry goes straight into the add (false for NaN); the s_nop is
                // the two wait states gfx950 wants between a VALU write of vcc and a VALU read of it (the compiler puts the
                // same s_nop between its own v_cmp / v_addc pairs; it cannot see into this block)
                asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\ts_nop 1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(a), "v"(bound) : "vcc");
            }
        }
        return __builtin_bitreverse32(mask);  // candidate e in bit e: visited in ascending order
    };
    // drain: one round per set bit of the fullest lane, straight-line: a lane without a bit offers +inf, and an offer
    // that does not beat the K-th distance leaves the list as it is (median of two neighbours and something not below
    // them), so only the log record is conditional -- no list value crosses a branch.
    auto drain = [&](int t, unsigned mask) {
        const float *drow = dist + (t & 1) * dist_floats + ct * kSplitPitch;
        const float *ts = tsq + (t % 3) * 32;
        if (__any(lcnt > kLogCap - 32)) compact();
        // (bottom-tested by hand: the compiler does not rotate a loop around a ballot, and copies the whole list on both
        // sides of a top test)
        if (__any(mask != 0)) {
            // (the LDS reads of the NEXT round's candidate are issued before this round's chain)
            int e = __builtin_ctz(mask | 0x80000000u);
            float dot = drow[e], nj = ts[e];
            do {
                const bool has = mask != 0;
                mask &= mask - 1;
                const int e_next = __builtin_ctz(mask | 0x80000000u);
                const float dot_next = drow[e_next], nj_next = ts[e_next];
                // reference CPU path: dist = -2*dot ; dist += |xj|^2 (column term) ; dist += |xi|^2 (row term)
                const float xe = __builtin_fmaf(-2.0f, dot, nj) + sq_q;
                const float xv = has ? xe : __builtin_inff();  // (never NaN: it passed a < bound with a finite bound)
                const float W = ld[K - 1];
                // (in place, tail first: written as asm so that no slot is copied around the loop)
#pragma unroll
                for (int s2 = K - 1; s2 > 0; s2--) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(ld[s2]) : "v"(ld[s2 - 1]), "v"(xv));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(ld[0]) : "v"(xv));
                if (xv < W) {
                    // (asm: with a store the compiler can see in this loop, its wait-count pass drains every load in
                    // flight -- the tile prefetch -- before the loop, once per stage; the explicit waits before the
                    // log is read back order these stores)
                    const unsigned long long rec =
                        ((unsigned long long)(unsigned)(t * 32 + e) << 32) | (unsigned long long)__float_as_uint(xv);
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(logp + lcnt * 256), "v"(rec) : "memory");
                }
                lcnt += xv < W ? 1 : 0;
                e = e_next;
                dot = dot_next;
                nj = nj_next;
            } while (__any(mask != 0));
        }
        refresh_bound();
    };
    auto step = [&](int t, float (&nxt)[E], float &nxt_sq, const float (&cur)[E], float cur_sq) {
        // (the loads fly during a whole stage; the commit comes BEFORE the rounds: behind the rounds' log stores its wait
        // for older loads would wait for the stores too)
        fetch(t + 2, nxt, nxt_sq);  // (unconditional, clamped past the end: behind a branch the compiler must wait as if it had not run)
        const unsigned mask = t > 0 ? screen(t - 1) : 0u;
        if (t + 1 < nstages) commit(t + 1, cur, cur_sq);
        if (t > 0) drain(t - 1, mask);
        __syncthreads();
    };
    fetch(0, pre_a, pre_sq_a);
    commit(0, pre_a, pre_sq_a);
    fetch(1, pre_b, pre_sq_b);
    __syncthreads();
    for (int t = 0; t < nstages; t += 2) {
        step(t, pre_a, pre_sq_a, pre_b, pre_sq_b);
        if (t + 1 < nstages) step(t + 1, pre_b, pre_sq_b, pre_a, pre_sq_a);
    }
    drain(nstages - 1, screen(nstages - 1));
    __syncthreads();  // every wave is done with the stage buffers

    // ---- the result from the log
    float2 *sel = reinterpret_cast<float2 *>(smem) + ct;          // [K][256] selected records below tau
    int *out = reinterpret_cast<int *>(smem + K * 256 * 8) + ct;  // [K][256] candidate of output slot o (-1: empty)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float tau = ld[0];
    int below = 0;  // list entries strictly below tau
#pragma unroll
    for (int s2 = 1; s2 < K; s2++) tau = (s2 == k - 1) ? ld[s2] : tau;
#pragma unroll
    for (int s2 = 0; s2 < K; s2++) {
        below += (s2 < k && ld[s2] < tau) ? 1 : 0;
        out[s2 * 256] = -1;
    }
    int nsel = 0, ties = below;  // ties: next output slot of a record equal to tau
    const int m = wave_max_nonneg(lcnt);
    // sixteen records in flight: the next eight are loaded before the current eight are looked at (one at a time, the
    // scan is a chain of memory round trips)
    float2 cur[8], nxt[8];
#pragma unroll
    for (int u = 0; u < 8; u++) cur[u] = log_load(logp + min(u, kLogCap - 1) * 256);
    for (int i0 = 0; i0 < m; i0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) nxt[u] = log_load(logp + min(i0 + 8 + u, kLogCap - 1) * 256);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (i0 + u < lcnt) {
                if (cur[u].x < tau) {
                    sel[nsel * 256] = cur[u];
                    nsel++;
                } else if (cur[u].x == tau && ties < k) {
                    out[ties * 256] = __float_as_int(cur[u].y);
                    ties++;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) cur[u] = nxt[u];
    }
    const int ms = wave_max_nonneg(nsel);
    for (int j = 0; j < ms; j++) {
        if (j < nsel) {
            const float2 e = sel[j * 256];
            int pos = 0;
#pragma unroll
            for (int s2 = 0; s2 < K; s2++) pos += ld[s2] < e.x ? 1 : 0;
            while (out[pos * 256] != -1) pos++;  // equal distances: the next free slot, in record (= candidate) order
            out[pos * 256] = __float_as_int(e.y);
        }
    }
    // a wave writes the rows of its 64 queries with consecutive lanes on consecutive words
    __builtin_amdgcn_wave_barrier();
    const int wq = (w - 4) * 64;  // first local query of this wave
    const int q0 = blockIdx.x * kSplitQ + wq;
    const int *wout = reinterpret_cast<const int *>(smem + K * 256 * 8) + wq;
    int64_t *dst = indices + ((size_t)smp * n + q0) * k;
    const int total = min(64, n - q0) * k;
    for (int e = lane; e < total; e += 64) {
        const int ql = e / k, o = e - ql * k;
        const int v = wout[o * 256 + ql];
        dst[e] = (int64_t)((unsigned)v < (unsigned)n ? v : n - 1);  // (an empty slot only when distances are NaN)
    }
}

template <int K>
int launch_small(int b, int c, int n, int k, const float *x, int64_t *indices, hipStream_t st) {
    pcc::ProfScope prof("knn_small_kernel", st);
    hipLaunchKernelGGL((knn_small_kernel<K, 4>), dim3(pcc::ceil_div(n, 64), b), dim3(256), 0, st, c, n, k, x, indices);
    return PCC_OK;
}

template <int K, int KPREV>
void launch_sorted(const KnnSortedArgs &a, hipStream_t st) {
    pcc::ProfScope prof("knn_sorted_kernel", st);
    const int waves = a.batch * a.nb;
    hipLaunchKernelGGL((knn_sorted_kernel<K, KPREV>), dim3(pcc::ceil_div(waves, kSW)), dim3(64 * kSW), 0, st, a);
}

struct SqBuf {  // stream-ordered workspace block, freed behind the work enqueued so far
    float *p = nullptr;
    hipStream_t st;
    explicit SqBuf(hipStream_t s) : st(s) {}
    ~SqBuf() {
        if (p) (void)pcc::ws_free(p, st);
    }
};

template <int K, int CP>
int launch_split(int b, int c, int n, int k, const float *x, const float *sq, int64_t *indices, hipStream_t st) {
    constexpr int lds = split_lds_bytes<K, CP>();
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_mfma_split_kernel<K, CP>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) {
        pcc::set_error((int)attr, "knn: cannot reserve the role-split kernel's LDS");
        return (int)attr;
    }
    SqBuf logs(st);  // the selection waves' records (stream-ordered: freed behind the kernel)
    if (pcc::ws_malloc(reinterpret_cast<void **>(&logs.p), split_log_bytes(b, n), st) != hipSuccess) {
        logs.p = nullptr;
        (void)hipGetLastError();
        pcc::set_error(PCC_ENOMEM, "knn: workspace allocation failed");
        return PCC_ENOMEM;
    }
    pcc::ProfScope prof("knn_mfma_split_kernel", st);
    hipLaunchKernelGGL((knn_mfma_split_kernel<K, CP>), dim3(pcc::ceil_div(n, kSplitQ), b), dim3(512), lds, st, c, n, k, x, sq,
                       reinterpret_cast<float2 *>(logs.p), indices);
    return PCC_OK;
}

template <int K>
int launch_mfma(int b, int c, int n, int k, const float *x, const float *sq, int64_t *indices, hipStream_t st) {
    // 256-query role-split workgroups (one per CU) once they fill three quarters of the chip; the 128-query kernel below
    // that (measured at n = 2048, k = 25: B = 32 c = 64 / 128: 286 / 417 us against 372 / 514; B = 16: 287 / 417 against 295 / 362)
    const int sw = pcc::tuning(PCC_TUNE_KNN_NOSPLIT);  // measurement switch: 1 = never, 2 = always
    // (and while the selection log -- 512 KB per workgroup -- stays a modest workspace)
    if (sw == 2 || (sw == 0 && (long long)pcc::ceil_div(n, kSplitQ) * b * 4 >= 3LL * pcc::device_cus() && split_log_bytes(b, n) <= (1ull << 30))) {
        if (c <= 16) return launch_split<K, 16>(b, c, n, k, x, sq, indices, st);
        if (c <= 32) return launch_split<K, 32>(b, c, n, k, x, sq, indices, st);
        if (c <= 64) return launch_split<K, 64>(b, c, n, k, x, sq, indices, st);
        return launch_split<K, 128>(b, c, n, k, x, sq, indices, st);
    }
    pcc::ProfScope prof("knn_mfma_kernel", st);
    const dim3 grid(pcc::ceil_div(n, 128), b);
    // two 32-candidate tiles (two accumulator chains) per stage while the double-buffered tiles leave room for two
    // workgroups per CU; 128 channels keep one
    if (c <= 8) hipLaunchKernelGGL((knn_mfma_kernel<K, 8, 1>), grid, dim3(256), 0, st, c, n, k, x, sq, indices);
    else if (c <= 16) hipLaunchKernelGGL((knn_mfma_kernel<K, 16, 1>), grid, dim3(256), 0, st, c, n, k, x, sq, indices);
    else if (c <= 32) hipLaunchKernelGGL((knn_mfma_kernel<K, 32, 1>), grid, dim3(256), 0, st, c, n, k, x, sq, indices);
    else if (c <= 64) hipLaunchKernelGGL((knn_mfma_kernel<K, 64, 1>), grid, dim3(256), 0, st, c, n, k, x, sq, indices);
    else hipLaunchKernelGGL((knn_mfma_kernel<K, 128, kTT128>), grid, dim3(256), 0, st, c, n, k, x, sq, indices);
    return PCC_OK;
}

}  // namespace

extern "C" int pcc_knn(int b, int c, int n, int k, const float *x, int64_t *indices, pcc_stream_t stream) {
    pcc::clear_error();
    if (b < 0 || c < 1 || n < 0 || k < 1) return pcc::invalid("knn: bad size");
    if (b == 0 || n == 0) return PCC_OK;
    if (k > n) return pcc::invalid("knn: k exceeds the number of points (torch.topk raises too)");
    if (k > 32) return pcc::invalid("knn: k > 32 is not supported");
    if (c > 128) return pcc::invalid("knn: more than 128 channels is not supported");
    if (b > 65535) return pcc::invalid("knn: batch too large");
    if (!x || !indices) return pcc::invalid("knn: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c <= 3 && n <= kSortedMaxN) {
        // sorted search: workspace = packed sorted rows | boxes | permutation
        const int nb = pcc::ceil_div(n, kSortBoxK);
        // (+256: the search loads whole 16-row blocks; the last block of the last sample may run past the cloud)
        const size_t aos_b = (size_t)b * n * 16 + 256, box_b = (size_t)b * nb * 32;
        SqBuf ws(st);
        if (pcc::ws_malloc(reinterpret_cast<void **>(&ws.p), aos_b + box_b + (size_t)b * n * 4, st) != hipSuccess) {
            ws.p = nullptr;
            (void)hipGetLastError();
            pcc::set_error(PCC_ENOMEM, "knn: workspace allocation failed");
            return PCC_ENOMEM;
        }
        char *base = reinterpret_cast<char *>(ws.p);
        KnnSortedArgs a{};
        a.n = n; a.nb = nb; a.batch = b; a.k = k;
        a.aos = reinterpret_cast<const float4 *>(base);
        a.box = reinterpret_cast<const float *>(base + aos_b);
        a.perm = reinterpret_cast<const int *>(base + aos_b + box_b);
        a.out = indices;
        if (int rc = pcc::sort_cloud_cmajor(b, c, n, x, reinterpret_cast<float4 *>(base), reinterpret_cast<float *>(base + aos_b),
                                            reinterpret_cast<int *>(base + aos_b + box_b), st))
            return rc;
        if (k <= 4) launch_sorted<4, 0>(a, st);
        else if (k <= 8) launch_sorted<8, 4>(a, st);
        else if (k <= 16) launch_sorted<16, 8>(a, st);
        else if (k <= 20) launch_sorted<20, 16>(a, st);
        else if (k <= 25) launch_sorted<25, 20>(a, st);
        else launch_sorted<32, 25>(a, st);
        return pcc::check_launch("knn(sorted)");
    }
    if (c <= 3) {  // clouds too large for the one-workgroup sort: exhaustive scan
        if (k <= 4) launch_small<4>(b, c, n, k, x, indices, st);
        else if (k <= 8) launch_small<8>(b, c, n, k, x, indices, st);
        else if (k <= 16) launch_small<16>(b, c, n, k, x, indices, st);
        else if (k <= 20) launch_small<20>(b, c, n, k, x, indices, st);
        else if (k <= 25) launch_small<25>(b, c, n, k, x, indices, st);
        else launch_small<32>(b, c, n, k, x, indices, st);
        return pcc::check_launch("knn(small)");
    }
    SqBuf sq(st);
    if (pcc::ws_malloc(reinterpret_cast<void **>(&sq.p), (size_t)b * n * sizeof(float), st) != hipSuccess) {
        sq.p = nullptr;
        (void)hipGetLastError();
        pcc::set_error(PCC_ENOMEM, "knn: workspace allocation failed");
        return PCC_ENOMEM;
    }
    hipLaunchKernelGGL(sqnorm_kernel, dim3(pcc::ceil_div(n, 256), b), dim3(256), 0, st, c, n, x, sq.p);
    if (int rc = pcc::check_launch("knn(sqnorm)")) return rc;
    int rc;
    if (k <= 4) rc = launch_mfma<4>(b, c, n, k, x, sq.p, indices, st);
    else if (k <= 8) rc = launch_mfma<8>(b, c, n, k, x, sq.p, indices, st);
    else if (k <= 16) rc = launch_mfma<16>(b, c, n, k, x, sq.p, indices, st);
    else if (k <= 20) rc = launch_mfma<20>(b, c, n, k, x, sq.p, indices, st);
    else if (k <= 25) rc = launch_mfma<25>(b, c, n, k, x, sq.p, indices, st);
    else rc = launch_mfma<32>(b, c, n, k, x, sq.p, indices, st);
    if (rc) return rc;
    return pcc::check_launch("knn(mfma)");
}
