// Neighbour gather, edge features, max-over-k and global pooling for gfx950 (MI355X).
//
// Replaces the torch.gather / cat / max compositions of the reference's src/utils/neighbour_ops.py:85-119
// and the global pooling of src/module/encoders.py:58 / classifier.py:63-64.  All of them are HBM-bound
// index-driven copies; the kernels keep every global access coalesced along the (n,k) axis and reuse one index
// load for all channels of a block; the backward scatter accumulates into per-workgroup LDS bins (no global
// atomics).
#include "pcc_common.hpp"

#include <cstdlib>
#include "pcc_neighbour.h"
#include "pcc_test_hooks.h"

namespace {

// A neighbour index outside [0, n) (an index array built for another cloud size, a padding slot) must never become an
// LDS address: it is replaced by the point itself (torch.gather would raise; a kernel cannot).  Memory safety only --
// the Python layer documents that indices must lie in [0, n).
__device__ __forceinline__ int nbr(long long v, int n, int self) { return (unsigned long long)v < (unsigned long long)n ? (int)v : self; }

// The k neighbours of the 64 points a wave works on, for the per-point modes (max / sum / min-max over k).  A lane that
// walks its own row of the [n][k] int64 list reads 8 bytes every 8k bytes: 64 cache lines per wave-load, and every channel
// block of the sample repeats it.  Instead the wave copies its 64 rows -- one contiguous run of 64k entries -- with
// coalesced loads into LDS as 16-bit indices (n <= 40960 here; 0xffff = "not a valid index": the point itself), and the
// lanes read their rows from there.  Wave-local: no workgroup barrier.  Without room in LDS (stage == nullptr) the rows
// are read from global memory as before.
struct NbrList {
    unsigned short *stage;   // this wave's [64][k] staging area, or nullptr
    const int64_t *ib;       // the sample's [n][k] list
    int n, k, lane;
    __device__ __forceinline__ void load(int first_point) {  // the wave's points are first_point .. first_point + 63
        if (!stage) return;
        __builtin_amdgcn_wave_barrier();  // (the previous tile's reads are done)
        const int cnt = max(0, min(64, n - first_point)) * k;
        const int64_t *src = ib + (size_t)first_point * k;
        for (int e = lane; e < cnt; e += 64) {
            const long long v = src[e];
            stage[e] = (unsigned long long)v < (unsigned long long)n ? (unsigned short)v : (unsigned short)0xffff;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ int at(int i, int j) const {  // neighbour j of point i (this lane's point)
        if (!stage) return nbr(ib[(size_t)i * k + j], n, i);
        const int s = stage[lane * k + j];
        return s == 0xffff ? i : s;
    }
};

// Forward: a workgroup owns CB channels of one sample, stages those rows of x in LDS (the gathers then hit
// LDS instead of 64 different cache lines per wave-instruction) and streams the (n,k) index list.
//   MODE 0: gather            out[b,c,n,j]  = x[b,c,idx]
//   MODE 1: graph features    out[b,c,n,j]  = x[b,c,idx] - x[b,c,n] ; out[b,C+c,n,j] = x[b,c,n]
//   MODE 2: max over k        out[b,c,n]    = max_j x[b,c,idx[n,j]] (+ argmax, first maximum like torch.max)
//   MODE 3: neighbour sum     out[b,c,n]    = sum_j x[b,c,idx[n,j]]
//   MODE 4: max AND min over k as TARGET indices: tsel[b,0,c,n] = idx[n, argmax_j], tsel[b,1,c,n] = idx[n, argmin_j]
//           (what the fused EdgeConv needs: the edge that survives max-over-k for either sign of the BN scale)
template <int MODE, int CB>
__global__ __launch_bounds__(1024) void gather_lds_kernel(int c, int n, int k, const float *__restrict__ x,
                                                           const int64_t *__restrict__ indices, float *__restrict__ out,
                                                           int32_t *__restrict__ argmax, int64_t *__restrict__ tsel,
                                                           int stage_off) {
    extern __shared__ __attribute__((aligned(16))) float rows[];  // [CB][n] (+ index staging, see NbrList)
    // one-dimensional launch, sample-major on XCD-contiguous ids: the channel blocks of a sample share an L2 (they all
    // stream the sample's index list; side by side on eight XCDs each would fetch it over the fabric)
    const int nblk = (c + CB - 1) / CB, lid = pcc::xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int smp = lid / nblk, c0 = (lid - smp * nblk) * CB;
    const int tid = threadIdx.x, T = 1024;
    const size_t nk = (size_t)n * k;
    const float *xb = x + ((size_t)smp * c + c0) * n;
    const int cb = min(CB, c - c0);
    for (int i = tid; i < cb * n; i += T) rows[i] = xb[i];
    __syncthreads();
    const int64_t *ib = indices + (size_t)smp * nk;
    NbrList nl;
    nl.ib = ib; nl.n = n; nl.k = k; nl.lane = tid & 63;
    nl.stage = stage_off ? reinterpret_cast<unsigned short *>(rows + stage_off) + (size_t)(tid >> 6) * 64 * k : nullptr;
    if (MODE == 3 || MODE == 4) {
        for (int base = 0; base < n; base += T) {
            const int i = base + tid;
            nl.load(base + (tid & ~63));
            if (i >= n) continue;
            float acc[CB], lo[CB];
            int tb[CB], tl[CB];
#pragma unroll
            for (int cc = 0; cc < CB; cc++) {
                acc[cc] = MODE == 3 ? 0.f : -__builtin_inff();
                lo[cc] = __builtin_inff();
                tb[cc] = tl[cc] = 0;
            }
            for (int j = 0; j < k; j++) {
                const int t = nl.at(i, j);
#pragma unroll
                for (int cc = 0; cc < CB; cc++) {
                    if (cc < cb) {
                        const float v = rows[cc * n + t];
                        if (MODE == 3) {
                            acc[cc] += v;
                        } else {
                            const bool gt = (j == 0) || v > acc[cc];
                            const bool lt = (j == 0) || v < lo[cc];
                            acc[cc] = gt ? v : acc[cc];
                            tb[cc] = gt ? t : tb[cc];
                            lo[cc] = lt ? v : lo[cc];
                            tl[cc] = lt ? t : tl[cc];
                        }
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < CB; cc++) {
                if (cc < cb) {
                    if (MODE == 3) {
                        out[((size_t)smp * c + c0 + cc) * n + i] = acc[cc];
                    } else {
                        tsel[(((size_t)smp * 2 + 0) * c + c0 + cc) * n + i] = tb[cc];
                        tsel[(((size_t)smp * 2 + 1) * c + c0 + cc) * n + i] = tl[cc];
                    }
                }
            }
        }
    } else if (MODE == 2) {
        for (int base = 0; base < n; base += T) {
            const int i = base + tid;
            nl.load(base + (tid & ~63));
            if (i >= n) continue;
            float best[CB];
            int bj[CB];
#pragma unroll
            for (int cc = 0; cc < CB; cc++) {
                best[cc] = -__builtin_inff();
                bj[cc] = 0;
            }
            for (int j = 0; j < k; j++) {
                const int t = nl.at(i, j);
#pragma unroll
                for (int cc = 0; cc < CB; cc++) {
                    if (cc < cb) {
                        const float v = rows[cc * n + t];
                        const bool gt = (j == 0) || v > best[cc];
                        best[cc] = gt ? v : best[cc];
                        bj[cc] = gt ? j : bj[cc];
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < CB; cc++) {
                if (cc < cb) {
                    out[((size_t)smp * c + c0 + cc) * n + i] = best[cc];
                    if (argmax) argmax[((size_t)smp * c + c0 + cc) * n + i] = bj[cc];
                }
            }
        }
    } else {
        const int out_c = MODE == 1 ? 2 * c : c;
        float *ob = out + (size_t)smp * out_c * nk;
        // (16-byte accesses: whole float4 groups and 16-byte aligned bases -- a tensor view with a storage offset need not be)
        if ((nk & 3) == 0 && k >= 4 && ((reinterpret_cast<uintptr_t>(ib) | reinterpret_cast<uintptr_t>(ob)) & 15) == 0) {
            // four consecutive edges per thread: the output is a write-only stream many times the L2 (838 MB at C=64,
            // k=25, B=32), stored as 16 bytes per lane; the index list as two 16-byte loads
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef long long v2l __attribute__((ext_vector_type(2)));
            for (size_t e4 = (size_t)tid * 4; e4 < nk; e4 += (size_t)T * 4) {
                const v2l ia = *reinterpret_cast<const v2l *>(ib + e4), ic = *reinterpret_cast<const v2l *>(ib + e4 + 2);
                const long long raw[4] = {ia.x, ia.y, ic.x, ic.y};
                const int i0 = (int)(e4 / k), r0 = (int)(e4 - (size_t)i0 * k);
                int t[4], i[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    i[q] = i0 + (r0 + q >= k ? 1 : 0);  // (k >= 4: four edges span at most two points)
                    t[q] = nbr(raw[q], n, i[q]);
                }
#pragma unroll
                for (int cc = 0; cc < CB; cc++) {
                    if (cc < cb) {
                        const float *row = rows + cc * n;
                        v4f nb = {row[t[0]], row[t[1]], row[t[2]], row[t[3]]};
                        if (MODE == 0) {
                            __builtin_nontemporal_store(nb, reinterpret_cast<v4f *>(ob + (size_t)(c0 + cc) * nk + e4));
                        } else {
                            const v4f self = {row[i[0]], row[i[1]], row[i[2]], row[i[3]]};
                            __builtin_nontemporal_store(nb - self, reinterpret_cast<v4f *>(ob + (size_t)(c0 + cc) * nk + e4));
                            __builtin_nontemporal_store(self, reinterpret_cast<v4f *>(ob + (size_t)(c + c0 + cc) * nk + e4));
                        }
                    }
                }
            }
            return;
        }
        for (size_t e = tid; e < nk; e += T) {
            const int i = (int)(e / k);
            const int t = nbr(ib[e], n, i);
#pragma unroll
            for (int cc = 0; cc < CB; cc++) {
                if (cc < cb) {
                    const float nb = rows[cc * n + t];
                    if (MODE == 0) {
                        ob[(size_t)(c0 + cc) * nk + e] = nb;
                    } else {
                        const float self = rows[cc * n + i];
                        ob[(size_t)(c0 + cc) * nk + e] = nb - self;
                        ob[(size_t)(c + c0 + cc) * nk + e] = self;
                    }
                }
            }
        }
    }
}

// ---- backward scatter --------------------------------------------------------------------------------
// grad_x[b,c,t] = sum over edges e=(i,j) with indices[b,i,j]==t of g[b,c,e]  (+ the self terms).
// A workgroup owns CB channels of one sample and keeps their n gradient bins in LDS; it streams the edge
// list with fully coalesced reads of `indices` and of the CB gradient rows and accumulates with ds_add_f32.
// kNN graphs in feature space are hubby (one point can be the neighbour of thousands), which ruins any
// one-thread-per-target gather; the bin scatter is insensitive to that.  Like torch's scatter_add (what the
// reference's gather backward runs) the float summation order is not fixed.
//   MODE 0: gather            MODE 1: graph features (adds sum_j g[C+c][i,j] - g[c][i,j] to bin i)
//   MODE 2: max pool (only the argmax edge of every (c,i) carries gradient)
//   MODE 3: neighbour sum (every edge (i,j) carries g[b,c,i])
template <int MODE, int CB>
__global__ __launch_bounds__(1024) void scatter_lds_kernel(int c, int n, int k, const int64_t *__restrict__ indices,
                                                            const int32_t *__restrict__ argmax,
                                                            const float *__restrict__ g, float *__restrict__ grad_x) {
    extern __shared__ __attribute__((aligned(16))) float bins[];  // [CB][n]
    // one-dimensional launch, sample-major on XCD-contiguous ids: the channel blocks of a sample share an L2 (they all
    // stream the sample's index list; side by side on eight XCDs each would fetch it over the fabric)
    const int nblk = (c + CB - 1) / CB, lid = pcc::xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int smp = lid / nblk, c0 = (lid - smp * nblk) * CB;
    const int tid = threadIdx.x, T = 1024;
    const size_t nk = (size_t)n * k;
    const int64_t *ib = indices + (size_t)smp * nk;
    const int gc = MODE == 1 ? 2 * c : c;
    for (int i = tid; i < CB * n; i += T) bins[i] = 0.f;
    __syncthreads();
    if (MODE == 2) {
        for (int i = tid; i < n; i += T) {
#pragma unroll
            for (int cc = 0; cc < CB; cc++) {
                const int ch = c0 + cc;
                if (ch < c) {
                    const int j = argmax[((size_t)smp * c + ch) * n + i];
                    const int t = nbr(ib[(size_t)i * k + j], n, i);
                    atomicAdd(&bins[cc * n + t], g[((size_t)smp * c + ch) * n + i]);
                }
            }
        }
    } else {
        for (size_t e = tid; e < nk; e += T) {
            const int i = (int)(e / k);
            const int t = nbr(ib[e], n, i);
#pragma unroll
            for (int cc = 0; cc < CB; cc++) {
                const int ch = c0 + cc;
                if (ch < c) {
                    const float v = MODE == 3 ? g[((size_t)smp * c + ch) * n + i] : g[((size_t)smp * gc + ch) * nk + e];
                    atomicAdd(&bins[cc * n + t], v);
                    if (MODE == 1) atomicAdd(&bins[cc * n + i], g[((size_t)smp * gc + c + ch) * nk + e] - v);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < CB * n; i += T) {
        const int cc = i / n, t = i - cc * n;
        if (c0 + cc < c) grad_x[((size_t)smp * c + c0 + cc) * n + t] = bins[i];
    }
}

// ---- neighbour-sum backward without per-edge float atomics ------------------------------------------------------
// grad_x[b,c,t] = sum over edges (i -> t) of g[b,c,i].  ds_add_f32 costs ~3.5 cycles per LANE (measured), 35x a
// ds_read, and the per-edge scatter above issues b*c*n*k of them (105 M at B=32, C=64, N=2048, k=25: 1 ms).  Here the
// edge list of a sample is first sorted by TARGET (counting sort in LDS with integer atomics, once per call, shared
// by all channels), then every wave streams 64 sorted edges at a time, gathers g[c][source] from the LDS-staged rows,
// runs a segmented prefix sum over equal targets (shuffles; the segment structure is computed once per 64 edges and
// reused by the CB channels) and only the last lane of every segment adds into the LDS bin: ~(n + n*k/64) float
// atomics per channel instead of n*k, never two lanes of one instruction on the same bin.
// The order of the edges inside a target's segment comes from integer atomics: like the scatter above (and torch's
// scatter_add) the float summation order is not fixed.
// rev[b][n*k] = source point (low 16 bits) | target point (high 16 bits), sorted by target; n <= 65536.
__global__ __launch_bounds__(1024) void edge_sort_kernel(int n, int k, const int64_t *__restrict__ indices,
                                                          unsigned *__restrict__ rev) {
    extern __shared__ __attribute__((aligned(16))) int cur[];  // [n] counts, then write cursors
    __shared__ int wave_tot[16];
    const int smp = blockIdx.x, tid = threadIdx.x, T = 1024, lane = tid & 63, w = tid >> 6;
    const size_t nk = (size_t)n * k;
    const int64_t *ib = indices + (size_t)smp * nk;
    unsigned *out = rev + (size_t)smp * nk;
    for (int i = tid; i < n; i += T) cur[i] = 0;
    __syncthreads();
    for (size_t e = tid; e < nk; e += T) atomicAdd(&cur[nbr(ib[e], n, (int)(e / k))], 1);
    __syncthreads();
    // exclusive scan of the n counts: every thread owns a contiguous run
    const int per = (n + T - 1) / T;
    const int beg = min(tid * per, n), end = min(beg + per, n);
    int mine = 0;
    for (int i = beg; i < end; i++) mine += cur[i];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        incl += lane >= off ? v : 0;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int before = 0;
    for (int i = 0; i < w; i++) before += wave_tot[i];
    int run = before + incl - mine;
    for (int i = beg; i < end; i++) {
        const int cnt = cur[i];
        cur[i] = run;
        run += cnt;
    }
    __syncthreads();
    for (size_t e = tid; e < nk; e += T) {
        const int t = nbr(ib[e], n, (int)(e / k));
        const int pos = atomicAdd(&cur[t], 1);
        out[pos] = (unsigned)(e / k) | ((unsigned)t << 16);
    }
}

template <int CB>
__global__ __launch_bounds__(1024) void nbrsum_bwd_sorted_kernel(int c, int n, int k, const unsigned *__restrict__ rev,
                                                                  const float *__restrict__ g,
                                                                  float *__restrict__ grad_x) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // rows [CB][n] | bins [CB][n]
    float *rows = lds, *bins = lds + (size_t)CB * n;
    // one-dimensional launch, sample-major on XCD-contiguous ids: the channel blocks of a sample share an L2 (they all
    // stream the sample's index list; side by side on eight XCDs each would fetch it over the fabric)
    const int nblk = (c + CB - 1) / CB, lid = pcc::xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int smp = lid / nblk, c0 = (lid - smp * nblk) * CB;
    const int tid = threadIdx.x, T = 1024, lane = tid & 63;
    const size_t nk = (size_t)n * k;
    const unsigned *rb = rev + (size_t)smp * nk;
    for (int i = tid; i < CB * n; i += T) {
        const int cc = i / n, p = i - cc * n;
        rows[i] = c0 + cc < c ? g[((size_t)smp * c + c0 + cc) * n + p] : 0.f;
        bins[i] = 0.f;
    }
    __syncthreads();
    const size_t span = (nk + 63) / 64 * 64;  // whole waves enter the loop (shuffles need every lane)
    for (size_t e = tid; e < span; e += T) {
        const bool valid = e < nk;
        const unsigned r = valid ? rb[e] : 0xffffffffu;
        const int src = (int)(r & 0xffffu) % n;  // (an invalid lane reads a harmless in-range address)
        const int t = valid ? (int)(r >> 16) : -1;
        // segment structure of these 64 edges (sorted by target): shared by the CB channels
        bool same[6];
#pragma unroll
        for (int s6 = 0; s6 < 6; s6++) {
            const int off = 1 << s6;
            const int tp = __shfl_up(t, off, 64);  // every lane takes part (no short-circuit: a lane masked off by
            same[s6] = lane >= off && tp == t;     // `lane >= off &&` would hand garbage to the lanes reading it)
        }
        const int tn = __shfl_down(t, 1, 64);
        const bool tail = valid && (lane == 63 || tn != t);
        float v[CB];
#pragma unroll
        for (int cc = 0; cc < CB; cc++) v[cc] = valid ? rows[cc * n + src] : 0.f;
#pragma unroll
        for (int s6 = 0; s6 < 6; s6++) {
#pragma unroll
            for (int cc = 0; cc < CB; cc++) {
                const float u = __shfl_up(v[cc], 1 << s6, 64);
                v[cc] += same[s6] ? u : 0.f;
            }
        }
        if (tail) {
#pragma unroll
            for (int cc = 0; cc < CB; cc++) atomicAdd(&bins[cc * n + t], v[cc]);
        }
    }
    __syncthreads();
    for (int i = tid; i < CB * n; i += T) {
        const int cc = i / n, p = i - cc * n;
        if (c0 + cc < c) grad_x[((size_t)smp * c + c0 + cc) * n + p] = bins[i];
    }
}

// ---- gather / edge-feature backward as a stream over the gradient ---------------------------------------------------------
// grad_x[b,c,t] = sum over edges e = (i,j) with indices[b,i,j] == t of g[b,c,e]          (get_neighbours, neighbour_ops.py:85-94)
//               + sum_j (g[b,C+c,t,j] - g[b,c,t,j])                                        (get_graph_features, :113-119)
// The gradient is per EDGE: [B,(2)C,N,k] floats, 838 MB at B=32, C=64, N=2048, k=25 -- it has to be read once, coalesced,
// and that read is the floor (~150 us).  scatter_lds_kernel does read it that way but pays one ds_add_f32 per edge
// and channel (~3.5 cycles per LANE): 1.6 ms.  Here:
//   * the edge list of a sample is cut into chunks of P source points (P*k <= kEsCE edges, what two channel rows of LDS
//     hold) and every chunk is counting-sorted by TARGET once per call (edge_chunk_sort_kernel: LDS integer atomics; the
//     graph is shared by all channels).  With every sorted entry goes a byte of flags: the segment structure of its
//     64-entry group (which of the six steps of the wave-wide prefix sum add), whether the entry closes its target's segment inside the run
//     of entries one wave walks, and whether that segment crosses a run boundary;
//   * a workgroup owns CB channels of a sample: it stages the chunk's gradient rows in LDS with 16-byte streaming loads
//     (the mandatory HBM traffic), its 16 waves each walk one contiguous run of the sorted entries -- LDS gather of
//     g[c][e], segmented prefix sum over equal targets with the precomputed flags, an open segment carried from one
//     64-entry group to the next in a register -- and only the entry that closes a segment touches the LDS bin: a plain
//     read-modify-write (no two waves ever hold the same target, except the <= 15 segments that cross a run boundary:
//     those use ds_add_f32);
//   * the self terms of the edge features are k consecutive values per point: the second half of the gradient only feeds
//     them and is reduced by its own streaming kernel (edge_self_sum_kernel); the first half's are summed from the staged
//     rows, one thread per (channel, point), plain updates.
// Like scatter_add in torch (what the reference's gather backward runs) the order inside a target's segment comes from
// integer atomics: the float summation order is not fixed.
constexpr int kEsCE = 7680;     // edges per chunk (two channel rows of 30 KB + bins: two workgroups per CU at n <= 2560)
constexpr int kEsWaves = 16;     // waves per workgroup of the stream kernel (<= 64 VGPRs: two workgroups, 8 waves per SIMD, per CU)
constexpr int kEsT = 64 * kEsWaves;
constexpr int kEsMaxGroups = ((kEsCE + kEsWaves - 1) / kEsWaves + 63) / 64;  // 64-entry groups in the longest run of one wave

__host__ __device__ inline int es_points_per_chunk(int n, int k) {
    int p = (kEsCE / k) & ~63;  // whole waves of points (and P*k a multiple of 4: 16-byte loads stay aligned)
    if (p < 64) return 0;
    return p < n ? p : ((n + 63) & ~63);
}
__host__ __device__ inline int es_run_len(int cnt) {  // entries one wave walks: whole 64-entry groups
    return ((cnt + kEsWaves - 1) / kEsWaves + 63) & ~63;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float es_dpp(float v) {  // the DPP-selected partner's value (0 where the lane has none)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float es_keep(float u, unsigned f, int bit) {  // u if bit `bit` of f is set, +0 otherwise
    return __int_as_float(__float_as_int(u) & -(int)((f >> bit) & 1u));
}

__global__ __launch_bounds__(1024) void edge_chunk_sort_kernel(int n, int k, int P, const int64_t *__restrict__ indices,
                                                                unsigned *__restrict__ ent, unsigned char *__restrict__ flg) {
    extern __shared__ __attribute__((aligned(16))) int es_lds[];  // start[n] | cur[n] | key[kEsCE]
    int *start = es_lds, *cur = es_lds + n;
    unsigned *key = reinterpret_cast<unsigned *>(es_lds + 2 * (size_t)n);
    __shared__ int wave_tot[16];
    const int ch = blockIdx.x, smp = blockIdx.y, tid = threadIdx.x, T = 1024, lane = tid & 63, w = tid >> 6;
    const int i0 = ch * P, pc = min(P, n - i0), cnt = pc * k;
    const size_t nk = (size_t)n * k;
    const int64_t *ib = indices + (size_t)smp * nk + (size_t)i0 * k;
    unsigned *E = ent + (size_t)smp * nk + (size_t)i0 * k;
    unsigned char *F = flg + (size_t)smp * nk + (size_t)i0 * k;
    for (int i = tid; i < n; i += T) cur[i] = 0;
    __syncthreads();
    for (int e = tid; e < cnt; e += T) atomicAdd(&cur[nbr(ib[e], n, i0 + e / k)], 1);
    __syncthreads();
    // exclusive scan of the n counts: every thread owns a contiguous run
    const int per = (n + T - 1) / T;
    const int beg = min(tid * per, n), end = min(beg + per, n);
    int mine = 0;
    for (int i = beg; i < end; i++) mine += cur[i];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        incl += lane >= off ? v : 0;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int before = 0;
    for (int i = 0; i < w; i++) before += wave_tot[i];
    int run = before + incl - mine;
    for (int i = beg; i < end; i++) {
        const int c = cur[i];
        cur[i] = run;
        start[i] = run;
        run += c;
    }
    __syncthreads();
    for (int e = tid; e < cnt; e += T) {
        const int t = nbr(ib[e], n, i0 + e / k);
        key[atomicAdd(&cur[t], 1)] = ((unsigned)t << 13) | (unsigned)e;  // e < kEsCE <= 8192
    }
    __syncthreads();
    const int L = es_run_len(cnt);
    for (int p = tid; p < cnt; p += T) {
        const unsigned kp = key[p];
        const int t = (int)(kp >> 13), ln = p & 63;
        // steps of the wave-wide segmented prefix sum of edge_stream_bwd_kernel (DPP: no LDS round trips): four shifts
        // inside the 16-lane row, then lane 15 / 47 into the next row, then lane 31 into the upper half
        unsigned f = 0;
        const int lr = ln & 15;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
            if (lr >= (1 << s4) && (int)(key[p - (1 << s4)] >> 13) == t) f |= 1u << s4;
        if ((ln & 16) && (int)(key[p - lr - 1] >> 13) == t) f |= 16u;
        if (ln >= 32 && (int)(key[p - ln + 31] >> 13) == t) f |= 32u;
        const int seg_first = start[t], seg_last = cur[t] - 1;  // (cur[t] has advanced to the end of the segment)
        const bool seg_end = p == seg_last, run_end = ((p + 1) % L == 0) || p == cnt - 1;
        if (seg_end || run_end) f |= 64u;                       // flush the running sum into the bin here
        if (seg_first / L != seg_last / L) f |= 128u;            // the segment is shared by two waves: atomic flush
        E[p] = kp;
        F[p] = (unsigned char)f;
    }
}

// The self term of the SECOND half of the edge-feature gradient: grad_x[b,c,i] = sum_j g[b, C + c, i, j] -- k consecutive
// values per point, a pure streaming reduction (419 MB at B=32, C=64, N=2048, k=25).  Written first; the stream kernel
// below adds the scattered terms and the first half's self term on top.  A workgroup takes 256 points of one (b, c) row:
// coalesced 16-byte loads into LDS, then one thread per point.
constexpr int kSelfPts = 256;
__global__ __launch_bounds__(256) void edge_self_sum_kernel(int c, int n, int k, const float *__restrict__ g, float *__restrict__ grad_x) {
    extern __shared__ __attribute__((aligned(16))) float ss_buf[];  // [kSelfPts * k]
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int tiles = (n + kSelfPts - 1) / kSelfPts;
    const int row = (int)(blockIdx.x / (unsigned)tiles);  // b * c + channel
    const int smp = row / c, ch = row - smp * c;
    const int i0 = (int)(blockIdx.x - (unsigned)row * tiles) * kSelfPts, pc = min(kSelfPts, n - i0), cnt = pc * k;
    const size_t nk = (size_t)n * k;
    const float *src = g + ((size_t)smp * 2 * c + c + ch) * nk + (size_t)i0 * k;
    const int tid = threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int c4 = cnt & ~3;
        for (int i = tid * 4; i < c4; i += 256 * 4)
            *reinterpret_cast<v4f *>(ss_buf + i) = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src + i));
        for (int i = c4 + tid; i < cnt; i += 256) ss_buf[i] = src[i];
    } else {
        for (int i = tid; i < cnt; i += 256) ss_buf[i] = src[i];
    }
    __syncthreads();
    if (tid < pc) {
        const float *r = ss_buf + (size_t)tid * k;
        float sum = 0.f;
        for (int j = 0; j < k; j++) sum += r[j];
        grad_x[(size_t)row * n + i0 + tid] = sum;
    }
}

// MODE 0: gather backward (grad_x written); MODE 1: edge-feature backward, first half of g (the scattered terms minus the
// self term sum_j g[c][i,j]), ADDED to what edge_self_sum_kernel has written.
template <int MODE, int CB>
__global__ __launch_bounds__(kEsT, 8) void edge_stream_bwd_kernel(int c, int n, int k, int P, const unsigned *__restrict__ ent,
                                                                const unsigned char *__restrict__ flg,
                                                                const float *__restrict__ g, float *__restrict__ grad_x) {
    extern __shared__ __attribute__((aligned(16))) float es_f[];  // buf[CB][kEsCE] | bins[CB][n]
    float *buf = es_f, *bins = es_f + (size_t)CB * kEsCE;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int nblk = (c + CB - 1) / CB, lid = pcc::xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int smp = lid / nblk, c0 = (lid - smp * nblk) * CB;
    const int tid = threadIdx.x, T = kEsT, lane = tid & 63, w = tid >> 6;
    const size_t nk = (size_t)n * k;
    const int gc = MODE == 1 ? 2 * c : c;
    const unsigned *Eb = ent + (size_t)smp * nk;
    const unsigned char *Fb = flg + (size_t)smp * nk;
    for (int i = tid; i < CB * n; i += T) bins[i] = 0.f;
    const float *rows[CB];
#pragma unroll
    for (int cc = 0; cc < CB; cc++) rows[cc] = g + ((size_t)smp * gc + min(c0 + cc, c - 1)) * nk;  // (a padded channel re-reads the last one; never stored)
    // 16-byte streaming loads when every row of this workgroup is 16-byte aligned and the chunks are whole float4 groups
    // (P * k is a multiple of 4, so with n * k a multiple of 4 every chunk is).  Measured and dropped: the rows of chunk
    // i+1 prefetched into registers during the walk over chunk i (8-wave workgroups at 96-111 VGPRs): 197 us against
    // 199 us for the gather backward -- the chunk's critical path is the walk's dependent LDS / cross-lane chain, which
    // more waves per SIMD hide better than a deeper load pipeline (16 waves at <= 64 VGPRs: 182 us).
    bool vec = (nk & 3) == 0;
#pragma unroll
    for (int cc = 0; cc < CB; cc++) vec = vec && (reinterpret_cast<uintptr_t>(rows[cc]) & 15) == 0;
    auto stage = [&](size_t e0, int cnt) {  // buf[cc][0 .. cnt) = the chunk of row cc
#pragma unroll
        for (int cc = 0; cc < CB; cc++) {
            float *dst = buf + (size_t)cc * kEsCE;
            if (vec) {
                for (int i = tid * 4; i < cnt; i += T * 4)
                    *reinterpret_cast<v4f *>(dst + i) = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(rows[cc] + e0 + i));
            } else {
                for (int i = tid; i < cnt; i += T) dst[i] = rows[cc][e0 + i];
            }
        }
    };
    for (int i0 = 0; i0 < n; i0 += P) {
        const int pc = min(P, n - i0), cnt = pc * k;
        const size_t e0 = (size_t)i0 * k;
        // this wave's run of the chunk's sorted entries: independent of the gradient rows, so the loads go out first
        const int L = es_run_len(cnt), run0 = w * L, p_end = min((w + 1) * L, cnt);
        unsigned kp_r[kEsMaxGroups], f_r[kEsMaxGroups];
#pragma unroll
        for (int gi = 0; gi < kEsMaxGroups; gi++) {
            const int p = run0 + gi * 64 + lane;
            const bool valid = p < p_end;
            kp_r[gi] = valid ? Eb[e0 + p] : 0xffffffffu;
            f_r[gi] = valid ? (unsigned)Fb[e0 + p] : 0u;
        }
        __syncthreads();  // bins zeroed / the previous chunk's buffer fully consumed
        stage(e0, cnt);
        __syncthreads();
        // ---- scattered terms: this wave's run of the chunk's target-sorted entries (fetched above, before the staging
        // barrier: inside the loop each group's entry load would be a dependent L2 round trip, ~1.5 us x 7 groups) ----
        {
            float carry[CB];
            int carry_t = -1;
#pragma unroll
            for (int cc = 0; cc < CB; cc++) carry[cc] = 0.f;
#pragma unroll
            for (int gi = 0; gi < kEsMaxGroups; gi++) {
                const int p0 = run0 + gi * 64;
                if (p0 < p_end) {  // (wave-uniform)
                    const bool valid = p0 + lane < p_end;
                    const unsigned kp = kp_r[gi], f = f_r[gi];
                    const int t = valid ? (int)(kp >> 13) : -2;
                    const int el = (int)(kp & 8191u) % kEsCE;  // (an invalid lane reads a harmless in-range address)
                    float v[CB];
#pragma unroll
                    for (int cc = 0; cc < CB; cc++) v[cc] = valid ? buf[(size_t)cc * kEsCE + el] : 0.f;
#pragma unroll
                    for (int cc = 0; cc < CB; cc++) {
                        // segmented inclusive prefix sum over the 64 lanes with cross-lane VALU operands (row_shr 1 2 4 8,
                        // row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3); a lane adds at a step iff its flag
                        // says the partner belongs to the same target
                        // (the partner's value is taken by ALL lanes and then masked with bit arithmetic: written as a
                        // select, the compiler moves the DPP move under the flag's exec mask, and a lane whose own flag is
                        // clear then is an inactive -- invalid -- source for its neighbour)
                        float x = v[cc];
                        x += es_keep(es_dpp<0x111, 0xf>(x), f, 0);
                        x += es_keep(es_dpp<0x112, 0xf>(x), f, 1);
                        x += es_keep(es_dpp<0x114, 0xf>(x), f, 2);
                        x += es_keep(es_dpp<0x118, 0xf>(x), f, 3);
                        x += es_keep(es_dpp<0x142, 0xa>(x), f, 4);
                        x += es_keep(es_dpp<0x143, 0xc>(x), f, 5);
                        v[cc] = x;
                    }
                    // the segment left open by the previous group of this run continues at the head of this one
                    const bool head = t == carry_t;
                    const int t63 = __builtin_amdgcn_readlane(t, 63);              // (wave-uniform: scalar registers)
                    const unsigned f63 = (unsigned)__builtin_amdgcn_readlane((int)f, 63);
#pragma unroll
                    for (int cc = 0; cc < CB; cc++) {
                        v[cc] += head ? carry[cc] : 0.f;
                        carry[cc] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[cc]), 63));
                    }
                    carry_t = (f63 & 64u) ? -1 : t63;  // flushed at lane 63 (or past the end): nothing to carry
                    if (valid && (f & 64u)) {
                        if (f & 128u) {
#pragma unroll
                            for (int cc = 0; cc < CB; cc++) atomicAdd(&bins[(size_t)cc * n + t], v[cc]);
                        } else {
#pragma unroll
                            for (int cc = 0; cc < CB; cc++) bins[(size_t)cc * n + t] += v[cc];
                        }
                    }
                }
            }
        }
        if (MODE == 1) {
            __syncthreads();
            // self term of the first half: bin i loses sum_j g[c][i,j] (one thread per (channel, point): plain updates)
            for (int q = tid; q < CB * pc; q += T) {
                const int cc = q / pc, il = q - cc * pc;
                const float *row = buf + (size_t)cc * kEsCE + (size_t)il * k;
                float sum = 0.f;
                for (int j = 0; j < k; j++) sum += row[j];
                bins[(size_t)cc * n + i0 + il] -= sum;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < CB * n; i += T) {
        const int cc = i / n, t = i - cc * n;
        if (c0 + cc < c) {
            float *dst = grad_x + ((size_t)smp * c + c0 + cc) * n + t;
            *dst = MODE == 1 ? *dst + bins[i] : bins[i];  // (MODE 1: on top of the second half's self term)
        }
    }
}

// One wave per (b,c) row: max (first maximum), argmax and mean over n.  HBM-bound (the encoder's [B,1024,N] tensor is
// 256 MiB at B=32, N=2048): rows are streamed with 16-byte non-temporal loads, eight in flight per lane before the first
// use; a lane's indices only grow, so inside a lane the strict compare already keeps the first maximum and the index
// tie-break is needed only when the lanes meet.
typedef float gp_v4f __attribute__((ext_vector_type(4)));

template <bool VEC>
__global__ __launch_bounds__(256) void global_pool_kernel(int rows, int n, const float *__restrict__ x,
                                                           float *__restrict__ out_max, int32_t *__restrict__ argmax,
                                                           float *__restrict__ out_mean) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *r = x + (size_t)row * n;
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
    float sum = 0.f;
    if (VEC) {
        constexpr int U = 8;
        const gp_v4f *r4 = reinterpret_cast<const gp_v4f *>(r);
        const int n4 = n >> 2;
        for (int i0 = lane; i0 < n4; i0 += 64 * U) {
            gp_v4f v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + 64 * u;
                v[u] = i < n4 ? __builtin_nontemporal_load(r4 + i)
                              : gp_v4f{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + 64 * u;
                if (i < n4) sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
                const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // NaN wins and sticks (torch.max propagates NaN): the first NaN of the lane is kept
                    const bool gt = e[q] > best || (e[q] != e[q] && best == best);
                    best = gt ? e[q] : best;
                    bi = gt ? 4 * i + q : bi;
                }
            }
        }
    } else {
        for (int i = lane; i < n; i += 64) {
            const float v = r[i];
            sum += v;
            const bool gt = v > best || (v != v && best == best);
            best = gt ? v : best;
            bi = gt ? i : bi;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        sum += __shfl_down(sum, off, 64);
        const bool on = ov != ov, bn = best != best;
        const bool gt = on ? (!bn || oi < bi) : (!bn && (ov > best || (ov == best && oi < bi)));
        best = gt ? ov : best;
        bi = gt ? oi : bi;
    }
    if (lane == 0) {
        if (out_max) out_max[row] = best;
        if (argmax) argmax[row] = bi;
        if (out_mean) out_mean[row] = sum / (float)n;
    }
}

int check(const char *who, int b, int c, int n, int k) {
    if (b < 0 || c < 1 || n < 0 || k < 1) return pcc::invalid(who);
    if (b > 65535 || (long long)n * k > 0x7fffffffLL) return pcc::invalid(who);
    return PCC_OK;
}

template <int MODE>
int gather_fwd(int b, int c, int n, int k, const float *x, const int64_t *indices, float *out, int32_t *argmax,
               hipStream_t st, const char *what, int64_t *tsel = nullptr) {
    int cb = 8;
    while (cb > 1 && (size_t)cb * n * sizeof(float) > 64 * 1024) cb >>= 1;
    size_t lds = (size_t)cb * n * sizeof(float);
    if (lds > 160 * 1024) return pcc::invalid("graph op: n too large for the LDS row tile");
    // the per-point modes stage each wave's 64 x k neighbour indices (16 bit) behind the rows when that still fits
    int stage_off = 0;
    const size_t rows_pad = (lds + 15) / 16 * 16, stage_bytes = (size_t)16 * 64 * k * sizeof(unsigned short);
    if (MODE >= 2 && rows_pad + stage_bytes <= 160 * 1024) {
        stage_off = (int)(rows_pad / sizeof(float));
        lds = rows_pad + stage_bytes;
    }
    const dim3 grid((unsigned)(pcc::ceil_div(c, cb) * b));
    pcc::ProfScope prof(what, st);
#define PCC_LAUNCH(CB)                                                                                              \
    do {                                                                                                            \
        static bool attr = hipFuncSetAttribute(reinterpret_cast<const void *>(gather_lds_kernel<MODE, CB>),         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; \
        (void)attr;                                                                                                 \
        hipLaunchKernelGGL((gather_lds_kernel<MODE, CB>), grid, dim3(1024), lds, st, c, n, k, x, indices, out, argmax, tsel, \
                           stage_off);                                                                              \
    } while (0)
    switch (cb) {
    case 8: PCC_LAUNCH(8); break;
    case 4: PCC_LAUNCH(4); break;
    case 2: PCC_LAUNCH(2); break;
    default: PCC_LAUNCH(1); break;
    }
#undef PCC_LAUNCH
    return pcc::check_launch(what);
}

// gather / edge-feature backward as a stream over the gradient (edge_chunk_sort_kernel + edge_stream_bwd_kernel);
// returns -1 when the sizes do not qualify (the caller then runs the per-edge scatter)
template <int MODE>
int edge_stream_bwd(int b, int c, int n, int k, const int64_t *indices, const float *g, float *grad_x, hipStream_t st) {
    const bool enabled = pcc::tuning(PCC_TUNE_EDGE_SCATTER) == 0;  // (measurement switch: the per-edge atomic scatter)
    const int P = es_points_per_chunk(n, k);
    if (!enabled || P == 0 || n > 32768) return -1;
    int cb = 2;
    auto lds_of = [&](int v) { return (size_t)v * kEsCE * sizeof(float) + (size_t)v * n * sizeof(float); };
    if (lds_of(cb) > 80 * 1024) cb = 1;
    const size_t lds = lds_of(cb), lds_sort = ((size_t)2 * n + kEsCE) * sizeof(int);
    if (lds > 160 * 1024 - 256 || lds_sort > 160 * 1024 - 256) return -1;
    const size_t nk = (size_t)n * k;
    char *ws = nullptr;
    const size_t ent_bytes = ((size_t)b * nk * sizeof(unsigned) + 15) & ~(size_t)15;
    if (pcc::ws_malloc(reinterpret_cast<void **>(&ws), ent_bytes + (size_t)b * nk, st) != hipSuccess) {
        (void)hipGetLastError();
        pcc::set_error(PCC_ENOMEM, "graph op backward: workspace allocation failed");
        return PCC_ENOMEM;
    }
    unsigned *ent = reinterpret_cast<unsigned *>(ws);
    unsigned char *flg = reinterpret_cast<unsigned char *>(ws + ent_bytes);
    static bool attr_sort = [] {
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(edge_chunk_sort_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        return ok;
    }();
    (void)attr_sort;
    {
        pcc::ProfScope prof("edge_chunk_sort_kernel", st);
        hipLaunchKernelGGL(edge_chunk_sort_kernel, dim3((unsigned)pcc::ceil_div(n, P), (unsigned)b), dim3(1024), lds_sort, st, n, k, P,
                           indices, ent, flg);
    }
    if (MODE == 1) {
        const long long wgs = (long long)pcc::ceil_div(n, kSelfPts) * b * c;
        if (wgs > 0x7fffffffLL) {
            (void)pcc::ws_free(ws, st);
            return -1;
        }
        static bool attr_self = [] {
            const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(edge_self_sum_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            return ok;
        }();
        (void)attr_self;
        pcc::ProfScope prof("edge_self_sum_kernel", st);
        hipLaunchKernelGGL(edge_self_sum_kernel, dim3((unsigned)wgs), dim3(256), (size_t)kSelfPts * k * sizeof(float), st, c, n, k, g,
                           grad_x);
    }
    const dim3 grid((unsigned)(pcc::ceil_div(c, cb) * b));
    {
        pcc::ProfScope prof(MODE == 1 ? "edge_stream_bwd_kernel<features>" : "edge_stream_bwd_kernel<gather>", st);
#define PCC_LAUNCH_E(CB)                                                                                                  \
    do {                                                                                                                  \
        static bool attr = [] {                                                                                           \
            const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(edge_stream_bwd_kernel<MODE, CB>),         \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) == hipSuccess; \
            if (!ok) (void)hipGetLastError();                                                                             \
            return ok;                                                                                                    \
        }();                                                                                                              \
        (void)attr;                                                                                                       \
        hipLaunchKernelGGL((edge_stream_bwd_kernel<MODE, CB>), grid, dim3(kEsT), lds, st, c, n, k, P, ent, flg, g, grad_x); \
    } while (0)
        if (cb == 2) PCC_LAUNCH_E(2);
        else PCC_LAUNCH_E(1);
#undef PCC_LAUNCH_E
    }
    (void)pcc::ws_free(ws, st);
    return pcc::check_launch("graph op backward (edge stream)");
}

template <int MODE>
int scatter_bwd(int b, int c, int n, int k, const int64_t *indices, const int32_t *argmax, const float *g,
                float *grad_x, hipStream_t st, const char *what) {
    if (MODE == 0 || MODE == 1) {
        const int rc = edge_stream_bwd<(MODE == 1 ? 1 : 0)>(b, c, n, k, indices, g, grad_x, st);
        if (rc >= 0) return rc;
    }
    // channels per workgroup: as many as fit 64 KiB of bins (two workgroups per CU)
    int cb = 8;
    while (cb > 1 && (size_t)cb * n * sizeof(float) > 64 * 1024) cb >>= 1;
    const size_t lds = (size_t)cb * n * sizeof(float);
    if (lds > 160 * 1024) return pcc::invalid("graph op backward: n too large for the LDS bins");
    const dim3 grid((unsigned)(pcc::ceil_div(c, cb) * b));
    pcc::ProfScope prof(what, st);
#define PCC_LAUNCH(CB)                                                                                               \
    do {                                                                                                             \
        static bool attr = hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_lds_kernel<MODE, CB>),         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; \
        (void)attr;                                                                                                  \
        hipLaunchKernelGGL((scatter_lds_kernel<MODE, CB>), grid, dim3(1024), lds, st, c, n, k, indices, argmax, g,   \
                           grad_x);                                                                                  \
    } while (0)
    switch (cb) {
    case 8: PCC_LAUNCH(8); break;
    case 4: PCC_LAUNCH(4); break;
    case 2: PCC_LAUNCH(2); break;
    default: PCC_LAUNCH(1); break;
    }
#undef PCC_LAUNCH
    return pcc::check_launch(what);
}

}  // namespace

extern "C" {

int pcc_gather_neighbours(int b, int c, int n, int k, const float *x, const int64_t *indices, float *out,
                          pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("gather_neighbours: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!x || !indices || !out) return pcc::invalid("gather_neighbours: null pointer");
    return gather_fwd<0>(b, c, n, k, x, indices, out, nullptr, static_cast<hipStream_t>(stream), "gather_lds_kernel<gather>");
}

int pcc_graph_features(int b, int c, int n, int k, const float *x, const int64_t *indices, float *out,
                       pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("graph_features: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!x || !indices || !out) return pcc::invalid("graph_features: null pointer");
    return gather_fwd<1>(b, c, n, k, x, indices, out, nullptr, static_cast<hipStream_t>(stream), "gather_lds_kernel<features>");
}

int pcc_graph_max_pool(int b, int c, int n, int k, const float *x, const int64_t *indices, float *out,
                       int32_t *argmax, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("graph_max_pool: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!x || !indices || !out) return pcc::invalid("graph_max_pool: null pointer");
    return gather_fwd<2>(b, c, n, k, x, indices, out, argmax, static_cast<hipStream_t>(stream), "gather_lds_kernel<maxpool>");
}

int pcc_gather_neighbours_bwd(int b, int c, int n, int k, const int64_t *indices, const float *grad_out,
                              float *grad_x, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("gather_neighbours_bwd: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!indices || !grad_out || !grad_x) return pcc::invalid("gather_neighbours_bwd: null pointer");
    return scatter_bwd<0>(b, c, n, k, indices, nullptr, grad_out, grad_x, static_cast<hipStream_t>(stream),
                          "scatter_bwd_kernel<gather>");
}

int pcc_graph_features_bwd(int b, int c, int n, int k, const int64_t *indices, const float *grad_out,
                           float *grad_x, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("graph_features_bwd: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!indices || !grad_out || !grad_x) return pcc::invalid("graph_features_bwd: null pointer");
    return scatter_bwd<1>(b, c, n, k, indices, nullptr, grad_out, grad_x, static_cast<hipStream_t>(stream),
                          "scatter_bwd_kernel<features>");
}

int pcc_graph_max_pool_bwd(int b, int c, int n, int k, const int64_t *indices, const int32_t *argmax,
                           const float *grad_out, float *grad_x, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("graph_max_pool_bwd: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!indices || !argmax || !grad_out || !grad_x) return pcc::invalid("graph_max_pool_bwd: null pointer");
    return scatter_bwd<2>(b, c, n, k, indices, argmax, grad_out, grad_x, static_cast<hipStream_t>(stream),
                          "scatter_bwd_kernel<maxpool>");
}

int pcc_neighbour_sum(int b, int c, int n, int k, const float *x, const int64_t *indices, float *out,
                      pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("neighbour_sum: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!x || !indices || !out) return pcc::invalid("neighbour_sum: null pointer");
    return gather_fwd<3>(b, c, n, k, x, indices, out, nullptr, static_cast<hipStream_t>(stream), "gather_lds_kernel<nbrsum>");
}

int pcc_neighbour_sum_bwd(int b, int c, int n, int k, const int64_t *indices, const float *grad_out, float *grad_x,
                          pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("neighbour_sum_bwd: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!indices || !grad_out || !grad_x) return pcc::invalid("neighbour_sum_bwd: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool sorted_enabled = pcc::tuning(PCC_TUNE_NBRSUM_SCATTER) == 0;  // (measurement switch: the per-edge atomic scatter)
    // sorted-edge schedule: needs 16-bit point ids and rows + bins of >= 1 channel in LDS
    if (sorted_enabled && n <= 65536 && (size_t)n * 8 <= 128 * 1024 && (size_t)n * 4 <= 160 * 1024 - 256) {
        unsigned *rev = nullptr;
        if (pcc::ws_malloc(reinterpret_cast<void **>(&rev), (size_t)b * n * k * sizeof(unsigned), st) != hipSuccess) {
            (void)hipGetLastError();
            pcc::set_error(PCC_ENOMEM, "neighbour_sum_bwd: workspace allocation failed");
            return PCC_ENOMEM;
        }
        static bool attr_sort = [] {
            const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(edge_sort_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            return ok;
        }();
        (void)attr_sort;
        {
            pcc::ProfScope prof("edge_sort_kernel", st);
            hipLaunchKernelGGL(edge_sort_kernel, dim3(b), dim3(1024), (size_t)n * sizeof(int), st, n, k, indices, rev);
        }
        int cb = 4;
        while (cb > 1 && (size_t)2 * cb * n * sizeof(float) > 64 * 1024) cb >>= 1;
        const size_t lds = (size_t)2 * cb * n * sizeof(float);
        const dim3 grid((unsigned)(pcc::ceil_div(c, cb) * b));
        {
            pcc::ProfScope prof("nbrsum_bwd_sorted_kernel", st);
#define PCC_LAUNCH_S(CB)                                                                                                  do {                                                                                                                      static bool attr = [] {                                                                                                   const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(nbrsum_bwd_sorted_kernel<CB>),                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;             if (!ok) (void)hipGetLastError();                                                                                     return ok;                                                                                                        }();                                                                                                                  (void)attr;                                                                                                           hipLaunchKernelGGL((nbrsum_bwd_sorted_kernel<CB>), grid, dim3(1024), lds, st, c, n, k, rev, grad_out, grad_x);     } while (0)
            switch (cb) {
            case 4: PCC_LAUNCH_S(4); break;
            case 2: PCC_LAUNCH_S(2); break;
            default: PCC_LAUNCH_S(1); break;
            }
#undef PCC_LAUNCH_S
        }
        (void)pcc::ws_free(rev, st);
        return pcc::check_launch("neighbour_sum_bwd(sorted)");
    }
    return scatter_bwd<3>(b, c, n, k, indices, nullptr, grad_out, grad_x, st, "scatter_lds_kernel<nbrsum>");
}

int pcc_neighbour_minmax_target(int b, int c, int n, int k, const float *x, const int64_t *indices, int64_t *tsel,
                                pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check("neighbour_minmax_target: bad size", b, c, n, k)) return rc;
    if (b == 0 || n == 0) return PCC_OK;
    if (!x || !indices || !tsel) return pcc::invalid("neighbour_minmax_target: null pointer");
    return gather_fwd<4>(b, c, n, k, x, indices, nullptr, nullptr, static_cast<hipStream_t>(stream),
                         "gather_lds_kernel<minmax>", tsel);
}

int pcc_global_pool(int b, int c, int n, const float *x, float *out_max, int32_t *argmax, float *out_mean,
                    pcc_stream_t stream) {
    pcc::clear_error();
    if (b < 0 || c < 0 || n < 1) return pcc::invalid("global_pool: bad size");
    if (b == 0 || c == 0) return PCC_OK;
    if (!x) return pcc::invalid("global_pool: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long rows = (long long)b * c;
    if (rows > 0x7fffffffLL) return pcc::invalid("global_pool: too many rows");
    pcc::ProfScope prof("global_pool_kernel", st);
    const bool vec = n % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(global_pool_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (int)rows, n, x,
                           out_max, argmax, out_mean);
    else
        hipLaunchKernelGGL(global_pool_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (int)rows, n, x,
                           out_max, argmax, out_mean);
    return pcc::check_launch("global_pool");
}

}  // extern "C"
