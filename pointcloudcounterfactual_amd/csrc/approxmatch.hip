// Approximate EMD (multi-scale soft matching) for gfx950 (MI355X), wave64.
//
// Replaces approxmatchkernel / matchcostkernel / matchcostgrad{1,2}kernel and their launchers
// (external/pytorch_structural_losses/src/approxmatch.cu:3-326).  Same recurrence, same outputs
// (match[b,m,n], temp[b,2(n+m)], cost[b], grad1, grad2); a different machine mapping:
//
//   reference                                   | here
//   --------------------------------------------+----------------------------------------------------
//   one 512-thread block per sample (32 blocks)  | every pass is a chip-wide launch: workgroup = 64*R
//   runs all 27 all-pairs passes serially        | owners x whole candidate cloud (SoA in LDS), S waves
//                                                | split the candidates, partial sums merged in LDS in a
//                                                | fixed order (deterministic)
//   pass C of level j and pass A of level j-1    | fused ("CA"): one distance evaluation feeds both
//   are separate sweeps                          | exponentials -> 19 launches instead of 27 sweeps
//   match zero-filled, then read-modify-written  | per-level ratio vectors (18(n+m) floats per sample)
//   once per level (9 x 1 GiB of traffic at      | are kept in a workspace and match is materialised by
//   B=32,N=2048)                                 | ONE write-only pass that re-evaluates the 9 levels in
//                                                | registers, summing them in the reference's order
//   exp via __expf(level*d2)                     | v_exp_f32((level*log2e)*d2): level is a power of 4, so
//                                                | the single rounded product is the same real number
//   matchcost / grad re-read match 3x            | materialising entry points: cost accumulated by the pass that
//                                                | writes match, both gradients from ONE read; the Python-level
//                                                | match_cost never stores match at all (am_pair_kernel: cost and
//                                                | gradients straight from registers, pcc_match_cost)
//   every pass evaluates all n*m pairs           | only terms that are not EXACTLY zero in float32: both clouds are
//                                                | Hilbert-sorted once per call (register bitonic sort); at the
//                                                | fine levels (64-owner group, 16-candidate block) pairs whose box
//                                                | distance makes every exp2(level*d2) underflow are skipped
//                                                | (V_CULL), and points whose capacity is used up drop out as
//                                                | candidates (V_CCAND / V_CLIST) and as owners (V_COWN)
//   one stream, one block per sample             | a large batch runs as two half-batch lanes on two streams so
//                                                | that the dependent launch chains fill each other's bubbles
//
// Rooflines: the 19 phase launches, am_pair_kernel and the materialise pass are f32-VALU /
// transcendental bound; matchcost and the fused gradient kernel of the materialising path are HBM bound
// (one read of match each).
#include "pcc_common.hpp"
#include "pcc_test_hooks.h"

#include <algorithm>
#include <functional>
#include <mutex>
#include <type_traits>

namespace {

using pcc::sq3;

typedef float v4f __attribute__((ext_vector_type(4)));  // for __builtin_nontemporal_load/store

constexpr int kLevels = 9;       // j = 7 .. -1, level = -4^j            (approxmatch.cu:24-25)
constexpr float kLog2e = 1.44269504088896340736f;

struct LevelConsts {
    float c[kLevels];            // level_j * log2(e), exact scalings of fl(log2 e)
};

__host__ LevelConsts make_levels() {
    LevelConsts lc;
    float level = -16384.0f;     // -4^7
    for (int i = 0; i < kLevels; i++) {
        lc.c[i] = level * kLog2e;
        level *= 0.25f;
    }
    return lc;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---------------------------------------------------------------------------------------------------
// Phase kernels: for every owner point o, S_c(o) = sum over candidates q of exp2(c_c * |o-q|^2) * w_c[q].
//   PH_A  (first level only): owners = set1, w0 = multiR (constant)      -> ratioL_0          (:29-62)
//   PH_B : owners = set2, w0 = ratioL_i                                  -> ratioR_i, remainR (:78-111)
//   PH_CA: owners = set1, w0 = ratioR_i, w1 = remainR, two exponents     -> remainL, ratioL_{i+1}
//          (pass C :130-163 without the match write, fused with pass A of the next level :29-62)
//   PH_C : last level, pass C only.
// ---------------------------------------------------------------------------------------------------
enum Phase { PH_A = 0, PH_B = 1, PH_CA = 2, PH_C = 3 };

constexpr int kDbgInts = 64 + 24 * 19;  // counters + per-phase stamps
constexpr int kBox = 16;          // points per bounding-box block (sorted order)
constexpr int kLiveRow = 16;      // ints per sample in the live-owner counters (one per level, padded)
constexpr int kInfSlot = 13;      // live-counter row, slots 13 / 14: set1 / set2 of the sample holds an infinite coordinate
constexpr float kZeroExp = 151.f; // exp2(x) == 0 exactly for x <= -150 (below the smallest f32 subnormal)

struct PhaseArgs {
    int n_own, n_cand, tiles, batch;   // tiles = ceil(n_own / (64 R))
    int own_n4, cand_n4, own_nb, cand_nb;
    const float *own_soa, *cand_soa;   // [b][3][n4] Hilbert-sorted coordinates
    const float *cand_box;             // [b][nb][8] per 16 sorted candidates (min xyz, 0, max xyz, 0)
    const float *own_box16;            // [b][own_nb][8] per 16 sorted owners (am_fine_kernel)
    const float *w0, *w1;              // per-candidate weights in sorted order (w0 may be null => w0c)
    long long w0_stride, w1_stride;    // per-sample strides in floats
    float w0c;
    float c0, c1;
    float cut2;                        // skip a candidate block when its box is farther than sqrt(cut2)
    float cut2_fine;                   // pass C/A: the radius of the FINER of its two levels (am_fine_kernel: one exponential beyond it)
    int first;                         // first level: remain* still hold their initial constants
    int level;                         // 0..8 (host bookkeeping only)
    float multiL, multiR;
    // epilogue operands, all indexed [sample * stride + owner (sorted position)]
    float *remain;                     // remainL (CA/C, updated in place by its owner) or remainR before pass B
    float *remain_out;                 // pass B: remainR after the pass (a second buffer: other workgroups of the
                                       // launch still read `remain` to find the live owners)
    long long remain_stride;
    const float *ratio_in;             // CA/C: ratioL_i
    float *ratio_out;                  // A: ratioL_0 ; B: ratioR_i ; CA: ratioL_{i+1}
    long long ratio_stride;            // per-sample stride of the level arrays
    float *clist;                      // dense candidate list [b][5][cl_n4]: x | y | z | ratioR | remainR (V_COWN writes, V_CLIST reads)
    int *clist_cnt;                    // [b] entries in the list
    int cl_n4;
    const unsigned *mask_in;           // [b][kLevels][mask_words] bit o of row i: set2 point o is live entering pass B of level i (V_COWN)
    unsigned *mask_out;                // row i + 1 (pass B of level 2 writes row 3 whole; later ones clear the bits of the owners they exhaust)
    int mask_words;
    const int *live_in;                // [b] live owners (remain != 0) of this pass B, counted by the previous pass B (V_COWN)
    int *live_out;                     // [b] pass B: live owners of the next level's pass B (integer atomics: order-free)
    int *dbg;                          // optional [2] counters: blocks visited / skipped (debug builds of the host)
    int *stamp;                        // optional [3][8] s_memrealtime stamps of the first / middle / last workgroup (PCC_AM_DEBUG=2)
};

// Work-skipping variants of a pass launch.  All of them only drop terms that are EXACTLY zero:
//   V_CULL  (levels 0-2, am_fine_kernel): a (16-owner group, 16-candidate block) pair is skipped when the box distance
//           makes every exp2(c*d2) underflow to 0;
//   V_COWN  (pass B from level 3 on): a query point whose capacity is used up has remainR == ratioR == 0 from then on
//           (approxmatch.cu:108-109; 54 % of them by level 3 and 95 % by level 8 on the bench clouds): its outputs are
//           ratioR = 0, remainR = 0 whatever the sum is, so only the live owners are gathered into tiles (the level
//           arrays are zero-filled beforehand) and workgroups beyond the live count leave at once;
//   V_CLIST (pass C/A from level 3 on): the same exhausted points as CANDIDATES: the live owners of the preceding
//           V_COWN pass B ARE the candidates with a non-zero weight (ratioR_i = consumption * remainR != 0 exactly for
//           them), so that pass writes their coordinates and new weights as a dense list and this one stages it with
//           straight float4 copies (no scan, no scattered LDS stores).
enum Var { V_PLAIN = 0, V_CULL = 1, V_COWN = 3, V_CLIST = 4 };

// Everything a pass needs is derived from this small description of one approxmatch call: the same function
// builds the arguments of pass p.
//   p = 0: pass A of level 0;  p = 1 + 2i: pass B of level i;  p = 2 + 2i: pass C of level i fused with pass A of
//   level i+1 (plain pass C for the last level).
struct Sched {
    int n, m, n4, m4, nb1, nb2;
    const float *soa1, *soa2, *box1, *box2;
    float *rem, *lv;               // sorted space: remain row = remainL(n4) | remainR ping(m4) | pong(m4); level rows
    float multiL, multiR;
    float *clist;                  // dense candidate list handed from pass B to pass C/A (null: pass C/A compacts itself)
    int *clist_cnt;
    int *live_cnt;                 // [b][kLiveRow] live set2 points entering pass B of level i (zeroed by the sort; i >= 1)
    unsigned *live_mask;           // [b][kLevels][ceil(m4 / 32)] live bits of set2 per level (rows 4.. preset to ones by the sort)
    int skip;                      // work-skipping variants enabled
    LevelConsts lc;
    int *dbg;
    int dbg_counts;                // 1: count visited / skipped blocks (slow: one atomic per block)
};

__host__ __device__ inline int sched_phases() { return 2 * kLevels + 1; }

__host__ __device__ inline PhaseArgs build_phase(const Sched &sc, int p, int *mode_out, int *var_out) {
    const long long nm4 = (long long)sc.n4 + sc.m4;
    const long long rs = (long long)sc.n4 + 2LL * sc.m4;
    PhaseArgs a{};
    a.multiL = sc.multiL;
    a.multiR = sc.multiR;
    const bool set1_owns = (p == 0) || (p % 2 == 0);
    if (set1_owns) {  // owners = set1, candidates = set2
        a.n_own = sc.n; a.n_cand = sc.m; a.own_n4 = sc.n4; a.cand_n4 = sc.m4; a.own_nb = sc.nb1; a.cand_nb = sc.nb2;
        a.own_soa = sc.soa1; a.cand_soa = sc.soa2; a.cand_box = sc.box2;
        a.own_box16 = sc.box1;
    } else {          // owners = set2, candidates = set1
        a.n_own = sc.m; a.n_cand = sc.n; a.own_n4 = sc.m4; a.cand_n4 = sc.n4; a.own_nb = sc.nb2; a.cand_nb = sc.nb1;
        a.own_soa = sc.soa2; a.cand_soa = sc.soa1; a.cand_box = sc.box1;
        a.own_box16 = sc.box2;
    }
    int mode, var;
    if (p == 0) {
        mode = PH_A;
        var = sc.skip ? V_CULL : V_PLAIN;
        a.level = 0;
        a.w0 = nullptr; a.w0c = sc.multiR; a.c0 = sc.lc.c[0]; a.first = 1;
        a.cut2 = kZeroExp / -sc.lc.c[0];
        a.ratio_out = sc.lv; a.ratio_stride = kLevels * nm4;
        a.dbg = sc.dbg_counts ? sc.dbg : nullptr;
    } else {
        const int i = (p - 1) / 2;
        float *ratioL = sc.lv + (size_t)i * nm4, *ratioR = ratioL + sc.n4;
        a.level = i;
        a.first = (i == 0);
        a.c0 = sc.lc.c[i];
        if (p % 2 == 1) {  // pass B of level i
            mode = PH_B;
            // box culling while the zero radius is small against the cloud (levels 0-2), live-owner compaction after
            var = !sc.skip ? V_PLAIN : i <= 2 ? V_CULL : V_COWN;
            a.w0 = ratioL; a.w0_stride = kLevels * nm4;
            a.cut2 = kZeroExp / -sc.lc.c[i];
            a.remain = sc.rem + sc.n4 + (i & 1) * sc.m4;
            a.remain_out = sc.rem + sc.n4 + ((i + 1) & 1) * sc.m4;
            a.remain_stride = rs;
            a.ratio_out = ratioR; a.ratio_stride = kLevels * nm4;
            if (var == V_COWN) { a.clist = sc.clist; a.clist_cnt = sc.clist_cnt; a.cl_n4 = sc.m4; }
            if (sc.live_mask) {
                const int words = (sc.m4 + 31) / 32;
                a.mask_words = words;
                if (i + 1 < kLevels && (var == V_COWN || i == 2)) a.mask_out = sc.live_mask + (size_t)(i + 1) * words;
                if (var == V_COWN) {
                    a.mask_in = sc.live_mask + (size_t)i * words;
                    // liveness comes from the mask, so remainR is updated IN PLACE by its owner from level 3 on (level 2
                    // left it in the second buffer): an exhausted owner holds 0 from the pass that exhausted it
                    a.remain = a.remain_out = sc.rem + sc.n4 + sc.m4;
                }
            }
            if (sc.live_cnt) {  // strided by kLiveRow ints per sample: the kernels index [smp * kLiveRow]
                a.live_in = (var == V_COWN && i >= 1) ? sc.live_cnt + i : nullptr;
                a.live_out = i + 1 < kLevels ? sc.live_cnt + i + 1 : nullptr;
            }
            a.dbg = sc.dbg_counts ? sc.dbg + 2 + 4 * i : nullptr;
        } else {           // pass C of level i (+ pass A of level i+1)
            mode = i + 1 < kLevels ? PH_CA : PH_C;
            // (the cull radius is the one of level i+1); from level 3 on pass B has left the dense candidate list
            // (level 2: the (16 x 16) boxes still drop half of the pairs at level 3's radius, more than compacting away
            // the exhausted candidates did)
            var = !sc.skip ? V_PLAIN : i <= 2 ? V_CULL : V_CLIST;
            if (var == V_CLIST) { a.clist = sc.clist; a.clist_cnt = sc.clist_cnt; a.cl_n4 = sc.m4; }
            a.w0 = ratioR; a.w0_stride = kLevels * nm4;
            a.w1 = sc.rem + sc.n4 + ((i + 1) & 1) * sc.m4; a.w1_stride = rs;
            a.remain = sc.rem; a.remain_stride = rs;
            a.ratio_in = ratioL; a.ratio_stride = kLevels * nm4;
            const int lc_i = i + 1 < kLevels ? i + 1 : i;
            a.cut2 = kZeroExp / -sc.lc.c[lc_i];  // the coarser of the two levels decides what is 0
            a.cut2_fine = kZeroExp / -sc.lc.c[i];
            if (i + 1 < kLevels) {
                a.c1 = sc.lc.c[i + 1];
                a.ratio_out = sc.lv + (size_t)(i + 1) * nm4;
            }
            a.dbg = sc.dbg_counts ? sc.dbg + 4 + 4 * i : nullptr;
        }
    }
    a.stamp = (sc.dbg && !sc.dbg_counts) ? sc.dbg + 64 + 24 * p : nullptr;
    *mode_out = mode;
    *var_out = var;
    return a;
}

// LDS footprint of one phase (floats / ints)
template <int NW, int R, int S, int CH>
struct PhaseLds {
    static constexpr int kC = (3 + NW) * CH;          // x | y | z | w0 | (w1)
    static constexpr int kRed = NW * S * 64 * R;      // partial sums [NW][S][TQ]
    static constexpr int kOwn = 64 * R;               // live-owner tile (ints)
    static constexpr int kWave = S;                   // (ints)
    static constexpr int floats = kC + kRed + kOwn + kWave + 4;
};

// G > 1 (owner-compacted passes of the late levels only): a wave holds 64 / G owners, each on G lanes that split the
// 16-candidate blocks among them (G-fold shorter pair loop for the few live owners left; the partial sums of the G
// lanes meet through log2(G) shuffles, in a fixed order).
template <int MODE, int R, int S, int CH, int VAR, int G = 1>
__device__ __forceinline__ void am_phase_body(const PhaseArgs &a, int smp, int tile, float *smem) {
    constexpr int T = 64 * S;
    constexpr int TQ = 64 * R / G;     // owners per workgroup
    constexpr int PQ = 64 * R;         // pitch of the partial-sum rows in LDS
    static_assert(G == 1 || (VAR == V_COWN && R == 1), "lane-split owners: owner-compacted launches only");
    constexpr int NW = (MODE == PH_CA) ? 2 : 1;
    constexpr bool W0_CONST = (MODE == PH_A);
    constexpr bool COWN = VAR == V_COWN, CLIST = VAR == V_CLIST;
    static_assert(VAR == V_PLAIN || COWN || CLIST, "the box-culled passes run on am_fine_kernel");
    static_assert(!(CLIST && W0_CONST), "pass A of the first level has constant weights");
    static_assert(TQ <= T, "one epilogue owner per thread");
    using L = PhaseLds<NW, R, S, CH>;  // (four candidate rows for the single-weight passes: 36 KB, four workgroups per CU)
    float *lds_c = smem;                               // x | y | z | w0 | (w1)
    float *red = smem + L::kC;                         // [NW][S][TQ]
    int *own_idx = reinterpret_cast<int *>(red + L::kRed);
    int *wave_cnt = own_idx + L::kOwn;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = G > 1 ? lane / TQ : 0;  // which share of every candidate block this lane takes
    const float *O = a.own_soa + (size_t)smp * 3 * a.own_n4;
    const float *C = a.cand_soa + (size_t)smp * 3 * a.cand_n4;
    const float *W0 = W0_CONST ? nullptr : a.w0 + (size_t)smp * a.w0_stride;
    const float *W1 = (NW == 2) ? a.w1 + (size_t)smp * a.w1_stride : nullptr;

    // debug stamps (100 MHz s_memrealtime) of the first, middle and last workgroup of the launch; a.stamp is null in
    // normal runs (one predicated-off scalar branch per stamp)
    unsigned long long tst0 = 0;
    int *stamp = nullptr;
    if (a.stamp && threadIdx.x == 0) {
        const unsigned bx = blockIdx.x, gx = gridDim.x;
        const int which = bx == 0 ? 0 : bx == gx / 2 ? 1 : bx == gx - 1 ? 2 : -1;
        if (which >= 0) {
            stamp = a.stamp + 8 * which;
            tst0 = __builtin_amdgcn_s_memrealtime();
            stamp[0] = (int)(tst0 & 0x7fffffff);
        }
    }
#define PCC_ST(k) do { if (stamp) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[(k)] = (int)(__builtin_amdgcn_s_memrealtime() - tst0); } } while (0)
    // ---- which owners does this workgroup hold? ----
    if (!COWN && tile * TQ >= a.n_own) return;
    // Candidate staging, split in two: the loads of a chunk (one float4 group per thread and row: CH / 4 <= T) and its
    // LDS stores.  The owner-compacted passes issue the loads of the first chunk HERE, above their owner scan, whose
    // round trips they then share.
    static_assert(CH / 4 <= T, "one float4 group per thread and chunk");
    float4 st_x, st_y, st_z, st_0, st_1;
    auto stage_load = [&](const float *Cc, const float *W0c, const float *W1c, int pitch, int q0, int ngroups) {
        const int i = min(tid, max(ngroups - 1, 0));  // (clamped: every thread loads, only tid < ngroups stores)
        st_x = reinterpret_cast<const float4 *>(Cc + q0)[i];
        st_y = reinterpret_cast<const float4 *>(Cc + (size_t)pitch + q0)[i];
        st_z = reinterpret_cast<const float4 *>(Cc + (size_t)2 * pitch + q0)[i];
        st_0 = make_float4(a.w0c, a.w0c, a.w0c, a.w0c);
        st_1 = st_0;
        if (!W0_CONST) st_0 = *reinterpret_cast<const float4 *>(W0c + q0 + 4 * i);
        if (NW == 2) st_1 = *reinterpret_cast<const float4 *>(W1c + q0 + 4 * i);
    };
    if (COWN) stage_load(C, W0, W1, a.cand_n4, 0, (min(CH, a.n_cand) + 3) / 4);
    int n_valid = min(TQ, a.n_own - tile * TQ);  // owners of this tile (sorted positions tile*TQ ...)
    if (COWN) {
        // live owners of the sample, in order; this workgroup takes the tile-th group of TQ.  Liveness is one bit per
        // owner (row `level` of the mask: written whole by pass B of level 2, later rows = the row before with the bits of
        // the owners exhausted since cleared): a thread takes one 32-owner word -- one count, ONE block scan, and the set
        // bits of the word dealt to the tile's slots.  (Until round 3 every workgroup scanned the 2048 remainR floats, 16
        // per thread, and carried the zeros of the exhausted owners into the second remainR buffer: 3-4 us at the head of
        // each of the six passes, PCC_AM_DEBUG=2 stamps.)
        const int lo = tile * TQ, hi = lo + TQ;
        const unsigned *mk = a.mask_in + (size_t)smp * kLevels * a.mask_words;
        unsigned *mnext = (tile == 0 && a.mask_out) ? a.mask_out + (size_t)smp * kLevels * a.mask_words : nullptr;
        const int wpt = (a.mask_words + T - 1) / T;  // words per thread: 1 up to 32 T = 16384 points
        unsigned word = 0;
        int mine = 0;
        if (wpt == 1) {
            word = mk[min(tid, a.mask_words - 1)];
            word &= -(unsigned)(tid < a.mask_words);
            if (mnext && tid < a.mask_words) atomicAnd(&mnext[tid], word);  // the next row starts as this one (its preset is all ones)
            mine = __popc(word);
        } else {
            for (int j = tid * wpt; j < min(tid * wpt + wpt, a.mask_words); j++) {
                const unsigned wj = mk[j];
                if (mnext) atomicAnd(&mnext[j], wj);
                mine += __popc(wj);
            }
        }
        // exclusive scan of `mine` over the workgroup: wave scan + S-entry LDS scan
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            incl += lane >= off ? v : 0;
        }
        if (lane == 63) wave_cnt[w] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < S; i++) {
            const int c = wave_cnt[i];
            before += i < w ? c : 0;
            total += c;
        }
        int pos = before + incl - mine - lo;  // slot of this thread's first live owner in the tile
        auto deal = [&](unsigned bits, int first_owner) {
            while (bits) {
                const int bit = __builtin_ctz(bits);
                bits &= bits - 1;
                if ((unsigned)pos < (unsigned)TQ) own_idx[pos] = first_owner + bit;
                pos++;
            }
        };
        if (wpt == 1) {
            deal(word, tid * 32);
        } else {
            for (int j = tid * wpt; j < min(tid * wpt + wpt, a.mask_words); j++) deal(mk[j], j * 32);
        }
        if (a.dbg && tile == 0 && tid == 0) {
            atomicAdd(&a.dbg[0], a.n_own);
            atomicAdd(&a.dbg[1], a.n_own - total);
        }
        if (a.clist_cnt && tile == 0 && tid == 0) a.clist_cnt[smp] = total;  // pass C/A stages exactly the live owners
        n_valid = min(TQ, total - lo);
        if (n_valid <= 0) return;  // wave-uniform: nothing live in this tile
        (void)hi;
        __syncthreads();
    }
    int own_e = -1;  // sorted position of the owner this THREAD finishes in the epilogue
    if (tid < TQ && tid < n_valid) own_e = COWN ? own_idx[tid] : tile * TQ + tid;

    float ox[R], oy[R], oz[R], s0[R], s1[R];
    float t0x = 0.f, t1x = 0.f;  // second accumulators of the R == 1 shapes
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = G > 1 ? (lane & (TQ - 1)) : r * 64 + lane;
        int o = COWN ? own_idx[e < n_valid ? e : 0] : tile * TQ + e;
        o = o < a.n_own ? o : a.n_own - 1;
        ox[r] = O[o];
        oy[r] = O[a.own_n4 + o];
        oz[r] = O[2 * a.own_n4 + o];
        s0[r] = 0.f;
        s1[r] = 0.f;
    }
    // operands of the epilogue do not depend on the pair loop: fetch them now, behind the staging traffic
    // (for every thread, from a clamped index: issued with the owner coordinates above, one round trip for all)
    float pre_rem = 0.f, pre_ratio = 0.f;
    {
        const int oe = max(own_e, 0);
        if (MODE != PH_A && !a.first) pre_rem = a.remain[(size_t)smp * a.remain_stride + oe];
        if (MODE == PH_CA || MODE == PH_C) pre_ratio = a.ratio_in[(size_t)smp * a.ratio_stride + oe];
    }
    PCC_ST(1);
    const float4 *X4 = reinterpret_cast<const float4 *>(lds_c);
    const float4 *Y4 = X4 + CH / 4;
    const float4 *Z4 = Y4 + CH / 4;
    const float4 *A4 = Z4 + CH / 4;
    const float4 *B4 = A4 + CH / 4;
    const float c0 = a.c0, c1 = a.c1;

    const int n_cand = CLIST ? __builtin_amdgcn_readfirstlane(a.clist_cnt[smp]) : a.n_cand;
    if (CLIST) {  // dense list of this sample: x | y | z | ratioR (w0) | remainR (w1), rows of cl_n4 floats
        C = a.clist + (size_t)smp * 5 * a.cl_n4;
        W0 = C + (size_t)3 * a.cl_n4;
        W1 = C + (size_t)4 * a.cl_n4;
    }
    const int cand_pitch = CLIST ? a.cl_n4 : a.cand_n4;
    for (int q0 = 0; q0 < n_cand; q0 += CH) {
        const int cnt = min(CH, n_cand - q0);
        int ngroups = (cnt + 3) / 4;
        if (q0) __syncthreads();
        {
            // sorted SoA rows and the weight rows are padded to a multiple of 4 (zeros): straight float4 copies
            float4 *dst4 = reinterpret_cast<float4 *>(lds_c);
            if (!(COWN && q0 == 0)) stage_load(C, W0, W1, cand_pitch, q0, ngroups);
            const int i = tid;
            if (i < ngroups) {
                float4 vx = st_x, vy = st_y, vz = st_z, v0 = st_0, v1 = st_1;
                if (CLIST && i * 4 + 3 >= cnt) {
                    // the dense list's tail is stale scratch: a padded candidate gets weight 0 below AND finite
                    // coordinates here (0 * exp2(NaN) would be NaN, not the exact 0 a padded candidate must add)
                    vx.y = i * 4 + 1 < cnt ? vx.y : 0.f; vx.z = i * 4 + 2 < cnt ? vx.z : 0.f; vx.w = 0.f;
                    vy.y = i * 4 + 1 < cnt ? vy.y : 0.f; vy.z = i * 4 + 2 < cnt ? vy.z : 0.f; vy.w = 0.f;
                    vz.y = i * 4 + 1 < cnt ? vz.y : 0.f; vz.z = i * 4 + 2 < cnt ? vz.z : 0.f; vz.w = 0.f;
                }
                if ((W0_CONST || CLIST) && i * 4 + 3 >= cnt) {  // padded candidates must weigh 0 (the list's tail is stale)
                    v0.x = i * 4 + 0 < cnt ? v0.x : 0.f;
                    v0.y = i * 4 + 1 < cnt ? v0.y : 0.f;
                    v0.z = i * 4 + 2 < cnt ? v0.z : 0.f;
                    v0.w = 0.f;
                    if (NW == 2) {
                        v1.x = i * 4 + 0 < cnt ? v1.x : 0.f;
                        v1.y = i * 4 + 1 < cnt ? v1.y : 0.f;
                        v1.z = i * 4 + 2 < cnt ? v1.z : 0.f;
                        v1.w = 0.f;
                    }
                }
                dst4[i] = vx;
                dst4[CH / 4 + i] = vy;
                dst4[2 * (CH / 4) + i] = vz;
                dst4[3 * (CH / 4) + i] = v0;
                if (NW == 2) dst4[4 * (CH / 4) + i] = v1;
            }
        }
        const int nblk = (ngroups + 3) / 4;  // blocks of 16 candidates (4 groups)
        __syncthreads();
        PCC_ST(3);
        // the S waves take the 16-candidate blocks round-robin
        for (int blk = w; blk < nblk; blk += S) {
            const int g_end = min(blk * 4 + 4, ngroups);
            for (int g = blk * 4 + sub; g < g_end; g += G) {
                const float4 x = X4[g], y = Y4[g], z = Z4[g], wa = A4[g];
                float4 wb;
                if (NW == 2) wb = B4[g];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    // (x2-x1)^2+(y2-y1)^2+(z2-z1)^2 with the oracle's rounding order (approxmatch.cu:54)
                    const float d0 = sq3(x.x - ox[r], y.x - oy[r], z.x - oz[r]);
                    const float d1 = sq3(x.y - ox[r], y.y - oy[r], z.y - oz[r]);
                    const float d2 = sq3(x.z - ox[r], y.z - oy[r], z.z - oz[r]);
                    const float d3 = sq3(x.w - ox[r], y.w - oy[r], z.w - oz[r]);
                    if (R == 1) {
                        // one owner per lane: a lone wave per SIMD would stall on a single dependent fma chain, so
                        // even and odd candidates go to two accumulators (added once, after the loop)
                        s0[r] = __builtin_fmaf(fast_exp2(c0 * d0), wa.x, s0[r]);
                        t0x = __builtin_fmaf(fast_exp2(c0 * d1), wa.y, t0x);
                        s0[r] = __builtin_fmaf(fast_exp2(c0 * d2), wa.z, s0[r]);
                        t0x = __builtin_fmaf(fast_exp2(c0 * d3), wa.w, t0x);
                        if (NW == 2) {
                            s1[r] = __builtin_fmaf(fast_exp2(c1 * d0), wb.x, s1[r]);
                            t1x = __builtin_fmaf(fast_exp2(c1 * d1), wb.y, t1x);
                            s1[r] = __builtin_fmaf(fast_exp2(c1 * d2), wb.z, s1[r]);
                            t1x = __builtin_fmaf(fast_exp2(c1 * d3), wb.w, t1x);
                        }
                    } else {
                        s0[r] = __builtin_fmaf(fast_exp2(c0 * d0), wa.x, s0[r]);
                        s0[r] = __builtin_fmaf(fast_exp2(c0 * d1), wa.y, s0[r]);
                        s0[r] = __builtin_fmaf(fast_exp2(c0 * d2), wa.z, s0[r]);
                        s0[r] = __builtin_fmaf(fast_exp2(c0 * d3), wa.w, s0[r]);
                        if (NW == 2) {
                            s1[r] = __builtin_fmaf(fast_exp2(c1 * d0), wb.x, s1[r]);
                            s1[r] = __builtin_fmaf(fast_exp2(c1 * d1), wb.y, s1[r]);
                            s1[r] = __builtin_fmaf(fast_exp2(c1 * d2), wb.z, s1[r]);
                            s1[r] = __builtin_fmaf(fast_exp2(c1 * d3), wb.w, s1[r]);
                        }
                    }
                }
            }
        }
    }
    PCC_ST(4);
    if (R == 1) {
        s0[0] += t0x;
        s1[0] += t1x;
    }
    if (G > 1) {  // the G lanes of an owner: a + b is the same float on both sides, so all of them end with the same sum
        s0[0] += __shfl_xor(s0[0], TQ, 64);
        if (G == 4) s0[0] += __shfl_xor(s0[0], 2 * TQ, 64);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        red[(0 * S + w) * PQ + r * 64 + lane] = s0[r];
        if (NW == 2) red[(1 * S + w) * PQ + r * 64 + lane] = s1[r];
    }
    __syncthreads();
    PCC_ST(5);
    if (own_e < 0) return;
    {
        const int e = tid;
        const int o = own_e;
        float t0 = red[e], t1 = 0.f;
        if (NW == 2) t1 = red[(1 * S) * PQ + e];
#pragma unroll
        for (int s = 1; s < S; s++) {
            t0 += red[(0 * S + s) * PQ + e];
            if (NW == 2) t1 += red[(1 * S + s) * PQ + e];
        }
        if (MODE == PH_A) {
            // ratioL[k] = remainL[k] / (1e-9 + sum)            approxmatch.cu:37,61 (remainL == multiL)
            a.ratio_out[(size_t)smp * a.ratio_stride + o] = a.multiL / (1e-9f + t0);
        } else if (MODE == PH_B) {
            // approxmatch.cu:106-109
            const float rR = a.first ? a.multiR : pre_rem;
            const float sumr = t0 * rR;
            const float consumption = __builtin_fminf(rR / (sumr + 1e-9f), 1.0f);
            const float ratio_new = consumption * rR, remain_new = __builtin_fmaxf(0.0f, rR - sumr);
            a.ratio_out[(size_t)smp * a.ratio_stride + o] = ratio_new;
            a.remain_out[(size_t)smp * a.remain_stride + o] = remain_new;
            if (a.live_out) {  // owners still live after this level = the owner count of the next pass B
                const unsigned long long alive = __ballot(remain_new != 0.f);
                if (lane == 0) atomicAdd(&a.live_out[(size_t)smp * kLiveRow], (int)__popcll(alive));
            }
            if (COWN && a.mask_out && remain_new == 0.f)  // exhausted here: not an owner from the next level on
                atomicAnd(&a.mask_out[(size_t)smp * kLevels * a.mask_words + (o >> 5)], ~(1u << (o & 31)));
            if (COWN && a.clist) {
                // this owner is the (tile * TQ + e)-th live one of its sample == its place in the next pass's candidate list
                float cx = ox[0], cy = oy[0], cz = oz[0];  // thread e = w * 64 + lane holds owner e in slot r = w
#pragma unroll
                for (int r = 1; r < R; r++) {
                    cx = w == r ? ox[r] : cx;
                    cy = w == r ? oy[r] : cy;
                    cz = w == r ? oz[r] : cz;
                }
                float *cl = a.clist + (size_t)smp * 5 * a.cl_n4 + tile * TQ + e;
                cl[0] = cx;
                cl[(size_t)a.cl_n4] = cy;
                cl[(size_t)2 * a.cl_n4] = cz;
                cl[(size_t)3 * a.cl_n4] = ratio_new;
                cl[(size_t)4 * a.cl_n4] = remain_new;
            }
        } else {
            // pass C: suml = sum_l e*ratioL[k]*ratioR[l] ; remainL = max(0, remainL - suml)   :154-162
            float *rem = a.remain + (size_t)smp * a.remain_stride + o;
            const float rl = pre_ratio;
            const float rL = a.first ? a.multiL : pre_rem;
            const float left = __builtin_fmaxf(0.0f, rL - rl * t0);
            *rem = left;
            // pass A of the next level: ratioL' = remainL / (1e-9 + sum_l e'*remainR[l])       :37,61
            if (MODE == PH_CA) a.ratio_out[(size_t)smp * a.ratio_stride + o] = left / (1e-9f + t1);
        }
    }
    PCC_ST(6);
#undef PCC_ST
}


template <int MODE, int R, int S, int CH, int VAR, int G = 1>
__global__ __launch_bounds__(64 * S) void am_phase_kernel(PhaseArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[PhaseLds<(MODE == PH_CA ? 2 : 1), R, S, CH>::floats];
    // V_COWN packs the live owners into the low tiles: dispatch those first (tile-major order), so the workgroups
    // that have nothing to do and exit after the scan are not in front of the ones that carry the launch
    constexpr bool COWN = VAR == V_COWN;
    // XCD affinity (cdna_hip_programming.md T1): blocks with equal blockIdx % 8 share an XCD and its L2.  All workgroups
    // of a sample stage the same candidate cloud at the same moment, so a sample's workgroups are given block ids of one
    // residue class: its cloud crosses the fabric once per launch instead of once per XCD (a pure speed choice).
    int smp, tile;
    const int bid = (int)blockIdx.x, nwg = (int)gridDim.x;
    if (COWN) {
        if (a.batch % 8 == 0) {  // samples congruent to the XCD label, tile-major inside the class
            const int per = a.batch / 8, i = bid / 8;
            smp = (bid % 8) + 8 * (i % per);
            tile = i / per;
        } else {
            smp = bid % a.batch;
            tile = bid / a.batch;
        }
        // the previous pass B counted the owners that are still live: workgroups beyond them leave at once (tile 0
        // stays: it carries the zeros of the exhausted owners into the output buffer)
        if (a.live_in && tile > 0 && tile * (64 * R / G) >= a.live_in[(size_t)smp * kLiveRow]) return;
    } else {
        const int lid = pcc::xcd_contiguous(bid, nwg);  // the blocks of one residue class get a contiguous run of logical ids
        smp = lid / a.tiles;
        tile = lid - smp * a.tiles;
    }
    am_phase_body<MODE, R, S, CH, VAR, G>(a, smp, tile, smem);
}

// ---------------------------------------------------------------------------------------------------
// Fine-grained culling for the passes of the fine levels (0-2, where the kernel radius is small against the cloud).
// am_phase_kernel's V_CULL tests (64-owner group, 16-candidate block) pairs of boxes; on Hilbert-sorted clouds a run of
// 64 points is several kernel radii wide, and 75-80 % of the pairs it keeps are still exact zeros.  Here a workgroup
// still owns 64 consecutive sorted owners and stages the candidate cloud once, but the owners are tested and walked in
// four groups of 16: a wave holds ONE group, each owner on 4 lanes that take one float4 (4 candidates) of every
// surviving 16-candidate block, and two waves share the block list of a group.  Measured on the bench clouds the
// (16 x 16) boxes keep 10 / 14 / 23 % of the pairs at levels 0 / 1 / 2 instead of 22 / 25 / 37 %.  Same sums as everywhere
// else in this file: only exact zeros are dropped, partial sums meet in a fixed order (two accumulators per lane,
// shuffles over the 4 lanes of an owner, the two waves of a group in LDS).
// ---------------------------------------------------------------------------------------------------
constexpr int kFineS = 8;                // waves per workgroup
constexpr int kFineOG = 16;              // owners per culling group (== kBox: the sort's 16-point boxes serve both sides)
constexpr int kFineGroups = 64 / kFineOG;
constexpr int kFineQ = 4;                // owners per lane

// Sum over the 16 lanes of a DPP row, in every lane of the row: pairs, quads, half rows, rows -- the tree of the xor
// butterfly (a + b and b + a are the same float), with cross-lane VALU operands instead of four trips through the LDS
// crossbar (ds_bpermute) per value.
__device__ __forceinline__ float row_sum16(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

template <int MODE, int CH>
// (<= 64 VGPRs for passes A and B: with their 34 KB of LDS four workgroups -- the 2 x 512 of the two half-batch lanes --
// then fit a CU together; at 70 VGPRs three fitted, a quarter of every launch's workgroups waited for a slot, and these
// launches are bound by a workgroup's own critical path: 445 -> 439 us per match_cost call, A/B across library builds.
// Pass C/A keeps its 81 registers and 43 KB -- three per CU: its second weight row read from global memory instead of
// LDS, to fit four, measured 454 us.)
__global__ __launch_bounds__(64 * kFineS, MODE == PH_CA ? 4 : 8) void am_fine_kernel(PhaseArgs a) {
    constexpr int T = 64 * kFineS;
    constexpr int NW = (MODE == PH_CA) ? 2 : 1;
    constexpr bool W0_CONST = (MODE == PH_A);
    constexpr int NBLK = CH / kBox;      // candidate blocks per staged chunk
    static_assert(kFineGroups * NBLK == T, "one (owner group, candidate block) box test per thread");
    static_assert(kFineOG == kBox && kFineOG * kFineQ == 64, "a wave = 16 candidates x 4 owner quads");
    __shared__ __attribute__((aligned(16))) float lds_c[(3 + NW) * CH];  // x | y | z | w0 | (w1)
    __shared__ float red[NW][2][64];
    __shared__ unsigned char items[kFineGroups][NBLK];
    __shared__ unsigned char need[NBLK];
    __shared__ int wave_cnt[kFineS];

    const int lid = pcc::xcd_contiguous((int)blockIdx.x, (int)gridDim.x);  // a sample's workgroups share an XCD (see am_phase_kernel)
    const int smp = lid / a.tiles;
    const int tile = lid - smp * a.tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = w & (kFineGroups - 1);      // owner group of this wave
    const int cs = w / kFineGroups;            // its share of the group's block list (0 / 1)
    const int cl = lane & (kBox - 1);          // the candidate of a block this lane holds
    const int quad = lane / kBox;              // its four owners: group-local 4 quad .. 4 quad + 3
    const float *O = a.own_soa + (size_t)smp * 3 * a.own_n4;
    const float *C = a.cand_soa + (size_t)smp * 3 * a.cand_n4;
    const float *W0 = W0_CONST ? nullptr : a.w0 + (size_t)smp * a.w0_stride;
    const float *W1 = (NW == 2) ? a.w1 + (size_t)smp * a.w1_stride : nullptr;

    // debug stamps as in am_phase_body (PCC_AM_DEBUG=2): [1] owners + boxes loaded, [2] lists built, [3] candidates staged,
    // [4] walk done, [5] reduced, [6] end
    unsigned long long tst0 = 0;
    int *stamp = nullptr;
    if (a.stamp && threadIdx.x == 0) {
        const unsigned bx = blockIdx.x, gx = gridDim.x;
        const int which = bx == 0 ? 0 : bx == gx / 2 ? 1 : bx == gx - 1 ? 2 : -1;
        if (which >= 0) {
            stamp = a.stamp + 8 * which;
            tst0 = __builtin_amdgcn_s_memrealtime();
            stamp[0] = (int)(tst0 & 0x7fffffff);
        }
    }
#define PCC_STF(k) do { if (stamp) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[(k)] = (int)(__builtin_amdgcn_s_memrealtime() - tst0); } } while (0)
    float ox[kFineQ], oy[kFineQ], oz[kFineQ], s0[kFineQ], s1[kFineQ];
#pragma unroll
    for (int j = 0; j < kFineQ; j++) {
        int o = tile * 64 + og * kFineOG + quad * kFineQ + j;
        o = o < a.n_own ? o : a.n_own - 1;
        ox[j] = O[o];
        oy[j] = O[a.own_n4 + o];
        oz[j] = O[2 * a.own_n4 + o];
        s0[j] = 0.f;
        s1[j] = 0.f;
    }
    // epilogue operands (thread e < 64 finishes owner tile * 64 + e): fetched now, behind the staging traffic
    const int own_e = (tid < 64 && tile * 64 + tid < a.n_own) ? tile * 64 + tid : -1;
    // (for every thread, from a clamped index: issued with the owner coordinates above, one round trip for all)
    float pre_rem = 0.f, pre_ratio = 0.f;
    {
        const int oe = max(own_e, 0);
        if (MODE != PH_A && !a.first) pre_rem = a.remain[(size_t)smp * a.remain_stride + oe];
        if (MODE == PH_CA || MODE == PH_C) pre_ratio = a.ratio_in[(size_t)smp * a.ratio_stride + oe];
    }
    // box of the owner group this THREAD tests (thread = (group tg, candidate block tb))
    const int tg = tid / NBLK, tb = tid - tg * NBLK;
    float4 glo = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
    float4 ghi = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.f);
    {
        const int g16 = tile * kFineGroups + tg;
        if (g16 < a.own_nb) {
            const float4 *ob = reinterpret_cast<const float4 *>(a.own_box16 + ((size_t)smp * a.own_nb + g16) * 8);
            glo = ob[0];
            ghi = ob[1];
        }
    }
    const float c0 = a.c0, c1 = a.c1, cut2 = a.cut2;
    PCC_STF(1);

    for (int q0 = 0; q0 < a.n_cand; q0 += CH) {
        const int cnt = min(CH, a.n_cand - q0);
        const int ngroups = (cnt + 3) / 4;
        const int nblk = (ngroups + 3) / 4;
        if (q0) __syncthreads();
        if (tid < NBLK) need[tid] = 0;
        // one box test per thread; the surviving blocks of a group are compacted, in order, into its list (the two
        // waves that hold the flags of a group: 2 tg and 2 tg + 1)
        bool keep = false, outer_only = false;
        if (tb < nblk) {
            const float4 *cb = reinterpret_cast<const float4 *>(a.cand_box + ((size_t)smp * a.cand_nb + q0 / kBox + tb) * 8);
            const float4 lo = cb[0], hi = cb[1];
            const float dx = fmaxf(fmaxf(glo.x - hi.x, lo.x - ghi.x), 0.f);
            const float dy = fmaxf(fmaxf(glo.y - hi.y, lo.y - ghi.y), 0.f);
            const float dz = fmaxf(fmaxf(glo.z - hi.z, lo.z - ghi.z), 0.f);
            const float bd2 = dx * dx + dy * dy + dz * dz;
            keep = !(bd2 > cut2);  // farther: every exponential of the pair of boxes is exactly 0
            // pass C/A walks the radius of the COARSER level; between the two radii the finer level's exponential is
            // exactly 0 for the whole pair of boxes: such a block is marked and costs one exponential, not two
            outer_only = NW == 2 && bd2 > a.cut2_fine;
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[w] = __popcll(bal);
        __syncthreads();
        {
            const int before = (w & 1) ? wave_cnt[w - 1] : 0;
            if (keep) {
                items[tg][before + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned char)(tb | (outer_only ? 128 : 0));  // (tb < 128)
                need[tb] = 1;  // (same value from every writer)
            }
        }
        __syncthreads();
        PCC_STF(2);
        {   // stage the blocks some group of this workgroup needs (on the fine levels a fraction of the cloud): the sorted
            // SoA rows and weight rows are padded to a multiple of 4 (zeros), so these are straight float4 copies
            float4 *dst4 = reinterpret_cast<float4 *>(lds_c);
            const float4 *sx = reinterpret_cast<const float4 *>(C + q0);
            const float4 *sy = reinterpret_cast<const float4 *>(C + (size_t)a.cand_n4 + q0);
            const float4 *sz = reinterpret_cast<const float4 *>(C + (size_t)2 * a.cand_n4 + q0);
            // (one float4 group per thread, no loop: as a loop over i += T every row pointer became a 64-bit induction
            // variable in a VGPR pair, hoisted out of the chunk loop -- most of what pass A / B spilled)
            static_assert(CH / 4 <= T, "one float4 group per thread and chunk");
            const int i = tid;
            if (i < nblk * 4 && need[i >> 2]) {
                float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vy = vx, vz = vx, v0 = vx, v1 = vx;
                if (i < ngroups) {
                    vx = sx[i]; vy = sy[i]; vz = sz[i];
                    v0 = W0_CONST ? make_float4(a.w0c, a.w0c, a.w0c, a.w0c) : *reinterpret_cast<const float4 *>(W0 + q0 + 4 * i);
                    if (NW == 2) v1 = *reinterpret_cast<const float4 *>(W1 + q0 + 4 * i);
                    if (W0_CONST && i * 4 + 3 >= cnt) {  // padded candidates must weigh 0
                        v0.x = i * 4 + 0 < cnt ? v0.x : 0.f;
                        v0.y = i * 4 + 1 < cnt ? v0.y : 0.f;
                        v0.z = i * 4 + 2 < cnt ? v0.z : 0.f;
                        v0.w = 0.f;
                    }
                }
                dst4[i] = vx;
                dst4[CH / 4 + i] = vy;
                dst4[2 * (CH / 4) + i] = vz;
                dst4[3 * (CH / 4) + i] = v0;
                if (NW == 2) dst4[4 * (CH / 4) + i] = v1;
            }
        }
        __syncthreads();
        PCC_STF(3);
        // a wave walks its half of the group's block list: a lane holds ONE candidate of the block (five scalar LDS
        // reads, 16 distinct addresses per wave) against its four owners in registers -- four independent fma chains
        const int nitems = wave_cnt[2 * og] + wave_cnt[2 * og + 1];
        for (int it = cs; it < nitems; it += 2) {
            const int item = __builtin_amdgcn_readfirstlane((int)items[og][it]);
            const int ci = (item & 127) * kBox + cl;
            const float x = lds_c[ci], y = lds_c[CH + ci], z = lds_c[2 * CH + ci];
            if (NW == 2 && (item & 128)) {  // (wave-uniform) beyond the finer level's radius: its terms are exact zeros
                const float wb = lds_c[4 * CH + ci];
#pragma unroll
                for (int j = 0; j < kFineQ; j++) s1[j] = __builtin_fmaf(fast_exp2(c1 * sq3(x - ox[j], y - oy[j], z - oz[j])), wb, s1[j]);
                continue;
            }
            const float wa = lds_c[3 * CH + ci];
            float wb = 0.f;
            if (NW == 2) wb = lds_c[4 * CH + ci];
#pragma unroll
            for (int j = 0; j < kFineQ; j++) {
                const float d = sq3(x - ox[j], y - oy[j], z - oz[j]);
                s0[j] = __builtin_fmaf(fast_exp2(c0 * d), wa, s0[j]);
                if (NW == 2) s1[j] = __builtin_fmaf(fast_exp2(c1 * d), wb, s1[j]);
            }
        }
    }
    PCC_STF(4);
    // the 16 candidate lanes of an owner meet through four butterfly steps (a + b is the same float on both sides, so
    // every lane ends with the same sum); lane cl == 0 of each quad hands the four sums to the epilogue
#pragma unroll
    for (int j = 0; j < kFineQ; j++) {
        s0[j] = row_sum16(s0[j]);
        if (NW == 2) s1[j] = row_sum16(s1[j]);
    }
    if (cl == 0) {
#pragma unroll
        for (int j = 0; j < kFineQ; j++) {
            red[0][cs][og * kFineOG + quad * kFineQ + j] = s0[j];
            if (NW == 2) red[NW - 1][cs][og * kFineOG + quad * kFineQ + j] = s1[j];
        }
    }
    __syncthreads();
    PCC_STF(5);
    if (own_e < 0) return;
    const float sum0 = red[0][0][tid] + red[0][1][tid];
    const float sum1 = NW == 2 ? red[NW - 1][0][tid] + red[NW - 1][1][tid] : 0.f;
    if (MODE == PH_A) {
        // ratioL[k] = remainL[k] / (1e-9 + sum)            approxmatch.cu:37,61 (remainL == multiL)
        a.ratio_out[(size_t)smp * a.ratio_stride + own_e] = a.multiL / (1e-9f + sum0);
    } else if (MODE == PH_B) {
        // approxmatch.cu:106-109
        const float rR = a.first ? a.multiR : pre_rem;
        const float sumr = sum0 * rR;
        const float consumption = __builtin_fminf(rR / (sumr + 1e-9f), 1.0f);
        const float ratio_new = consumption * rR, remain_new = __builtin_fmaxf(0.0f, rR - sumr);
        a.ratio_out[(size_t)smp * a.ratio_stride + own_e] = ratio_new;
        a.remain_out[(size_t)smp * a.remain_stride + own_e] = remain_new;
        if (a.live_out) {  // owners still live after this level = the owner count of the next pass B
            const unsigned long long alive = __ballot(remain_new != 0.f);
            if (lane == 0) atomicAdd(&a.live_out[(size_t)smp * kLiveRow], (int)__popcll(alive));
            if (a.mask_out && lane == 0) {  // ... and their bits: this tile's 64 owners are two whole words of the next row
                unsigned *mo = a.mask_out + (size_t)smp * kLevels * a.mask_words;
                if (2 * tile < a.mask_words) mo[2 * tile] = (unsigned)alive;
                if (2 * tile + 1 < a.mask_words) mo[2 * tile + 1] = (unsigned)(alive >> 32);
            }
        }
    } else {
        // pass C: suml = sum_l e*ratioL[k]*ratioR[l] ; remainL = max(0, remainL - suml)   :154-162
        const float rL = a.first ? a.multiL : pre_rem;
        const float left = __builtin_fmaxf(0.0f, rL - pre_ratio * sum0);
        a.remain[(size_t)smp * a.remain_stride + own_e] = left;
        // pass A of the next level: ratioL' = remainL / (1e-9 + sum_l e'*remainR[l])       :37,61
        if (MODE == PH_CA) a.ratio_out[(size_t)smp * a.ratio_stride + own_e] = left / (1e-9f + sum1);
    }
    PCC_STF(6);
#undef PCC_STF
}

// ---------------------------------------------------------------------------------------------------
// The seven passes of the fine levels (A0 B0 CA0 B1 CA1 B2 CA2) as ONE resident launch.
// On am_fine_kernel these passes are bound by their fixed costs -- a dependent launch (~5 us of gap plus a cold L2:
// the producer ran on other XCDs) and ~5 us of workgroup prologue for 1-4 us of arithmetic -- 97 us of each lane's
// chain at B=32, N=2048.  Here a workgroup keeps tile t of BOTH clouds (64 + 64 owners) for all seven passes:
//   * the two sorted clouds are staged in LDS once (coordinates never change), the (16-owner group, 16-candidate block)
//     box distances are computed once and kept in a register per direction -- the balls of the levels nest, so a pass
//     only compares that distance with its own radius;
//   * an owner's running state (remainL, ratioL_i, remainR) stays in the register of the thread that finishes it;
//   * per pass only the weights of the blocks some group needs are re-read (agent-scope loads: written by other
//     workgroups during this launch) and one or two values per owner are written (agent-scope stores);
//   * a pass boundary is a per-SAMPLE barrier (one counter per sample in the live-counter row; tiles of a sample get
//     consecutive block ids of one XCD): no launch, no cache-wide fence.
// The pair walk, the reductions and the epilogue formulas are am_fine_kernel's, in the same order: the level rows carry
// the same bits as with one launch per pass (tests/test_gpu_structural.py::test_resident_fine_levels_equal_one_launch_per_pass).
// Residency: a sample's barrier needs its `tiles` workgroups resident together, and the hardware does NOT dispatch
// workgroups strictly in block-id order (measured: a launch of more workgroups than the chip holds leaves samples
// half-resident for milliseconds -- B=32 as two 512-workgroup lanes ran 4 ms per call, with barrier time-outs).  So, like
// the auction's cluster kernel, the host uses this form only when the WHOLE launch fits the device with room to spare
// (batch x tiles <= compute units while two workgroups fit a CU: b <= 8 at N = 2048, the reference's default
// per-device batch, default_train.yaml:6) and never lets two of these launches run at the same time (an event orders
// them across streams).  Spins are bounded all the same: a barrier that times out raises the sample's error slot (the
// finish kernel then reports NaN for the sample) and a sticky host word (the next call on the device fails) instead
// of hanging.  Measured (N=2048, b=8): 68 us against 91 us for the seven launches -- the passes are bound by the
// latency of one workgroup's own work (list, weights, walk, reduction: ~10 us), not by the launch boundary.
// ---------------------------------------------------------------------------------------------------
constexpr int kFpPasses = 7;   // A0 B0 CA0 B1 CA1 B2 CA2
constexpr int kFpCH = 2048;    // largest cloud the resident form takes
constexpr int kBarSlot = 12;   // live-counter row: the sample's barrier counter (cleared by the sort kernel)
constexpr int kErrSlot = 15;   // live-counter row: the sample's resident passes did not complete

struct FinePersistArgs {
    int n, m, n4, m4, nb1, nb2, tiles;
    const float *soa1, *soa2, *box1, *box2;
    float *rem, *lv;               // sorted space (see Sched)
    float multiL, multiR;
    float c[4], cut2[4];           // levels 0..3
    int *live_cnt;                 // [b][kLiveRow]
    unsigned *live_mask;           // [b][kLevels][mask_words] (row 3 is written here: PhaseArgs::mask_out)
    int mask_words;
    unsigned *host_err;            // sticky word in mapped host memory (or null)
};

__device__ __forceinline__ float fp_ld(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fp_st(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64 * kFineS) void am_fine_persist_kernel(FinePersistArgs a) {
    constexpr int T = 64 * kFineS, CH = kFpCH, NBLK = CH / kBox;
    static_assert(kFineGroups * NBLK == T, "one (owner group, candidate block) box test per thread and direction");
    __shared__ __attribute__((aligned(16))) float lds_p[2][3 * CH];  // sorted coordinates x | y | z of set1, set2
    __shared__ __attribute__((aligned(16))) float lds_w[2][CH];      // w0 | w1 of the current pass
    __shared__ float red[2][2][64];
    __shared__ unsigned char items[kFineGroups][NBLK];
    __shared__ unsigned char need[NBLK];
    __shared__ int wave_cnt[kFineS];
    __shared__ int bar_failed;

    const int lid = pcc::xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int smp = lid / a.tiles;
    const int tile = lid - smp * a.tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = w & (kFineGroups - 1), cs = w / kFineGroups;
    const int cl = lane & (kBox - 1), quad = lane / kBox;
    const long long nm4 = (long long)a.n4 + a.m4, rs = (long long)a.n4 + 2LL * a.m4;
    const float *S1 = a.soa1 + (size_t)smp * 3 * a.n4;
    const float *S2 = a.soa2 + (size_t)smp * 3 * a.m4;
    float *LV = a.lv + (size_t)smp * kLevels * nm4;
    float *REM = a.rem + (size_t)smp * rs;
    int *row = a.live_cnt + (size_t)smp * kLiveRow;
    unsigned *bar = reinterpret_cast<unsigned *>(row + kBarSlot);
    unsigned *err = reinterpret_cast<unsigned *>(row + kErrSlot);

    // ---- once: both clouds into LDS (rows are padded to a multiple of 4 with zeros; beyond that zeros: a padded
    // candidate sits at the origin with weight 0) ----
    for (int s = 0; s < 2; s++) {
        const float *S = s ? S2 : S1;
        const int p4 = s ? a.m4 : a.n4, ngroups = p4 / 4, nblk = (ngroups + 3) / 4;
        float4 *dst4 = reinterpret_cast<float4 *>(lds_p[s]);
        const float4 *sx = reinterpret_cast<const float4 *>(S);
        const float4 *sy = reinterpret_cast<const float4 *>(S + p4);
        const float4 *sz = reinterpret_cast<const float4 *>(S + 2 * (size_t)p4);
        for (int i = tid; i < nblk * 4; i += T) {
            float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vy = vx, vz = vx;
            if (i < ngroups) { vx = sx[i]; vy = sy[i]; vz = sz[i]; }
            dst4[i] = vx;
            dst4[CH / 4 + i] = vy;
            dst4[2 * (CH / 4) + i] = vz;
        }
    }
    // ---- once: this lane's owners of both roles, and the box distance of its (group, block) test per direction ----
    float ox1[kFineQ], oy1[kFineQ], oz1[kFineQ], ox2[kFineQ], oy2[kFineQ], oz2[kFineQ];
#pragma unroll
    for (int j = 0; j < kFineQ; j++) {
        const int o = tile * 64 + og * kFineOG + quad * kFineQ + j;
        const int o1 = min(o, a.n - 1), o2 = min(o, a.m - 1);
        ox1[j] = S1[o1]; oy1[j] = S1[a.n4 + o1]; oz1[j] = S1[2 * a.n4 + o1];
        ox2[j] = S2[o2]; oy2[j] = S2[a.m4 + o2]; oz2[j] = S2[2 * a.m4 + o2];
    }
    const int tg = tid / NBLK, tb = tid - tg * NBLK;
    auto box_d2 = [&](const float *own_box, int own_nb, const float *cand_box, int cand_nb) -> float {
        float4 glo = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
        float4 ghi = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.f);
        const int g16 = tile * kFineGroups + tg;
        if (g16 < own_nb) {
            const float4 *ob = reinterpret_cast<const float4 *>(own_box + ((size_t)smp * own_nb + g16) * 8);
            glo = ob[0];
            ghi = ob[1];
        }
        if (tb >= cand_nb) return __builtin_inff();
        const float4 *cb = reinterpret_cast<const float4 *>(cand_box + ((size_t)smp * cand_nb + tb) * 8);
        const float4 lo = cb[0], hi = cb[1];
        const float dx = fmaxf(fmaxf(glo.x - hi.x, lo.x - ghi.x), 0.f);
        const float dy = fmaxf(fmaxf(glo.y - hi.y, lo.y - ghi.y), 0.f);
        const float dz = fmaxf(fmaxf(glo.z - hi.z, lo.z - ghi.z), 0.f);
        return dx * dx + dy * dy + dz * dz;
    };
    const float d2_role0 = box_d2(a.box1, a.nb1, a.box2, a.nb2);  // set1 owners against set2 candidate blocks
    const float d2_role1 = box_d2(a.box2, a.nb2, a.box1, a.nb1);  // set2 owners against set1 candidate blocks
    // running state of the owner thread e < 64 finishes in each role
    const int own1 = (tid < 64 && tile * 64 + tid < a.n) ? tile * 64 + tid : -1;
    const int own2 = (tid < 64 && tile * 64 + tid < a.m) ? tile * 64 + tid : -1;
    float remL = 0.f, ratL = 0.f, remR = 0.f;
    if (tid == 0) bar_failed = 0;
    unsigned arrivals = 0;
    bool ok = true;

    // one pass: MODE decides the role (A / CA: set1 owners, B: set2 owners); i = level
    auto pass = [&](auto mode_tag, int i) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int NW = MODE == PH_CA ? 2 : 1;
        constexpr int ROLE = MODE == PH_B ? 1 : 0;
        const int n_cand = ROLE ? a.n : a.m, cand_n4 = ROLE ? a.n4 : a.m4;
        const int nblk = (cand_n4 / 4 + 3) / 4;
        const float c0 = a.c[i], c1 = a.c[MODE == PH_CA ? i + 1 : i];
        const float cut2 = a.cut2[MODE == PH_CA ? i + 1 : i];  // the coarser of the two levels decides what is 0
        const float *P = lds_p[ROLE ? 0 : 1];
        if (tid < NBLK) need[tid] = 0;
        const float bd2 = ROLE ? d2_role1 : d2_role0;
        const bool keep = tb < nblk && !(bd2 > cut2);
        const bool outer_only = NW == 2 && bd2 > a.cut2[i];  // beyond the finer level's radius (see am_fine_kernel)
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[w] = __popcll(bal);
        __syncthreads();
        {
            const int before = (w & 1) ? wave_cnt[w - 1] : 0;
            if (keep) {
                items[tg][before + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned char)(tb | (outer_only ? 128 : 0));
                need[tb] = 1;
            }
        }
        __syncthreads();
        // weights of the blocks some group needs
        {
            const float *W0 = MODE == PH_A ? nullptr : MODE == PH_B ? LV + (size_t)i * nm4 : LV + (size_t)i * nm4 + a.n4;
            const float *W1 = MODE == PH_CA ? REM + a.n4 + (size_t)((i + 1) & 1) * a.m4 : nullptr;
            for (int idx = tid; idx < nblk * kBox; idx += T) {
                if (!need[idx >> 4]) continue;
                float v0, v1 = 0.f;
                if (MODE == PH_A) v0 = idx < n_cand ? a.multiR : 0.f;
                else v0 = idx < cand_n4 ? fp_ld(W0 + idx) : 0.f;
                if (NW == 2) v1 = idx < cand_n4 ? fp_ld(W1 + idx) : 0.f;
                lds_w[0][idx] = v0;
                if (NW == 2) lds_w[1][idx] = v1;
            }
        }
        __syncthreads();
        float s0[kFineQ], s1[kFineQ];
#pragma unroll
        for (int j = 0; j < kFineQ; j++) s0[j] = s1[j] = 0.f;
        const int nitems = wave_cnt[2 * og] + wave_cnt[2 * og + 1];
        for (int it = cs; it < nitems; it += 2) {
            const int item = __builtin_amdgcn_readfirstlane((int)items[og][it]);
            const int ci = (item & 127) * kBox + cl;
            const float x = P[ci], y = P[CH + ci], z = P[2 * CH + ci];
            if (NW == 2 && (item & 128)) {  // (wave-uniform; pass C/A only, whose owners are set1)
                const float wb = lds_w[1][ci];
#pragma unroll
                for (int j = 0; j < kFineQ; j++) s1[j] = __builtin_fmaf(fast_exp2(c1 * sq3(x - ox1[j], y - oy1[j], z - oz1[j])), wb, s1[j]);
                continue;
            }
            const float wa = lds_w[0][ci];
            float wb = 0.f;
            if (NW == 2) wb = lds_w[1][ci];
#pragma unroll
            for (int j = 0; j < kFineQ; j++) {
                const float d = ROLE ? sq3(x - ox2[j], y - oy2[j], z - oz2[j]) : sq3(x - ox1[j], y - oy1[j], z - oz1[j]);
                s0[j] = __builtin_fmaf(fast_exp2(c0 * d), wa, s0[j]);
                if (NW == 2) s1[j] = __builtin_fmaf(fast_exp2(c1 * d), wb, s1[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < kFineQ; j++) {
            s0[j] = row_sum16(s0[j]);
            if (NW == 2) s1[j] = row_sum16(s1[j]);
        }
        if (cl == 0) {
#pragma unroll
            for (int j = 0; j < kFineQ; j++) {
                red[0][cs][og * kFineOG + quad * kFineQ + j] = s0[j];
                if (NW == 2) red[1][cs][og * kFineOG + quad * kFineQ + j] = s1[j];
            }
        }
        __syncthreads();
        const int own_e = ROLE ? own2 : own1;
        if (own_e >= 0) {
            const float sum0 = red[0][0][tid] + red[0][1][tid];
            const float sum1 = NW == 2 ? red[1][0][tid] + red[1][1][tid] : 0.f;
            if (MODE == PH_A) {
                ratL = a.multiL / (1e-9f + sum0);                                       // approxmatch.cu:37,61
                fp_st(LV + own_e, ratL);
            } else if (MODE == PH_B) {
                const float rR = i == 0 ? a.multiR : remR;                               // approxmatch.cu:106-109
                const float sumr = sum0 * rR;
                const float consumption = __builtin_fminf(rR / (sumr + 1e-9f), 1.0f);
                const float ratio_new = consumption * rR, remain_new = __builtin_fmaxf(0.0f, rR - sumr);
                fp_st(LV + (size_t)i * nm4 + a.n4 + own_e, ratio_new);
                fp_st(REM + a.n4 + (size_t)((i + 1) & 1) * a.m4 + own_e, remain_new);
                remR = remain_new;
                if (i == 2) {  // owners still live after level 2 = the owner count of pass B of level 3 (V_COWN)
                    const unsigned long long alive = __ballot(remain_new != 0.f);
                    if (lane == 0) {
                        atomicAdd(&row[3], (int)__popcll(alive));
                        unsigned *mo = a.live_mask + ((size_t)smp * kLevels + 3) * a.mask_words;  // ... and their bits
                        if (2 * tile < a.mask_words) mo[2 * tile] = (unsigned)alive;
                        if (2 * tile + 1 < a.mask_words) mo[2 * tile + 1] = (unsigned)(alive >> 32);
                    }
                }
            } else {
                const float rL = i == 0 ? a.multiL : remL;                               // approxmatch.cu:154-162
                const float left = __builtin_fmaxf(0.0f, rL - ratL * sum0);
                fp_st(REM + own_e, left);
                remL = left;
                ratL = left / (1e-9f + sum1);                                            // :37,61 of the next level
                fp_st(LV + (size_t)(i + 1) * nm4 + own_e, ratL);
            }
        }
    };
    // sample barrier: every wave's agent-scope stores have left the CU, then one arrival per workgroup
    auto sample_barrier = [&]() -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        arrivals += (unsigned)a.tiles;
        if (tid == 0) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrivals) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    const unsigned seen = __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned was = __hip_atomic_exchange(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // diagnosis for the host: 1 | barrier << 4 | arrivals seen << 8 | sample << 20 (first reporter wins)
                    if (a.host_err && !was) {
                        unsigned expected = 0;
                        __hip_atomic_compare_exchange_strong(a.host_err, &expected, 1u | ((arrivals / (unsigned)a.tiles) << 4) | ((seen & 0xfffu) << 8) | ((unsigned)smp << 20),
                                                             __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    bar_failed = 1;
                    break;
                }
            }
        }
        __syncthreads();
        return bar_failed == 0;
    };
    __syncthreads();  // clouds staged, bar_failed initialised
    pass(std::integral_constant<int, PH_A>{}, 0);
    for (int i = 0; i < 3 && ok; i++) {
        ok = sample_barrier();
        if (!ok) break;
        pass(std::integral_constant<int, PH_B>{}, i);
        ok = sample_barrier();
        if (!ok) break;
        pass(std::integral_constant<int, PH_CA>{}, i);
    }
}

// ---------------------------------------------------------------------------------------------------
// Nearest neighbours on the sorted clouds (the Chamfer half of the reference's ChamferEMD loss, nndistance.cu:2-128, when
// it is computed in the same call as the approximate EMD: pcc_chamfer_emd).  The exhaustive scan of nn_fwd_kernel
// evaluates every pair; here the Hilbert-sorted points, the 16-point boxes and the permutations of THIS call's sort are
// reused, and a group of 16 consecutive sorted queries only visits the candidate blocks that can still hold a nearest
// neighbour:
//   1. the candidate blocks of a window are ordered by the distance between their box and the group's box (a lower
//      bound of every distance between the two boxes) and visited nearest first;
//   2. the group's radius is the largest of its queries' best distances so far; the first block whose box is farther
//      than the radius ends the walk (everything behind it is farther still): exact, nothing that could win -- or tie
//      with a lower index -- is skipped.  On the bench clouds a group visits 11 of 128 blocks on average (43 at most);
//   3. ties go to the lowest ORIGINAL candidate index (the reference's rule; the sorted order is not the original one):
//      the best so far is one 64-bit key (distance bits : original index).  Distances are the oracle's fmaf chain:
//      indices and distances carry the bits of nn_fwd_kernel / the oracle (tests/test_gpu_structural.py).
// Measured: 42 us per half-batch launch at B=32, N=2048 -- on a par with the exhaustive kernel (the 16 x 16 tile steps cost
// ~100 instructions each, 4x the exhaustive kernel's cost per pair, on 9 % of the pairs); what the fused call saves is
// the separate loss-reduction launch (it rides in the finish launch) and the second read of the clouds.
// ---------------------------------------------------------------------------------------------------
struct NNSortedArgs {
    int n_q, n_c, q_n4, c_n4, q_nb, c_nb, groups, batch;
    const float *q_soa;            // [b][3][n4]
    const float4 *c_aos;           // [b][n_c] (x, y, z, original index) per sorted candidate
    const float *q_box, *c_box;    // [b][nb][8]
    const int *q_perm;             // [b][n] sorted position -> original index
    float *out_d;                  // [b][n_q] in the caller's query order
    int *out_i;
};

// minimum over the 16 lanes of a DPP row, in every lane of the row: four cross-lane VALU operands (xor 1, xor 2, then the
// two mirror steps) instead of four trips through the LDS crossbar (ds_bpermute)
__device__ __forceinline__ float row_min16(float v) {
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));  // row_mirror
    return v;
}

// A WAVE owns one group of 16 consecutive sorted queries and works alone (no LDS, no barrier: thousands of independent
// waves hide each other's latency): lane = (candidate slot cl of a 16-candidate block, query quad), four queries in
// registers.  128 candidate blocks per window: every lane tests two of them against the group's box, the survivors are
// two 64-bit ballots that the wave walks bit by bit; a lane loads ITS candidate of the block straight from the sorted
// rows (L2-resident), the next block's loads are issued before the current one is consumed.
constexpr int kNNWaves = 4;  // independent waves per workgroup

__global__ __launch_bounds__(64 * kNNWaves) void nn_sorted_kernel(NNSortedArgs a0, NNSortedArgs a1, int waves0) {
    const int lane = threadIdx.x & 63;
    int gw = (int)blockIdx.x * kNNWaves + (int)(threadIdx.x >> 6);   // global wave = (direction, sample, group)
    const bool second = gw >= waves0;
    const NNSortedArgs &a = second ? a1 : a0;
    if (second) gw -= waves0;
    const int smp = gw / a.groups;
    const int grp = gw - smp * a.groups;
    if (smp >= a.batch) return;  // (whole wave)
    const int cl = lane & (kBox - 1), quad = lane / kBox;
    const float *Q = a.q_soa + (size_t)smp * 3 * a.q_n4;
    const float4 *C = a.c_aos + (size_t)smp * a.n_c;

    // best so far per query as ONE 64-bit key (distance bits : original candidate index): squared distances are
    // non-negative floats, which order like unsigned integers, so a single 64-bit compare is the reference's rule
    // "smaller distance, lowest index on ties"
    float qx[kFineQ], qy[kFineQ], qz[kFineQ];
    unsigned long long bk[kFineQ];
#pragma unroll
    for (int j = 0; j < kFineQ; j++) {
        int q = grp * kBox + quad * kFineQ + j;
        q = q < a.n_q ? q : a.n_q - 1;
        qx[j] = Q[q];
        qy[j] = Q[a.q_n4 + q];
        qz[j] = Q[2 * a.q_n4 + q];
        bk[j] = ((unsigned long long)__float_as_uint(__builtin_inff()) << 32) | 0x7fffffffull;
    }
    const float4 *gb = reinterpret_cast<const float4 *>(a.q_box + ((size_t)smp * a.q_nb + grp) * 8);
    const float4 glo = gb[0], ghi = gb[1];

    struct Cand {
        float x, y, z;
        int o;
    };
    auto load_block = [&](int blk) -> Cand {  // this lane's candidate of block `blk` (+inf / INT_MAX beyond the cloud)
        // unconditional loads of a clamped index (a branch around them would serialise the software pipeline below),
        // then the select
        const int ci = blk * kBox + cl;
        const bool real = ci < a.n_c;
        const unsigned cc = (unsigned)(real ? ci : a.n_c - 1);
        const float4 v = C[cc];  // one 16-byte load per lane and block
        Cand c;
        c.x = real ? v.x : __builtin_inff();
        c.y = v.y;
        c.z = v.z;
        c.o = real ? __float_as_int(v.w) : 0x7fffffff;
        return c;
    };
    auto scan = [&](const Cand &c) {
#pragma unroll
        for (int j = 0; j < kFineQ; j++) {
            const float d = sq3(c.x - qx[j], c.y - qy[j], c.z - qz[j]);
            const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)c.o;
            bk[j] = k < bk[j] ? k : bk[j];
        }
    };
    // the group's radius: every query's best so far over its 16 candidate lanes, the largest over the 16 queries
    auto group_radius = [&]() -> float {
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < kFineQ; j++) {
            const float m = row_min16(__uint_as_float((unsigned)(bk[j] >> 32)));
            r = fmaxf(r, grp * kBox + quad * kFineQ + j < a.n_q ? m : 0.f);
        }
        // the four rows (query quads) meet through scalar reads
        const int ri = __float_as_int(r);
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(ri, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(ri, 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(ri, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(ri, 48));
        return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    };

    float r = __builtin_inff();  // the group's radius: the largest of its queries' best distances so far
    for (int b0 = 0; b0 < a.c_nb; b0 += 128) {  // windows of 128 candidate blocks (2048 candidates)
        // every lane: two blocks of the window, keyed by the lower bound of every distance between the group's box and
        // the block's box.  key = lower bound with its 7 lowest mantissa bits replaced by the block's slot: positive
        // floats order like unsigned integers, and the truncation only makes the bound smaller (conservative)
        unsigned key[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int blk = b0 + lane + 64 * h;
            key[h] = 0xffffffffu;
            if (blk < a.c_nb) {
                const float4 *cb = reinterpret_cast<const float4 *>(a.c_box + ((size_t)smp * a.c_nb + blk) * 8);
                const float4 lo = cb[0], hi = cb[1];
                const float dx = fmaxf(fmaxf(glo.x - hi.x, lo.x - ghi.x), 0.f);
                const float dy = fmaxf(fmaxf(glo.y - hi.y, lo.y - ghi.y), 0.f);
                const float dz = fmaxf(fmaxf(glo.z - hi.z, lo.z - ghi.z), 0.f);
                key[h] = (__float_as_uint(sq3(dx, dy, dz)) & ~127u) | (unsigned)(lane + 64 * h);  // (the candidates' own chain: monotone)
            }
        }
        // ascending bitonic sort of the 128 keys held by the wave (element lane + 64 h): nearest boxes first
#pragma unroll
        for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j == 64) {
                    const unsigned mn = min(key[0], key[1]), mx = max(key[0], key[1]);
                    key[0] = mn;
                    key[1] = mx;
                } else {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int i = lane + 64 * h;
                        const unsigned other = (unsigned)__shfl_xor((int)key[h], j, 64);
                        const bool take_min = ((i & j) == 0) == ((i & k) == 0);
                        key[h] = take_min ? min(key[h], other) : max(key[h], other);
                    }
                }
            }
        }
        // walk the window nearest-first; a block farther than the radius ends it (everything behind is farther still):
        // nothing that could win, or tie with a lower original index, is skipped.  Two blocks are in flight ahead.
        // (branch-free: a branch around the look-ahead loads makes the compiler drain them before every use)
        auto key_at = [&](int p) -> unsigned {
            const int pl = __builtin_amdgcn_readfirstlane(p) & 63;
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)key[0], pl);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)key[1], pl);
            return p < 64 ? lo : hi;
        };
        const int nwin = min(128, a.c_nb - b0);
        // batches of four blocks: their sixteen loads are issued together, each block is consumed as soon as ITS loads
        // have landed (straight-line code: the compiler counts the outstanding loads exactly), the radius is refreshed
        // after every batch
        bool done = false;
        for (int p = 0; p < nwin && !done; p += 4) {
            unsigned kk[4];
            Cand cc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                kk[u] = key_at(min(p + u, nwin - 1));  // (past the end: the last block again, never consumed)
                cc[u] = load_block(b0 + (int)(kk[u] & 127u));
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!done) {
                    if (p + u >= nwin || __uint_as_float(kk[u] & ~127u) > r) done = true;
                    else scan(cc[u]);
                }
            }
            r = fminf(r, group_radius());
        }
    }
    // every query: best over its 16 candidate lanes
#pragma unroll
    for (int j = 0; j < kFineQ; j++) {
#pragma unroll
        for (int off = 1; off < kBox; off <<= 1) {
            const unsigned hi = (unsigned)__shfl_xor((int)(bk[j] >> 32), off, 64);
            const unsigned lo = (unsigned)__shfl_xor((int)(bk[j] & 0xffffffffu), off, 64);
            const unsigned long long ok = ((unsigned long long)hi << 32) | lo;
            bk[j] = ok < bk[j] ? ok : bk[j];
        }
    }
    if (cl == 0) {
#pragma unroll
        for (int j = 0; j < kFineQ; j++) {
            const int qs = grp * kBox + quad * kFineQ + j;
            if (qs < a.n_q) {
                const int orig = a.q_perm[(size_t)smp * a.n_q + qs];
                a.out_d[(size_t)smp * a.n_q + orig] = __uint_as_float((unsigned)(bk[j] >> 32));
                a.out_i[(size_t)smp * a.n_q + orig] = (int)(bk[j] & 0xffffffffu);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Spatial sort: one workgroup per (sample, cloud) orders the points along a 30-bit Hilbert curve with a
// bitonic sort of (code << 32 | index) keys in LDS and writes the sorted SoA coordinates, the inverse
// permutation (rank) and one bounding box per 16 consecutive sorted points.  Clouds too large for the LDS
// sort (> 16384 points) keep their original order: culling then simply finds little to skip.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned part1by2(unsigned v) {
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ff;
    v = (v | (v << 8)) & 0x0300f00f;
    v = (v | (v << 4)) & 0x030c30c3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

// 30-bit Hilbert index of a 10-bit lattice point (Skilling's axes-to-transpose, then bit interleave).
// Unlike the Morton order, every contiguous run of the Hilbert order is spatially compact, so all owner
// tiles get similar, small bounding boxes (a Morton run that straddles an octant boundary spans the cloud).
__device__ __forceinline__ unsigned hilbert3(unsigned x0, unsigned x1, unsigned x2) {
    unsigned X[3] = {x0, x1, x2};
    const unsigned M = 1u << 9;
    for (unsigned Q = M; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const unsigned t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t;
    X[1] ^= t;
    X[2] ^= t;
    return (part1by2(X[0]) << 2) | (part1by2(X[1]) << 1) | part1by2(X[2]);
}

struct SortArgs {  // one entry per cloud; blockIdx.y selects it
    int n[2], n4[2], nb[2], npad[2];
    const float *xyz[2];
    // input layout: coordinate c of point i of sample s sits at xyz[s*sstride + i*pstride + c*cstride]; channels
    // >= nch read as 0 (the k-NN graph sorts channels-major clouds of 1..3 channels with the same kernel)
    long long sstride[2], pstride[2], cstride[2];
    int nch[2];
    float *soa[2];
    int *rank[2];
    int *perm[2];      // sorted position -> original index (inverse of rank)
    float4 *aos[2];    // optional [b][n]: (x, y, z, original index as bits) per sorted point (nn_sorted_kernel)
    float *box[2];
    // zero-fill riding along (replaces two memset launches): the two workgroups of a sample clear one region each
    float *zero[2];
    long long zero_stride[2], zero_count[2];  // per-sample stride and length in floats (multiples of 4)
    int *live_cnt;                            // [b][kLiveRow] live-owner counters of the passes B, cleared here
    unsigned *live_mask;                      // [b][kLevels][mask_words] live bits of set2: rows 4.. are preset to ones here
    int mask_words;
};

// Bitonic sort of NPAD = kSortT*SLOTS 32-bit keys held in registers (element i = tid*SLOTS + slot) by an 8-wave
// workgroup: strides < SLOTS are exchanges between a thread's registers, strides < 64*SLOTS between lanes (DPP where the
// partner is a quad / row permutation), only the three longest strides go through LDS.  Fully unrolled so that every
// register index is static.  A key is (truncated Hilbert code << idx_bits) | point index: keys are unique and
// one v_min_u32 / v_max_u32 pair is a whole compare-exchange.
constexpr int kSortT = 512;

template <int SLOTS>
__device__ __forceinline__ void bitonic_sort(unsigned (&key)[SLOTS], unsigned *lds, int tid) {
    // element tid * SLOTS + s sits in slot s of thread tid (a thread's keys are neighbours): the SHORT strides -- the ones
    // every merge repeats -- are exchanges between registers, the middle ones between lanes, and only the three longest
    // strides (6 stages of the 66 at 2048 keys) cross waves through LDS.  (With element tid + 512 s the three strides 64 /
    // 128 / 256 went through LDS, 12 stages with two barriers each: half of the sort's time by in-kernel stamps.)
    constexpr int NPAD = kSortT * SLOTS;
#pragma unroll
    for (int kk = 2; kk <= NPAD; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j < SLOTS) {
#pragma unroll
                for (int s = 0; s < SLOTS; s++) {
                    const int sp = s ^ j;
                    if (sp > s) {
                        const bool asc = ((tid * SLOTS + s) & kk) == 0;
                        const unsigned mn = min(key[s], key[sp]), mx = max(key[s], key[sp]);
                        key[s] = asc ? mn : mx;
                        key[sp] = asc ? mx : mn;
                    }
                }
            } else {
                const int L = j / SLOTS;  // the partner is slot s of thread tid ^ L
                if (L >= 64) {
                    __syncthreads();
#pragma unroll
                    for (int s = 0; s < SLOTS; s++) lds[tid + kSortT * s] = key[s];
                    __syncthreads();
                }
#pragma unroll
                for (int s = 0; s < SLOTS; s++) {
                    // the partner lane ^ L: one DPP move for L = 1, 2 (quad permutations) and 8 (a rotation by 8 of the row
                    // of 16 IS lane ^ 8), two rotations and a select for 4; the LDS crossbar (ds_bpermute) for 16 and 32
                    unsigned other;
                    if (L >= 64) other = lds[(tid ^ L) + kSortT * s];
                    else if (L == 1) other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key[s], 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
                    else if (L == 2) other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key[s], 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
                    else if (L == 8) other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key[s], 0x128, 0xf, 0xf, false);  // row_ror:8
                    else if (L == 4) {
                        // (row_ror:n hands lane i the value of lane i - n of its row)
                        const unsigned lo4 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key[s], 0x124, 0xf, 0xf, false);   // from lane - 4
                        const unsigned hi4 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key[s], 0x12C, 0xf, 0xf, false);   // from lane - 12 = lane + 4
                        other = (tid & 4) ? lo4 : hi4;
                    } else other = (unsigned)__shfl_xor((int)key[s], L, 64);
                    const bool take_min = ((tid & L) == 0) == (((tid * SLOTS + s) & kk) == 0);
                    key[s] = take_min ? min(key[s], other) : max(key[s], other);
                }
            }
        }
    }
}

template <int SLOTS>
__global__ __launch_bounds__(kSortT) void am_sort_kernel(SortArgs a) {
    constexpr bool MIRROR = SLOTS <= 8;  // clouds of up to 4096 points keep their coordinates in LDS (48 KB) for the gather
    __shared__ unsigned lds_keys[kSortT * SLOTS];
    __shared__ float lds_xyz[MIRROR ? 3 * kSortT * SLOTS : 1];
    __shared__ float red[6][16];
    const int which = blockIdx.y;
    const int n = a.n[which], n4 = a.n4[which], nb = a.nb[which], npad = a.npad[which];
    const int smp = blockIdx.x, tid = threadIdx.x, T = kSortT;
    if (a.live_cnt && which == 0 && (threadIdx.x < kInfSlot || threadIdx.x == kErrSlot))  // (kInfSlot.. are set below)
        a.live_cnt[(size_t)blockIdx.x * kLiveRow + threadIdx.x] = 0;
    if (a.live_mask && which == 1)  // (rows 4.. of the live masks start as all ones: PhaseArgs::mask_out)
        for (int i = 4 * a.mask_words + threadIdx.x; i < kLevels * a.mask_words; i += kSortT)
            a.live_mask[(size_t)blockIdx.x * kLevels * a.mask_words + i] = ~0u;
    if (a.zero[which]) {  // fire-and-forget stores, hidden under the sort
        float4 *z = reinterpret_cast<float4 *>(a.zero[which] + (size_t)smp * a.zero_stride[which]);
        const long long cnt4 = a.zero_count[which] / 4;
        for (long long i = tid; i < cnt4; i += T) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float *p = a.xyz[which] + (size_t)smp * a.sstride[which];
    const long long ps = a.pstride[which], cs = a.cstride[which];
    const int nch = a.nch[which];
    auto coord = [&](int i, int c) -> float { return c < nch ? p[i * ps + c * cs] : 0.f; };
    float *so = a.soa[which] ? a.soa[which] + (size_t)smp * 3 * n4 : nullptr;
    int *rk = a.rank[which] ? a.rank[which] + (size_t)smp * n : nullptr;
    int *pm = a.perm[which] + (size_t)smp * n;
    float4 *ao = a.aos[which] ? a.aos[which] + (size_t)smp * n : nullptr;
    float *bx = a.box[which] + (size_t)smp * nb * 8;
    int idx_bits = 10;
    while ((1 << idx_bits) < npad) idx_bits++;  // npad >= 1024
    const unsigned idx_mask = (1u << idx_bits) - 1;
    if (npad) {
        const int code_shift = 30 - 3 * ((32 - idx_bits) / 3);  // keep the leading 3*floor((32-idx_bits)/3) code bits
        // The thread's points: every load issued before anything consumes one (ONE round trip; read in a loop with the
        // min/max next to each load, and again for the keys, the kernel waited out eight), kept in registers for the
        // bounding box and the keys, and mirrored in LDS where the cloud fits, for the gather behind the sort.
        // (up to 16 slots -- 8192 points -- stay in registers; larger clouds read their points twice, as they come)
        constexpr bool KEEP = SLOTS <= 16;
        constexpr int BATCH = SLOTS < 8 ? SLOTS : 8;
        constexpr int NKEEP = KEEP ? SLOTS : BATCH;
        float px[NKEEP], py[NKEEP], pz[NKEEP];
        const int c1 = min(1, nch - 1), c2 = min(2, nch - 1);
        const int k1 = -(int)(nch > 1), k2 = -(int)(nch > 2);  // channels >= nch read as 0
        auto load_batch = [&](int s0, int r0) {  // slots s0 .. s0 + BATCH - 1 into registers r0 ..
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                const long long i = min(tid * SLOTS + (s0 + u), n - 1);
                px[r0 + u] = p[i * ps];
                py[r0 + u] = p[i * ps + c1 * cs];
                pz[r0 + u] = p[i * ps + c2 * cs];
            }
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                py[r0 + u] = __int_as_float(__float_as_int(py[r0 + u]) & k1);
                pz[r0 + u] = __int_as_float(__float_as_int(pz[r0 + u]) & k2);
            }
        };
        float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
        float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        if (!KEEP) {
            for (int i = tid; i < n; i += T)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float v = coord(i, c);
                    lo[c] = fminf(lo[c], v);
                    hi[c] = fmaxf(hi[c], v);
                }
        }
#pragma unroll
        for (int s0 = 0; s0 < (KEEP ? SLOTS : 0); s0 += BATCH) {
            const int r0 = s0;
            load_batch(s0, r0);
#pragma unroll
            for (int u = 0; u < BATCH; u++) {  // (a slot past the cloud repeats the last point: no effect on the box)
                lo[0] = fminf(lo[0], px[r0 + u]); hi[0] = fmaxf(hi[0], px[r0 + u]);
                lo[1] = fminf(lo[1], py[r0 + u]); hi[1] = fmaxf(hi[1], py[r0 + u]);
                lo[2] = fminf(lo[2], pz[r0 + u]); hi[2] = fmaxf(hi[2], pz[r0 + u]);
                if (MIRROR) {
                    lds_xyz[tid * SLOTS + (s0 + u)] = px[r0 + u];
                    lds_xyz[kSortT * SLOTS + tid * SLOTS + (s0 + u)] = py[r0 + u];
                    lds_xyz[2 * kSortT * SLOTS + tid * SLOTS + (s0 + u)] = pz[r0 + u];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // (inside the rows of 16 lanes by DPP rotations, across the four rows through the LDS crossbar)
#define PCC_ROR(v, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xf, 0xf, false))
            lo[c] = fminf(lo[c], PCC_ROR(lo[c], 8)); hi[c] = fmaxf(hi[c], PCC_ROR(hi[c], 8));
            lo[c] = fminf(lo[c], PCC_ROR(lo[c], 4)); hi[c] = fmaxf(hi[c], PCC_ROR(hi[c], 4));
            lo[c] = fminf(lo[c], PCC_ROR(lo[c], 2)); hi[c] = fmaxf(hi[c], PCC_ROR(hi[c], 2));
            lo[c] = fminf(lo[c], PCC_ROR(lo[c], 1)); hi[c] = fmaxf(hi[c], PCC_ROR(hi[c], 1));
#undef PCC_ROR
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                lo[c] = fminf(lo[c], __shfl_xor(lo[c], off, 64));
                hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off, 64));
            }
            if ((tid & 63) == 0) {
                red[c][tid >> 6] = lo[c];
                red[3 + c][tid >> 6] = hi[c];
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float l = red[c][0], h = red[3 + c][0];
            for (int i = 1; i < kSortT / 64; i++) {
                l = fminf(l, red[c][i]);
                h = fmaxf(h, red[3 + c][i]);
            }
            lo[c] = l;
            hi[c] = h > l ? 1023.f / (h - l) : 0.f;  // scale
        }
        unsigned key[SLOTS];
        if (!KEEP) {
#pragma unroll
            for (int s2 = 0; s2 < SLOTS; s2++) {
                const int i = tid * SLOTS + s2;
                key[s2] = ~0u;
                if (i < n) {
                    const unsigned qx = (unsigned)fminf(fmaxf((coord(i, 0) - lo[0]) * hi[0], 0.f), 1023.f);
                    const unsigned qy = (unsigned)fminf(fmaxf((coord(i, 1) - lo[1]) * hi[1], 0.f), 1023.f);
                    const unsigned qz = (unsigned)fminf(fmaxf((coord(i, 2) - lo[2]) * hi[2], 0.f), 1023.f);
                    key[s2] = ((hilbert3(qx, qy, qz) >> code_shift) << idx_bits) | (unsigned)i;
                }
            }
        }
#pragma unroll
        for (int s0 = 0; s0 < (KEEP ? SLOTS : 0); s0 += BATCH) {
            const int r0 = s0;
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                const int i = tid * SLOTS + (s0 + u);
                const unsigned qx = (unsigned)fminf(fmaxf((px[r0 + u] - lo[0]) * hi[0], 0.f), 1023.f);
                const unsigned qy = (unsigned)fminf(fmaxf((py[r0 + u] - lo[1]) * hi[1], 0.f), 1023.f);
                const unsigned qz = (unsigned)fminf(fmaxf((pz[r0 + u] - lo[2]) * hi[2], 0.f), 1023.f);
                const unsigned kv = ((hilbert3(qx, qy, qz) >> code_shift) << idx_bits) | (unsigned)i;
                key[s0 + u] = i < n ? kv : ~0u;
            }
        }
        bitonic_sort<SLOTS>(key, lds_keys, tid);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; s++) lds_keys[tid * SLOTS + s] = key[s];
        __syncthreads();
    }
    // sorted SoA rows + inverse permutation; the box of every 16 consecutive sorted points falls out of a
    // 16-lane min/max butterfly on the coordinates the lanes already hold
    const int span = ((max(n4, nb * kBox) + 63) / 64) * 64;
    int has_inf = 0;
    auto emit = [&](int s) {  // sorted position s: its point, its rows, its box
        float x = 0.f, y = 0.f, z = 0.f;
        const bool real = s < n;
        if (real) {
            const int orig = npad ? (int)(lds_keys[s] & idx_mask) : s;
            if (MIRROR && npad) {
                x = lds_xyz[orig];
                y = lds_xyz[kSortT * SLOTS + orig];
                z = lds_xyz[2 * kSortT * SLOTS + orig];
            } else {
                x = coord(orig, 0);
                y = coord(orig, 1);
                z = coord(orig, 2);
            }
            has_inf |= (__builtin_isinf(x) || __builtin_isinf(y) || __builtin_isinf(z)) ? 1 : 0;
            if (rk) rk[orig] = s;
            pm[s] = orig;
            if (ao) ao[s] = make_float4(x, y, z, __int_as_float(orig));
        }
        if (so && s < n4) {
            so[s] = x;
            so[n4 + s] = y;
            so[2 * n4 + s] = z;
        }
        float l0 = real ? x : __builtin_inff(), l1 = real ? y : __builtin_inff(), l2 = real ? z : __builtin_inff();
        float h0 = real ? x : -__builtin_inff(), h1 = real ? y : -__builtin_inff(), h2 = real ? z : -__builtin_inff();
        // (min / max over the 16 lanes of a DPP row, in every lane: four rotations of the row)
        auto rot = [](float v, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, false)); };
        static_assert(kBox == 16, "a box is a DPP row");
#define PCC_ROW16(op, v)                                                        \
    v = op(v, rot(v, std::integral_constant<int, 0x128>{}));                    \
    v = op(v, rot(v, std::integral_constant<int, 0x124>{}));                    \
    v = op(v, rot(v, std::integral_constant<int, 0x122>{}));                    \
    v = op(v, rot(v, std::integral_constant<int, 0x121>{}));
        PCC_ROW16(fminf, l0) PCC_ROW16(fminf, l1) PCC_ROW16(fminf, l2)
        PCC_ROW16(fmaxf, h0) PCC_ROW16(fmaxf, h1) PCC_ROW16(fmaxf, h2)
#undef PCC_ROW16
        const int bb = s / kBox;
        if ((s & (kBox - 1)) == 0 && bb < nb) {
            float4 *dst = reinterpret_cast<float4 *>(bx + (size_t)bb * 8);
            dst[0] = make_float4(l0, l1, l2, 0.f);
            dst[1] = make_float4(h0, h1, h2, 0.f);
        }
    };
    if (npad && SLOTS <= 8) {  // (unrolled: the LDS reads of all of a thread's positions are in flight together)
#pragma unroll
        for (int k2 = 0; k2 < SLOTS; k2++)
            if (tid + k2 * T < span) emit(tid + k2 * T);  // (span is a multiple of 64: whole waves take the branch)
    } else {
        for (int s = tid; s < span; s += T) emit(s);
    }
    // An infinite coordinate makes every pair of its point exp(-inf) = 0 -- skipped here as an exact zero -- while the
    // reference goes on to multiply that 0 by sqrt(inf): its cost and gradients of the sample are NaN (approxmatch.cu:207,
    // 247-248).  The sample is flagged and the finish kernel reports NaN.  (NaN coordinates need no flag: they reach the
    // sums through the distances, as in the reference.)
    if (a.live_cnt) {
        const int any_inf = __syncthreads_or(has_inf);
        if (tid == 0) a.live_cnt[(size_t)blockIdx.x * kLiveRow + kInfSlot + which] = any_inf ? 1 : 0;
    }
}

// The phases run in Hilbert-sorted index space; this puts the nine (ratioL | ratioR) level vectors back into
// the caller's point order for the materialise pass (contiguous loads there) and fills
// temp = remainL | remainR | ratioL | ratioR of the last level (approxmatch.cu:4).
__global__ __launch_bounds__(256) void am_unpermute_kernel(int n, int m, int n4, int m4,
                                                            const float *__restrict__ lv_sorted,
                                                            const float *__restrict__ rem_sorted,
                                                            const int *__restrict__ rank1,
                                                            const int *__restrict__ rank2,
                                                            float *__restrict__ lv, float *__restrict__ temp) {
    // sorted-space rows are [ratioL (n4) | ratioR (m4)] (16-byte aligned halves); outputs are dense [n | m]
    const int smp = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n + m) return;
    const int s = i < n ? rank1[(size_t)smp * n + i] : n4 + rank2[(size_t)smp * m + (i - n)];
    const size_t nm = (size_t)n + m, nm4 = (size_t)n4 + m4;
    const float *src = lv_sorted + (size_t)smp * kLevels * nm4;
    float *dst = lv + (size_t)smp * kLevels * nm;
    float last = 0.f;
#pragma unroll
    for (int l = 0; l < kLevels; l++) {
        last = src[(size_t)l * nm4 + s];
        dst[(size_t)l * nm + i] = last;
    }
    // remain row: remainL (n4) | remainR ping (m4) | pong (m4); the nine passes B leave the final remainR in pong
    float *tb = temp + (size_t)smp * 2 * nm;
    tb[i] = rem_sorted[(size_t)smp * (nm4 + m4) + (i < n ? s : s + m4)];
    tb[nm + i] = last;
}

// ---------------------------------------------------------------------------------------------------
// Materialise: match[b,l,k] = sum_i (exp2(c_i d2) * ratioL_i[k]) * ratioR_i[l], i = 0..8 in the
// reference's accumulation order (approxmatch.cu:154-155).  Write-only on match (float4 rows).
// Optionally also accumulates cost partials sum match*sqrt(d2) (matchcost, :207-208) so that the
// Python-level match_cost forward needs no second pass over match.
// ---------------------------------------------------------------------------------------------------
constexpr int kMatLT = 64;   // l rows per workgroup
constexpr int kMatKT = 256;  // k columns per workgroup (4 per lane)

template <bool COST, bool VEC>
__global__ __launch_bounds__(256) void am_materialise_kernel(int n, int m, const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2,
                                                              const float *__restrict__ lv, LevelConsts lc,
                                                              float *__restrict__ match,
                                                              float *__restrict__ cost_part) {
    __shared__ float4 lds_l[kMatLT][3];  // (x,y,z,rr0) (rr1..rr4) (rr5..rr8)
    __shared__ float lds_red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int smp = blockIdx.z;
    const int l0 = blockIdx.y * kMatLT;
    const int k0 = blockIdx.x * kMatKT + lane * 4;
    const size_t lvs = (size_t)kLevels * (n + m);
    const float *lvb = lv + (size_t)smp * lvs;
    const float *p1 = xyz1 + (size_t)smp * n * 3;
    const float *p2 = xyz2 + (size_t)smp * m * 3;
    const int lcnt = min(kMatLT, m - l0);

    if (tid < lcnt) {
        const int l = l0 + tid;
        float rr[kLevels];
#pragma unroll
        for (int i = 0; i < kLevels; i++) rr[i] = lvb[(size_t)i * (n + m) + n + l];
        lds_l[tid][0] = make_float4(p2[l * 3 + 0], p2[l * 3 + 1], p2[l * 3 + 2], rr[0]);
        lds_l[tid][1] = make_float4(rr[1], rr[2], rr[3], rr[4]);
        lds_l[tid][2] = make_float4(rr[5], rr[6], rr[7], rr[8]);
    }
    float x1[4], y1[4], z1[4], rl[kLevels][4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int k = k0 + q;
        k = k < n ? k : n - 1;
        x1[q] = p1[k * 3 + 0];
        y1[q] = p1[k * 3 + 1];
        z1[q] = p1[k * 3 + 2];
#pragma unroll
        for (int i = 0; i < kLevels; i++) rl[i][q] = lvb[(size_t)i * (n + m) + k];
    }
    __syncthreads();
    float csum = 0.f;
    for (int li = w; li < lcnt; li += 4) {
        const float4 A = lds_l[li][0], B = lds_l[li][1], Cc = lds_l[li][2];
        const float rr[kLevels] = {A.w, B.x, B.y, B.z, B.w, Cc.x, Cc.y, Cc.z, Cc.w};
        // A query point whose capacity is used up has ratioR == 0 exactly at every later level (remainR is
        // clamped to 0, approxmatch.cu:109, and ratioR = consumption * remainR, :108): that level adds exactly 0
        // to the whole row, so its exponentials are skipped (wave-uniform).  Typically 4 of 9 levels are live.
        int live[kLevels];
#pragma unroll
        for (int i = 0; i < kLevels; i++) live[i] = __builtin_amdgcn_readfirstlane((int)(rr[i] != 0.f));
        float d[4], acc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            d[q] = sq3(A.x - x1[q], A.y - y1[q], A.z - z1[q]);
            acc[q] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < kLevels; i++) {
            if (live[i]) {
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] += (fast_exp2(lc.c[i] * d[q]) * rl[i][q]) * rr[i];
            }
        }
        float out[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            out[q] = acc[q];
            if (COST && k0 + q < n) csum = __builtin_fmaf(acc[q], __builtin_amdgcn_sqrtf(d[q]), csum);
        }
        float *row = match + ((size_t)smp * m + (l0 + li)) * n;
        if (VEC) {
            if (k0 + 3 < n) {
                // match is written once and read back only after the whole 512 MiB (far beyond the 256 MiB Infinity
                // Cache): non-temporal stores (A/B on MI355X: materialise 150 -> 127 us)
                v4f o4 = {out[0], out[1], out[2], out[3]};
                __builtin_nontemporal_store(o4, reinterpret_cast<v4f *>(row + k0));
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (k0 + q < n) row[k0 + q] = out[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (k0 + q < n) row[k0 + q] = out[q];
        }
    }
    if (COST) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) csum += __shfl_down(csum, off, 64);
        if (lane == 0) lds_red[w] = csum;
        __syncthreads();
        if (tid == 0)
            cost_part[(size_t)smp * gridDim.x * gridDim.y + blockIdx.y * gridDim.x + blockIdx.x] =
                ((lds_red[0] + lds_red[1]) + lds_red[2]) + lds_red[3];
    }
}

// out[b] = sum_p part[b][p] in index order (deterministic second stage of every cost reduction).
__global__ __launch_bounds__(256) void reduce_rows_kernel(int parts, const float *__restrict__ part,
                                                           float *__restrict__ out) {
    __shared__ float red[256];
    const int smp = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < parts; i += 256) s += part[(size_t)smp * parts + i];
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) out[smp] = red[0];
}

// ---------------------------------------------------------------------------------------------------
// matchcost / matchcostgrad2: "row" kernels.  A workgroup takes RT rows (query points l of set2) of
// one sample; set1 is staged SoA in LDS chunk by chunk; each wave streams whole rows of match with
// coalesced float4 loads (1 KiB per wave-instruction).
//   MODE 0: cost partial  = sum match * sqrt(d2)                         (approxmatch.cu:200-209)
//   MODE 1: grad2[k,:]    = sum_j match[k,j] (p2_k - p1_j) rsqrt(max(d2,1e-20))   (:239-246)
// ---------------------------------------------------------------------------------------------------
constexpr int kRowRT = 32;  // rows per workgroup -> 8 per wave

template <int MODE, bool VEC, int CH>
__global__ __launch_bounds__(256) void am_row_kernel(int n, int m, const float *__restrict__ xyz1,
                                                      const float *__restrict__ xyz2,
                                                      const float *__restrict__ match, float *__restrict__ out) {
    constexpr int RPW = kRowRT / 4;
    __shared__ __attribute__((aligned(16))) float lds_p[3 * CH];
    __shared__ float lds_red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int smp = blockIdx.y;
    const int r0 = blockIdx.x * kRowRT;
    const float *p1 = xyz1 + (size_t)smp * n * 3;
    const float *p2 = xyz2 + (size_t)smp * m * 3;
    const float4 *X4 = reinterpret_cast<const float4 *>(lds_p);
    const float4 *Y4 = X4 + CH / 4;
    const float4 *Z4 = Y4 + CH / 4;

    float acc[RPW][3];
#pragma unroll
    for (int i = 0; i < RPW; i++) acc[i][0] = acc[i][1] = acc[i][2] = 0.f;
    float csum = 0.f;

    for (int q0 = 0; q0 < n; q0 += CH) {
        const int cnt = min(CH, n - q0);
        if (q0) __syncthreads();
        for (int i = tid; i < cnt * 3; i += 256) {
            const float v = p1[(size_t)q0 * 3 + i];
            const int p = i / 3;
            lds_p[(i - p * 3) * CH + p] = v;
        }
        for (int i = cnt + tid; i < ((cnt + 3) & ~3); i += 256) lds_p[i] = lds_p[CH + i] = lds_p[2 * CH + i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPW; i++) {
            const int row = r0 + w + 4 * i;
            const bool live = row < m;  // wave-uniform
            const int rowc = live ? row : m - 1;
            const float x2 = p2[rowc * 3 + 0], y2 = p2[rowc * 3 + 1], z2 = p2[rowc * 3 + 2];
            const float *mrow = match + ((size_t)smp * m + rowc) * n + q0;
            for (int k = lane * 4; live && k < cnt; k += 256) {
                float mv[4];
                if (VEC && k + 3 < cnt) {
                    const v4f t4 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(mrow + k));
                    const float4 t = make_float4(t4.x, t4.y, t4.z, t4.w);
                    mv[0] = t.x; mv[1] = t.y; mv[2] = t.z; mv[3] = t.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) mv[q] = (k + q < cnt) ? mrow[k + q] : 0.f;
                }
                const float4 xs = X4[k >> 2], ys = Y4[k >> 2], zs = Z4[k >> 2];
                const float px[4] = {xs.x, xs.y, xs.z, xs.w};
                const float py[4] = {ys.x, ys.y, ys.z, ys.w};
                const float pz[4] = {zs.x, zs.y, zs.z, zs.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float dx = x2 - px[q], dy = y2 - py[q], dz = z2 - pz[q];
                    const float d = sq3(dx, dy, dz);
                    if (MODE == 0) {
                        csum = __builtin_fmaf(mv[q], __builtin_amdgcn_sqrtf(d), csum);
                    } else {
                        const float f = mv[q] * __builtin_amdgcn_rsqf(__builtin_fmaxf(d, 1e-20f));
                        acc[i][0] = __builtin_fmaf(dx, f, acc[i][0]);
                        acc[i][1] = __builtin_fmaf(dy, f, acc[i][1]);
                        acc[i][2] = __builtin_fmaf(dz, f, acc[i][2]);
                    }
                }
            }
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) csum += __shfl_down(csum, off, 64);
        if (lane == 0) lds_red[w] = csum;
        __syncthreads();
        if (tid == 0) out[(size_t)smp * gridDim.x + blockIdx.x] = ((lds_red[0] + lds_red[1]) + lds_red[2]) + lds_red[3];
    } else {
#pragma unroll
        for (int i = 0; i < RPW; i++) {
            const int row = r0 + w + 4 * i;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float v = acc[i][c];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0 && row < m) out[((size_t)smp * m + row) * 3 + c] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// matchcostgrad1: "column" kernel.  grad1[l,:] = sum_k match[k,l] (p1_l - p2_k) rsqrt(max(d2,1e-20))
// (approxmatch.cu:277-285).  A lane owns 4 consecutive columns l; the rows k are split RS ways over
// workgroups and 4 ways over waves; partial sums go to a workspace and are added in a fixed order.
// ---------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void am_col_kernel(int n, int m, int rs, const float *__restrict__ xyz1,
                                                      const float *__restrict__ xyz2,
                                                      const float *__restrict__ match, float *__restrict__ part) {
    __shared__ float red[3][4][64 * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int smp = blockIdx.z;
    const int split = blockIdx.y;
    const int k0 = blockIdx.x * 256 + lane * 4;
    const float *p1 = xyz1 + (size_t)smp * n * 3;
    const float *p2 = xyz2 + (size_t)smp * m * 3;
    const int rbeg = (int)((long long)m * split / rs), rend = (int)((long long)m * (split + 1) / rs);
    float x1[4], y1[4], z1[4], g[4][3];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int k = k0 + q;
        k = k < n ? k : n - 1;
        x1[q] = p1[k * 3 + 0];
        y1[q] = p1[k * 3 + 1];
        z1[q] = p1[k * 3 + 2];
        g[q][0] = g[q][1] = g[q][2] = 0.f;
    }
    for (int row = rbeg + w; row < rend; row += 4) {
        const float x2 = p2[row * 3 + 0], y2 = p2[row * 3 + 1], z2 = p2[row * 3 + 2];
        const float *mrow = match + ((size_t)smp * m + row) * n;
        float mv[4];
        if (VEC && k0 + 3 < n) {
            const float4 t = *reinterpret_cast<const float4 *>(mrow + k0);
            mv[0] = t.x; mv[1] = t.y; mv[2] = t.z; mv[3] = t.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) mv[q] = (k0 + q < n) ? mrow[k0 + q] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float dx = x1[q] - x2, dy = y1[q] - y2, dz = z1[q] - z2;
            const float f = mv[q] * __builtin_amdgcn_rsqf(__builtin_fmaxf(sq3(dx, dy, dz), 1e-20f));
            g[q][0] = __builtin_fmaf(dx, f, g[q][0]);
            g[q][1] = __builtin_fmaf(dy, f, g[q][1]);
            g[q][2] = __builtin_fmaf(dz, f, g[q][2]);
        }
    }
    // merge the 4 waves (fixed order), then store this split's partial [b][split][n][3]
    if (w > 0) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) red[w - 1][q][lane * 3 + c] = g[q][c];
    }
    __syncthreads();
    if (w == 0) {
        float *dst = part + (((size_t)smp * rs + split) * n) * 3;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (k0 + q >= n) continue;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = ((g[q][c] + red[0][q][lane * 3 + c]) + red[1][q][lane * 3 + c]) + red[2][q][lane * 3 + c];
                dst[(size_t)(k0 + q) * 3 + c] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// matchcostgrad, fused: ONE read of match produces both gradients (the reference reads it twice,
// approxmatch.cu:319-320).  A workgroup takes RT rows (points k of set2) x a 2048-column slab (points l of set1);
// a wave streams whole row segments with float4 loads; per element t = d * match * rsqrt(max(|d|^2,1e-20)):
//   grad1[l] += t   (column sums: 4 columns x 3 components per lane per 256-column step, kept in registers,
//                    merged over the 4 waves in LDS, written as one partial per row tile)
//   grad2[k] -= t   (row sums: per-lane partials, wave butterfly at the end of the row segment)
// Partials are combined in a fixed order by reduce_splits_kernel / the slab loop: deterministic.
// ---------------------------------------------------------------------------------------------------
constexpr int kGradRT = 64;     // rows per workgroup (16 per wave)
constexpr int kGradSlab = 1024;  // columns per slab = 4 steps of 256 (48 column-sum registers per lane; 2048 -> 175 us, 1024 -> 129 us, 512 -> 134 us at B=32,N=2048)

template <bool VEC>
__global__ __launch_bounds__(256) void am_grad_fused_kernel(int n, int m, int row_tiles,
                                                             const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2,
                                                             const float *__restrict__ match,
                                                             float *__restrict__ part1,  // [b][row_tiles][n][3]
                                                             float *__restrict__ part2,  // [b][slabs][m][3]
                                                             const float *__restrict__ scale2)  // applied when part2 IS grad2
{
    constexpr int STEPS = kGradSlab / 256;
    // set1 slab SoA (24 KiB); after the row loop the same bytes carry one wave's column sums at a time to wave 0
    __shared__ __attribute__((aligned(16))) float lds_p[3 * kGradSlab > STEPS * 12 * 64 ? 3 * kGradSlab : STEPS * 12 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int smp = blockIdx.z, slab = blockIdx.y, rt = blockIdx.x;
    const int c0 = slab * kGradSlab;
    const int cnt = min(kGradSlab, n - c0);
    const float *p1 = xyz1 + ((size_t)smp * n + c0) * 3;
    const float *p2 = xyz2 + (size_t)smp * m * 3;
    for (int i = tid; i < cnt * 3; i += 256) {
        const float v = p1[i];
        const int p = i / 3;
        lds_p[(i - p * 3) * kGradSlab + p] = v;
    }
    for (int i = cnt + tid; i < kGradSlab; i += 256) lds_p[i] = lds_p[kGradSlab + i] = lds_p[2 * kGradSlab + i] = 0.f;
    __syncthreads();
    const float4 *X4 = reinterpret_cast<const float4 *>(lds_p);
    const float4 *Y4 = X4 + kGradSlab / 4;
    const float4 *Z4 = Y4 + kGradSlab / 4;

    float g1[STEPS][4][3];
#pragma unroll
    for (int st = 0; st < STEPS; st++)
#pragma unroll
        for (int q = 0; q < 4; q++) g1[st][q][0] = g1[st][q][1] = g1[st][q][2] = 0.f;

    const int r_begin = rt * kGradRT, r_end = min(r_begin + kGradRT, m);
    const bool full = VEC && cnt == kGradSlab;  // whole slab, aligned: branch-free body, 8 row loads in flight
    for (int row = r_begin + w; row < r_end; row += 4) {
        const float x2 = p2[row * 3 + 0], y2 = p2[row * 3 + 1], z2 = p2[row * 3 + 2];
        const float *mrow = match + ((size_t)smp * m + row) * n + c0;
        float rx = 0.f, ry = 0.f, rz = 0.f;
        float mv[STEPS][4];
        if (full) {
#pragma unroll
            for (int st = 0; st < STEPS; st++) {
                const v4f t4 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(mrow + st * 256 + lane * 4));
                const float4 t = make_float4(t4.x, t4.y, t4.z, t4.w);  // read once: non-temporal (128 -> 119 us)
                mv[st][0] = t.x; mv[st][1] = t.y; mv[st][2] = t.z; mv[st][3] = t.w;
            }
        } else {
#pragma unroll
            for (int st = 0; st < STEPS; st++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int k = st * 256 + lane * 4 + q;
                    mv[st][q] = k < cnt ? mrow[k] : 0.f;  // columns past the slab contribute exactly 0
                }
        }
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            const int k = st * 256 + lane * 4;
            const float4 xs = X4[k >> 2], ys = Y4[k >> 2], zs = Z4[k >> 2];
            const float px[4] = {xs.x, xs.y, xs.z, xs.w};
            const float py[4] = {ys.x, ys.y, ys.z, ys.w};
            const float pz[4] = {zs.x, zs.y, zs.z, zs.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // grad1 uses (p1 - p2) (approxmatch.cu:281-284); grad2 the negated vector (:240-246)
                const float dx = px[q] - x2, dy = py[q] - y2, dz = pz[q] - z2;
                const float f = mv[st][q] * __builtin_amdgcn_rsqf(__builtin_fmaxf(sq3(dx, dy, dz), 1e-20f));
                const float tx = dx * f, ty = dy * f, tz = dz * f;
                g1[st][q][0] += tx;
                g1[st][q][1] += ty;
                g1[st][q][2] += tz;
                rx -= tx;
                ry -= ty;
                rz -= tz;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rx += __shfl_down(rx, off, 64);
            ry += __shfl_down(ry, off, 64);
            rz += __shfl_down(rz, off, 64);
        }
        if (lane == 0) {
            float *dst = part2 + (((size_t)smp * gridDim.y + slab) * m + row) * 3;
            const float sc = scale2 ? scale2[smp] : 1.0f;
            dst[0] = scale2 ? rx * sc : rx;
            dst[1] = scale2 ? ry * sc : ry;
            dst[2] = scale2 ? rz * sc : rz;
        }
    }
    // column partials: waves 1, 2, 3 hand their sums to wave 0 one after the other (fixed order)
    float *red = lds_p;
    for (int src = 1; src < 4; src++) {
        __syncthreads();
        if (w == src) {
#pragma unroll
            for (int st = 0; st < STEPS; st++)
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int c = 0; c < 3; c++) red[((st * 4 + q) * 3 + c) * 64 + lane] = g1[st][q][c];
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int st = 0; st < STEPS; st++)
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int c = 0; c < 3; c++) g1[st][q][c] += red[((st * 4 + q) * 3 + c) * 64 + lane];
        }
    }
    if (w == 0) {
        float *dst = part1 + (((size_t)smp * row_tiles + rt) * n + c0) * 3;
#pragma unroll
        for (int st = 0; st < STEPS; st++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int k = st * 256 + lane * 4 + q;
                if (k < cnt) {
#pragma unroll
                    for (int c = 0; c < 3; c++) dst[(size_t)k * 3 + c] = g1[st][q][c];
                }
            }
    }
}

// grad1[b][i] = sum_s part[b][s][i]  (i over n*3), fixed order.
__global__ __launch_bounds__(256) void reduce_splits_kernel(int rs, size_t per_sample, const float *__restrict__ part,
                                                             const float *__restrict__ scale, float *__restrict__ out) {
    const int smp = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_sample) return;
    const float *p = part + (size_t)smp * rs * per_sample + i;
    float s = p[0];
    for (int t = 1; t < rs; t++) s += p[t * per_sample];
    out[(size_t)smp * per_sample + i] = scale ? s * scale[smp] : s;  // optional upstream gradient (match_cost.py:41-42)
}

// ---------------------------------------------------------------------------------------------------
// Implicit match ("pair" kernel): what the Python-level match_cost needs is cost[b] and, when the clouds require
// gradients, grad1 / grad2 -- never the 512 MiB match tensor itself (reference match_cost.py:25-27,39-42 keeps it on
// ctx only to feed MatchCostGrad, and the gradient treats match as a constant).  This kernel evaluates every match
// element in registers exactly as am_materialise_kernel does (same level order, same rounding) and feeds it straight
// into the cost sum (approxmatch.cu:207-208) and both gradient sums (:239-246, :277-285): no store, no re-read.
// It runs in the Hilbert-sorted index space of the phase kernels, which buys two exact skips the materialising path
// cannot have (match is laid out in the caller's order there):
//   * a level whose exp2(c_i d2) underflows to 0 for every pair (row point, 64Q-column box) is skipped
//     (wave-uniform; the same test as V_CULL, applied to all levels);
//   * a row whose live levels are all skipped for this column box contributes exactly 0: no distance, no sqrt.
// Mapping: workgroup = kPairRT rows (set2 points, sorted) x 64Q columns (set1 points, sorted); a lane owns Q
// consecutive columns (coordinates, the nine ratioL values and the column sums stay in registers), the 4 waves deal
// the rows round-robin, row data is broadcast from LDS.  Row sums: per-lane partials are parked in LDS and folded
// eight rows at a time (48 lanes x 32 sequential adds + one shuffle), so the VALU never runs a 64-lane butterfly per
// row.  All partials are combined in a fixed order by the second-stage kernels: deterministic.
// ---------------------------------------------------------------------------------------------------
constexpr int kPairRT = 128;  // rows per workgroup (32 per wave; 256: partials halve, 437 vs 432 us per call)
constexpr int kPairRB = 8;    // rows per row-sum fold
constexpr int kPairPad = 65;  // stash row pitch (floats): lanes of one fold hit distinct banks

struct PairArgs {
    int n, m, n4, m4;
    const float *soa1, *soa2;   // [b][3][n4] / [b][3][m4] sorted coordinates
    const float *lv;            // [b][9][n4 + m4] sorted level rows: ratioL | ratioR
    LevelConsts lc;
    float cut2[kLevels];        // a level is exactly 0 beyond this squared distance
    float *cost_part;           // [b][gridDim.y * gridDim.x]
    float *part1;               // [b][row_tiles][n4][3]   column sums (grad1, sorted space)
    float *part2;               // [b][col_blocks][m4][3]  row sums    (grad2, sorted space)
    int col_blocks, row_tiles, bc;  // 1-D grid of col_blocks * row_tiles * bc workgroups (see the kernel)
    int plain_order;
};

template <int Q, bool GRAD>
__global__ __launch_bounds__(256) void am_pair_kernel(PairArgs a) {
    __shared__ float4 lds_l[kPairRT][3];  // (x,y,z,rr0) (rr1..rr4) (rr5..rr8)
    __shared__ int lds_mask[kPairRT];     // bit i: level i contributes to this row segment
    __shared__ float stash[GRAD ? 4 * kPairRB * 3 * kPairPad : 4];
    __shared__ float lds_red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Dispatch order = work order.  A workgroup's duration goes from ~0 (every row of the tile masked out for this column
    // box) to the full 128 x 256 pairs on all live levels, and the heavy ones are the (row tile, column block) pairs that
    // are CLOSE in space -- close along the two Hilbert orders.  In (x, y, z) grid order the last workgroups dispatched
    // were as likely heavy as light and the chip idled 22 % of the kernel behind them (SQ_BUSY_CU_CYCLES).  The 1-D grid
    // is read shift-major instead: for every shift 0, +1, -1, +2, ... of the row tile against the column block's own
    // position along the curve, every column block, every sample -- near pairs first, far (short) ones last.
    const int cbn = a.col_blocks, rtn = a.row_tiles;
    const int item = (int)blockIdx.x / a.bc, smp = (int)blockIdx.x - item * a.bc;
    const int shift_k = item / cbn, cblk = item - shift_k * cbn;
    const int base_r = (int)(((long long)(2 * cblk + 1) * rtn) / (2 * cbn));
    const int shift = ((shift_k + 1) >> 1) * ((shift_k & 1) ? 1 : -1);  // 0, +1, -1, +2, ... : a complete residue system mod rtn
    int rtile = ((base_r + shift) % rtn + rtn) % rtn;
    if (a.plain_order) rtile = shift_k;  // (A/B switch 1: row tiles in index order)
    const int l0 = rtile * kPairRT;
    const int kb = cblk * 64 * Q;
    const int k0 = kb + lane * Q;
    const size_t nm4 = (size_t)a.n4 + a.m4;
    const float *lvb = a.lv + (size_t)smp * kLevels * nm4;
    const float *s1 = a.soa1 + (size_t)smp * 3 * a.n4;
    const float *s2 = a.soa2 + (size_t)smp * 3 * a.m4;
    const int lcnt = min(kPairRT, a.m - l0);

    // this thread's row (the first lcnt threads finish one row each once the column box is known)
    float rowx = 0.f, rowy = 0.f, rowz = 0.f, rowr[kLevels];
#pragma unroll
    for (int i = 0; i < kLevels; i++) rowr[i] = 0.f;
    if (tid < lcnt) {
        const int l = l0 + tid;
        rowx = s2[l];
        rowy = s2[a.m4 + l];
        rowz = s2[2 * a.m4 + l];
#pragma unroll
        for (int i = 0; i < kLevels; i++) rowr[i] = lvb[(size_t)i * nm4 + a.n4 + l];
    }
    float x1[Q], y1[Q], z1[Q], rl[kLevels][Q];
    float blo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float bhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const bool real = k0 + q < a.n;
        const int k = real ? k0 + q : a.n - 1;
        x1[q] = s1[k];
        y1[q] = s1[a.n4 + k];
        z1[q] = s1[2 * a.n4 + k];
#pragma unroll
        for (int i = 0; i < kLevels; i++) rl[i][q] = real ? lvb[(size_t)i * nm4 + k] : 0.f;  // a padded column weighs 0
        blo[0] = fminf(blo[0], x1[q]); bhi[0] = fmaxf(bhi[0], x1[q]);
        blo[1] = fminf(blo[1], y1[q]); bhi[1] = fmaxf(bhi[1], y1[q]);
        blo[2] = fminf(blo[2], z1[q]); bhi[2] = fmaxf(bhi[2], z1[q]);
    }
    // bounding box of the 64Q columns of this workgroup (every wave holds the same columns)
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            blo[c] = fminf(blo[c], __shfl_xor(blo[c], off, 64));
            bhi[c] = fmaxf(bhi[c], __shfl_xor(bhi[c], off, 64));
        }
    }
    if (tid < lcnt) {
        // level i adds exactly 0 to the whole row segment if ratioR_i == 0 (exhausted query point,
        // approxmatch.cu:108-109) or if every exponential underflows: every pair (row, column of the box) has
        // d2 >= bd2
        const float bx = fmaxf(fmaxf(blo[0] - rowx, rowx - bhi[0]), 0.f);
        const float by = fmaxf(fmaxf(blo[1] - rowy, rowy - bhi[1]), 0.f);
        const float bz = fmaxf(fmaxf(blo[2] - rowz, rowz - bhi[2]), 0.f);
        const float bd2 = bx * bx + by * by + bz * bz;
        int mask = 0;
#pragma unroll
        for (int i = 0; i < kLevels; i++) mask |= (rowr[i] != 0.f && !(bd2 > a.cut2[i])) ? (1 << i) : 0;
        lds_l[tid][0] = make_float4(rowx, rowy, rowz, rowr[0]);
        lds_l[tid][1] = make_float4(rowr[1], rowr[2], rowr[3], rowr[4]);
        lds_l[tid][2] = make_float4(rowr[5], rowr[6], rowr[7], rowr[8]);
        lds_mask[tid] = mask;
    }
    float g1[Q][3];
#pragma unroll
    for (int q = 0; q < Q; q++) g1[q][0] = g1[q][1] = g1[q][2] = 0.f;
    float csum = 0.f;
    float *my_stash = stash + (GRAD ? w * kPairRB * 3 * kPairPad : 0);
    __syncthreads();

    // wave w takes rows w, w+4, ...; kPairRB of them per fold
    for (int base = 0; base < lcnt; base += 4 * kPairRB) {
#pragma unroll 1
        for (int s = 0; s < kPairRB; s++) {
            const int li = base + 4 * s + w;
            float rx = 0.f, ry = 0.f, rz = 0.f;
            const int mask = li < lcnt ? __builtin_amdgcn_readfirstlane(lds_mask[li]) : 0;
            if (mask) {
                const float4 A = lds_l[li][0], B = lds_l[li][1], Cc = lds_l[li][2];
                const float rr[kLevels] = {A.w, B.x, B.y, B.z, B.w, Cc.x, Cc.y, Cc.z, Cc.w};
                float ex[Q], ey[Q], ez[Q], d[Q], acc[Q];
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    ex[q] = A.x - x1[q];  // p2 - p1 (approxmatch.cu:148-150)
                    ey[q] = A.y - y1[q];
                    ez[q] = A.z - z1[q];
                    d[q] = sq3(ex[q], ey[q], ez[q]);
                    acc[q] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < kLevels; i++) {
                    if (mask & (1 << i)) {
                        // w = exp(level d2) ratioL ratioR; match += w (approxmatch.cu:153-155): the product is added
                        // with one fma (the contraction a compiler applies to `match += a * b`); am_materialise_kernel
                        // rounds the product first -- the two differ by half an ulp of the product
#pragma unroll
                        for (int q = 0; q < Q; q++)
                            acc[q] = __builtin_fmaf(fast_exp2(a.lc.c[i] * d[q]) * rl[i][q], rr[i], acc[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    if (GRAD) {
                        // max(d2, 1e-20): d2 is never NaN, so the bare instruction (no canonicalising pre-pass)
                        float dm;
                        asm("v_max_f32 %0, %1, %2" : "=v"(dm) : "v"(d[q]), "v"(1e-20f));
                        const float f = acc[q] * __builtin_amdgcn_rsqf(dm);
                        // sqrt(d2) = d2 * rsqrt(d2): one transcendental serves both sums (d2 < 1e-20 moves the cost by < 1e-10)
                        csum = __builtin_fmaf(f, d[q], csum);
                        // t = (p2 - p1) match / |p1 - p2|: grad2 (rows) accumulates +t (approxmatch.cu:240-246), grad1
                        // (columns) the negated vector (:281-284)
                        g1[q][0] = __builtin_fmaf(-ex[q], f, g1[q][0]);
                        g1[q][1] = __builtin_fmaf(-ey[q], f, g1[q][1]);
                        g1[q][2] = __builtin_fmaf(-ez[q], f, g1[q][2]);
                        rx = __builtin_fmaf(ex[q], f, rx);
                        ry = __builtin_fmaf(ey[q], f, ry);
                        rz = __builtin_fmaf(ez[q], f, rz);
                    } else {
                        csum = __builtin_fmaf(acc[q], __builtin_amdgcn_sqrtf(d[q]), csum);
                    }
                }
            }
            if (GRAD) {
                my_stash[(s * 3 + 0) * kPairPad + lane] = rx;
                my_stash[(s * 3 + 1) * kPairPad + lane] = ry;
                my_stash[(s * 3 + 2) * kPairPad + lane] = rz;
            }
        }
        if (GRAD) {
            // fold the eight rows: lane (v, h) adds half h of vector v = (slot, component) in index order, the two
            // halves meet through one shuffle.  The stash is private to the wave: no workgroup barrier.
            const int v = lane % 24, h = lane / 24;
            float t = 0.f;
            __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order; keep the compiler to it
            if (lane < 48) {
                const float *src = my_stash + v * kPairPad + h * 32;
#pragma unroll 8
                for (int i = 0; i < 32; i++) t += src[i];
            }
            __builtin_amdgcn_wave_barrier();
            const float hi = __shfl(t, lane + 24, 64);
            const int sl = v / 3, c = v - sl * 3;
            const int li = base + 4 * sl + w;
            if (lane < 24 && li < lcnt)
                a.part2[(((size_t)smp * cbn + cblk) * a.m4 + (l0 + li)) * 3 + c] = t + hi;
        }
    }
    // cost partial of this workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) csum += __shfl_down(csum, off, 64);
    if (lane == 0) lds_red[w] = csum;
    __syncthreads();
    if (tid == 0)
        a.cost_part[(size_t)smp * cbn * rtn + rtile * cbn + cblk] =
            ((lds_red[0] + lds_red[1]) + lds_red[2]) + lds_red[3];
    if (GRAD) {
        // column sums: waves 1..3 hand theirs to wave 0 one after the other (fixed order); the stash is free now
        static_assert(Q * 3 * 64 <= 4 * kPairRB * 3 * kPairPad, "column merge reuses the stash");
        for (int src = 1; src < 4; src++) {
            __syncthreads();
            if (w == src) {
#pragma unroll
                for (int q = 0; q < Q; q++)
#pragma unroll
                    for (int c = 0; c < 3; c++) stash[(q * 3 + c) * 64 + lane] = g1[q][c];
            }
            __syncthreads();
            if (w == 0) {
#pragma unroll
                for (int q = 0; q < Q; q++)
#pragma unroll
                    for (int c = 0; c < 3; c++) g1[q][c] += stash[(q * 3 + c) * 64 + lane];
            }
        }
        if (w == 0) {
            float *dst = a.part1 + (((size_t)smp * rtn + rtile) * a.n4) * 3;
#pragma unroll
            for (int q = 0; q < Q; q++) {
                if (k0 + q < a.n) {
#pragma unroll
                    for (int c = 0; c < 3; c++) dst[(size_t)(k0 + q) * 3 + c] = g1[q][c];
                }
            }
        }
    }
}

// Second stage of the implicit path: partials are added in index order (deterministic) and the gradients carried from
// the sorted index space back to the caller's point order through `rank`.
// The three second-stage reductions of the implicit path in ONE launch (blockIdx.z: 0 = grad1, 1 = grad2, 2 = cost).
struct FinishArgs {
    int parts[3], npts[2], pitch[2];
    const float *part[3];
    const int *perm[2];   // sorted position -> caller's point index
    const float *scale;
    float *out[3];
    // the Chamfer half of a ChamferEMD call rides along (blockIdx.z == 3): loss[b] = sum / mean of the two distance rows
    const float *ch_d1, *ch_d2;
    float *ch_loss;
    int ch_n, ch_m, ch_mean;
    const int *flags;  // live-counter rows [b][kLiveRow] (slots kInfSlot, kInfSlot + 1), or null
};
__global__ __launch_bounds__(256) void pair_finish_kernel(FinishArgs f) {
    __shared__ float red[256];
    const int which = blockIdx.z, smp = blockIdx.y, tid = threadIdx.x;
    if (which == 3) {  // the same fixed-order tree as chamfer_reduce_kernel (chamfer.hip): the same bits
        if (blockIdx.x) return;
        __shared__ float red2[256];
        float s1 = 0.f, s2 = 0.f;
        // (eight loads in flight, added in the same order: one at a time this slice was a chain of n / 256 round trips, the
        // longest of the launch)
        auto strided_sum = [&](const float *d, int cnt) -> float {
            float acc = 0.f;
            for (int i0 = tid; i0 < cnt; i0 += 8 * 256) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = d[min(i0 + u * 256, cnt - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (i0 + u * 256 < cnt) acc += v[u];
            }
            return acc;
        };
        s1 = strided_sum(f.ch_d1 + (size_t)smp * f.ch_n, f.ch_n);
        s2 = strided_sum(f.ch_d2 + (size_t)smp * f.ch_m, f.ch_m);
        red[tid] = s1;
        red2[tid] = s2;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                red[tid] += red[tid + off];
                red2[tid] += red2[tid + off];
            }
            __syncthreads();
        }
        if (tid == 0) f.ch_loss[smp] = f.ch_mean ? red2[0] / (float)f.ch_m + red[0] / (float)f.ch_n : red[0] + red2[0];
        return;
    }
    // a sample with an infinite coordinate: NaN cost and gradients (see am_sort_kernel)
    // ... or whose resident fine-level passes did not complete (am_fine_persist_kernel: a sample barrier timed out)
    const bool poisoned = f.flags && (f.flags[(size_t)smp * kLiveRow + kInfSlot] | f.flags[(size_t)smp * kLiveRow + kInfSlot + 1] |
                                      f.flags[(size_t)smp * kLiveRow + kErrSlot]);
    if (which == 2) {  // cost[b] = sum of the workgroup partials, fixed order
        if (blockIdx.x) return;
        const int parts = f.parts[2];
        float s = 0.f;
        for (int i = tid; i < parts; i += 256) s += f.part[2][(size_t)smp * parts + i];
        red[tid] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) f.out[2][smp] = poisoned ? __builtin_nanf("") : red[0];
        return;
    }
    if (!f.out[which]) return;
    // a thread owns component c of SORTED position s: the partial rows are read as straight coalesced streams (they
    // are the bulk: parts x npts x 12 bytes); only the 12-byte result is scattered to the caller's point order
    const int npts = f.npts[which], pitch = f.pitch[which], parts = f.parts[which];
    const int i = blockIdx.x * 256 + tid;
    if (i >= npts * 3) return;
    const int s = i / 3, c = i - s * 3;
    const float *p = f.part[which] + (size_t)smp * parts * pitch * 3 + i;
    const int pt = f.perm[which][(size_t)smp * npts + s];
    // the partials in the fixed order t = 0, 1, 2 ..., eight loads in flight (one at a time, the fold is a chain of
    // `parts` memory round trips: most of this kernel's time)
    const size_t stride = (size_t)pitch * 3;
    float acc = p[0];
    int t = 1;
    for (; t + 8 <= parts; t += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(size_t)(t + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    {
        float v[7];
#pragma unroll
        for (int u = 0; u < 7; u++) v[u] = p[(size_t)min(t + u, parts - 1) * stride];
#pragma unroll
        for (int u = 0; u < 7; u++)
            if (t + u < parts) acc += v[u];
    }
    f.out[which][((size_t)smp * npts + pt) * 3 + c] = poisoned ? __builtin_nanf("") : f.scale ? acc * f.scale[smp] : acc;
}

// ---- host side -------------------------------------------------------------------------------------
constexpr int kPhCH = 2048;
// the dense list of pass C/A holds the LIVE candidates only (under half of the cloud from level 3 on): chunks of 1024
// (25-29 KB of LDS: five or six workgroups per CU instead of three; a longer list takes a second chunk)
constexpr int kClistCH = 1024;

static bool cull_enabled() {  // measurement switch (pcc_test_hooks.h): every work-skipping variant off
    return pcc::tuning(PCC_TUNE_AM_NOCULL) == 0;
}

template <int MODE>
const char *phase_name(int level) {
    static const char *const names[4][kLevels] = {
        {"am_phase_kernel<A> L0", "", "", "", "", "", "", "", ""},
        {"am_phase_kernel<B> L0", "am_phase_kernel<B> L1", "am_phase_kernel<B> L2", "am_phase_kernel<B> L3",
         "am_phase_kernel<B> L4", "am_phase_kernel<B> L5", "am_phase_kernel<B> L6", "am_phase_kernel<B> L7",
         "am_phase_kernel<B> L8"},
        {"am_phase_kernel<CA> L0", "am_phase_kernel<CA> L1", "am_phase_kernel<CA> L2", "am_phase_kernel<CA> L3",
         "am_phase_kernel<CA> L4", "am_phase_kernel<CA> L5", "am_phase_kernel<CA> L6", "am_phase_kernel<CA> L7", ""},
        {"", "", "", "", "", "", "", "", "am_phase_kernel<C> L8"}};
    return names[MODE][level];
}

template <int MODE, int R, int S, int G = 1>
int launch_phase_rs(PhaseArgs a, int b, int var, hipStream_t st, const char *what) {
    a.tiles = pcc::ceil_div(a.n_own, 64 * R / G);
    a.batch = b;
    const long long grid = (long long)b * a.tiles;
    if (grid > 0x7fffffffLL) return pcc::invalid("approxmatch: grid too large");
    {
        pcc::ProfScope prof(phase_name<MODE>(a.level), st);
        const dim3 g((unsigned)grid), blk(64 * S);
        // only the combinations the schedule uses are instantiated
        if (var == V_CLIST && (MODE == PH_CA || MODE == PH_C))
            hipLaunchKernelGGL((am_phase_kernel<(MODE == PH_CA || MODE == PH_C) ? MODE : PH_C, R, S, kClistCH, V_CLIST>), g, blk, 0, st, a);
        else if (var == V_COWN && MODE == PH_B)
            hipLaunchKernelGGL((am_phase_kernel<PH_B, (G > 1 ? 1 : R), S, kPhCH, V_COWN, G>), g, blk, 0, st, a);
        else hipLaunchKernelGGL((am_phase_kernel<MODE, R, S, kPhCH, V_PLAIN>), g, blk, 0, st, a);
    }
    return pcc::check_launch(what);
}

template <int MODE>
int launch_fine(PhaseArgs a, int b, hipStream_t st, const char *what) {
    a.tiles = pcc::ceil_div(a.n_own, 64);
    a.batch = b;
    const long long grid = (long long)b * a.tiles;
    if (grid > 0x7fffffffLL) return pcc::invalid("approxmatch: grid too large");
    {
        pcc::ProfScope prof(phase_name<MODE>(a.level), st);
        hipLaunchKernelGGL((am_fine_kernel<(MODE == PH_C ? PH_CA : MODE), kPhCH>), dim3((unsigned)grid), dim3(64 * kFineS), 0, st, a);
    }
    return pcc::check_launch(what);
}

template <int MODE>
int launch_phase(const PhaseArgs &a, int b, int var, hipStream_t st, const char *what) {
    // the box-culled passes of the fine levels: 16-owner groups
    if (var == V_CULL) return launch_fine<MODE>(a, b, st, what);
    // Owner compaction packs the live owners into the first tiles; a full tile takes as long as before (just on fewer
    // CUs), so these launches use the smallest tile (64 owners x 8 waves), and where few owners are left (recon / uniform
    // clouds: ~25 % live at level 5, 5 % at level 8) an owner is spread over 2 / 4 lanes: the pair loop of a workgroup,
    // which is the launch's critical path, gets that much shorter (first levels 3 / 4 and 4 / 6: 2-5 us slower)
    if (var == V_COWN) {
        if (a.live_in && a.level >= 5) return launch_phase_rs<MODE, 1, 8, 4>(a, b, var, st, what);
        if (a.live_in && a.level >= 4) return launch_phase_rs<MODE, 1, 8, 2>(a, b, var, st, what);
        return launch_phase_rs<MODE, 1, 8>(a, b, var, st, what);
    }
    // Workgroup shape: R owners per lane x S candidate slices (waves).  One wave can issue a VALU instruction only every
    // 4 cycles while a SIMD retires one every 2, so a phase needs >= 2 (better 4) waves per SIMD = 2048-4096 waves on
    // 256 CUs; R is spent only once the chip is full (each LDS broadcast read is then amortised over R owners).
    // (A large batch arrives here as two concurrent half-batch lanes: 32768 owners per launch at B=32, N=2048, where
    // 128-owner tiles measured 492 us per forward+backward against 509 us for 64-owner tiles.)
    const long long owners = (long long)b * a.n_own;
    // pass C/A on the dense list of live candidates: 25 KB of LDS per 64-owner workgroup (chunks of kClistCH), so the
    // 2 x 512 workgroups of two half-batch lanes are resident together and every owner's loop is half as long as on the
    // 128-owner tile (432.9 -> 429.1 us per match_cost call; with the 45 KB carve of 2048-candidate chunks it measured slower)
    if (var == V_CLIST && owners < 4LL * 65536) return launch_phase_rs<MODE, 1, 8>(a, b, var, st, what);
    if (owners >= 4LL * 65536) return launch_phase_rs<MODE, 4, 8>(a, b, var, st, what);
    if (owners >= 32768) return launch_phase_rs<MODE, 2, 8>(a, b, var, st, what);
    return launch_phase_rs<MODE, 1, 8>(a, b, var, st, what);
}

struct StreamBuf {  // stream-ordered scratch from the library's private pool (pcc::ws_malloc / ws_free)
    void *p = nullptr;
    hipStream_t st;
    explicit StreamBuf(hipStream_t s) : st(s) {}
    int alloc(size_t bytes) {
        if (pcc::ws_malloc(&p, bytes, st) != hipSuccess) {
            pcc::set_error(PCC_ENOMEM, "workspace allocation failed");
            return PCC_ENOMEM;
        }
        return PCC_OK;
    }
    ~StreamBuf() {
        if (p) (void)pcc::ws_free(p, st);
    }
};

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

size_t cost_parts(int n, int m) { return (size_t)pcc::ceil_div(n, kMatKT) * pcc::ceil_div(m, kMatLT); }

// Workspace carve (bytes, every section 16-byte aligned).
inline int mask_words(int m4) { return (m4 + 31) / 32; }

struct WsLayout {
    int n4, m4, nb1, nb2;
    size_t soa1, soa2, rank1, rank2, perm1, perm2, box1, box2, rem, lv, lv_orig, cpart, clist, clist_cnt, live_cnt, live_mask, aos1, aos2, total;
    WsLayout(int b, int n, int m) {
        auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
        n4 = (n + 3) & ~3;
        m4 = (m + 3) & ~3;
        nb1 = pcc::ceil_div(n, kBox);
        nb2 = pcc::ceil_div(m, kBox);
        size_t o = 0;
        soa1 = o; o = up(o + (size_t)b * 3 * n4 * 4);
        soa2 = o; o = up(o + (size_t)b * 3 * m4 * 4);
        rank1 = o; o = up(o + (size_t)b * n * 4);
        rank2 = o; o = up(o + (size_t)b * m * 4);
        perm1 = o; o = up(o + (size_t)b * n * 4);
        perm2 = o; o = up(o + (size_t)b * m * 4);
        box1 = o; o = up(o + (size_t)b * nb1 * 8 * 4);
        box2 = o; o = up(o + (size_t)b * nb2 * 8 * 4);
        rem = o; o = up(o + (size_t)b * ((size_t)n4 + 2 * (size_t)m4) * 4);    // sorted space: remainL | remainR x2
        lv = o; o = up(o + (size_t)b * kLevels * ((size_t)n4 + m4) * 4);      // sorted space, padded halves
        lv_orig = o; o = up(o + (size_t)b * kLevels * ((size_t)n + m) * 4);
        cpart = o; o = up(o + (size_t)b * cost_parts(n, m) * 4);
        clist = o; o = up(o + (size_t)b * 5 * m4 * 4);   // dense candidate list handed from pass B to pass C/A
        clist_cnt = o; o = up(o + (size_t)b * 4);
        live_cnt = o; o = up(o + (size_t)b * kLiveRow * 4);
        live_mask = o; o = up(o + (size_t)b * kLevels * mask_words(m4) * 4);  // live bits of set2 per level (V_COWN)
        aos1 = o; o = up(o + (size_t)b * n * 16);   // packed sorted points for the nearest-neighbour search of pcc_chamfer_emd
        aos2 = o; o = up(o + (size_t)b * m * 16);
        total = o;
    }
};

void launch_sort(const SortArgs &a, int slots, dim3 grid, hipStream_t st) {
    pcc::ProfScope prof("am_sort_kernel", st);
    switch (slots) {
    case 4: hipLaunchKernelGGL((am_sort_kernel<4>), grid, dim3(kSortT), 0, st, a); break;
    case 8: hipLaunchKernelGGL((am_sort_kernel<8>), grid, dim3(kSortT), 0, st, a); break;
    case 16: hipLaunchKernelGGL((am_sort_kernel<16>), grid, dim3(kSortT), 0, st, a); break;
    case 32: hipLaunchKernelGGL((am_sort_kernel<32>), grid, dim3(kSortT), 0, st, a); break;
    default: hipLaunchKernelGGL((am_sort_kernel<64>), grid, dim3(kSortT), 0, st, a); break;
    }
}

int sort_clouds(int b, const WsLayout &L, int n, int m, const float *xyz1, const float *xyz2, float *soa1, float *soa2,
                int *rank1, int *rank2, int *perm1, int *perm2, float *box1, float *box2, float *rem, float *lv,
                int *live_cnt, unsigned *live_mask, float4 *aos1, float4 *aos2, hipStream_t st) {
    SortArgs a{};
    a.live_cnt = live_cnt;
    a.live_mask = live_mask;
    a.mask_words = mask_words(L.m4);
    a.aos[0] = aos1; a.aos[1] = aos2;
    // the padded tails of the weight rows are staged as float4: they must be finite (their candidates sit at the
    // origin with these weights), and V_COWN relies on zero-filled level arrays for the exhausted owners it never
    // touches: remain rows are cleared by the workgroup sorting set1, level rows by the one sorting set2
    a.zero[0] = rem; a.zero_stride[0] = a.zero_count[0] = (long long)L.n4 + 2LL * L.m4;
    a.zero[1] = lv; a.zero_stride[1] = a.zero_count[1] = (long long)kLevels * ((long long)L.n4 + L.m4);
    const int nn[2] = {n, m};
    int slots = 4;
    for (int w = 0; w < 2; w++) {
        int npad = 4 * kSortT;
        while (npad < nn[w]) npad <<= 1;
        if (npad > 64 * kSortT) npad = 0;  // > 16384 points: keep the original order (nothing is culled)
        a.n[w] = nn[w];
        a.npad[w] = npad;
        slots = std::max(slots, npad / kSortT);
    }
    for (int w = 0; w < 2; w++)
        if (a.npad[w]) a.npad[w] = kSortT * slots;  // one SLOTS instantiation serves both clouds
    a.n4[0] = L.n4; a.n4[1] = L.m4; a.nb[0] = L.nb1; a.nb[1] = L.nb2;
    a.xyz[0] = xyz1; a.xyz[1] = xyz2; a.soa[0] = soa1; a.soa[1] = soa2;
    for (int w = 0; w < 2; w++) {
        a.sstride[w] = (long long)nn[w] * 3; a.pstride[w] = 3; a.cstride[w] = 1; a.nch[w] = 3;
    }
    a.rank[0] = rank1; a.rank[1] = rank2; a.perm[0] = perm1; a.perm[1] = perm2; a.box[0] = box1; a.box[1] = box2;
    launch_sort(a, slots, dim3(b, 2), st);
    return pcc::check_launch("approxmatch(sort)");
}

// One internal side stream per device: the second half of a large batch runs its 19 dependent phase launches there
// while the first half runs on the caller's stream, so that one half's kernels fill the launch / drain bubbles of
// the other's (each launch is a chain link of ~20 us with 4-8 us of fixed cost).  Fork and join are events on the
// caller's stream: for the caller the call still is "enqueue on `stream`, no host synchronisation".
constexpr int kMaxLanes = 4;
hipStream_t side_stream(int which) {  // which = 0 .. kMaxLanes - 2
    static std::mutex mu;
    static hipStream_t streams[64][kMaxLanes - 1] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || which < 0 || which >= kMaxLanes - 1) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!streams[dev][which]) {
        // HIP multiplexes the streams of a process onto a few hardware queues per PRIORITY class, in creation order: a
        // plain side stream created after an application has made many streams of its own (RCCL does, at
        // init_process_group) can land on the hardware queue of the caller's stream, and the two lanes then serialise
        // (measured: 0.53 -> 0.73 ms per bench step).  A high-priority stream comes from a different queue pool than the
        // caller's normal-priority stream.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
            (void)hipGetLastError();
            least = greatest = 0;
        }
        if (hipStreamCreateWithPriority(&streams[dev][which], hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&streams[dev][which], hipStreamNonBlocking) != hipSuccess) {
                streams[dev][which] = nullptr;
                (void)hipGetLastError();
            }
        }
    }
    return streams[dev][which];
}

struct ForkJoin {  // the side streams wait for everything enqueued on main so far; at scope exit main waits for them
    hipStream_t main, side[kMaxLanes - 1] = {};
    int nside = 0;
    bool ok = false;
    ForkJoin(hipStream_t m, int want) : main(m) {
        if (want <= 0) return;
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return;
        if (hipEventRecord(ev, main) == hipSuccess) {
            for (int i = 0; i < want && i < kMaxLanes - 1; i++) {
                hipStream_t s = side_stream(i);
                if (!s || hipStreamWaitEvent(s, ev, 0) != hipSuccess) break;
                side[nside++] = s;
            }
        }
        (void)hipEventDestroy(ev);  // released once the recorded work has completed
        ok = nside > 0;
    }
    ~ForkJoin() {
        for (int i = 0; i < nside; i++) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                (void)hipStreamSynchronize(side[i]);  // cannot order the streams any other way
                continue;
            }
            (void)hipEventRecord(ev, side[i]);
            (void)hipStreamWaitEvent(main, ev, 0);
            (void)hipEventDestroy(ev);
        }
    }
};

// Per-device state of the resident fine-level launch (am_fine_persist_kernel): a sticky failure word in mapped host
// memory, the event of the last such launch (two of them never run at the same time: a launch on another stream first
// waits for the previous one) and whether the device holds enough workgroups of the kernel at once.
struct ResidentState {
    unsigned *host_word = nullptr, *dev_word = nullptr;
    hipEvent_t last = nullptr;
    hipStream_t last_stream = nullptr;
    int wg_per_cu = -1, cus = 0;  // resident workgroups of am_fine_persist_kernel per CU (-1: not asked yet), compute units
};
std::mutex g_resident_mu;
ResidentState *resident_state() {  // (call with g_resident_mu held)
    static ResidentState st[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    ResidentState &r = st[dev];
    if (r.wg_per_cu < 0) {
        r.wg_per_cu = 0;
        void *h = nullptr, *d = nullptr;
        int per_cu = 0, cus = 0;
        if (hipHostMalloc(&h, sizeof(unsigned), hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess &&
            hipEventCreateWithFlags(&r.last, hipEventDisableTiming) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(am_fine_persist_kernel), 64 * kFineS, 0) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
            r.host_word = static_cast<unsigned *>(h);
            r.dev_word = static_cast<unsigned *>(d);
            *r.host_word = 0;
            r.wg_per_cu = per_cu;
            r.cus = cus;
        } else {
            (void)hipGetLastError();
        }
    }
    return &r;
}
// nonzero if an earlier resident launch on this device timed out (and clears the word)
unsigned take_resident_failure() {
    std::lock_guard<std::mutex> lk(g_resident_mu);
    ResidentState *r = resident_state();
    if (!r || !r->host_word) return 0;
    return __atomic_exchange_n(r->host_word, 0u, __ATOMIC_RELAXED);
}
static bool resident_enabled() {  // measurement switch (pcc_test_hooks.h): one launch per pass at the fine levels too
    return pcc::tuning(PCC_TUNE_AM_NORESIDENT) == 0;
}
// the seven fine-level passes of the samples of `sc` as one resident launch; returns -1 when the device / the sizes do
// not qualify (the caller then runs one launch per pass)
int launch_fine_resident(const Sched &sc, int bc, hipStream_t st) {
    if (!sc.skip || !sc.live_cnt || !sc.live_mask || sc.dbg || !resident_enabled()) return -1;
    if (sc.n > kFpCH || sc.m > kFpCH || sc.n < 1 || sc.m < 1) return -1;
    const int tiles = pcc::ceil_div(std::max(sc.n, sc.m), 64);
    FinePersistArgs a{};
    a.n = sc.n; a.m = sc.m; a.n4 = sc.n4; a.m4 = sc.m4; a.nb1 = sc.nb1; a.nb2 = sc.nb2; a.tiles = tiles;
    a.soa1 = sc.soa1; a.soa2 = sc.soa2; a.box1 = sc.box1; a.box2 = sc.box2;
    a.rem = sc.rem; a.lv = sc.lv; a.multiL = sc.multiL; a.multiR = sc.multiR;
    for (int i = 0; i < 4; i++) {
        a.c[i] = sc.lc.c[i];
        a.cut2[i] = kZeroExp / -sc.lc.c[i];
    }
    a.live_cnt = sc.live_cnt;
    a.live_mask = sc.live_mask;
    a.mask_words = mask_words(sc.m4);
    {
        std::lock_guard<std::mutex> lk(g_resident_mu);
        ResidentState *r = resident_state();
        // the whole launch resident at once on half of the device's workgroup slots
        if (!r || r->wg_per_cu < 2 || (long long)bc * tiles > r->cus) return -1;
        a.host_err = r->dev_word;
        if (r->last_stream && r->last_stream != st) (void)hipStreamWaitEvent(st, r->last, 0);
        {
            pcc::ProfScope prof("am_fine_persist_kernel", st);
            hipLaunchKernelGGL(am_fine_persist_kernel, dim3((unsigned)(bc * tiles)), dim3(64 * kFineS), 0, st, a);
        }
        if (hipEventRecord(r->last, st) == hipSuccess) r->last_stream = st;
    }
    return pcc::check_launch("approxmatch(resident fine levels)");
}

// Sort + the 19 passes: leaves the nine (ratioL | ratioR) level rows and remainL | remainR in the workspace, in the
// Hilbert-sorted index space.
int run_levels(int b, int n, int m, const float *xyz1, const float *xyz2, const WsLayout &L, char *base, hipStream_t st,
               const std::function<int(int, int, hipStream_t)> &lane_tail = nullptr,
               const std::function<int(int, int, hipStream_t)> &after_sort = nullptr) {
    if (const unsigned fw = take_resident_failure()) {
        char buf[320];
        std::snprintf(buf, sizeof buf, "approxmatch: an earlier call on this device did not complete (a sample barrier of the resident "
                      "fine-level launch timed out: barrier %u of sample %u saw %u arrivals; the implicit path reported NaN for those "
                      "samples); this call was not started", (fw >> 4) & 0xfu, fw >> 20, (fw >> 8) & 0xfffu);
        return pcc::invalid(buf);
    }
    const LevelConsts lc = make_levels();
    float multiL, multiR;  // approxmatch.cu:6-12 (integer division)
    if (n >= m) { multiL = 1; multiR = (float)(n / m); }
    else { multiL = (float)(m / n); multiR = 1; }
    const long long nm4 = (long long)L.n4 + L.m4;  // sorted-space level row: ratioL (n4) | ratioR (m4)
    const long long rs = (long long)L.n4 + 2LL * L.m4;  // remain row: remainL (n4) | remainR ping (m4) | pong (m4)

    static int *dbg_counters = [] {
        int *p = nullptr;
        const char *e = std::getenv("PCC_AM_DEBUG");
        if (e && (e[0] == '1' || e[0] == '2') && hipMalloc(reinterpret_cast<void **>(&p), kDbgInts * sizeof(int)) == hipSuccess)
            (void)hipMemset(p, 0, kDbgInts * sizeof(int));
        return p;
    }();
    static const int dbg_counts = [] {
        const char *e = std::getenv("PCC_AM_DEBUG");
        return (e && e[0] == '1') ? 1 : 0;
    }();

    const bool split_enabled = pcc::tuning(PCC_TUNE_AM_NOSPLIT) == 0;  // (measurement switch: everything on the caller's stream)

    // Lanes: disjoint sample ranges that run the same schedule on different streams.  Two lanes when each half still
    // is a sizeable launch (B=32, N=2048: EMD forward+backward 530 -> 49x us); every workspace section is indexed
    // [sample][...], so a lane is the same schedule on pointers offset to its first sample.
    struct Lane {
        int s0, bc;
        hipStream_t st;
        Sched sc;
    };
    Lane lanes[kMaxLanes];
    int nlanes = 1, want_side = 0;
    if (split_enabled && !dbg_counters && b >= 8 && (long long)b * std::max(n, m) >= 32768) {
        // not while the caller's stream is being captured into a graph: the capture stays a single-stream chain
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
            (void)hipGetLastError();
            cap = hipStreamCaptureStatusNone;
        }
        if (cap == hipStreamCaptureStatusNone) {
            want_side = 1;
            const int t = pcc::tuning(PCC_TUNE_AM_LANES);  // (measurement switch: lane count)
            if (t >= 1 && t <= kMaxLanes) want_side = t - 1;
        }
    }
    ForkJoin fj(st, want_side);
    if (fj.ok) nlanes = 1 + fj.nside;
    for (int l = 0; l < nlanes; l++) {
        Lane &ln = lanes[l];
        ln.s0 = (int)((long long)b * l / nlanes);
        ln.bc = (int)((long long)b * (l + 1) / nlanes) - ln.s0;
        ln.st = l == 0 ? st : fj.side[l - 1];
        const size_t s0 = (size_t)ln.s0;
        Sched &sc = ln.sc;
        sc = Sched{};
        sc.n = n; sc.m = m; sc.n4 = L.n4; sc.m4 = L.m4; sc.nb1 = L.nb1; sc.nb2 = L.nb2;
        sc.soa1 = reinterpret_cast<float *>(base + L.soa1) + s0 * 3 * L.n4;
        sc.soa2 = reinterpret_cast<float *>(base + L.soa2) + s0 * 3 * L.m4;
        sc.box1 = reinterpret_cast<float *>(base + L.box1) + s0 * L.nb1 * 8;
        sc.box2 = reinterpret_cast<float *>(base + L.box2) + s0 * L.nb2 * 8;
        sc.rem = reinterpret_cast<float *>(base + L.rem) + s0 * rs;
        sc.lv = reinterpret_cast<float *>(base + L.lv) + s0 * kLevels * nm4;
        sc.multiL = multiL; sc.multiR = multiR;
        sc.skip = cull_enabled() ? 1 : 0;
        sc.lc = lc;
        sc.dbg = l == 0 ? dbg_counters : nullptr;
        sc.dbg_counts = dbg_counts;
        {
            sc.clist = reinterpret_cast<float *>(base + L.clist) + s0 * 5 * L.m4;
            sc.clist_cnt = reinterpret_cast<int *>(base + L.clist_cnt) + s0;
        }
        if (sc.skip) sc.live_cnt = reinterpret_cast<int *>(base + L.live_cnt) + s0 * kLiveRow;
        if (sc.skip) sc.live_mask = reinterpret_cast<unsigned *>(base + L.live_mask) + s0 * kLevels * mask_words(L.m4);
    }
    int rc = PCC_OK;
    const bool nn_all_at_head = pcc::tuning(PCC_TUNE_NN_HEAD) != 0;  // (measurement switch)
    auto enqueue_head = [&](int l) -> int {
        const Lane &ln = lanes[l];
        const size_t s0 = (size_t)ln.s0;
        int r = sort_clouds(ln.bc, L, n, m, xyz1 + s0 * n * 3, xyz2 + s0 * m * 3, const_cast<float *>(ln.sc.soa1),
                            const_cast<float *>(ln.sc.soa2), reinterpret_cast<int *>(base + L.rank1) + s0 * n,
                            reinterpret_cast<int *>(base + L.rank2) + s0 * m, reinterpret_cast<int *>(base + L.perm1) + s0 * n,
                            reinterpret_cast<int *>(base + L.perm2) + s0 * m, const_cast<float *>(ln.sc.box1),
                            const_cast<float *>(ln.sc.box2), ln.sc.rem, ln.sc.lv,
                            reinterpret_cast<int *>(base + L.live_cnt) + s0 * kLiveRow,
                            reinterpret_cast<unsigned *>(base + L.live_mask) + s0 * kLevels * mask_words(L.m4),
                            after_sort ? reinterpret_cast<float4 *>(base + L.aos1) + s0 * n : nullptr,
                            after_sort ? reinterpret_cast<float4 *>(base + L.aos2) + s0 * m : nullptr, ln.st);
        // work that only needs the sorted clouds of this lane's samples (pcc_chamfer_emd: the nearest-neighbour search).
        // Even lanes run it here, odd lanes behind their passes: two searches at the same moment halve each other (each
        // wants every SIMD); against the other lane's pass chain a search costs less (chamfer_emd 447.7 -> 440.4 us, step
        // 470 -> 462.6 us, tools/ab_nn.py; before passes 3 / 7 / 11 / 15 of the odd lane: 451 / 445 / 445 / 445 us; on a
        // stream of its own: 522 us).
        if (!r && after_sort && (l % 2 == 0 || nn_all_at_head)) r = after_sort(ln.s0, ln.bc, ln.st);
        return r;
    };
    auto enqueue_pass = [&](int l, int p) -> int {
        const Lane &ln = lanes[l];
        int mode, var;
        const PhaseArgs a = build_phase(ln.sc, p, &mode, &var);
        switch (mode) {
        case PH_A: return launch_phase<PH_A>(a, ln.bc, var, ln.st, "approxmatch(A)");
        case PH_B: return launch_phase<PH_B>(a, ln.bc, var, ln.st, "approxmatch(B)");
        case PH_CA: return launch_phase<PH_CA>(a, ln.bc, var, ln.st, "approxmatch(CA)");
        default: return launch_phase<PH_C>(a, ln.bc, var, ln.st, "approxmatch(C)");
        }
    };
    int first_pass[kMaxLanes] = {};
    {
        for (int l = 0; l < nlanes && !rc; l++) rc = enqueue_head(l);
        if (rc) return rc;
        {
            // pass p of every lane is enqueued before pass p+1 of any: the streams advance together
            pcc::ProfScope seq0("am_phase_sequence", lanes[0].st, true);
            pcc::ProfScope seq1("am_phase_sequence", lanes[nlanes > 1 ? 1 : 0].st, true, nlanes >= 2);
            // the seven passes of levels 0-2 as one resident launch per lane where the device and the sizes allow it
            for (int l = 0; l < nlanes && !rc; l++) {
                const int r = launch_fine_resident(lanes[l].sc, lanes[l].bc, lanes[l].st);
                if (r > 0) rc = r;
                else if (r == 0) first_pass[l] = kFpPasses;
            }
            for (int p = 0; p < sched_phases() && !rc; p++)
                for (int l = 0; l < nlanes && !rc; l++)
                    if (p >= first_pass[l]) rc = enqueue_pass(l, p);
            if (rc) return rc;
        }
        // what follows the passes for one lane's samples (the implicit path's pair + finish kernels) goes on that lane's
        // stream: the lane that finishes its passes first starts at once instead of waiting for the join
        if (after_sort && !nn_all_at_head)
            for (int l = 1; l < nlanes; l += 2)
                if (int rc2 = after_sort(lanes[l].s0, lanes[l].bc, lanes[l].st)) return rc2;
        if (lane_tail) {
            for (int l = 0; l < nlanes; l++)
                if (int rc2 = lane_tail(lanes[l].s0, lanes[l].bc, lanes[l].st)) return rc2;
        }
    }
    if (dbg_counters) {
        static int h[kDbgInts];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, dbg_counters, sizeof h, hipMemcpyDeviceToHost);
        std::fprintf(stderr, "[pcc dbg] A: %d/%d skipped;", h[1], h[0]);
        for (int i = 0; i < kLevels; i++) std::fprintf(stderr, " B%d %d/%d CA%d %d/%d;", i, h[3 + 4 * i], h[2 + 4 * i], i, h[5 + 4 * i], h[4 + 4 * i]);
        if (!dbg_counts) {
            std::fprintf(stderr, "\n[pcc dbg] phase stamps x10ns (first | middle | last workgroup): start-after-first [prologue loads staged loop reduced end]");
            for (int q = 0; q < sched_phases(); q++) {
                const int *s = h + 64 + 24 * q;
                std::fprintf(stderr, "\n[pcc dbg]  p%02d", q);
                for (int k = 0; k < 3; k++) {
                    const int *t = s + 8 * k;
                    std::fprintf(stderr, " | +%d [%d %d %d %d %d %d]", (t[0] - s[0]) & 0x7fffffff, t[1], t[2], t[3], t[4], t[5], t[6]);
                }
            }
        }
        std::fprintf(stderr, "\n");
        (void)hipMemset(dbg_counters, 0, sizeof h);
    }
    return PCC_OK;
}

int approxmatch_impl(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                     void *workspace, size_t workspace_bytes, float *cost_out, hipStream_t st) {
    const WsLayout L(b, n, m);
    if (workspace_bytes < L.total) return pcc::invalid("approxmatch: workspace too small");
    if (!aligned16(workspace)) return pcc::invalid("approxmatch: workspace must be 16-byte aligned");
    char *base = static_cast<char *>(workspace);
    int rc = run_levels(b, n, m, xyz1, xyz2, L, base, st);
    if (rc) return rc;
    int *rank1 = reinterpret_cast<int *>(base + L.rank1), *rank2 = reinterpret_cast<int *>(base + L.rank2);
    float *rem = reinterpret_cast<float *>(base + L.rem);
    float *lv = reinterpret_cast<float *>(base + L.lv);
    float *lv_orig = reinterpret_cast<float *>(base + L.lv_orig);
    float *cpart = reinterpret_cast<float *>(base + L.cpart);
    const LevelConsts lc = make_levels();
    hipLaunchKernelGGL(am_unpermute_kernel, dim3(pcc::ceil_div(n + m, 256), b), dim3(256), 0, st, n, m, L.n4, L.m4, lv, rem,
                       rank1, rank2, lv_orig, temp);
    rc = pcc::check_launch("approxmatch(unpermute)");
    if (rc) return rc;
    const dim3 grid(pcc::ceil_div(n, kMatKT), pcc::ceil_div(m, kMatLT), b);
    const bool vec = (n % 4 == 0) && aligned16(match);
    if (cost_out) {
        {
            pcc::ProfScope prof("am_materialise_kernel<cost>", st);
            if (vec) hipLaunchKernelGGL((am_materialise_kernel<true, true>), grid, dim3(256), 0, st, n, m, xyz1, xyz2, lv_orig, lc, match, cpart);
            else hipLaunchKernelGGL((am_materialise_kernel<true, false>), grid, dim3(256), 0, st, n, m, xyz1, xyz2, lv_orig, lc, match, cpart);
        }
        rc = pcc::check_launch("approxmatch(materialise+cost)");
        if (rc) return rc;
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(b), dim3(256), 0, st, (int)cost_parts(n, m), cpart, cost_out);
        return pcc::check_launch("approxmatch(cost reduce)");
    }
    {
        pcc::ProfScope prof("am_materialise_kernel", st);
        if (vec) hipLaunchKernelGGL((am_materialise_kernel<false, true>), grid, dim3(256), 0, st, n, m, xyz1, xyz2, lv_orig, lc, match, nullptr);
        else hipLaunchKernelGGL((am_materialise_kernel<false, false>), grid, dim3(256), 0, st, n, m, xyz1, xyz2, lv_orig, lc, match, nullptr);
    }
    return pcc::check_launch("approxmatch(materialise)");
}

// cost[b] (and grad1 / grad2 when both are non-null) of the Python-level match_cost without materialising match.
template <int Q>
int launch_pair(const PairArgs &pa, dim3 grid, bool grad, hipStream_t st) {
    pcc::ProfScope prof(grad ? "am_pair_kernel<grad>" : "am_pair_kernel<cost>", st);
    if (grad) hipLaunchKernelGGL((am_pair_kernel<Q, true>), grid, dim3(256), 0, st, pa);
    else hipLaunchKernelGGL((am_pair_kernel<Q, false>), grid, dim3(256), 0, st, pa);
    return pcc::check_launch("match_cost(pair)");
}

int match_cost_implicit_impl(int b, int n, int m, const float *xyz1, const float *xyz2, const float *grad_cost,
                             float *cost, float *grad1, float *grad2, hipStream_t st,
                             const pcc::ChamferOut *chamfer = nullptr) {
    const WsLayout L(b, n, m);
    constexpr int q_cols = 4;  // columns per lane of am_pair_kernel (2 measured slower)
    const bool grad = grad1 && grad2;
    const int col_blocks = pcc::ceil_div(n, 64 * q_cols), row_tiles = pcc::ceil_div(m, kPairRT);
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t cpart_off = up(L.total);
    const size_t part1_off = up(cpart_off + (size_t)b * col_blocks * row_tiles * 4);
    const size_t part2_off = up(part1_off + (grad ? (size_t)b * row_tiles * L.n4 * 3 * 4 : 0));
    const size_t total = up(part2_off + (grad ? (size_t)b * col_blocks * L.m4 * 3 * 4 : 0));
    StreamBuf ws(st);
    if (int rc = ws.alloc(total)) return rc;
    char *base = static_cast<char *>(ws.p);
    const size_t nm4 = (size_t)L.n4 + L.m4;
    // pair + finish kernels of the samples [s0, s0 + bc) on `lst` (every section of the workspace is indexed by sample)
    auto tail = [&](int s0, int bc, hipStream_t lst) -> int {
        const size_t o = (size_t)s0;
        PairArgs pa{};
        pa.n = n; pa.m = m; pa.n4 = L.n4; pa.m4 = L.m4;
        pa.soa1 = reinterpret_cast<const float *>(base + L.soa1) + o * 3 * L.n4;
        pa.soa2 = reinterpret_cast<const float *>(base + L.soa2) + o * 3 * L.m4;
        pa.lv = reinterpret_cast<const float *>(base + L.lv) + o * kLevels * nm4;
        pa.lc = make_levels();
        for (int i = 0; i < kLevels; i++) pa.cut2[i] = kZeroExp / -pa.lc.c[i];
        pa.cost_part = reinterpret_cast<float *>(base + cpart_off) + o * col_blocks * row_tiles;
        pa.part1 = grad ? reinterpret_cast<float *>(base + part1_off) + o * row_tiles * L.n4 * 3 : nullptr;
        pa.part2 = grad ? reinterpret_cast<float *>(base + part2_off) + o * col_blocks * L.m4 * 3 : nullptr;
        pa.col_blocks = col_blocks; pa.row_tiles = row_tiles; pa.bc = bc;
        pa.plain_order = pcc::tuning(PCC_TUNE_PAIR_PLAIN_ORDER);
        if ((long long)col_blocks * row_tiles * bc > 0x7fffffffLL) return pcc::invalid("match_cost: grid too large");
        const dim3 grid((unsigned)(col_blocks * row_tiles * bc));
        if (int rc = launch_pair<q_cols>(pa, grid, grad, lst)) return rc;
        FinishArgs f{};
        f.parts[0] = row_tiles; f.parts[1] = col_blocks; f.parts[2] = col_blocks * row_tiles;
        f.npts[0] = n; f.npts[1] = m; f.pitch[0] = L.n4; f.pitch[1] = L.m4;
        f.part[0] = pa.part1; f.part[1] = pa.part2; f.part[2] = pa.cost_part;
        f.perm[0] = reinterpret_cast<const int *>(base + L.perm1) + o * n;
        f.perm[1] = reinterpret_cast<const int *>(base + L.perm2) + o * m;
        f.scale = grad_cost ? grad_cost + o : nullptr;
        f.out[0] = grad ? grad1 + o * n * 3 : nullptr;
        f.out[1] = grad ? grad2 + o * m * 3 : nullptr;
        f.out[2] = cost + o;
        f.flags = reinterpret_cast<const int *>(base + L.live_cnt) + o * kLiveRow;  // (written by this call's sort)
        if (chamfer) {
            f.ch_d1 = chamfer->dist1 + o * n; f.ch_d2 = chamfer->dist2 + o * m; f.ch_loss = chamfer->loss + o;
            f.ch_n = n; f.ch_m = m; f.ch_mean = chamfer->mean;
        }
        {
            pcc::ProfScope prof("pair_finish_kernel", lst);
            const int blocks = grad ? pcc::ceil_div(std::max(n, m) * 3, 256) : 1;
            hipLaunchKernelGGL(pair_finish_kernel, dim3(blocks, bc, chamfer ? 4 : 3), dim3(256), 0, lst, f);
        }
        return pcc::check_launch("match_cost(reduce)");
    };
    // the Chamfer half of a ChamferEMD call: nearest neighbours on the clouds this call has just sorted, both directions
    // in one launch, then the loss reduction -- on the lane's stream, between the sort and the first pass
    auto nn_after_sort = [&](int s0, int bc, hipStream_t lst) -> int {
        const size_t o = (size_t)s0;
        NNSortedArgs q1{}, q2{};
        q1.n_q = n; q1.n_c = m; q1.q_n4 = L.n4; q1.c_n4 = L.m4; q1.q_nb = L.nb1; q1.c_nb = L.nb2;
        q1.groups = L.nb1; q1.batch = bc;
        q1.q_soa = reinterpret_cast<const float *>(base + L.soa1) + o * 3 * L.n4;
        q1.c_aos = reinterpret_cast<const float4 *>(base + L.aos2) + o * m;
        q1.q_box = reinterpret_cast<const float *>(base + L.box1) + o * L.nb1 * 8;
        q1.c_box = reinterpret_cast<const float *>(base + L.box2) + o * L.nb2 * 8;
        q1.q_perm = reinterpret_cast<const int *>(base + L.perm1) + o * n;
        q1.out_d = chamfer->dist1 + o * n; q1.out_i = chamfer->idx1 + o * n;
        q2.n_q = m; q2.n_c = n; q2.q_n4 = L.m4; q2.c_n4 = L.n4; q2.q_nb = L.nb2; q2.c_nb = L.nb1;
        q2.groups = L.nb2; q2.batch = bc;
        q2.q_soa = reinterpret_cast<const float *>(base + L.soa2) + o * 3 * L.m4;
        q2.c_aos = reinterpret_cast<const float4 *>(base + L.aos1) + o * n;
        q2.q_box = q1.c_box; q2.c_box = q1.q_box;
        q2.q_perm = reinterpret_cast<const int *>(base + L.perm2) + o * m;
        q2.out_d = chamfer->dist2 + o * m; q2.out_i = chamfer->idx2 + o * m;
        const long long w0 = (long long)bc * q1.groups, w1 = (long long)bc * q2.groups;
        const long long grid = (w0 + w1 + kNNWaves - 1) / kNNWaves;
        if (w0 + w1 > 0x7fffffffLL) return pcc::invalid("chamfer_emd: grid too large");
        {
            pcc::ProfScope prof("nn_sorted_kernel", lst);
            hipLaunchKernelGGL(nn_sorted_kernel, dim3((unsigned)grid), dim3(64 * kNNWaves), 0, lst, q1, q2, (int)w0);
        }
        return pcc::check_launch("chamfer_emd(nearest neighbours)");  // (the loss reduction rides in the finish launch)
    };
    if (chamfer) return run_levels(b, n, m, xyz1, xyz2, L, base, st, tail, nn_after_sort);
    return run_levels(b, n, m, xyz1, xyz2, L, base, st, tail);
}

int check_sizes(const char *who, int b, int n, int m) {
    if (b < 0 || n < 0 || m < 0) return pcc::invalid(who);
    if ((long long)n * 3 > 0x7fffffffLL || (long long)m * 3 > 0x7fffffffLL) return pcc::invalid(who);
    return PCC_OK;
}

}  // namespace

namespace pcc {
int match_cost_with_chamfer(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1,
                            float *grad2, hipStream_t st, const ChamferOut &chamfer) {
    return match_cost_implicit_impl(b, n, m, xyz1, xyz2, nullptr, cost, grad1, grad2, st, &chamfer);
}

// Hilbert sort of ONE channels-major cloud per sample (x[b][c][n], 1 <= c <= 3) for the k-NN graph (knn.hip): packed
// sorted rows (x, y, z, original index), the 16-point boxes and the sorted -> original permutation.
int sort_cloud_cmajor(int b, int c, int n, const float *x, float4 *aos, float *box16, int *perm, hipStream_t st) {
    SortArgs a{};
    int npad = 4 * kSortT;
    while (npad < n) npad <<= 1;
    if (npad > 64 * kSortT) npad = 0;  // > 16384 points: original order
    a.n[0] = n; a.npad[0] = npad; a.n4[0] = (n + 3) & ~3; a.nb[0] = pcc::ceil_div(n, kBox);
    a.xyz[0] = x; a.sstride[0] = (long long)c * n; a.pstride[0] = 1; a.cstride[0] = n; a.nch[0] = c;
    a.aos[0] = aos; a.box[0] = box16; a.perm[0] = perm;
    launch_sort(a, npad ? npad / kSortT : 4, dim3(b, 1), st);
    return pcc::check_launch("knn(sort)");
}
}  // namespace pcc

extern "C" {

size_t pcc_approxmatch_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return WsLayout(b, n, m).total;
}

int pcc_approxmatch_ws(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                       void *workspace, size_t workspace_bytes, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check_sizes("approxmatch: bad size", b, n, m)) return rc;
    if (b == 0 || n == 0 || m == 0) return PCC_OK;  // nothing to match (reference: empty loops)
    if (!xyz1 || !xyz2 || !match || !temp || !workspace) return pcc::invalid("approxmatch: null pointer");
    return approxmatch_impl(b, n, m, xyz1, xyz2, match, temp, workspace, workspace_bytes, nullptr,
                            static_cast<hipStream_t>(stream));
}

int pcc_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                    pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check_sizes("approxmatch: bad size", b, n, m)) return rc;
    if (b == 0 || n == 0 || m == 0) return PCC_OK;
    if (!xyz1 || !xyz2 || !match || !temp) return pcc::invalid("approxmatch: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    StreamBuf ws(st);
    const size_t bytes = WsLayout(b, n, m).total;
    if (int rc = ws.alloc(bytes)) return rc;
    return approxmatch_impl(b, n, m, xyz1, xyz2, match, temp, ws.p, bytes, nullptr, st);
}

int pcc_approxmatch_cost(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                         float *cost, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check_sizes("approxmatch_cost: bad size", b, n, m)) return rc;
    if (b == 0) return PCC_OK;
    if (!cost) return pcc::invalid("approxmatch_cost: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0 || m == 0) {
        hipError_t e = hipMemsetAsync(cost, 0, (size_t)b * sizeof(float), st);
        return e == hipSuccess ? PCC_OK : (pcc::set_error((int)e, "approxmatch_cost: memset failed"), (int)e);
    }
    if (!xyz1 || !xyz2 || !match || !temp) return pcc::invalid("approxmatch_cost: null pointer");
    StreamBuf ws(st);
    const size_t bytes = pcc_approxmatch_workspace_bytes(b, n, m);
    if (int rc = ws.alloc(bytes)) return rc;
    return approxmatch_impl(b, n, m, xyz1, xyz2, match, temp, ws.p, bytes, cost, st);
}

int pcc_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *grad_cost, float *cost,
                   float *grad1, float *grad2, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check_sizes("match_cost: bad size", b, n, m)) return rc;
    if (b == 0) return PCC_OK;
    if (!cost) return pcc::invalid("match_cost: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0 || m == 0) {  // empty sums
        hipError_t e = hipMemsetAsync(cost, 0, (size_t)b * sizeof(float), st);
        if (n && grad1 && e == hipSuccess) e = hipMemsetAsync(grad1, 0, (size_t)b * n * 3 * sizeof(float), st);
        if (m && grad2 && e == hipSuccess) e = hipMemsetAsync(grad2, 0, (size_t)b * m * 3 * sizeof(float), st);
        return e == hipSuccess ? PCC_OK : (pcc::set_error((int)e, "match_cost: memset failed"), (int)e);
    }
    if (!xyz1 || !xyz2) return pcc::invalid("match_cost: null pointer");
    if ((grad1 == nullptr) != (grad2 == nullptr)) return pcc::invalid("match_cost: grad1 and grad2 go together");
    return match_cost_implicit_impl(b, n, m, xyz1, xyz2, grad_cost, cost, grad1, grad2, st);
}

void approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                 pcc_stream_t stream) {
    (void)pcc_approxmatch(b, n, m, xyz1, xyz2, match, temp, stream);
}

int pcc_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out,
                  pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check_sizes("matchcost: bad size", b, n, m)) return rc;
    if (b == 0) return PCC_OK;
    if (!out) return pcc::invalid("matchcost: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0 || m == 0) {  // empty sums
        hipError_t e = hipMemsetAsync(out, 0, (size_t)b * sizeof(float), st);
        return e == hipSuccess ? PCC_OK : (pcc::set_error((int)e, "matchcost: memset failed"), (int)e);
    }
    if (!xyz1 || !xyz2 || !match) return pcc::invalid("matchcost: null pointer");
    const int tiles = pcc::ceil_div(m, kRowRT);
    StreamBuf ws(st);
    if (int rc = ws.alloc((size_t)b * tiles * sizeof(float))) return rc;
    float *part = static_cast<float *>(ws.p);
    const bool vec = (n % 4 == 0) && aligned16(match);
    {
        pcc::ProfScope prof("am_row_kernel<cost>", st);
        if (vec) hipLaunchKernelGGL((am_row_kernel<0, true, 2048>), dim3(tiles, b), dim3(256), 0, st, n, m, xyz1, xyz2, match, part);
        else hipLaunchKernelGGL((am_row_kernel<0, false, 2048>), dim3(tiles, b), dim3(256), 0, st, n, m, xyz1, xyz2, match, part);
    }
    if (int rc = pcc::check_launch("matchcost")) return rc;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(b), dim3(256), 0, st, tiles, part, out);
    return pcc::check_launch("matchcost(reduce)");
}

void matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *out,
               pcc_stream_t stream) {
    (void)pcc_matchcost(b, n, m, xyz1, xyz2, match, out, stream);
}

int pcc_matchcostgrad_scaled(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                             const float *grad_cost, float *grad1, float *grad2, pcc_stream_t stream) {
    pcc::clear_error();
    if (int rc = check_sizes("matchcostgrad: bad size", b, n, m)) return rc;
    if (b == 0) return PCC_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0 || m == 0) {
        hipError_t e = hipSuccess;
        if (n && grad1) e = hipMemsetAsync(grad1, 0, (size_t)b * n * 3 * sizeof(float), st);
        if (m && grad2 && e == hipSuccess) e = hipMemsetAsync(grad2, 0, (size_t)b * m * 3 * sizeof(float), st);
        return e == hipSuccess ? PCC_OK : (pcc::set_error((int)e, "matchcostgrad: memset failed"), (int)e);
    }
    if (!xyz1 || !xyz2 || !match || !grad1 || !grad2) return pcc::invalid("matchcostgrad: null pointer");
    const bool vec = (n % 4 == 0) && aligned16(match);
    const int row_tiles = pcc::ceil_div(m, kGradRT), slabs = pcc::ceil_div(n, kGradSlab);
    StreamBuf ws(st);
    const size_t p1_elems = (size_t)b * row_tiles * n * 3, p2_elems = slabs > 1 ? (size_t)b * slabs * m * 3 : 0;
    if (int rc = ws.alloc((p1_elems + p2_elems) * sizeof(float))) return rc;
    float *part1 = static_cast<float *>(ws.p);
    float *part2 = slabs > 1 ? part1 + p1_elems : grad2;  // a single slab writes grad2 directly
    {
        pcc::ProfScope prof("am_grad_fused_kernel", st);
        const dim3 grid(row_tiles, slabs, b);
        const float *sc2 = slabs > 1 ? nullptr : grad_cost;
        if (vec) hipLaunchKernelGGL((am_grad_fused_kernel<true>), grid, dim3(256), 0, st, n, m, row_tiles, xyz1, xyz2, match, part1, part2, sc2);
        else hipLaunchKernelGGL((am_grad_fused_kernel<false>), grid, dim3(256), 0, st, n, m, row_tiles, xyz1, xyz2, match, part1, part2, sc2);
    }
    if (int rc = pcc::check_launch("matchcostgrad(fused)")) return rc;
    const size_t per1 = (size_t)n * 3, per2 = (size_t)m * 3;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((per1 + 255) / 256), b), dim3(256), 0, st, row_tiles, per1, part1, grad_cost, grad1);
    if (slabs > 1)
        hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((per2 + 255) / 256), b), dim3(256), 0, st, slabs, per2, part2, grad_cost, grad2);
    return pcc::check_launch("matchcostgrad(reduce)");
}

int pcc_matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *grad1,
                      float *grad2, pcc_stream_t stream) {
    return pcc_matchcostgrad_scaled(b, n, m, xyz1, xyz2, match, nullptr, grad1, grad2, stream);
}

void matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *grad1,
                   float *grad2, pcc_stream_t stream) {
    (void)pcc_matchcostgrad(b, n, m, xyz1, xyz2, match, grad1, grad2, stream);
}

}  // extern "C"
