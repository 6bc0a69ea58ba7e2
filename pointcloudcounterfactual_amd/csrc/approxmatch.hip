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
//   matchcost / grad re-read match 3x            | cost: one read; grads: one read each, deterministic
//                                                | two-stage reductions (no float atomics)
//
// Rooflines: the 19 phase launches and the materialise pass are f32-VALU/transcendental
// bound (10 / 14 / 60 issue slots per pair); matchcost and the two gradient kernels are HBM bound
// (one read of match each).
#include "pcc_common.hpp"

namespace {

using pcc::sq3;

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

struct PhaseArgs {
    int n_own, n_cand, tiles;          // tiles = ceil(n_own / (64 R))
    const float *own_xyz, *cand_xyz;   // [b, n_own, 3], [b, n_cand, 3]
    const float *w0, *w1;              // per-candidate weights (w0 may be null => w0c)
    long long w0_stride, w1_stride;    // per-sample strides in floats
    float w0c;
    float c0, c1;
    int first;                         // first level: remain* still hold their initial constants
    float multiL, multiR;
    // epilogue operands, all indexed [sample * stride + owner]
    float *remain;                     // remainL (CA/C) or remainR (B)
    long long remain_stride;
    const float *ratio_in;             // CA/C: ratioL_i
    float *ratio_out;                  // A: ratioL_0 ; B: ratioR_i ; CA: ratioL_{i+1}
    long long ratio_stride;            // per-sample stride of the level arrays
};

template <int MODE, int R, int S, int CH>
__global__ __launch_bounds__(64 * S) void am_phase_kernel(PhaseArgs a) {
    constexpr int T = 64 * S;
    constexpr int TQ = 64 * R;
    constexpr int NW = (MODE == PH_CA) ? 2 : 1;
    constexpr bool W0_CONST = (MODE == PH_A);
    __shared__ __attribute__((aligned(16))) float lds_c[(3 + NW) * CH];  // x | y | z | w0 | (w1)
    __shared__ float red[NW][S][TQ];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int smp = blockIdx.x / a.tiles;
    const int tile = blockIdx.x - smp * a.tiles;
    const float *O = a.own_xyz + (size_t)smp * a.n_own * 3;
    const float *C = a.cand_xyz + (size_t)smp * a.n_cand * 3;
    const float *W0 = W0_CONST ? nullptr : a.w0 + (size_t)smp * a.w0_stride;
    const float *W1 = (NW == 2) ? a.w1 + (size_t)smp * a.w1_stride : nullptr;

    float ox[R], oy[R], oz[R], s0[R], s1[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        int o = tile * TQ + r * 64 + lane;
        o = o < a.n_own ? o : a.n_own - 1;
        ox[r] = O[o * 3 + 0];
        oy[r] = O[o * 3 + 1];
        oz[r] = O[o * 3 + 2];
        s0[r] = 0.f;
        s1[r] = 0.f;
    }
    const float4 *X4 = reinterpret_cast<const float4 *>(lds_c);
    const float4 *Y4 = X4 + CH / 4;
    const float4 *Z4 = Y4 + CH / 4;
    const float4 *A4 = Z4 + CH / 4;
    const float4 *B4 = A4 + CH / 4;
    const float c0 = a.c0, c1 = a.c1;

    for (int q0 = 0; q0 < a.n_cand; q0 += CH) {
        const int cnt = min(CH, a.n_cand - q0);
        const int ngroups = (cnt + 3) / 4;
        if (q0) __syncthreads();
        const float *src = C + (size_t)q0 * 3;
        for (int i = tid; i < cnt * 3; i += T) {
            const float v = src[i];
            const int p = i / 3;
            lds_c[(i - p * 3) * CH + p] = v;
        }
        for (int i = tid; i < ngroups * 4; i += T) {
            const bool ok = i < cnt;
            if (!ok) {
                lds_c[i] = 0.f;
                lds_c[CH + i] = 0.f;
                lds_c[2 * CH + i] = 0.f;
            }
            lds_c[3 * CH + i] = ok ? (W0_CONST ? a.w0c : W0[q0 + i]) : 0.f;  // padded candidates weigh 0
            if (NW == 2) lds_c[4 * CH + i] = ok ? W1[q0 + i] : 0.f;
        }
        __syncthreads();
        const int gs = (ngroups + S - 1) / S;
        const int g_begin = w * gs;
        const int g_end = min(g_begin + gs, ngroups);
        for (int g = g_begin; g < g_end; g++) {
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
#pragma unroll
    for (int r = 0; r < R; r++) {
        red[0][w][r * 64 + lane] = s0[r];
        if (NW == 2) red[1][w][r * 64 + lane] = s1[r];
    }
    __syncthreads();
    for (int e = tid; e < TQ; e += T) {
        const int o = tile * TQ + e;
        if (o >= a.n_own) continue;
        float t0 = red[0][0][e], t1 = 0.f;
        if (NW == 2) t1 = red[1][0][e];
#pragma unroll
        for (int s = 1; s < S; s++) {
            t0 += red[0][s][e];
            if (NW == 2) t1 += red[1][s][e];
        }
        if (MODE == PH_A) {
            // ratioL[k] = remainL[k] / (1e-9 + sum)            approxmatch.cu:37,61 (remainL == multiL)
            a.ratio_out[(size_t)smp * a.ratio_stride + o] = a.multiL / (1e-9f + t0);
        } else if (MODE == PH_B) {
            // approxmatch.cu:106-109
            float *rem = a.remain + (size_t)smp * a.remain_stride + o;
            const float rR = a.first ? a.multiR : *rem;
            const float sumr = t0 * rR;
            const float consumption = __builtin_fminf(rR / (sumr + 1e-9f), 1.0f);
            a.ratio_out[(size_t)smp * a.ratio_stride + o] = consumption * rR;
            *rem = __builtin_fmaxf(0.0f, rR - sumr);
        } else {
            // pass C: suml = sum_l e*ratioL[k]*ratioR[l] ; remainL = max(0, remainL - suml)   :154-162
            float *rem = a.remain + (size_t)smp * a.remain_stride + o;
            const float rl = a.ratio_in[(size_t)smp * a.ratio_stride + o];
            const float rL = a.first ? a.multiL : *rem;
            const float left = __builtin_fmaxf(0.0f, rL - rl * t0);
            *rem = left;
            // pass A of the next level: ratioL' = remainL / (1e-9 + sum_l e'*remainR[l])       :37,61
            if (MODE == PH_CA) a.ratio_out[(size_t)smp * a.ratio_stride + o] = left / (1e-9f + t1);
        }
    }
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
                                                              float *__restrict__ match, float *__restrict__ temp,
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
    // The first tile of each sample also publishes the last level's ratios into temp
    // (temp = remainL | remainR | ratioL | ratioR, approxmatch.cu:4).
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        float *tb = temp + (size_t)smp * 2 * (n + m);
        const float *last = lvb + (size_t)(kLevels - 1) * (n + m);
        for (int i = tid; i < n; i += 256) tb[n + m + i] = last[i];
        for (int i = tid; i < m; i += 256) tb[n + m + n + i] = last[n + i];
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
        float out[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float d = sq3(A.x - x1[q], A.y - y1[q], A.z - z1[q]);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < kLevels; i++) {
                const float wgt = (fast_exp2(lc.c[i] * d) * rl[i][q]) * rr[i];
                acc += wgt;
            }
            out[q] = acc;
            if (COST && k0 + q < n) csum = __builtin_fmaf(acc, __builtin_amdgcn_sqrtf(d), csum);
        }
        float *row = match + ((size_t)smp * m + (l0 + li)) * n;
        if (VEC) {
            if (k0 + 3 < n) {
                *reinterpret_cast<float4 *>(row + k0) = make_float4(out[0], out[1], out[2], out[3]);
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
                    const float4 t = *reinterpret_cast<const float4 *>(mrow + k);
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

// grad1[b][i] = sum_s part[b][s][i]  (i over n*3), fixed order.
__global__ __launch_bounds__(256) void reduce_splits_kernel(int rs, size_t per_sample, const float *__restrict__ part,
                                                             float *__restrict__ out) {
    const int smp = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_sample) return;
    const float *p = part + (size_t)smp * rs * per_sample + i;
    float s = p[0];
    for (int t = 1; t < rs; t++) s += p[t * per_sample];
    out[(size_t)smp * per_sample + i] = s;
}

// ---- host side -------------------------------------------------------------------------------------
constexpr int kPhCH = 2048;

// Workgroup shape: R owners per lane x S candidate slices (waves).  One wave can issue a VALU
// instruction only every 4 cycles while a SIMD retires one every 2, so a phase needs >= 2 (better 4)
// waves per SIMD = 2048-4096 waves on 256 CUs; R is spent only once the chip is full (each LDS
// broadcast read is then amortised over R owners).
static int phase_cfg_override() {
    static const int v = [] {
        const char *e = std::getenv("PCC_AM_CFG");
        return e ? std::atoi(e) : 0;
    }();
    return v;
}

template <int MODE, int R, int S>
int launch_phase_rs(PhaseArgs a, int b, hipStream_t st, const char *what) {
    a.tiles = pcc::ceil_div(a.n_own, 64 * R);
    const long long grid = (long long)b * a.tiles;
    if (grid > 0x7fffffffLL) return pcc::invalid("approxmatch: grid too large");
    {
        pcc::ProfScope prof(MODE == PH_CA ? "am_phase_kernel<CA>" : MODE == PH_B ? "am_phase_kernel<B>"
                            : MODE == PH_A ? "am_phase_kernel<A>" : "am_phase_kernel<C>", st);
        hipLaunchKernelGGL((am_phase_kernel<MODE, R, S, kPhCH>), dim3((unsigned)grid), dim3(64 * S), 0, st, a);
    }
    return pcc::check_launch(what);
}

template <int MODE>
int launch_phase(const PhaseArgs &a, int b, hipStream_t st, const char *what) {
    int cfg = phase_cfg_override();
    if (cfg == 0) {
        const long long owners = (long long)b * a.n_own;
        // waves = owners / (64 R) * S ; aim for >= 4096
        if (owners >= 4LL * 65536) cfg = 44;
        else if (owners >= 2LL * 65536) cfg = 48;
        else if (owners >= 65536) cfg = 28;
        else cfg = 18;
    }
    switch (cfg) {
    case 44: return launch_phase_rs<MODE, 4, 4>(a, b, st, what);
    case 48: return launch_phase_rs<MODE, 4, 8>(a, b, st, what);
    case 24: return launch_phase_rs<MODE, 2, 4>(a, b, st, what);
    case 28: return launch_phase_rs<MODE, 2, 8>(a, b, st, what);
    case 14: return launch_phase_rs<MODE, 1, 4>(a, b, st, what);
    case 216: return launch_phase_rs<MODE, 2, 16>(a, b, st, what);
    default: return launch_phase_rs<MODE, 1, 8>(a, b, st, what);
    }
}

struct StreamBuf {  // stream-ordered scratch (hipMallocAsync / hipFreeAsync)
    void *p = nullptr;
    hipStream_t st;
    explicit StreamBuf(hipStream_t s) : st(s) {}
    int alloc(size_t bytes) {
        static bool pool_tuned = [] {
            int dev = 0;
            hipMemPool_t pool;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
                uint64_t keep = ~0ull;  // keep freed blocks cached in the pool between calls
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            }
            return true;
        }();
        (void)pool_tuned;
        hipError_t e = hipMallocAsync(&p, bytes, st);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            pcc::set_error(PCC_ENOMEM, "workspace hipMallocAsync failed");
            return PCC_ENOMEM;
        }
        return PCC_OK;
    }
    ~StreamBuf() {
        if (p) (void)hipFreeAsync(p, st);
    }
};

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

size_t levels_bytes(int b, int n, int m) { return (size_t)b * kLevels * ((size_t)n + m) * sizeof(float); }
size_t cost_parts(int n, int m) { return (size_t)pcc::ceil_div(n, kMatKT) * pcc::ceil_div(m, kMatLT); }

int approxmatch_impl(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                     void *workspace, size_t workspace_bytes, float *cost_out, hipStream_t st) {
    const size_t need = levels_bytes(b, n, m) + (cost_out ? (size_t)b * cost_parts(n, m) * sizeof(float) : 0);
    if (workspace_bytes < need) return pcc::invalid("approxmatch: workspace too small");
    float *lv = static_cast<float *>(workspace);
    float *cpart = lv + (size_t)b * kLevels * ((size_t)n + m);
    const LevelConsts lc = make_levels();
    float multiL, multiR;  // approxmatch.cu:6-12 (integer division)
    if (n >= m) { multiL = 1; multiR = (float)(n / m); }
    else { multiL = (float)(m / n); multiR = 1; }
    const long long nm = (long long)n + m;

    PhaseArgs a{};
    a.multiL = multiL; a.multiR = multiR;
    // pass A of the first level
    a.n_own = n; a.n_cand = m;
    a.own_xyz = xyz1; a.cand_xyz = xyz2;
    a.w0 = nullptr; a.w0c = multiR; a.c0 = lc.c[0]; a.first = 1;
    a.ratio_out = lv; a.ratio_stride = kLevels * nm;
    int rc = launch_phase<PH_A>(a, b, st, "approxmatch(A)");
    if (rc) return rc;
    for (int i = 0; i < kLevels; i++) {
        float *ratioL = lv + (size_t)i * nm, *ratioR = ratioL + n;
        PhaseArgs pb{};
        pb.multiL = multiL; pb.multiR = multiR; pb.first = (i == 0);
        pb.n_own = m; pb.n_cand = n;
        pb.own_xyz = xyz2; pb.cand_xyz = xyz1;
        pb.w0 = ratioL; pb.w0_stride = kLevels * nm; pb.c0 = lc.c[i];
        pb.remain = temp + n; pb.remain_stride = 2 * nm;
        pb.ratio_out = ratioR; pb.ratio_stride = kLevels * nm;
        rc = launch_phase<PH_B>(pb, b, st, "approxmatch(B)");
        if (rc) return rc;
        PhaseArgs pc{};
        pc.multiL = multiL; pc.multiR = multiR; pc.first = (i == 0);
        pc.n_own = n; pc.n_cand = m;
        pc.own_xyz = xyz1; pc.cand_xyz = xyz2;
        pc.w0 = ratioR; pc.w0_stride = kLevels * nm; pc.c0 = lc.c[i];
        pc.w1 = temp + n; pc.w1_stride = 2 * nm;
        pc.remain = temp; pc.remain_stride = 2 * nm;
        pc.ratio_in = ratioL; pc.ratio_stride = kLevels * nm;
        if (i + 1 < kLevels) {
            pc.c1 = lc.c[i + 1];
            pc.ratio_out = lv + (size_t)(i + 1) * nm;
            rc = launch_phase<PH_CA>(pc, b, st, "approxmatch(CA)");
        } else {
            rc = launch_phase<PH_C>(pc, b, st, "approxmatch(C)");
        }
        if (rc) return rc;
    }
    const dim3 grid(pcc::ceil_div(n, kMatKT), pcc::ceil_div(m, kMatLT), b);
    const bool vec = (n % 4 == 0) && aligned16(match);
    if (cost_out) {
        {
            pcc::ProfScope prof("am_materialise_kernel<cost>", st);
            if (vec) hipLaunchKernelGGL((am_materialise_kernel<true, true>), grid, dim3(256), 0, st, n, m, xyz1, xyz2, lv, lc, match, temp, cpart);
            else hipLaunchKernelGGL((am_materialise_kernel<true, false>), grid, dim3(256), 0, st, n, m, xyz1, xyz2, lv, lc, match, temp, cpart);
        }
        rc = pcc::check_launch("approxmatch(materialise+cost)");
        if (rc) return rc;
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(b), dim3(256), 0, st, (int)cost_parts(n, m), cpart, cost_out);
        return pcc::check_launch("approxmatch(cost reduce)");
    }
    {
        pcc::ProfScope prof("am_materialise_kernel", st);
        if (vec) hipLaunchKernelGGL((am_materialise_kernel<false, true>), grid, dim3(256), 0, st, n, m, xyz1, xyz2, lv, lc, match, temp, nullptr);
        else hipLaunchKernelGGL((am_materialise_kernel<false, false>), grid, dim3(256), 0, st, n, m, xyz1, xyz2, lv, lc, match, temp, nullptr);
    }
    return pcc::check_launch("approxmatch(materialise)");
}

int check_sizes(const char *who, int b, int n, int m) {
    if (b < 0 || n < 0 || m < 0) return pcc::invalid(who);
    if ((long long)n * 3 > 0x7fffffffLL || (long long)m * 3 > 0x7fffffffLL) return pcc::invalid(who);
    return PCC_OK;
}

}  // namespace

extern "C" {

size_t pcc_approxmatch_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return levels_bytes(b, n, m) + (size_t)b * cost_parts(n, m) * sizeof(float);
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
    const size_t bytes = levels_bytes(b, n, m);
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

int pcc_matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *grad1,
                      float *grad2, pcc_stream_t stream) {
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
    // grad2: row kernel
    const int tiles = pcc::ceil_div(m, kRowRT);
    {
        pcc::ProfScope prof("am_row_kernel<grad2>", st);
        if (vec) hipLaunchKernelGGL((am_row_kernel<1, true, 2048>), dim3(tiles, b), dim3(256), 0, st, n, m, xyz1, xyz2, match, grad2);
        else hipLaunchKernelGGL((am_row_kernel<1, false, 2048>), dim3(tiles, b), dim3(256), 0, st, n, m, xyz1, xyz2, match, grad2);
    }
    if (int rc = pcc::check_launch("matchcostgrad(grad2)")) return rc;
    // grad1: column kernel with RS row splits, then ordered sum of the partials
    const int ctiles = pcc::ceil_div(n, 256);
    int rs = pcc::ceil_div(2048, ctiles * b);
    rs = std::max(1, std::min(rs, std::min(64, pcc::ceil_div(m, 16))));
    StreamBuf ws(st);
    if (int rc = ws.alloc((size_t)b * rs * n * 3 * sizeof(float))) return rc;
    float *part = static_cast<float *>(ws.p);
    {
        pcc::ProfScope prof("am_col_kernel<grad1>", st);
        if (vec) hipLaunchKernelGGL((am_col_kernel<true>), dim3(ctiles, rs, b), dim3(256), 0, st, n, m, rs, xyz1, xyz2, match, part);
        else hipLaunchKernelGGL((am_col_kernel<false>), dim3(ctiles, rs, b), dim3(256), 0, st, n, m, rs, xyz1, xyz2, match, part);
    }
    if (int rc = pcc::check_launch("matchcostgrad(grad1)")) return rc;
    const size_t per = (size_t)n * 3;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((per + 255) / 256), b), dim3(256), 0, st, rs, per, part, grad1);
    return pcc::check_launch("matchcostgrad(reduce)");
}

void matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *grad1,
                   float *grad2, pcc_stream_t stream) {
    (void)pcc_matchcostgrad(b, n, m, xyz1, xyz2, match, grad1, grad2, stream);
}

}  // extern "C"
