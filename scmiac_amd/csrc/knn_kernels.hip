// Fused pairwise-distance + top-k search for gfx950 (MI355X), and the exact
// refine / direct-evaluation kernels that make its result bit-identical to a
// float32 direct-difference search.
//
// Pipeline of one scmiac_knn_search call (all on one stream, no host sync):
//   locality       (large sets) both sets regrouped by nearest pivot, the cells laid out along a
//                  nearest-neighbour chain over the pivots: permutations, and per query block the
//                  stages of the reference image round its queries' own cells ("home")
//   pack_rows      Q, R  -> MFMA-friendly padded copies (in cell order), |r|^2, max |r|^2
//   seed_tau       (merge mode) thresholds to start from, out of the running k-th distances
//   fused_knn      seed pass over the home rows (group minima -> a starting threshold per query),
//                  then for every query the 64 references with the smallest
//                  s(q,r) = |r|^2 - 2 q.r   (fp32 MFMA 32x32x2, refs staged in LDS,
//                  running threshold in a register, per-lane candidate buffers in
//                  global memory (L2) or LDS, wave-wide bitonic merges) -- the N x M
//                  matrix never leaves the CU
//   refine         exact D(q,r) for the 64*S shortlisted references, (D, index) sort,
//                  top-k, and a rounding-error certificate that nothing outside the
//                  shortlist can belong to the top-k
//   exact_rows     rows whose certificate failed are searched by direct evaluation
// The fused kernel itself lives in fused_knn.h and is instantiated by knn_fused_0..3.hip
// (four translation units, so that its many variants compile side by side); knn_fused_4.hip holds
// the pruned product kernels (blocks skip the cells out of their queries' reach) and
// knn_fused_5.hip the stream-sharing kernels of fused_knn_share.h (single-round plans: blocks out
// of work join unfinished streams).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "kernels.h"
#include "fused_common.h"
#include "fused_variants.h"

namespace scmiac {


// --------------------------------------------------------------------------------
// pack_rows: X[n][d] -> Xp[n_pad][ld] with lane-half-major features
//   Xp[row][h*D2P + s] = scale * X[row][h*D2 + s]   (0 where h*D2+s >= d or s >= D2)
// so that lane half h of an MFMA 32x32x2 step s reads feature (h, s) with one aligned
// 16-byte LDS read per four steps; columns 2*D2P..ld-1 are zero (a block of rows is then
// one contiguous image of an LDS stage).  norms[row] = fmaf chain of x_i^2 (feature
// order), 1.7e38 on the padding rows (they can then never be candidates).
// --------------------------------------------------------------------------------
// One block packs PACK_ROWS consecutive rows: the rows are read as one contiguous, coalesced
// range into LDS, then written out column by column in the packed order (coalesced again);
// one thread per row runs the norm's fmaf chain out of LDS.  HBM-bound: n*d*4 bytes in,
// n*ld*4 out.
static constexpr int PACK_ROWS = 64;
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ X, long long n, int d, int D2,
                                                        int D2P, float scale, float* __restrict__ Xp, int ld,
                                                        float* __restrict__ norms, long long n_pad,
                                                        unsigned* __restrict__ maxnorm_bits,
                                                        const int* __restrict__ perm /* packed row -> row of X, or NULL */) {
    __shared__ float rows[PACK_ROWS * SCMIAC_KNN_MAX_D_INTERNAL];
    const long long row0 = (long long)blockIdx.x * PACK_ROWS;
    if (row0 >= n_pad) return;
    const int nrows = (int)((n_pad - row0) < PACK_ROWS ? (n_pad - row0) : PACK_ROWS);
    const long long real = n - row0;  // rows of this block that exist in X (may be <= 0)
    const int nreal = real < 0 ? 0 : (real < nrows ? (int)real : nrows);
    if (perm) {  // gather: whole rows (d contiguous floats each) from where the permutation says
        for (int i = threadIdx.x; i < nreal * d; i += blockDim.x) {
            const int r = i / d;
            rows[i] = X[(long long)perm[row0 + r] * d + (i - r * d)];
        }
    } else {
        const float* src = X + row0 * d;
        for (int i = threadIdx.x; i < nreal * d; i += blockDim.x) rows[i] = src[i];
    }
    __syncthreads();
    float* dst = Xp + row0 * ld;
    for (int i = threadIdx.x; i < nrows * ld; i += blockDim.x) {
        const int r = i / ld, c = i - r * ld;
        const int h = c / D2P, s = c - h * D2P;
        const int f = h * D2 + s;
        float v = 0.0f;
        if (r < nreal && h < 2 && s < D2 && f < d) v = rows[r * d + f] * scale;
        dst[i] = v;
    }
    if (norms && (int)threadIdx.x < nrows) {
        const int r = threadIdx.x;
        if (r >= nreal) {
            norms[row0 + r] = __uint_as_float(0x7f000000u);  // finite, above every real score
        } else {
            // rows[] is read with a stride of d floats: conflict-free for odd d, a few-way
            // conflict otherwise -- irrelevant next to the HBM traffic
            float acc = 0.0f;
            for (int i = 0; i < d; ++i) acc = __fmaf_rn(rows[r * d + i], rows[r * d + i], acc);
            norms[row0 + r] = acc;
            atomicMax(maxnorm_bits, __float_as_uint(acc));  // acc >= 0: bit order == value order
        }
    }
}

// --------------------------------------------------------------------------------
// Locality partition.  The candidate path of the fused kernel is cheap once a query's
// threshold is tight, and it becomes tight as soon as the query has seen its near
// neighbours.  So both sets are regrouped by nearest pivot (C reference rows taken at even
// spacing): queries of a block then share a cell, and the block streams the reference rows
// of that cell and of its neighbours in the chain order -- its "home" stages -- before the rest.  Any assignment gives a correct
// search (the result is put back in the caller's row order and row numbers); a good one
// makes it fast.  Kernels: pivots -> cell of every row + per-block cell counts -> offsets
// -> stable scatter into a permutation -> home stages per query block.
// --------------------------------------------------------------------------------
static constexpr int CELL_NT = 256;       // threads per block of the assignment kernel
static constexpr int CELL_TP = 8;         // pivots scored together per thread
static constexpr int CELL_PT = 64;        // pivots per LDS tile

// rows per block of the partition kernels: as many as leave the block's rows (feature-major) and
// one pivot tile within the LDS
static int cell_rows_per_block(int d) { return d <= 64 ? 256 : (d <= 128 ? 128 : 64); }

// Pruned search: pivots by farthest-point traversal over a sample of the reference rows (each
// new pivot is the sample row farthest from all pivots so far), so that every cluster of the data
// gets pivots of its own and no cell has to take in the rows of a cluster without one: the cells'
// radii, which the pruning bound subtracts, stay at the size of a cluster.  One block; sample row
// s is row s * m / FPS_SAMPLE of R.
static constexpr int FPS_SAMPLE = 4096, FPS_NT = 1024, FPS_PER = FPS_SAMPLE / FPS_NT;
int knn_fps_sample_rows() { return FPS_SAMPLE; }
// The traversal reads every sample row once per pivot, so the sample is first copied feature-major
// (sampleT[i][s] = feature i of sample row s): the threads of a wave then read consecutive
// addresses where a row-major sample would cost them 64 cache lines per load.
__global__ __launch_bounds__(FPS_NT) void fps_pivots_kernel(const float* __restrict__ R, long long m, int d, int C,
                                                            int* __restrict__ piv_rows, float* __restrict__ sampleT) {
    __shared__ float pv[SCMIAC_KNN_MAX_D_INTERNAL];
    __shared__ float red_v[FPS_NT / 64];
    __shared__ int red_i[FPS_NT / 64];
    __shared__ int cur;
    const int tid = threadIdx.x;
    const long long S = m < FPS_SAMPLE ? m : FPS_SAMPLE;
    for (long long e = tid; e < S * d; e += FPS_NT) {
        const long long s = e / d;
        const int i = (int)(e - s * d);
        sampleT[(long long)i * FPS_SAMPLE + s] = R[(s * m / S) * d + i];
    }
    float mind[FPS_PER];
#pragma unroll
    for (int u = 0; u < FPS_PER; ++u) mind[u] = __builtin_inff();
    if (tid == 0) { cur = 0; piv_rows[0] = 0; }
    __syncthreads();  // (the block's own global writes above are visible to it after the barrier)
    for (int c = 1; c <= C; ++c) {
        const int pcur = cur;
        for (int i = tid; i < d; i += FPS_NT) pv[i] = sampleT[(long long)i * FPS_SAMPLE + pcur];
        __syncthreads();
        float acc[FPS_PER];
#pragma unroll
        for (int u = 0; u < FPS_PER; ++u) acc[u] = 0.0f;
        for (int i = 0; i < d; ++i) {
            const float p = pv[i];
            const float* col = sampleT + (long long)i * FPS_SAMPLE + tid;
#pragma unroll
            for (int u = 0; u < FPS_PER; ++u) {
                const float t = col[u * FPS_NT] - p;  // (sample rows past S are never looked at below)
                acc[u] = __fmaf_rn(t, t, acc[u]);
            }
        }
        float best = -1.0f;
        int besti = 0;
#pragma unroll
        for (int u = 0; u < FPS_PER; ++u) {
            const long long sidx = (long long)u * FPS_NT + tid;
            if (sidx < S) {
                mind[u] = fminf(mind[u], acc[u] == acc[u] ? acc[u] : 0.0f);  // (NaN rows never become pivots)
                if (mind[u] > best) { best = mind[u]; besti = (int)sidx; }
            }
        }
        for (int off = 32; off; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(besti, off);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_i[tid >> 6] = besti; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < FPS_NT / 64; ++w)
                if (red_v[w] > best || (red_v[w] == best && red_i[w] < besti)) { best = red_v[w]; besti = red_i[w]; }
            cur = besti;
            if (c < C) piv_rows[c] = (int)((long long)besti * m / S);
        }
        __syncthreads();
    }
}

// Order of the cells in the packed images: a greedy nearest-neighbour chain over the pivots (start
// at pivot 0, go on to the nearest pivot not yet visited), so that the cells of one cluster of the
// data -- whose pivots are closer to each other than to any other cluster's -- follow each other.
// A query block's home window, a run of consecutive packed rows round its own cells, then holds its
// cluster-mates in the neighbouring cells too.  The chain itself is serial, so it runs over at most
// CHAIN_MAX "lead" pivots (every stride-th one), one wave, their mutual distances in LDS (half
// precision is plenty: the order is a heuristic, any order gives the same table): a few hundred
// cycles per step.  Every other pivot follows the lead it is nearest to.
static constexpr int CHAIN_MAX = 256;
__global__ __launch_bounds__(256) void pivot_dist_kernel(const float* __restrict__ pivT, int d, int C, int Cpad, int stride,
                                                         int S, _Float16* __restrict__ D /* [S][S] */,
                                                         int* __restrict__ near /* [C]: nearest lead of every pivot */) {
    __shared__ float mine[SCMIAC_KNN_MAX_D_INTERNAL];
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < d; i += blockDim.x) mine[i] = pivT[(long long)i * Cpad + c];
    __syncthreads();
    const bool lead = (c % stride == 0) && (c / stride < S);
    float best = 3.0e38f;
    int besti = 0x7fffffff;
    if (tid < S) {  // S <= 256 = blockDim
        const int t = tid * stride;
        float acc = 0.0f;
        for (int i = 0; i < d; ++i) { const float v = pivT[(long long)i * Cpad + t] - mine[i]; acc = __fmaf_rn(v, v, acc); }
        float dist = sqrtf(acc);
        if (!(dist < 60000.0f)) dist = 60000.0f;  // (half precision's range; NaN pivots come last)
        if (lead) D[(long long)(c / stride) * S + tid] = (_Float16)dist;
        best = dist; besti = tid;
    }
    for (int off = 32; off; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (ov < best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_i[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (red_v[w] < best || (red_v[w] == best && red_i[w] < besti)) { best = red_v[w]; besti = red_i[w]; }
        near[c] = lead ? c / stride : besti;
    }
}

// chain over the S leads (one wave); chain[pos] = lead
__global__ __launch_bounds__(64) void chain_order_kernel(const _Float16* __restrict__ D, int S, int* __restrict__ chain) {
    extern __shared__ __attribute__((aligned(16))) _Float16 dm[];  // [S][S]
    const int lane = threadIdx.x;
    for (int i = lane; i < S * S; i += 64) dm[i] = D[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // lane l owns leads l, l + 64, ...: bit b of `seen` = lead l + 64 b visited
    unsigned seen = 0u;
    int cur = 0;
    if (lane == 0) { seen = 1u; chain[0] = 0; }
    for (int step = 1; step < S; ++step) {
        const _Float16* row = dm + cur * S;
        float best = 3.0e38f;
        int besti = 0x7fffffff;
        for (int b = 0, c = lane; c < S; ++b, c += 64) {
            const float v = (float)row[c];
            if (!((seen >> b) & 1u) && v < best) { best = v; besti = c; }   // (ascending c: the lowest number among equals)
        }
        for (int off = 32; off; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(besti, off);
            if (ov < best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        cur = besti;  // (step < S: some lead is unvisited, so besti is one)
        if ((cur & 63) == lane) seen |= 1u << (cur >> 6);
        if (lane == 0) chain[step] = cur;
    }
}

// order[pos] = pivot: the leads in chain order, each followed by the pivots nearest to it (ascending)
__global__ __launch_bounds__(1024) void chain_expand_kernel(const int* __restrict__ chain, const int* __restrict__ near, int C,
                                                            int S, int* __restrict__ order) {
    __shared__ int rank[CHAIN_MAX], start[CHAIN_MAX + 1], cnt[CHAIN_MAX];
    __shared__ unsigned short nr[KNN_MAX_CELLS];
    const int tid = threadIdx.x;
    for (int i = tid; i < S; i += 1024) { rank[chain[i]] = i; cnt[i] = 0; }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
        const int r = rank[near[c]];
        nr[c] = (unsigned short)r;
        atomicAdd(&cnt[r], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < S; ++i) { start[i] = acc; acc += cnt[i]; }
        start[S] = acc;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
        const int r = nr[c];
        int before = 0;
        for (int o = 0; o < c; ++o) before += (nr[o] == r) ? 1 : 0;
        order[start[r] + before] = c;
    }
}

// the pivots in chain order: pivT2[i][pos] = pivT[i][chain[pos]], likewise the norms
__global__ void permute_pivots_kernel(const float* __restrict__ pivT, const float* __restrict__ pnorm, int d, int C, int Cpad,
                                      const int* __restrict__ chain, float* __restrict__ pivT2, float* __restrict__ pnorm2,
                                      float* __restrict__ pivRM /* [Cpad][d]: the same, one pivot per row */) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= Cpad) return;
    const int c = pos < C ? chain[pos] : pos;
    for (int i = 0; i < d; ++i) {
        const float v = pivT[(long long)i * Cpad + c];
        pivT2[(long long)i * Cpad + pos] = v;
        pivRM[(long long)pos * d + i] = v;
    }
    pnorm2[pos] = pnorm[c];
}

// pivT[i][c] = feature i of pivot c (= row (c * m) / C of R, or row piv_rows[c]), pnorm[c] = |pivot c|^2
__global__ void pivots_kernel(const float* __restrict__ R, long long m, int d, int C, int Cpad,
                              float* __restrict__ pivT, float* __restrict__ pnorm, const int* __restrict__ piv_rows) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cpad) return;
    float acc = 0.0f;
    const float* row = R + (piv_rows != nullptr && c < C ? (long long)piv_rows[c] : (long long)c * m / C) * d;
    for (int i = 0; i < d; ++i) {
        const float v = c < C ? row[i] : 0.0f;
        pivT[i * Cpad + c] = v;
        acc = __fmaf_rn(v, v, acc);
    }
    pnorm[c] = c < C ? acc : __builtin_inff();
}

// cell[row] = pivot with the smallest |p|^2 - 2 x.p (lowest pivot number among equals);
// blockhist[block][c] = rows of this block in cell c.
// A block holds RB rows feature-major in LDS for the whole pass and streams the pivots through
// LDS in tiles of CELL_PT (any number of cells, any d up to 256); the 256 / RB threads of a row
// share a tile's groups of eight pivots among them and their results meet in LDS at the end.
template <int RB>
__global__ __launch_bounds__(CELL_NT) void assign_cells_kernel(const float* __restrict__ X, long long n, int d,
                                                               const float* __restrict__ pivT,
                                                               const float* __restrict__ pnorm, int C, int Cpad,
                                                               cell_t* __restrict__ cell,
                                                               int* __restrict__ blockhist,
                                                               float* __restrict__ qpd /* [Cpad][n] or NULL */,
                                                               unsigned* __restrict__ crad /* [Cpad] or NULL */) {
    constexpr int PG = CELL_NT / RB;       // threads per row
    extern __shared__ __attribute__((aligned(16))) float csm[];
    float* xT = csm;                       // [d][RB + 1]: feature-major, conflict-free both ways
    float* pT = xT + ((d * (RB + 1) + 3) & ~3);  // [d][CELL_PT], 16-byte aligned
    float* pn = pT + d * CELL_PT;          // [CELL_PT]
    float* best_s = pn + CELL_PT;          // [PG][RB]
    int* bestc_s = reinterpret_cast<int*>(best_s + PG * RB);  // [PG][RB]
    int* hist = bestc_s + PG * RB;         // [Cpad]
    const int tid = threadIdx.x;
    const int r = tid % RB, pg = tid / RB;
    const long long row0 = (long long)blockIdx.x * RB;
    const int nrows = (int)((n - row0) < RB ? (n - row0) : RB);
    for (int i = tid; i < Cpad; i += CELL_NT) hist[i] = 0;
    const float* src = X + row0 * d;
    for (int e = tid; e < nrows * d; e += CELL_NT) {
        const int rr = e / d, i = e - rr * d;
        xT[i * (RB + 1) + rr] = src[e];
    }
    for (int e = nrows * d + tid; e < RB * d; e += CELL_NT) {  // rows past the end: zeros, never written out
        const int rr = e / d, i = e - rr * d;
        xT[i * (RB + 1) + rr] = 0.0f;
    }
    __syncthreads();
    float best = __builtin_inff();
    int bestc = 0;
    // (pruned search) |x|^2, to turn the scores |p|^2 - 2 x.p into distances with a rounding-error
    // slack: the float32 chain is off by at most about 8e-6 (|p|^2 + |x|^2) up to 64 features, and
    // in proportion to the number of features beyond
    float xn = 0.0f;
    if (qpd != nullptr || crad != nullptr)
        for (int i = 0; i < d; ++i) { const float x = xT[i * (RB + 1) + r]; xn = __fmaf_rn(x, x, xn); }
    const float slack_coef = 1.6e-5f * (d > 64 ? (float)d / 64.0f : 1.0f);
    for (int c0 = 0; c0 < Cpad; c0 += CELL_PT) {
        if (c0 > 0) __syncthreads();  // everyone is done with the previous tile
        for (int e = tid; e < d * CELL_PT; e += CELL_NT) {
            const int i = e / CELL_PT, cc = e - i * CELL_PT;
            pT[e] = (c0 + cc < Cpad) ? pivT[(long long)i * Cpad + c0 + cc] : 0.0f;
        }
        if (tid < CELL_PT) pn[tid] = (c0 + tid < Cpad) ? pnorm[c0 + tid] : __builtin_inff();
        __syncthreads();
        for (int g8 = pg * CELL_TP; g8 < CELL_PT && c0 + g8 < Cpad; g8 += PG * CELL_TP) {
            float acc[CELL_TP];
#pragma unroll
            for (int u = 0; u < CELL_TP; ++u) acc[u] = pn[g8 + u];
            for (int i = 0; i < d; ++i) {
                const float x2 = -2.0f * xT[i * (RB + 1) + r];
                const float4 pa = *reinterpret_cast<const float4*>(pT + i * CELL_PT + g8);
                const float4 pb = *reinterpret_cast<const float4*>(pT + i * CELL_PT + g8 + 4);
                acc[0] = __fmaf_rn(x2, pa.x, acc[0]); acc[1] = __fmaf_rn(x2, pa.y, acc[1]);
                acc[2] = __fmaf_rn(x2, pa.z, acc[2]); acc[3] = __fmaf_rn(x2, pa.w, acc[3]);
                acc[4] = __fmaf_rn(x2, pb.x, acc[4]); acc[5] = __fmaf_rn(x2, pb.y, acc[5]);
                acc[6] = __fmaf_rn(x2, pb.z, acc[6]); acc[7] = __fmaf_rn(x2, pb.w, acc[7]);
            }
#pragma unroll
            for (int u = 0; u < CELL_TP; ++u)
                if (acc[u] < best) { best = acc[u]; bestc = c0 + g8 + u; }  // NaN rows stay in cell 0
            if (qpd != nullptr && r < nrows) {
#pragma unroll
                for (int u = 0; u < CELL_TP; ++u) {  // a lower bound of the distance to pivot c0 + g8 + u
                    const float slack = slack_coef * (pn[g8 + u] + xn);
                    const float lo2 = acc[u] + xn - slack;
                    qpd[(long long)(c0 + g8 + u) * n + row0 + r] = lo2 > 0.0f ? sqrtf(lo2) * 0.999999f : 0.0f;
                }
            }
        }
    }
    if (PG > 1) {  // the threads of a row: smallest score, lowest pivot number among equals
        best_s[pg * RB + r] = best;
        bestc_s[pg * RB + r] = bestc;
        __syncthreads();
        if (pg == 0) {
#pragma unroll
            for (int o = 1; o < PG; ++o) {
                const float ob = best_s[o * RB + r];
                const int oc = bestc_s[o * RB + r];
                if (ob < best || (ob == best && oc < bestc)) { best = ob; bestc = oc; }
            }
        }
    }
    if (pg == 0 && r < nrows) {
        if (bestc >= C) bestc = 0;
        if (crad != nullptr) {  // an upper bound of this row's distance to its own pivot
            const float pnb = pnorm[bestc];
            const float hi2 = best + xn + slack_coef * (pnb + xn);
            const float hi = hi2 > 0.0f ? sqrtf(hi2) * 1.000001f : 0.0f;  // (NaN rows: inf, their cell is never pruned)
            atomicMax(crad + bestc, __float_as_uint(hi == hi ? hi : __builtin_inff()));
        }
        cell[row0 + r] = (cell_t)bestc;
        atomicAdd(&hist[bestc], 1);
    }
    __syncthreads();
    for (int i = tid; i < C; i += CELL_NT) blockhist[(long long)blockIdx.x * C + i] = hist[i];
}

// The same assignment on the matrix cores, for the dense search (no bounds to write out): the
// scores |p|^2 - 2 x.p of 32 rows against 32 pivots are one chain of fp32 MFMAs (32x32x2), the
// rows' features the B operand in registers, a tile of pivots the A operand out of LDS -- lane
// (j, hh) ends up with 16 scores of ITS row j, so the running minimum needs no cross-lane traffic
// until the two lane halves meet at the end.  Which pivot a minimum belongs to rides in the low
// four mantissa bits of the score (the register number) and in the number of the tile.  The
// perturbation (16 ulp) can prefer one of two nearly equidistant pivots over the other; any
// assignment gives the same table.  WAVES x GROUPS x 32 rows per block.
template <int D2MAX, int GROUPS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void assign_cells_mfma_kernel(const float* __restrict__ X, long long n, int d,
                                                                       const float* __restrict__ pivRM /* [C][d] */,
                                                                       const float* __restrict__ pnorm, int C,
                                                                       cell_t* __restrict__ cell,
                                                                       int* __restrict__ blockhist) {
    constexpr int NT = WAVES * 64, RB = WAVES * GROUPS * 32;
    const int D2 = (d + 1) / 2, D2P = (D2 + 3) & ~3;
    const int LSP = 2 * D2P + ((((2 * D2P) / 4) % 2 == 0) ? 4 : 8);  // odd number of 16-byte slots per pivot row
    extern __shared__ __attribute__((aligned(16))) float asm_[];
    float* pt = asm_;                               // [32][LSP]: a tile of pivots, lane-half-major features
    float* pnl = pt + 32 * LSP;                     // [32]
    int* hist = reinterpret_cast<int*>(pnl + 32);   // [C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const long long row0 = (long long)blockIdx.x * RB;
    for (int i = tid; i < C; i += NT) hist[i] = 0;
    // B operand: -2 x features hh * D2 + s of this lane's row, per group
    float qf[GROUPS][D2MAX];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const long long row = row0 + (long long)(wave * GROUPS + g) * 32 + j;
#pragma unroll
        for (int sidx = 0; sidx < D2MAX; ++sidx) {
            const int f = hh * D2 + sidx;
            qf[g][sidx] = (row < n && sidx < D2 && f < d) ? -2.0f * X[row * d + f] : 0.0f;
        }
    }
    float best[GROUPS];
    int best_tile[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) { best[g] = __builtin_inff(); best_tile[g] = 0; }
    const int tiles = (C + 31) / 32;
    for (int t = 0; t < tiles; ++t) {
        __syncthreads();  // everyone is done with the previous tile
        {   // NT / 32 threads per pivot row, no divisions in the loop
            constexpr int TPR = NT / 32;
            const int pj = tid / TPR, pc = t * 32 + pj;
            const float* src = pivRM + (long long)pc * d;
            for (int c = tid % TPR; c < 2 * D2P; c += TPR) {
                const int ph = c >= D2P ? 1 : 0, ps = c - ph * D2P;
                const int f = ph * D2 + ps;
                pt[pj * LSP + c] = (ps < D2 && f < d && pc < C) ? src[f] : 0.0f;
            }
        }
        if (tid < 32) pnl[tid] = (t * 32 + tid < C) ? pnorm[t * 32 + tid] : 3.0e38f;
        __syncthreads();
        f32x16 acc[GROUPS];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 nv = lds_read_b128(pnl + 8 * q4 + 4 * hh);
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                acc[g][4 * q4 + 0] = nv.x; acc[g][4 * q4 + 1] = nv.y; acc[g][4 * q4 + 2] = nv.z; acc[g][4 * q4 + 3] = nv.w;
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < D2MAX / 4; ++s4) {
            if (4 * s4 < D2) {  // (wave-uniform)
                const float4 a4 = lds_read_b128(pt + j * LSP + hh * D2P + 4 * s4);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * s4 + e < D2) {
#pragma unroll
                        for (int g = 0; g < GROUPS; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], qf[g][4 * s4 + e], acc[g], 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            float key[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) key[r] = score_key(acc[g][r], r);
            const float a0 = fminf(fminf(key[0], key[1]), fminf(key[2], key[3]));
            const float a1 = fminf(fminf(key[4], key[5]), fminf(key[6], key[7]));
            const float a2 = fminf(fminf(key[8], key[9]), fminf(key[10], key[11]));
            const float a3 = fminf(fminf(key[12], key[13]), fminf(key[14], key[15]));
            const float kb = fminf(fminf(a0, a1), fminf(a2, a3));
            if (kb < best[g]) { best[g] = kb; best_tile[g] = t; }   // strict: the earlier tile among equals
        }
    }
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const unsigned r = __float_as_uint(best[g]) & 15u;
        int pc = best_tile[g] * 32 + (int)((r & 3u) + 8u * (r >> 2)) + 4 * hh;
        float b = best[g];
        const float ob = __shfl_xor(b, 32);
        const int oc = __shfl_xor(pc, 32);
        if (ob < b || (ob == b && oc < pc)) { b = ob; pc = oc; }
        const long long row = row0 + (long long)(wave * GROUPS + g) * 32 + j;
        if (hh == 0 && row < n) {
            if (pc >= C || !(b < __builtin_inff())) pc = 0;  // NaN rows stay in cell 0
            cell[row] = (cell_t)pc;
            atomicAdd(&hist[pc], 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < C; i += NT) blockhist[(long long)blockIdx.x * C + i] = hist[i];
}

// blockhist[b][c] -> rows of cell c in blocks before b (one block per cell scans its column);
// col_total[c] = rows in cell c
__global__ __launch_bounds__(256) void cell_column_scan_kernel(int* __restrict__ blockhist, int nblk, int C,
                                                               int* __restrict__ col_total) {
    __shared__ int part[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int per = (nblk + 255) / 256;
    const int b0 = tid * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += blockhist[(long long)b * C + c];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int t = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    int run = part[tid] - sum;  // rows of this cell in the blocks before b0
    for (int b = b0; b < b1; ++b) {
        const int v = blockhist[(long long)b * C + c];
        blockhist[(long long)b * C + c] = run;
        run += v;
    }
    if (tid == 255) col_total[c] = part[255];
}

// cell_start[c] = rows in cells before c (cell_start[C] = all rows); C is at most a few hundred
__global__ void cell_starts_kernel(const int* __restrict__ col_total, int C, int* __restrict__ cell_start) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) { cell_start[c] = acc; acc += col_total[c]; }
        cell_start[C] = acc;
    }
}

// perm[position] = row: rows ordered by (cell, row number).  Launched with the assignment
// kernel's rows per block as block size (the per-block histograms are of those blocks).
__global__ __launch_bounds__(256) void cell_scatter_kernel(const cell_t* __restrict__ cell, long long n,
                                                           int C, const int* __restrict__ blockoff,
                                                           const int* __restrict__ cell_start,
                                                           int* __restrict__ perm) {
    __shared__ cell_t cs[256];
    const int tid = threadIdx.x;
    const long long row = (long long)blockIdx.x * blockDim.x + tid;
    cs[tid] = row < n ? cell[row] : (cell_t)0xffff;
    __syncthreads();
    if (row >= n) return;
    const int mine = cs[tid];
    int before = 0;
    for (int t = 0; t < tid; ++t) before += (cs[t] == mine) ? 1 : 0;
    perm[cell_start[mine] + blockoff[(long long)blockIdx.x * C + mine] + before] = (int)row;
}

// home[b] = {first stage, stages} of query block b's home window and {first stage, stages} of its
// own cells inside it.  The own cells are the reference rows that share a cell with the block's
// queries; the window is that range widened, evenly to both sides, to max_stages stages of
// consecutive packed rows -- the neighbouring cells of the chain order, i.e. the rest of the
// queries' cluster.  (Own cells longer than the window: the window is their head.)
__global__ void home_stages_kernel(const cell_t* __restrict__ qcell, const int* __restrict__ perm_q,
                                   long long n, int qpb, int nqb, const int* __restrict__ rcell_start, int RT,
                                   long long m_pad, int max_stages, int widen, int* __restrict__ home) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nqb) return;
    const long long first = (long long)b * qpb;
    long long last = first + qpb - 1;
    if (last > n - 1) last = n - 1;
    int w0 = 0, wn = 0, o0 = 0, on = 0;
    if (first < n) {
        const int c_lo = qcell[perm_q[first]], c_hi = qcell[perm_q[last]];
        const long long r0 = rcell_start[c_lo], r1 = rcell_start[c_hi + 1];
        const long long total = m_pad / RT;
        long long s0 = r0 / RT, s1 = (r1 + RT - 1) / RT;
        if (s1 > total) s1 = total;
        if (s1 > s0 + max_stages) s1 = s0 + max_stages;
        if (s1 > s0) {
            o0 = (int)s0; on = (int)(s1 - s0);
            long long spare = widen ? max_stages - (s1 - s0) : 0;  // stages the window may add round the own cells
            long long a = s0 - spare / 2, z = s1 + (spare - spare / 2);
            if (a < 0) { z -= a; a = 0; }
            if (z > total) { a -= z - total; z = total; }
            if (a < 0) a = 0;
            w0 = (int)a; wn = (int)(z - a);
        }
    }
    home[4 * b] = w0; home[4 * b + 1] = wn; home[4 * b + 2] = o0; home[4 * b + 3] = on;
}

// Merge mode: a threshold to start from.  The running result's k-th distance D_k bounds what
// this reference block can still contribute; in score space (score = D - |q|^2 up to the
// rounding error the certificate knows) that is tau0, chosen so that the certificate's lower
// bound for "score >= tau0" lies just above D_k.
__global__ void seed_tau_kernel(const float* __restrict__ Q, const int* __restrict__ perm_q, int d, long long n,
                                const int* __restrict__ idx, const float* __restrict__ d2, int k,
                                double err_abs_coef, double rel_keep, const unsigned* __restrict__ maxnorm_bits,
                                float* __restrict__ tau0) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const long long qo = perm_q ? perm_q[q] : q;
    float t0 = __uint_as_float(TAU_MAX_BITS);
    if (idx[qo * k + k - 1] >= 0) {
        const double Dk = (double)d2[qo * k + k - 1];
        double qn = 0.0;
        for (int i = 0; i < d; ++i) qn += (double)Q[qo * d + i] * (double)Q[qo * d + i];
        const double rmax2 = (double)__uint_as_float(*maxnorm_bits) * (1.0 + 1e-5);
        double t = Dk / rel_keep - qn + err_abs_coef * (rmax2 + qn);
        t += (fabs(t) + Dk + qn) * 1e-6;
        float f = (float)t;
        if ((double)f < t) f = nextafterf(f, __builtin_inff());
        if (f < t0) t0 = f;
    }
    tau0[q] = t0;
}

// --------------------------------------------------------------------------------
// refine: one wave per query.  Exact D for the shortlisted references (oracle
// arithmetic: t = q_i - r_i; acc = fmaf(t, t, acc), feature order), (D, index) sort,
// top-k, certificate.
// --------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void refine_kernel(
    const float* __restrict__ Q, const float* __restrict__ Qp, const float* __restrict__ Rp, int rowf, int ls,
    int d, long long n, long long n_pad,
    const u64* __restrict__ lists, int S, FusedTail tail, int k, int kt, long long col_offset, int merge,
    double err_abs_coef,  // 2*G: |s + |q|^2 - T| <= err_abs_coef * (|r|^2 + |q|^2)
    double rel_keep,      // 1 - gamma: D >= T * rel_keep
    const unsigned* __restrict__ maxnorm_bits,
    const float* __restrict__ tau_lists,  // [shortlists]: the threshold each list ended with
    const int* __restrict__ perm_q,       // packed query position -> caller's row (NULL: the same)
    const int* __restrict__ perm_r,       // packed reference row -> caller's row (NULL: the same)
    const unsigned* __restrict__ share_claims,  // [streams] helper lists claimed (NULL: no stream sharing)
    int share_lists, long long share_row0,      // helper lists per stream; their first shortlist row
    int* __restrict__ out_idx,
    float* __restrict__ out_d2, float* __restrict__ out_dist, double* __restrict__ lower_bound,
    int* __restrict__ fail_rows, int* __restrict__ n_fail) {
    const int lane = threadIdx.x & 63;
    const long long qp = (long long)blockIdx.x * (blockDim.x >> 6) +
                         __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // position in the packed order
    if (qp >= n) return;
    const long long q = perm_q ? (long long)perm_q[qp] : qp;                 // the caller's row: outputs go there
    const float* qrow = Q + q * d;

    double qn = 0.0;
    for (int i = 0; i < d; ++i) qn += (double)qrow[i] * (double)qrow[i];
    const double rmax2 = (double)__uint_as_float(*maxnorm_bits) * (1.0 + 1e-5);

    u64 best = KEY_SENTINEL;
    double lb = __builtin_inf();
    if (merge) {
        if (lane < k) {
            int pi = out_idx[q * k + lane];
            if (pi >= 0) best = ((u64)__float_as_uint(out_d2[q * k + lane]) << 32) | (u64)(unsigned)pi;
        }
        lb = lower_bound[q];
    }
    const long long tail_q0 = (long long)tail.qb0 * tail.qpb;
    const bool in_tail = qp >= tail_q0;
    const int S_q = in_tail ? tail.St : S;
    // With stream sharing a stream (chunk s of a main-region query block: stream s qb0 + qb; piece s
    // of a tail query block: stream block0 + (qb - qb0) St + s) may have helper lists as well: slot
    // h for every h below the stream's claim count.
    const int per_chunk = share_claims != nullptr ? 1 + share_lists : 1;
    const long long n_streams = (long long)tail.block0 + (tail.rows / tail.qpb) * tail.St;
    for (int it = 0; it < S_q * per_chunk; ++it) {
        const int s = it / per_chunk, h = it - s * per_chunk - 1;  // h = -1: the chunk's own list
        long long lrow = (long long)s * n_pad + qp;
        if (in_tail && s > 0) lrow = tail.lists_row0 + (long long)(s - 1) * tail.rows + (qp - tail_q0);
        if (h >= 0) {
            const long long qb = qp / tail.qpb;
            const long long vstream = in_tail ? tail.block0 + (qb - tail.qb0) * tail.St + s : (long long)s * tail.qb0 + qb;
            if ((unsigned)h >= share_claims[vstream]) continue;  // (wave-uniform) nobody filled this one
            lrow = share_row0 + ((long long)h * n_streams + vstream) * tail.qpb + (qp - qb * tail.qpb);
        }
        u64 e = lists[lrow * 64 + lane];
        unsigned ridx = (unsigned)e;
        // every reference of this chunk outside the list scores at least this much; a threshold
        // that never came down from its ceiling means the list holds all there was
        float tau_s = tau_lists[lrow];
        if (tau_s >= __uint_as_float(TAU_MAX_BITS)) tau_s = __builtin_inff();
        // Which entries are worth their 240-byte gather.  The list is ascending in the fused
        // kernel's keys a (score within the certificate's error of the true |r|^2 - 2 q.r), so its
        // first k entries have D <= (a_k + |q|^2 + E)(1 + gamma), and so has the running result's
        // k-th entry, exactly.  An entry whose own lower bound (a + |q|^2 - E)(1 - gamma) lies
        // above either cannot be among the k nearest: k others are strictly closer.
        bool worth = ridx != 0xffffffffu;
        {
            const double E = err_abs_coef * (rmax2 + qn);
            const unsigned kth_a = __builtin_amdgcn_readlane((unsigned)(e >> 32), k - 1);  // orderable bits, sentinel: +inf
            double upper = ((double)ord2f(kth_a) + qn + E) * (2.0 - rel_keep);
            const unsigned run_hi = __builtin_amdgcn_readlane((unsigned)(best >> 32), k - 1);
            if (run_hi != 0xffffffffu) {
                const double run_k = (double)__uint_as_float(run_hi);
                upper = run_k < upper ? run_k : upper;
            }
            const double lower = (double)ord2f((unsigned)(e >> 32)) + qn - E;
            if (lower > 0.0 && lower * rel_keep > upper) worth = false;
        }
        u64 key = KEY_SENTINEL;
        if (worth) {
            float D = exact_dist2_packed(reinterpret_cast<const float4*>(Qp + qp * rowf),
                                         reinterpret_cast<const float4*>(Rp + (long long)ridx * ls), rowf / 4);
            const long long rrow = perm_r ? (long long)perm_r[ridx] : (long long)ridx;
            key = ((u64)__float_as_uint(D) << 32) | (u64)(unsigned)(rrow + col_offset);
        }
        key = wave_sort<64>(key, lane);
        u64 rev = __shfl(key, 63 - lane);
        best = wave_merge(umin64(best, rev), lane);
        if (tau_s < __builtin_inff()) {
            double lbs = ((double)tau_s + qn - err_abs_coef * (rmax2 + qn)) * rel_keep;
            lb = lbs < lb ? lbs : lb;
        }
    }
    // k-th smallest exact key (sentinel when fewer than k candidates exist)
    unsigned kth_hi = __builtin_amdgcn_readlane((unsigned)(best >> 32), k - 1);
    unsigned kth_lo = __builtin_amdgcn_readlane((unsigned)best, k - 1);
    const bool have_k = !(kth_hi == 0xffffffffu && kth_lo == 0xffffffffu);
    if (lane < k) {
        bool valid = best != KEY_SENTINEL;
        float D = __uint_as_float((unsigned)(best >> 32));
        out_idx[q * k + lane] = valid ? (int)(unsigned)best : -1;
        out_d2[q * k + lane] = valid ? D : __builtin_inff();
        if (out_dist) out_dist[q * k + lane] = valid ? sqrt_rn_f32(D) : __builtin_inff();
    }
    if (lane == 0) {
        lower_bound[q] = lb;
        bool certified;
        if (lb == __builtin_inf()) certified = true;  // every reference seen so far is shortlisted
        else certified = have_k && ((double)__uint_as_float(kth_hi) < lb);
        if (!certified) {
            int p = atomicAdd(n_fail, 1);
            fail_rows[p] = (int)q;
        }
    }
}

// --------------------------------------------------------------------------------
// exact_rows: direct evaluation against every reference row for the listed queries;
// merges into idx/dist2 (top-k of the union).
//   MODE 0  one 256-thread block per listed row streams all m references.
//   MODE 1  block (chunk, row): exact top-k of one reference chunk among the keys not
//           above the row's current k-th key -> parts[row][chunk][0..k)
//   MODE 2  one block per listed row merges its nch partial lists with the prior result.
// Modes 1+2 serve the first `row_limit` listed rows (a handful of rows then use the
// whole GPU instead of one CU each); MODE 0 serves the rows from `first_row` on.
// --------------------------------------------------------------------------------
static constexpr int EX_CAP = 2048;
static constexpr int EX_NT = 256;
static constexpr int EX_PART_STRIDE = 64;  // keys reserved per (row, chunk) partial list

__device__ void block_bitonic_sort(u64* v, int n_pow2, int tid) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += EX_NT) {
                int p = i ^ j;
                if (p > i) {
                    bool up = (i & k) == 0;
                    u64 a = v[i], b = v[p];
                    if ((a > b) == up) { v[i] = b; v[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(EX_NT) void exact_rows_kernel(
    const float* __restrict__ Q, const float* __restrict__ R, int d, long long m, int k,
    long long col_offset, const int* __restrict__ rows, const int* __restrict__ n_rows,
    int* __restrict__ idx_io, float* __restrict__ d2_io, float* __restrict__ dist_out,
    u64* __restrict__ parts, int nch, long long chunk_len, int first_row, int row_limit) {
    __shared__ u64 cand[EX_CAP];
    __shared__ u64 prior[SCMIAC_KNN_MAX_K_INTERNAL];
    __shared__ float qs[SCMIAC_KNN_MAX_D_INTERNAL];
    __shared__ unsigned cnt;
    __shared__ u64 thr;
    const int tid = threadIdx.x;
    int total = *n_rows;
    int f0, fstep;
    if (MODE == 0) {
        f0 = first_row + (int)blockIdx.x; fstep = (int)gridDim.x;
    } else {
        total = total < row_limit ? total : row_limit;
        f0 = (MODE == 1) ? (int)blockIdx.y : (int)blockIdx.x;
        fstep = (MODE == 1) ? (int)gridDim.y : (int)gridDim.x;
    }
    // the stream of keys this block looks at: reference rows [x_begin, x_end) (modes 0, 1)
    // or the nch * k partial keys of the row (mode 2)
    long long x_begin = 0, x_end = m;
    if (MODE == 1) {
        x_begin = (long long)blockIdx.x * chunk_len;
        x_end = x_begin + chunk_len < m ? x_begin + chunk_len : m;
    }
    if (MODE == 2) x_end = (long long)nch * k;
    for (int f = f0; f < total; f += fstep) {
        const long long q = rows[f];
        __syncthreads();
        if (MODE != 2)
            for (int i = tid; i < d; i += EX_NT) qs[i] = Q[q * d + i];
        if (tid < k) {
            int pi = idx_io[q * k + tid];
            u64 key = pi >= 0 ? (((u64)__float_as_uint(d2_io[q * k + tid]) << 32) | (u64)(unsigned)pi)
                              : KEY_SENTINEL;
            prior[tid] = key;
            if (MODE != 1) cand[tid] = key;
        }
        __syncthreads();
        if (tid == 0) {
            int nv = 0;
            while (nv < k && prior[nv] != KEY_SENTINEL) ++nv;
            // prior rows are sorted ascending: compact form, valid entries first
            cnt = (MODE == 1) ? 0u : (unsigned)nv;
            thr = (nv == k) ? prior[k - 1] : (KEY_SENTINEL - 1);
        }
        __syncthreads();
        for (long long base = x_begin; base < x_end; base += EX_NT) {
            long long x = base + tid;
            if (x < x_end) {
                u64 key;
                if (MODE == 2) {
                    key = parts[((long long)f * nch + x / k) * EX_PART_STRIDE + x % k];
                } else {
                    float D = exact_dist2(qs, R + x * d, d);
                    key = ((u64)__float_as_uint(D) << 32) | (u64)(unsigned)(x + col_offset);
                }
                if (key <= thr) {
                    bool dup = false;
                    if (MODE != 1)  // partial lists keep entries of the prior result; mode 2 drops them
                        for (int p = 0; p < k; ++p) dup |= (prior[p] == key);
                    if (!dup) {
                        unsigned slot = atomicAdd(&cnt, 1u);
                        cand[slot] = key;  // slot < EX_CAP: compaction below keeps EX_NT free
                    }
                }
            }
            __syncthreads();
            const unsigned c = cnt;  // every thread reads the same value ...
            __syncthreads();         // ... before any thread of the next round can add to it
            if (c > (unsigned)(EX_CAP - EX_NT)) {
                for (int i = c + tid; i < EX_CAP; i += EX_NT) cand[i] = KEY_SENTINEL;
                __syncthreads();
                block_bitonic_sort(cand, EX_CAP, tid);
                if (tid == 0) {
                    cnt = c < (unsigned)k ? c : (unsigned)k;
                    if (c >= (unsigned)k) thr = cand[k - 1];
                }
                __syncthreads();
            }
        }
        {
            unsigned c = cnt;
            int np2 = 64;
            while (np2 < (int)c) np2 <<= 1;
            for (int i = c + tid; i < np2; i += EX_NT) cand[i] = KEY_SENTINEL;
            __syncthreads();
            block_bitonic_sort(cand, np2, tid);
            if (tid < k) {
                u64 key = (tid < (int)c) ? cand[tid] : KEY_SENTINEL;
                if (MODE == 1) {
                    parts[((long long)f * nch + blockIdx.x) * EX_PART_STRIDE + tid] = key;
                } else {
                    bool valid = key != KEY_SENTINEL;
                    float D = __uint_as_float((unsigned)(key >> 32));
                    idx_io[q * k + tid] = valid ? (int)(unsigned)key : -1;
                    d2_io[q * k + tid] = valid ? D : __builtin_inff();
                    if (dist_out) dist_out[q * k + tid] = valid ? sqrt_rn_f32(D) : __builtin_inff();
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------
// host-side launchers
// --------------------------------------------------------------------------------
static const int kD2Sizes[] = {8, 15, 16, 20, 25, 32, 40, 50, 64, 80, 96, 100, 112, 128};
int knn_pick_d2(int d) {
    int need = (d + 1) / 2;
    for (int v : kD2Sizes)
        if (v >= need) return v;
    return -1;
}

// Diagnostic / tuning state (set through scmiac_knn_set_tuning only; all zero in normal use).
// One mutex guards it together with the occupancy cache below: host threads may search
// concurrently.
static std::mutex g_plan_mu;
static FusedTuning g_tuning{};

static bool fused_variant_compiled(int D2, int nw, int rt, int pcap, int gp);

FusedTuning knn_get_tuning() {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    return g_tuning;
}

bool knn_set_tuning(const FusedTuning& t) {
    if (t.has_variant) {  // must be a compiled combination for at least one feature size
        bool any = false;
        for (int v : kD2Sizes) any = any || fused_variant_compiled(v, t.nw, t.rt, t.pcap, t.gp);
        if (!any) return false;
    }
    if (t.slots < 0 || t.tail_min_rows < 0 || t.ablate < 0 || t.home_rows < 0 || t.seed_rows < 0 || t.cell_rows < 0)
        return false;
    // (runs of fewer than two stages would ask for the next run before the answer to the last request is in)
    if (t.share < -1 || t.share > 8 || t.share_chunk < 0 || t.share_chunk == 1 || t.share_min_left < 0 || t.share_order < 0 ||
        t.share_order > 2)
        return false;
    std::lock_guard<std::mutex> lock(g_plan_mu);
    g_tuning = t;
    return true;
}

// (waves per block, reference rows per LDS stage) of the fused kernel
static void fused_variant(int D2, const FusedTuning& tn, int* nw, int* rt, int* pcap, int* gp) {
    // 12 waves per block (3 per SIMD) wherever the registers allow (up to 100 features), the
    // pending-candidate buffers in global memory and all of the LDS given to two stages as
    // long as it holds: 320 rows up to 50 features, 256 up to 64, 128 up to 128, 64 beyond
    // (fewer barriers, and the waves' uneven candidate work averages out over a longer stage).
    *nw = (D2 <= 50) ? 12 : 8;
    *rt = (D2 <= 25) ? 320 : (D2 <= 32) ? 256 : (D2 <= 64) ? 128 : 64;
    *pcap = 64;  // 64 pending candidates per query: half as many merges as with 32
    *gp = 1;
    // tuning aid: another compiled combination for this feature size, when there is one
    if (tn.has_variant && fused_variant_compiled(D2, tn.nw, tn.rt, tn.pcap, tn.gp)) {
        *nw = tn.nw; *rt = tn.rt; *pcap = tn.pcap; *gp = tn.gp;
    }
}

static bool fused_variant_compiled(int D2, int nw, int rt, int pcap, int gp) {
#define SCMIAC_HAS_CASE(V_, NW_, RT_, PC_, GP_) \
    if (D2 == V_ && nw == NW_ && rt == RT_ && pcap == PC_ && gp == GP_) return true;
    SCMIAC_FUSED_VARIANTS(SCMIAC_HAS_CASE)
#undef SCMIAC_HAS_CASE
    return false;
}
// Blocks of the fused kernel the current device runs at once.  Asked of the HIP runtime once per
// (device, variant) -- which initialises the runtime on the first planning call of a process with
// a GPU; 256 (one block on each of an MI355X's compute units) where no device answers.
static int fused_concurrent_blocks(const KnnGeometry& g) {
    static int cache[64][3];  // (device, variant) -> slots; a handful of variants exist
    static int n_cached = 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    std::lock_guard<std::mutex> lock(g_plan_mu);
    const int tag = (((g.D2 * 100 + g.nw) * 1000 + g.RT) * 100 + g.pcap) * 2 + g.gp;
    for (int i = 0; i < n_cached; ++i)
        if (cache[i][0] == tag && cache[i][2] == dev) return cache[i][1];
    int slots = 256;
    if (dev >= 0)
        for (auto occ : {fused_occupancy_tu0, fused_occupancy_tu1, fused_occupancy_tu2, fused_occupancy_tu3}) {
            const int v = occ(g);
            if (v > 0) slots = v;  // (0: the variant is not that file's)
        }
    if (n_cached < 64) { cache[n_cached][0] = tag; cache[n_cached][1] = slots; cache[n_cached][2] = dev; ++n_cached; }
    return slots;
}

bool fused_pruned_compiled(const KnnGeometry& g) { return fused_pruned_compiled_tu4(g) || fused_pruned_compiled_tu6(g); }
static bool fused_share_compiled(const KnnGeometry& g) { return fused_share_compiled_tu5(g); }

KnnGeometry knn_geometry(long long n, long long m, int d, int k, int splits, int mode, const FusedTuning& tn) {
    KnnGeometry g{};
    g.D2 = knn_pick_d2(d);
    if (g.D2 < 0) return g;
    g.D2P = (g.D2 + 3) & ~3;
    g.rowf = 2 * g.D2P;
    g.ls = g.rowf + (((g.rowf / 4) % 2 == 0) ? 4 : 8);
    fused_variant(g.D2, tn, &g.nw, &g.RT, &g.pcap, &g.gp);
    // locality partition: worth its passes over both sets once the stream is long enough to have
    // a candidate-dense start worth cutting
    g.cells = 0; g.cells_pad = 0; g.home_max = 0; g.seed_stages = 0;
    if (!tn.no_locality && m >= 16384 && n >= 2048) {
        long long c = m / (tn.cell_rows > 0 ? tn.cell_rows : 1024);
        // (assigning a row costs one distance per cell -- on the matrix cores for the dense search,
        // on the vector units with a bound per (query, cell) written out for the pruned one: the
        // library's own choice stops at 1024 and 248 cells, an explicit cell size may ask for up to
        // KNN_MAX_CELLS)
        const long long cmax = tn.cell_rows > 0 ? KNN_MAX_CELLS : (mode == 1 ? 248 : 1024);
        c = c < 8 ? 8 : (c > cmax ? cmax : c);
        g.cells = (int)c;
        g.cells_pad = (g.cells + 7) & ~7;
        // a block streams at most this many home stages first ...
        g.home_max = ((tn.home_rows > 0 ? tn.home_rows : 8192) + g.RT - 1) / g.RT;
        // ... and looks at this many of them for its thresholds
        g.seed_stages = ((tn.seed_rows > 0 ? tn.seed_rows : 6144) + g.RT - 1) / g.RT;
    }
    g.qpb = g.nw * 32;
    g.n_pad = ((n + g.qpb - 1) / g.qpb) * g.qpb;
    g.nqb = (int)(g.n_pad / g.qpb);
    const long long rows_rt = (m + g.RT - 1) / g.RT;  // reference rows in units of one LDS stage
    int slots = fused_concurrent_blocks(g);
    long long min_chunk = g.cells > 0 ? 2048 : 8192;  // shortest reference chunk considered
    // test aids: pretend the GPU runs fewer blocks at once / allow short chunks, so that
    // small problems exercise every region of the grid
    if (tn.slots > 0) slots = tn.slots;
    if (tn.tail_min_rows > 0) min_chunk = tn.tail_min_rows;
    const bool no_tail = tn.no_tail != 0;  // tuning aid

    // Grid plan.  Query blocks [0, qb0) ("main") are searched in S reference chunks each,
    // qb0 * S being a whole number of rounds of `slots` concurrent blocks; the remaining
    // query blocks ("tail") would leave the last round partly empty and are searched in St
    // chunks each instead.  More chunks = more, shorter blocks (fuller rounds) but every
    // chunk refills its shortlist from scratch and adds a list per query to the refine
    // pass.  (S, St) minimise a cycle model of the fused kernel fitted on MI355X.
    int S = 1, St = 1, qb0 = g.nqb;
    // A pruned search streams a query block's own cluster and little else: cutting its reference
    // range into chunks would buy no balance and cost every chunk but one the sight of its
    // queries' own cells (the thresholds the reach test lives on).  One block per query block.
    // (One case apart: a few query blocks more than one round -- 261 of them are two rounds on 256
    // compute units, which doubles the time of a search whose data turn out not to prune.  There
    // the dense plan is kept, whose main region is one round of whole streams and whose tail pieces
    // fill the rest.  Only there: every piece pays a seed pass over its queries' own cells and a
    // reach test of its own, which doubled the time of a search that does prune at 326 query blocks
    // and tripled it at 652.)
    const bool want_prune = mode == 1 && g.cells > 0 && fused_pruned_compiled(g);
    if (want_prune && splits <= 0 && !(g.nqb > slots && g.nqb - slots <= slots / 8)) {
        // (S = St = 1)
    } else if (splits > 0) {
        S = splits;
        if (S > rows_rt) S = (int)(rows_rt > 0 ? rows_rt : 1);
    } else {
        const double kt = (double)knn_threshold_rank(k, d);
        auto block_cycles = [&](double chunk) {
            if (chunk < 64.0) chunk = 64.0;
            const double sub = chunk / 32.0;                   // 32-row subtiles
            if (g.cells > 0) {
                // with the locality partition the thresholds start tight: what remains is the seed
                // pass, the home rows' candidates and the closing merges -- a fixed price per
                // block (fitted on MI355X: about 1.6 M cycles next to the seed pass) -- and a
                // stream that runs at 84 % of the MFMA rate in chunk-sized pieces
                const double mfma = g.D2 * 64.0 * (g.nw / 4);
                return sub * mfma / 0.84 + g.seed_stages * (g.RT / 32.0) * mfma / 0.9 + 1.6e6;
            }
            const double dense = 1024.0 * kt;                  // rows over which every subtile has candidates
            const double entries = chunk <= dense ? sub : dense / 32.0 * (1.0 + log(chunk / dense));
            const double merges = 64.0 + kt * log(chunk / 64.0);
            // MFMA time of a subtile with nw/4 waves sharing each SIMD, at the measured no-selection efficiency
            return sub * (g.D2 * 64.0 * (g.nw / 4) / 0.86) + entries * 2300.0 + merges * 4400.0;
        };
        long long s_max = (long long)rows_rt * g.RT / min_chunk;
        if (s_max > 64) s_max = 64;
        if (s_max < 1) s_max = 1;
        double best = 1e300;
        for (int s = 1; s <= s_max; ++s) {
            int gcd = s, b = slots;
            while (b) { int tmp = gcd % b; gcd = b; b = tmp; }
            const int step = slots / gcd;  // main query blocks come in multiples of this
            int q0 = (g.nqb / step) * step;
            if (no_tail) q0 = g.nqb;
            const double chunk_s = (double)m / s;
            const double main_cost = (double)((long long)q0 * s / slots) * block_cycles(chunk_s);
            const int tail_blocks = g.nqb - q0;
            const double refine_main = 43.0 * (double)q0 * g.qpb * s;
            if (tail_blocks == 0) {
                const double c = main_cost + refine_main;
                if (c < best) { best = c; S = s; St = 1; qb0 = q0; }
                continue;
            }
            if (s > 1 && q0 == 0) continue;  // an all-tail plan does not depend on s: counted once
            for (int st = 1; st <= s_max; ++st) {
                const long long rounds = ((long long)tail_blocks * st + slots - 1) / slots;
                const double c = main_cost + refine_main + (double)rounds * block_cycles((double)m / st) +
                                 43.0 * (double)tail_blocks * g.qpb * st;
                if (c < best) { best = c; S = s; St = st; qb0 = q0; }
            }
        }
    }
    long long per = (rows_rt + S - 1) / S;
    g.chunk_rows = per * g.RT;
    if (g.chunk_rows == 0) g.chunk_rows = g.RT;
    g.S = (int)((rows_rt * g.RT + g.chunk_rows - 1) / g.chunk_rows);
    if (g.S < 1) g.S = 1;
    g.m_pad = (long long)g.S * g.chunk_rows;

    FusedTail& t = g.tail;
    t.qb0 = qb0; t.St = St; t.block0 = qb0 * g.S; t.qpb = g.qpb; t.m_pad = g.m_pad;
    t.rows = (long long)(g.nqb - qb0) * g.qpb;
    t.lists_row0 = (long long)g.S * g.n_pad;
    {
        long long per_t = (g.m_pad / g.RT + St - 1) / St;
        t.chunk_rows = per_t * g.RT;
    }
    g.list_rows = (long long)g.S * g.n_pad + (long long)(t.St - 1) * t.rows;
    g.blocks = t.block0 + (g.nqb - t.qb0) * t.St;
    // pruned search: where the partition applies, the variant has a pruned instantiation and a
    // block's stage list fits 16-bit stage numbers; the dense search otherwise
    const long long longest = g.chunk_rows > t.chunk_rows ? g.chunk_rows : t.chunk_rows;
    g.prune = (want_prune && longest / g.RT < 65535) ? 1 : 0;
    // Stream sharing: when the main region is a single round of the grid (every block owns its
    // stream from the first stage to the last, and the launch would end with the slowest of them)
    // and the streams are long enough to be worth handing out in runs.  Two helper lists per
    // stream.  (Not for the diagnostic instantiation, which has no such kernel.)
    g.share = 0; g.share_chunk = 0; g.share_min_left = 0; g.share_tail_first = 0; g.share_row0 = g.list_rows;
    if (mode == 0 && !g.prune && tn.share >= 0 && !tn.diag && !(tn.ablate & (63 | 128)) && g.blocks >= 2 &&
        g.blocks <= 4096 && fused_share_compiled(g)) {
        const int chunk = tn.share_chunk > 0 ? tn.share_chunk : 4;
        const bool one_round = t.block0 <= slots && g.blocks <= 2 * slots && longest / g.RT >= 12 * chunk;
        if (tn.share > 0 || one_round) {
            g.share = tn.share > 0 ? tn.share : 2;
            g.share_chunk = chunk;
            g.share_min_left = tn.share_min_left > 0 ? tn.share_min_left : 8;
            g.share_tail_first = tn.share_order == 2 ? 0 : 1;  // the library's choice: tail pieces first (measured)
            g.list_rows += (long long)g.share * g.blocks * g.qpb;
        }
    }
    return g;
}

hipError_t launch_pack(const float* X, long long n, int d, const KnnGeometry& g, float scale, float* Xp,
                       int ld, float* norms, long long n_pad, unsigned* maxnorm_bits, const int* perm,
                       hipStream_t stream) {
    dim3 block(256), grid((unsigned)((n_pad + PACK_ROWS - 1) / PACK_ROWS));
    hipLaunchKernelGGL(pack_rows_kernel, grid, block, 0, stream, X, n, d, g.D2, g.D2P, scale, Xp, ld, norms,
                       n_pad, maxnorm_bits, perm);
    return hipGetLastError();
}

// One side of the locality partition: cell of every row of X (nearest pivot), the permutation
// that orders the rows by (cell, row number) and the first position of every cell.
template <int RB>
static hipError_t launch_assign(const float* X, long long n, int d, int C, int Cpad, const float* pivT, const float* pnorm,
                                cell_t* cell, int* blockhist, float* qpd, unsigned* crad, hipStream_t stream) {
    constexpr int PG = CELL_NT / RB;
    const int nblk = (int)((n + RB - 1) / RB);
    const size_t lds = sizeof(float) * ((size_t)((d * (RB + 1) + 3) & ~3) + (size_t)d * CELL_PT + CELL_PT + (size_t)PG * RB) +
                       sizeof(int) * ((size_t)PG * RB + Cpad);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&assign_cells_kernel<RB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(assign_cells_kernel<RB>, dim3((unsigned)nblk), dim3(CELL_NT), lds, stream, X, n, d, pivT, pnorm, C,
                       Cpad, cell, blockhist, qpd, crad);
    return hipGetLastError();
}

template <int D2MAX, int GROUPS, int WAVES>
static hipError_t launch_assign_mfma(const float* X, long long n, int d, int C, const float* pivRM, const float* pnorm,
                                     cell_t* cell, int* blockhist, hipStream_t stream) {
    constexpr int RB = WAVES * GROUPS * 32;
    const int D2P = (((d + 1) / 2) + 3) & ~3;
    const int LSP = 2 * D2P + ((((2 * D2P) / 4) % 2 == 0) ? 4 : 8);
    const size_t lds = sizeof(float) * (32 * (size_t)LSP + 32) + sizeof(int) * (size_t)C;
    const int nblk = (int)((n + RB - 1) / RB);
    hipLaunchKernelGGL((assign_cells_mfma_kernel<D2MAX, GROUPS, WAVES>), dim3((unsigned)nblk), dim3(WAVES * 64), lds, stream, X, n,
                       d, pivRM, pnorm, C, cell, blockhist);
    return hipGetLastError();
}

static hipError_t launch_cells(const float* X, long long n, int d, const KnnGeometry& g, const float* pivT,
                               const float* pivRM, const float* pnorm, cell_t* cell, int* blockhist, int* cell_start,
                               int* perm, float* qpd, unsigned* crad, hipStream_t stream) {
    const int C = g.cells, Cpad = g.cells_pad;
    const int rb = cell_rows_per_block(d);
    const int nblk = (int)((n + rb - 1) / rb);
    hipError_t e;
    if (qpd == nullptr && crad == nullptr) {  // no bounds wanted: the matrix cores (same rows per block)
        e = rb == 256 ? launch_assign_mfma<32, 2, 4>(X, n, d, C, pivRM, pnorm, cell, blockhist, stream)
          : rb == 128 ? launch_assign_mfma<64, 1, 4>(X, n, d, C, pivRM, pnorm, cell, blockhist, stream)
                      : launch_assign_mfma<128, 1, 2>(X, n, d, C, pivRM, pnorm, cell, blockhist, stream);
    } else {
        e = rb == 256 ? launch_assign<256>(X, n, d, C, Cpad, pivT, pnorm, cell, blockhist, qpd, crad, stream)
          : rb == 128 ? launch_assign<128>(X, n, d, C, Cpad, pivT, pnorm, cell, blockhist, qpd, crad, stream)
                      : launch_assign<64>(X, n, d, C, Cpad, pivT, pnorm, cell, blockhist, qpd, crad, stream);
    }
    if (e != hipSuccess) return e;
    // (the totals pass through the tail of the start array's own allocation: C + 1 .. 2 C + 1)
    int* col_total = cell_start + C + 1;
    hipLaunchKernelGGL(cell_column_scan_kernel, dim3((unsigned)C), dim3(256), 0, stream, blockhist, nblk, C, col_total);
    hipLaunchKernelGGL(cell_starts_kernel, dim3(1), dim3(64), 0, stream, col_total, C, cell_start);
    hipLaunchKernelGGL(cell_scatter_kernel, dim3((unsigned)nblk), dim3(rb), 0, stream, cell, n, C, blockhist,
                       cell_start, perm);
    return hipGetLastError();
}

// rows per block of the partition kernels for d features (the per-block histograms' row count)
int knn_cell_block_rows(int d) { return cell_rows_per_block(d); }

hipError_t launch_locality(const float* Q, long long n, const float* R, long long m, int d, const KnnGeometry& g,
                           const LocalityWs& w, hipStream_t stream) {
    const int* piv_rows = nullptr;
    if (g.prune && w.crad != nullptr && w.fps_sample != nullptr) {
        hipLaunchKernelGGL(fps_pivots_kernel, dim3(1), dim3(FPS_NT), 0, stream, R, m, d, g.cells, w.piv_rows, w.fps_sample);
        piv_rows = w.piv_rows;
    }
    // the pivots as chosen (scratch copy), the leads' mutual distances, the chain, the pivots in chain order
    const unsigned pgrid = (unsigned)((g.cells_pad + 63) / 64);
    hipLaunchKernelGGL(pivots_kernel, dim3(pgrid), dim3(64), 0, stream, R, m, d, g.cells, g.cells_pad, w.pivT_tmp, w.pnorm_tmp,
                       piv_rows);
    {
        const int stride = (g.cells + CHAIN_MAX - 1) / CHAIN_MAX;
        const int S = (g.cells + stride - 1) / stride;
        int* near = w.piv_rows + g.cells_pad;         // [cells]
        int* chain = near + g.cells_pad;              // [S]
        int* order = chain + g.cells_pad;             // [cells]
        hipLaunchKernelGGL(pivot_dist_kernel, dim3((unsigned)g.cells), dim3(256), 0, stream, w.pivT_tmp, d, g.cells, g.cells_pad,
                           stride, S, w.pivD, near);
        const size_t lds = (sizeof(_Float16) * (size_t)S * S + 15) & ~(size_t)15;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_order_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(chain_order_kernel, dim3(1), dim3(64), lds, stream, w.pivD, S, chain);
        hipLaunchKernelGGL(chain_expand_kernel, dim3(1), dim3(1024), 0, stream, chain, near, g.cells, S, order);
        hipLaunchKernelGGL(permute_pivots_kernel, dim3(pgrid), dim3(64), 0, stream, w.pivT_tmp, w.pnorm_tmp, d, g.cells,
                           g.cells_pad, order, w.pivT, w.pnorm, w.pivRM);
    }
    // (a pruned search needs exact bounds on both sides: both use the VALU kernel)
    hipError_t e = launch_cells(R, m, d, g, w.pivT, w.pivRM, w.pnorm, w.cell_r, w.hist_r, w.cstart_r, w.perm_r, nullptr, w.crad, stream);
    if (e != hipSuccess) return e;
    e = launch_cells(Q, n, d, g, w.pivT, w.pivRM, w.pnorm, w.cell_q, w.hist_q, w.cstart_q, w.perm_q, w.qpd, nullptr, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(home_stages_kernel, dim3((unsigned)((g.nqb + 63) / 64)), dim3(64), 0, stream, w.cell_q, w.perm_q,
                       n, g.qpb, g.nqb, w.cstart_r, g.RT, g.m_pad, g.home_max, g.prune ? 0 : 1, w.home);
    return hipGetLastError();
}

// error model shared by the refine certificate and the merge-mode seed (see launch_refine)
static void knn_error_model(int d, const KnnGeometry& g, double* err_abs_coef, double* rel_keep) {
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double G = 2.0 * (double)(d + 2 * g.D2 + 4 + 32) * u;
    *err_abs_coef = 2.0 * G;
    *rel_keep = 1.0 - 2.0 * (double)(d + 4) * u;
}

hipError_t launch_seed_tau(const float* Q, const int* perm_q, int d, long long n, const KnnGeometry& g, const int* idx,
                           const float* d2, int k, const unsigned* maxnorm_bits, float* tau0, hipStream_t stream) {
    double coef, keep;
    knn_error_model(d, g, &coef, &keep);
    hipLaunchKernelGGL(seed_tau_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, Q, perm_q, d, n, idx, d2,
                       k, coef, keep, maxnorm_bits, tau0);
    return hipGetLastError();
}

hipError_t launch_fused(const KnnGeometry& g, const FusedTuning& tn, const float* Qp, const float* Rp, const float* rnorm,
                        unsigned long long* lists, void* pend_g_, long long n, int kt, const int* home,
                        const float* tau0, float* tau_out, const PruneArgs& prune, const ShareArgs& share,
                        unsigned long long* diag, hipStream_t stream) {
    uint2* pend_g = static_cast<uint2*>(pend_g_);
    // Three instantiations of every variant: the product kernel (dense), the pruned product kernel
    // (library variants) and a diagnostic one that carries per-wave counters,
    // the ablations (tuning bits: 1 no selection, 2 no staging, 4 no barrier, 8 empty candidate
    // path, 16 candidate arithmetic only, 32 no seed pass, 128 no per-subtile test) and the pruning code as well.
    const int ablate = tn.ablate & (63 | 128);
    // A fourth one where the plan shares streams (kind 3, fused_knn_share.h).
    const int kind = (diag != nullptr || ablate != 0) ? 1 : (prune.qpd != nullptr ? 2 : (g.share > 0 && share.cursor != nullptr ? 3 : 0));
    const FusedLaunch a{Qp, Rp, rnorm, lists, pend_g, n, kt, ablate, kind, home, tau0, tau_out, prune, share, diag, stream};
    for (auto launch : {launch_fused_tu0, launch_fused_tu1, launch_fused_tu2, launch_fused_tu3, launch_fused_tu4,
                        launch_fused_tu5, launch_fused_tu6}) {
        const hipError_t e = launch(g, a);
        if (e != hipErrorNotFound) return e;  // hipErrorNotFound: the variant (of this kind) is not that file's
    }
    return hipErrorInvalidValue;
}

PruneArgs knn_prune_args(const KnnGeometry& g, const LocalityWs& loc, const unsigned* maxnorm, long long n, long long m,
                         int d) {
    PruneArgs a{};
    if (!g.prune || loc.qpd == nullptr) return a;
    double err_abs_coef, rel_keep;
    knn_error_model(d, g, &err_abs_coef, &rel_keep);
    a = PruneArgs{loc.qpd, n, loc.crad, loc.cstart_r, loc.perm_q, loc.cell_q, maxnorm, (float)(err_abs_coef * 1.001), g.cells,
                  loc.order, loc.order_stride, m, loc.stats};
    return a;
}

hipError_t launch_refine(const float* Q, const float* Qp, const float* Rp, int d, long long n, const KnnGeometry& g,
                         const unsigned long long* lists, int k, int kt, long long col_offset, int merge,
                         const unsigned* maxnorm_bits, const float* tau_lists, const int* perm_q, const int* perm_r,
                         const unsigned* share_claims, int* idx, float* d2, float* dist, double* lb, int* fail_rows,
                         int* n_fail, hipStream_t stream) {
    // rounding-error model of the fused score s against T = |q|^2 + |r|^2 - 2 q.r (exact),
    // with gamma_n = n u / (1 - n u), u = 2^-24: the MFMA chain (2*D2 products) and the norm
    // chain (d terms) are each a float32 inner product, so
    //   |s + |q|^2 - T| <= G (|r| + |q|)^2 <= 2 G (|r|^2 + |q|^2),  G = gamma_{d + 2*D2 + 1}
    //   D >= T (1 - gamma_{d+2})
    // both taken with a factor-two margin.
    // (+ 32 u in G: the selection key replaces the low four mantissa bits of the score)
    double err_abs_coef, rel_keep;
    knn_error_model(d, g, &err_abs_coef, &rel_keep);
    dim3 block(256), grid((unsigned)((n + 3) / 4));
    hipLaunchKernelGGL(refine_kernel, grid, block, 0, stream, Q, Qp, Rp, g.rowf, g.ls, d, n, g.n_pad, lists, g.S, g.tail, k, kt,
                       col_offset, merge, err_abs_coef, rel_keep, maxnorm_bits, tau_lists, perm_q, perm_r,
                       g.share > 0 ? share_claims : nullptr, g.share, g.share_row0, idx, d2, dist, lb, fail_rows, n_fail);
    return hipGetLastError();
}

hipError_t launch_exact_rows(const float* Q, const float* R, int d, long long n, long long m, int k,
                             long long col_offset, const int* rows, const int* n_rows, int* idx,
                             float* d2, float* dist, void* scratch, size_t scratch_bytes,
                             hipStream_t stream) {
    // With scratch memory the first rows of the list are searched by many blocks each.
    int nch = (int)((m + 1023) / 1024);
    if (nch > 256) nch = 256;
    long long row_limit = 0;
    if (scratch && nch >= 2) {
        row_limit = (long long)(scratch_bytes / (sizeof(u64) * EX_PART_STRIDE * (size_t)nch));
        if (row_limit > n) row_limit = n;
        if (row_limit > 4096) row_limit = 4096;
    }
    u64* parts = static_cast<u64*>(scratch);
    if (row_limit > 0) {
        long long chunk_len = (m + nch - 1) / nch;
        unsigned ry = (unsigned)(row_limit < 16 ? row_limit : 16);
        hipLaunchKernelGGL(exact_rows_kernel<1>, dim3((unsigned)nch, ry), dim3(EX_NT), 0, stream, Q, R, d, m, k,
                           col_offset, rows, n_rows, idx, d2, dist, parts, nch, chunk_len, 0, (int)row_limit);
        unsigned gb = (unsigned)(row_limit < 256 ? row_limit : 256);
        hipLaunchKernelGGL(exact_rows_kernel<2>, dim3(gb), dim3(EX_NT), 0, stream, Q, R, d, m, k, col_offset,
                           rows, n_rows, idx, d2, dist, parts, nch, chunk_len, 0, (int)row_limit);
    }
    long long rest = n - row_limit;  // listed rows beyond the scratch capacity
    if (rest > 0) {
        long long gmax = rest < 2048 ? rest : 2048;
        hipLaunchKernelGGL(exact_rows_kernel<0>, dim3((unsigned)gmax), dim3(EX_NT), 0, stream, Q, R, d, m, k,
                           col_offset, rows, n_rows, idx, d2, dist, parts, nch, 0ll, (int)row_limit, 0);
    }
    return hipGetLastError();
}

}  // namespace scmiac
