// Integer kernels that follow the k-NN tables: mutual-nearest-neighbour anchors,
// shared-neighbour scores, score normalisation tail + cell-type filter.  All are
// HBM-bound gather work; outputs keep the reference's row order through an ordered
// (stable) compaction: flags -> per-block counts -> scan of the counts -> scatter.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace scmiac {

typedef long long i64;
static constexpr int CB_NT = 256;          // threads per compaction block
static constexpr int CB_ROWS = 4;          // items per thread
static constexpr int CB_ITEMS = CB_NT * CB_ROWS;

// workspace carve-up (all offsets multiples of 256 bytes)
struct CompactWs {
    unsigned char* flag_a;  // [n_items]
    unsigned char* flag_b;  // [n_items]
    i64* bsum_a;            // [nblocks + 1]
    i64* bsum_b;            // [nblocks + 1]
};
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline i64 compact_blocks(i64 n_items) { return (n_items + CB_ITEMS - 1) / CB_ITEMS; }
size_t compact_workspace_bytes(long long n_items) {
    if (n_items < 1) n_items = 1;
    size_t nb = (size_t)compact_blocks(n_items) + 1;
    return 2 * up256((size_t)n_items) + 2 * up256(nb * sizeof(i64));
}
static CompactWs carve(void* ws, i64 n_items) {
    if (n_items < 1) n_items = 1;
    size_t nb = (size_t)compact_blocks(n_items) + 1;
    unsigned char* p = static_cast<unsigned char*>(ws);
    CompactWs c;
    c.flag_a = p; p += up256((size_t)n_items);
    c.flag_b = p; p += up256((size_t)n_items);
    c.bsum_a = reinterpret_cast<i64*>(p); p += up256(nb * sizeof(i64));
    c.bsum_b = reinterpret_cast<i64*>(p);
    return c;
}

// sum of `v` over the block, returned to every thread
__device__ __forceinline__ int block_sum(int v, int* sh /*[4]*/) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// Item (row, thread) of a block sits at base + row*CB_NT + thread.  Given one flag per
// item, returns in rank[row] the number of set flags before the item in that order.
__device__ __forceinline__ void block_ordered_ranks(const bool (&flag)[CB_ROWS], int (&rank)[CB_ROWS],
                                                    int* sh /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int running = 0;
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
        unsigned long long b = __ballot(flag[r]);
        int before = __popcll(b & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) sh[wave] = __popcll(b);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int c = sh[w];
            woff += (w < wave) ? c : 0;
            tot += c;
        }
        rank[r] = running + woff + before;
        running += tot;
    }
}

// exclusive scan of bsum[0..nb) in place, total to bsum[nb]; one block
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(i64* __restrict__ a, i64* __restrict__ b,
                                                               i64 nb) {
    __shared__ i64 sh[1024];
    __shared__ i64 carry;
    i64* arrs[2] = {a, b};
    for (int which = 0; which < 2; ++which) {
        i64* v = arrs[which];
        if (v == nullptr) continue;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (i64 base = 0; base < nb; base += 1024) {
            i64 i = base + threadIdx.x;
            i64 x = (i < nb) ? v[i] : 0;
            sh[threadIdx.x] = x;
            __syncthreads();
            for (int o = 1; o < 1024; o <<= 1) {
                i64 t = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0;
                __syncthreads();
                sh[threadIdx.x] += t;
                __syncthreads();
            }
            i64 incl = sh[threadIdx.x];
            i64 c = carry;
            if (i < nb) v[i] = c + incl - x;
            __syncthreads();
            if (threadIdx.x == 1023) carry = c + incl;
            __syncthreads();
        }
        if (threadIdx.x == 0) v[nb] = carry;
        __syncthreads();
    }
}

// ------------------------------------------------------------------ find_mnn
__global__ __launch_bounds__(CB_NT) void mnn_flag_kernel(const int* __restrict__ g12, i64 n1, i64 ld12,
                                                         const int* __restrict__ g21, i64 n2, i64 ld21,
                                                         int ka, unsigned char* __restrict__ flag,
                                                         i64* __restrict__ bsum, int* __restrict__ status) {
    __shared__ int sh[4];
    const i64 n_items = n1 * ka;
    const i64 base = (i64)blockIdx.x * CB_ITEMS;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
        i64 e = base + r * CB_NT + threadIdx.x;
        if (e < n_items) {
            i64 i = e / ka;
            int j = (int)(e - i * ka);
            i64 y = g12[i * ld12 + j];
            if (y < 0) y += n2;  // numpy wraps a negative index once
            bool hit = false;
            if (y < 0 || y >= n2) {
                *status = 1;
            } else {
                const int* row = g21 + y * ld21;
                for (int t = 0; t < ka; ++t) hit |= ((i64)row[t] == i);
            }
            flag[e] = hit ? 1 : 0;
            cnt += hit ? 1 : 0;
        }
    }
    int tot = block_sum(cnt, sh);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(CB_NT) void mnn_scatter_kernel(const int* __restrict__ g12, i64 n1, i64 ld12,
                                                            int ka, const unsigned char* __restrict__ flag,
                                                            const i64* __restrict__ boff, i64 nb,
                                                            i64* __restrict__ pairs, i64* __restrict__ n_pairs) {
    __shared__ int sh[4];
    const i64 n_items = n1 * ka;
    const i64 base = (i64)blockIdx.x * CB_ITEMS;
    bool f[CB_ROWS];
    int rank[CB_ROWS];
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
        i64 e = base + r * CB_NT + threadIdx.x;
        f[r] = (e < n_items) && flag[e];
    }
    block_ordered_ranks(f, rank, sh);
    const i64 off = boff[blockIdx.x];
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
        if (f[r]) {
            i64 e = base + r * CB_NT + threadIdx.x;
            i64 i = e / ka;
            int j = (int)(e - i * ka);
            i64 p = off + rank[r];
            pairs[2 * p] = i;
            pairs[2 * p + 1] = (i64)g12[i * ld12 + j];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_pairs = boff[nb];
}

hipError_t launch_find_mnn(const int* g12, long long n1, long long ld12, const int* g21, long long n2,
                           long long ld21, int ka, long long* pairs, long long* n_pairs, int* status,
                           void* ws, hipStream_t stream) {
    const i64 n_items = n1 * ka;
    CompactWs c = carve(ws, n_items);
    const i64 nb = compact_blocks(n_items < 1 ? 1 : n_items);
    hipError_t e = hipMemsetAsync(status, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mnn_flag_kernel, dim3((unsigned)nb), dim3(CB_NT), 0, stream, g12, n1, ld12, g21, n2,
                       ld21, ka, c.flag_a, c.bsum_a, status);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, c.bsum_a, (i64*)nullptr, nb);
    hipLaunchKernelGGL(mnn_scatter_kernel, dim3((unsigned)nb), dim3(CB_NT), 0, stream, g12, n1, ld12, ka,
                       c.flag_a, c.bsum_a, nb, pairs, n_pairs);
    return hipGetLastError();
}

// ------------------------------------------------------------------ snn score
// One wave per anchor.  len(set(A) & set(B)) = number of positions p whose value is
// the first occurrence in A and occurs somewhere in B.
__global__ __launch_bounds__(256) void snn_score_kernel(const i64* __restrict__ pairs,
                                                        const i64* __restrict__ n_pairs,
                                                        const int* __restrict__ g11, const int* __restrict__ g12,
                                                        const int* __restrict__ g21, const int* __restrict__ g22,
                                                        i64 n1, i64 n2, i64 ld, int ks, int* __restrict__ score,
                                                        i64* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) int shm[];  // [4 waves][4 rows][ks] + hist[2*ks+1]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* A1 = shm + wave * 4 * ks;
    int* B1 = A1 + ks;
    int* A2 = B1 + ks;
    int* B2 = A2 + ks;
    int* lh = shm + 16 * ks;
    const int nbin = 2 * ks + 1;
    for (int i = threadIdx.x; i < nbin; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const i64 np = *n_pairs;
    for (i64 a = (i64)blockIdx.x * 4 + wave; a < np; a += (i64)gridDim.x * 4) {
        i64 x = pairs[2 * a], y = pairs[2 * a + 1];
        if (x < 0) x += n1;
        if (y < 0) y += n2;
        int total = -1;
        if (x >= 0 && x < n1 && y >= 0 && y < n2) {
            for (int t = lane; t < ks; t += 64) {
                A1[t] = g11[x * ld + t];
                B1[t] = g21[y * ld + t];
                A2[t] = g12[x * ld + t];
                B2[t] = g22[y * ld + t];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            int c = 0;
            for (int p = lane; p < ks; p += 64) {
                const int v1 = A1[p], v2 = A2[p];
                bool first1 = true, in1 = false, first2 = true, in2 = false;
                for (int t = 0; t < ks; ++t) {
                    int a1 = A1[t], b1 = B1[t], a2 = A2[t], b2 = B2[t];
                    first1 &= !(t < p && a1 == v1);
                    in1 |= (b1 == v1);
                    first2 &= !(t < p && a2 == v2);
                    in2 |= (b2 == v2);
                }
                c += (first1 && in1) ? 1 : 0;
                c += (first2 && in2) ? 1 : 0;
            }
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            total = c;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (lane == 0) {
            score[a] = total;
            if (hist && total >= 0) atomicAdd(&lh[total], 1);
        }
    }
    __syncthreads();
    if (hist)
        for (int i = threadIdx.x; i < nbin; i += blockDim.x)
            if (lh[i]) atomicAdd(reinterpret_cast<unsigned long long*>(hist + i), (unsigned long long)lh[i]);
}

hipError_t launch_snn_score(const long long* pairs, const long long* n_pairs, long long n_pairs_max,
                            const int* g11, const int* g12, const int* g21, const int* g22,
                            long long n1, long long n2, long long ld, int ks, int* score,
                            long long* hist, hipStream_t stream) {
    if (hist) {
        hipError_t e = hipMemsetAsync(hist, 0, sizeof(long long) * (2 * ks + 1), stream);
        if (e != hipSuccess) return e;
    }
    long long blocks = (n_pairs_max + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    size_t lds = sizeof(int) * (size_t)(16 * ks + 2 * ks + 1);
    hipLaunchKernelGGL(snn_score_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, pairs, n_pairs, g11,
                       g12, g21, g22, n1, n2, ld, ks, score, hist);
    return hipGetLastError();
}

// ------------------------------------------------------------------ score tail + label filter
__device__ __forceinline__ double norm_score(int raw, double p_lo, double p_hi) {
    // (tmp - tmin) / (tmax - tmin); tmp[tmp > 1] = 1; tmp[tmp < 0] = 0  -- NaN stays NaN
    double s = ((double)raw - p_lo) / (p_hi - p_lo);
    if (s > 1.0) s = 1.0;
    if (s < 0.0) s = 0.0;
    return s;
}

__global__ __launch_bounds__(CB_NT) void filter_flag_kernel(
    const i64* __restrict__ pairs, const i64* __restrict__ n_pairs, const int* __restrict__ raw, double p_lo,
    double p_hi, double thr, const int* __restrict__ codes1, i64 n1, const int* __restrict__ codes2, i64 n2,
    int ct_filter, unsigned char* __restrict__ flag_pass, unsigned char* __restrict__ flag_keep,
    i64* __restrict__ bsum_pass, i64* __restrict__ bsum_keep, double* __restrict__ all_score) {
    __shared__ int sh[4];
    const i64 np = *n_pairs;
    const i64 base = (i64)blockIdx.x * CB_ITEMS;
    int cp = 0, ck = 0;
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
        i64 e = base + r * CB_NT + threadIdx.x;
        if (e < np) {
            double s = norm_score(raw[e], p_lo, p_hi);
            i64 x = pairs[2 * e], y = pairs[2 * e + 1];
            if (x < 0) x += n1;
            if (y < 0) y += n2;
            bool same = (x >= 0 && x < n1 && y >= 0 && y < n2) && (codes1[x] == codes2[y]);
            bool pass = s > thr;
            bool keep = pass && (!ct_filter || same);
            flag_pass[e] = (unsigned char)((pass ? 1 : 0) | (same ? 2 : 0));
            flag_keep[e] = keep ? 1 : 0;
            if (all_score) all_score[e] = s;
            cp += pass ? 1 : 0;
            ck += keep ? 1 : 0;
        }
    }
    int tp = block_sum(cp, sh);
    int tk = block_sum(ck, sh);
    if (threadIdx.x == 0) {
        bsum_pass[blockIdx.x] = tp;
        bsum_keep[blockIdx.x] = tk;
    }
}

__global__ __launch_bounds__(CB_NT) void filter_scatter_kernel(
    const i64* __restrict__ pairs, const i64* __restrict__ n_pairs, const int* __restrict__ raw, double p_lo,
    double p_hi, const unsigned char* __restrict__ flag_pass, const unsigned char* __restrict__ flag_keep,
    const i64* __restrict__ boff_pass, const i64* __restrict__ boff_keep, i64 nb, i64* __restrict__ out_pairs,
    double* __restrict__ out_score, unsigned char* __restrict__ out_same, i64* __restrict__ out_index,
    i64* __restrict__ counts) {
    __shared__ int sh[4];
    const i64 np = *n_pairs;
    const i64 base = (i64)blockIdx.x * CB_ITEMS;
    bool fp[CB_ROWS], fk[CB_ROWS], same[CB_ROWS];
    int rp[CB_ROWS], rk[CB_ROWS];
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
        i64 e = base + r * CB_NT + threadIdx.x;
        unsigned char a = (e < np) ? flag_pass[e] : 0;
        fp[r] = a & 1;
        same[r] = (a & 2) != 0;
        fk[r] = (e < np) && flag_keep[e];
    }
    block_ordered_ranks(fp, rp, sh);
    block_ordered_ranks(fk, rk, sh);
    const i64 op = boff_pass[blockIdx.x], ok = boff_keep[blockIdx.x];
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
        if (fk[r]) {
            i64 e = base + r * CB_NT + threadIdx.x;
            i64 p = ok + rk[r];
            out_pairs[2 * p] = pairs[2 * e];
            out_pairs[2 * p + 1] = pairs[2 * e + 1];
            out_score[p] = norm_score(raw[e], p_lo, p_hi);
            out_same[p] = same[r] ? 1 : 0;
            out_index[p] = op + rp[r];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counts[0] = boff_pass[nb];
        counts[1] = boff_keep[nb];
    }
}

hipError_t launch_anchor_filter(const long long* pairs, const long long* n_pairs, long long n_pairs_max,
                                const int* raw, double p_lo, double p_hi, double thr, const int* codes1,
                                long long n1, const int* codes2, long long n2, int ct_filter,
                                long long* out_pairs, double* out_score, unsigned char* out_same,
                                long long* out_index, long long* counts, double* all_score, void* ws,
                                hipStream_t stream) {
    CompactWs c = carve(ws, n_pairs_max);
    const i64 nb = compact_blocks(n_pairs_max < 1 ? 1 : n_pairs_max);
    hipLaunchKernelGGL(filter_flag_kernel, dim3((unsigned)nb), dim3(CB_NT), 0, stream, pairs, n_pairs, raw, p_lo,
                       p_hi, thr, codes1, n1, codes2, n2, ct_filter, c.flag_a, c.flag_b, c.bsum_a, c.bsum_b,
                       all_score);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, c.bsum_a, c.bsum_b, nb);
    hipLaunchKernelGGL(filter_scatter_kernel, dim3((unsigned)nb), dim3(CB_NT), 0, stream, pairs, n_pairs, raw,
                       p_lo, p_hi, c.flag_a, c.flag_b, c.bsum_a, c.bsum_b, nb, out_pairs, out_score, out_same,
                       out_index, counts);
    return hipGetLastError();
}

// ------------------------------------------------------------------ LU row permutation
// perm[0..n) <- identity (all blocks), then LAPACK's row interchanges in order:
// swap(perm[i], perm[piv[i]-1]) for i = 0..m-1 (m is the number of columns of a tall block: a few
// dozen).  The swaps depend on each other, so one thread of a second, one-block launch applies
// them.  A pivot outside 1..n is no LAPACK pivot (a factorisation of NaNs leaves such entries):
// it is skipped and *status becomes 1, so that the caller can refuse the result.
__global__ __launch_bounds__(256) void identity_perm_kernel(long long n, long long* __restrict__ perm) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        perm[i] = i;
}
__global__ void lu_row_swaps_kernel(const int* __restrict__ piv, int m, long long n, long long* __restrict__ perm,
                                    int* __restrict__ status) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    bool bad = false;
    for (int i = 0; i < m; ++i) {
        const long long p = (long long)piv[i] - 1;
        if (p < 0 || p >= n) { bad = true; continue; }
        const long long a = perm[i], b = perm[p];
        perm[i] = b;
        perm[p] = a;
    }
    if (bad && status != nullptr) *status = 1;
}

hipError_t launch_lu_row_perm(const int* piv, int m, long long n, long long* perm, int* status, hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(identity_perm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, perm);
    hipLaunchKernelGGL(lu_row_swaps_kernel, dim3(1), dim3(64), 0, stream, piv, m, n, perm, status);
    return hipGetLastError();
}

}  // namespace scmiac
