// Symmetric NT-Xent loss between the anchor cells' two embeddings (the consumer of the
// anchor table; scmiac/modeling/loss.py:85-133), forward and backward, one launch each.
//
//   zn = z / max(|z|, 1e-12)            S = zn_i . zn_j^T / T
//   loss = 0.5 * ( mean_t(lse_j S[t,j] - S[t,t]) + mean_t(lse_i S[i,t] - S[t,t]) )
//
// Work split: item t (one anchor pair) owns row t and column t of S.  A block has 4
// waves x 64 items and sees one slice of the other side's rows (grid y: up to 16 slices, so
// that a batch of 256 already makes 64 blocks); the slice passes through LDS in chunks of
// JT rows, each wave taking a quarter of every chunk (LDS reads of a row are broadcasts),
// and the four partial results of an item meet in LDS.  A second, small kernel lets the
// slices of an item meet.  S is recomputed in the backward pass instead of being stored.
//
// From a batch of kMfmaMinBatch items on, the first kernel of each pass is the MFMA one
// (ntxent2_*_mfma_kernel): a wave owns 32 items as the B operand of v_mfma_f32_32x32x2f32 (lane
// <-> item), the other side's rows are the A operand out of LDS, so a lane ends up with 16
// entries of its own item's row (or column) of S and the log-sum-exp stays lane-local.  In the
// backward pass the weights p of a 32 x 32 tile are, register for register, already laid out as
// the B operand of the second product (gradient = sum_j p[j] zn[j]).  Both write the same
// per-slice partial results as the FMA kernels; the finishing kernels are shared.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace scmiac {

namespace {

constexpr int kItems = 64;     // items per block = lanes of a wave
constexpr int kWaves = 4;
constexpr float kNormEps = 1e-12f;

template <int DP>
struct NtCfg {
    static constexpr int JT = (DP <= 32) ? 128 : 64;                 // rows per chunk and side
    static constexpr int kChunkFloats = 2 * JT * DP + 2 * JT;        // rows of both sides + their two lse values
    static constexpr int kRedFloats = kWaves * kItems * DP;          // per-wave partial gradients of one side
    static constexpr int kLdsFloats = (kChunkFloats > kRedFloats) ? kChunkFloats : kRedFloats;
};

// own row of item t, normalised, in registers; returns 1 / max(|z|, eps)
template <int DP>
__device__ inline float load_own_row(const float* __restrict__ z, int t, int d, float (&row)[DP]) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < DP; ++c) {
        row[c] = (c < d) ? z[(size_t)t * d + c] : 0.f;
        ss = fmaf(row[c], row[c], ss);
    }
    float inv = 1.f / fmaxf(sqrtf(ss), kNormEps);
#pragma unroll
    for (int c = 0; c < DP; ++c) row[c] *= inv;
    return inv;
}

// rows [c0, c0+JT) of both sides, normalised, into LDS (one row per thread)
template <int DP>
__device__ inline void stage_chunk(const float* __restrict__ zi, const float* __restrict__ zj, int B, int d, int c0,
                                   float* lds) {
    constexpr int JT = NtCfg<DP>::JT;
    for (int r = threadIdx.x; r < 2 * JT; r += blockDim.x) {
        int side = r / JT, j = c0 + (r % JT);
        float* dst = lds + (size_t)r * DP;
        if (j < B) {
            const float* src = (side ? zj : zi) + (size_t)j * d;
            float ss = 0.f;
            for (int c = 0; c < d; ++c) ss = fmaf(src[c], src[c], ss);
            float inv = 1.f / fmaxf(sqrtf(ss), kNormEps);
            for (int c = 0; c < DP; ++c) dst[c] = (c < d) ? src[c] * inv : 0.f;
        } else {
            for (int c = 0; c < DP; ++c) dst[c] = 0.f;
        }
    }
}

template <int DP>
__device__ inline float dot_lds(const float (&a)[DP], const float* row) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DP; c += 4) {
        float4 v = *reinterpret_cast<const float4*>(row + c);
        s = fmaf(a[c], v.x, s);
        s = fmaf(a[c + 1], v.y, s);
        s = fmaf(a[c + 2], v.z, s);
        s = fmaf(a[c + 3], v.w, s);
    }
    return s;
}

__device__ inline void lse_push(float s, float& m, float& sum) {
    if (s > m) {
        sum = sum * expf(m - s) + 1.f;
        m = s;
    } else {
        sum += expf(s - m);
    }
}

// Forward, block (x, y): items 64 x .. 64 x + 63 against the rows of slice y of both sides.
// part[y][t] = (max, sum exp) of row t of S over the slice's columns, and of column t over
// the slice's rows.
template <int DP>
__global__ __launch_bounds__(kItems* kWaves) void ntxent2_forward_kernel(
    const float* __restrict__ zi, const float* __restrict__ zj, int B, int d, float inv_t, int slice_len,
    float4* __restrict__ part) {
    constexpr int JT = NtCfg<DP>::JT;
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * kItems + lane;
    const int tc = (t < B) ? t : B - 1;
    const int j_lo = blockIdx.y * slice_len;
    const int j_hi = (j_lo + slice_len < B) ? j_lo + slice_len : B;
    float a[DP], b[DP];
    load_own_row<DP>(zi, tc, d, a);
    load_own_row<DP>(zj, tc, d, b);

    float mr = -INFINITY, sr = 0.f, mc = -INFINITY, sc = 0.f;
    for (int c0 = j_lo; c0 < j_hi; c0 += JT) {
        __syncthreads();
        stage_chunk<DP>(zi, zj, j_hi, d, c0, lds);
        __syncthreads();
        int j0 = w * (JT / kWaves), j1 = j0 + JT / kWaves;
        if (c0 + j1 > j_hi) j1 = j_hi - c0;
        for (int j = j0; j < j1; ++j) {
            lse_push(dot_lds<DP>(a, lds + (size_t)(JT + j) * DP) * inv_t, mr, sr);  // S[t, c0+j]
            lse_push(dot_lds<DP>(b, lds + (size_t)j * DP) * inv_t, mc, sc);         // S[c0+j, t]
        }
    }
    __syncthreads();
    float4* meet = reinterpret_cast<float4*>(lds);
    meet[w * kItems + lane] = make_float4(mr, sr, mc, sc);
    __syncthreads();
    if (w == 0 && t < B) {
        float M = -INFINITY, Mc = -INFINITY;
        for (int q = 0; q < kWaves; ++q) {
            float4 p = meet[q * kItems + lane];
            M = fmaxf(M, p.x);
            Mc = fmaxf(Mc, p.z);
        }
        float S = 0.f, Sc = 0.f;
        for (int q = 0; q < kWaves; ++q) {
            float4 p = meet[q * kItems + lane];
            if (p.y > 0.f) S += p.y * expf(p.x - M);
            if (p.w > 0.f) Sc += p.w * expf(p.z - Mc);
        }
        part[(size_t)blockIdx.y * B + t] = make_float4(M, S, Mc, Sc);
    }
}

// Forward, second step: the slices of an item meet; log-sum-exp of its row and column, its
// share of the loss; one partial loss per block of 256 items.
__global__ __launch_bounds__(256) void ntxent2_forward_finish_kernel(
    const float* __restrict__ zi, const float* __restrict__ zj, int B, int d, float inv_t, int n_slices,
    const float4* __restrict__ part, float* __restrict__ lse_row, float* __restrict__ lse_col,
    float* __restrict__ loss_parts) {
    __shared__ float red[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    float contrib = 0.f;
    if (t < B) {
        float M = -INFINITY, Mc = -INFINITY;
        for (int y = 0; y < n_slices; ++y) {
            float4 p = part[(size_t)y * B + t];
            if (p.y > 0.f) M = fmaxf(M, p.x);
            if (p.w > 0.f) Mc = fmaxf(Mc, p.z);
        }
        float S = 0.f, Sc = 0.f;
        for (int y = 0; y < n_slices; ++y) {
            float4 p = part[(size_t)y * B + t];
            if (p.y > 0.f) S += p.y * expf(p.x - M);
            if (p.w > 0.f) Sc += p.w * expf(p.z - Mc);
        }
        const float lr = M + logf(S), lc = Mc + logf(Sc);
        const float* a = zi + (size_t)t * d;
        const float* b = zj + (size_t)t * d;
        // one pass over the two rows, eight features' loads in flight at a time (a loop that waits
        // for every load before the next costs more than the rest of the kernel)
        float na = 0.f, nb = 0.f, ab = 0.f;
#pragma unroll 8
        for (int c = 0; c < d; ++c) {
            const float x = a[c], y = b[c];
            na = fmaf(x, x, na);
            nb = fmaf(y, y, nb);
            ab = fmaf(x, y, ab);
        }
        const float ia = 1.f / fmaxf(sqrtf(na), kNormEps), ib = 1.f / fmaxf(sqrtf(nb), kNormEps);
        const float diag = ab * ia * ib * inv_t;
        lse_row[t] = lr;
        lse_col[t] = lc;
        contrib = 0.5f * ((lr - diag) + (lc - diag));
    }
    for (int off = 32; off; off >>= 1) contrib += __shfl_xor(contrib, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) loss_parts[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
}

// Backward, block (x, y): the gradient with respect to the normalised rows of items
// 64 x .. 64 x + 63, summed over slice y: gpart[y][side][t][0..DP).
template <int DP>
__global__ __launch_bounds__(kItems* kWaves) void ntxent2_backward_kernel(
    const float* __restrict__ zi, const float* __restrict__ zj, int B, int d, float inv_t, int slice_len,
    const float* __restrict__ lse_row, const float* __restrict__ lse_col, const float* __restrict__ grad_out,
    float* __restrict__ gpart) {
    constexpr int JT = NtCfg<DP>::JT;
    extern __shared__ __align__(16) float lds[];
    float* lse_chunk = lds + 2 * JT * DP;  // [0, JT): lse_row of the chunk's rows, [JT, 2JT): lse_col
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * kItems + lane;
    const int tc = (t < B) ? t : B - 1;
    const int j_lo = blockIdx.y * slice_len;
    const int j_hi = (j_lo + slice_len < B) ? j_lo + slice_len : B;
    float a[DP], b[DP], ga[DP], gb[DP];
    load_own_row<DP>(zi, tc, d, a);
    load_own_row<DP>(zj, tc, d, b);
#pragma unroll
    for (int c = 0; c < DP; ++c) ga[c] = gb[c] = 0.f;
    const float lr_t = lse_row[tc], lc_t = lse_col[tc];
    const float coef = 0.5f * grad_out[0] / (float)B * inv_t;

    for (int c0 = j_lo; c0 < j_hi; c0 += JT) {
        __syncthreads();
        stage_chunk<DP>(zi, zj, j_hi, d, c0, lds);
        for (int r = threadIdx.x; r < 2 * JT; r += blockDim.x) {
            int j = c0 + (r % JT);
            lse_chunk[r] = (j < j_hi) ? ((r < JT) ? lse_row[j] : lse_col[j]) : 0.f;
        }
        __syncthreads();
        int j0 = w * (JT / kWaves), j1 = j0 + JT / kWaves;
        if (c0 + j1 > j_hi) j1 = j_hi - c0;
        for (int j = j0; j < j1; ++j) {
            const float* rj = lds + (size_t)(JT + j) * DP;  // zn_j[c0+j]
            const float* ri = lds + (size_t)j * DP;         // zn_i[c0+j]
            float same = (c0 + j == t) ? 2.f : 0.f;
            float s = dot_lds<DP>(a, rj) * inv_t;           // S[t, c0+j]
            float p = coef * (expf(s - lr_t) + expf(s - lse_chunk[JT + j]) - same);
            float s2 = dot_lds<DP>(b, ri) * inv_t;          // S[c0+j, t]
            float p2 = coef * (expf(s2 - lse_chunk[j]) + expf(s2 - lc_t) - same);
#pragma unroll
            for (int c = 0; c < DP; c += 4) {
                float4 vj = *reinterpret_cast<const float4*>(rj + c);
                float4 vi = *reinterpret_cast<const float4*>(ri + c);
                ga[c] = fmaf(p, vj.x, ga[c]);
                ga[c + 1] = fmaf(p, vj.y, ga[c + 1]);
                ga[c + 2] = fmaf(p, vj.z, ga[c + 2]);
                ga[c + 3] = fmaf(p, vj.w, ga[c + 3]);
                gb[c] = fmaf(p2, vi.x, gb[c]);
                gb[c + 1] = fmaf(p2, vi.y, gb[c + 1]);
                gb[c + 2] = fmaf(p2, vi.z, gb[c + 2]);
                gb[c + 3] = fmaf(p2, vi.w, gb[c + 3]);
            }
        }
    }
    // the four partial gradients of an item meet in LDS, one side after the other;
    // column-major per wave ([w][c][lane]) so that neighbouring lanes hit neighbouring banks
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        __syncthreads();
        float(&g)[DP] = side ? gb : ga;
#pragma unroll
        for (int c = 0; c < DP; ++c) lds[((size_t)w * DP + c) * kItems + lane] = g[c];
        __syncthreads();
        if (w == 0 && t < B) {
            float* out = gpart + (((size_t)blockIdx.y * 2 + side) * B + t) * DP;
#pragma unroll
            for (int c = 0; c < DP; c += 4) {
                float4 v;
                float* e = &v.x;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float acc = 0.f;
                    for (int q = 0; q < kWaves; ++q) acc += lds[((size_t)q * DP + c + u) * kItems + lane];
                    e[u] = acc;
                }
                *reinterpret_cast<float4*>(out + c) = v;
            }
        }
    }
}

// Backward, second step: thread (t, side) adds up the slices and takes the gradient through
// the row normalisation: dz = (g - zn (zn . g)) / max(|z|, eps).
template <int DP>
__global__ __launch_bounds__(128) void ntxent2_backward_finish_kernel(
    const float* __restrict__ zi, const float* __restrict__ zj, int B, int d, int n_slices,
    const float* __restrict__ gpart, float* __restrict__ dzi, float* __restrict__ dzj) {
    const int id = blockIdx.x * 128 + threadIdx.x;
    if (id >= 2 * B) return;
    const int side = id / B, t = id - side * B;
    float zn[DP], g[DP];
    const float inv = load_own_row<DP>(side ? zj : zi, t, d, zn);
#pragma unroll
    for (int c = 0; c < DP; ++c) g[c] = 0.f;
    for (int y = 0; y < n_slices; ++y) {
        const float4* src = reinterpret_cast<const float4*>(gpart + (((size_t)y * 2 + side) * B + t) * DP);
#pragma unroll
        for (int c = 0; c < DP / 4; ++c) {
            float4 v = src[c];
            g[4 * c] += v.x; g[4 * c + 1] += v.y; g[4 * c + 2] += v.z; g[4 * c + 3] += v.w;
        }
    }
    float proj = 0.f;
#pragma unroll
    for (int c = 0; c < DP; ++c) proj = fmaf(zn[c], g[c], proj);
    float* out = (side ? dzj : dzi) + (size_t)t * d;
#pragma unroll
    for (int c = 0; c < DP; ++c)
        if (c < d) out[c] = (g[c] - zn[c] * proj) * inv;
}


// ------------------------------------------------------------------ MFMA path (large batches)
typedef float nt_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMfmaMinBatch = 512;  // below this the FMA kernels' 64-item blocks fill the GPU better
constexpr int kOwn = 128;           // items per block: 4 waves x 32

template <int DP>
struct NtM {
    static constexpr int JT = 128;                               // rows of the other side per chunk
    static constexpr int LS = DP + 1;                            // odd row stride: column reads hit distinct banks
    static constexpr int kLdsFloats = JT * LS + JT;              // the rows + their lse values
};

// Both sides' rows, normalised and zero-padded to DP features: zn[side][row][DP].  Sixteen lanes
// share a row (independent loads, one shuffle reduction), so a batch of thousands takes microseconds.
template <int DP>
__global__ __launch_bounds__(256) void ntxent2_normalise_kernel(const float* __restrict__ zi,
                                                                const float* __restrict__ zj, int B, int d,
                                                                float* __restrict__ zn) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;  // row of [2 B]
    if (row >= 2 * B) return;  // (whole 16-lane groups leave together)
    const float* src = (row < B ? zi : zj) + (size_t)(row < B ? row : row - B) * d;
    float v[DP / 16];
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < DP / 16; ++u) {
        const int c = sub + 16 * u;
        v[u] = (c < d) ? src[c] : 0.f;
        ss = fmaf(v[u], v[u], ss);
    }
#pragma unroll
    for (int off = 8; off; off >>= 1) ss += __shfl_xor(ss, off);
    const float inv = 1.f / fmaxf(sqrtf(ss), kNormEps);
#pragma unroll
    for (int u = 0; u < DP / 16; ++u) zn[(size_t)row * DP + sub + 16 * u] = v[u] * inv;
}

// rows [c0, c0+JT) of one side (already normalised) into LDS with row stride LS; rows from j_hi on are zero
template <int DP>
__device__ inline void stage_chunk_m(const float* __restrict__ zn_other, int j_hi, int c0, float* lds) {
    constexpr int JT = NtM<DP>::JT, LS = NtM<DP>::LS;
    constexpr int V4 = DP / 4;  // float4 per row
    for (int e = threadIdx.x; e < JT * V4; e += 256) {
        const int r = e / V4, q = e - r * V4;
        const int j = c0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < j_hi) v = *reinterpret_cast<const float4*>(zn_other + (size_t)j * DP + 4 * q);
        float* dst = lds + (size_t)r * LS + 4 * q;  // (odd row stride: scalar stores)
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}

// B operand of the score product: features 2 s + h of item t's normalised row, s = 0 .. DP/2 - 1
template <int DP>
__device__ inline void load_own_operand(const float* __restrict__ zn_side, int t, int h, float (&q)[DP / 2]) {
    const float2* src = reinterpret_cast<const float2*>(zn_side + (size_t)t * DP);
#pragma unroll
    for (int s = 0; s < DP / 2; ++s) {
        const float2 v = src[s];
        q[s] = h ? v.y : v.x;
    }
}

// scores of the 32 rows at `rows` (LDS, stride LS) against the wave's 32 items: register r of lane
// (n, h) = row (r & 3) + 8 (r >> 2) + 4 h  .  item n
template <int DP>
__device__ inline nt_f32x16 score_tile(const float* rows, const float (&q)[DP / 2], int n, int h) {
    constexpr int LS = NtM<DP>::LS;
    nt_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* a = rows + (size_t)n * LS + h;
#pragma unroll
    for (int s = 0; s < DP / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s], q[s], acc, 0, 0, 0);
    return acc;
}

// running (max, sum exp) of a lane over the 16 scores of a tile; `valid` = rows of the tile that exist
__device__ inline void lse_tile(const nt_f32x16& acc, float inv_t, int h, int valid, float& m, float& sum) {
    float v[16];
    float tm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        v[r] = (row < valid) ? acc[r] * inv_t : -INFINITY;
        tm = fmaxf(tm, v[r]);
    }
    if (tm > m) {  // (also the first tile with a valid row: m = -inf, sum = 0)
        sum *= expf(m - tm);
        m = tm;
    }
    if (m > -INFINITY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += expf(v[r] - m);
    }
}

// Forward, block (x, y, side): items 128 x .. 128 x + 127 of `side` (0: rows of S, own rows from
// z_i; 1: columns of S, own rows from z_j) against the rows of slice y of the other side.
// part[y][t] = (max, sum exp of row t; max, sum exp of column t): each side writes its half.
template <int DP>
__global__ __launch_bounds__(256) void ntxent2_forward_mfma_kernel(
    const float* __restrict__ zn, int B, float inv_t, int slice_len, float4* __restrict__ part) {
    constexpr int JT = NtM<DP>::JT, LS = NtM<DP>::LS;
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int side = blockIdx.z;
    const int t = blockIdx.x * kOwn + w * 32 + n;
    const int tc = (t < B) ? t : B - 1;
    const int j_lo = blockIdx.y * slice_len;
    const int j_hi = (j_lo + slice_len < B) ? j_lo + slice_len : B;
    const float* zn_other = zn + (size_t)(1 - side) * B * DP;
    float q[DP / 2];
    load_own_operand<DP>(zn + (size_t)side * B * DP, tc, h, q);

    float m = -INFINITY, sum = 0.f;
    for (int c0 = j_lo; c0 < j_hi; c0 += JT) {
        __syncthreads();
        stage_chunk_m<DP>(zn_other, j_hi, c0, lds);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < JT / 32; ++u) {
            const int valid = j_hi - (c0 + 32 * u);  // wave-uniform
            if (valid <= 0) break;
            lse_tile(score_tile<DP>(lds + (size_t)(32 * u) * LS, q, n, h), inv_t, h, valid, m, sum);
        }
    }
    // the two lane halves of an item saw different rows: let them meet
    const float m2 = __shfl_xor(m, 32), sum2 = __shfl_xor(sum, 32);
    if (h == 0 && t < B) {
        const float M = fmaxf(m, m2);
        float S = 0.f;
        if (sum > 0.f) S += sum * expf(m - M);
        if (sum2 > 0.f) S += sum2 * expf(m2 - M);
        float2* out = reinterpret_cast<float2*>(part + (size_t)blockIdx.y * B + t);
        out[side] = make_float2(M, S);
    }
}

// Forward, second step of the MFMA path: as ntxent2_forward_finish_kernel, with the diagonal
// S[t, t] taken from the normalised rows (compile-time width: sixteen-byte loads, all in flight
// together, where the other kernel walks the raw rows feature by feature).
template <int DP>
__global__ __launch_bounds__(256) void ntxent2_forward_finish_zn_kernel(
    const float* __restrict__ zn, int B, float inv_t, int n_slices, const float4* __restrict__ part,
    float* __restrict__ lse_row, float* __restrict__ lse_col, float* __restrict__ loss_parts) {
    __shared__ float red[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    float contrib = 0.f;
    if (t < B) {
        float M = -INFINITY, Mc = -INFINITY;
        for (int y = 0; y < n_slices; ++y) {
            float4 p = part[(size_t)y * B + t];
            if (p.y > 0.f) M = fmaxf(M, p.x);
            if (p.w > 0.f) Mc = fmaxf(Mc, p.z);
        }
        float S = 0.f, Sc = 0.f;
        for (int y = 0; y < n_slices; ++y) {
            float4 p = part[(size_t)y * B + t];
            if (p.y > 0.f) S += p.y * expf(p.x - M);
            if (p.w > 0.f) Sc += p.w * expf(p.z - Mc);
        }
        const float lr = M + logf(S), lc = Mc + logf(Sc);
        const float4* a = reinterpret_cast<const float4*>(zn + (size_t)t * DP);
        const float4* b = reinterpret_cast<const float4*>(zn + ((size_t)B + t) * DP);
        float ab = 0.f;
#pragma unroll
        for (int c = 0; c < DP / 4; ++c) {
            const float4 x = a[c], y = b[c];
            ab = fmaf(x.x, y.x, ab); ab = fmaf(x.y, y.y, ab); ab = fmaf(x.z, y.z, ab); ab = fmaf(x.w, y.w, ab);
        }
        const float diag = ab * inv_t;
        lse_row[t] = lr;
        lse_col[t] = lc;
        contrib = 0.5f * ((lr - diag) + (lc - diag));
    }
    for (int off = 32; off; off >>= 1) contrib += __shfl_xor(contrib, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) loss_parts[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
}

// gradient product of one tile: g[mb] += sum over the tile's 32 rows of p[row] * zn[row][32 mb + c].
// The weights sit in the score tile's layout, which IS the B operand layout step by step: register
// r of lane (n, h) belongs to row (r & 3) + 8 (r >> 2) + 4 h, i.e. k = h of MFMA step r.
template <int DP>
__device__ inline void grad_tile(const float* rows, const nt_f32x16& p, int n, int h, nt_f32x16 (&g)[DP / 32]) {
    constexpr int LS = NtM<DP>::LS;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* a = rows + (size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * LS + n;  // A[feature n][k = h]
#pragma unroll
        for (int mb = 0; mb < DP / 32; ++mb)
            g[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[32 * mb], p[r], g[mb], 0, 0, 0);
    }
}

// Backward, block (x, y, side): gradient with respect to the normalised own rows of `side`, summed
// over slice y of the other side: gpart[y][side][t][0..DP).  With s = S[t, j] (side 0) or S[j, t]
// (side 1) the weight of the other side's row j is coef (exp(s - lse_own[t]) + exp(s - lse_other[j])
// - 2 [j == t]), lse_own / lse_other = lse_row / lse_col for side 0 and the other way round for side 1.
template <int DP>
__global__ __launch_bounds__(256) void ntxent2_backward_mfma_kernel(
    const float* __restrict__ zn, int B, float inv_t, int slice_len,
    const float* __restrict__ lse_row, const float* __restrict__ lse_col, const float* __restrict__ grad_out,
    float* __restrict__ gpart) {
    constexpr int JT = NtM<DP>::JT, LS = NtM<DP>::LS;
    extern __shared__ __align__(16) float lds[];
    float* lse_chunk = lds + JT * LS;  // lse_other of the chunk's rows
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int side = blockIdx.z;
    const int t = blockIdx.x * kOwn + w * 32 + n;
    const int tc = (t < B) ? t : B - 1;
    const int j_lo = blockIdx.y * slice_len;
    const int j_hi = (j_lo + slice_len < B) ? j_lo + slice_len : B;
    const float* zn_other = zn + (size_t)(1 - side) * B * DP;
    const float* lse_own = side ? lse_col : lse_row;
    const float* lse_other = side ? lse_row : lse_col;
    float q[DP / 2];
    load_own_operand<DP>(zn + (size_t)side * B * DP, tc, h, q);
    nt_f32x16 g[DP / 32];
#pragma unroll
    for (int mb = 0; mb < DP / 32; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[mb][r] = 0.f;
    const float l_t = lse_own[tc];
    const float coef = 0.5f * grad_out[0] / (float)B * inv_t;

    for (int c0 = j_lo; c0 < j_hi; c0 += JT) {
        __syncthreads();
        stage_chunk_m<DP>(zn_other, j_hi, c0, lds);
        if (threadIdx.x < JT) {
            const int j = c0 + threadIdx.x;
            lse_chunk[threadIdx.x] = (j < j_hi) ? lse_other[j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < JT / 32; ++u) {
            if (j_hi - (c0 + 32 * u) <= 0) break;  // wave-uniform; rows beyond j_hi inside a tile are zero rows
            const float* rows = lds + (size_t)(32 * u) * LS;
            nt_f32x16 p = score_tile<DP>(rows, q, n, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ju = 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float s = p[r] * inv_t;
                const float same = (c0 + ju == t) ? 2.f : 0.f;
                p[r] = coef * (expf(s - l_t) + expf(s - lse_chunk[ju]) - same);
            }
            grad_tile<DP>(rows, p, n, h, g);
        }
    }
    // register r of lane (n, h), block mb: feature 32 mb + (r & 3) + 8 (r >> 2) + 4 h of item n
    if (t < B) {
        float* out = gpart + (((size_t)blockIdx.y * 2 + side) * B + t) * DP;
#pragma unroll
        for (int mb = 0; mb < DP / 32; ++mb)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<float4*>(out + 32 * mb + 8 * k + 4 * h) =
                    make_float4(g[mb][4 * k], g[mb][4 * k + 1], g[mb][4 * k + 2], g[mb][4 * k + 3]);
    }
}

// how the other side's rows are cut into slices: enough blocks to fill the GPU
struct NtPlan {
    int blocks_x, n_slices, slice_len;
    bool mfma;
};
inline NtPlan nt_plan(int B) {
    NtPlan p;
    p.mfma = B >= kMfmaMinBatch;
    const int own = p.mfma ? kOwn : kItems;
    p.blocks_x = (B + own - 1) / own;
    // (the MFMA grid has a block per side as well: 256 blocks in x and y make 512 in all, two or
    // three per compute unit, so that one block's staging and barriers hide behind another's products)
    const int want = 256;
    int s = (p.blocks_x >= want) ? 1 : (want + p.blocks_x - 1) / p.blocks_x;
    if (s > 16) s = 16;
    p.slice_len = (B + s - 1) / s;
    p.n_slices = (B + p.slice_len - 1) / p.slice_len;
    return p;
}

// floats of the per-slice partial results (forward: 4 per item and slice, backward: 2 DP); the MFMA
// path keeps the normalised rows (2 B DP floats) behind them
inline size_t nt_partial_floats(const NtPlan& p, int B, int dp) {
    const size_t fwd = (size_t)p.n_slices * B * 4, bwd = (size_t)p.n_slices * 2 * B * dp;
    return fwd > bwd ? fwd : bwd;
}

template <int DP>
hipError_t launch_fwd(const float* zi, const float* zj, int B, int d, float inv_t, float* lse_row, float* lse_col,
                      float* loss_parts, float* ws, hipStream_t stream) {
    const NtPlan p = nt_plan(B);
    size_t lds = (size_t)NtCfg<DP>::kLdsFloats * sizeof(float);
    float4* part = reinterpret_cast<float4*>(ws);
    if (p.mfma) {
        float* zn = ws + nt_partial_floats(p, B, DP);  // behind the per-slice partial results
        ntxent2_normalise_kernel<DP><<<(2 * B + 15) / 16, 256, 0, stream>>>(zi, zj, B, d, zn);
        ntxent2_forward_mfma_kernel<DP><<<dim3(p.blocks_x, p.n_slices, 2), 256, NtM<DP>::kLdsFloats * sizeof(float), stream>>>(
            zn, B, inv_t, p.slice_len, part);
        ntxent2_forward_finish_zn_kernel<DP><<<(B + 255) / 256, 256, 0, stream>>>(zn, B, inv_t, p.n_slices, part, lse_row,
                                                                                lse_col, loss_parts);
        return hipGetLastError();
    } else
        ntxent2_forward_kernel<DP><<<dim3(p.blocks_x, p.n_slices), kItems * kWaves, lds, stream>>>(
            zi, zj, B, d, inv_t, p.slice_len, part);
    ntxent2_forward_finish_kernel<<<(B + 255) / 256, 256, 0, stream>>>(zi, zj, B, d, inv_t, p.n_slices, part,
                                                                      lse_row, lse_col, loss_parts);
    return hipGetLastError();
}

template <int DP>
hipError_t launch_bwd(const float* zi, const float* zj, int B, int d, float inv_t, const float* lse_row,
                      const float* lse_col, const float* grad_out, float* dzi, float* dzj, float* ws,
                      hipStream_t stream) {
    const NtPlan p = nt_plan(B);
    size_t lds = (size_t)NtCfg<DP>::kLdsFloats * sizeof(float);
    if (p.mfma) {
        float* zn = ws + nt_partial_floats(p, B, DP);
        ntxent2_normalise_kernel<DP><<<(2 * B + 15) / 16, 256, 0, stream>>>(zi, zj, B, d, zn);
        ntxent2_backward_mfma_kernel<DP><<<dim3(p.blocks_x, p.n_slices, 2), 256, NtM<DP>::kLdsFloats * sizeof(float), stream>>>(
            zn, B, inv_t, p.slice_len, lse_row, lse_col, grad_out, ws);
    } else
        ntxent2_backward_kernel<DP><<<dim3(p.blocks_x, p.n_slices), kItems * kWaves, lds, stream>>>(
            zi, zj, B, d, inv_t, p.slice_len, lse_row, lse_col, grad_out, ws);
    ntxent2_backward_finish_kernel<DP><<<(2 * B + 127) / 128, 128, 0, stream>>>(zi, zj, B, d, p.n_slices, ws, dzi,
                                                                              dzj);
    return hipGetLastError();
}

}  // namespace

int ntxent2_blocks(int B) { return (B + 255) / 256; }

size_t ntxent2_workspace_bytes(int B, int d) {
    const NtPlan p = nt_plan(B);
    const int dp = (d <= 32) ? 32 : 64;
    return sizeof(float) * (nt_partial_floats(p, B, dp) + (p.mfma ? (size_t)2 * B * dp : 0));
}

hipError_t launch_ntxent2_forward(const float* zi, const float* zj, int B, int d, float temperature, float* lse_row,
                                  float* lse_col, float* loss_parts, float* ws, hipStream_t stream) {
    float inv_t = 1.f / temperature;
    if (d <= 32) return launch_fwd<32>(zi, zj, B, d, inv_t, lse_row, lse_col, loss_parts, ws, stream);
    return launch_fwd<64>(zi, zj, B, d, inv_t, lse_row, lse_col, loss_parts, ws, stream);
}

hipError_t launch_ntxent2_backward(const float* zi, const float* zj, int B, int d, float temperature,
                                   const float* lse_row, const float* lse_col, const float* grad_out, float* dzi,
                                   float* dzj, float* ws, hipStream_t stream) {
    float inv_t = 1.f / temperature;
    if (d <= 32) return launch_bwd<32>(zi, zj, B, d, inv_t, lse_row, lse_col, grad_out, dzi, dzj, ws, stream);
    return launch_bwd<64>(zi, zj, B, d, inv_t, lse_row, lse_col, grad_out, dzi, dzj, ws, stream);
}

}  // namespace scmiac
