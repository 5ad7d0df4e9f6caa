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
        float na = 0.f, nb = 0.f, ab = 0.f;
        for (int c = 0; c < d; ++c) {
            na = fmaf(a[c], a[c], na);
            nb = fmaf(b[c], b[c], nb);
        }
        const float ia = 1.f / fmaxf(sqrtf(na), kNormEps), ib = 1.f / fmaxf(sqrtf(nb), kNormEps);
        for (int c = 0; c < d; ++c) ab = fmaf(a[c] * ia, b[c] * ib, ab);
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

// how the other side's rows are cut into slices: enough blocks to fill the GPU
struct NtPlan {
    int blocks_x, n_slices, slice_len;
};
inline NtPlan nt_plan(int B) {
    NtPlan p;
    p.blocks_x = (B + kItems - 1) / kItems;
    int s = (p.blocks_x >= 256) ? 1 : (256 + p.blocks_x - 1) / p.blocks_x;
    if (s > 16) s = 16;
    p.slice_len = (B + s - 1) / s;
    p.n_slices = (B + p.slice_len - 1) / p.slice_len;
    return p;
}

template <int DP>
hipError_t launch_fwd(const float* zi, const float* zj, int B, int d, float inv_t, float* lse_row, float* lse_col,
                      float* loss_parts, float* ws, hipStream_t stream) {
    const NtPlan p = nt_plan(B);
    size_t lds = (size_t)NtCfg<DP>::kLdsFloats * sizeof(float);
    float4* part = reinterpret_cast<float4*>(ws);
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
    size_t fwd = (size_t)p.n_slices * B * 4, bwd = (size_t)p.n_slices * 2 * B * dp;
    return sizeof(float) * (fwd > bwd ? fwd : bwd);
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
