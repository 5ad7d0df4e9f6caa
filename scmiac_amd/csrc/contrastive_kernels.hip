// Symmetric NT-Xent loss between the anchor cells' two embeddings (the consumer of the
// anchor table; scmiac/modeling/loss.py:85-133), forward and backward, one launch each.
//
//   zn = z / max(|z|, 1e-12)            S = zn_i . zn_j^T / T
//   loss = 0.5 * ( mean_t(lse_j S[t,j] - S[t,t]) + mean_t(lse_i S[i,t] - S[t,t]) )
//
// Work split: item t (one anchor pair) owns row t and column t of S.  A block has 4
// waves x 64 items; the rows of the other side pass through LDS in chunks of JT rows,
// each wave taking a quarter of every chunk (LDS reads of a row are broadcasts), and
// the four partial results of an item meet in LDS.  S is recomputed in the backward
// pass instead of being stored: B^2 * d is tiny next to a launch.
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

template <int DP>
__global__ __launch_bounds__(kItems* kWaves) void ntxent2_forward_kernel(
    const float* __restrict__ zi, const float* __restrict__ zj, int B, int d, float inv_t,
    float* __restrict__ lse_row, float* __restrict__ lse_col, float* __restrict__ loss_parts) {
    constexpr int JT = NtCfg<DP>::JT;
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * kItems + lane;
    const int tc = (t < B) ? t : B - 1;
    float a[DP], b[DP];
    load_own_row<DP>(zi, tc, d, a);
    load_own_row<DP>(zj, tc, d, b);

    float mr = -INFINITY, sr = 0.f, mc = -INFINITY, sc = 0.f;
    for (int c0 = 0; c0 < B; c0 += JT) {
        __syncthreads();
        stage_chunk<DP>(zi, zj, B, d, c0, lds);
        __syncthreads();
        int j0 = w * (JT / kWaves), j1 = j0 + JT / kWaves;
        if (c0 + j1 > B) j1 = B - c0;
        for (int j = j0; j < j1; ++j) {
            lse_push(dot_lds<DP>(a, lds + (size_t)(JT + j) * DP) * inv_t, mr, sr);  // S[t, c0+j]
            lse_push(dot_lds<DP>(b, lds + (size_t)j * DP) * inv_t, mc, sc);         // S[c0+j, t]
        }
    }
    __syncthreads();
    float4* part = reinterpret_cast<float4*>(lds);
    part[w * kItems + lane] = make_float4(mr, sr, mc, sc);
    __syncthreads();
    if (w == 0) {
        float M = -INFINITY, Mc = -INFINITY;
        for (int q = 0; q < kWaves; ++q) {
            float4 p = part[q * kItems + lane];
            M = fmaxf(M, p.x);
            Mc = fmaxf(Mc, p.z);
        }
        float S = 0.f, Sc = 0.f;
        for (int q = 0; q < kWaves; ++q) {
            float4 p = part[q * kItems + lane];
            if (p.y > 0.f) S += p.y * expf(p.x - M);
            if (p.w > 0.f) Sc += p.w * expf(p.z - Mc);
        }
        float lr = M + logf(S), lc = Mc + logf(Sc);
        float diag = 0.f;
#pragma unroll
        for (int c = 0; c < DP; ++c) diag = fmaf(a[c], b[c], diag);
        diag *= inv_t;
        float contrib = 0.f;
        if (t < B) {
            lse_row[t] = lr;
            lse_col[t] = lc;
            contrib = 0.5f * ((lr - diag) + (lc - diag));
        }
        for (int off = 32; off; off >>= 1) contrib += __shfl_xor(contrib, off);
        if (lane == 0) loss_parts[blockIdx.x] = contrib / (float)B;
    }
}

template <int DP>
__global__ __launch_bounds__(kItems* kWaves) void ntxent2_backward_kernel(
    const float* __restrict__ zi, const float* __restrict__ zj, int B, int d, float inv_t,
    const float* __restrict__ lse_row, const float* __restrict__ lse_col, const float* __restrict__ grad_out,
    float* __restrict__ dzi, float* __restrict__ dzj) {
    constexpr int JT = NtCfg<DP>::JT;
    extern __shared__ __align__(16) float lds[];
    float* lse_chunk = lds + 2 * JT * DP;  // [0, JT): lse_row of the chunk's rows, [JT, 2JT): lse_col
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * kItems + lane;
    const int tc = (t < B) ? t : B - 1;
    float a[DP], b[DP], ga[DP], gb[DP];
    const float inv_a = load_own_row<DP>(zi, tc, d, a);
    const float inv_b = load_own_row<DP>(zj, tc, d, b);
#pragma unroll
    for (int c = 0; c < DP; ++c) ga[c] = gb[c] = 0.f;
    const float lr_t = lse_row[tc], lc_t = lse_col[tc];
    const float coef = 0.5f * grad_out[0] / (float)B * inv_t;

    for (int c0 = 0; c0 < B; c0 += JT) {
        __syncthreads();
        stage_chunk<DP>(zi, zj, B, d, c0, lds);
        for (int r = threadIdx.x; r < 2 * JT; r += blockDim.x) {
            int j = c0 + (r % JT);
            lse_chunk[r] = (j < B) ? ((r < JT) ? lse_row[j] : lse_col[j]) : 0.f;
        }
        __syncthreads();
        int j0 = w * (JT / kWaves), j1 = j0 + JT / kWaves;
        if (c0 + j1 > B) j1 = B - c0;
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
        if (w == 0) {
            float(&zn)[DP] = side ? b : a;
            float tot[DP];
            float proj = 0.f;
#pragma unroll
            for (int c = 0; c < DP; ++c) {
                float v = 0.f;
                for (int q = 0; q < kWaves; ++q) v += lds[((size_t)q * DP + c) * kItems + lane];
                tot[c] = v;
                proj = fmaf(zn[c], v, proj);
            }
            if (t < B) {
                float inv = side ? inv_b : inv_a;
                float* out = (side ? dzj : dzi) + (size_t)t * d;
#pragma unroll
                for (int c = 0; c < DP; ++c)
                    if (c < d) out[c] = (tot[c] - zn[c] * proj) * inv;
            }
        }
    }
}

template <int DP>
hipError_t launch_fwd(const float* zi, const float* zj, int B, int d, float inv_t, float* lse_row, float* lse_col,
                      float* loss_parts, hipStream_t stream) {
    size_t lds = (size_t)NtCfg<DP>::kLdsFloats * sizeof(float);
    int blocks = (B + kItems - 1) / kItems;
    ntxent2_forward_kernel<DP><<<blocks, kItems * kWaves, lds, stream>>>(zi, zj, B, d, inv_t, lse_row, lse_col,
                                                                       loss_parts);
    return hipGetLastError();
}

template <int DP>
hipError_t launch_bwd(const float* zi, const float* zj, int B, int d, float inv_t, const float* lse_row,
                      const float* lse_col, const float* grad_out, float* dzi, float* dzj, hipStream_t stream) {
    size_t lds = (size_t)NtCfg<DP>::kLdsFloats * sizeof(float);
    int blocks = (B + kItems - 1) / kItems;
    ntxent2_backward_kernel<DP><<<blocks, kItems * kWaves, lds, stream>>>(zi, zj, B, d, inv_t, lse_row, lse_col,
                                                                        grad_out, dzi, dzj);
    return hipGetLastError();
}

}  // namespace

int ntxent2_blocks(int B) { return (B + kItems - 1) / kItems; }

hipError_t launch_ntxent2_forward(const float* zi, const float* zj, int B, int d, float temperature, float* lse_row,
                                  float* lse_col, float* loss_parts, hipStream_t stream) {
    float inv_t = 1.f / temperature;
    if (d <= 32) return launch_fwd<32>(zi, zj, B, d, inv_t, lse_row, lse_col, loss_parts, stream);
    return launch_fwd<64>(zi, zj, B, d, inv_t, lse_row, lse_col, loss_parts, stream);
}

hipError_t launch_ntxent2_backward(const float* zi, const float* zj, int B, int d, float temperature,
                                   const float* lse_row, const float* lse_col, const float* grad_out, float* dzi,
                                   float* dzj, hipStream_t stream) {
    float inv_t = 1.f / temperature;
    if (d <= 32) return launch_bwd<32>(zi, zj, B, d, inv_t, lse_row, lse_col, grad_out, dzi, dzj, stream);
    return launch_bwd<64>(zi, zj, B, d, inv_t, lse_row, lse_col, grad_out, dzi, dzj, stream);
}

}  // namespace scmiac
