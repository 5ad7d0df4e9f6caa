// For every query row, the number of reference rows that are closer to it than its true
// match: the integer behind FOSCTTM (scmiac/evaluation/benchmark.py:75-112, 115-159).
// The reference builds the dense N x M distance matrix; here the matrix is produced tile
// by tile by the same fp32-MFMA stage loop as the k-NN search and compared on the spot.
//
//   closer_thresholds   per query: D_match (float32 direct-difference chain, the same
//                       arithmetic as the k-NN path), and two score thresholds around it:
//                       every score below `lo` certainly belongs to a reference closer
//                       than the match, every score at or above `hi` certainly does not
//                       (rounding-error model of the MFMA score, as in the refine kernel)
//   fused_count         s(q,r) = |r|^2 - 2 q.r by MFMA; scores below `lo` are counted in a
//                       register; the few in [lo, hi) are parked per lane and settled by
//                       the exact chain, so the count is exactly
//                       #{ r : D(q, r) < D(q, match) } in float32 direct-difference terms.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "fused_common.h"

namespace scmiac {

namespace {

constexpr int kPendCap = 32;  // parked references per lane; settled when one lane may run out of room

__global__ __launch_bounds__(256) void closer_thresholds_kernel(
    const float* __restrict__ Q, const float* __restrict__ R, const int* __restrict__ match, int d, long long n,
    long long n_pad, long long m, double err_abs_coef, double gamma, const unsigned* __restrict__ maxnorm_bits,
    float* __restrict__ lo, float* __restrict__ hi, float* __restrict__ dmatch, unsigned* __restrict__ counts,
    int* __restrict__ status, const int* __restrict__ perm_q /* packed position -> caller's row, or NULL */) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // position in the packed order
    if (q >= n_pad) return;
    counts[q] = 0u;
    float lf = -__builtin_inff(), hf = -__builtin_inff(), D = 0.0f;
    if (q < n) {
        const long long qo = perm_q ? (long long)perm_q[q] : q;
        const long long mi = match ? (long long)match[qo] : qo;
        if (mi < 0 || mi >= m) {
            *status = 1;  // the row keeps count 0
        } else {
            const float* qrow = Q + qo * d;
            D = exact_dist2(qrow, R + mi * d, d);
            double qn = 0.0;
            for (int i = 0; i < d; ++i) qn += (double)qrow[i] * (double)qrow[i];
            const double rmax2 = (double)__uint_as_float(*maxnorm_bits) * (1.0 + 1e-5);
            const double E = err_abs_coef * (rmax2 + qn);
            // D(q,r) <= (s + |q|^2 + E)(1 + gamma)  and  D(q,r) >= (s + |q|^2 - E)(1 - gamma)
            const double l = (double)D / (1.0 + gamma) - qn - E;
            const double h = (double)D / (1.0 - gamma) - qn + E;
            lf = (float)l;
            if ((double)lf > l) lf = nextafterf(lf, -__builtin_inff());
            hf = (float)h;
            if ((double)hf < h) hf = nextafterf(hf, __builtin_inff());
        }
    }
    lo[q] = lf;
    hi[q] = hf;
    dmatch[q] = D;
}

// what the stage loop does with a finished subtile: nothing, unless some score lies below
// `hi` (smallest of the lane's 16 scores as a depth-3 tree, as in the k-NN search).  Where
// nearly every subtile has such scores the question is not worth its 9 instructions:
// `dense` (set per stage from the stage before) skips it.
#define SCMIAC_TEST_PREV(PRV_)                                                                      \
    do {                                                                                            \
        bool go_ = dense;                                                                           \
        if (!go_) {                                                                                 \
            const float m0_ = min3_after_mfma(accs[PRV_][0], accs[PRV_][1], accs[PRV_][2]);         \
            const float m1_ = min3_raw_after(accs[PRV_][3], accs[PRV_][4], accs[PRV_][5], m0_);     \
            const float m2_ = min3_raw_after(accs[PRV_][6], accs[PRV_][7], accs[PRV_][8], m0_);     \
            const float m3_ = min3_raw_after(accs[PRV_][9], accs[PRV_][10], accs[PRV_][11], m0_);   \
            const float m4_ = min3_raw_after(accs[PRV_][12], accs[PRV_][13], accs[PRV_][14], m0_);  \
            const float mn_ = min2_raw(min3_raw(m0_, m1_, m2_), min3_raw(m3_, m4_, accs[PRV_][15])); \
            go_ = __any(mn_ < hi);                                                                  \
            hits += go_ ? 1 : 0;                                                                    \
        }                                                                                           \
        if (go_) count_slow(accs[PRV_], ref0s[PRV_]);                                               \
    } while (0)

template <int D2, int NW_, int RT_>
__global__ __launch_bounds__(NW_ * 64, (NW_ >= 8 ? NW_ / 4 : 2))
__attribute__((amdgpu_waves_per_eu((NW_ >= 8 ? NW_ / 4 : 2), (NW_ >= 8 ? NW_ / 4 : 2)))) void fused_count_kernel(
    const float* __restrict__ Qp,      // [n_pad][ROWF], scaled by -2
    const float* __restrict__ Rp,      // [m_pad][LS]
    const float* __restrict__ rnorm,   // [m_pad], 1.7e38 on padding rows
    const float* __restrict__ thr_lo,  // [n_pad]
    const float* __restrict__ thr_hi,  // [n_pad]
    const float* __restrict__ dmatch,  // [n_pad]
    unsigned* __restrict__ counts,     // [n_pad], zeroed
    unsigned* __restrict__ pend_g,     // [grid waves][64][kPendCap]
    long long chunk_rows, FusedTail tail) {
    using C = FusedCfg<D2, NW_, RT_, 32, true>;
    constexpr int D2P = C::D2P, ROWF = C::ROWF, LS = C::LS, RT = C::RT, NW = C::NW;
    constexpr int NSUB = RT / 32, NB = C::NB;
    constexpr int STAGE_BYTES = RT * LS * 4;
    constexpr int NCHUNK = STAGE_BYTES / 1024;
    constexpr int LD_ITERS = (NCHUNK + NW - 1) / NW;
    constexpr int NORM_CHUNKS = RT / 64;
    static_assert(STAGE_BYTES % 1024 == 0 && RT % 64 == 0 && NORM_CHUNKS <= NW, "stage geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tile = reinterpret_cast<float*>(smem);  // [NB][RT*LS]
    float* tnorm = tile + NB * RT * LS;            // [NB][RT]
    const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tile;
    const unsigned tnorm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tnorm;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;
    // same cut of the work as the k-NN search: (reference chunk, query block) per block, the
    // last, partly filled round of the grid cut into shorter chunks
    int split, qb;
    long long r_begin, rows_here;
    if ((int)blockIdx.x < tail.block0) {
        split = blockIdx.x / tail.qb0; qb = blockIdx.x % tail.qb0;
        r_begin = (long long)split * chunk_rows;
        rows_here = chunk_rows;
    } else {
        const int tb_ = (int)blockIdx.x - tail.block0;
        split = tb_ % tail.St; qb = tail.qb0 + tb_ / tail.St;
        r_begin = (long long)split * tail.chunk_rows;
        rows_here = tail.m_pad - r_begin;
        rows_here = rows_here < 0 ? 0 : (rows_here < tail.chunk_rows ? rows_here : tail.chunk_rows);
    }
    const long long q0 = (long long)qb * C::QPB + wave * 32;
    const int T = (int)(rows_here / RT);

    unsigned* pend_l = pend_g + (((long long)blockIdx.x * NW + wave) * 64 + lane) * kPendCap;
    unsigned pcnt = 0u, cnt = 0u;
    const float4* qrow4 = reinterpret_cast<const float4*>(Qp + (q0 + j) * ROWF);

    float qf[D2P];
    {
        const float4* qp4 = reinterpret_cast<const float4*>(Qp + (q0 + j) * ROWF + hh * D2P);
#pragma unroll
        for (int c = 0; c < D2P / 4; ++c) {
            float4 v = qp4[c];
            qf[4 * c + 0] = v.x; qf[4 * c + 1] = v.y; qf[4 * c + 2] = v.z; qf[4 * c + 3] = v.w;
        }
    }
    float lo = thr_lo[q0 + j], hi = thr_hi[q0 + j];
    const float dq = dmatch[q0 + j];
#pragma unroll
    for (int c = 0; c < D2P; ++c) asm volatile("" : "+v"(qf[c]));
    asm volatile("" : "+v"(lo), "+v"(hi));

    // settle the parked references of every lane with the exact chain
    auto settle = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the parked entries have reached L2
        for (unsigned e = 0; __any(e < pcnt); ++e) {
            if (e < pcnt) {
                const unsigned ref = __hip_atomic_load(pend_l + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float D = exact_dist2_packed(qrow4, reinterpret_cast<const float4*>(Rp + (long long)ref * LS),
                                                   ROWF / 4);
                cnt += (D < dq) ? 1u : 0u;
            }
        }
        pcnt = 0u;
    };
    // Some score of the subtile may lie below `hi`.  t = score - lo (two registers per packed
    // subtract; the sign of a float difference is exact): the sign bits are shifted into four short
    // histories (one funnel shift per register, four independent chains) and counted with one
    // population count; the smallest non-negative t (an unsigned minimum over the raw bits,
    // negative floats sort last) tells whether any score lies in [lo, hi), since
    // score < hi implies t <= fl(hi - lo).  The references in the band are parked as before.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 lo2 = {lo, lo};
    const unsigned band_bits = (hi > lo) ? __float_as_uint(hi - lo) : 0u;  // padding queries: lo = hi = -inf
    auto count_slow = [&](const f32x16& acc, unsigned ref0_wave) __attribute__((always_inline)) {
        f32x16 sc = acc;
        asm volatile("" : "+v"(sc));  // keeps this path's arithmetic behind the branch that guards it
        unsigned tbits[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 t2 = (f32x2){sc[r], sc[r + 1]} - lo2;
            tbits[r] = __float_as_uint(t2[0]);
            tbits[r + 1] = __float_as_uint(t2[1]);
        }
        unsigned hist[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < 4; ++c) hist[c] = __builtin_amdgcn_alignbit(hist[c], tbits[4 * c + r], 31);
        }
        cnt += (unsigned)__builtin_popcount(hist[0] | (hist[1] << 4) | (hist[2] << 8) | (hist[3] << 12));
        auto umin3 = [](unsigned a, unsigned b, unsigned c) __attribute__((always_inline)) {
            return min(min(a, b), c);
        };
        const unsigned u0 = umin3(tbits[0], tbits[1], tbits[2]), u1 = umin3(tbits[3], tbits[4], tbits[5]), u2 = umin3(tbits[6], tbits[7], tbits[8]);
        const unsigned u3 = umin3(tbits[9], tbits[10], tbits[11]), u4 = umin3(tbits[12], tbits[13], tbits[14]);
        const unsigned un = min(umin3(u0, u1, u2), umin3(u3, u4, tbits[15]));
        if (__builtin_expect(__any(un <= band_bits), 0)) {
            if (__any(pcnt > (unsigned)(kPendCap - 16))) settle();
            const unsigned ref0 = ref0_wave + 4u * (unsigned)hh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool b = sc[r] >= lo && sc[r] < hi;
                if (__any(b)) {
                    if (b) {
                        pend_l[pcnt] = ref0 + (unsigned)((r & 3) + 8 * (r >> 2));
                        ++pcnt;
                    }
                }
            }
        }
    };

    constexpr int NGRP = D2P / 4;
    static_assert(NGRP >= 2, "two groups of lookahead");
    f32x16 accs[2];
    float fr[3][4];
    unsigned ref0s[2] = {0u, 0u};
    const float big = __uint_as_float(TAU_MAX_BITS);
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[1][r] = big;  // "previous subtile" of the first one: nothing below hi

    if (T > 0) SCMIAC_STAGE(r_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool dense = false;  // wave-uniform: skip the per-subtile question in this stage
    int hits = 0;        // subtiles of the current stage that had scores below `hi`
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) SCMIAC_STAGE(r_begin + (long long)(t + 1) * RT, (t + 1) & 1);
        // asked again every eighth stage, so a stream that turns sparse is noticed
        dense = (t & 7) != 0 && (dense || hits >= NSUB - 1);
        hits = 0;
        const float* tb = tile + (t & 1) * RT * LS;
        const float* nb = tnorm + (t & 1) * RT;
        const long long cur_row0 = r_begin + (long long)t * RT;  // first reference row of this stage
        SCMIAC_LOAD_NORMS(0, 0);
        SCMIAC_LOAD_GROUP(0, 0, 0);
        SCMIAC_LOAD_GROUP(0, 1, 1);
        static_assert(NSUB == 2 || NSUB == 4 || NSUB == 8 || NSUB == 10, "stage of 64, 128, 256 or 320 rows");
        SCMIAC_SUBTILE(0, 1, 0);
        SCMIAC_SUBTILE(1, 0, 1);
        if constexpr (NSUB > 2) {
            SCMIAC_SUBTILE(0, 1, 2);
            SCMIAC_SUBTILE(1, 0, 3);
        }
        if constexpr (NSUB > 4) {
            SCMIAC_SUBTILE(0, 1, 4);
            SCMIAC_SUBTILE(1, 0, 5);
            SCMIAC_SUBTILE(0, 1, 6);
            SCMIAC_SUBTILE(1, 0, 7);
        }
        if constexpr (NSUB > 8) {
            SCMIAC_SUBTILE(0, 1, 8);
            SCMIAC_SUBTILE(1, 0, 9);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (T > 0) count_slow(accs[1], ref0s[1]);  // the last subtile
    settle();
    if (cnt) atomicAdd(counts + q0 + j, cnt);
}
#undef SCMIAC_TEST_PREV

template <int D2, int NW, int RT>
hipError_t launch_count_t(const KnnGeometry& g, const float* Qp, const float* Rp, const float* rnorm,
                          const float* lo, const float* hi, const float* dmatch, unsigned* counts, unsigned* pend,
                          hipStream_t stream) {
    using C = FusedCfg<D2, NW, RT, 32, true>;
    size_t lds = C::lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_count_kernel<D2, NW, RT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)g.blocks), block(NW * 64);
    hipLaunchKernelGGL((fused_count_kernel<D2, NW, RT>), grid, block, lds, stream, Qp, Rp, rnorm, lo, hi, dmatch, counts,
                       pend, g.chunk_rows, g.tail);
    return hipGetLastError();
}

}  // namespace

// counts of the packed order back to the caller's rows
__global__ void scatter_counts_kernel(const unsigned* __restrict__ packed, const int* __restrict__ perm_q, long long n,
                                      unsigned* __restrict__ out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[perm_q ? perm_q[q] : q] = packed[q];
}

hipError_t launch_scatter_counts(const unsigned* packed, const int* perm_q, long long n, unsigned* out,
                                 hipStream_t stream) {
    hipLaunchKernelGGL(scatter_counts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, packed, perm_q, n,
                       out);
    return hipGetLastError();
}

// bytes of the parked-reference buffers for a grid of this geometry
size_t closer_count_pend_bytes(const KnnGeometry& g) {
    return (size_t)g.blocks * g.nw * 64 * kPendCap * sizeof(unsigned);
}

hipError_t launch_closer_count(const KnnGeometry& g, const float* Q, const float* R, const int* match, int d,
                               long long n, long long m, const float* Qp, const float* Rp, const float* rnorm,
                               const unsigned* maxnorm_bits, float* lo, float* hi, float* dmatch, unsigned* counts,
                               unsigned* pend, int* status, const int* perm_q, hipStream_t stream) {
    // the rounding-error model of the refine kernel's certificate, without the key term
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double G = 2.0 * (double)(d + 2 * g.D2 + 4) * u;
    const double err_abs_coef = 2.0 * G;
    const double gamma = 2.0 * (double)(d + 4) * u;
    hipLaunchKernelGGL(closer_thresholds_kernel, dim3((unsigned)((g.n_pad + 255) / 256)), dim3(256), 0, stream, Q, R,
                       match, d, n, g.n_pad, m, err_abs_coef, gamma, maxnorm_bits, lo, hi, dmatch, counts, status, perm_q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
#define SCMIAC_COUNT_CASE(V_, NW_, RT_) \
    if (g.D2 == V_ && g.nw == NW_ && g.RT == RT_) \
        return launch_count_t<V_, NW_, RT_>(g, Qp, Rp, rnorm, lo, hi, dmatch, counts, pend, stream);
    SCMIAC_COUNT_CASE(8, 12, 320) SCMIAC_COUNT_CASE(15, 12, 320) SCMIAC_COUNT_CASE(16, 12, 320)
    SCMIAC_COUNT_CASE(20, 12, 320) SCMIAC_COUNT_CASE(25, 12, 320) SCMIAC_COUNT_CASE(32, 12, 256)
    SCMIAC_COUNT_CASE(40, 12, 128) SCMIAC_COUNT_CASE(50, 12, 128) SCMIAC_COUNT_CASE(64, 8, 128)
    SCMIAC_COUNT_CASE(80, 8, 64) SCMIAC_COUNT_CASE(96, 8, 64) SCMIAC_COUNT_CASE(100, 8, 64) SCMIAC_COUNT_CASE(112, 8, 64)
    SCMIAC_COUNT_CASE(128, 8, 64)
#undef SCMIAC_COUNT_CASE
    return hipErrorInvalidValue;
}

}  // namespace scmiac
