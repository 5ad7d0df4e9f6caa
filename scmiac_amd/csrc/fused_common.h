// Pieces shared by the fused kernels that stream the reference rows through LDS and
// score them against register-resident queries with fp32 MFMA (the k-NN search and the
// closer-than-the-match count): key/minimum helpers, the stage geometry, LDS-DMA loads,
// the exact float32 distance, and the software-pipelined stage loop as macros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace scmiac {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

static constexpr u64 KEY_SENTINEL = 0xffffffffffffffffull;  // sorts after every real key
// orderable(+inf) in the high word, 0xffffffff index: "no candidate" in the shortlist
static constexpr u64 LIST_SENTINEL = 0xff800000ffffffffull;

__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
    return __uint_as_float(u);
}
// correctly rounded float32 square root: the double result rounded once more is exact
// for sqrt because 53 >= 2*24 + 2
__device__ __forceinline__ float sqrt_rn_f32(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a < b ? b : a; }

// Bitonic sort of one key per lane, ascending across the wave.  WIDTH = 32 sorts
// lanes 0..31 ascending (and 32..63 descending); WIDTH = 64 sorts all 64 lanes.
template <int WIDTH>
__device__ __forceinline__ u64 wave_sort(u64 key, int lane) {
#pragma unroll
    for (int k = 2; k <= WIDTH; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            u64 other = __shfl_xor(key, j);
            bool up = (lane & k) == 0;
            bool low = (lane & j) == 0;
            key = (up == low) ? umin64(key, other) : umax64(key, other);
        }
    }
    return key;
}
// A bitonic sequence across the 64 lanes -> ascending.
__device__ __forceinline__ u64 wave_merge(u64 key, int lane) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        u64 other = __shfl_xor(key, j);
        key = ((lane & j) == 0) ? umin64(key, other) : umax64(key, other);
    }
    return key;
}

static constexpr unsigned KEY_MASK = 0xfffffff0u;
static constexpr unsigned TAU_MAX_BITS = 0x7f000000u;  // 1.7e38: above every real score, below inf
__device__ __forceinline__ float score_key(float score, int r) {
    return __uint_as_float((__float_as_uint(score) & KEY_MASK) | (unsigned)r);
}
// min of three / two floats as single instructions.  fminf() would first canonicalise
// its operands (two extra VALU instructions per subtile, each worth 4 cycles of MFMA issue);
// the scores are never signalling NaNs.
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The same, ordered after the statement that produced `after` (an operand the instruction
// does not use): the compiler is free to move independent asm statements past each other,
// and a read of MFMA results must not overtake the one that carries the wait states.
__device__ __forceinline__ float min3_raw_after(float a, float b, float c, float after) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(after));
    return r;
}
// The same, for operands that an MFMA may have written only just before: the compiler
// does not see through inline assembly, so the wait states the ISA asks for between a
// 16-pass MFMA's write and a VALU read of the same registers (18) are spelled out here.
__device__ __forceinline__ float min3_after_mfma(float a, float b, float c) {
    float r;
    asm("s_nop 15\n\ts_nop 1\n\tv_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min2_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int D2, int NW_, int RT_, int PCAP_, bool GP_>
struct FusedCfg {
    static constexpr int D2P = (D2 + 3) & ~3;
    static constexpr int ROWF = 2 * D2P;  // feature floats per packed row
    // row stride of the reference image (global and LDS): an odd number of 16-byte slots
    // per row keeps the ds_read_b128 fragment reads conflict-free
    static constexpr int LS = ROWF + (((ROWF / 4) % 2 == 0) ? 4 : 8);
    static constexpr int RT = RT_;                     // reference rows per LDS stage
    static constexpr int NW = NW_;                     // waves per block
    static constexpr int PCAP = PCAP_;                 // pending candidates per query
    static constexpr int NB = 2;                       // LDS stage buffers
    // pending candidates in global memory (L2) instead of LDS: leaves the LDS to longer stages
    static constexpr bool GP = GP_;
    static constexpr int QPB = NW * 32;                // queries per block
    static constexpr size_t lds_bytes() {
        return sizeof(float) * (NB * RT * LS + NB * RT) + (GP ? 0 : sizeof(uint2) * (NW * 32 * PCAP));
    }
};

// One wave-wide LDS-DMA load: every lane fetches 16 (or 4) bytes from its own global
// address and the data lands at the wave-uniform LDS byte address lds_dst + lane * size.
// The compiler does not count these loads: the caller waits with s_waitcnt vmcnt before
// the barrier that publishes the data.  M0 (the LDS base of the DMA) is saved and restored
// inside the statement because the compiler reserves it.
__device__ __forceinline__ void lds_dma_16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ void lds_dma_4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
// 16-byte LDS read through an LDS-address-space pointer (stays one ds_read_b128)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_read_b128(const float* p) {
    f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(
        (const __attribute__((address_space(3))) void*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float exact_dist2(const float* __restrict__ a, const float* __restrict__ b, int d) {
    float acc = 0.0f;
    for (int i = 0; i < d; ++i) {
        float t = __fsub_rn(a[i], b[i]);
        acc = __fmaf_rn(t, t, acc);
    }
    return acc;
}

// Exact squared distance from the packed copies (16-byte loads): the packed order is the
// feature order with zeros in between, and a zero pair leaves the fmaf chain's value as it
// is, so this is bit-identical to exact_dist2 on the original rows.  qp holds -2 q.
__device__ __forceinline__ float exact_dist2_packed(const float4* __restrict__ qp, const float4* __restrict__ rp,
                                                    int chunks) {
    float acc = 0.0f;
    for (int c = 0; c < chunks; ++c) {
        const float4 a = qp[c], b = rp[c];
        float t = __fsub_rn(-0.5f * a.x, b.x); acc = __fmaf_rn(t, t, acc);
        t = __fsub_rn(-0.5f * a.y, b.y); acc = __fmaf_rn(t, t, acc);
        t = __fsub_rn(-0.5f * a.z, b.z); acc = __fmaf_rn(t, t, acc);
        t = __fsub_rn(-0.5f * a.w, b.w); acc = __fmaf_rn(t, t, acc);
    }
    return acc;
}

// ---- the stage loop, as macros expanded inside a kernel -------------------------------
// They use the kernel's locals by name: tile_lds, tnorm_lds, Rp, rnorm, wave, lane,
// j, hh, tb, nb, cur_row0 (first reference row of the stage being scored, wave-uniform),
// fr[3][4], accs[2], qf[], ref0s[2] and the constants RT, LS, D2P, NW,
// NCHUNK, LD_ITERS, NORM_CHUNKS, STAGE_BYTES, NGRP, NSUB, D2.  SCMIAC_TEST_PREV(PRV_) is
// supplied by the kernel: what to do with the finished scores in accs[PRV_].
// Stage copy global -> LDS without registers: the RT packed rows from ROW0_ on are one
// contiguous byte range in both, and so are their norms.
#define SCMIAC_STAGE(ROW0_, BUF_)                                                                   \
    do {                                                                                            \
        const long long row0_ = (ROW0_);                                                            \
        const char* src_ = reinterpret_cast<const char*>(Rp + row0_ * LS);                          \
        const unsigned dst_ = tile_lds + (unsigned)((BUF_) * STAGE_BYTES);                          \
        _Pragma("unroll") for (int i = 0; i < LD_ITERS; ++i) {                                      \
            const int ch = wave + i * NW; /* wave-uniform */                                        \
            if (ch < NCHUNK) lds_dma_16(src_ + ch * 1024 + lane * 16, dst_ + ch * 1024);            \
        }                                                                                           \
        if (wave < NORM_CHUNKS)                                                                     \
            lds_dma_4(rnorm + row0_ + wave * 64 + lane, tnorm_lds + (unsigned)(((BUF_) * RT + wave * 64) * 4)); \
    } while (0)
// ring slot SLOT_ <- steps 4 G_ .. 4 G_ + 3 of subtile SUB_'s fragments
#define SCMIAC_LOAD_GROUP(SUB_, G_, SLOT_)                                                          \
    do {                                                                                            \
        float4 v = lds_read_b128(tb + ((SUB_) * 32 + j) * LS + hh * D2P + 4 * (G_));                \
        fr[SLOT_][0] = v.x; fr[SLOT_][1] = v.y; fr[SLOT_][2] = v.z; fr[SLOT_][3] = v.w;             \
    } while (0)
// accumulator set SET_ <- |r|^2 of subtile SUB_ (the MFMA chain then adds -2 q.r on top)
#define SCMIAC_LOAD_NORMS(SET_, SUB_)                                                               \
    do {                                                                                            \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
            float4 nv = lds_read_b128(nb + (SUB_) * 32 + 8 * g + 4 * hh);                           \
            accs[SET_][4 * g + 0] = nv.x; accs[SET_][4 * g + 1] = nv.y;                             \
            accs[SET_][4 * g + 2] = nv.z; accs[SET_][4 * g + 3] = nv.w;                             \
        }                                                                                           \
    } while (0)
// One subtile: CUR_/PRV_ are the compile-time indices of the two accumulator sets.  On entry
// accs[CUR_] holds the subtile's |r|^2 and the ring its first two fragment groups, all
// fetched while the previous subtile's chain ran, so the first MFMA follows the previous
// chain's last one without an LDS wait (the waves of a SIMD run in step and would all be
// waiting at the same time).  The previous subtile's scores are tested after the first
// group; once they are dealt with, their registers take the next subtile's |r|^2.
// one group of a subtile's chain: the LDS read two groups ahead, then this group's MFMAs
#define SCMIAC_GROUP(CUR_, SUB_, G_)                                                                \
    do {                                                                                            \
        const int ph_ = ((SUB_) * NGRP) % 3; /* ring slot of this subtile's group 0 */              \
        if ((G_) + 2 < NGRP) SCMIAC_LOAD_GROUP((SUB_), (G_) + 2, (ph_ + (G_) + 2) % 3);             \
        else if ((SUB_) + 1 < NSUB) SCMIAC_LOAD_GROUP((SUB_) + 1, (G_) + 2 - NGRP, (ph_ + (G_) + 2) % 3); \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                               \
            if (4 * (G_) + e < D2)                                                                  \
                accs[CUR_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[(ph_ + (G_)) % 3][e], qf[4 * (G_) + e], \
                                                                  accs[CUR_], 0, 0, 0);             \
        /* tell the scheduler to keep this order: the LDS read, then this group's MFMAs */          \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                          \
        if (4 * (G_) + 4 <= D2) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                  \
        else __builtin_amdgcn_sched_group_barrier(0x008, (D2 % 4) ? (D2 % 4) : 4, 0);               \
    } while (0)
// TEST_ names the macro that deals with the finished scores in accs[PRV_].
#define SCMIAC_SUBTILE_T(CUR_, PRV_, SUB_, TEST_)                                                   \
    do {                                                                                            \
        SCMIAC_GROUP(CUR_, SUB_, 0);                                                                \
        {                                                                                           \
            TEST_(PRV_);                                                                            \
            if ((SUB_) + 1 < NSUB) {                                                                \
                SCMIAC_LOAD_NORMS(PRV_, (SUB_) + 1);                                                \
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                  \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int g = 1; g < NGRP; ++g) SCMIAC_GROUP(CUR_, SUB_, g);               \
        ref0s[CUR_] = (unsigned)(cur_row0 + (SUB_) * 32); /* wave-uniform */                        \
    } while (0)
#define SCMIAC_SUBTILE(CUR_, PRV_, SUB_) SCMIAC_SUBTILE_T(CUR_, PRV_, SUB_, SCMIAC_TEST_PREV)
// the subtiles of a stage, written out: their numbers must be compile-time constants (ring
// slots, LDS offsets), whatever the optimizer thinks of unrolling a loop this big
#define SCMIAC_STAGE_SUBTILES(TEST_)                                                                \
    do {                                                                                            \
        static_assert(NSUB == 2 || NSUB == 4 || NSUB == 8 || NSUB == 10, "stage of 64, 128, 256 or 320 rows"); \
        SCMIAC_SUBTILE_T(0, 1, 0, TEST_);                                                           \
        SCMIAC_SUBTILE_T(1, 0, 1, TEST_);                                                           \
        if constexpr (NSUB > 2) {                                                                   \
            SCMIAC_SUBTILE_T(0, 1, 2, TEST_);                                                       \
            SCMIAC_SUBTILE_T(1, 0, 3, TEST_);                                                       \
        }                                                                                           \
        if constexpr (NSUB > 4) {                                                                   \
            SCMIAC_SUBTILE_T(0, 1, 4, TEST_);                                                       \
            SCMIAC_SUBTILE_T(1, 0, 5, TEST_);                                                       \
            SCMIAC_SUBTILE_T(0, 1, 6, TEST_);                                                       \
            SCMIAC_SUBTILE_T(1, 0, 7, TEST_);                                                       \
        }                                                                                           \
        if constexpr (NSUB > 8) {                                                                   \
            SCMIAC_SUBTILE_T(0, 1, 8, TEST_);                                                       \
            SCMIAC_SUBTILE_T(1, 0, 9, TEST_);                                                       \
        }                                                                                           \
    } while (0)

}  // namespace scmiac
