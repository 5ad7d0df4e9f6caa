// The fused distance + top-k kernel of the k-NN search (template) with its launcher and
// occupancy query.  Included by the translation units knn_fused_0..3.hip, each of which
// instantiates a share of the variants listed in fused_variants.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "fused_common.h"

namespace scmiac {

// --------------------------------------------------------------------------------
// fused_knn
//   block = NW waves, each wave owns 32 queries for the whole pass (MFMA B operand,
//   columns of the 32x32 result tile, kept in registers); all waves share the reference
//   stage in LDS (MFMA A operand, rows).  Lane l: query j = l & 31; accumulator register
//   r holds the score of reference row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the subtile,
//   so each lane tests its 16 scores against one threshold, its query's.
//   Selection works on KEYS: the score with its low four mantissa bits replaced by the
//   register number r.  A key is a float within 16 ulp of the score, the 16 keys of a
//   lane are distinct, and the smallest key names its own register -- so the candidate
//   path needs no search for "which reference was it".  The 16-ulp perturbation is part
//   of the error bound the refine kernel's certificate uses.
// --------------------------------------------------------------------------------


// Merge query q's pending candidates into its 64-entry shortlist (global memory),
// all 64 lanes cooperating.  Returns the new threshold (key of the kt-th entry).
// The pending candidates of a query sit in two buffers of HCAP entries, one per lane half
// of the MFMA layout (c0 and c1 entries in use).
template <int HCAP, bool GP>
__device__ __forceinline__ float flush_query(unsigned c0, unsigned c1, const uint2* pend_q, u64* list_q,
                                             bool inited, int lane, int kt) {
    static_assert(2 * HCAP <= 64, "both buffers are sorted by one bitonic network across the wave");
    u64 key = LIST_SENTINEL;
    const unsigned h = (unsigned)lane / HCAP, i = (unsigned)lane % HCAP;
    if (h < 2u && i < (h ? c1 : c0)) {
        uint2 e;  // buffer h, entry i
        if constexpr (GP) {  // written by this wave a while ago: read from L2, past the L1
            const u64 raw = __hip_atomic_load(reinterpret_cast<const u64*>(pend_q) + lane, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
            e = make_uint2((unsigned)raw, (unsigned)(raw >> 32));
        } else {
            e = pend_q[lane];
        }
        key = ((u64)f2ord(__uint_as_float(e.x)) << 32) | (u64)e.y;
    }
    // ascending across the lanes that hold entries (lanes 32..63 are all sentinel when 2 HCAP <= 32)
    key = (2 * HCAP <= 32) ? wave_sort<32>(key, lane) : wave_sort<64>(key, lane);
    u64 old = LIST_SENTINEL;
    if (inited)  // written earlier by this same wave: read it back from L2, not a stale L1 line
        old = __hip_atomic_load(list_q + (63 - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    key = umin64(key, old);  // the 64 smallest of the union, as a bitonic sequence
    key = wave_merge(key, lane);
    list_q[lane] = key;
    // the threshold is the kt-th smallest key, not the 64th: fewer candidates pass, and
    // everything outside the 64-entry list still scores at least this much
    unsigned top = __builtin_amdgcn_readlane((unsigned)(key >> 32), kt - 1);
    return ord2f(top);
}

// what the stage loop does with a finished subtile: smallest score against the threshold
#define SCMIAC_TEST_PREV(PRV_)                                                                      \
    do {                                                                                            \
        if (DIAG_ && (ablate & 128)) break; /* diagnostic: the stream without the per-subtile test */ \
            /* smallest of the 16 scores as a depth-3 tree: a wave waits its turn for every */      \
            /* instruction that depends on the one before: depth costs, not count           */      \
            const float m0_ = min3_after_mfma(accs[PRV_][0], accs[PRV_][1], accs[PRV_][2]);         \
            const float m1_ = min3_raw_after(accs[PRV_][3], accs[PRV_][4], accs[PRV_][5], m0_);     \
            const float m2_ = min3_raw_after(accs[PRV_][6], accs[PRV_][7], accs[PRV_][8], m0_);     \
            const float m3_ = min3_raw_after(accs[PRV_][9], accs[PRV_][10], accs[PRV_][11], m0_);   \
            const float m4_ = min3_raw_after(accs[PRV_][12], accs[PRV_][13], accs[PRV_][14], m0_);  \
            const float mn_ = min2_raw(min3_raw(m0_, m1_, m2_), min3_raw(m3_, m4_, accs[PRV_][15])); \
            if (__builtin_expect(__any(mn_ < tau_hi), 0)) select_slow(accs[PRV_], ref0s[PRV_]);     \
    } while (0)

// ... and during the seed pass over the home stages: running minimum of every accumulator
// register, even and odd subtiles apart (the accumulator set is the subtile's parity)
#define SCMIAC_SEED_PREV(PRV_)                                                                      \
    do {                                                                                            \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) gmin[PRV_][r] = fminf(gmin[PRV_][r], accs[PRV_][r]); \
    } while (0)

// ascending sort of 32 registers (bitonic network, every index a compile-time constant)
__device__ __forceinline__ void sort32_regs(float (&v)[32]) {
#pragma unroll
    for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int p = i ^ j;
                if (p > i) {
                    const float lo = fminf(v[i], v[p]), hi = fmaxf(v[i], v[p]);
                    if ((i & k) == 0) { v[i] = lo; v[p] = hi; } else { v[i] = hi; v[p] = lo; }
                }
            }
        }
    }
}

// KIND_: 0 the product kernel of the dense search; 2 the product kernel of the pruned search
// (blocks skip the cells of the reference image out of their queries' reach); 1 the diagnostic
// instantiation: per-wave counters, ablations, and the pruning code as well.
template <int D2, int NW_, int RT_, int PCAP_, bool GP_, int KIND_>
__global__ __launch_bounds__(NW_ * 64, (NW_ >= 8 ? NW_ / 4 : 2))
// exactly this many waves per SIMD: the LDS allows one block per CU, so registers saved
// beyond that buy nothing, and the scheduler should not trade LDS-read lookahead for them
__attribute__((amdgpu_waves_per_eu((NW_ >= 8 ? NW_ / 4 : 2), (NW_ >= 8 ? NW_ / 4 : 2)))) void fused_knn_kernel(
    const float* __restrict__ Qp,     // [n_pad][ROWF], already scaled by -2
    const float* __restrict__ Rp,     // [m_pad][LS], zero pad columns
    const float* __restrict__ rnorm,  // [m_pad], 1.7e38 on padding rows
    u64* __restrict__ lists,          // [S][n_pad][64], then the tail region's extra lists
    uint2* __restrict__ pend_g,       // [shortlists][PCAP] pending candidates (GP_ only)
    long long n, long long n_pad, int nqb, long long chunk_rows, FusedTail tail, int kt, int ablate_in,
    const int* __restrict__ home,     // [nqb][4] home window and own cells of every query block (stages), or NULL
    int seed_stages,                  // most home stages the seed pass looks at
    const float* __restrict__ tau0,   // [n_pad] threshold to start from, or NULL
    float* __restrict__ tau_out,      // [shortlists] threshold each shortlist ended with
    PruneArgs prune,                  // pruned search (KIND_ 1 and 2; qpd == NULL: search everything)
    u64* __restrict__ diag_out /* per wave 16 counters (diagnostic instantiation only) */) {
    constexpr bool DIAG_ = KIND_ == 1, PRUNE_ = KIND_ != 0;
    u64* const diag = DIAG_ ? diag_out : nullptr;  // compile-time NULL in the product kernels (may be NULL in the other too)
    const int ablate = DIAG_ ? ablate_in : 0;      // ablations exist in the diagnostic instantiation only
    using C = FusedCfg<D2, NW_, RT_, PCAP_, GP_>;
    constexpr int D2P = C::D2P, ROWF = C::ROWF, LS = C::LS, RT = C::RT, NW = C::NW;
    constexpr int PCAP = C::PCAP, NSUB = RT / 32, NB = C::NB;
    constexpr int STAGE_BYTES = RT * LS * 4;
    constexpr int NCHUNK = STAGE_BYTES / 1024;  // 1 KiB = one wave-wide 16-byte LDS-DMA load
    constexpr int LD_ITERS = (NCHUNK + NW - 1) / NW;
    constexpr int NORM_CHUNKS = RT / 64;        // 256 B = one wave-wide 4-byte LDS-DMA load
    static_assert(STAGE_BYTES % 1024 == 0, "a stage is a whole number of wave-wide loads");
    static_assert(RT % 64 == 0 && NORM_CHUNKS <= NW, "norms of a stage: one load per wave at most");
    (void)ROWF;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tile = reinterpret_cast<float*>(smem);             // [NB][RT*LS]
    float* tnorm = tile + NB * RT * LS;                       // [NB][RT]
    uint2* pend = reinterpret_cast<uint2*>(tnorm + NB * RT);  // [NW][32][PCAP]
    const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tile;
    const unsigned tnorm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tnorm;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;
    // Blocks below tail.block0 search reference chunk `split` of query block `qb`; the
    // blocks from tail.block0 on belong to the last, partly filled round of the grid and
    // cut their query blocks' reference range into tail.St shorter chunks instead.
    int split, qb;
    long long r_begin, rows_here;
    long long list_row;  // row of the [.][64] shortlist array this wave's first query uses
    if ((int)blockIdx.x < tail.block0) {
        // Blocks are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8), each with an L2
        // of its own.  Within every aligned group of 256 block numbers, XCD x takes 32 CONSECUTIVE
        // places of the (chunk, query block) order: neighbouring query blocks share cells, so their
        // seed passes and home stages read the same reference rows and take about as long, and the
        // 32 blocks then walk the rest of their chunk in step -- out of their XCD's L2.
        int b = (int)blockIdx.x;
        if ((b | 255) < tail.block0) b = (b & ~255) | ((b & 7) << 5) | ((b & 255) >> 3);
        split = b / tail.qb0; qb = b % tail.qb0;  // main region: qb0 query blocks
        r_begin = (long long)split * chunk_rows;
        rows_here = chunk_rows;
        list_row = (long long)split * n_pad + (long long)qb * C::QPB + wave * 32;
    } else {
        const int tb = (int)blockIdx.x - tail.block0;
        split = tb % tail.St; qb = tail.qb0 + tb / tail.St;
        r_begin = (long long)split * tail.chunk_rows;
        rows_here = tail.m_pad - r_begin;
        rows_here = rows_here < 0 ? 0 : (rows_here < tail.chunk_rows ? rows_here : tail.chunk_rows);
        const long long qrow = (long long)qb * C::QPB + wave * 32;
        list_row = split == 0 ? qrow
                              : tail.lists_row0 + (long long)(split - 1) * tail.rows + (qrow - (long long)tail.qb0 * C::QPB);
    }
    const long long q0 = (long long)qb * C::QPB + wave * 32;  // first query of this wave
    const int T = (int)(rows_here / RT);
    // The order in which this block streams the reference stages of its chunk (global stage g =
    // rows g RT .. g RT + RT - 1 of the packed image): the home stages of its queries that lie
    // inside the chunk first -- their near neighbours, which bring the thresholds down to where
    // they will stay -- then the rest.
    const int G0 = (int)(r_begin / RT);
    int H0 = 0, HN = 0, OW0 = 0, OWN = 0;  // home window; the queries' own cells inside it
    if (home != nullptr && T > 0) { H0 = home[4 * qb]; HN = home[4 * qb + 1]; OW0 = home[4 * qb + 2]; OWN = home[4 * qb + 3]; }
    int O0 = H0 > G0 ? H0 : G0;                                  // home stages inside the chunk: [O0, O0 + oN)
    const int O1 = (H0 + HN < G0 + T) ? H0 + HN : G0 + T;
    const int oN = O1 > O0 ? O1 - O0 : 0;
    if (oN == 0) O0 = G0;
    // Blocks that start together should also walk the rest of the reference rows together
    // (each stage then comes out of the L2 for all but the first of them), although their seed
    // passes and home stages differ in length: a block that has spent `shift` stage times on
    // those enters the rest of its chunk `shift` stages in and wraps round at the end.
    // seed pass stages (own cells' head): no more than a quarter of the stages the block is about to
    // stream -- the short pieces of the tail would otherwise spend as long on it as on their chunk
    int SN = (home != nullptr && T > 0) ? (OWN < seed_stages ? OWN : seed_stages) : 0;
    { const int cap = T / 4 > 3 ? T / 4 : 3; SN = SN < cap ? SN : cap; }
    // (pruned search: the reach test needs a threshold that every query has earned among its own
    // cell's rows.  A chunk that holds those rows streams them before the test; a block whose chunk
    // does not -- one of several chunks of a query block -- looks at all of them in its seed pass.)
    if (PRUNE_ && prune.qpd != nullptr && SN > 0 && !(OW0 >= G0 && OW0 + OWN <= G0 + T)) SN = OWN;
    const int rest = T - oN;
    const int shift = rest > 0 ? (SN + oN) % rest : 0;
    // (pruned search: after its seed pass a block may decide to stream only part of the rest
    // of its chunk; the chunk-relative numbers of those stages then stand in `order`)
    const unsigned short* order = nullptr;
    int T_loop = T;
    auto stage_row = [&](int i) __attribute__((always_inline)) {  // first reference row of the i-th stage streamed
        int g;
        if (PRUNE_ && order != nullptr && i >= oN) {
            g = G0 + (int)__hip_atomic_load(order + (i - oN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (i < oN) {
            g = O0 + i;
        } else {
            int x = i - oN + shift;
            if (x >= rest) x -= rest;
            g = G0 + x;
            if (g >= O0) g += oN;
        }
        return (long long)g * RT;
    };
    constexpr int HCAP = PCAP / 2;  // each lane appends to a buffer of its own: no atomics
    uint2* pend_w = GP_ ? pend_g + list_row * PCAP : pend + wave * 32 * PCAP;
    uint2* pend_l = pend_w + (j * 2 + hh) * HCAP;  // this lane's buffer: query j, lane half hh
    unsigned pcnt = 0u;                            // entries in it
    u64* lists_w = lists + list_row * 64;

    // B operand: this lane's query features, one register per MFMA step
    float qf[D2P];
    {
        const float4* qp4 = reinterpret_cast<const float4*>(Qp + (q0 + j) * ROWF + hh * D2P);
#pragma unroll
        for (int c = 0; c < D2P / 4; ++c) {
            float4 v = qp4[c];
            qf[4 * c + 0] = v.x; qf[4 * c + 1] = v.y; qf[4 * c + 2] = v.z; qf[4 * c + 3] = v.w;
        }
    }
    // The loads above must be complete before the stage loop starts: a compiler-inserted
    // wait for them inside the loop would also drain the LDS-DMA loads of the next stage.
#pragma unroll
    for (int c = 0; c < D2P; ++c) asm volatile("" : "+v"(qf[c]));
    // running threshold: score of the 64th best reference so far; padding queries take nothing
    const float tau_max = __uint_as_float(TAU_MAX_BITS);  // padding rows score exactly this
    float tau_start = tau_max;
    if (tau0 != nullptr) tau_start = fminf(tau0[q0 + j], tau_max);
    float tau = (q0 + j < n && !(ablate & 1)) ? tau_start : -__builtin_inff();
    // The per-subtile test looks at raw scores; a key is within 16 ulp (2^-19 relative) of
    // its score, so "some key below tau" implies "some score below tau_hi".
    auto widen = [](float t) __attribute__((always_inline)) {
        return t > -__builtin_inff() ? t + fabsf(t) * 3.814697265625e-06f + 1e-30f : t;  // 2^-18 relative
    };
    float tau_hi = widen(tau);
    unsigned inited = 0u;  // bit q: query q's shortlist exists in global memory (wave-uniform)


    u64 dg_slow = 0, dg_flush = 0, dg_nslow = 0, dg_nflush = 0, dg_npush = 0, dg_iters = 0;
    u64 dg_seg1 = 0, dg_seg2 = 0, dg_ngen = 0, dg_tseed = 0, dg_thome = 0, dg_tloop = 0;
    const u64 dg_t0 = diag ? (u64)clock64() : 0;

    // merge every query of this wave that has a full buffer (only_full) or any pending
    // candidate, or, at the end, no shortlist in memory yet
    auto flush_full = [&](bool only_full, bool force_uninit) __attribute__((always_inline)) {
        const u64 tf0 = diag ? (u64)clock64() : 0;
        // the appended entries have reached L2 (stores write through; the merge reads past the L1)
        if constexpr (GP_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bool need = only_full ? (pcnt >= (unsigned)HCAP) : (pcnt > 0u);
        need = need || (force_uninit && !((inited >> j) & 1u) && (q0 + j < n));
        const u64 m64 = __ballot(need);
        u64 mask = (m64 | (m64 >> 32)) & 0xffffffffull;
        while (mask) {
            int q = __builtin_ctzll(mask);
            mask &= mask - 1;
            const unsigned c0 = __builtin_amdgcn_readlane(pcnt, q);
            const unsigned c1 = __builtin_amdgcn_readlane(pcnt, q + 32);
            float nt = flush_query<HCAP, GP_>(c0, c1, pend_w + q * PCAP, lists_w + (long long)q * 64,
                                         (inited >> q) & 1u, lane, kt);
            inited |= 1u << q;
            // (never above the threshold the query had: a list that holds fewer than kt entries so far
            // says nothing, and the bound of the seed pass or of the running result still stands)
            if (j == q) { tau = fminf(tau, fminf(nt, tau_max)); tau_hi = widen(tau); pcnt = 0u; }
            if (diag) ++dg_nflush;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (diag) dg_flush += (u64)clock64() - tf0;
    };

    // Candidate path (entered when some lane's smallest score is below its threshold): the
    // lanes whose keys pass append them to their own buffer.  Every
    // step that depends on the one before costs a wave hundreds of cycles while its SIMD
    // partners keep the MFMA pipe busy, so the path is kept shallow: no cross-lane
    // traffic, no atomics, no search for minima.
    auto select_slow = [&](const f32x16& acc, unsigned ref0_wave) __attribute__((always_inline)) {
        const unsigned ref0 = ref0_wave + 4u * (unsigned)hh;  // first reference row of this lane half
        const u64 ts0 = diag ? (u64)clock64() : 0;
        if (diag) ++dg_nslow;
        if (ablate & 8) {  // diagnostic: price of reaching this path and returning
            if (diag) dg_slow += (u64)clock64() - ts0;
            return;
        }
        // (the empty statement keeps the compiler from hoisting this path's arithmetic above
        // the branch that guards it, into the per-subtile code every wave runs)
        f32x16 sc = acc;
        asm volatile("" : "+v"(sc));
        float key[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) key[r] = score_key(sc[r], r);
        // smallest key of the lane, as a depth-3 tree (every dependent step costs a turn)
        const float a0 = min3_raw(key[0], key[1], key[2]), a1 = min3_raw(key[3], key[4], key[5]);
        const float a2 = min3_raw(key[6], key[7], key[8]), a3 = min3_raw(key[9], key[10], key[11]);
        const float a4 = min3_raw(key[12], key[13], key[14]);
        const float kmin = min2_raw(min3_raw(a0, a1, a2), min3_raw(a3, a4, key[15]));
        // which of the lane's 16 keys pass: one bit each, gathered by a shallow tree
        unsigned bit[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bit[r] = (key[r] < tau) ? (1u << r) : 0u;
        const unsigned m0 = bit[0] | bit[1] | bit[2], m1 = bit[3] | bit[4] | bit[5], m2 = bit[6] | bit[7] | bit[8];
        const unsigned m3 = bit[9] | bit[10] | bit[11], m4 = bit[12] | bit[13] | bit[14];
        const unsigned mask = (m0 | m1 | m2) | (m3 | m4 | bit[15]);
        const unsigned cnt = (unsigned)__builtin_popcount(mask);
        const u64 ts1 = diag ? (u64)clock64() : 0;
        if (diag) dg_seg1 += ts1 - ts0;  // keys and pass mask
        if (ablate & 16) {  // diagnostic: price of the arithmetic above without any store
            asm volatile("" ::"v"(mask), "v"(kmin));
            return;
        }
        if (__builtin_expect(!__any(pcnt + cnt > (unsigned)HCAP), 1)) {
            // every lane has room for all its candidates
            if (!__any(cnt > 1u)) {
                // at most one per lane, and then it is the lane's smallest key, which names its own
                // register (key register r holds subtile row (r & 3) + 8 (r >> 2) = r + (r & 12))
                const unsigned kb = __float_as_uint(kmin);
                if (cnt) pend_l[pcnt] = make_uint2(kb, ref0 + (kb & 15u) + (kb & 12u));
            } else {
                // sixteen independent predicated stores, each to the slot its rank among the lane's
                // passing keys gives it
                if (diag) ++dg_ngen;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned pos = pcnt + (unsigned)__builtin_popcount(mask & ((1u << r) - 1u));
                    if (mask & (1u << r))
                        pend_l[pos] = make_uint2(__float_as_uint(key[r]), ref0 + (unsigned)((r & 3) + 8 * (r >> 2)));
                }
            }
            pcnt += cnt;
            if (diag) { ++dg_iters; dg_npush += __popcll(__ballot(cnt > 0u)); dg_seg2 += (u64)clock64() - ts1; }
        } else {
            unsigned skipped = 0u;  // bit r: key[r] passed but this lane's buffer was full
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool p = key[r] < tau;
                if (__any(p)) {
                    const bool room = pcnt < (unsigned)HCAP;
                    if (p && room) {
                        pend_l[pcnt] = make_uint2(__float_as_uint(key[r]), ref0 + (unsigned)((r & 3) + 8 * (r >> 2)));
                        ++pcnt;
                    }
                    if (p && !room) skipped |= 1u << r;
                    if (diag) { ++dg_iters; dg_npush += __popcll(__ballot(p && room)); }
                }
            }
            while (__any(skipped != 0u)) {  // merge the full buffers, then the keys left over
                flush_full(true, false);
                unsigned still = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool p = ((skipped >> r) & 1u) && key[r] < tau;  // against the tightened threshold
                    if (__any(p)) {
                        const bool room = pcnt < (unsigned)HCAP;  // a buffer shorter than 16 can fill up again
                        if (p && room) {
                            pend_l[pcnt] = make_uint2(__float_as_uint(key[r]), ref0 + (unsigned)((r & 3) + 8 * (r >> 2)));
                            ++pcnt;
                        }
                        if (p && !room) still |= 1u << r;
                        if (diag) { ++dg_iters; dg_npush += __popcll(__ballot(p && room)); }
                    }
                }
                skipped = still;
            }
        }
        if (__builtin_expect(__any(pcnt >= (unsigned)HCAP), 0)) flush_full(true, false);  // keep room for next time
        if (diag) dg_slow += (u64)clock64() - ts0;
    };

    // Software pipeline over subtiles (32 references each) and, within a subtile, over groups
    // of four MFMA steps.  The A-operand fragments live in a ring of three four-register
    // groups: the LDS read for group g + 2 is issued just before the MFMAs of group g (two
    // groups = eight MFMAs ahead, far more than the LDS latency), running on into the next
    // subtile's first groups.  A whole double-buffered fragment set would cost 44 more VGPRs,
    // and with them the compiler's freedom to place these reads.  While subtile s's chain
    // runs, the wave also tests subtile s-1 (other accumulator set) after its first group.
    constexpr int NGRP = D2P / 4;  // groups of four steps (the last one may be partly padding)
    static_assert(NGRP >= 2, "two groups of lookahead");
    f32x16 accs[2];
    float fr[3][4];
    unsigned ref0s[2] = {0u, 0u};
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[1][r] = tau_max;  // "previous subtile" of the first one

    const bool do_stage = !(ablate & 2);  // diagnostic: keep re-reading the first stage
    const bool do_sync = !(ablate & 4);   // diagnostic: no block barrier (results are garbage)


    // Seed pass.  Before anything is selected the block streams (some of) its queries' home
    // stages once with no selection at all: every lane keeps the running minimum of each of its
    // 16 accumulator registers, even and odd subtiles apart -- 32 minima over 32 DISJOINT sets
    // of home rows per lane, 64 per query.  If v is the 18th smallest of a lane's 32, at least
    // 18 rows of that lane half score no more than v; with the other lane half's v, at least
    // 36 >= kt rows score no more than the larger of the two: an upper bound of the query's
    // kt-th smallest score, typically near the 60th smallest score of the rows looked at.  The
    // search starts from it instead of from "everything passes", which removes the cold start
    // (hundreds of candidates and half a dozen merges per query).
    if (SN > 0 && !(ablate & 32)) {
        float gmin[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { gmin[0][r] = tau_max; gmin[1][r] = tau_max; }
        SCMIAC_STAGE((long long)OW0 * RT, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < SN; ++t) {
            if (t + 1 < SN) SCMIAC_STAGE((long long)(OW0 + t + 1) * RT, (t + 1) & 1);
            const float* tb = tile + (t & 1) * RT * LS;
            const float* nb = tnorm + (t & 1) * RT;
            const long long cur_row0 = (long long)(OW0 + t) * RT;
            SCMIAC_LOAD_NORMS(0, 0);
            SCMIAC_LOAD_GROUP(0, 0, 0);
            SCMIAC_LOAD_GROUP(0, 1, 1);
            SCMIAC_STAGE_SUBTILES(SCMIAC_SEED_PREV);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        {
            SCMIAC_SEED_PREV(1);  // the last subtile
            float all[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) { all[r] = gmin[0][r]; all[16 + r] = gmin[1][r]; }
            sort32_regs(all);
            static_assert(SCMIAC_KNN_MAX_K_INTERNAL <= 64, "half of kt fits a lane half's 32 minima");
            // the (kt / 2 rounded up)-th smallest of each lane half, the larger of the two halves
            float mine = all[0];
            const int need = (kt + 1) / 2;  // 1..32, wave-uniform
#pragma unroll
            for (int i = 1; i < 32; ++i) mine = (i < need) ? all[i] : mine;
            const float other = __shfl_xor(mine, 32);
            float bound = fmaxf(mine, other);
            // the search compares keys (scores with four replaced mantissa bits): widen like tau_hi
            bound = bound < tau_max ? widen(bound) : tau_max;
            if (q0 + j < n && !(ablate & 1)) { tau = fminf(tau, bound); tau_hi = widen(tau); }
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[1][r] = tau_max;  // "previous subtile" of the first one, again
        }
    }

    if (diag) dg_tseed = (u64)clock64() - dg_t0;

    // Pruned search.  The partition's pivots give, for every cell c of the reference image,
    // a lower bound of the distance from a query to anything in it: dist(q, p_c) - rho_c (rho_c
    // the largest distance of the cell's rows to their pivot).  At least kt rows score no more
    // than the query's threshold tau, i.e. lie within reach = sqrt(tau + |q|^2 + E) of it (E the
    // score's error bound), and tau only falls from here on: a cell beyond the reach of all of
    // the block's queries cannot hold one of their neighbours.  The block streams its home stages
    // and the stages of the cells within reach, nothing else; every row left out has a true
    // distance above every bound the refine certificate uses, so the result stays exact.
    // It happens in two steps, because the queries of a block can belong to two cells while its
    // home stages (at most 8 192 rows) may cover only part of the first: the queries of the
    // second would then still carry a loose threshold and keep every cell.  Step 1 (before the
    // stage loop, no test): after the home stages the block streams the rest of its queries' own
    // cells.  Step 2 (when those are nearly done and every query has met its own cell): the reach
    // test decides which other cells follow.
    bool pruning = false;
    int replan_at = -1;          // iteration at whose start step 2 runs
    int own_lo = 0, own_hi = -1; // chunk-relative stages of the queries' own cells
    unsigned short* my_order = nullptr;
    auto plan_rest = [&]() __attribute__((always_inline)) {  // step 2; every thread of the block calls it
        if constexpr (PRUNE_) {
            constexpr int QPB = NW_ * 32, NT = NW_ * 64;
            __shared__ float reach_s[QPB];
            __shared__ int qrow_s[QPB];
            __shared__ unsigned cellmask[KNN_MAX_CELLS / 32];
            __shared__ int wave_cnt[NW_];
            float part = 0.0f;
#pragma unroll
            for (int c = 0; c < D2P; ++c) part = fmaf(0.25f * qf[c], qf[c], part);  // qf holds -2 q
            const float qn = part + __shfl_xor(part, 32);
            const float rmax2 = __uint_as_float(*prune.maxnorm) * 1.00001f;
            float reach = -__builtin_inff();  // padding queries need nothing
            if (q0 + j < n) {
                const float t2 = (tau + qn + prune.err_coef * (rmax2 + qn)) * 1.00001f;
                reach = tau < tau_max ? sqrtf(fmaxf(t2, 0.0f)) * 1.000001f : __builtin_inff();
            }
            if (hh == 0) {
                reach_s[wave * 32 + j] = reach;
                qrow_s[wave * 32 + j] = (q0 + j < n) ? prune.perm_q[q0 + j] : -1;
            }
            const int C = prune.cells;
            for (int w = tid; w < (C + 31) / 32; w += NT) cellmask[w] = 0u;
            __syncthreads();
            for (int idx = tid; idx < C * QPB; idx += NT) {
                const int c = idx / QPB, ql = idx - c * QPB;
                const int qrow = qrow_s[ql];
                bool need = false;
                if (qrow >= 0) {
                    const float dq = prune.qpd[(long long)c * prune.n + qrow];
                    const float rho = __uint_as_float(prune.crad[c]);
                    need = !((dq - rho) * 0.99999f > reach_s[ql]);  // (NaN anywhere: the cell stays)
                }
                // (a wave's 64 pairs share the cell or straddle two of them: QPB is a multiple of 64)
                if (__any(need) && lane == 0) atomicOr(&cellmask[c >> 5], 1u << (c & 31));
            }
            __syncthreads();
            // The stages that hold rows of a cell within reach, ascending, without the home stages and
            // the own cells' (already streamed): every thread looks at one stage per round -- which
            // cells its rows belong to (the cells lie one after the other in the packed image) -- and
            // the kept stage numbers are appended in order (ballot ranks within a wave, wave totals
            // through LDS).
            int cnt = T_loop - oN;  // entries of the list already there (step 1)
            for (int g0 = 0; g0 < T; g0 += NT) {
                const int g = g0 + tid;
                bool keep = false;
                if (g < T && !(oN > 0 && g >= O0 - G0 && g < O0 - G0 + oN) && !(g >= own_lo && g <= own_hi)) {
                    const long long ra = (long long)(G0 + g) * RT, rb = ra + RT;  // rows [ra, rb)
                    if (ra < prune.m) {
                        // c1: the cell that holds row ra (last cell starting at or before it);
                        // c2: the last cell starting before rb
                        int lo = 0, hi = C;  // invariant: cstart[lo] <= ra < cstart[hi] (cstart[C] = m > ra)
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if ((long long)prune.cstart_r[mid] <= ra) lo = mid; else hi = mid;
                        }
                        const int c1 = lo;
                        int c2 = c1;
                        while (c2 + 1 < C && (long long)prune.cstart_r[c2 + 1] < rb) ++c2;  // a stage spans few cells
                        for (int c = c1; c <= c2; ++c) keep = keep || ((cellmask[c >> 5] >> (c & 31)) & 1u);
                    }
                }
                const u64 bal = __ballot(keep);
                if (lane == 0) wave_cnt[wave] = __popcll(bal);
                __syncthreads();
                int before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < NW_; ++w) { const int c = wave_cnt[w]; before += (w < wave) ? c : 0; total += c; }
                if (keep) my_order[cnt + before + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)g;
                cnt += total;
                __syncthreads();
            }
            // (the list was written with plain stores by many threads and is read back from the L2)
            __threadfence();
            __syncthreads();
            T_loop = oN + cnt;
            if (tid == 0 && prune.stats != nullptr) {
                atomicAdd(prune.stats, (u64)T_loop);
                atomicAdd(prune.stats + 1, (u64)T);
            }
        }
    };
    if constexpr (PRUNE_) {
        if (prune.qpd != nullptr && home != nullptr && T > 0 && T < prune.order_stride) {
            constexpr int NT = NW_ * 64;
            my_order = prune.order + (long long)blockIdx.x * prune.order_stride;
            // step 1: the stages of the queries' own cells (the block's queries are consecutive in
            // the packed order, i.e. sorted by cell), without the home stages
            const long long qa = (long long)qb * (NW_ * 32);
            long long qz = qa + NW_ * 32 - 1;
            if (qz >= n) qz = n - 1;
            int n_own = 0;
            if (qa < n) {
                const int c_lo = prune.cell_q[prune.perm_q[qa]], c_hi = prune.cell_q[prune.perm_q[qz]];
                const long long a = prune.cstart_r[c_lo], b = prune.cstart_r[c_hi + 1];
                if (b > a) {
                    int lo = (int)(a / RT) - G0, hi = (int)((b - 1) / RT) - G0;
                    if (lo < 0) lo = 0;
                    if (hi >= T) hi = T - 1;
                    if (hi >= lo) {
                        own_lo = lo; own_hi = hi;
                        // the home stages inside [lo, hi] are left out: [hl, hr)
                        int hl = O0 - G0, hr = O0 - G0 + oN;
                        if (hl < lo) hl = lo;
                        if (hr > hi + 1) hr = hi + 1;
                        const int cut = (oN > 0 && hr > hl) ? hr - hl : 0;
                        n_own = hi - lo + 1 - cut;
                        for (int i = tid; i < n_own; i += NT) {
                            int g = lo + i;
                            if (cut > 0 && g >= hl) g += cut;
                            my_order[i] = (unsigned short)g;
                        }
                    }
                }
            }
            __threadfence();
            __syncthreads();
            order = my_order;
            T_loop = oN + n_own;
            pruning = true;
            if (T_loop == 0) plan_rest();       // nothing of its own in this chunk: decide at once
            else replan_at = T_loop - 1;        // ... else while the last home / own stage is being fetched
        }
    }


    {   // Two stage buffers, one block barrier per stage.
        if (T_loop > 0) SCMIAC_STAGE(stage_row(0), 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < T_loop; ++t) {
            if (diag && t == oN) dg_thome = (u64)clock64() - dg_t0;
            if (PRUNE_ && pruning && t == replan_at) plan_rest();
            if (do_stage && t + 1 < T_loop) SCMIAC_STAGE(stage_row(t + 1), (t + 1) & 1);
            const float* tb = tile + (do_stage ? (t & 1) : 0) * RT * LS;
            const float* nb = tnorm + (do_stage ? (t & 1) : 0) * RT;
            const long long cur_row0 = stage_row(t);
            // |r|^2 and fragments of the stage's first subtile (the stage only became readable at
            // the barrier; set 0 was tested during the previous stage's last subtile)
            SCMIAC_LOAD_NORMS(0, 0);
            SCMIAC_LOAD_GROUP(0, 0, 0);
            SCMIAC_LOAD_GROUP(0, 1, 1);
            SCMIAC_STAGE_SUBTILES(SCMIAC_TEST_PREV);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA loads have landed
            if (do_sync) __syncthreads();
        }
    }
    if (diag) { dg_tloop = (u64)clock64() - dg_t0; if (oN >= T_loop) dg_thome = dg_tloop; }
    if (T_loop > 0) {  // the last subtile is still untested
        float mn_ = accs[1][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mn_ = fminf(mn_, accs[1][r]);
        if (__any(mn_ < tau_hi)) select_slow(accs[1], ref0s[1]);
    }
    flush_full(false, true);
    // what the refine pass may assume of every reference this block streamed and did not list
    if (hh == 0) tau_out[list_row + j] = tau;
    if (diag && lane == 0) {
        u64* o = diag + ((long long)blockIdx.x * NW + wave) * 16;
        o[0] = (u64)clock64() - dg_t0; o[1] = dg_slow; o[2] = dg_flush; o[3] = dg_nslow;
        o[4] = dg_nflush; o[5] = dg_npush; o[6] = dg_iters; o[7] = dg_seg1;
        o[8] = dg_seg2; o[9] = dg_ngen; o[10] = dg_tseed; o[11] = dg_thome;
        o[12] = dg_tloop; o[13] = (u64)T_loop; o[14] = (u64)oN; o[15] = (u64)SN;
    }
}

#undef SCMIAC_TEST_PREV
#undef SCMIAC_SEED_PREV

template <int D2, int NW, int RT, int PC, bool GP, int KIND>
hipError_t launch_fused_t(const float* Qp, const float* Rp, const float* rnorm, u64* lists, uint2* pend_g,
                                 long long n, long long n_pad, int nqb, int S, long long chunk_rows,
                                 const FusedTail& tail, int kt, int ablate, const int* home, int seed_stages,
                                 const float* tau0, float* tau_out, const PruneArgs& prune, u64* diag, hipStream_t stream) {
    using C = FusedCfg<D2, NW, RT, PC, GP>;
    size_t lds = C::lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_knn_kernel<D2, NW, RT, PC, GP, KIND>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(tail.block0 + (nqb - tail.qb0) * tail.St)), block(NW * 64);
    hipLaunchKernelGGL((fused_knn_kernel<D2, NW, RT, PC, GP, KIND>), grid, block, lds, stream, Qp, Rp, rnorm, lists, pend_g, n,
                       n_pad, nqb, chunk_rows, tail, kt, ablate, home, seed_stages, tau0, tau_out, prune, diag);
    return hipGetLastError();
}


// blocks of the fused kernel the GPU runs at once (256 one-block CUs when no device answers)
template <int D2, int NW, int RT, int PC, bool GP>
int fused_occupancy_t() {
    constexpr int KIND = 0;
    using C = FusedCfg<D2, NW, RT, PC, GP>;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        return 256;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_knn_kernel<D2, NW, RT, PC, GP, KIND>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes());
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_knn_kernel<D2, NW, RT, PC, GP, KIND>, NW * 64,
                                                     C::lds_bytes()) != hipSuccess || per_cu <= 0) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    return cus * per_cu;
}

// dispatch of one translation unit's variants (VARIANTS_ = its list macro): the dense product
// kernel and the diagnostic instantiation
#define SCMIAC_DEFINE_FUSED_TU(N_, VARIANTS_)                                                                          \
    hipError_t launch_fused_tu##N_(const KnnGeometry& g, const FusedLaunch& a) {                                       \
        if (a.kind >= 2) return hipErrorNotFound; /* pruned and stream-sharing kernels live in units of their own */   \
        VARIANTS_(SCMIAC_TU_LAUNCH_CASE)                                                                               \
        return hipErrorNotFound;                                                                                       \
    }                                                                                                                  \
    int fused_occupancy_tu##N_(const KnnGeometry& g) {                                                                 \
        VARIANTS_(SCMIAC_TU_OCC_CASE)                                                                                  \
        return 0;                                                                                                      \
    }
#define SCMIAC_TU_ARGS_(PRUNE_)                                                                                        \
    a.Qp, a.Rp, a.rnorm, a.lists, static_cast<uint2*>(a.pend_g), a.n, g.n_pad, g.nqb, g.S, g.chunk_rows, g.tail, a.kt,  \
        a.ablate, a.home, g.seed_stages, a.tau0, a.tau_out, PRUNE_, a.diag, a.stream
#define SCMIAC_TU_LAUNCH_CASE(V_, NW_, RT_, PC_, GP_)                                                                  \
    if (g.D2 == V_ && g.nw == NW_ && g.RT == RT_ && g.pcap == PC_ && g.gp == GP_)                                      \
        return a.kind == 1 ? launch_fused_t<V_, NW_, RT_, PC_, (GP_ != 0), 1>(SCMIAC_TU_ARGS_(a.prune))                \
                           : launch_fused_t<V_, NW_, RT_, PC_, (GP_ != 0), 0>(SCMIAC_TU_ARGS_(PruneArgs{}));
#define SCMIAC_TU_OCC_CASE(V_, NW_, RT_, PC_, GP_)                                                                     \
    if (g.D2 == V_ && g.nw == NW_ && g.RT == RT_ && g.pcap == PC_ && g.gp == GP_)                                      \
        return fused_occupancy_t<V_, NW_, RT_, PC_, (GP_ != 0)>();
// ... and the unit of the pruned product kernels
#define SCMIAC_DEFINE_PRUNED_TU(N_, VARIANTS_)                                                                         \
    hipError_t launch_fused_tu##N_(const KnnGeometry& g, const FusedLaunch& a) {                                       \
        if (a.kind != 2) return hipErrorNotFound;                                                                      \
        VARIANTS_(SCMIAC_TU_PRUNED_CASE)                                                                               \
        return hipErrorNotFound;                                                                                       \
    }                                                                                                                  \
    bool fused_pruned_compiled_tu##N_(const KnnGeometry& g) {                                                          \
        VARIANTS_(SCMIAC_TU_HAS_CASE)                                                                                  \
        return false;                                                                                                  \
    }
#define SCMIAC_TU_PRUNED_CASE(V_, NW_, RT_, PC_, GP_)                                                                  \
    if (g.D2 == V_ && g.nw == NW_ && g.RT == RT_ && g.pcap == PC_ && g.gp == GP_)                                      \
        return launch_fused_t<V_, NW_, RT_, PC_, (GP_ != 0), 2>(SCMIAC_TU_ARGS_(a.prune));
#define SCMIAC_TU_HAS_CASE(V_, NW_, RT_, PC_, GP_)                                                                     \
    if (g.D2 == V_ && g.nw == NW_ && g.RT == RT_ && g.pcap == PC_ && g.gp == GP_) return true;

}  // namespace scmiac
