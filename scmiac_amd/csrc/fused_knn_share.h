// The fused distance + top-k kernel for searches whose main region is a single round of the
// grid: every block then owns one (query block, reference chunk) stream for the whole launch, and
// the launch ends with its slowest block -- the one whose queries met the most candidates --
// while the compute units of the others idle.  Here a stream's stages after the home stages are
// handed out in small runs through a counter, to its owner and to whichever blocks have run out
// of work of their own ("helpers": they take the owner's queries and the thresholds it published
// after its home stages, and keep a shortlist of their own that the refine pass merges like a
// chunk's).  The tail blocks -- the short pieces of the query blocks beyond the round -- come
// first in the grid, so that what is left to balance at the end is the finely divisible part.
// Same arithmetic, same selection, same lists as fused_knn_kernel (fused_knn.h); the result of a
// search does not depend on who streamed which stage.
#pragma once
#include "fused_knn.h"

namespace scmiac {

// smallest score of a finished subtile against the lane's threshold (see fused_knn.h)
#define SCMIAC_SHARE_TEST_PREV(PRV_)                                                                \
    do {                                                                                            \
        const float m0_ = min3_after_mfma(accs[PRV_][0], accs[PRV_][1], accs[PRV_][2]);             \
        const float m1_ = min3_raw_after(accs[PRV_][3], accs[PRV_][4], accs[PRV_][5], m0_);         \
        const float m2_ = min3_raw_after(accs[PRV_][6], accs[PRV_][7], accs[PRV_][8], m0_);         \
        const float m3_ = min3_raw_after(accs[PRV_][9], accs[PRV_][10], accs[PRV_][11], m0_);       \
        const float m4_ = min3_raw_after(accs[PRV_][12], accs[PRV_][13], accs[PRV_][14], m0_);      \
        const float mn_ = min2_raw(min3_raw(m0_, m1_, m2_), min3_raw(m3_, m4_, accs[PRV_][15]));    \
        if (__builtin_expect(__any(mn_ < tau_hi), 0)) select_slow(accs[PRV_], ref0s[PRV_]);         \
    } while (0)
#define SCMIAC_SHARE_SEED_PREV(PRV_)                                                                \
    do {                                                                                            \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) gmin[PRV_][r] = fminf(gmin[PRV_][r], accs[PRV_][r]); \
    } while (0)

template <int D2, int NW_, int RT_, int PCAP_>
__global__ __launch_bounds__(NW_ * 64, (NW_ >= 8 ? NW_ / 4 : 2))
__attribute__((amdgpu_waves_per_eu((NW_ >= 8 ? NW_ / 4 : 2), (NW_ >= 8 ? NW_ / 4 : 2)))) void fused_knn_share_kernel(
    const float* __restrict__ Qp, const float* __restrict__ Rp, const float* __restrict__ rnorm,
    u64* __restrict__ lists, uint2* __restrict__ pend_g, long long n, long long n_pad, int nqb, long long chunk_rows,
    FusedTail tail, int kt, const int* __restrict__ home, int seed_stages, const float* __restrict__ tau0,
    float* __restrict__ tau_out, ShareArgs sh) {
    using C = FusedCfg<D2, NW_, RT_, PCAP_, true>;
    constexpr int D2P = C::D2P, ROWF = C::ROWF, LS = C::LS, RT = C::RT, NW = C::NW;
    constexpr int PCAP = C::PCAP, NSUB = RT / 32, NB = C::NB, QPB = C::QPB, NT = NW_ * 64;
    constexpr int STAGE_BYTES = RT * LS * 4;
    constexpr int NCHUNK = STAGE_BYTES / 1024;
    constexpr int LD_ITERS = (NCHUNK + NW - 1) / NW;
    constexpr int NORM_CHUNKS = RT / 64;
    constexpr int HCAP = PCAP / 2;
    constexpr int NGRP = D2P / 4;
    constexpr int MAX_HELP = 4;  // streams a block joins after its own, at most
    static_assert(STAGE_BYTES % 1024 == 0, "a stage is a whole number of wave-wide loads");
    static_assert(RT % 64 == 0 && NORM_CHUNKS <= NW, "norms of a stage: one load per wave at most");
    static_assert(NGRP >= 2, "two groups of lookahead");
    (void)ROWF;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tile = reinterpret_cast<float*>(smem);  // [NB][RT*LS]
    float* tnorm = tile + NB * RT * LS;            // [NB][RT]
    const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tile;
    const unsigned tnorm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tnorm;
    __shared__ int sh_word[3];         // [0] base of the next run of stages, [1] helper list claimed, [2] base of an item's first run
    __shared__ unsigned sh_best[NW_];  // per wave: the stream most worth joining

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const float tau_max = __uint_as_float(TAU_MAX_BITS);
    auto widen = [](float t) __attribute__((always_inline)) {
        return t > -__builtin_inff() ? t + fabsf(t) * 3.814697265625e-06f + 1e-30f : t;  // 2^-18 relative
    };

    // Streams: one per grid block of the plan -- (chunk, query block) of the main region, then
    // (query block, piece) of the tail.
    const int nmain = tail.block0;
    const int ntail = (nqb - tail.qb0) * tail.St;
    const int nstreams = nmain + ntail;
    if (tid == 0) atomicAdd(sh.started, 1u);
    // A trace entry per block, after the two counters of sh.stats: {start, end of its own stream,
    // end, stages it scored for others, stages of its own stream it scored itself, first stream it
    // joined + 1, when, streams joined, the first four streams joined + 1, stages scored in each};
    // times in ticks of the constant 100 MHz clock (tools/exp/share_trace.py draws its picture from it).
    u64* const trace = sh.stats != nullptr ? sh.stats + 2 + 16 * (u64)blockIdx.x : nullptr;
    u64 helped_stages = 0, n_helped = 0;
    if (tid == 0 && trace != nullptr) trace[0] = wall_clock64();

    // ---- what this block works on: first its own stream, then up to MAX_HELP streams of others ----
    int v;            // the stream the item at hand draws its stages from
    int hslot = -1;   // helper list of that stream this block fills; -1: the stream is its own
    {
        // Order of the grid: the tail pieces first (the library's choice: what is left to balance at
        // the end is then the finely divisible part), or the main region's whole streams first and the
        // tail pieces as compute units fall free (a block that ends while others still wait for a
        // compute unit goes home instead of helping, see below -- so nobody helps until the last
        // tail piece has a compute unit, which is why this order measured worse).
        int b = (int)blockIdx.x;
        bool tail_block;
        if (sh.tail_first) { tail_block = b < ntail; b = tail_block ? b : b - ntail; }
        else { tail_block = b >= nmain; b = tail_block ? b - nmain : b; }
        if (tail_block) {
            v = nmain + b;
        } else {
            // (the same dealing of consecutive query blocks to an XCD as in fused_knn_kernel)
            if ((b | 255) < nmain) b = (b & ~255) | ((b & 7) << 5) | ((b & 255) >> 3);
            v = b;
        }
    }

    for (int item = 0;; ++item) {
        const bool own = hslot < 0;  // owner of the stream: seed pass, home stages, publishes its thresholds
        int split, qb;
        long long r_begin, rows_here, list_row;
        if (v >= nmain) {
            const int tb = v - nmain;
            split = tb % tail.St; qb = tail.qb0 + tb / tail.St;
            r_begin = (long long)split * tail.chunk_rows;
            rows_here = tail.m_pad - r_begin;
            rows_here = rows_here < 0 ? 0 : (rows_here < tail.chunk_rows ? rows_here : tail.chunk_rows);
            const long long qrow = (long long)qb * QPB + wave * 32;
            list_row = split == 0 ? qrow
                                  : tail.lists_row0 + (long long)(split - 1) * tail.rows + (qrow - (long long)tail.qb0 * QPB);
        } else {
            split = v / tail.qb0; qb = v % tail.qb0;
            r_begin = (long long)split * chunk_rows;
            rows_here = chunk_rows;
            list_row = (long long)split * n_pad + (long long)qb * QPB + wave * 32;
        }
        if (!own) list_row = sh.row0 + ((long long)hslot * nstreams + v) * QPB + wave * 32;
        const long long q0 = (long long)qb * QPB + wave * 32;
        const int T = (int)(rows_here / RT);
        const int G0 = (int)(r_begin / RT);
        int H0 = 0, HN = 0, OW0 = 0, OWN = 0;
        if (home != nullptr && T > 0) { H0 = home[4 * qb]; HN = home[4 * qb + 1]; OW0 = home[4 * qb + 2]; OWN = home[4 * qb + 3]; }
        int O0 = H0 > G0 ? H0 : G0;
        const int O1 = (H0 + HN < G0 + T) ? H0 + HN : G0 + T;
        const int oN = O1 > O0 ? O1 - O0 : 0;
        if (oN == 0) O0 = G0;
        int SN = (home != nullptr && T > 0) ? (OWN < seed_stages ? OWN : seed_stages) : 0;
        { const int cap = T / 4 > 3 ? T / 4 : 3; SN = SN < cap ? SN : cap; }  // (as in fused_knn_kernel)
        const int rest = T - oN;
        const int shift = rest > 0 ? (SN + oN) % rest : 0;

        uint2* pend_w = pend_g + list_row * PCAP;
        uint2* pend_l = pend_w + (j * 2 + hh) * HCAP;
        unsigned pcnt = 0u;
        u64* lists_w = lists + list_row * 64;

        float qf[D2P];
        {
            const float4* qp4 = reinterpret_cast<const float4*>(Qp + (q0 + j) * ROWF + hh * D2P);
#pragma unroll
            for (int c = 0; c < D2P / 4; ++c) {
                float4 x = qp4[c];
                qf[4 * c + 0] = x.x; qf[4 * c + 1] = x.y; qf[4 * c + 2] = x.z; qf[4 * c + 3] = x.w;
            }
        }
        float tau_start = tau_max;
        if (own) {
            if (tau0 != nullptr) tau_start = fminf(tau0[q0 + j], tau_max);
        } else {  // what the owner held after its home stages (written by another block: read past the L1)
            tau_start = __uint_as_float(__hip_atomic_load(
                reinterpret_cast<const unsigned*>(sh.tau_pub) + (long long)v * QPB + wave * 32 + j, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT));
        }
#pragma unroll
        for (int c = 0; c < D2P; ++c) asm volatile("" : "+v"(qf[c]));
        float tau = (q0 + j < n) ? fminf(tau_start, tau_max) : -__builtin_inff();
        float tau_hi = widen(tau);
        unsigned inited = 0u;

        auto flush_full = [&](bool only_full, bool force_uninit) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool need = only_full ? (pcnt >= (unsigned)HCAP) : (pcnt > 0u);
            need = need || (force_uninit && !((inited >> j) & 1u) && (q0 + j < n));
            const u64 m64 = __ballot(need);
            u64 mask = (m64 | (m64 >> 32)) & 0xffffffffull;
            while (mask) {
                int q = __builtin_ctzll(mask);
                mask &= mask - 1;
                const unsigned c0 = __builtin_amdgcn_readlane(pcnt, q);
                const unsigned c1 = __builtin_amdgcn_readlane(pcnt, q + 32);
                float nt = flush_query<HCAP, true>(c0, c1, pend_w + q * PCAP, lists_w + (long long)q * 64,
                                                   (inited >> q) & 1u, lane, kt);
                inited |= 1u << q;
                // (never above the threshold the query had: a list of fewer than kt entries says nothing)
                if (j == q) { tau = fminf(tau, fminf(nt, tau_max)); tau_hi = widen(tau); pcnt = 0u; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };

        // candidate path: see fused_knn_kernel
        auto select_slow = [&](const f32x16& acc, unsigned ref0_wave) __attribute__((always_inline)) {
            const unsigned ref0 = ref0_wave + 4u * (unsigned)hh;
            f32x16 sc = acc;
            asm volatile("" : "+v"(sc));
            float key[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) key[r] = score_key(sc[r], r);
            const float a0 = min3_raw(key[0], key[1], key[2]), a1 = min3_raw(key[3], key[4], key[5]);
            const float a2 = min3_raw(key[6], key[7], key[8]), a3 = min3_raw(key[9], key[10], key[11]);
            const float a4 = min3_raw(key[12], key[13], key[14]);
            const float kmin = min2_raw(min3_raw(a0, a1, a2), min3_raw(a3, a4, key[15]));
            unsigned bit[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bit[r] = (key[r] < tau) ? (1u << r) : 0u;
            const unsigned m0 = bit[0] | bit[1] | bit[2], m1 = bit[3] | bit[4] | bit[5], m2 = bit[6] | bit[7] | bit[8];
            const unsigned m3 = bit[9] | bit[10] | bit[11], m4 = bit[12] | bit[13] | bit[14];
            const unsigned mask = (m0 | m1 | m2) | (m3 | m4 | bit[15]);
            const unsigned cnt = (unsigned)__builtin_popcount(mask);
            if (__builtin_expect(!__any(pcnt + cnt > (unsigned)HCAP), 1)) {
                if (!__any(cnt > 1u)) {
                    const unsigned kb = __float_as_uint(kmin);
                    if (cnt) pend_l[pcnt] = make_uint2(kb, ref0 + (kb & 15u) + (kb & 12u));
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned p = pcnt + (unsigned)__builtin_popcount(mask & ((1u << r) - 1u));
                        if (mask & (1u << r))
                            pend_l[p] = make_uint2(__float_as_uint(key[r]), ref0 + (unsigned)((r & 3) + 8 * (r >> 2)));
                    }
                }
                pcnt += cnt;
            } else {
                unsigned skipped = 0u;
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
                    }
                }
                while (__any(skipped != 0u)) {
                    flush_full(true, false);
                    unsigned still = 0u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool p = ((skipped >> r) & 1u) && key[r] < tau;
                        if (__any(p)) {
                            const bool room = pcnt < (unsigned)HCAP;
                            if (p && room) {
                                pend_l[pcnt] = make_uint2(__float_as_uint(key[r]), ref0 + (unsigned)((r & 3) + 8 * (r >> 2)));
                                ++pcnt;
                            }
                            if (p && !room) still |= 1u << r;
                        }
                    }
                    skipped = still;
                }
            }
            if (__builtin_expect(__any(pcnt >= (unsigned)HCAP), 0)) flush_full(true, false);
        };

        f32x16 accs[2];
        float fr[3][4];
        unsigned ref0s[2] = {0u, 0u};
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[1][r] = tau_max;

        // the first run of stages of a queued stream: asked for now, there when the stage loop starts
        int nbase;
        {
            if (tid == 0) sh_word[2] = (int)atomicAdd(sh.cursor + v, (unsigned)sh.chunk);
            __syncthreads();
            nbase = __builtin_amdgcn_readfirstlane(sh_word[2]);
        }

        // seed pass over the head of the queries' own cells (see fused_knn_kernel)
        if (own && SN > 0) {
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
                SCMIAC_STAGE_SUBTILES(SCMIAC_SHARE_SEED_PREV);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            {
                SCMIAC_SHARE_SEED_PREV(1);
                float all[32];
#pragma unroll
                for (int r = 0; r < 16; ++r) { all[r] = gmin[0][r]; all[16 + r] = gmin[1][r]; }
                sort32_regs(all);
                float mine = all[0];
                const int need = (kt + 1) / 2;
#pragma unroll
                for (int i = 1; i < 32; ++i) mine = (i < need) ? all[i] : mine;
                const float other = __shfl_xor(mine, 32);
                float bound = fmaxf(mine, other);
                bound = bound < tau_max ? widen(bound) : tau_max;
                if (q0 + j < n) { tau = fminf(tau, bound); tau_hi = widen(tau); }
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[1][r] = tau_max;
            }
        }

        // ---- the stages of this item, one ahead of the one being scored ----
        int hi = own ? 0 : oN;            // home stages taken so far (a helper's were streamed by the owner)
        int pos = 0, cend = 0;            // current run of rest positions [pos, cend)
        bool need_req = false, req = false;
        unsigned reply = 0u;
        auto take = [&]() __attribute__((always_inline)) -> long long {  // block-uniform
            if (hi < oN) return (long long)(O0 + hi++) * RT;
            if (pos >= cend) {
                if (cend >= rest) return -1;           // the run that just ended reached the end
                if (nbase >= rest) return -1;          // (the base asked for when the previous run began has arrived: runs are >= 2 stages)
                pos = nbase;
                cend = nbase + sh.chunk < rest ? nbase + sh.chunk : rest;
                need_req = true;
            }
            int x = pos++ + shift;
            if (x >= rest) x -= rest;
            int g = G0 + x;
            if (g >= O0) g += oN;
            return (long long)g * RT;
        };
        auto ask = [&]() __attribute__((always_inline)) {  // the run after the one just begun
            if (need_req) {
                if (tid == 0) reply = atomicAdd(sh.cursor + v, (unsigned)sh.chunk);
                need_req = false; req = true;
            }
        };
        bool want_pub = own, pub_pending = false;
        int done = 0;
        auto publish = [&]() __attribute__((always_inline)) {  // thresholds for whoever joins this stream
            if (hh == 0) sh.tau_pub[(long long)v * QPB + wave * 32 + j] = tau;
            want_pub = false; pub_pending = true;
        };
        if (want_pub && oN == 0) publish();

        long long cur_row = take();
        ask();
        if (cur_row >= 0) SCMIAC_STAGE(cur_row, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (req && tid == 0) sh_word[0] = (int)reply;
        __syncthreads();
        int buf = 0;
        while (cur_row >= 0) {
            if (pub_pending) {  // the published thresholds have reached the L2 (vmcnt(0) + barrier above)
                if (tid == 0) __hip_atomic_store(sh.pub + v, (unsigned)rest + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                pub_pending = false;
            }
            if (req) { nbase = __builtin_amdgcn_readfirstlane(sh_word[0]); req = false; }
            const long long nxt_row = take();
            ask();
            if (nxt_row >= 0) SCMIAC_STAGE(nxt_row, buf ^ 1);
            const float* tb = tile + buf * RT * LS;
            const float* nb = tnorm + buf * RT;
            const long long cur_row0 = cur_row;
            SCMIAC_LOAD_NORMS(0, 0);
            SCMIAC_LOAD_GROUP(0, 0, 0);
            SCMIAC_LOAD_GROUP(0, 1, 1);
            SCMIAC_STAGE_SUBTILES(SCMIAC_SHARE_TEST_PREV);
            ++done;
            if (want_pub && done >= oN) publish();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (req && tid == 0) sh_word[0] = (int)reply;
            __syncthreads();
            cur_row = nxt_row;
            buf ^= 1;
        }
        if (done > 0) {  // the last subtile is still untested
            float mn_ = accs[1][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mn_ = fminf(mn_, accs[1][r]);
            if (__any(mn_ < tau_hi)) select_slow(accs[1], ref0s[1]);
        }
        flush_full(false, true);
        if (hh == 0) tau_out[list_row + j] = tau;
        if (want_pub) publish();  // (a stream without stages: nothing for anyone to join, but the flag is owed)
        if (pub_pending) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(sh.pub + v, (unsigned)rest + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!own && tid == 0 && sh.stats != nullptr) {
            atomicAdd(sh.stats, 1ull);
            atomicAdd(sh.stats + 1, (u64)done);
        }
        if (own) { if (tid == 0 && trace != nullptr) { trace[1] = wall_clock64(); trace[4] = (u64)done; } }
        else {
            if (tid == 0 && trace != nullptr && n_helped < 4) { trace[8 + n_helped] = (u64)v + 1; trace[12 + n_helped] = (u64)done; }
            helped_stages += (u64)done; ++n_helped;
        }

        // ---- the next item: the published stream with the most stages left and a helper list free ----
        // (only once every block of the grid has started: a compute unit that a waiting block could
        // have is worth more to the launch than a helper)
        // (One thread reads the counter and the block agrees on what it read: the counter moves
        // while blocks end, and threads that each read it for themselves could part ways here,
        // some leaving, some waiting at the barriers below.)
        if (item >= MAX_HELP) break;
        if (tid == 0) sh_word[1] = (int)__hip_atomic_load(sh.started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned started_now = (unsigned)__builtin_amdgcn_readfirstlane(sh_word[1]);
        __syncthreads();
        if (started_now < gridDim.x) break;
        // Blocks run out of work in droves, and droves that all went for the one stream with the most
        // stages left would find its lists taken and go home while the next-longest streams stay
        // alone.  So the stages left count in units of four, and among streams of one unit every
        // block has an order of its own (a hash of stream, block and attempt).
        int found = -1, slot = -1;
        for (int attempt = 0; attempt < 6 && found < 0; ++attempt) {
            unsigned best = 0u;  // (stages left / 4, capped) << 20 | hash << 12 | (4095 - stream): the larger the better
            for (int vv = tid; vv < nstreams; vv += NT) {
                const unsigned p = __hip_atomic_load(sh.pub + vv, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (p == 0u) continue;
                const unsigned c = __hip_atomic_load(sh.cursor + vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned cl = __hip_atomic_load(sh.claims + vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const long long left = (long long)(p - 1u) - (long long)c;
                if (left >= (long long)sh.min_left && cl < (unsigned)sh.lists_per) {
                    const unsigned unit = left / 4 > 0xfffll ? 0xfffu : (unsigned)(left / 4);
                    const unsigned hash = (((unsigned)vv * 2654435761u) ^ ((unsigned)blockIdx.x * 40503u + (unsigned)attempt * 977u)) >> 24;
                    const unsigned key = (unit << 20) | (hash << 12) | (unsigned)(4095 - vv);
                    best = key > best ? key : best;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned other = __shfl_xor(best, o);
                best = other > best ? other : best;
            }
            __syncthreads();  // (everyone is done with sh_best / sh_word of the attempt before)
            if (lane == 0) sh_best[wave] = best;
            __syncthreads();
            best = 0u;
#pragma unroll
            for (int w = 0; w < NW_; ++w) best = sh_best[w] > best ? sh_best[w] : best;
            if (best == 0u) break;  // nothing worth joining
            const int cand = 4095 - (int)(best & 4095u);
            if (tid == 0) sh_word[1] = (int)atomicAdd(sh.claims + cand, 1u);
            __syncthreads();
            const int got = __builtin_amdgcn_readfirstlane(sh_word[1]);
            if (got < sh.lists_per) { found = cand; slot = got; }
        }
        if (found < 0) break;
        if (tid == 0 && trace != nullptr && n_helped == 0) { trace[5] = (u64)found + 1; trace[6] = wall_clock64(); }
        v = found; hslot = slot;
    }
    if (tid == 0 && trace != nullptr) { trace[2] = wall_clock64(); trace[3] = helped_stages; trace[7] = n_helped; }
}

#undef SCMIAC_SHARE_TEST_PREV
#undef SCMIAC_SHARE_SEED_PREV

template <int D2, int NW, int RT, int PC>
hipError_t launch_fused_share_t(const float* Qp, const float* Rp, const float* rnorm, u64* lists, uint2* pend_g, long long n,
                                long long n_pad, int nqb, long long chunk_rows, const FusedTail& tail, int kt,
                                const int* home, int seed_stages, const float* tau0, float* tau_out, const ShareArgs& sh,
                                hipStream_t stream) {
    using C = FusedCfg<D2, NW, RT, PC, true>;
    size_t lds = C::lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_knn_share_kernel<D2, NW, RT, PC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(tail.block0 + (nqb - tail.qb0) * tail.St)), block(NW * 64);
    hipLaunchKernelGGL((fused_knn_share_kernel<D2, NW, RT, PC>), grid, block, lds, stream, Qp, Rp, rnorm, lists, pend_g, n,
                       n_pad, nqb, chunk_rows, tail, kt, home, seed_stages, tau0, tau_out, sh);
    return hipGetLastError();
}

// the unit of the stream-sharing kernels (kind 3)
#define SCMIAC_DEFINE_SHARE_TU(N_, VARIANTS_)                                                                          \
    hipError_t launch_fused_tu##N_(const KnnGeometry& g, const FusedLaunch& a) {                                       \
        if (a.kind != 3) return hipErrorNotFound;                                                                      \
        VARIANTS_(SCMIAC_TU_SHARE_CASE)                                                                                \
        return hipErrorNotFound;                                                                                       \
    }                                                                                                                  \
    bool fused_share_compiled_tu##N_(const KnnGeometry& g) {                                                           \
        VARIANTS_(SCMIAC_TU_HAS_CASE)                                                                                  \
        return false;                                                                                                  \
    }
#define SCMIAC_TU_SHARE_CASE(V_, NW_, RT_, PC_, GP_)                                                                   \
    if (g.D2 == V_ && g.nw == NW_ && g.RT == RT_ && g.pcap == PC_ && g.gp == GP_ && GP_ == 1)                          \
        return launch_fused_share_t<V_, NW_, RT_, PC_>(a.Qp, a.Rp, a.rnorm, a.lists, static_cast<uint2*>(a.pend_g), a.n, \
                                                       g.n_pad, g.nqb, g.chunk_rows, g.tail, a.kt, a.home, g.seed_stages, \
                                                       a.tau0, a.tau_out, a.share, a.stream);

}  // namespace scmiac
