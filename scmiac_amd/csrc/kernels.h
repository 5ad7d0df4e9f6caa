// Internal declarations shared by the kernel translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define SCMIAC_KNN_MAX_K_INTERNAL 64
#define SCMIAC_KNN_MAX_D_INTERNAL 256

namespace scmiac {

// Shapes derived from (n, m, d, splits): padded row counts, packed row length and
// how the reference rows are cut into independently searched chunks.
// The query blocks of the grid's last, partly filled round ("tail") search the
// references in St chunks each; everything before them is one block per (chunk, query block).
struct FusedTail {
    int qb0;               // first query block of the tail (== nqb when there is none)
    int St;                // reference chunks per tail query block (1: no tail)
    int block0;            // first grid block of the tail
    int qpb;               // queries per block
    long long chunk_rows;  // rows per tail chunk (multiple of RT)
    long long m_pad;       // padded reference rows
    long long rows;        // query rows in the tail
    long long lists_row0;  // first shortlist row of the tail's chunks 1..St-1
};

struct KnnGeometry {
    int D2;                // MFMA K steps per tile (two features per step)
    int D2P;               // D2 rounded up to a multiple of 4
    int rowf;              // feature floats per packed row (2 * D2P)
    int ls;                // row stride of the packed reference image (features, zero pad)
    int RT;                // reference rows per LDS stage
    int nw;                // waves per block of the fused kernel
    int pcap;              // pending candidates per query (buffer length)
    int gp;                // 1: those buffers live in global memory (workspace), 0: in LDS
    int qpb;               // queries per block
    int nqb;               // query blocks
    int S;                 // reference chunks searched independently
    long long n_pad;       // queries padded to a multiple of qpb
    long long chunk_rows;  // rows per chunk (multiple of RT)
    long long m_pad;       // S * chunk_rows
    FusedTail tail;
    long long list_rows;   // 64-entry shortlists in the workspace
    int blocks;            // grid size of the fused kernel
    int cells;             // pivots of the locality partition (0: rows keep the caller's order)
    int cells_pad;         // cells rounded up to a multiple of 8
    int home_max;          // most home stages a query block streams ahead of its chunk
    int seed_stages;       // home stages the fused kernel's seed pass looks at
    int prune;             // 1: pruned search (SCMIAC_KNN_PRUNED): blocks skip the cells out of their queries' reach
    // Stream sharing (dense search whose main region is a single round of the grid): blocks that
    // have run out of work join the streams of blocks that have not.  share = helper shortlists per
    // stream (0: off; a stream = a block of the plan's grid: main region, then tail); their rows
    // follow the ordinary ones from share_row0 on.
    int share;
    int share_chunk;       // stages handed out at a time
    int share_min_left;    // stages a stream must have left to be worth joining
    int share_tail_first;  // 1: the tail pieces are the first blocks of the grid
    long long share_row0;  // first helper shortlist: row share_row0 + (slot * blocks + stream) * qpb + query
};

typedef unsigned short cell_t;  // number of a pivot cell (at most KNN_MAX_CELLS of them)
static constexpr int KNN_MAX_CELLS = 2048;

// device scratch of the locality partition (carved out of the search workspace)
struct LocalityWs {
    float* pivT;            // [d][cells_pad]
    float* pnorm;           // [cells_pad]
    cell_t* cell_q;         // [n]
    cell_t* cell_r;         // [m]
    int* hist_q;            // [ceil(n / 256)][cells]
    int* hist_r;            // [ceil(m / 256)][cells]
    int* cstart_q;          // [2 (cells + 1)]: first position of every cell, then scratch for the cell sizes
    int* cstart_r;          // [2 (cells + 1)]
    int* perm_q;            // [n]: packed query position -> caller's row
    int* perm_r;            // [m]: packed reference row -> caller's row
    int* home;              // [nqb][4]: home window (first stage, stages), own cells inside it (first stage, stages)
    int* piv_rows;          // [4 cells_pad]: rows of the pivots as chosen (pruned search); scratch of the chain order
    float* pivT_tmp;        // [d][cells_pad], [cells_pad]: the pivots as chosen, before the chain order
    float* pnorm_tmp;
    _Float16* pivD;         // [256][256]: mutual distances of the chain's lead pivots
    float* pivRM;           // [cells_pad][d]: the pivots in chain order, one per row (A operand of the MFMA assignment)
    // pruned search only (NULL otherwise)
    float* qpd;             // [cells_pad][n]: lower bound of dist(query row, pivot)
    unsigned* crad;         // [cells_pad]: float bits, upper bound of max dist(reference row, its pivot)
    float* fps_sample;      // [d][knn_fps_sample_rows()]: the farthest-point traversal's sample, feature-major
    unsigned short* order;  // [grid blocks][order_stride]: the stages a block decided to stream
    int order_stride;       // entries per block in `order`: the longest chunk's stages + 1
    unsigned long long* stats;  // [2]: stages streamed by all blocks, stages a dense search would have streamed
};

// Diagnostic / tuning knobs of the fused kernel's grid plan (all zero = product behaviour).
struct FusedTuning {
    int slots;          // > 0: pretend the GPU runs this many blocks at once
    int tail_min_rows;  // > 0: shortest reference chunk the planner may choose
    int no_tail;        // 1: never split the last round's query blocks further
    int ablate;         // ablation bits of the diagnostic kernel instantiation (1, 2, 4, 8, 16, 32)
    int diag;           // 1: per-wave cycle counters into the workspace
    int no_locality;    // 1: stream the references in the caller's order (no pivot partition)
    int home_rows;      // > 0: most home rows a query block streams first (default 8192)
    int seed_rows;      // > 0: home rows the seed pass looks at (default 6144)
    int cell_rows;      // > 0: reference rows per pivot cell aimed at (default 1024)
    int has_variant;    // 1: use (nw, rt, pcap, gp) where that combination is compiled
    int nw, rt, pcap, gp;
    int share;          // stream sharing: 0 the library's choice, -1 never, > 0 this many helper lists per stream wherever the plan allows
    int share_chunk;    // > 0: stages handed out at a time (at least 2; default 4)
    int share_min_left; // > 0: stages a stream must have left to be joined (default 8)
    int share_order;    // 0 the library's choice, 1 tail pieces first in the grid, 2 tail pieces last
};
// One snapshot per C-ABI call: everything that call plans, carves and launches derives from it, so
// a concurrent scmiac_knn_set_tuning cannot pair a geometry planned under one setting with a
// kernel chosen under another.
FusedTuning knn_get_tuning();
bool knn_set_tuning(const FusedTuning& t);  // false: rejected (uncompiled variant, negative value)

int knn_pick_d2(int d);
// rank of the shortlist entry whose key serves as the running threshold: the k wanted
// neighbours plus a margin that the rounding-error certificate can absorb
// (the certificate's error bound grows with the number of features while the gaps between
// consecutive neighbour distances shrink: beyond 64 features the margin is 24 ranks instead of
// 6, which turns thousands of uncertified rows per 200 000 into a handful)
inline int knn_threshold_rank(int k, int d) {
    const int margin = d > 64 ? 24 : 6;
    return (k + margin < 64) ? k + margin : 64;
}
// mode: 0 dense search, 1 pruned search (honoured where the pruned kernel is compiled and the
// locality partition applies; the dense search otherwise)
KnnGeometry knn_geometry(long long n, long long m, int d, int k, int splits, int mode, const FusedTuning& tn);
int knn_fps_sample_rows();       // sample rows of the pruned search's farthest-point traversal
int knn_cell_block_rows(int d);  // rows per block of the partition kernels (their per-block histograms' granularity)
bool fused_pruned_compiled(const KnnGeometry& g);  // is there a pruned instantiation of this geometry's variant?

// Pruned search: what the fused kernel needs to decide, per block, which cells of the reference
// image can hold a neighbour of its queries.
struct PruneArgs {
    const float* qpd;          // NULL: no pruning
    long long n;               // row stride of qpd
    const unsigned* crad;
    const int* cstart_r;       // [cells + 1] first packed row of every cell
    const int* perm_q;         // packed query position -> caller's row
    const cell_t* cell_q;      // cell of every query (caller's rows)
    const unsigned* maxnorm;   // max |r|^2 (float bits)
    float err_coef;            // |score + |q|^2 - T| <= err_coef (|r|^2 + |q|^2)
    int cells;
    unsigned short* order;     // [grid blocks][order_stride]
    int order_stride;
    long long m;               // reference rows (stages past them hold padding only)
    unsigned long long* stats; // [2]: stages streamed, stages of the dense stream (atomic adds)
};

// Stream sharing: the queues of the streams and what their helpers need.
struct ShareArgs {
    unsigned* started;          // [1] blocks of the grid that have started
    unsigned* cursor;           // [streams] next position of the stream's rest to hand out (stages); NULL: no sharing
    unsigned* pub;              // [streams] 0: the owner has not published its thresholds yet; else stages of the rest + 1
    unsigned* claims;           // [streams] helper lists claimed (may overshoot lists_per)
    float* tau_pub;             // [streams][qpb] the owner's thresholds after its home stages
    long long row0;             // first helper shortlist row
    int lists_per;              // helper lists per stream
    int chunk;                  // stages handed out at a time (>= 2)
    int min_left;               // stages a stream must have left to be worth joining
    int tail_first;             // 1: the tail pieces are the first blocks of the grid, 0: the last
    unsigned long long* stats;  // [2]: streams joined, stages scored by helpers (atomic adds)
};

// arguments of one launch of the fused kernel that are not part of the geometry
struct FusedLaunch {
    const float* Qp;
    const float* Rp;
    const float* rnorm;
    unsigned long long* lists;
    void* pend_g;
    long long n;
    int kt;
    int ablate;                // ablation bits (diagnostic instantiation only)
    int kind;                  // 0 product dense kernel, 1 diagnostic instantiation, 2 product pruned kernel, 3 stream sharing
    const int* home;
    const float* tau0;
    float* tau_out;
    PruneArgs prune;
    ShareArgs share;
    unsigned long long* diag;
    hipStream_t stream;
};
// one per translation unit of fused-kernel variants: hipErrorNotFound / 0 when the geometry's
// variant is not among that unit's
hipError_t launch_fused_tu0(const KnnGeometry& g, const FusedLaunch& a);
hipError_t launch_fused_tu1(const KnnGeometry& g, const FusedLaunch& a);
hipError_t launch_fused_tu2(const KnnGeometry& g, const FusedLaunch& a);
hipError_t launch_fused_tu3(const KnnGeometry& g, const FusedLaunch& a);
int fused_occupancy_tu0(const KnnGeometry& g);
int fused_occupancy_tu1(const KnnGeometry& g);
int fused_occupancy_tu2(const KnnGeometry& g);
int fused_occupancy_tu3(const KnnGeometry& g);
// pruned product instantiations of the library's own variants: units of their own (4: up to 64
// features, 6: wider rows)
hipError_t launch_fused_tu4(const KnnGeometry& g, const FusedLaunch& a);
bool fused_pruned_compiled_tu4(const KnnGeometry& g);
hipError_t launch_fused_tu6(const KnnGeometry& g, const FusedLaunch& a);
bool fused_pruned_compiled_tu6(const KnnGeometry& g);
// stream-sharing instantiations (the same variants): a unit of their own
hipError_t launch_fused_tu5(const KnnGeometry& g, const FusedLaunch& a);
bool fused_share_compiled_tu5(const KnnGeometry& g);

hipError_t launch_pack(const float* X, long long n, int d, const KnnGeometry& g, float scale, float* Xp,
                       int ld, float* norms, long long n_pad, unsigned* maxnorm_bits, const int* perm,
                       hipStream_t stream);
hipError_t launch_locality(const float* Q, long long n, const float* R, long long m, int d, const KnnGeometry& g,
                           const LocalityWs& w, hipStream_t stream);
hipError_t launch_seed_tau(const float* Q, const int* perm_q, int d, long long n, const KnnGeometry& g, const int* idx,
                           const float* d2, int k, const unsigned* maxnorm_bits, float* tau0, hipStream_t stream);
// the pruned search's kernel arguments (qpd == NULL unless the geometry asks for it)
PruneArgs knn_prune_args(const KnnGeometry& g, const LocalityWs& loc, const unsigned* maxnorm, long long n, long long m,
                         int d);
hipError_t launch_fused(const KnnGeometry& g, const FusedTuning& tn, const float* Qp, const float* Rp, const float* rnorm,
                        unsigned long long* lists, void* pend_g, long long n, int kt, const int* home,
                        const float* tau0, float* tau_out, const PruneArgs& prune, const ShareArgs& share,
                        unsigned long long* diag, hipStream_t stream);
// (share_claims: helper lists claimed per stream, NULL without stream sharing)
hipError_t launch_refine(const float* Q, const float* Qp, const float* Rp, int d, long long n, const KnnGeometry& g,
                         const unsigned long long* lists, int k, int kt, long long col_offset, int merge,
                         const unsigned* maxnorm_bits, const float* tau_lists, const int* perm_q, const int* perm_r,
                         const unsigned* share_claims, int* idx, float* d2, float* dist, double* lb, int* fail_rows,
                         int* n_fail, hipStream_t stream);
hipError_t launch_exact_rows(const float* Q, const float* R, int d, long long n, long long m, int k,
                             long long col_offset, const int* rows, const int* n_rows, int* idx,
                             float* d2, float* dist, void* scratch, size_t scratch_bytes,
                             hipStream_t stream);

// graph kernels (mutual nearest neighbours, shared-neighbour score, label filter)
size_t compact_workspace_bytes(long long n_items);
hipError_t launch_find_mnn(const int* g12, long long n1, long long ld12, const int* g21, long long n2,
                           long long ld21, int ka, long long* pairs, long long* n_pairs, int* status,
                           void* ws, hipStream_t stream);
hipError_t launch_snn_score(const long long* pairs, const long long* n_pairs, long long n_pairs_max,
                            const int* g11, const int* g12, const int* g21, const int* g22,
                            long long n1, long long n2, long long ld, int ks, int* score,
                            long long* hist, hipStream_t stream);
hipError_t launch_anchor_filter(const long long* pairs, const long long* n_pairs, long long n_pairs_max,
                                const int* raw, double p_lo, double p_hi, double thr, const int* codes1,
                                long long n1, const int* codes2, long long n2, int ct_filter,
                                long long* out_pairs, double* out_score, unsigned char* out_same,
                                long long* out_index, long long* counts, double* all_score, void* ws,
                                hipStream_t stream);

hipError_t launch_lu_row_perm(const int* piv, int m, long long n, long long* perm, int* status, hipStream_t stream);

// count_kernels.hip: references closer than the true match, per query row
size_t closer_count_pend_bytes(const KnnGeometry& g);
hipError_t launch_closer_count(const KnnGeometry& g, const float* Q, const float* R, const int* match, int d,
                               long long n, long long m, const float* Qp, const float* Rp, const float* rnorm,
                               const unsigned* maxnorm_bits, float* lo, float* hi, float* dmatch, unsigned* counts,
                               unsigned* pend, int* status, const int* perm_q, hipStream_t stream);
hipError_t launch_scatter_counts(const unsigned* packed, const int* perm_q, long long n, unsigned* out,
                                 hipStream_t stream);

// contrastive_kernels.hip
int ntxent2_blocks(int B);
size_t ntxent2_workspace_bytes(int B, int d);
hipError_t launch_ntxent2_forward(const float* zi, const float* zj, int B, int d, float temperature, float* lse_row,
                                  float* lse_col, float* loss_parts, float* ws, hipStream_t stream);
hipError_t launch_ntxent2_backward(const float* zi, const float* zj, int B, int d, float temperature,
                                   const float* lse_row, const float* lse_col, const float* grad_out, float* dzi,
                                   float* dzj, float* ws, hipStream_t stream);

}  // namespace scmiac
