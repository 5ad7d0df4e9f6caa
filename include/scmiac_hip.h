/*
 * libscmiac_hip.so -- C ABI of the MI355X (gfx950) anchor-search kernels.
 *
 * Every entry point returns 0 on success and a non-zero status otherwise;
 * scmiac_last_error() then holds a human-readable reason (thread-local).  All
 * array arguments are DEVICE pointers owned by the caller; nothing is allocated,
 * freed or synchronised inside (work is enqueued on `stream`, a hipStream_t
 * passed as void*; NULL is the default stream).  No C++ types cross the boundary.
 *
 * The reference (TianLab-Bioinfo/scMIAC) is pure Python and has no FFI of its
 * own; each entry point names the reference function whose arithmetic it
 * replaces (paths relative to the reference checkout).
 */
#ifndef SCMIAC_HIP_H
#define SCMIAC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SCMIAC_API __attribute__((visibility("default")))
#else
#define SCMIAC_API
#endif

#define SCMIAC_OK 0
#define SCMIAC_ERR_ARG 1       /* bad argument (shape, NULL pointer, unsupported size) */
#define SCMIAC_ERR_WORKSPACE 2 /* workspace too small */
#define SCMIAC_ERR_HIP 3       /* a HIP runtime call failed */

#define SCMIAC_KNN_MAX_K 64   /* neighbours per row one search can return */
#define SCMIAC_KNN_MAX_D 256  /* feature dimension supported by the fused kernel */

/* Library/ABI version (major*1000 + minor). */
SCMIAC_API int scmiac_version(void);

/* Message for the last non-zero status returned on this thread. */
SCMIAC_API const char *scmiac_last_error(void);

/*
 * How scmiac_knn_search goes through the reference rows (its `mode` argument).  Both give the
 * same, exact table.
 *   SCMIAC_KNN_DENSE   every query block streams every reference row: the dense contraction.
 *   SCMIAC_KNN_PRUNED  both sets are grouped into cells round pivots taken from the reference rows
 *                      (farthest-point traversal of a sample), and a query block streams only the
 *                      cells that can hold a neighbour of one of its queries: those whose pivot
 *                      distance minus cell radius lies within the queries' reach (triangle
 *                      inequality, with the rounding-error slack of the search's own certificate).
 *                      Many times faster on clustered data (latent embeddings of cell atlases);
 *                      on data without structure every cell stays within reach and the search
 *                      costs what the dense one costs plus the choice of pivots.  Available from
 *                      16 384 reference / 2 048 query rows on (any width the search takes); other
 *                      shapes run dense.  No counterpart in the reference (its index, pynndescent, prunes
 *                      approximately; this is exact).
 */
#define SCMIAC_KNN_DENSE 0
#define SCMIAC_KNN_PRUNED 1

/*
 * Bytes of device scratch scmiac_knn_search needs for an (n x d) query block
 * against an (m x d) reference block.  `splits` = 0 lets the library choose how
 * many reference chunks to search independently (more chunks fill the GPU when n
 * is small); pass the same `splits` and `mode` to scmiac_knn_search.
 * (Planning reads the tuning knobs once per call and asks the HIP runtime, once per device and
 * kernel variant, how many blocks the current device runs at once -- the first planning call of a
 * process with a GPU therefore initialises the runtime; without a device it assumes an MI355X.)
 */
SCMIAC_API int scmiac_knn_workspace_size(int64_t n, int64_t m, int d, int k, int splits, int mode, size_t *bytes);

/*
 * Exact k nearest reference rows for every query row -- the arithmetic behind
 *   scmiac/preprocessing/anchors.py:546-565
 *     pynndescent.NNDescent(ref, metric="euclidean", n_neighbors=k+1).neighbor_graph
 *     index.query(q, k=k)
 * (third-party, approximate there; exact here).  Squared distance
 *   D(a,b) = fmaf chain over (a_i-b_i)^2 in float32, feature index ascending,
 * rows ordered by (D, reference row number) ascending.
 *
 *   q      [n*d]  float32 row-major queries            (device)
 *   r      [m*d]  float32 row-major references         (device)
 *   col_offset    added to every reported reference row number (reference block
 *                 m0..m0+m of a larger set; 0 otherwise)
 *   merge         0: idx/dist2/lower_bound are outputs only.
 *                 1: they also hold the result of earlier calls over OTHER
 *                    reference blocks of the same set; the result is the exact
 *                    top-k of the union (ring-passed / tiled reference sets).
 *   idx    [n*k]  int32   neighbour row numbers (+col_offset); -1 past the end
 *   dist2  [n*k]  float32 D, ascending; +inf past the end
 *   dist   [n*k]  float32 sqrtf(D), or NULL
 *   lower_bound [n] float64: every reference row seen so far that is NOT in
 *                 idx has D >= lower_bound (proved from the fused kernel's
 *                 rounding-error bound); rows whose k-th D is below it are exact.
 *   uncertified_rows [n] int32, n_uncertified [1] int32: rows for which that
 *                 proof failed (distance ties / near-ties at the shortlist edge).
 *   exact_fallback 1: those rows are re-searched against `r` by direct
 *                 evaluation before returning, so the result is exact for every
 *                 row (use 0 only when `r` is one block of several; then call
 *                 scmiac_knn_exact_rows for each block afterwards).
 *   mode          SCMIAC_KNN_DENSE or SCMIAC_KNN_PRUNED (above); the workspace must have been
 *                 sized for the same mode.  One call reads the tuning knobs once: its plan, its
 *                 carve-up of the workspace and its choice of kernel always agree, and a workspace
 *                 sized under other knobs is refused (SCMIAC_ERR_WORKSPACE) when too small.
 *   1 <= k <= SCMIAC_KNN_MAX_K, 1 <= d <= SCMIAC_KNN_MAX_D, n, m < 2^31.
 *   Inputs must be finite with squared row norms below 1e37 (latent embeddings are
 *   many orders of magnitude inside that).  Rows containing NaN or infinities do not
 *   fault or hang the search, but their results are unspecified.
 */
SCMIAC_API int scmiac_knn_search(const float *q, int64_t n, const float *r, int64_t m, int d, int k,
                      int64_t col_offset, int merge, int splits, int mode, int32_t *idx, float *dist2,
                      float *dist, double *lower_bound, int32_t *uncertified_rows,
                      int32_t *n_uncertified, int exact_fallback, void *workspace,
                      size_t workspace_bytes, void *stream);

/*
 * Direct-evaluation search for a list of query rows: merges the exact top-k of
 * reference block `r` into idx/dist2 (which must already hold k valid-or-padded
 * entries per row, e.g. from scmiac_knn_search).  `rows` [*n_rows] int32 and
 * n_rows [1] int32 live on the device.  Same ordering rule as scmiac_knn_search;
 * same reference call sites.  `workspace` is optional device scratch (NULL: none):
 * with 128 KiB per listed row (up to 4096 rows) each row's scan of `r` is spread
 * over the whole GPU instead of one compute unit; the result is the same.
 */
SCMIAC_API int scmiac_knn_exact_rows(const float *q, int64_t n, const float *r, int64_t m, int d, int k,
                          int64_t col_offset, const int32_t *rows, const int32_t *n_rows,
                          int32_t *idx, float *dist2, float *dist, void *workspace,
                          size_t workspace_bytes, void *stream);

/*
 * Mutual-nearest-neighbour anchors -- scmiac/preprocessing/anchors.py:569-572
 *   [[i, G12[i,j]] for i in range(n1) for j in range(ka) if i in G21[G12[i,j], :ka]]
 *   g12 [n1*ld12], g21 [n2*ld21] int32 row-major neighbour tables
 *   pairs [n1*ka*2] int64 out, rows in (i, j) order; n_pairs [1] int64 out
 *   status [1] int32 out: 0, or 1 when an entry of g12 is not a row of g21
 *   workspace: scmiac_compact_workspace_size(n1*ka) bytes
 */
SCMIAC_API int scmiac_find_mnn(const int32_t *g12, int64_t n1, int64_t ld12, const int32_t *g21, int64_t n2,
                    int64_t ld21, int ka, int64_t *pairs, int64_t *n_pairs, int32_t *status,
                    void *workspace, size_t workspace_bytes, void *stream);

/*
 * scmiac_find_mnn for callers that cannot allocate device memory or carry a workspace (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  Same arguments and result as scmiac_find_mnn minus the
 * workspace: the compaction scratch is a static device array inside the library
 * (about 16.1 MiB on every GPU the library is used on, held until the process ends), which
 * bounds the problem: n1 * ka <= SCMIAC_MNN_NOALLOC_MAX_ITEMS, else SCMIAC_ERR_ARG.
 * Calls no hipMalloc / hipFree and no allocating routine; work is enqueued on `stream`.
 * The first call on each device looks the array's address up (make that call outside graph
 * capture).  All calls on one GPU share the array: calls that overlap in time on the same
 * GPU (different streams or threads) race; serialise them.
 */
#define SCMIAC_MNN_NOALLOC_MAX_ITEMS (8 * 1024 * 1024)
SCMIAC_API int scmiac_find_mnn_noalloc(const int32_t *g12, int64_t n1, int64_t ld12, const int32_t *g21,
                                       int64_t n2, int64_t ld21, int ka, int64_t *pairs, int64_t *n_pairs,
                                       int32_t *status, void *stream);

/* Scratch bytes for the ordered-compaction passes over `n_items` flags. */
SCMIAC_API int scmiac_compact_workspace_size(int64_t n_items, size_t *bytes);

/*
 * Shared-neighbour anchor score before normalisation --
 * scmiac/preprocessing/anchors.py:660-664
 *   len(set(G11[x,:ks]) & set(G21[y,:ks])) + len(set(G12[x,:ks]) & set(G22[y,:ks]))
 *   pairs [n_pairs*2] int64, *n_pairs read on the device (n_pairs_max bounds the launch)
 *   g11,g12 [n1*ld], g21,g22 [n2*ld] int32;  score [n_pairs] int32 out
 *   hist [2*ks+1] int64 out (may be NULL): histogram of the scores
 */
SCMIAC_API int scmiac_snn_score(const int64_t *pairs, const int64_t *n_pairs, int64_t n_pairs_max,
                     const int32_t *g11, const int32_t *g12, const int32_t *g21,
                     const int32_t *g22, int64_t n1, int64_t n2, int64_t ld, int ks,
                     int32_t *score, int64_t *hist, void *stream);

/*
 * Score normalisation tail and cell-type filter --
 * scmiac/preprocessing/anchors.py:578-580 (given the two percentiles) and
 * scmiac/modeling/scmiac.py:91-99
 *   score = clip((raw - p_lo) / (p_hi - p_lo), 0, 1)           (float64)
 *   is_same = codes1[x1] == codes2[x2];  keep score > threshold;  then is_same
 * Outputs are compacted in input order:
 *   out_pairs [n*2] int64, out_score [n] float64, out_is_same [n] uint8,
 *   out_index [n] int64 (row label after the score filter's reset_index),
 *   counts [2] int64: {rows with score > threshold, rows written}
 *   all_score [n] float64 out (may be NULL): normalised score of every input row
 *   ct_filter 0 keeps rows whatever is_same says (they are still reported).
 *   workspace: scmiac_compact_workspace_size(n_pairs_max) bytes
 */
SCMIAC_API int scmiac_anchor_filter(const int64_t *pairs, const int64_t *n_pairs, int64_t n_pairs_max,
                         const int32_t *raw_score, double p_lo, double p_hi, double threshold,
                         const int32_t *codes1, int64_t n1, const int32_t *codes2, int64_t n2,
                         int ct_filter, int64_t *out_pairs, double *out_score,
                         uint8_t *out_is_same, int64_t *out_index, int64_t *counts,
                         double *all_score, void *workspace, size_t workspace_bytes,
                         void *stream);

/*
 * Diagnostic and tuning knobs of the fused kernel (no counterpart in the reference; the
 * library reads no environment variables).  All zero -- the state at load time -- is the
 * product behaviour; tests and tuning tools set, and reset, them explicitly:
 *   slots          > 0: plan the grid as if the GPU ran this many blocks at once
 *   tail_min_rows  > 0: shortest reference chunk the planner may choose (default 8192)
 *   no_tail        1: do not split the last round's query blocks further
 *   ablate         ablation bits honoured by the diagnostic kernel instantiation only
 *                  (1 no selection, 2 no staging, 4 no barrier, 8 empty candidate path,
 *                  16 candidate arithmetic only, 32 no seed pass): results are then NOT a k-NN
 *                  table (bit 32 excepted)
 *   diag           1: per-wave counters, see scmiac_knn_diag_region
 *   nw, rt, pcap, gp  kernel variant (waves per block, rows per LDS stage, pending
 *                  candidates per query, 1 = pending buffers in global memory); all zero
 *                  = the library's choice.  A combination that is not compiled for any
 *                  feature size is rejected with SCMIAC_ERR_ARG; one that is compiled for
 *                  other feature sizes only is ignored for the rest.
 *   no_locality    1: stream the reference rows in the caller's order instead of regrouping
 *                  both sets by nearest pivot first (same result, A/B timing aid)
 *   home_rows, seed_rows, cell_rows  > 0: override how many of its queries' home rows a
 *                  block streams first (8192), how many of them the seed pass looks at
 *                  (6144) and the reference rows per pivot cell aimed at (1024; the library's
 *                  own choice stops at 1024 cells -- 248 for a pruned search --, an explicit value
 *                  may ask for up to 2048)
 * Thread-safe; every later call of the library reads the knobs once, at its start.
 */
SCMIAC_API int scmiac_knn_set_tuning(int slots, int tail_min_rows, int no_tail, int ablate, int diag, int nw,
                                     int rt, int pcap, int gp, int no_locality, int home_rows, int seed_rows,
                                     int cell_rows);

/*
 * Diagnostic aid (no counterpart in the reference).  When the `diag` knob of
 * scmiac_knn_set_tuning is set, the fused kernel of scmiac_knn_search writes twelve
 * uint64 counters per wave into the workspace at `offset`:
 *   {wave cycles, cycles in the candidate path, cycles merging shortlists,
 *    candidate-path entries, shortlist merges, candidates taken, register steps,
 *    cycles up to the duplicate check, cycles of the one-candidate-per-lane store,
 *    entries that took the general path, 0, 0}.
 */
SCMIAC_API int scmiac_knn_diag_region(int64_t n, int64_t m, int d, int k, int splits, int mode, size_t *offset,
                                      size_t *bytes, int *waves);

/*
 * Where a pruned search (SCMIAC_KNN_PRUNED) of this shape leaves its two uint64 counters in the
 * workspace: {stages of reference rows its blocks streamed, stages the dense search would have
 * streamed}; their ratio is the share of the dense contraction that was still performed.
 * SCMIAC_ERR_ARG when a search of this shape does not prune.  No counterpart in the reference.
 */
SCMIAC_API int scmiac_knn_prune_stats_offset(int64_t n, int64_t m, int d, int k, int splits, size_t *offset);

/*
 * Stream sharing (no counterpart in the reference).  When the main region of a dense search's grid
 * is a single round -- every block owns one (query block, reference chunk) stream from the first
 * stage to the last, and the launch would end with the slowest of them -- the stages of a stream
 * are handed out in runs through a counter, to its owner and to blocks that have run out of work
 * of their own; those fill shortlists of their own, which the refine pass merges like a chunk's.
 * The result does not depend on who scored which stage.
 *   scmiac_knn_set_sharing(lists, run, min_left, order): lists 0 = the library's choice (two helper lists
 *     per stream where the plan is a single round), -1 = never, 1..8 = this many wherever the plan
 *     has more than one block (test aid); run = stages handed out at a time (0: 4; at least 2);
 *     min_left = stages a stream must have left to be joined (0: 8); order = 0 the library's choice, 1 the
 *     tail pieces first in the grid, 2 last (A/B aid).  Part of the tuning state
 *     (read once per call; scmiac_knn_set_tuning leaves these three as they are).
 *   scmiac_knn_share_info: info[4] (host) = {helper lists per stream (0: a search of this shape
 *     does not share), stages per run, min_left, streams}; *stats_offset = where such a search
 *     leaves its counters in its workspace: two uint64 {streams joined, stages scored by helpers},
 *     then sixteen uint64 per block of the grid {start, end of its own stream, end of the block,
 *     stages scored for others, stages of its own stream it scored itself, first stream joined + 1,
 *     when, streams joined, the first four streams joined + 1, stages scored in each}; times in ticks
 *     of the constant 100 MHz clock.
 */
SCMIAC_API int scmiac_knn_set_sharing(int lists, int run, int min_left, int order);
SCMIAC_API int scmiac_knn_share_info(int64_t n, int64_t m, int d, int k, int splits, int64_t *info,
                                     size_t *stats_offset);

/*
 * How scmiac_knn_search would lay its fused kernel's grid out for this problem (no
 * counterpart in the reference; no GPU work).  plan[10] (host memory):
 *   {query blocks, queries per block, S, rows per chunk, main query blocks,
 *    St, rows per tail chunk, grid blocks, pivot cells (0: no partition), 1 if the search prunes}
 * Query blocks below "main query blocks" are searched in S reference chunks each; the
 * rest -- what would be a partly filled last round of the grid -- in St chunks each.
 */
SCMIAC_API int scmiac_knn_grid_plan(int64_t n, int64_t m, int d, int k, int splits, int mode, int64_t *plan);

/*
 * Name of the fused kernel instantiation scmiac_knn_search would launch for this problem,
 * as a profiler prints it without the namespace, e.g.
 * "fused_knn_kernel<25, 12, 320, 64, true, 0>" (the last parameter: 0 dense product kernel,
 * 2 pruned product kernel, 1 diagnostic instantiation; no counterpart in the reference; no GPU
 * work).  Lets a measurement say which compiled variant it timed.
 */
SCMIAC_API int scmiac_knn_kernel_name(int64_t n, int64_t m, int d, int k, int splits, int mode, char *buf,
                                      size_t buf_bytes);

/*
 * Optional in-library timing of the fused distance + top-k kernel (no counterpart
 * in the reference).  While enabled, every scmiac_knn_search brackets that kernel's
 * launch with a pair of HIP events on `stream` (up to 256 searches; enabling again
 * restarts the count).  scmiac_knn_fused_ms waits for the recorded events and
 * returns the summed kernel time and the number of launches it covers.
 */
SCMIAC_API int scmiac_knn_set_timing(int enabled);
SCMIAC_API int scmiac_knn_fused_ms(float *total_ms, int *launches);

/*
 * Closer-than-the-match counts, the integer behind FOSCTTM: replaces the dense
 * distance matrix and the `d < diag` comparison of foscttm
 * (scmiac/evaluation/benchmark.py:99-106) and of partially_matched_foscttm
 * (benchmark.py:148-157).  For query row i with true match r[match[i]] (match == NULL:
 * row i of r, which then needs m == n):
 *     counts[i] = #{ j : D(q_i, r_j) < D(q_i, r_match(i)) },   uint32 [n]
 * with D the float32 direct-difference squared distance of the k-NN entry points
 * (strict comparison, like the reference's `<`).  The N x M matrix is never formed.
 * FOSCTTM of row i is counts[i] / m.  *status (int32, device) becomes 1 if a match
 * index lies outside [0, m).  Workspace: scmiac_closer_count_workspace_size bytes.
 * Same limits as scmiac_knn_search (d <= 256, finite inputs).
 */
SCMIAC_API int scmiac_closer_count_workspace_size(int64_t n, int64_t m, int d, size_t *bytes);
SCMIAC_API int scmiac_closer_count(const float *q, int64_t n, const float *r, int64_t m, int d,
                                   const int32_t *match, uint32_t *counts, int32_t *status, void *workspace,
                                   size_t workspace_bytes, void *stream);

/*
 * Row permutation of a partially pivoted LU factorisation, on the device: the `permute_l`
 * part of scipy.linalg.lu as scikit-learn's randomized range finder uses it for
 * TruncatedSVD, which scmiac/preprocessing/anchors.py:196-198 calls.  pivots [m] int32 are
 * LAPACK's 1-based row interchanges (torch.linalg.lu_factor's second result), n the number
 * of rows; perm [n] int64 receives the identity with the interchanges applied in order
 * (perm[i] <-> perm[pivots[i]-1], i = 0..m-1), so that (P L)[perm] = L.  An entry of pivots
 * outside 1..n is no LAPACK pivot (a factorisation of non-finite input leaves such entries): it
 * is skipped and *status (int32, device, may be NULL; never cleared here) becomes 1.
 */
SCMIAC_API int scmiac_lu_row_perm(const int32_t *pivots, int m, int64_t n, int64_t *perm, int32_t *status,
                                  void *stream);

/*
 * Symmetric NT-Xent loss between the two embeddings of a batch of anchor cells:
 * replaces NTXentLoss2.forward (scmiac/modeling/loss.py:96-133) and its autograd
 * backward, one launch each.  zi, zj: float32 [batch, d] row-major on the device,
 * 1 <= d <= 64, batch >= 1.  Rows are L2-normalised inside (x / max(|x|, 1e-12)).
 *
 * forward: loss_parts[scmiac_ntxent2_parts(batch)] receives partial sums whose total
 * is the loss; lse_row / lse_col (float32 [batch]) receive the row and column
 * log-sum-exp of the similarity matrix, which backward reads.
 * backward: grad_out points at one float32 on the device (d loss); dzi, dzj receive
 * d loss / d zi, d loss / d zj.  The batch x batch similarity matrix is recomputed, not
 * stored.  Both take scmiac_ntxent2_workspace_size(batch, d) bytes of device scratch
 * (partial results of the up to 16 slices the other side's rows are cut into and, from
 * 512 rows on, where the products run on MFMA, the normalised rows of both sides).
 */
SCMIAC_API int scmiac_ntxent2_parts(int64_t batch);
SCMIAC_API int scmiac_ntxent2_workspace_size(int64_t batch, int d, size_t *bytes);
SCMIAC_API int scmiac_ntxent2_forward(const float *zi, const float *zj, int64_t batch, int d, float temperature,
                                      float *lse_row, float *lse_col, float *loss_parts, void *workspace,
                                      size_t workspace_bytes, void *stream);
SCMIAC_API int scmiac_ntxent2_backward(const float *zi, const float *zj, int64_t batch, int d, float temperature,
                                       const float *lse_row, const float *lse_col, const float *grad_out,
                                       float *dzi, float *dzj, void *workspace, size_t workspace_bytes,
                                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SCMIAC_HIP_H */
