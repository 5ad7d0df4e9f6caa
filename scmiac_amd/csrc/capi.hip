// extern "C" boundary of libscmiac_hip.so (declared in include/scmiac_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/scmiac_hip.h"
#include "kernels.h"

using namespace scmiac;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
static int hip_fail(hipError_t e, const char* what) {
    return fail(SCMIAC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(expr, what)                              \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return hip_fail(_e, what); \
    } while (0)

static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// workspace carve-up of one search
struct KnnWs {
    float* qp;
    float* rp;
    float* rnorm;
    unsigned long long* lists;
    void* pend;  // pending-candidate buffers, when the fused kernel keeps them in global memory
    unsigned* maxnorm;
    float* tau_out;  // [list_rows] threshold every shortlist ended with
    float* tau0;     // [n_pad] thresholds to start from (merge mode)
    LocalityWs loc;  // pivot partition (all NULL when the geometry has no cells)
    unsigned long long* diag;  // per-wave counters of the fused kernel (diagnostic runs)
    size_t diag_offset, diag_bytes;
    size_t stats_offset;       // pruned search: {stages streamed, stages of the dense stream} (0: none)
    ShareArgs share;           // stream sharing: queues, published thresholds, counters (cursor NULL: none)
    size_t share_ctl_offset, share_ctl_bytes;  // what a search clears first: cursor, pub, claims, counters
    size_t share_stats_offset;
    size_t total;
};
static KnnWs knn_carve(void* ws, long long n, long long m, int d, const KnnGeometry& g) {
    unsigned char* p = static_cast<unsigned char*>(ws);
    KnnWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* at = p + off;
        off += up256(bytes);
        return at;
    };
    w.qp = reinterpret_cast<float*>(p + off);
    off += up256(sizeof(float) * (size_t)g.n_pad * g.rowf);
    w.rp = reinterpret_cast<float*>(p + off);
    off += up256(sizeof(float) * (size_t)g.m_pad * g.ls);
    w.rnorm = reinterpret_cast<float*>(p + off);
    off += up256(sizeof(float) * (size_t)g.m_pad);
    w.lists = reinterpret_cast<unsigned long long*>(p + off);
    off += up256(sizeof(unsigned long long) * (size_t)g.list_rows * 64);
    w.pend = p + off;
    if (g.gp) off += up256(8 * (size_t)g.list_rows * g.pcap);
    w.maxnorm = reinterpret_cast<unsigned*>(p + off);
    off += 256;
    w.tau_out = reinterpret_cast<float*>(take(sizeof(float) * (size_t)g.list_rows));
    w.tau0 = reinterpret_cast<float*>(take(sizeof(float) * (size_t)g.n_pad));
    w.loc = LocalityWs{};
    w.stats_offset = 0;
    if (g.cells > 0) {
        const size_t rb = (size_t)knn_cell_block_rows(d);  // rows per block of the partition kernels
        const size_t nbq = ((size_t)n + rb - 1) / rb, nbr = ((size_t)m + rb - 1) / rb;
        w.loc.pivT = reinterpret_cast<float*>(take(sizeof(float) * (size_t)d * g.cells_pad));
        w.loc.pnorm = reinterpret_cast<float*>(take(sizeof(float) * (size_t)g.cells_pad));
        w.loc.cell_q = reinterpret_cast<cell_t*>(take(sizeof(cell_t) * (size_t)n));
        w.loc.cell_r = reinterpret_cast<cell_t*>(take(sizeof(cell_t) * (size_t)m));
        w.loc.hist_q = reinterpret_cast<int*>(take(sizeof(int) * nbq * g.cells));
        w.loc.hist_r = reinterpret_cast<int*>(take(sizeof(int) * nbr * g.cells));
        w.loc.cstart_q = reinterpret_cast<int*>(take(sizeof(int) * 2 * (size_t)(g.cells + 1)));  // starts, then totals
        w.loc.cstart_r = reinterpret_cast<int*>(take(sizeof(int) * 2 * (size_t)(g.cells + 1)));
        w.loc.perm_q = reinterpret_cast<int*>(take(sizeof(int) * (size_t)n));
        w.loc.perm_r = reinterpret_cast<int*>(take(sizeof(int) * (size_t)m));
        w.loc.home = reinterpret_cast<int*>(take(sizeof(int) * 4 * (size_t)g.nqb));
        w.loc.piv_rows = reinterpret_cast<int*>(take(sizeof(int) * 4 * (size_t)g.cells_pad));
        w.loc.pivT_tmp = reinterpret_cast<float*>(take(sizeof(float) * (size_t)d * g.cells_pad));
        w.loc.pnorm_tmp = reinterpret_cast<float*>(take(sizeof(float) * (size_t)g.cells_pad));
        w.loc.pivD = reinterpret_cast<_Float16*>(take(sizeof(_Float16) * 256 * 256 + 16));
        w.loc.pivRM = reinterpret_cast<float*>(take(sizeof(float) * (size_t)d * g.cells_pad));
        if (g.prune) {  // pruned search: distances to the pivots, cell radii, per-block stage lists, counters
            w.loc.qpd = reinterpret_cast<float*>(take(sizeof(float) * (size_t)g.cells_pad * (size_t)n));
            w.loc.crad = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * (size_t)g.cells_pad));
            w.loc.fps_sample = reinterpret_cast<float*>(take(sizeof(float) * (size_t)d * (size_t)knn_fps_sample_rows()));
            // a block's list holds at most the stages of its chunk; main and tail blocks differ in that
            const long long longest = g.chunk_rows > g.tail.chunk_rows ? g.chunk_rows : g.tail.chunk_rows;
            w.loc.order_stride = (int)(longest / g.RT + 1);
            size_t order_bytes = sizeof(unsigned short) * (size_t)g.blocks * (size_t)w.loc.order_stride;
            w.loc.order = reinterpret_cast<unsigned short*>(take(order_bytes));
            w.stats_offset = off;
            w.loc.stats = reinterpret_cast<unsigned long long*>(take(2 * sizeof(unsigned long long)));
        }
    }
    w.share = ShareArgs{};
    w.share_ctl_offset = w.share_ctl_bytes = w.share_stats_offset = 0;
    if (g.share > 0) {
        const size_t streams = (size_t)g.blocks;
        w.share_ctl_offset = off;
        w.share.started = reinterpret_cast<unsigned*>(take(sizeof(unsigned)));
        w.share.cursor = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * streams));
        w.share.pub = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * streams));
        w.share.claims = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * streams));
        w.share_stats_offset = off;
        w.share.stats = reinterpret_cast<unsigned long long*>(take((2 + 16 * streams) * sizeof(unsigned long long)));  // counters, then a trace entry per block
        w.share_ctl_bytes = off - w.share_ctl_offset;
        w.share.tau_pub = reinterpret_cast<float*>(take(sizeof(float) * streams * (size_t)g.qpb));
        w.share.row0 = g.share_row0;
        w.share.lists_per = g.share;
        w.share.chunk = g.share_chunk;
        w.share.min_left = g.share_min_left;
        w.share.tail_first = g.share_tail_first;
    }
    w.diag = reinterpret_cast<unsigned long long*>(p + off);
    w.diag_offset = off;
    w.diag_bytes = sizeof(unsigned long long) * 16 * (size_t)g.blocks * g.nw;
    off += up256(w.diag_bytes);
    w.total = off;
    return w;
}

static int check_knn_mode(int mode) {
    if (mode != SCMIAC_KNN_DENSE && mode != SCMIAC_KNN_PRUNED)
        return fail(SCMIAC_ERR_ARG, "mode %d is neither SCMIAC_KNN_DENSE nor SCMIAC_KNN_PRUNED", mode);
    return SCMIAC_OK;
}

static int check_knn_args(int64_t n, int64_t m, int d, int k) {
    if (n < 0 || m < 0) return fail(SCMIAC_ERR_ARG, "negative row count (n=%lld, m=%lld)", (long long)n, (long long)m);
    if (n >= (1ll << 31) || m >= (1ll << 31)) return fail(SCMIAC_ERR_ARG, "row counts must be below 2^31");
    if (d < 1 || d > SCMIAC_KNN_MAX_D) return fail(SCMIAC_ERR_ARG, "feature dimension %d outside 1..%d", d, SCMIAC_KNN_MAX_D);
    if (k < 1 || k > SCMIAC_KNN_MAX_K) return fail(SCMIAC_ERR_ARG, "k=%d outside 1..%d", k, SCMIAC_KNN_MAX_K);
    return SCMIAC_OK;
}

// optional timing of the fused kernel: one event pair per search since timing was enabled
// (g_timing_mu guards all of it: host threads may search concurrently)
static constexpr int kMaxTimed = 256;
static std::mutex g_timing_mu;
static bool g_timing = false;
static hipEvent_t g_ev0[kMaxTimed], g_ev1[kMaxTimed];
static int g_ev_created = 0;
static int g_ev_used = 0;

extern "C" {

int scmiac_version(void) { return 1000; }

const char* scmiac_last_error(void) { return g_err; }

int scmiac_knn_workspace_size(int64_t n, int64_t m, int d, int k, int splits, int mode, size_t* bytes) {
    if (!bytes) return fail(SCMIAC_ERR_ARG, "bytes is NULL");
    int rc = check_knn_args(n, m, d, k);
    if (rc) return rc;
    if ((rc = check_knn_mode(mode))) return rc;
    KnnGeometry g = knn_geometry(n, m, d, k, splits, mode, knn_get_tuning());
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    *bytes = knn_carve(nullptr, n, m, d, g).total;
    return SCMIAC_OK;
}

int scmiac_knn_diag_region(int64_t n, int64_t m, int d, int k, int splits, int mode, size_t* offset, size_t* bytes,
                           int* waves) {
    if (!offset || !bytes || !waves) return fail(SCMIAC_ERR_ARG, "NULL output pointer");
    int rc = check_knn_args(n, m, d, k);
    if (rc) return rc;
    if ((rc = check_knn_mode(mode))) return rc;
    KnnGeometry g = knn_geometry(n, m, d, k, splits, mode, knn_get_tuning());
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    KnnWs w = knn_carve(nullptr, n, m, d, g);
    *offset = w.diag_offset;
    *bytes = w.diag_bytes;
    *waves = g.blocks * g.nw;
    return SCMIAC_OK;
}

int scmiac_knn_prune_stats_offset(int64_t n, int64_t m, int d, int k, int splits, size_t* offset) {
    if (!offset) return fail(SCMIAC_ERR_ARG, "offset is NULL");
    int rc = check_knn_args(n, m, d, k);
    if (rc) return rc;
    KnnGeometry g = knn_geometry(n, m, d, k, splits, SCMIAC_KNN_PRUNED, knn_get_tuning());
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    if (!g.prune) return fail(SCMIAC_ERR_ARG, "a search of this shape does not prune (it runs dense)");
    *offset = knn_carve(nullptr, n, m, d, g).stats_offset;
    return SCMIAC_OK;
}

int scmiac_knn_share_info(int64_t n, int64_t m, int d, int k, int splits, int64_t* info, size_t* stats_offset) {
    if (!info || !stats_offset) return fail(SCMIAC_ERR_ARG, "NULL output pointer");
    int rc = check_knn_args(n, m, d, k);
    if (rc) return rc;
    KnnGeometry g = knn_geometry(n, m, d, k, splits, SCMIAC_KNN_DENSE, knn_get_tuning());
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    info[0] = g.share; info[1] = g.share_chunk; info[2] = g.share_min_left; info[3] = g.share > 0 ? g.blocks : 0;
    *stats_offset = g.share > 0 ? knn_carve(nullptr, n, m, d, g).share_stats_offset : 0;
    return SCMIAC_OK;
}

int scmiac_knn_grid_plan(int64_t n, int64_t m, int d, int k, int splits, int mode, int64_t* plan) {
    if (!plan) return fail(SCMIAC_ERR_ARG, "plan is NULL");
    int rc = check_knn_args(n, m, d, k);
    if (rc) return rc;
    if ((rc = check_knn_mode(mode))) return rc;
    KnnGeometry g = knn_geometry(n, m, d, k, splits, mode, knn_get_tuning());
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    plan[0] = g.nqb; plan[1] = g.qpb; plan[2] = g.S; plan[3] = g.chunk_rows;
    plan[4] = g.tail.qb0; plan[5] = g.tail.St; plan[6] = g.tail.chunk_rows; plan[7] = g.blocks;
    plan[8] = g.cells; plan[9] = g.prune;
    return SCMIAC_OK;
}

int scmiac_knn_kernel_name(int64_t n, int64_t m, int d, int k, int splits, int mode, char* buf, size_t buf_bytes) {
    if (!buf || buf_bytes == 0) return fail(SCMIAC_ERR_ARG, "buf is NULL or empty");
    int rc = check_knn_args(n, m, d, k);
    if (rc) return rc;
    if ((rc = check_knn_mode(mode))) return rc;
    const FusedTuning tn = knn_get_tuning();
    KnnGeometry g = knn_geometry(n, m, d, k, splits, mode, tn);
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    const int kind = (tn.diag || (tn.ablate & (63 | 128))) ? 1 : (g.prune ? 2 : 0);
    if (kind == 0 && g.share > 0)
        snprintf(buf, buf_bytes, "fused_knn_share_kernel<%d, %d, %d, %d>", g.D2, g.nw, g.RT, g.pcap);
    else
        snprintf(buf, buf_bytes, "fused_knn_kernel<%d, %d, %d, %d, %s, %d>", g.D2, g.nw, g.RT, g.pcap,
                 g.gp ? "true" : "false", kind);
    return SCMIAC_OK;
}

int scmiac_knn_search(const float* q, int64_t n, const float* r, int64_t m, int d, int k,
                      int64_t col_offset, int merge, int splits, int mode, int32_t* idx, float* dist2, float* dist,
                      double* lower_bound, int32_t* uncertified_rows, int32_t* n_uncertified,
                      int exact_fallback, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = check_knn_args(n, m, d, k);
    if (rc) return rc;
    if ((rc = check_knn_mode(mode))) return rc;
    if (col_offset < 0 || col_offset + m >= (1ll << 31)) return fail(SCMIAC_ERR_ARG, "col_offset out of range");
    if (n == 0) return SCMIAC_OK;
    if (!q || !idx || !dist2 || !lower_bound || !uncertified_rows || !n_uncertified || !workspace)
        return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    if (m > 0 && !r) return fail(SCMIAC_ERR_ARG, "NULL reference pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // one snapshot of the tuning state for the whole call: plan, carve-up and kernel choice agree
    const FusedTuning tn = knn_get_tuning();
    KnnGeometry g = knn_geometry(n, m, d, k, splits, mode, tn);
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    KnnWs w = knn_carve(workspace, n, m, d, g);
    if (w.total > workspace_bytes)
        return fail(SCMIAC_ERR_WORKSPACE, "workspace holds %zu bytes, %zu needed (sized for another mode, shape or "
                    "tuning state?)", workspace_bytes, w.total);

    HIP_TRY(hipMemsetAsync(w.maxnorm, 0, 256, stream), "memset maxnorm");
    HIP_TRY(hipMemsetAsync(n_uncertified, 0, sizeof(int32_t), stream), "memset n_uncertified");
    if (g.prune) {
        HIP_TRY(hipMemsetAsync(w.loc.crad, 0, sizeof(unsigned) * (size_t)g.cells_pad, stream), "memset radii");
        HIP_TRY(hipMemsetAsync(w.loc.stats, 0, 2 * sizeof(unsigned long long), stream), "memset counters");
    }
    if (g.share > 0)
        HIP_TRY(hipMemsetAsync(static_cast<unsigned char*>(workspace) + w.share_ctl_offset, 0, w.share_ctl_bytes, stream),
                "memset stream queues");
    if (g.cells > 0) HIP_TRY(launch_locality(q, n, r, m, d, g, w.loc, stream), "locality partition");
    HIP_TRY(launch_pack(q, n, d, g, -2.0f, w.qp, g.rowf, nullptr, g.n_pad, w.maxnorm, w.loc.perm_q, stream), "pack queries");
    HIP_TRY(launch_pack(r, m, d, g, 1.0f, w.rp, g.ls, w.rnorm, g.m_pad, w.maxnorm, w.loc.perm_r, stream), "pack references");
    // merge mode: the running k-th distances bound what this block can contribute
    const float* tau0 = nullptr;
    if (merge && m > 0) {
        HIP_TRY(launch_seed_tau(q, w.loc.perm_q, d, n, g, idx, dist2, k, w.maxnorm, w.tau0, stream), "seed thresholds");
        tau0 = w.tau0;
    }
    int slot = -1;  // event pair of this search, when timing is on
    {
        std::lock_guard<std::mutex> lock(g_timing_mu);
        if (g_timing && g_ev_used < kMaxTimed) {
            if (g_ev_used >= g_ev_created) {
                HIP_TRY(hipEventCreate(&g_ev0[g_ev_created]), "event create");
                HIP_TRY(hipEventCreate(&g_ev1[g_ev_created]), "event create");
                ++g_ev_created;
            }
            slot = g_ev_used++;
        }
    }
    if (slot >= 0) HIP_TRY(hipEventRecord(g_ev0[slot], stream), "event record");
    HIP_TRY(launch_fused(g, tn, w.qp, w.rp, w.rnorm, w.lists, w.pend, n, knn_threshold_rank(k, d), w.loc.home, tau0, w.tau_out,
                         knn_prune_args(g, w.loc, w.maxnorm, n, m, d), w.share, tn.diag ? w.diag : nullptr, stream),
            "fused distance + top-k");
    if (slot >= 0) HIP_TRY(hipEventRecord(g_ev1[slot], stream), "event record");
    HIP_TRY(launch_refine(q, w.qp, w.rp, d, n, g, w.lists, k, knn_threshold_rank(k, d), col_offset, merge, w.maxnorm, w.tau_out,
                          w.loc.perm_q, w.loc.perm_r, w.share.claims, idx, dist2, dist, lower_bound, uncertified_rows,
                          n_uncertified, stream),
            "refine");
    if (exact_fallback)  // the shortlists are dead once refine has run: their memory is the scratch
        HIP_TRY(launch_exact_rows(q, r, d, n, m, k, col_offset, uncertified_rows, n_uncertified, idx, dist2,
                                  dist, w.lists, sizeof(unsigned long long) * (size_t)g.list_rows * 64, stream),
                "exact rows");
    return SCMIAC_OK;
}

int scmiac_knn_exact_rows(const float* q, int64_t n, const float* r, int64_t m, int d, int k,
                          int64_t col_offset, const int32_t* rows, const int32_t* n_rows, int32_t* idx,
                          float* dist2, float* dist, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = check_knn_args(n, m, d, k);
    if (rc) return rc;
    if (n == 0) return SCMIAC_OK;
    if (!q || !rows || !n_rows || !idx || !dist2) return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    if (m > 0 && !r) return fail(SCMIAC_ERR_ARG, "NULL reference pointer");
    HIP_TRY(launch_exact_rows(q, r, d, n, m, k, col_offset, rows, n_rows, idx, dist2, dist, workspace,
                              workspace ? workspace_bytes : 0, static_cast<hipStream_t>(stream_)),
            "exact rows");
    return SCMIAC_OK;
}

int scmiac_compact_workspace_size(int64_t n_items, size_t* bytes) {
    if (!bytes) return fail(SCMIAC_ERR_ARG, "bytes is NULL");
    if (n_items < 0) return fail(SCMIAC_ERR_ARG, "negative item count");
    *bytes = compact_workspace_bytes(n_items);
    return SCMIAC_OK;
}

int scmiac_find_mnn(const int32_t* g12, int64_t n1, int64_t ld12, const int32_t* g21, int64_t n2,
                    int64_t ld21, int ka, int64_t* pairs, int64_t* n_pairs, int32_t* status,
                    void* workspace, size_t workspace_bytes, void* stream_) {
    if (n1 < 0 || n2 < 0 || ka < 0) return fail(SCMIAC_ERR_ARG, "negative size");
    if (ka > ld12 || ka > ld21)
        return fail(SCMIAC_ERR_ARG, "k_anchor=%d exceeds the neighbour columns (%lld, %lld)", ka,
                    (long long)ld12, (long long)ld21);
    if (!n_pairs || !status || !workspace) return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    if (n1 * (int64_t)ka > 0 && (!g12 || !g21 || !pairs)) return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    size_t need = compact_workspace_bytes(n1 * (int64_t)ka);
    if (need > workspace_bytes)
        return fail(SCMIAC_ERR_WORKSPACE, "workspace holds %zu bytes, %zu needed", workspace_bytes, need);
    HIP_TRY(launch_find_mnn(g12, n1, ld12, g21, n2, ld21, ka, reinterpret_cast<long long*>(pairs),
                            reinterpret_cast<long long*>(n_pairs), status, workspace,
                            static_cast<hipStream_t>(stream_)),
            "find_mnn");
    return SCMIAC_OK;
}

// ---- allocation-free variant of scmiac_find_mnn ------------------------------------------
// Its compaction scratch (flags + per-block counts) is this file-scope device array instead of
// a caller workspace.  Every GPU the library runs on holds its own copy at its own address;
// `scratch` below is only the host-side handle of the symbol, never a device address: host
// code uses the per-device pointer from hipGetSymbolAddress wherever the memory is meant.
static constexpr size_t kMnnStaticBytes =
    2 * (size_t)SCMIAC_MNN_NOALLOC_MAX_ITEMS + 2 * (((size_t)SCMIAC_MNN_NOALLOC_MAX_ITEMS / 1024 + 2) * 8 + 256) + 512;
static __device__ __align__(256) unsigned char scratch[kMnnStaticBytes];
static constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU, once resolved

// Address of this device's copy of `scratch` (resolved on the first call per device).
static int scratch_for_current_device(void** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
    if (dev < 0 || dev >= kMaxDevices) return fail(SCMIAC_ERR_ARG, "device ordinal %d outside 0..%d", dev, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return hip_fail(e, "hipGetSymbolAddress(scratch)");
        if (!ptr) return fail(SCMIAC_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return SCMIAC_OK;
}

int scmiac_find_mnn_noalloc(const int32_t* g12, int64_t n1, int64_t ld12, const int32_t* g21, int64_t n2,
                            int64_t ld21, int ka, int64_t* pairs, int64_t* n_pairs, int32_t* status, void* stream_) {
    if (n1 < 0 || n2 < 0 || ka < 0) return fail(SCMIAC_ERR_ARG, "negative size");
    if (ka > ld12 || ka > ld21)
        return fail(SCMIAC_ERR_ARG, "k_anchor=%d exceeds the neighbour columns (%lld, %lld)", ka,
                    (long long)ld12, (long long)ld21);
    const int64_t items = n1 * (int64_t)ka;
    if (items > (int64_t)SCMIAC_MNN_NOALLOC_MAX_ITEMS)
        return fail(SCMIAC_ERR_ARG, "n1 * k_anchor = %lld exceeds the static scratch (%lld candidate pairs); "
                    "use scmiac_find_mnn with a workspace", (long long)items, (long long)SCMIAC_MNN_NOALLOC_MAX_ITEMS);
    if (!n_pairs || !status) return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    if (items > 0 && (!g12 || !g21 || !pairs)) return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    if (compact_workspace_bytes(items) > kMnnStaticBytes)
        return fail(SCMIAC_ERR_WORKSPACE, "internal: static scratch smaller than the compaction needs");
    void* ws = nullptr;
    if (int rc = scratch_for_current_device(&ws)) return rc;
    // every byte of the scratch the kernels read they have written in this call (flags of all
    // candidate pairs, counts of all blocks): nothing left over from the previous call matters
    HIP_TRY(launch_find_mnn(g12, n1, ld12, g21, n2, ld21, ka, reinterpret_cast<long long*>(pairs),
                            reinterpret_cast<long long*>(n_pairs), status, ws, static_cast<hipStream_t>(stream_)),
            "find_mnn");
    return SCMIAC_OK;
}

int scmiac_snn_score(const int64_t* pairs, const int64_t* n_pairs, int64_t n_pairs_max, const int32_t* g11,
                     const int32_t* g12, const int32_t* g21, const int32_t* g22, int64_t n1, int64_t n2,
                     int64_t ld, int ks, int32_t* score, int64_t* hist, void* stream_) {
    if (n_pairs_max < 0 || n1 < 0 || n2 < 0) return fail(SCMIAC_ERR_ARG, "negative size");
    if (ks < 1 || ks > ld) return fail(SCMIAC_ERR_ARG, "k_score=%d outside 1..%lld", ks, (long long)ld);
    if (ks > 2048) return fail(SCMIAC_ERR_ARG, "k_score=%d too large (max 2048)", ks);
    if (n_pairs_max == 0) return SCMIAC_OK;
    if (!pairs || !n_pairs || !g11 || !g12 || !g21 || !g22 || !score)
        return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    HIP_TRY(launch_snn_score(reinterpret_cast<const long long*>(pairs),
                             reinterpret_cast<const long long*>(n_pairs), n_pairs_max, g11, g12, g21, g22, n1,
                             n2, ld, ks, score, reinterpret_cast<long long*>(hist),
                             static_cast<hipStream_t>(stream_)),
            "snn_score");
    return SCMIAC_OK;
}

int scmiac_anchor_filter(const int64_t* pairs, const int64_t* n_pairs, int64_t n_pairs_max,
                         const int32_t* raw_score, double p_lo, double p_hi, double threshold,
                         const int32_t* codes1, int64_t n1, const int32_t* codes2, int64_t n2, int ct_filter,
                         int64_t* out_pairs, double* out_score, uint8_t* out_is_same, int64_t* out_index,
                         int64_t* counts, double* all_score, void* workspace, size_t workspace_bytes,
                         void* stream_) {
    if (n_pairs_max < 0 || n1 < 0 || n2 < 0) return fail(SCMIAC_ERR_ARG, "negative size");
    if (!n_pairs || !counts || !workspace) return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    if (n_pairs_max > 0 && (!pairs || !raw_score || !codes1 || !codes2 || !out_pairs || !out_score ||
                            !out_is_same || !out_index))
        return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    size_t need = compact_workspace_bytes(n_pairs_max);
    if (need > workspace_bytes)
        return fail(SCMIAC_ERR_WORKSPACE, "workspace holds %zu bytes, %zu needed", workspace_bytes, need);
    HIP_TRY(launch_anchor_filter(reinterpret_cast<const long long*>(pairs),
                                 reinterpret_cast<const long long*>(n_pairs), n_pairs_max, raw_score, p_lo,
                                 p_hi, threshold, codes1, n1, codes2, n2, ct_filter,
                                 reinterpret_cast<long long*>(out_pairs), out_score, out_is_same,
                                 reinterpret_cast<long long*>(out_index), reinterpret_cast<long long*>(counts),
                                 all_score, workspace, static_cast<hipStream_t>(stream_)),
            "anchor_filter");
    return SCMIAC_OK;
}

int scmiac_lu_row_perm(const int32_t* pivots, int m, int64_t n, int64_t* perm, int32_t* status, void* stream_) {
    if (m < 0 || n < 0 || m > n) return fail(SCMIAC_ERR_ARG, "need 0 <= m <= n (m=%d, n=%lld)", m, (long long)n);
    if (n == 0) return SCMIAC_OK;
    if (!perm || (m > 0 && !pivots)) return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    HIP_TRY(launch_lu_row_perm(pivots, m, n, reinterpret_cast<long long*>(perm), status, static_cast<hipStream_t>(stream_)),
            "lu_row_perm");
    return SCMIAC_OK;
}

int scmiac_knn_set_timing(int enabled) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    g_timing = enabled != 0;
    g_ev_used = 0;
    return SCMIAC_OK;
}

int scmiac_knn_set_tuning(int slots, int tail_min_rows, int no_tail, int ablate, int diag, int nw, int rt, int pcap,
                          int gp, int no_locality, int home_rows, int seed_rows, int cell_rows) {
    FusedTuning t{};
    t.slots = slots; t.tail_min_rows = tail_min_rows; t.no_tail = no_tail; t.ablate = ablate; t.diag = diag;
    t.no_locality = no_locality; t.home_rows = home_rows; t.seed_rows = seed_rows; t.cell_rows = cell_rows;
    t.has_variant = (nw > 0 || rt > 0) ? 1 : 0;
    t.nw = nw; t.rt = rt; t.pcap = pcap; t.gp = gp;
    {   // (the sharing knobs have a call of their own and stay as they are)
        const FusedTuning cur = knn_get_tuning();
        t.share = cur.share; t.share_chunk = cur.share_chunk; t.share_min_left = cur.share_min_left;
        t.share_order = cur.share_order;
    }
    if (!knn_set_tuning(t))
        return fail(SCMIAC_ERR_ARG, "tuning rejected: negative value, or kernel variant %dx%dx%dx%d is not compiled",
                    nw, rt, pcap, gp);
    return SCMIAC_OK;
}

int scmiac_knn_set_sharing(int lists, int chunk, int min_left, int order) {
    FusedTuning t = knn_get_tuning();
    t.share = lists; t.share_chunk = chunk; t.share_min_left = min_left; t.share_order = order;
    if (!knn_set_tuning(t))
        return fail(SCMIAC_ERR_ARG, "sharing knobs rejected: lists %d outside -1..8, run of %d stages (0 or at least 2), "
                    "min_left %d negative, order %d outside 0..2", lists, chunk, min_left, order);
    return SCMIAC_OK;
}

int scmiac_knn_fused_ms(float* total_ms, int* launches) {
    if (!total_ms || !launches) return fail(SCMIAC_ERR_ARG, "NULL output pointer");
    std::lock_guard<std::mutex> lock(g_timing_mu);
    float sum = 0.f;
    for (int i = 0; i < g_ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(g_ev1[i]), "event synchronize");
        HIP_TRY(hipEventElapsedTime(&ms, g_ev0[i], g_ev1[i]), "event elapsed");
        sum += ms;
    }
    *total_ms = sum;
    *launches = g_ev_used;
    return SCMIAC_OK;
}

// workspace of a count: the k-NN carve-up, with the thresholds and counts in the (unused)
// shortlist region and the parked references in the pending-candidate region
// (the count kernel shares no streams: no helper lists in its plan)
static FusedTuning count_tuning() {
    FusedTuning t = knn_get_tuning();
    t.share = -1;
    return t;
}
static size_t count_pend_offset_total(const KnnGeometry& g, int64_t n, int64_t m, int d, KnnWs* w_out) {
    KnnWs w = knn_carve(nullptr, n, m, d, g);
    if (w_out) *w_out = w;
    return up256(w.total) + up256(closer_count_pend_bytes(g));
}

int scmiac_closer_count_workspace_size(int64_t n, int64_t m, int d, size_t* bytes) {
    if (!bytes) return fail(SCMIAC_ERR_ARG, "bytes is NULL");
    int rc = check_knn_args(n, m, d, 1);
    if (rc) return rc;
    KnnGeometry g = knn_geometry(n, m, d, 30, 0, SCMIAC_KNN_DENSE, count_tuning());
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    *bytes = count_pend_offset_total(g, n, m, d, nullptr);
    return SCMIAC_OK;
}

int scmiac_closer_count(const float* q, int64_t n, const float* r, int64_t m, int d, const int32_t* match,
                        uint32_t* counts, int32_t* status, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = check_knn_args(n, m, d, 1);
    if (rc) return rc;
    if (!match && n != m) return fail(SCMIAC_ERR_ARG, "without a match array the two sides need equal row counts");
    if (n == 0) return SCMIAC_OK;
    if (m == 0) return fail(SCMIAC_ERR_ARG, "no reference rows");
    if (!q || !r || !counts || !status || !workspace) return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KnnGeometry g = knn_geometry(n, m, d, 30, 0, SCMIAC_KNN_DENSE, count_tuning());
    if (g.D2 < 0) return fail(SCMIAC_ERR_ARG, "unsupported feature dimension %d", d);
    KnnWs w0;
    size_t total = count_pend_offset_total(g, n, m, d, &w0);
    if (total > workspace_bytes)
        return fail(SCMIAC_ERR_WORKSPACE, "workspace holds %zu bytes, %zu needed", workspace_bytes, total);
    KnnWs w = knn_carve(workspace, n, m, d, g);
    unsigned* pend = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + up256(w.total));
    float* lo = reinterpret_cast<float*>(w.lists);
    float* hi = lo + g.n_pad;
    float* dmatch = hi + g.n_pad;
    unsigned* cnt = reinterpret_cast<unsigned*>(dmatch + g.n_pad);
    if ((size_t)g.n_pad * 16 > sizeof(unsigned long long) * (size_t)g.list_rows * 64)
        return fail(SCMIAC_ERR_WORKSPACE, "internal: threshold region too small");
    HIP_TRY(hipMemsetAsync(w.maxnorm, 0, 256, stream), "memset maxnorm");
    HIP_TRY(hipMemsetAsync(status, 0, sizeof(int32_t), stream), "memset status");
    // the locality partition serves the count as well: with both sets in cell order the references
    // closer than a query's match sit in few stages, and every other subtile skips the counting
    if (g.share > 0)
        HIP_TRY(hipMemsetAsync(static_cast<unsigned char*>(workspace) + w.share_ctl_offset, 0, w.share_ctl_bytes, stream),
                "memset stream queues");
    if (g.cells > 0) HIP_TRY(launch_locality(q, n, r, m, d, g, w.loc, stream), "locality partition");
    HIP_TRY(launch_pack(q, n, d, g, -2.0f, w.qp, g.rowf, nullptr, g.n_pad, w.maxnorm, w.loc.perm_q, stream), "pack queries");
    HIP_TRY(launch_pack(r, m, d, g, 1.0f, w.rp, g.ls, w.rnorm, g.m_pad, w.maxnorm, w.loc.perm_r, stream), "pack references");
    HIP_TRY(launch_closer_count(g, q, r, match, d, n, m, w.qp, w.rp, w.rnorm, w.maxnorm, lo, hi, dmatch, cnt, pend,
                                status, w.loc.perm_q, stream),
            "closer count");
    HIP_TRY(launch_scatter_counts(cnt, w.loc.perm_q, n, counts, stream), "counts to the caller's rows");
    return SCMIAC_OK;
}

int scmiac_ntxent2_parts(int64_t batch) { return batch > 0 ? ntxent2_blocks((int)batch) : 0; }

static int check_ntxent_args(int64_t batch, int d, float temperature) {
    if (batch < 1 || batch > (1 << 24)) return fail(SCMIAC_ERR_ARG, "batch=%lld outside 1..2^24", (long long)batch);
    if (d < 1 || d > 64) return fail(SCMIAC_ERR_ARG, "d=%d outside 1..64", d);
    if (!(temperature > 0.f)) return fail(SCMIAC_ERR_ARG, "temperature must be positive");
    return SCMIAC_OK;
}

int scmiac_ntxent2_workspace_size(int64_t batch, int d, size_t* bytes) {
    if (!bytes) return fail(SCMIAC_ERR_ARG, "bytes is NULL");
    if (int rc = check_ntxent_args(batch, d, 1.0f)) return rc;
    *bytes = ntxent2_workspace_bytes((int)batch, d);
    return SCMIAC_OK;
}

int scmiac_ntxent2_forward(const float* zi, const float* zj, int64_t batch, int d, float temperature, float* lse_row,
                           float* lse_col, float* loss_parts, void* workspace, size_t workspace_bytes,
                           void* stream_) {
    if (int rc = check_ntxent_args(batch, d, temperature)) return rc;
    if (!zi || !zj || !lse_row || !lse_col || !loss_parts || !workspace)
        return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    size_t need = ntxent2_workspace_bytes((int)batch, d);
    if (need > workspace_bytes)
        return fail(SCMIAC_ERR_WORKSPACE, "workspace holds %zu bytes, %zu needed", workspace_bytes, need);
    HIP_TRY(launch_ntxent2_forward(zi, zj, (int)batch, d, temperature, lse_row, lse_col, loss_parts,
                                   static_cast<float*>(workspace), static_cast<hipStream_t>(stream_)),
            "ntxent2_forward");
    return SCMIAC_OK;
}

int scmiac_ntxent2_backward(const float* zi, const float* zj, int64_t batch, int d, float temperature,
                            const float* lse_row, const float* lse_col, const float* grad_out, float* dzi,
                            float* dzj, void* workspace, size_t workspace_bytes, void* stream_) {
    if (int rc = check_ntxent_args(batch, d, temperature)) return rc;
    if (!zi || !zj || !lse_row || !lse_col || !grad_out || !dzi || !dzj || !workspace)
        return fail(SCMIAC_ERR_ARG, "NULL device pointer");
    size_t need = ntxent2_workspace_bytes((int)batch, d);
    if (need > workspace_bytes)
        return fail(SCMIAC_ERR_WORKSPACE, "workspace holds %zu bytes, %zu needed", workspace_bytes, need);
    HIP_TRY(launch_ntxent2_backward(zi, zj, (int)batch, d, temperature, lse_row, lse_col, grad_out, dzi, dzj,
                                    static_cast<float*>(workspace), static_cast<hipStream_t>(stream_)),
            "ntxent2_backward");
    return SCMIAC_OK;
}

}  // extern "C"
