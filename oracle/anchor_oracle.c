/*
 * TEST INFRASTRUCTURE -- CPU oracle for the anchor-search path.  Not part of the
 * shipped product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * Plain-C restatement of the reference algorithm in
 *   scmiac/preprocessing/anchors.py:525-566  (find_neighbor: four k-NN tables)
 *   scmiac/preprocessing/anchors.py:569-572  (find_mnn)
 *   scmiac/preprocessing/anchors.py:660-664  (score_anchor: shared-neighbour count)
 *
 * The k-NN arithmetic itself lives in the third-party package pynndescent
 * (pyproject.toml:38, unpinned, absent from the reference tree), which is an
 * APPROXIMATE nearest-neighbour index.  This oracle restates its published
 * contract with the approximation removed: Euclidean metric on float32-cast
 * inputs, squared distance accumulated directly as sum((a-b)^2) in float32,
 * rows sorted by ascending distance.  The definition used here, which the HIP
 * path must reproduce bit for bit:
 *
 *     D(a,b) = acc_d,  acc_0 = +0.0f,  acc_{i+1} = fmaf(a_i - b_i, a_i - b_i, acc_i)
 *
 * (feature index ascending, one rounding for the subtraction, one for the fused
 * multiply-add), neighbours ordered by the key (D, index) ascending, reported
 * distance sqrtf(D).  Parity at the pynndescent seam itself is unpinned: the
 * reference holds no test or fixture for it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

static inline float dist2_f32(const float *a, const float *b, int d) {
    float acc = 0.0f;
    for (int i = 0; i < d; ++i) {
        float t = a[i] - b[i];
        acc = fmaf(t, t, acc);
    }
    return acc;
}

/* threads an oracle_knn call with nthreads <= 0 will use */
ORACLE_API int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads of the parallel loops from now on (those of oracle_knn calls with nthreads <= 0 too) */
ORACLE_API void oracle_set_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

/* (D, idx) lexicographic "less than" */
static inline int key_less(float da, int64_t ia, float db, int64_t ib) {
    return (da < db) || (da == db && ia < ib);
}

/*
 * Exact k nearest rows of R for every row of Q.
 *   idx  [n*k] int32 : neighbour row numbers, ascending (D, index)
 *   dist2[n*k] float : D as defined above (callers take sqrtf for the distance)
 * k may exceed m: missing entries are index -1, distance +inf.
 * Follows the call shape of anchors.py:546-565 (index build + query collapse to
 * one exact search: neighbor_graph == search(data, data, k+1), query == search).
 */
ORACLE_API int oracle_knn(const float *Q, int64_t n, const float *R, int64_t m, int d,
                          int k, int32_t *idx, float *dist2, int nthreads) {
    if (n < 0 || m < 0 || d <= 0 || k <= 0) return 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        float *bd = (float *)malloc(sizeof(float) * (size_t)k);
        int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
#pragma omp for schedule(dynamic, 16)
        for (int64_t q = 0; q < n; ++q) {
            const float *a = Q + q * (int64_t)d;
            int cnt = 0;
            for (int64_t r0 = 0; r0 < m; r0 += 8) {
                /* eight independent accumulation chains: same per-pair arithmetic,
                 * more instruction-level parallelism than one chain at a time */
                float acc[8];
                int nb = (m - r0) < 8 ? (int)(m - r0) : 8;
                const float *b0 = R + r0 * (int64_t)d;
                if (nb == 8) {
                    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
                    for (int i = 0; i < d; ++i) {
                        float ai = a[i];
                        for (int j = 0; j < 8; ++j) {
                            float t = ai - b0[(int64_t)j * d + i];
                            acc[j] = fmaf(t, t, acc[j]);
                        }
                    }
                } else {
                    for (int j = 0; j < nb; ++j) acc[j] = dist2_f32(a, b0 + (int64_t)j * d, d);
                }
                for (int j = 0; j < nb; ++j) {
                    float dv = acc[j];
                    int64_t iv = r0 + j;
                    if (cnt == k && !key_less(dv, iv, bd[k - 1], bi[k - 1])) continue;
                    /* insertion into the sorted list (ascending key) */
                    int p = (cnt < k) ? cnt : k - 1;
                    while (p > 0 && key_less(dv, iv, bd[p - 1], bi[p - 1])) {
                        bd[p] = bd[p - 1];
                        bi[p] = bi[p - 1];
                        --p;
                    }
                    bd[p] = dv;
                    bi[p] = (int32_t)iv;
                    if (cnt < k) ++cnt;
                }
            }
            for (int j = 0; j < k; ++j) {
                idx[q * (int64_t)k + j] = (j < cnt) ? bi[j] : -1;
                dist2[q * (int64_t)k + j] = (j < cnt) ? bd[j] : INFINITY;
            }
        }
        free(bd);
        free(bi);
    }
    return 0;
}

/*
 * find_mnn (anchors.py:569-572):
 *   [[i, G12[i,j]] for i in range(n1) for j in range(ka) if i in G21[G12[i,j], :ka]]
 * pairs must hold n1*ka rows of two int64; returns the number of rows written,
 * in row-major (i, j) order, or -1 when an entry of G12 is not a row of G21.
 */
ORACLE_API int64_t oracle_find_mnn(const int32_t *G12, int64_t n1, int64_t ld12,
                                   const int32_t *G21, int64_t n2, int64_t ld21, int ka,
                                   int64_t *pairs) {
    int64_t cnt = 0;
    for (int64_t i = 0; i < n1; ++i) {
        for (int j = 0; j < ka; ++j) {
            int64_t y = G12[i * ld12 + j];
            /* numpy indexing semantics: negative indices wrap once */
            if (y < 0) y += n2;
            if (y < 0 || y >= n2) return -1;
            const int32_t *row = G21 + y * ld21;
            int hit = 0;
            for (int t = 0; t < ka; ++t)
                if ((int64_t)row[t] == i) { hit = 1; break; }
            if (hit) {
                pairs[2 * cnt] = i;
                pairs[2 * cnt + 1] = (int64_t)G12[i * ld12 + j];
                ++cnt;
            }
        }
    }
    return cnt;
}

/* number of DISTINCT values of a[0..ks) that occur in b[0..ks)  ==
 * len(set(a).intersection(b)) */
static int set_intersection_count(const int32_t *a, const int32_t *b, int ks) {
    int c = 0;
    for (int i = 0; i < ks; ++i) {
        int first = 1;
        for (int p = 0; p < i; ++p)
            if (a[p] == a[i]) { first = 0; break; }
        if (!first) continue;
        for (int j = 0; j < ks; ++j)
            if (b[j] == a[i]) { ++c; break; }
    }
    return c;
}

/*
 * Raw shared-neighbour score (anchors.py:660-664), before min_max:
 *   len(set(G11[x,:ks]) & set(G21[y,:ks])) + len(set(G12[x,:ks]) & set(G22[y,:ks]))
 */
ORACLE_API int oracle_snn_score(const int64_t *pairs, int64_t n_anchor, const int32_t *G11,
                                const int32_t *G12, const int32_t *G21, const int32_t *G22,
                                int64_t n1, int64_t n2, int64_t ld, int ks, int32_t *score) {
    int bad = 0;
    /* anchors are independent of each other: one per loop iteration, any order */
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t a = 0; a < n_anchor; ++a) {
        int64_t x = pairs[2 * a], y = pairs[2 * a + 1];
        if (x < 0) x += n1;
        if (y < 0) y += n2;
        if (x < 0 || x >= n1 || y < 0 || y >= n2) { bad |= 1; continue; }
        score[a] = set_intersection_count(G11 + x * ld, G21 + y * ld, ks) +
                   set_intersection_count(G12 + x * ld, G22 + y * ld, ks);
    }
    return bad;
}

/*
 * For every row i of Q: how many rows of R are closer to it than R[match[i]]
 * (match == NULL: R[i]), with the float32 direct-difference squared distance and a
 * strict comparison.  This is the integer behind FOSCTTM
 * (scmiac/evaluation/benchmark.py:102-106: dist matrix, `d < diag`, mean), taken
 * with the k-NN path's distance arithmetic instead of the reference's
 * x_sq + y_sq - 2 x.y^T, whose rounding depends on the BLAS in use.
 * Returns 1 if a match index is out of range.
 */
ORACLE_API int oracle_closer_count(const float *Q, int64_t n, const float *R, int64_t m, int d,
                                   const int32_t *match, uint32_t *counts) {
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(| : bad)
    for (int64_t i = 0; i < n; ++i) {
        int64_t mi = match ? (int64_t)match[i] : i;
        counts[i] = 0;
        if (mi < 0 || mi >= m) { bad |= 1; continue; }
        const float *q = Q + i * (int64_t)d;
        float dm = dist2_f32(q, R + mi * (int64_t)d, d);
        uint32_t c = 0;
        for (int64_t j = 0; j < m; ++j) c += dist2_f32(q, R + j * (int64_t)d, d) < dm;
        counts[i] = c;
    }
    return bad;
}
