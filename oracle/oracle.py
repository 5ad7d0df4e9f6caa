"""TEST INFRASTRUCTURE -- CPU oracle for the anchor-search path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  It restates, on the CPU, what the reference computes
on the path

    find_neighbor -> find_mnn -> score_anchor/min_max -> label filter

(reference files: ``scmiac/preprocessing/anchors.py:525-581, 631-679, 779-789,
874-890, 971-987, 1021-1037`` and ``scmiac/modeling/scmiac.py:89-103``), with the
approximate third-party index ``pynndescent.NNDescent`` (``pyproject.toml:38``,
unpinned, not vendored) replaced by an exact search whose definition is written
out in ``anchor_oracle.c``.  Integer work is numpy / C; the float64 score
normalisation calls ``numpy.percentile`` exactly as the reference does.

Pinning: ``tests/golden/*.npz`` were produced by running the reference's own
``anchors.py`` / ``scmiac.py`` (loaded by path, absent third-party modules
stubbed, this module's exact search standing in for ``NNDescent``) -- see
``oracle/gen_golden.py``.  At the pynndescent seam itself parity is unpinned.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np
import pandas as pd

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    """Compile ``liboracle.so`` with the committed Makefile."""
    so = _HERE / "liboracle.so"
    if force or not so.exists() or so.stat().st_mtime < (_HERE / "anchor_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "liboracle.so"
        if not so.exists():
            build()
        lib = ctypes.CDLL(str(so))
        i64, i32p, f32p, i64p = (ctypes.c_int64, ctypes.POINTER(ctypes.c_int32),
                                 ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64))
        lib.oracle_knn.argtypes = [f32p, i64, f32p, i64, ctypes.c_int, ctypes.c_int, i32p, f32p,
                                   ctypes.c_int]
        lib.oracle_knn.restype = ctypes.c_int
        lib.oracle_find_mnn.argtypes = [i32p, i64, i64, i32p, i64, i64, ctypes.c_int, i64p]
        lib.oracle_find_mnn.restype = i64
        lib.oracle_snn_score.argtypes = [i64p, i64, i32p, i32p, i32p, i32p, i64, i64, i64,
                                         ctypes.c_int, i32p]
        lib.oracle_snn_score.restype = ctypes.c_int
        lib.oracle_closer_count.argtypes = [f32p, i64, f32p, i64, ctypes.c_int, i32p,
                                            ctypes.POINTER(ctypes.c_uint32)]
        lib.oracle_closer_count.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def host_threads() -> int:
    """Threads the C search uses by default (OpenMP's choice, e.g. OMP_NUM_THREADS)."""
    lib = _lib()
    lib.oracle_max_threads.restype = ctypes.c_int
    return int(lib.oracle_max_threads())


def set_threads(nthreads: int) -> None:
    """Threads of the C loops (k-NN search, shared-neighbour counts) from now on."""
    lib = _lib()
    lib.oracle_set_threads.argtypes = [ctypes.c_int]
    lib.oracle_set_threads.restype = None
    lib.oracle_set_threads(int(nthreads))


# --------------------------------------------------------------------------- k-NN
def knn(query, ref, k, nthreads: int = 0):
    """Exact k nearest rows of ``ref`` for every row of ``query``.

    Returns ``(idx int32 [n,k], dist float32 [n,k])``; distance is ``sqrtf(D)``
    of the float32 squared distance ``D`` that orders the rows (ties by index).
    """
    q = np.ascontiguousarray(np.asarray(query), dtype=np.float32)
    r = np.ascontiguousarray(np.asarray(ref), dtype=np.float32)
    if q.ndim != 2 or r.ndim != 2 or q.shape[1] != r.shape[1]:
        raise ValueError("query and ref must be 2-D with the same number of columns")
    n, d = q.shape
    m = r.shape[0]
    idx = np.empty((n, k), dtype=np.int32)
    d2 = np.empty((n, k), dtype=np.float32)
    rc = _lib().oracle_knn(_p(q, ctypes.c_float), n, _p(r, ctypes.c_float), m, d, int(k),
                           _p(idx, ctypes.c_int32), _p(d2, ctypes.c_float), int(nthreads))
    if rc != 0:
        raise ValueError("oracle_knn: bad arguments")
    return idx, np.sqrt(d2, dtype=np.float32)


def knn_dist2(query, ref, k, nthreads: int = 0):
    """As :func:`knn` but returns the ordering key ``D`` (squared distance)."""
    q = np.ascontiguousarray(np.asarray(query), dtype=np.float32)
    r = np.ascontiguousarray(np.asarray(ref), dtype=np.float32)
    n, d = q.shape
    idx = np.empty((n, k), dtype=np.int32)
    d2 = np.empty((n, k), dtype=np.float32)
    _lib().oracle_knn(_p(q, ctypes.c_float), n, _p(r, ctypes.c_float), r.shape[0], d, int(k),
                      _p(idx, ctypes.c_int32), _p(d2, ctypes.c_float), int(nthreads))
    return idx, d2


class ExactNNDescent:
    """Exact stand-in with the three members ``find_neighbor`` uses of
    ``pynndescent.NNDescent`` (call sites ``anchors.py:546-565``): the
    constructor, ``neighbor_graph`` and ``query``."""

    def __init__(self, data, metric="euclidean", n_neighbors=30, random_state=None,
                 parallel_batch_queries=False, n_jobs=None, **_ignored):
        if metric != "euclidean":
            raise ValueError("only the euclidean metric is restated")
        self._data = np.ascontiguousarray(np.asarray(data), dtype=np.float32)
        self.n_neighbors = int(n_neighbors)
        self._graph = None

    @property
    def neighbor_graph(self):
        if self._graph is None:
            self._graph = knn(self._data, self._data, self.n_neighbors)
        return self._graph

    def query(self, query_data, k=10, epsilon=0.1):
        return knn(query_data, self._data, k)


def find_neighbor(cc1, cc2, k, random_state=0, n_jobs=-1):
    """anchors.py:525-566 with the exact index."""
    index = ExactNNDescent(cc1, metric="euclidean", n_neighbors=k + 1)
    G11 = index.neighbor_graph[0][:, 1:k + 1]
    G21 = index.query(cc2, k=k)[0]
    index = ExactNNDescent(cc2, metric="euclidean", n_neighbors=k + 1)
    G22 = index.neighbor_graph[0][:, 1:k + 1]
    G12 = index.query(cc1, k=k)[0]
    return G11, G12, G21, G22


# ------------------------------------------------------------------ MNN / scoring
def find_mnn_loops(G12, G21, kanchor):
    """anchors.py:569-572 as plain Python loops (small inputs only)."""
    anchor = []
    for i in range(G12.shape[0]):
        for j in range(kanchor):
            row = G21[G12[i, j], :kanchor]
            if any(int(v) == i for v in row):
                anchor.append([i, G12[i, j]])
    return np.array(anchor)


def find_mnn(G12, G21, kanchor):
    """anchors.py:569-572; C loop, same row order and the same ``(0,)``-shaped
    result when no pair is mutual."""
    G12 = np.asarray(G12)
    G21 = np.asarray(G21)
    if kanchor > G12.shape[1] or kanchor > G21.shape[1]:
        raise IndexError("kanchor exceeds the number of neighbour columns")
    g12 = np.ascontiguousarray(G12, dtype=np.int32)
    g21 = np.ascontiguousarray(G21, dtype=np.int32)
    pairs = np.empty((g12.shape[0] * kanchor, 2), dtype=np.int64)
    cnt = _lib().oracle_find_mnn(_p(g12, ctypes.c_int32), g12.shape[0], g12.shape[1],
                                 _p(g21, ctypes.c_int32), g21.shape[0], g21.shape[1],
                                 int(kanchor), _p(pairs, ctypes.c_int64))
    if cnt < 0:
        raise IndexError("neighbour index out of bounds")
    if cnt == 0:
        return np.array([])
    return pairs[:cnt].copy()


def min_max(tmp, q_left=1, q_right=90):
    """anchors.py:575-581."""
    tmin, tmax = np.percentile(tmp, [q_left, q_right])
    tmp = (tmp - tmin) / (tmax - tmin)
    tmp[tmp > 1] = 1
    tmp[tmp < 0] = 0
    return tmp


def snn_counts(anchor, G11, G12, G21, G22, k_score=30):
    """The integer list ``tmp`` of anchors.py:660-664."""
    anchor = np.ascontiguousarray(np.asarray(anchor), dtype=np.int64)
    if anchor.size == 0:
        return np.empty(0, dtype=np.int32)
    tabs = [np.ascontiguousarray(np.asarray(g)[:, :k_score], dtype=np.int32)
            for g in (G11, G12, G21, G22)]
    ks = min(t.shape[1] for t in tabs)
    if any(t.shape[1] != ks for t in tabs):
        raise ValueError("neighbour tables must share their column count")
    out = np.empty(anchor.shape[0], dtype=np.int32)
    rc = _lib().oracle_snn_score(_p(anchor, ctypes.c_int64), anchor.shape[0],
                                 _p(tabs[0], ctypes.c_int32), _p(tabs[1], ctypes.c_int32),
                                 _p(tabs[2], ctypes.c_int32), _p(tabs[3], ctypes.c_int32),
                                 tabs[0].shape[0], tabs[2].shape[0], ks, ks,
                                 _p(out, ctypes.c_int32))
    if rc != 0:
        raise IndexError("anchor index out of bounds")
    return out


def snn_counts_loops(anchor, G11, G12, G21, G22, k_score=30):
    """anchors.py:660-664 with Python sets (small inputs only)."""
    return [
        len(set(G11[x, :k_score]).intersection(G21[y, :k_score]))
        + len(set(G12[x, :k_score]).intersection(G22[y, :k_score]))
        for x, y in anchor
    ]


def score_anchor(anchor, G11, G12, G21, G22, k_score=30, Gp1=None, Gp2=None, k_local=50):
    """anchors.py:631-679 (``k_local`` branch included for completeness)."""
    table = pd.DataFrame(anchor, columns=["x1", "x2"])
    table["score"] = min_max([int(v) for v in snn_counts(anchor, G11, G12, G21, G22, k_score)])
    if k_local:
        for col, local, joint, cells in (("score_local1", Gp1, G11, table["x1"].values),
                                         ("score_local2", Gp2, G22, table["x2"].values)):
            # per cell: how many local neighbours are among its first k_local joint-space ones
            shared = np.array([len(set(own) & set(row[:k_local])) for own, row in zip(local, joint)])
            table[col] = min_max([shared[c] for c in cells])
        table["score"] = table["score"] * table["score_local1"] * table["score_local2"]
    return table


# ---------------------------------------------------------------- upstream producer
def top_loading_features(data, n_features):
    """Genes with the largest absolute loadings over the components (anchors.py:132-168)."""
    n_cc = data.shape[0]
    per_dim = min(n_features * 10 // n_cc, data.shape[1] - 1)
    rows = np.arange(n_cc)[:, None]
    data = np.abs(data)
    idx = np.argpartition(-data, per_dim, axis=1)[:, :per_dim]
    idx = idx[rows, np.argsort(-data[rows, idx], axis=1)]
    for i in range(n_features // n_cc + 1, per_dim):
        features_idx = np.unique(idx[:, :i].flatten())
        if len(features_idx) > n_features:
            return features_idx
    return np.unique(idx[:, :per_dim].flatten())


def cca(data1, data2, scale1=True, scale2=True, n_components=50, max_cc_cell=20000, chunk_size=50000,
        random_state=0, k_filter=None, n_features=200):
    """Canonical correlation vectors as the reference computes them (anchors.py:171-242
    with the sub-sampling of 480-507): scikit-learn's StandardScaler and randomized
    TruncatedSVD on the dense ``X1 @ X2.T``; cells beyond ``max_cc_cell`` are projected
    afterwards in chunks.  Dense inputs only."""
    from sklearn.decomposition import TruncatedSVD
    from sklearn.preprocessing import StandardScaler

    np.random.seed(random_state)
    np.random.seed(random_state)
    sub, scalers = [], []
    for data, scale in ((data1, scale1), (data2, scale2)):
        n = data.shape[0]
        tf = data[np.random.choice(np.arange(n), max_cc_cell, False)] if n > max_cc_cell else data.copy()
        sub.append(np.asarray(tf, dtype=np.float64))
    for pos, scale in enumerate((scale1, scale2)):
        scalers.append(StandardScaler() if scale else None)
        if scale:
            sub[pos] = scalers[pos].fit_transform(sub[pos])
    tf1, tf2 = sub

    model = TruncatedSVD(n_components=n_components, algorithm="randomized", random_state=random_state)
    U = model.fit_transform(tf1.dot(tf2.T.copy()))
    keep = model.singular_values_ != 0
    sv = model.singular_values_[keep]
    V = model.components_[keep].T
    U = U[:, keep] / sv
    features = None
    if k_filter:
        features = top_loading_features(
            np.concatenate([U, V], axis=0).T.dot(np.concatenate([tf1, tf2], axis=0)), n_features)

    def chunks(data, scaler):
        for start in range(0, data.shape[0], chunk_size):
            tmp = np.asarray(data[start:start + chunk_size], dtype=np.float64)
            yield scaler.transform(tmp) if scaler is not None else tmp

    if data2.shape[0] > max_cc_cell:
        left = np.dot(U.T, tf1)
        V = np.concatenate([np.dot(left, tmp.T).T for tmp in chunks(data2, scalers[1])], axis=0) / sv
    if data1.shape[0] > max_cc_cell:
        right = np.dot(model.components_[keep], tf2).T
        U = np.concatenate([np.dot(tmp, right) for tmp in chunks(data1, scalers[0])], axis=0) / sv
    return U, V, features


# ------------------------------------------------------------------------ evaluation
def closer_count(query, ref, match=None):
    """``counts[i] = #{j : D(q_i, r_j) < D(q_i, r_match(i))}`` (uint32), float32
    direct-difference distances, strict comparison."""
    q = np.ascontiguousarray(np.asarray(query), dtype=np.float32)
    r = np.ascontiguousarray(np.asarray(ref), dtype=np.float32)
    n, d = q.shape
    counts = np.empty(n, dtype=np.uint32)
    mp = None
    if match is not None:
        match = np.ascontiguousarray(np.asarray(match), dtype=np.int32)
        mp = _p(match, ctypes.c_int32)
    elif r.shape[0] != n:
        raise ValueError("Shapes do not match!")
    if _lib().oracle_closer_count(_p(q, ctypes.c_float), n, _p(r, ctypes.c_float), r.shape[0], d, mp,
                                  _p(counts, ctypes.c_uint32)):
        raise IndexError("match index out of bounds")
    return counts


def foscttm(x, y):
    """Fraction of samples closer than the true match, both directions
    (evaluation/benchmark.py:75-112), as float32 like the reference's ``.float().mean()``."""
    x = np.asarray(x)
    y = np.asarray(y)
    if x.shape != y.shape:
        raise ValueError("Shapes do not match!")
    n = np.float32(x.shape[0])
    return (closer_count(x, y).astype(np.float32) / n, closer_count(y, x).astype(np.float32) / n)


def neighbor_conservation(orig_data, latent_data, n_neighbors=10, exclude_top=0):
    """Mean Jaccard index of the k-NN sets in the two spaces (evaluation/benchmark.py:35-68),
    with the exact (distance, index)-ordered search."""
    k = n_neighbors + exclude_top
    a = knn(orig_data, orig_data, k)[0][:, exclude_top:]
    b = knn(latent_data, latent_data, k)[0][:, exclude_top:]
    scores = [len(set(u) & set(v)) / len(set(u) | set(v)) for u, v in zip(a, b)]
    return np.mean(scores)


def knn_matching(rna_latent, atac_latent, k=1):
    """Share of cells whose true match is among the k most cosine-similar cells of the
    other modality (evaluation/benchmark.py:342-374): on unit rows, cosine order is
    Euclidean order."""
    a = l2_normalize_rows(rna_latent).astype(np.float32)
    b = l2_normalize_rows(atac_latent).astype(np.float32)
    hit_a = (knn(a, b, k)[0] == np.arange(len(a))[:, None]).any(axis=1).sum()
    hit_b = (knn(b, a, k)[0] == np.arange(len(b))[:, None]).any(axis=1).sum()
    return {"rna_to_atac_accuracy": np.round(hit_a / len(a), 5),
            "atac_to_rna_accuracy": np.round(hit_b / len(b), 5),
            "overall_accuracy": np.round((hit_a + hit_b) / (len(a) + len(b)), 5)}


# ---------------------------------------------------------------- downstream consumer
def ntxent2(z_i, z_j, temperature=0.5):
    """Symmetric NT-Xent loss (modeling/loss.py:108-132) in float64 numpy; returns
    ``(loss, d loss / d z_i, d loss / d z_j)``."""
    z_i = np.asarray(z_i, dtype=np.float64)
    z_j = np.asarray(z_j, dtype=np.float64)
    B = z_i.shape[0]
    ni = np.maximum(np.linalg.norm(z_i, axis=1, keepdims=True), 1e-12)
    nj = np.maximum(np.linalg.norm(z_j, axis=1, keepdims=True), 1e-12)
    a, b = z_i / ni, z_j / nj
    S = a @ b.T / temperature

    def lse(M, axis):
        m = M.max(axis=axis, keepdims=True)
        return np.squeeze(m, axis) + np.log(np.exp(M - m).sum(axis=axis))

    lr, lc = lse(S, 1), lse(S, 0)
    diag = np.diag(S)
    loss = 0.5 * ((lr - diag).mean() + (lc - diag).mean())
    dS = 0.5 / B * (np.exp(S - lr[:, None]) + np.exp(S - lc[None, :]) - 2.0 * np.eye(B))
    ga, gb = dS @ b / temperature, dS.T @ a / temperature
    dzi = (ga - a * (a * ga).sum(1, keepdims=True)) / ni
    dzj = (gb - b * (b * gb).sum(1, keepdims=True)) / nj
    return loss, dzi, dzj


# ------------------------------------------------------------ pairwise orchestration
def l2_normalize_rows(X):
    """sklearn.preprocessing.normalize(X, axis=1) for dense input
    (anchors.py:877-878): rows divided by their L2 norm, zero rows kept."""
    X = np.array(X, dtype=np.float64, copy=True)
    norms = np.sqrt(np.einsum("ij,ij->i", X, X))
    norms[norms == 0.0] = 1.0
    X /= norms[:, np.newaxis]
    return X


def pairwise_anchors(U, V, k_anchor=5, k_score=30, k_local=None, normalize_rows=False):
    """Steps 3, 4 and 6 of ``_pairwise_find_anchor`` (anchors.py:874-890 for the
    "cca" branch with ``normalize_rows``; 971-987 for "precomputed"; 1021-1037)."""
    if normalize_rows:
        U = l2_normalize_rows(U)
        V = l2_normalize_rows(V)
    min_sample = min(U.shape[0], V.shape[0])
    _k = max(t for t in [k_anchor, k_local, k_score] if t is not None)
    _k = min(min_sample - 2, _k)
    G11, G12, G21, G22 = find_neighbor(U, V, k=_k)
    raw_anchors = find_mnn(G12, G21, k_anchor)
    anchor_df = score_anchor(raw_anchors, G11, G12, G21, G22, k_score=k_score, k_local=k_local)
    return anchor_df, (G11, G12, G21, G22), raw_anchors


def label_filter(anchor_df, rna_labels, atac_labels, ct_filter=True, threshold=0.2):
    """scmiac/modeling/scmiac.py:90-99."""
    anchor_df = anchor_df.copy()
    anchor_df["x1_ct"] = np.asarray(rna_labels)[anchor_df["x1"].values]
    anchor_df["x2_ct"] = np.asarray(atac_labels)[anchor_df["x2"].values]
    anchor_df["is_same"] = anchor_df["x1_ct"].astype(str) == anchor_df["x2_ct"].astype(str)
    anchor_df = anchor_df[anchor_df["score"] > threshold].reset_index(drop=True)
    if ct_filter:
        anchor_df = anchor_df[anchor_df["is_same"]]
    return anchor_df
