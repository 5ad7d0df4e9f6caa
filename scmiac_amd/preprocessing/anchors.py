"""Drop-in for the anchor engine of the reference
(``scmiac/preprocessing/anchors.py``): same names, arguments and return types,
with the k-NN / MNN / SNN arithmetic running on the MI355X through
``libscmiac_hip.so``.

Mirrored entry points (reference line numbers in parentheses):
``find_neighbor`` (525), ``find_mnn`` (569), ``min_max`` (575), ``filter_anchor``
(584), ``score_anchor`` (631), ``SeuratIntegration`` (696) with ``find_anchor``
(1039) for ``dim_red`` in {"cca", "pca", "precomputed"}, ``cca`` (171),
``downsample`` (480), ``get_cca_adata_list`` (26) and an exact index object with
the three members the reference uses of ``pynndescent.NNDescent``.

numpy arrays in, numpy / pandas out; inputs are never modified.  ``random_state``
and ``n_jobs`` are accepted for signature compatibility and have no effect: the
search is exact and runs on the GPU.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import pandas as pd
import torch

from .. import _hip
from .. import anchor_search as _as

VALID_DIM_RED = ["pca", "cca", "lsi", "lsi-cca", "rpca", "rlsi", "precomputed"]


def _device():
    _hip.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _to_device_f32(x):
    a = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
    return torch.from_numpy(a).to(_device())


def _to_device_i32(x):
    a = np.ascontiguousarray(np.asarray(x), dtype=np.int32)
    return torch.from_numpy(a).to(_device())


# ------------------------------------------------------------------------ index
class NNDescent:
    """Exact Euclidean index with the interface ``find_neighbor`` expects of
    ``pynndescent.NNDescent`` (call sites anchors.py:546-565, 616-624, 732-740):
    ``NNDescent(data, metric, n_neighbors, ...)``, ``.neighbor_graph`` and
    ``.query(q, k)``, each returning ``(int32 indices, float32 distances)`` with
    rows sorted by ascending distance (ties by index).  More than 64 neighbours
    are served by merging group-wise searches (:mod:`scmiac_amd.wide_knn`)."""

    def __init__(self, data, metric="euclidean", n_neighbors=30, random_state=None,
                 parallel_batch_queries=False, n_jobs=None, **kwargs):
        if metric != "euclidean":
            raise ValueError(f"metric {metric!r} is not supported (euclidean only)")
        self._data = data if isinstance(data, torch.Tensor) else _to_device_f32(data)
        self.n_neighbors = int(n_neighbors)
        self._graph = None

    @property
    def neighbor_graph(self):
        if self._graph is None:
            k = min(self.n_neighbors, self._data.shape[0])
            self._graph = self._search(self._data, k)
        return self._graph

    def _search(self, q, k):
        if k > _hip.KNN_MAX_K:
            from ..wide_knn import knn_search_wide
            idx, _, dist = knn_search_wide(q, self._data, k)
            return idx.cpu().numpy(), dist.cpu().numpy()
        st = _hip.knn_search(q, self._data, k)
        return st.idx.cpu().numpy(), st.dist.cpu().numpy()

    def query(self, query_data, k=10, epsilon=0.1):
        q = query_data if isinstance(query_data, torch.Tensor) else _to_device_f32(query_data)
        return self._search(q, int(k))


# ------------------------------------------------------------------- hot path
def find_neighbor(cc1, cc2, k, random_state=0, n_jobs=-1):
    """Four k-NN tables between two embeddings (anchors.py:525-566).

    Returns ``G11, G12, G21, G22``: int32 arrays of shape (n_cell, k)."""
    tables = _as.find_neighbor_device(_to_device_f32(cc1), _to_device_f32(cc2), int(k))
    return tables.numpy()


def find_mnn(G12, G21, kanchor):
    """Mutual nearest neighbours (anchors.py:569-572): int64 array (n_anchor, 2)
    in (i, j) order; ``np.array([])`` (shape ``(0,)``) when there is none."""
    G12 = np.asarray(G12)
    G21 = np.asarray(G21)
    if kanchor > G12.shape[1] or kanchor > G21.shape[1]:
        raise IndexError(f"index {kanchor - 1} is out of bounds for the neighbour columns")
    pairs, n_pairs, status = _hip.find_mnn(_to_device_i32(G12), _to_device_i32(G21), int(kanchor))
    n = int(n_pairs.item())
    if int(status.item()) != 0:
        raise IndexError("neighbour index out of bounds")
    if n == 0:
        return np.array([])
    return pairs[:n].cpu().numpy()


def min_max(tmp, q_left=1, q_right=90):
    """Rescale to the [q_left, q_right] percentile range and clip (anchors.py:575-581)."""
    tmin, tmax = np.percentile(tmp, [q_left, q_right])
    tmp = (tmp - tmin) / (tmax - tmin)
    tmp[tmp > 1] = 1
    tmp[tmp < 0] = 0
    return tmp


def _snn_raw(anchor, G11, G12, G21, G22, k_score):
    anchor = np.asarray(anchor)
    for x, y in anchor:  # the reference unpacks rows: a (0,)-shaped array passes, a 1-D one raises
        break
    if anchor.size == 0:
        return []
    dev = _device()
    pairs = torch.from_numpy(np.ascontiguousarray(anchor, dtype=np.int64)).to(dev)
    n_pairs = torch.tensor([pairs.shape[0]], dtype=torch.int64, device=dev)
    ks = int(min(k_score, *(np.asarray(g).shape[1] for g in (G11, G12, G21, G22))))
    tabs = [_to_device_i32(np.asarray(g)[:, :ks]) for g in (G11, G12, G21, G22)]
    raw, _ = _hip.snn_score(pairs, n_pairs, *tabs, ks)
    raw = raw[:pairs.shape[0]].cpu().numpy()
    if (raw < 0).any():
        raise IndexError("anchor index out of bounds")
    return [int(v) for v in raw]


def score_anchor(anchor, G11, G12, G21, G22, k_score=30, Gp1=None, Gp2=None, k_local=50):
    """Shared-nearest-neighbour anchor score (anchors.py:631-679):
    ``DataFrame[x1, x2, score]`` (+ the local-structure columns when ``k_local``)."""
    table = pd.DataFrame(anchor, columns=["x1", "x2"])
    table["score"] = min_max(_snn_raw(anchor, G11, G12, G21, G22, k_score))
    if not k_local:
        return table

    # intra-dataset agreement (off the product path: find_anchors passes k_local=None): how many
    # of a cell's local neighbours are also among its first k_local neighbours in the joint space
    def local_agreement(local, joint, cells):
        shared = np.fromiter((len(set(own) & set(row[:k_local])) for own, row in zip(local, joint)),
                             dtype=np.int64, count=len(local))
        return min_max([shared[c] for c in cells])

    table["score_local1"] = local_agreement(Gp1, G11, table["x1"].values)
    table["score_local2"] = local_agreement(Gp2, G22, table["x2"].values)
    table["score"] = table["score"] * table["score_local1"] * table["score_local2"]
    return table


def filter_anchor(anchor, adata_ref=None, adata_qry=None, scale_ref=False, scale_qry=False,
                  high_dim_feature=None, k_filter=200, random_state=0, n_jobs=-1):
    """Keep anchors that are still neighbours in the high-loading feature space
    (anchors.py:584-628).  Up to 256 features."""
    from scipy.sparse import issparse
    from scipy.stats import zscore
    from sklearn.preprocessing import normalize

    def _prep(adata, scale):
        X = adata.X[:, high_dim_feature]
        X = X.toarray() if issparse(X) else X.copy()
        if scale:
            X = zscore(X, axis=0)
        return normalize(X, axis=1)

    ref_data = _prep(adata_ref, scale_ref)
    qry_data = _prep(adata_qry, scale_qry)
    neighbours = NNDescent(ref_data, metric="euclidean", n_neighbors=k_filter).query(qry_data, k=k_filter)[0]
    n_before = anchor.shape[0]
    kept = np.array([pair for pair in anchor if pair[0] in neighbours[pair[1]]])
    print(f"Anchor selected with high CC feature graph: {kept.shape[0]} / {n_before}")
    return kept


# -------------------------------------------------------------- upstream producer
def downsample(data1, data2, scale1, scale2, todense, max_cc_cell=20000, random_state=0):
    """Cell subsampling + optional standardisation before CCA (anchors.py:480-507)."""
    from scipy.sparse import issparse
    from sklearn.preprocessing import StandardScaler

    scalers = [None, None]
    out = []
    np.random.seed(random_state)
    for pos, (data, scale) in enumerate(((data1, scale1), (data2, scale2))):
        if data.shape[0] > max_cc_cell:
            sel = np.random.choice(np.arange(data.shape[0]), min(max_cc_cell, data.shape[0]), False)
            tf = data[sel]
        else:
            tf = data.copy()
        if todense and issparse(tf):
            tf = tf.toarray()
        out.append(tf)
    for pos, scale in enumerate((scale1, scale2)):
        if scale:
            scalers[pos] = StandardScaler()
            out[pos] = scalers[pos].fit_transform(out[pos])
    return out[0], out[1], scalers[0], scalers[1]


def cca(data1, data2, scale1=True, scale2=True, n_components=50, max_cc_cell=20000, chunk_size=50000,
        random_state=0, svd_algorithm="randomized", k_filter=None, n_features=200):
    """Canonical correlation vectors by truncated SVD of ``X1 @ X2.T``
    (anchors.py:171-242), on the GPU and without forming the cell x cell matrix
    (:mod:`scmiac_amd.cca_device`).  Returns ``(U, V, high_dim_feature)`` as float64
    numpy arrays.  Only the reference's default ``svd_algorithm="randomized"`` is
    provided."""
    from ..cca_device import cca_device

    if svd_algorithm != "randomized":
        raise NotImplementedError(f"svd_algorithm={svd_algorithm!r}: only 'randomized' is provided")
    return cca_device(data1, data2, scale1=scale1, scale2=scale2, n_components=n_components,
                      max_cc_cell=max_cc_cell, chunk_size=chunk_size, random_state=random_state,
                      k_filter=k_filter, n_features=n_features)


def top_features_idx(data, n_features):
    """Features with the highest absolute loadings over the components (anchors.py:132-168):
    per component the features ranked by |loading|; the shortest common rank prefix whose
    union over the components exceeds ``n_features`` features is returned (sorted, unique)."""
    loadings = np.abs(data)
    n_comp, n_all = loadings.shape
    depth = min(n_features * 10 // n_comp, n_all - 1)
    rows = np.arange(n_comp)[:, None]
    top = np.argpartition(-loadings, depth, axis=1)[:, :depth]          # the `depth` largest, unordered
    ranked = top[rows, np.argsort(-loadings[rows, top], axis=1)]        # ... ordered by loading
    for prefix in range(n_features // n_comp + 1, depth):
        chosen = np.unique(ranked[:, :prefix].ravel())
        if chosen.size > n_features:
            return chosen
    return np.unique(ranked.ravel())


def get_cca_adata_list(adatas, all_nfeatures=3000, single_nfeatures=3000):
    """Genes for the CCA (anchors.py:26-90): among the genes all data sets share, the
    ``all_nfeatures`` with the smallest sum of per-data-set ranks of normalised dispersion
    (scanpy's "seurat" flavour).  Returns the raw-count AnnData objects cut to those genes.
    Needs scanpy."""
    try:
        import scanpy as sc
    except ImportError as exc:  # pragma: no cover - scanpy is an optional dependency
        raise ImportError("get_cca_adata_list needs scanpy (highly_variable_genes)") from exc

    raws = []
    for adata in adatas:
        raw = adata.raw.to_adata()
        raw.raw = raw
        raws.append(raw)
    shared = set(raws[0].var_names).intersection(*(set(r.var_names) for r in raws[1:]))
    shared = pd.Index(list(shared))
    print(f"Number of common genes: {len(shared)}")

    ranks = pd.DataFrame(index=shared)
    for pos, raw in enumerate(raws, start=1):
        sc.pp.highly_variable_genes(raw, flavor="seurat", n_top_genes=single_nfeatures, inplace=True)
        order = raw.var.loc[shared].sort_values(by="dispersions_norm", ascending=False).index
        ranks[f"adata{pos}_rank"] = pd.Series(np.arange(1, len(order) + 1), index=order)
    ranks["rank_sum"] = ranks.sum(axis=1)
    ranks = ranks.sort_values(by="rank_sum", ascending=True)
    genes = ranks.nsmallest(all_nfeatures, "rank_sum").index.tolist()
    return [raw[:, genes].copy() for raw in raws]


# ------------------------------------------------------------------ orchestration
class SeuratIntegration:
    """Anchor finder with the reference class's surface (anchors.py:696-1187):
    ``find_anchor(adata_list, ...)`` then ``.anchor[(0, 1)]``, ``.mutual_knn``,
    ``.raw_anchor``, ``.U``, ``.V``."""

    def __init__(self, n_jobs=-1, random_state=0):
        self.n_jobs, self.random_state = n_jobs, random_state
        # the data sets and what find_anchor is asked for
        self.adata_dict, self.n_dataset, self.n_cells = OrderedDict(), 0, []
        self.k_local, self.key_local, self.local_knn = None, None, []
        self.alignments, self.all_pairs = None, np.array([])
        # results, keyed by the pair of data-set numbers
        self.anchor, self.mutual_knn, self.raw_anchor = {}, {}, {}
        self.label_transfer_results = {}
        # device-resident result of the last pairwise search, for the label filter
        self.device_scores = {}

    def _calculate_local_knn(self):
        if self.k_local is None:
            self.local_knn = [None] * len(self.adata_dict)
            return
        print("Find neighbors within datasets")
        for adata in self.adata_dict.values():
            graph = NNDescent(adata.obsm[self.key_local], metric="euclidean", n_neighbors=self.k_local + 1,
                              random_state=self.random_state).neighbor_graph[0]
            self.local_knn.append(graph[:, 1:])  # column 0 is the cell itself

    def _get_all_pairs(self):
        if self.alignments is None:
            self.all_pairs = np.array([])
            return
        names = [f"{min(a, b)}-{max(a, b)}" for left, right in self.alignments for a in left for b in right]
        self.all_pairs = np.unique(names)

    def _prepare_matrix(self, i, j, key_anchor):
        adata1, adata2 = self.adata_dict[i], self.adata_dict[j]
        if key_anchor == "X":
            if (adata1.shape[1] != adata2.shape[1]) or (
                    (adata1.var.index == adata2.var.index).sum() < adata1.shape[1]):
                sel_b = adata1.var.index.intersection(adata2.var.index)
                return adata1[:, sel_b].X.copy(), adata2[:, sel_b].X.copy()
            return adata1.X.copy(), adata2.X.copy()
        return adata1.obsm[key_anchor], adata2.obsm[key_anchor]

    def _calculate_mutual_knn_and_raw_anchors(self, i, j, U, V, k, k_anchor, k_score=None):
        """anchors.py:779-789; the tables stay on the device for the scoring step."""
        tables = _as.find_neighbor_device(_to_device_f32(U), _to_device_f32(V), int(k))
        scores, n_pairs = _as.mnn_and_score(tables, k_anchor=k_anchor,
                                            k_score=min(k_score if k_score else k, k))
        G11, G12, G21, G22 = tables.numpy()
        raw_anchors = scores.pairs[:scores.n].cpu().numpy() if scores.n else np.array([])
        self.mutual_knn[(i, j)] = (G11, G12, G21, G22)
        self.raw_anchor[(i, j)] = raw_anchors
        self.device_scores[(i, j)] = (scores, n_pairs)
        return G11, G12, G21, G22, raw_anchors

    def _pairwise_find_anchor(self, i, i_sel, j, j_sel, dim_red, key_anchor, svd_algorithm, scale1, scale2,
                              k_anchor, k_local, k_score, ncc, max_cc_cell, k_filter, n_features, chunk_size,
                              random_state, signorm):
        """Anchors between two datasets (anchors.py:791-1037)."""
        from sklearn.preprocessing import normalize

        adata1, adata2 = self.adata_dict[i], self.adata_dict[j]
        min_sample = min(adata1.shape[0], adata2.shape[0])
        if i_sel is not None or j_sel is not None:
            # The reference binds the selected cells to local names (anchors.py:819-822) that
            # _prepare_matrix never sees (758-777 read self.adata_dict again), so its search runs on
            # ALL cells and find_anchor then indexes the group's cell list with all-cell row numbers
            # (1155-1156): an IndexError or silently wrong pairs.  There is no behaviour to mirror.
            raise NotImplementedError("per-group anchor search (key_match) is not provided")
        high_dim_feature = None

        if dim_red in ("cca", "pca"):
            print("1. prepare input matrix for CCA")
            U, V = self._prepare_matrix(i, j, key_anchor=key_anchor)
            print("2. run cca between datasets")
            print("Run CCA")
            U, V, high_dim_feature = cca(data1=U, data2=V, scale1=scale1, scale2=scale2, n_components=ncc,
                                         max_cc_cell=max_cc_cell, k_filter=k_filter, n_features=n_features,
                                         chunk_size=chunk_size, svd_algorithm=svd_algorithm,
                                         random_state=random_state)
            print("3. normalize CCV per sample/row")
            U = normalize(U, axis=1)
            V = normalize(V, axis=1)
        elif dim_red == "precomputed":
            print("———————————————————————————————————" * 2)  # anchors.py:974
            print("1. get U,V")
            U, V = self._prepare_matrix(i, j, key_anchor=key_anchor)
            print(f"U.shape:{U.shape}")
            print(f"V.shape:{V.shape}")
        elif dim_red in ("lsi", "lsi-cca", "rpca", "rlsi"):
            raise NotImplementedError(
                f"dim_red={dim_red!r}: only the 'cca'/'pca' and 'precomputed' branches are provided")
        else:
            raise ValueError(f"Dimension reduction method {dim_red} is not supported.")

        rule = "———————————————————————————————————" * 2
        if dim_red == "precomputed":  # the reference rules this step off in that branch only (anchors.py:980)
            print(rule)
        print("4. find MNN of U and V to find anchors")
        _k = max(_temp for _temp in [k_anchor, k_local, k_score] if _temp is not None)
        _k = min(min_sample - 2, _k)
        print(f"Find Anchors using k={_k}")
        G11, G12, G21, G22, raw_anchors = self._calculate_mutual_knn_and_raw_anchors(
            i=i, j=j, U=U, V=V, k=_k, k_anchor=k_anchor, k_score=k_score)

        filtered = False
        if dim_red != "precomputed":  # announced whether or not the filter runs (anchors.py:894)
            print("5. filter anchors by high dimensional neighbors")
        if k_filter is not None and high_dim_feature is not None:
            if dim_red == "precomputed":  # ... and only with the filter here (anchors.py:989-992)
                print(rule)
                print("5. filter anchors by high dimensional neighbors")
            if self.n_cells[i] >= self.n_cells[j]:
                raw_anchors = filter_anchor(anchor=raw_anchors, adata_ref=adata1, adata_qry=adata2,
                                            scale_ref=scale1, scale_qry=scale2,
                                            high_dim_feature=high_dim_feature, k_filter=k_filter)
            else:
                raw_anchors = filter_anchor(anchor=raw_anchors[:, ::-1], adata_ref=adata2, adata_qry=adata1,
                                            scale_ref=scale2, scale_qry=scale1,
                                            high_dim_feature=high_dim_feature, k_filter=k_filter)[:, ::-1]
            filtered = True

        print("6. score anchors with snn and local structure preservation")
        scores, _ = self.device_scores[(i, j)]
        if filtered or k_local:
            self.device_scores.pop((i, j), None)
            anchor_df = score_anchor(anchor=raw_anchors, G11=G11, G12=G12, G21=G21, G22=G22, k_score=k_score,
                                     k_local=k_local, Gp1=self.local_knn[i], Gp2=self.local_knn[j])
        else:
            # the integer counts were produced on the device together with the MNN pairs
            anchor_df = pd.DataFrame(raw_anchors, columns=["x1", "x2"])
            anchor_df["score"] = scores.score
        self.U = U
        self.V = V
        return anchor_df

    def find_nearest_anchor(self, data, data_qry, ref, qry, key_correct="X_pca", npc=30, k_weight=100, sd=1,
                            random_state=0):
        """For every query cell its ``k_weight`` nearest anchors (among the query-side anchor
        cells) and their Gaussian-kernel weights (anchors.py:1189-1241).  Returns
        ``(anchor, G, D, cum_qry)`` like the reference."""
        print("Initialize")
        # row offsets of the data sets inside the concatenated reference / query matrices
        cum_ref = [0] + [int(v) for v in np.cumsum([data[r].shape[0] for r in ref])]
        cum_qry = [0] + [int(v) for v in np.cumsum([data[q].shape[0] for q in qry])]

        frames = []
        for ri, r in enumerate(ref):
            for qi, q in enumerate(qry):
                stored = self.anchor[(min(r, q), max(r, q))].copy()
                if r > q:  # stored the other way round: swap the two index columns
                    stored[["x1", "x2"]] = stored[["x2", "x1"]]
                stored["x1"] += cum_ref[ri]
                stored["x2"] += cum_qry[qi]
                frames.append(stored)
        table = pd.concat(frames)
        score = table["score"].values
        anchor = table[["x1", "x2"]].values

        if key_correct == "X":
            from sklearn.decomposition import PCA
            reduce_qry = PCA(n_components=npc, svd_solver="arpack", random_state=random_state).fit_transform(data_qry)
        else:
            reduce_qry = data_qry[:, :npc]

        print("Find nearest anchors", end=". ")
        index = NNDescent(reduce_qry[anchor[:, 1]], metric="euclidean", n_neighbors=k_weight,
                          random_state=random_state)
        k_weight = max(5, min(k_weight, anchor.shape[0] - 5))
        print("k_weight: ", k_weight, end="\n")
        G, D = index.query(reduce_qry, k=k_weight)

        print("Normalize graph")
        farthest = D[:, -1]
        all_on_cell = farthest == 0                     # every listed anchor sits on the cell itself
        W = (1 - D / farthest[:, None]) * score[G]      # closeness relative to the farthest, times anchor score
        W[all_on_cell] = score[G[all_on_cell]]
        W = 1 - np.exp(-W * (sd ** 2) / 4)
        W = W / (np.sum(W, axis=1) + 1e-6)[:, None]
        return anchor, G, W, cum_qry

    def find_anchor(self, adata_list, adata_names=None, k_local=None, key_local="X_pca", key_anchor="X",
                    dim_red="pca", svd_algorithm="randomized", scale1=True, scale2=True, scale_list=None,
                    k_filter=None, n_features=200, n_components=None, max_cc_cells=50000, chunk_size=50000,
                    k_anchor=5, k_score=30, alignments=None, random_state=0, signorm=True, key_match=None):
        """Find anchors for each dataset pair (anchors.py:1039-1187)."""
        if dim_red not in VALID_DIM_RED:
            raise ValueError(f"Dimension reduction method {dim_red} is not supported.")
        if key_match is not None:
            raise NotImplementedError("per-group anchor search (key_match) is not provided")
        if adata_names is None:
            adata_names = list(range(len(adata_list)))
        if len(adata_names) != len(adata_list):
            print("length of adata_names does not match length of adata_list.")

        self.adata_dict = {k: v for k, v in zip(adata_names, adata_list)}
        self.n_dataset = len(adata_list)
        self.n_cells = [adata.shape[0] for adata in adata_list]
        self.k_local = k_local
        self.key_local = key_local
        self._calculate_local_knn()
        self.alignments = alignments
        self._get_all_pairs()

        print("Find anchors across datasets.")
        for i in range(self.n_dataset - 1):
            for j in range(i + 1, self.n_dataset):
                if scale_list is not None:
                    scale1, scale2 = scale_list[i], scale_list[j]
                    print("Get scale1 and scale2 from scale_list")
                    print(f"dataset {i} scale: {scale1}")
                    print(f"dataset {j} scale: {scale2}")
                anchor_df = self._pairwise_find_anchor(
                    i=i, i_sel=None, j=j, j_sel=None, dim_red=dim_red, key_anchor=key_anchor,
                    svd_algorithm=svd_algorithm, scale1=scale1, scale2=scale2, k_anchor=k_anchor,
                    k_local=k_local, k_score=k_score, ncc=n_components, max_cc_cell=max_cc_cells,
                    k_filter=k_filter, n_features=n_features, chunk_size=chunk_size,
                    random_state=random_state, signorm=signorm)
                self.anchor[(i, j)] = anchor_df.copy()
        return
