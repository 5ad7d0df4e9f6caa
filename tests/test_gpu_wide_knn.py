"""More than 64 neighbours per row (the counts ``filter_anchor`` and ``find_nearest_anchor``
ask for, anchors.py:616-624, 1222-1233): merged group searches against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle
from scmiac_amd.preprocessing import anchors as A
from scmiac_amd.wide_knn import knn_search_wide

pytestmark = pytest.mark.gpu


def _check(q, r, k, **kw):
    idx_o, d2_o = oracle.knn_dist2(q, r, k)
    idx, d2, dist = knn_search_wide(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), k, **kw)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(d2.cpu().numpy(), d2_o)
    np.testing.assert_array_equal(dist.cpu().numpy(), np.sqrt(d2_o.astype(np.float64)).astype(np.float32))


@pytest.mark.parametrize("n,m,d,k", [(700, 5000, 30, 100), (300, 9000, 50, 200), (50, 150, 8, 100), (40, 130, 5, 130)])
def test_wide_knn_matches_oracle(n, m, d, k):
    rng = np.random.default_rng(m + k)
    centres = rng.normal(size=(6, d)) * 3
    r = (centres[np.sort(rng.integers(0, 6, m))] + rng.normal(size=(m, d))).astype(np.float32)  # clustered by row number
    q = (centres[rng.integers(0, 6, n)] + rng.normal(size=(n, d))).astype(np.float32)
    r[10:40] = r[50:80]  # duplicated rows: ties go by row number
    _check(q, r, k)


def test_wide_knn_cut_short_groups_are_searched_again():
    """Two groups of 64 listed rows cannot hold 100 neighbours unless they split 50/50:
    nearly every row takes the finer regrouping, and the result is still exact."""
    rng = np.random.default_rng(4)
    r = rng.normal(size=(4000, 16)).astype(np.float32)
    q = rng.normal(size=(200, 16)).astype(np.float32)
    _check(q, r, 100, n_groups=2)
    with pytest.raises(ValueError):
        knn_search_wide(torch.from_numpy(q).cuda(), torch.from_numpy(r[:50]).cuda(), 60 + 4)


def test_wide_knn_groups_of_exactly_64_rows_are_complete():
    """Groups of exactly 64 rows end in a real 64th key, but their lists are the whole group:
    a list that ends before the k-th merged key is then no sign of a row left out."""
    rng = np.random.default_rng(21)
    n_groups, d = 8, 12
    m = 64 * n_groups
    # rows dealt round-robin: group g holds rows g, g + 8, ...; put group 0 entirely near the queries
    r = (rng.normal(size=(m, d)) + 20.0).astype(np.float32)
    r[0::n_groups] = rng.normal(size=(64, d)).astype(np.float32) * 0.1
    q = rng.normal(size=(30, d)).astype(np.float32) * 0.1
    _check(q, r, 100, n_groups=n_groups)     # group 0's 64 rows all rank before the 100th key
    _check(q, r[:2 * 64], 70, n_groups=2)    # reached by the regrouping of a two-group search as well


def test_index_object_serves_wide_queries():
    rng = np.random.default_rng(8)
    data = rng.normal(size=(3000, 30)).astype(np.float32)
    qry = rng.normal(size=(500, 30)).astype(np.float32)
    index = A.NNDescent(data, metric="euclidean", n_neighbors=100, random_state=0)
    G, D = index.query(qry, k=100)                       # find_nearest_anchor's call (anchors.py:1233)
    idx_o, d2_o = oracle.knn_dist2(qry, data, 100)
    np.testing.assert_array_equal(G, idx_o)
    assert G.dtype == np.int32 and D.dtype == np.float32 and D.shape == (500, 100)
    graph = index.neighbor_graph[0]
    np.testing.assert_array_equal(graph, oracle.knn_dist2(data, data, 100)[0])


@pytest.mark.parametrize("n,m,d,k", [(600, 7000, 129, 30), (450, 5000, 160, 30), (500, 6000, 192, 31), (400, 5000, 200, 30),
                                     (350, 4000, 224, 30), (300, 4000, 256, 64)])
def test_wide_feature_rows_match_oracle(n, m, d, k):
    """129..256 features per row (8 waves per block, 64-row stages)."""
    from scmiac_amd import _hip
    rng = np.random.default_rng(d)
    centres = rng.normal(size=(5, d)) * 2
    r = (centres[rng.integers(0, 5, m)] + rng.normal(size=(m, d))).astype(np.float32)
    q = (centres[rng.integers(0, 5, n)] + rng.normal(size=(n, d))).astype(np.float32)
    q[:10] = r[:10]
    idx_o, d2_o = oracle.knn_dist2(q, r, k)
    st = _hip.knn_search(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), k)
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)
    y = (q + 2.0 * rng.normal(size=q.shape)).astype(np.float32)
    np.testing.assert_array_equal(_hip.closer_count(torch.from_numpy(q).cuda(), torch.from_numpy(y).cuda()).cpu().numpy(),
                                  oracle.closer_count(q, y).astype(np.int64))


def test_filter_anchor_with_its_default_neighbour_count():
    """``filter_anchor`` (anchors.py:584-628) with k_filter=200 on a 230-gene feature set:
    anchors survive iff the reference cell is among the query cell's 200 nearest cells in
    the row-normalised feature space."""
    from sklearn.preprocessing import normalize

    class _Ad:
        def __init__(self, X):
            self.X = X

    rng = np.random.default_rng(12)
    genes, feats = 400, np.sort(rng.choice(400, 230, replace=False))
    prof = rng.gamma(2.0, 1.0, size=(6, genes))
    l1, l2 = rng.integers(0, 6, 1500), rng.integers(0, 6, 1300)
    X1 = np.log1p(rng.poisson(prof[l1])).astype(np.float64)
    X2 = np.log1p(rng.poisson(prof[l2])).astype(np.float64)
    anchors = np.stack([rng.integers(0, 1500, 4000), rng.integers(0, 1300, 4000)], axis=1)
    got = A.filter_anchor(anchors, adata_ref=_Ad(X1), adata_qry=_Ad(X2), high_dim_feature=feats, k_filter=200)
    ref_data, qry_data = normalize(X1[:, feats], axis=1), normalize(X2[:, feats], axis=1)
    G = oracle.knn(qry_data, ref_data, 200)[0]
    want = np.array([xx for xx in anchors if xx[0] in G[xx[1]]])
    np.testing.assert_array_equal(got, want)
    assert 0 < len(want) < len(anchors)
