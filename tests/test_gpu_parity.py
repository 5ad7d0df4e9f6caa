"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
golden vectors produced by the reference's own code.  Needs an MI355X.

Bar: neighbour indices, anchor pairs and integer scores bit-exact; float64
``score`` bit-exact; distances identical to the oracle's float32 value (and
within 1e-5 relative of a float64 evaluation)."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle
from scmiac_amd import _hip, anchor_search
from scmiac_amd.preprocessing import anchors as A
from scmiac_amd.modeling import scmiac as M


def _dev(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to("cuda", dtype)


def _mixture(seed, n, d, C=8, unit=False, offset=False):
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(1000 + d).normal(scale=3.0, size=(C, d))
    X = centres[rng.integers(0, C, n)] + rng.normal(size=(n, d))
    if offset:
        X = X + np.random.default_rng(2000 + d).normal(scale=0.25, size=d)
    if unit:
        X = X / np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(np.float32)


def _check_knn(Q, R, k, **kw):
    st = _hip.knn_search(_dev(Q), _dev(R), k, **kw)
    torch.cuda.synchronize()
    idx_o, d2_o = oracle.knn_dist2(Q, R, k)
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)  # same float32 arithmetic
    np.testing.assert_array_equal(st.dist.cpu().numpy(), np.sqrt(d2_o, dtype=np.float32))
    return st


@pytest.mark.parametrize("case", ["a", "b", "c", "dup", "tiny"])
def test_find_neighbor_golden(golden, case):
    f = golden["f3_find_neighbor"]
    k = int(f[f"{case}_k"])
    G = A.find_neighbor(f[f"{case}_U"], f[f"{case}_V"], k)
    for name, g in zip(("G11", "G12", "G21", "G22"), G):
        assert g.dtype == np.int32 and g.shape == f[f"{case}_{name}"].shape
        np.testing.assert_array_equal(g, f[f"{case}_{name}"])


@pytest.mark.parametrize("n,m,d,k", [
    (1, 1, 1, 1), (5, 3, 2, 3), (300, 70, 16, 31), (257, 513, 30, 30), (1000, 2049, 49, 31),
    (1500, 1500, 50, 30), (700, 900, 64, 64), (513, 777, 100, 20), (260, 300, 128, 10),
    (100, 40, 7, 50),  # k > m: padded with -1 / inf
])
def test_knn_matches_oracle_shapes(n, m, d, k):
    Q = _mixture(n + d, n, d)
    R = _mixture(m + 7 * d, m, d, offset=True)
    _check_knn(Q, R, k)


def test_knn_unit_norm_and_self():
    X = _mixture(5, 3000, 30, unit=True)
    st = _check_knn(X, X, 31)
    # the self table's first column is the row itself (no duplicates in this set)
    np.testing.assert_array_equal(st.idx[:, 0].cpu().numpy(), np.arange(3000))


def test_knn_duplicates_use_direct_evaluation():
    """Exact duplicates tie at the shortlist edge: the certificate must fail for
    those rows and the direct-evaluation pass must restore (D, index) order."""
    R = _mixture(9, 2000, 30)
    R[100:400] = R[7]          # 300 copies: more than the 64-entry shortlist holds
    Q = np.concatenate([R[:50], _mixture(10, 200, 30)])
    st = _check_knn(Q, R, 31)
    assert int(st.n_uncertified.item()) > 0


def test_knn_many_splits_and_single_split_agree():
    Q = _mixture(21, 600, 50)
    R = _mixture(22, 5000, 50, offset=True)
    a = _check_knn(Q, R, 30, splits=1)
    b = _check_knn(Q, R, 30, splits=7)
    assert torch.equal(a.idx, b.idx)


def test_knn_merge_over_reference_blocks():
    """Searching reference blocks one after the other (ring / tiled mode) gives
    the top-k of the union."""
    Q = _mixture(31, 500, 50)
    R = _mixture(32, 3000, 50, offset=True)
    idx_o, d2_o = oracle.knn_dist2(Q, R, 30)
    q = _dev(Q)
    st = None
    for lo, hi in ((0, 1100), (1100, 1900), (1900, 3000)):
        st = _hip.knn_search(q, _dev(R[lo:hi]), 30, col_offset=lo, state=st, exact_fallback=False)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)
    kth = st.dist2[:, -1].double()
    assert bool((kth < st.lower_bound).all())  # every row certified without the fallback


def test_knn_exact_rows_alone():
    Q = _mixture(41, 64, 30)
    R = _mixture(42, 1500, 30)
    st = _hip.knn_search(_dev(Q), _dev(R[:200]), 31, exact_fallback=False)
    rows = torch.arange(64, dtype=torch.int32, device="cuda")
    n_rows = torch.tensor([64], dtype=torch.int32, device="cuda")
    _hip.knn_exact_rows(_dev(Q), _dev(R[200:]), 31, rows, n_rows, st, col_offset=200)
    torch.cuda.synchronize()
    idx_o, d2_o = oracle.knn_dist2(Q, R, 31)
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)


@pytest.mark.parametrize("scratch_rows", [0, 1, 64])
def test_knn_exact_rows_scratch_modes(scratch_rows):
    """Without scratch every listed row is scanned by one block; with scratch the first
    rows are scanned by many blocks each (scratch_rows=1 covers 51 of the 64 rows here,
    the rest take the one-block path).  Same result either way."""
    Q = _mixture(43, 64, 50)
    R = _mixture(44, 5300, 50)
    st = _hip.knn_search(_dev(Q), _dev(R[:300]), 30, exact_fallback=False)
    rows = torch.arange(64, dtype=torch.int32, device="cuda")
    n_rows = torch.tensor([64], dtype=torch.int32, device="cuda")
    _hip.knn_exact_rows(_dev(Q), _dev(R[300:]), 30, rows, n_rows, st, col_offset=300, scratch_rows=scratch_rows)
    torch.cuda.synchronize()
    idx_o, d2_o = oracle.knn_dist2(Q, R, 30)
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)


@pytest.fixture
def tuning():
    """Tuning knobs of the fused kernel, reset after the test."""
    try:
        yield _hip.set_tuning
    finally:
        _hip.set_tuning()


def test_knn_tail_region(tuning):
    """The query blocks of the grid's last round search the references in several
    chunks each.  Forced at a small size by pretending the GPU runs 8 blocks at once."""
    tuning(slots=8, tail_min_rows=1024)
    Q = _mixture(61, 5000, 50)              # 14 query blocks of 384: 8 + a tail of 6
    R = _mixture(62, 20000, 50, offset=True)
    R[5000:5100] = R[3]                      # ties across tail chunks -> direct evaluation too
    assert _hip.knn_grid_plan(3300, 20000, 50, 31)["St"] > 1
    _check_knn(Q, R, 30)
    _check_knn(Q[:3300], R, 31)             # 9 blocks: tail of 1


def test_knn_distances_within_tolerance_of_float64():
    Q = _mixture(51, 800, 50)
    R = _mixture(52, 4000, 50, offset=True)
    st = _hip.knn_search(_dev(Q), _dev(R), 30)
    idx = st.idx.cpu().numpy()
    ref = np.sqrt(((Q[:, None, :].astype(np.float64) - R[idx].astype(np.float64)) ** 2).sum(-1))
    np.testing.assert_allclose(st.dist.cpu().numpy(), ref, rtol=1e-5)  # north-star tolerance


def test_knn_rejects_bad_arguments():
    q = torch.zeros((4, 8), device="cuda")
    with pytest.raises(ValueError):
        _hip.knn_search(q, q, 65)
    with pytest.raises(ValueError):
        _hip.knn_search(torch.zeros((4, 257), device="cuda"), torch.zeros((4, 257), device="cuda"), 3)
    with pytest.raises(ValueError):
        _hip.knn_search(q, torch.zeros((4, 9), device="cuda"), 3)


def test_find_mnn_golden(golden):
    f = golden["f1_find_mnn"]
    for ka in (1, 5, 30):
        got = A.find_mnn(f["G12"], f["G21"], ka)
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, f[f"pairs_ka{ka}"])
    empty = A.find_mnn(f["G12_empty"], f["G21_empty"], 3)
    assert empty.shape == (0,)
    with pytest.raises(IndexError):
        A.find_mnn(f["G12"], f["G21"], 31)
    bad = f["G12"].copy()
    bad[3, 0] = 10_000
    with pytest.raises(IndexError):
        A.find_mnn(bad, f["G21"], 5)


def test_find_mnn_large_random_matches_oracle():
    rng = np.random.default_rng(3)
    n1, n2, k = 40_000, 35_000, 30
    G12 = rng.integers(0, n2, size=(n1, k), dtype=np.int32)
    G21 = rng.integers(0, n1, size=(n2, k), dtype=np.int32)
    # plant mutual pairs so the result is not tiny
    rows = rng.integers(0, n1, 20_000)
    G21[G12[rows, 2], 1] = rows
    np.testing.assert_array_equal(A.find_mnn(G12, G21, 5), oracle.find_mnn(G12, G21, 5))


def test_score_anchor_golden(golden):
    f = golden["f2_score_anchor"]
    for ks in (30, 10):
        df = A.score_anchor(f["pairs"], f["G11"], f["G12"], f["G21"], f["G22"], k_score=ks, k_local=None)
        assert list(df.columns) == ["x1", "x2", "score"] and df["score"].dtype == np.float64
        np.testing.assert_array_equal(df["score"].values, f[f"score_ks{ks}"])
        raw = A._snn_raw(f["pairs"], f["G11"], f["G12"], f["G21"], f["G22"], ks)
        np.testing.assert_array_equal(np.asarray(raw, dtype=np.int32), f[f"raw_ks{ks}"])
    with pytest.raises(ValueError):
        A.score_anchor(np.array([]), f["G11"], f["G12"], f["G21"], f["G22"], k_local=None)


class _FakeAnnData:
    def __init__(self, X, obsm=None, obs=None):
        self.X = np.asarray(X)
        self.shape = self.X.shape
        self.obsm = dict(obsm or {})
        self.obs = obs
        self.var = type("V", (), {"index": pd.Index([f"g{i}" for i in range(self.X.shape[1])])})()


def _run_precomputed(f):
    a1 = _FakeAnnData(np.zeros((f["pre_U"].shape[0], 3)), obsm={"lat": f["pre_U"]})
    a2 = _FakeAnnData(np.zeros((f["pre_V"].shape[0], 3)), obsm={"lat": f["pre_V"]})
    integ = A.SeuratIntegration()
    integ.find_anchor([a1, a2], k_local=None, key_anchor="lat", dim_red="precomputed", k_anchor=5,
                      k_score=30, k_filter=None, alignments=[[[0], [1]]])
    return integ


def test_find_anchor_precomputed_golden(golden, capsys):
    f = golden["f4_find_anchor"]
    integ = _run_precomputed(f)
    adf = integ.anchor[(0, 1)]
    np.testing.assert_array_equal(adf["x1"].values, f["pre_x1"])
    np.testing.assert_array_equal(adf["x2"].values, f["pre_x2"])
    np.testing.assert_array_equal(adf["score"].values, f["pre_score"])
    assert adf["x1"].dtype == np.int64 and adf["score"].dtype == np.float64
    np.testing.assert_array_equal(integ.raw_anchor[(0, 1)], f["pre_raw"])
    for name, g in zip(("G11", "G12", "G21", "G22"), integ.mutual_knn[(0, 1)]):
        np.testing.assert_array_equal(g, f[f"pre_{name}"])
    assert "Find Anchors using k=30" in capsys.readouterr().out


def test_score_anchor_local_branch_golden(golden):
    """``score_anchor`` with ``k_local`` (anchors.py:670-678): the three score columns of the
    reference, bit for bit."""
    f = golden["f2_score_anchor"]
    df = A.score_anchor(f["pairs"], f["G11"], f["G12"], f["G21"], f["G22"], k_score=30,
                        Gp1=f["Gp1"], Gp2=f["Gp2"], k_local=20)
    assert list(df.columns) == ["x1", "x2", "score", "score_local1", "score_local2"]
    np.testing.assert_array_equal(df["score_local1"].values, f["local_1"])
    np.testing.assert_array_equal(df["score_local2"].values, f["local_2"])
    np.testing.assert_array_equal(df["score"].values, f["local_score"])


def test_find_nearest_anchor_golden(golden):
    """100 nearest anchors per query cell and their weights (anchors.py:1189-1241) against
    the reference's own method run on the exact stand-in index."""
    f = golden["f4_find_anchor"]
    integ = _run_precomputed(f)
    anchor, G, D, cum = integ.find_nearest_anchor([f["pre_U"], f["pre_V"]], f["pre_V"], [0], [1],
                                                  key_correct="lat", npc=30, k_weight=100)
    np.testing.assert_array_equal(anchor, f["nearest_anchor"])
    np.testing.assert_array_equal(G, f["nearest_G"])
    np.testing.assert_array_equal(D, f["nearest_D"])   # float64 arithmetic on bit-equal float32 distances
    assert list(cum) == list(f["nearest_cum"])


@pytest.mark.parametrize("ct_filter,tag", [(True, "filt"), (False, "nofilt")])
def test_label_filter_and_csv_golden(golden, ct_filter, tag):
    f = golden["f4_find_anchor"]
    integ = _run_precomputed(f)
    df, n_pass = M.label_filter(integ, f["pre_lab1"], f["pre_lab2"], ct_filter=ct_filter)
    assert list(df.columns) == ["x1", "x2", "score", "x1_ct", "x2_ct", "is_same"]
    np.testing.assert_array_equal(df["x1"].values, f[f"{tag}_x1"])
    np.testing.assert_array_equal(df["x2"].values, f[f"{tag}_x2"])
    np.testing.assert_array_equal(df["score"].values, f[f"{tag}_score"])
    np.testing.assert_array_equal(df["is_same"].values, f[f"{tag}_is_same"])
    np.testing.assert_array_equal(df.index.values, f[f"{tag}_index"])
    assert n_pass == len(f["nofilt_x1"])
    assert df.to_csv(index=False) == str(f[f"{tag}_csv"])  # anchors.csv text, byte for byte


def test_find_anchor_cca_branch_golden(golden):
    """Whole "cca" branch: CCA on the device (matrix-free randomized SVD, float64) feeding
    the GPU search; U within 1e-10 of the reference's (float64 sums taken in another
    order), anchors and scores equal to the reference's."""
    f = golden["f4_find_anchor"]
    integ = A.SeuratIntegration()
    integ.find_anchor([_FakeAnnData(f["cca_X1"]), _FakeAnnData(f["cca_X2"])], k_local=None, key_anchor="X",
                      dim_red="cca", max_cc_cells=50000, k_anchor=5, k_score=30, scale1=True, scale2=True,
                      n_components=20, k_filter=None, alignments=[[[0], [1]]])
    np.testing.assert_allclose(integ.U, f["cca_U"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(integ.V, f["cca_V"], rtol=0, atol=1e-10)
    adf = integ.anchor[(0, 1)]
    np.testing.assert_array_equal(adf["x1"].values, f["cca_x1"])
    np.testing.assert_array_equal(adf["x2"].values, f["cca_x2"])
    np.testing.assert_array_equal(adf["score"].values, f["cca_score"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_cca_device_golden(golden, tag):
    """``cca`` on the device against the reference's own ``cca`` (fixture F5): all cells
    decomposed in both orientations, and the sub-sampled cases with chunked projection.
    Tolerance 1e-10 absolute on entries of magnitude <= 0.3 (float64 rounding only)."""
    f = golden["f5_cca"]
    nc, cap, chunk, kf, scale2 = (int(v) for v in f[f"{tag}_args"])
    U, V, hdf = A.cca(f[f"{tag}_X1"], f[f"{tag}_X2"], scale1=True, scale2=bool(scale2), n_components=nc,
                      max_cc_cell=cap, chunk_size=chunk, random_state=0, k_filter=kf or None)
    assert U.dtype == np.float64 and U.shape == f[f"{tag}_U"].shape and V.shape == f[f"{tag}_V"].shape
    np.testing.assert_allclose(U, f[f"{tag}_U"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(V, f[f"{tag}_V"], rtol=0, atol=1e-10)
    if kf:
        np.testing.assert_array_equal(hdf, f[f"{tag}_hdf"])
    else:
        assert hdf is None


def test_cca_device_sparse_and_large_against_oracle():
    """scipy-sparse inputs and more cells than the decomposition keeps, against the CPU
    restatement (which forms the dense product as the reference does)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    g, n1, n2 = 300, 6000, 5000
    prof = rng.gamma(2.0, 1.0, size=(8, g))
    X1 = np.log1p(rng.poisson(prof[rng.integers(0, 8, n1)] * 0.6)).astype(np.float64)
    X2 = np.log1p(rng.poisson(prof[rng.integers(0, 8, n2)] * 0.4)).astype(np.float64)
    want = oracle.cca(X1, X2, n_components=30, max_cc_cell=2000, chunk_size=1500)
    got = A.cca(sp.csr_matrix(X1), sp.csr_matrix(X2), n_components=30, max_cc_cell=2000, chunk_size=1500)
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got[1], want[1], rtol=0, atol=1e-9)
    with pytest.raises(NotImplementedError):
        A.cca(X1, X2, svd_algorithm="arpack")


def test_rare_label_atlas_against_oracle():
    """200 labels, 50 of them rare (0.1 % each), 10 % label noise on the ATAC side:
    the filtered anchor table must equal the oracle's, rare labels included."""
    rng = np.random.default_rng(3003)
    C, d, n1, n2 = 200, 50, 12_000, 11_000
    prior = np.r_[np.full(50, 0.001), (1.0 / np.arange(1, 151)) / (1.0 / np.arange(1, 151)).sum() * 0.95]
    centres = rng.normal(scale=3.0, size=(C, d))
    l1, l2 = rng.choice(C, n1, p=prior), rng.choice(C, n2, p=prior)
    U = (centres[l1] + rng.normal(size=(n1, d))).astype(np.float32)
    V = (centres[l2] + rng.normal(scale=0.25, size=d) + rng.normal(size=(n2, d))).astype(np.float32)
    noisy = l2.copy()
    flip = rng.random(n2) < 0.10
    noisy[flip] = (noisy[flip] + rng.integers(1, C, flip.sum())) % C

    odf, _, _ = oracle.pairwise_anchors(U, V, k_anchor=5, k_score=30)
    want = oracle.label_filter(odf, l1, noisy, ct_filter=True)

    tables = anchor_search.find_neighbor_device(_dev(U), _dev(V), 30)
    scores, n_pairs = anchor_search.mnn_and_score(tables, 5, 30)
    c1, c2 = anchor_search.label_codes(l1, noisy)
    got = anchor_search.filter_anchors(scores, n_pairs, c1, c2, ct_filter=True)
    np.testing.assert_array_equal(got.x1, want["x1"].values)
    np.testing.assert_array_equal(got.x2, want["x2"].values)
    np.testing.assert_array_equal(got.score, want["score"].values)
    np.testing.assert_array_equal(got.index, want.index.values)
    rare_kept = np.isin(l1[got.x1], np.arange(50)).sum()
    assert rare_kept == np.isin(l1[want["x1"].values], np.arange(50)).sum()


def test_large_search_sampled_against_oracle():
    """100k x 100k, d = 50 (the single-GPU benchmark shape): 2 000 sampled rows
    checked bit-exactly, plus size-independent properties on all rows."""
    n = m = 100_000
    g = torch.Generator(device="cuda").manual_seed(1001)
    centres = torch.randn((50, 50), device="cuda", generator=g) * 3
    Q = centres[torch.randint(0, 50, (n,), device="cuda", generator=g)] + torch.randn((n, 50), device="cuda", generator=g)
    R = centres[torch.randint(0, 50, (m,), device="cuda", generator=g)] + torch.randn((m, 50), device="cuda", generator=g) + 0.1
    st = _hip.knn_search(Q, R, 30)
    torch.cuda.synchronize()
    d2 = st.dist2
    assert bool((d2[:, 1:] >= d2[:, :-1]).all())                      # rows sorted
    assert bool((st.idx >= 0).all()) and bool((st.idx < m).all())
    srt = torch.sort(st.idx, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                    # no reference twice in a row
    rows = np.random.default_rng(5).choice(n, 2000, replace=False)
    idx_o, d2_o = oracle.knn_dist2(Q[rows].cpu().numpy(), R.cpu().numpy(), 30)
    np.testing.assert_array_equal(st.idx[rows].cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2[rows].cpu().numpy(), d2_o)
    # the same search again gives the same table: a read of matrix-core results that
    # overtook its wait states would show up as run-to-run differences long before it
    # showed up in a 2 000-row sample
    first = st.idx.clone()
    for _ in range(2):
        again = _hip.knn_search(Q, R, 30)
        assert torch.equal(again.idx, first)


def test_ring_search_unit_norm_d64_single_rank(tmp_path):
    """Ring mode with the real kernels (one rank: the transport is covered by the gloo
    tests): L2-normalised d=64 latents, duplicates that force the direct-evaluation trip."""
    import torch.distributed as dist
    from scmiac_amd import sharded

    Q = _mixture(71, 2000, 64, unit=True)
    R = _mixture(72, 30000, 64, unit=True, offset=True)
    R[1000:1200] = R[5]  # 200 copies: ties beyond the shortlist
    Q[:20] = R[5]
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        st = sharded.knn_ring(_dev(Q), _dev(R), 30)
        torch.cuda.synchronize()
        full = sharded.gather_rows(st.idx, len(Q))
        # replicated mode: the four tables through the sharded driver == the single-process ones
        U, V = _dev(_mixture(73, 700, 64, unit=True)), _dev(_mixture(74, 900, 64, unit=True))
        a = sharded.find_neighbor_sharded(U, V, 30)
        b = anchor_search.find_neighbor_device(U, V, 30)
        for x, y in zip((a.G11, a.G12, a.G21, a.G22), (b.G11, b.G12, b.G21, b.G22)):
            assert torch.equal(x, y)
        # the search sharded by rows of the first set (only G21 / G22 gathered) == the single-process one
        pairs, raw = sharded.anchors_sharded(U, V, 30)
        want = anchor_search.mnn_and_score(b, k_anchor=5, k_score=30)[0]
        assert torch.equal(pairs, want.pairs[:want.n]) and torch.equal(raw, want.raw[:want.n]) and want.n > 0
        # closer-than-match counts through the sharded driver == the oracle's
        fx, fy = sharded.foscttm_sharded(U, _dev(_mixture(75, 700, 64, unit=True)))
        want_x, want_y = oracle.foscttm(U.cpu().numpy(), _mixture(75, 700, 64, unit=True))
        np.testing.assert_array_equal(fx, want_x)
        np.testing.assert_array_equal(fy, want_y)
    finally:
        dist.destroy_process_group()
    idx_o, d2_o = oracle.knn_dist2(Q, R, 30)
    np.testing.assert_array_equal(full.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)
    assert int(st.n_uncertified.item()) > 0


@pytest.mark.parametrize("variant", ["12x256x16x1", "12x128x32x0", "12x128x32x1"])
def test_knn_kernel_variants_agree_with_oracle(tuning, variant):
    """Other compiled shapes of the fused kernel (short pending buffers that refill within
    one subtile, buffers in LDS, 128-row stages) give the same exact result."""
    tuning(variant=variant)
    Q = _mixture(81, 1500, 50)
    R = _mixture(82, 12000, 50, offset=True)
    Q[:40] = R[:40]  # queries sitting on references: many small scores early in the stream
    _check_knn(Q, R, 31)


def test_knn_empty_query_and_empty_reference():
    """No query rows: nothing to do.  No reference rows: every slot is padding (-1 / inf),
    as for k > m."""
    R = _mixture(91, 50, 20)
    st = _hip.knn_search(torch.zeros((0, 20), device="cuda"), _dev(R), 5)
    assert st.idx.shape == (0, 5) and int(st.n_uncertified.item()) == 0
    Q = _mixture(92, 70, 20)
    st = _hip.knn_search(_dev(Q), torch.zeros((0, 20), device="cuda"), 5)
    torch.cuda.synchronize()
    assert bool((st.idx == -1).all()) and bool(torch.isinf(st.dist2).all())
    assert int(st.n_uncertified.item()) == 0


def test_knn_non_finite_rows_do_not_disturb_the_rest():
    """NaN / infinite rows are outside the contract, but they must neither hang the search
    nor change the neighbours of finite query rows among finite reference rows."""
    Q = _mixture(93, 300, 30)
    R = _mixture(94, 2000, 30, offset=True)
    bad = [17, 900]
    Rb = R.copy()
    Rb[17] = np.nan
    Rb[900, 3] = np.inf
    Qb = Q.copy()
    Qb[5] = np.nan
    st = _hip.knn_search(_dev(Qb), _dev(Rb), 10)
    torch.cuda.synchronize()
    keep = np.setdiff1d(np.arange(2000), bad)
    idx_o, d2_o = oracle.knn_dist2(Q, R[keep], 10)
    rows = np.setdiff1d(np.arange(300), [5])
    np.testing.assert_array_equal(st.idx.cpu().numpy()[rows], keep[idx_o][rows])
    np.testing.assert_array_equal(st.dist2.cpu().numpy()[rows], d2_o[rows])


def test_lu_row_permutation_on_the_device():
    """``P @ L`` of a tall block's pivoted LU, the normaliser of the randomized range finder
    behind ``cca`` (anchors.py:196-198 -> TruncatedSVD): permutation built by
    ``scmiac_lu_row_perm`` on the device against scipy's."""
    import scipy.linalg
    from scmiac_amd.cca_device import _lu_rows

    rng = np.random.default_rng(17)
    for n, r in ((5000, 40), (64, 64), (300, 1)):
        y = rng.normal(size=(n, r))
        y[::7] *= 50.0                     # pivots far from the diagonal
        got = _lu_rows(torch.from_numpy(y).cuda()).cpu().numpy()
        want = scipy.linalg.lu(y, permute_l=True)[0]
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    # more pivots than rows are refused before the launch; an entry outside 1..n is skipped and reported
    piv = torch.tensor([3, 99, 2], dtype=torch.int32, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    perm = _hip.lu_row_perm(piv, 4, status).cpu().numpy()
    np.testing.assert_array_equal(perm, [2, 0, 1, 3])
    assert int(status.item()) == 1
    status.zero_()
    _hip.lu_row_perm(torch.tensor([3, 2, 3], dtype=torch.int32, device="cuda"), 300000, status)
    assert int(status.item()) == 0
    from scmiac_amd.cca_device import randomized_svd_of_product
    bad = torch.full((500, 40), float("nan"), dtype=torch.float64, device="cuda")
    with pytest.raises((ValueError, RuntimeError)):  # refused pivots, or the factorisation's own complaint about the NaNs
        randomized_svd_of_product(bad, bad.clone(), 5)
    with pytest.raises(ValueError):
        _hip.lu_row_perm(torch.ones(5, dtype=torch.int32, device="cuda"), 3)
