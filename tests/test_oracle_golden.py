"""The CPU oracle against the golden vectors produced by the reference's own
code (oracle/gen_golden.py).  No GPU needed."""
import numpy as np
import pandas as pd
import pytest

from oracle import oracle


def test_find_mnn_matches_reference(golden):
    f = golden["f1_find_mnn"]
    for ka in (1, 5, 30):
        got = oracle.find_mnn(f["G12"], f["G21"], ka)
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, f[f"pairs_ka{ka}"])
    # loop restatement agrees with the C loop
    np.testing.assert_array_equal(oracle.find_mnn_loops(f["G12"], f["G21"], 5), f["pairs_ka5"])
    empty = oracle.find_mnn(f["G12_empty"], f["G21_empty"], 3)
    assert tuple(empty.shape) == tuple(f["pairs_empty_shape"]) == (0,)


def test_score_anchor_matches_reference(golden):
    f = golden["f2_score_anchor"]
    for ks in (30, 10):
        raw = oracle.snn_counts(f["pairs"], f["G11"], f["G12"], f["G21"], f["G22"], ks)
        np.testing.assert_array_equal(raw, f[f"raw_ks{ks}"])
        np.testing.assert_array_equal(
            raw, oracle.snn_counts_loops(f["pairs"], f["G11"], f["G12"], f["G21"], f["G22"], ks))
        df = oracle.score_anchor(f["pairs"], f["G11"], f["G12"], f["G21"], f["G22"],
                                 k_score=ks, k_local=None)
        assert list(df.columns) == ["x1", "x2", "score"]
        assert df["score"].dtype == np.float64
        np.testing.assert_array_equal(df["score"].values, f[f"score_ks{ks}"])  # bit-exact


def test_score_anchor_local_branch_matches_reference(golden):
    f = golden["f2_score_anchor"]
    df = oracle.score_anchor(f["pairs"], f["G11"], f["G12"], f["G21"], f["G22"], k_score=30,
                             Gp1=f["Gp1"], Gp2=f["Gp2"], k_local=20)
    np.testing.assert_array_equal(df["score_local1"].values, f["local_1"])
    np.testing.assert_array_equal(df["score_local2"].values, f["local_2"])
    np.testing.assert_array_equal(df["score"].values, f["local_score"])


def test_min_max_matches_reference(golden):
    f = golden["f2_score_anchor"]
    out = oracle.min_max([int(v) for v in f["minmax_in"]])
    np.testing.assert_array_equal(out, f["minmax_out"])
    with np.errstate(all="ignore"):
        const = oracle.min_max([7] * 50)
    assert np.isnan(const).all() and np.isnan(f["minmax_const_out"]).all()


@pytest.mark.parametrize("case", ["a", "b", "c", "dup", "tiny"])
def test_find_neighbor_matches_reference(golden, case):
    f = golden["f3_find_neighbor"]
    U, V, k = f[f"{case}_U"], f[f"{case}_V"], int(f[f"{case}_k"])
    G = oracle.find_neighbor(U, V, k)
    for name, g in zip(("G11", "G12", "G21", "G22"), G):
        assert g.dtype == np.int32 and g.shape[1] == k
        np.testing.assert_array_equal(g, f[f"{case}_{name}"])


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_exact_search_against_float64_bruteforce(golden, case):
    """Independent pin of the k-NN definition: a float64 brute force must give
    the same neighbours wherever its k-th/(k+1)-th gap exceeds float32 noise."""
    f = golden["f3_find_neighbor"]
    U, V, k = f[f"{case}_U"].astype(np.float64), f[f"{case}_V"].astype(np.float64), int(f[f"{case}_k"])
    d2 = (U * U).sum(1)[:, None] + (V * V).sum(1)[None, :] - 2.0 * U @ V.T
    order = np.argsort(d2, axis=1, kind="stable")[:, :k + 1]
    srt = np.take_along_axis(d2, order, axis=1)
    gaps = np.diff(srt, axis=1)
    safe = (gaps > 1e-5 * srt[:, 1:]).all(axis=1)
    assert safe.mean() > 0.9
    np.testing.assert_array_equal(f[f"{case}_G12"][safe], order[safe, :k])
    # reported distances agree with float64 to 1e-5 relative
    idx, dist = oracle.knn(f[f"{case}_U"], f[f"{case}_V"], k)
    np.testing.assert_allclose(dist, f[f"{case}_D12"], rtol=1e-5, atol=1e-7)


def test_pairwise_precomputed_matches_reference(golden):
    f = golden["f4_find_anchor"]
    df, G, raw = oracle.pairwise_anchors(f["pre_U"], f["pre_V"], k_anchor=5, k_score=30)
    for name, g in zip(("G11", "G12", "G21", "G22"), G):
        np.testing.assert_array_equal(g, f[f"pre_{name}"])
    np.testing.assert_array_equal(raw, f["pre_raw"])
    np.testing.assert_array_equal(df["x1"].values, f["pre_x1"])
    np.testing.assert_array_equal(df["x2"].values, f["pre_x2"])
    np.testing.assert_array_equal(df["score"].values, f["pre_score"])


@pytest.mark.parametrize("ct_filter,tag", [(True, "filt"), (False, "nofilt")])
def test_label_filter_matches_reference(golden, ct_filter, tag):
    f = golden["f4_find_anchor"]
    df = pd.DataFrame({"x1": f["pre_x1"], "x2": f["pre_x2"], "score": f["pre_score"]})
    out = oracle.label_filter(df, f["pre_lab1"], f["pre_lab2"], ct_filter=ct_filter)
    np.testing.assert_array_equal(out["x1"].values, f[f"{tag}_x1"])
    np.testing.assert_array_equal(out["x2"].values, f[f"{tag}_x2"])
    np.testing.assert_array_equal(out["score"].values, f[f"{tag}_score"])
    np.testing.assert_array_equal(out["is_same"].values, f[f"{tag}_is_same"])
    np.testing.assert_array_equal(out.index.values, f[f"{tag}_index"])
    assert out.to_csv(index=False) == str(f[f"{tag}_csv"])


def test_cca_branch_search_matches_reference(golden):
    """Steps 3-6 of the "cca" branch on the reference's own U, V (already
    row-normalised by the reference; normalising again must be a no-op up to
    rounding, so the search is run on them as stored)."""
    f = golden["f4_find_anchor"]
    df, _, _ = oracle.pairwise_anchors(f["cca_U"], f["cca_V"], k_anchor=5, k_score=30)
    np.testing.assert_array_equal(df["x1"].values, f["cca_x1"])
    np.testing.assert_array_equal(df["x2"].values, f["cca_x2"])
    np.testing.assert_array_equal(df["score"].values, f["cca_score"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_cca_restatement_matches_reference(golden, tag):
    """The oracle's ``cca`` against the reference's own (fixture F5): same scikit-learn
    calls on the same dense product, so the agreement is to the last few bits."""
    f = golden["f5_cca"]
    nc, cap, chunk, kf, scale2 = (int(v) for v in f[f"{tag}_args"])
    U, V, hdf = oracle.cca(f[f"{tag}_X1"], f[f"{tag}_X2"], scale1=True, scale2=bool(scale2),
                           n_components=nc, max_cc_cell=cap, chunk_size=chunk, k_filter=kf or None)
    np.testing.assert_allclose(U, f[f"{tag}_U"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(V, f[f"{tag}_V"], rtol=0, atol=1e-13)
    if kf:
        np.testing.assert_array_equal(hdf, f[f"{tag}_hdf"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_matrix_free_cca_host_logic(golden, tag, monkeypatch):
    """Host logic of the matrix-free CCA (thin products, LU row permutation, sign
    convention, sub-sampling, chunked projection) with torch's CPU float64 backend
    standing in for the device; the GPU run of the same code is a ``gpu`` test."""
    import torch
    from scmiac_amd import cca_device

    def row_perm_on_the_host(pivots, n, status=None):
        # stand-in for the device kernel: LAPACK's row interchanges applied in order
        perm = np.arange(n, dtype=np.int64)
        for i, p in enumerate(pivots.numpy() - 1):
            perm[i], perm[p] = perm[p], perm[i]
        return torch.from_numpy(perm)

    monkeypatch.setattr(cca_device, "_device", lambda: torch.device("cpu"))
    monkeypatch.setattr(cca_device._hip, "lu_row_perm", row_perm_on_the_host)
    f = golden["f5_cca"]
    nc, cap, chunk, kf, scale2 = (int(v) for v in f[f"{tag}_args"])
    U, V, hdf = cca_device.cca_device(f[f"{tag}_X1"], f[f"{tag}_X2"], scale1=True, scale2=bool(scale2),
                                      n_components=nc, max_cc_cell=cap, chunk_size=chunk,
                                      k_filter=kf or None)
    np.testing.assert_allclose(U, f[f"{tag}_U"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(V, f[f"{tag}_V"], rtol=0, atol=1e-10)
    if kf:
        np.testing.assert_array_equal(hdf, f[f"{tag}_hdf"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_ntxent_restatement_matches_reference(golden, tag):
    """The float64 restatement of the symmetric NT-Xent loss against the reference's
    module (fixture F6: float32 on the CPU, gradients of 300 x loss)."""
    f = golden["f6_ntxent"]
    loss, dzi, dzj = oracle.ntxent2(f[f"{tag}_zi"], f[f"{tag}_zj"], float(f[f"{tag}_T"]))
    np.testing.assert_allclose(loss, float(f[f"{tag}_loss"]), rtol=1e-6)
    for got, want in ((300 * dzi, f[f"{tag}_dzi"]), (300 * dzj, f[f"{tag}_dzj"])):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6 * float(np.abs(want).max()))


def test_evaluation_restatements_match_reference(golden):
    """FOSCTTM (both directions, raw and unit rows, partially matched), neighbour
    conservation and k-NN matching of the oracle against the reference's own
    evaluation/benchmark.py outputs (fixture F7)."""
    f = golden["f7_evaluation"]
    for a, b, wa, wb in (("x", "y", "fos_x", "fos_y"), ("xu", "yu", "fosu_x", "fosu_y")):
        fx, fy = oracle.foscttm(f[a], f[b])
        np.testing.assert_array_equal(fx, f[wa])
        np.testing.assert_array_equal(fy, f[wb])
    x, y = f["x_clean"], f["y_clean"]
    assert oracle.neighbor_conservation(x, y, 10) == pytest.approx(float(f["nc_10"]), abs=1e-12)
    assert oracle.neighbor_conservation(x, y, 15, 1) == pytest.approx(float(f["nc_15_ex1"]), abs=1e-12)
    for k in (1, 5):
        res = oracle.knn_matching(x, y, k)
        np.testing.assert_array_equal(
            [res["rna_to_atac_accuracy"], res["atac_to_rna_accuracy"], res["overall_accuracy"]], f[f"match_k{k}"])
    with pytest.raises(IndexError):
        oracle.closer_count(x[:4], y[:4], [0, 1, 2, 4])
