"""Distance-based evaluation metrics on the GPU against the reference's own outputs
(fixture F7, made by evaluation/benchmark.py) and the CPU oracle.  The closer-than-match
counts are integers and must be equal; the k-NN based metrics are exact on data without
tied distances."""
import numpy as np
import pytest
import torch

from oracle import oracle
from scmiac_amd import _hip
from scmiac_amd.evaluation import foscttm, knn_matching, neighbor_conservation, partially_matched_foscttm

pytestmark = pytest.mark.gpu


def test_foscttm_matches_reference_fixture(golden):
    f = golden["f7_evaluation"]
    for a, b, wa, wb in (("x", "y", "fos_x", "fos_y"), ("xu", "yu", "fosu_x", "fosu_y")):
        fx, fy = foscttm(f[a], f[b])
        assert fx.dtype == np.float32 and fx.shape == (600,)
        np.testing.assert_array_equal(fx, f[wa])   # counts / n in float32, as .float().mean() gives
        np.testing.assert_array_equal(fy, f[wb])
    with pytest.raises(ValueError, match="Shapes do not match!"):
        foscttm(f["x"][:10], f["y"][:11])


def test_partially_matched_foscttm_matches_reference_fixture(golden, capsys):
    f = golden["f7_evaluation"]
    names = np.array([f"cell{i:04d}" for i in range(600)])
    px, py = f["part_px"], f["part_py"]
    fx, fy = partially_matched_foscttm(f["x"][px].astype(np.float64), f["y"][py].astype(np.float64),
                                       names[px], names[py])
    assert fx.dtype == np.float64
    np.testing.assert_array_equal(fx, f["part_fx"])
    np.testing.assert_array_equal(fy, f["part_fy"])
    assert "D_x.shape:(380, 450)" in capsys.readouterr().out
    with pytest.raises(ValueError, match="No matching barcodes"):
        partially_matched_foscttm(f["x"][:5], f["y"][:5], names[:5], names[5:10])


def test_knn_metrics_match_reference_fixture(golden):
    f = golden["f7_evaluation"]
    x, y = f["x_clean"], f["y_clean"]
    assert neighbor_conservation(x, y, n_neighbors=10) == pytest.approx(float(f["nc_10"]), abs=1e-12)
    assert neighbor_conservation(x, y, n_neighbors=15, exclude_top=1) == pytest.approx(float(f["nc_15_ex1"]), abs=1e-12)
    for k in (1, 5):
        res = knn_matching(x, y, k=k)
        got = [res["rna_to_atac_accuracy"], res["atac_to_rna_accuracy"], res["overall_accuracy"]]
        np.testing.assert_array_equal(got, f[f"match_k{k}"])


@pytest.mark.parametrize("n,m,d,noise", [(1, 1, 2, 0.0), (33, 33, 3, 0.5), (1000, 1000, 30, 3.0),
                                         (5000, 5000, 50, 10.0), (3000, 3000, 64, 0.2), (2000, 2000, 100, 5.0),
                                         (777, 777, 128, 1.0), (20000, 20000, 16, 2.0)])
def test_closer_count_equals_oracle(n, m, d, noise):
    """Every count is an integer over exact float32 distances: equality, also where the
    match is far away (noise 10: about half of all cells are closer) and thousands of scores
    fall into the band that has to be settled by the exact chain."""
    rng = np.random.default_rng(n + d)
    x = rng.normal(size=(n, d)).astype(np.float32)
    y = (x + noise * rng.normal(size=(n, d))).astype(np.float32)
    want = oracle.closer_count(x, y)
    got = _hip.closer_count(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.cpu().numpy(), want.astype(np.int64))
    if noise >= 3.0:
        assert want.mean() > 0.05 * m


def test_closer_count_unit_rows_duplicates_and_match_array():
    rng = np.random.default_rng(9)
    n, m, d = 4000, 6500, 30
    r = rng.normal(size=(m, d)).astype(np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    match = rng.integers(0, m, n).astype(np.int32)
    q = r[match] + 0.3 * rng.normal(size=(n, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = q.astype(np.float32)
    r[100:200] = r[300:400]            # duplicated references: ties must not count as closer
    q[5] = r[match[5]]                 # distance zero to the match: nothing is closer
    want = oracle.closer_count(q, r, match)
    got = _hip.closer_count(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), match=torch.from_numpy(match).cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), want.astype(np.int64))
    assert got[5].item() == 0
    bad = match.copy()
    bad[17] = m
    with pytest.raises(IndexError):
        _hip.closer_count(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), match=torch.from_numpy(bad).cuda())
    with pytest.raises(ValueError):
        _hip.closer_count(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda())   # n != m without match
    assert _hip.closer_count(torch.zeros((0, d), device="cuda"), torch.from_numpy(r).cuda(),
                             match=torch.zeros(0, dtype=torch.int32, device="cuda")).shape == (0,)


def test_closer_count_large_is_repeatable_and_sane():
    """200 000 x 200 000, d = 30: the grid has main and tail regions and several reference
    chunks per query block; counts of a row sample equal the oracle's, two runs agree."""
    rng = np.random.default_rng(77)
    n, d = 200000, 30
    centres = rng.normal(size=(40, d)).astype(np.float32) * 2
    lab = rng.integers(0, 40, n)
    x = centres[lab] + rng.normal(size=(n, d)).astype(np.float32)
    y = x + 1.5 * rng.normal(size=(n, d)).astype(np.float32)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    got = _hip.closer_count(xd, yd).cpu().numpy()
    again = _hip.closer_count(xd, yd).cpu().numpy()
    np.testing.assert_array_equal(got, again)
    rows = rng.choice(n, 300, replace=False).astype(np.int32)
    want = oracle.closer_count(x[rows], y, rows)
    np.testing.assert_array_equal(got[rows], want.astype(np.int64))
