"""mode="auto" of the host layer: a probe search (a few clumps of queries, pruned mode) says what
share of the dense stream the pruned search would go through, and the search is run in the mode
that this favours.  Whatever it picks, the table is the oracle's."""
import numpy as np
import pytest
import torch

from oracle import oracle
from scmiac_amd import _hip

pytestmark = pytest.mark.gpu


@pytest.fixture
def small_probe(monkeypatch):
    """Let searches of test size be probed (the library's own limits keep the probe for searches
    that take tens of milliseconds)."""
    monkeypatch.setattr(_hip, "AUTO_MIN_PAIRS", 1e9)
    monkeypatch.setattr(_hip, "AUTO_MIN_QUERIES", 8192)
    yield
    _hip.set_default_mode("dense")


def _data(kind, seed, n, d):
    rng = np.random.default_rng(seed)
    if kind == "clustered":
        centres = np.random.default_rng(900 + d).normal(scale=3.0, size=(40, d))
        return (centres[rng.integers(0, 40, n)] + rng.normal(size=(n, d))).astype(np.float32)
    return rng.normal(size=(n, d)).astype(np.float32)


@pytest.mark.parametrize("kind,expect", [("clustered", "pruned"), ("structureless", "dense")])
def test_auto_picks_the_mode_the_data_favour(kind, expect, small_probe):
    n, m, d, k = 40000, 90000, 50, 30
    Q, R = _data(kind, 61, n, d), _data(kind, 62, m, d)
    q, r = torch.from_numpy(Q).cuda(), torch.from_numpy(R).cuda()
    share = _hip.knn_kept_fraction(q, r, k)
    assert (share < 0.3) if expect == "pruned" else (share > 0.9), share
    assert _hip.choose_mode(q, r, k) == expect
    st = _hip.knn_search(q, r, k, mode="auto")
    torch.cuda.synchronize()
    rows = np.random.default_rng(6).choice(n, 1200, replace=False)
    idx_o, d2_o = oracle.knn_dist2(Q[rows], R, k)
    np.testing.assert_array_equal(st.idx.cpu().numpy()[rows], idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy()[rows], d2_o)
    for mode in ("dense", "pruned"):
        other = _hip.knn_search(q, r, k, mode=mode)
        assert torch.equal(other.idx, st.idx) and torch.equal(other.dist2, st.dist2)
    # as the default of searches that do not name a mode, and over reference blocks merged in turn
    _hip.set_default_mode("auto")
    merged = None
    for lo, hi in ((0, 50000), (50000, m)):
        merged = _hip.knn_search(q, r[lo:hi], k, col_offset=lo, state=merged)
    assert torch.equal(merged.idx, st.idx) and torch.equal(merged.dist2, st.dist2)
    assert merged.auto_mode == _hip.choose_mode(q, r[:50000], k)     # the first block's choice, kept for the second


def test_small_searches_are_not_probed():
    q = torch.zeros((3000, 30), device="cuda")
    r = torch.zeros((50000, 30), device="cuda")
    assert _hip.choose_mode(q, r, 30) == "dense"
    assert _hip.choose_mode(torch.zeros((40000, 50), device="cuda"), torch.zeros((90000, 50), device="cuda"), 30) == "dense"
