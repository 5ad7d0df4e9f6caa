"""The locality partition of the search (rows regrouped by nearest pivot, every query block
streaming its cell's reference rows first, thresholds seeded from those rows): the table must
be the oracle's whatever the row order and however the grid is cut, and identical to the
table of the plain caller-order stream."""
import numpy as np
import pytest
import torch

from oracle import oracle
from scmiac_amd import _hip

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _clustered(seed, n, d, C, spread=1.0, unit=False):
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(500 + d).normal(scale=3.0, size=(C, d))
    lab = rng.integers(0, C, n)
    X = centres[lab] + spread * rng.normal(size=(n, d))
    if unit:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(np.float32), lab


@pytest.fixture
def tuning():
    try:
        yield _hip.set_tuning
    finally:
        _hip.set_tuning()


def _check(Q, R, k, **kw):
    st = _hip.knn_search(_dev(Q), _dev(R), k, **kw)
    torch.cuda.synchronize()
    idx_o, d2_o = oracle.knn_dist2(Q, R, k)
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)
    return st


@pytest.mark.parametrize("order", ["random", "sorted by cluster", "sorted, queries reversed"])
def test_any_row_order_gives_the_oracle_table(order, tuning):
    Q, lq = _clustered(1, 6000, 50, 12)
    R, lr = _clustered(2, 40000, 50, 12)
    if order != "random":
        R = R[np.argsort(lr, kind="stable")]
        Q = Q[np.argsort(lq, kind="stable")]
    if order.endswith("reversed"):
        Q = Q[::-1].copy()
    st = _check(Q, R, 30)
    tuning(no_locality=True)
    plain = _hip.knn_search(_dev(Q), _dev(R), 30)
    assert torch.equal(plain.idx, st.idx) and torch.equal(plain.dist2, st.dist2)


def test_chunks_and_tail_with_home_rows_inside_and_outside(tuning):
    """A pretend 8-block GPU: query blocks searched in several reference chunks, most of which
    do not contain the block's home rows (their thresholds still come from the seed pass)."""
    Q, _ = _clustered(3, 5000, 30, 9, unit=True)
    R, _ = _clustered(4, 50000, 30, 9, unit=True)
    R[7000:7100] = R[11]      # 100 copies of one row: ties that outnumber a shortlist
    Q[:10] = R[11]
    tuning(slots=8, tail_min_rows=1024)
    plan = _hip.knn_grid_plan(3300, 50000, 30, 31)   # 9 query blocks: 8 + a tail of 1, cut into chunks
    assert plan["St"] > 1
    _check(Q[:3300], R, 31)
    _check(Q, R, 31)
    _check(Q, R, 30, splits=5)
    for knobs in (dict(cell_rows=300, home_rows=700, seed_rows=320), dict(cell_rows=6000, home_rows=20000, seed_rows=9000)):
        tuning(slots=8, tail_min_rows=1024, **knobs)
        _check(Q[:2500], R, 30)


def test_merge_over_blocks_starts_from_the_running_distances():
    """Reference blocks one after the other: later blocks start from thresholds derived from the
    running k-th distance; duplicates across blocks tie at exactly that distance."""
    Q, _ = _clustered(5, 4096, 64, 10, unit=True)
    R, _ = _clustered(6, 60000, 64, 10, unit=True)
    R[20000:20040] = R[3]     # block 2 repeats a row of block 1: equal distances, larger row numbers
    R[45000] = R[3]
    Q[:5] = R[3]
    idx_o, d2_o = oracle.knn_dist2(Q, R, 30)
    q, st = _dev(Q), None
    cuts = (0, 18000, 41000, 60000)
    for lo, hi in zip(cuts, cuts[1:]):
        st = _hip.knn_search(q, _dev(R[lo:hi]), 30, col_offset=lo, state=st, exact_fallback=False)
    n_unc = int(st.n_uncertified.item())
    rows = st.uncertified_rows[:max(n_unc, 1)].clone()
    n_rows = torch.tensor([n_unc], dtype=torch.int32, device="cuda")
    for lo, hi in zip(cuts, cuts[1:]):
        _hip.knn_exact_rows(q, _dev(R[lo:hi]), 30, rows, n_rows, st, col_offset=lo)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)
    # a block that holds nothing closer leaves the result alone and still certifies it
    far = (R[:20000] + 50.0).astype(np.float32)
    st2 = _hip.knn_search(q, _dev(far), 30, col_offset=60000, state=st, exact_fallback=False)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st2.idx.cpu().numpy()[5:], idx_o[5:])


def test_structureless_and_degenerate_inputs():
    rng = np.random.default_rng(7)
    Q = rng.normal(size=(3000, 20)).astype(np.float32)          # no clusters at all
    R = rng.normal(size=(20000, 20)).astype(np.float32)
    _check(Q, R, 30)
    R2 = np.repeat(R[:200], 100, axis=0)                          # 200 distinct rows, 100 copies each
    _check(Q[:2100], R2, 30)
    Q3 = np.zeros((2500, 8), dtype=np.float32)                    # every query the same point
    R3 = rng.normal(size=(17000, 8)).astype(np.float32)
    _check(Q3, R3, 10)


@pytest.mark.parametrize("d,unit", [(50, False), (30, True), (64, True)])
def test_pruned_search_stays_exact(d, unit, tuning):
    """``mode="pruned"`` (SCMIAC_KNN_PRUNED): farthest-point pivots, and every block streams only the
    cells within reach of its queries.  Well separated clusters (most cells pruned), overlapping
    ones, outliers, duplicates, chunks and a tail, more than 255 cells, and blocks merged one after
    the other: always the oracle's table, and the dense search's."""
    for spread, C in ((1.0, 40), (4.0, 12)):
        Q, _ = _clustered(11, 5000, d, C, spread=spread, unit=unit)
        R, _ = _clustered(12, 70000, d, C, spread=spread, unit=unit)
        assert _hip.knn_grid_plan(5000, 70000, d, 30, mode="pruned")["prunes"] == 1
        assert _hip.knn_kernel_name(5000, 70000, d, 30, mode="pruned").endswith(", 2>")
        if spread == 1.0 and not unit:      # separated clusters: most of the dense stream is skipped
            _check(Q, R, 30, mode="pruned")
            kept, dense = _hip.knn_prune_stats(5000, 70000, d, 30)
            assert 0 < kept < 0.5 * dense
        R[100:140] = R[5]                   # duplicates
        R[::997] *= 3.0                     # outliers far from every cluster (they inflate the cells' radii)
        Q[:7] = R[5]
        st = _check(Q, R, 30, mode="pruned")
        kept, dense = _hip.knn_prune_stats(5000, 70000, d, 30)
        assert 0 < kept <= dense
        plain = _hip.knn_search(_dev(Q), _dev(R), 30, mode="dense")
        assert torch.equal(plain.idx, st.idx) and torch.equal(plain.dist2, st.dist2)
        tuning(slots=8, tail_min_rows=1024)  # several chunks per query block and a tail region
        _check(Q[:3300], R, 31, mode="pruned")
        tuning(cell_rows=128)                # 546 cells: cell numbers beyond one byte
        assert _hip.knn_grid_plan(5000, 70000, d, 30, mode="pruned")["cells"] == 546
        _check(Q, R, 30, mode="pruned")
        tuning()
        idx_o, d2_o = oracle.knn_dist2(Q, R, 30)
        q, st = _dev(Q), None
        cuts = (0, 25000, 70000)
        for lo, hi in zip(cuts, cuts[1:]):
            st = _hip.knn_search(q, _dev(R[lo:hi]), 30, col_offset=lo, state=st, mode="pruned")
        torch.cuda.synchronize()
        np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
        np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)


def test_pruned_mode_on_structureless_data_and_where_it_does_not_apply():
    """No clusters: every cell stays within reach, the table is still the oracle's.  Shapes the
    pruned kernel does not serve (too few rows for a partition) run the dense search."""
    rng = np.random.default_rng(17)
    Q = rng.normal(size=(3000, 20)).astype(np.float32)
    R = rng.normal(size=(20000, 20)).astype(np.float32)
    _check(Q, R, 30, mode="pruned")
    kept, dense = _hip.knn_prune_stats(3000, 20000, 20, 30)
    assert kept > 0.9 * dense
    assert _hip.knn_grid_plan(500, 3000, 20, 30, mode="pruned")["prunes"] == 0
    _check(Q[:500], R[:3000], 30, mode="pruned")
    with pytest.raises(ValueError):
        _hip.knn_search(_dev(Q), _dev(R), 30, mode="sparse")


@pytest.mark.parametrize("d", [72, 100, 128, 160, 192, 200, 224, 256])
def test_pruned_search_serves_wide_rows(d):
    """Rows of more than 64 features (the widths of filter_anchor's searches among them): the pruned
    kernels of the wide variants skip most of the stream on clustered data and return the oracle's
    table, which is also the dense search's."""
    Q, _ = _clustered(23, 3000, d, 10)
    R, _ = _clustered(24, 40000, d, 10)
    R[700:730] = R[5]
    Q[:4] = R[5]
    assert _hip.knn_grid_plan(3000, 40000, d, 30, mode="pruned")["prunes"] == 1
    assert _hip.knn_kernel_name(3000, 40000, d, 30, mode="pruned").endswith(", 2>")
    st = _check(Q, R, 30, mode="pruned")
    kept, dense = _hip.knn_prune_stats(3000, 40000, d, 30)
    assert 0 < kept < 0.5 * dense, (kept, dense)
    plain = _hip.knn_search(_dev(Q), _dev(R), 30, mode="dense")
    assert torch.equal(plain.idx, st.idx) and torch.equal(plain.dist2, st.dist2)
    _check(Q[:2900], R, 31, mode="pruned", splits=3)
    rng = np.random.default_rng(d)          # no clusters: nothing to skip, the same table
    Qn = rng.normal(size=(2500, d)).astype(np.float32)
    Rn = rng.normal(size=(20000, d)).astype(np.float32)
    _check(Qn, Rn, 30, mode="pruned")


@pytest.mark.parametrize("d", [100, 128, 200, 256])
def test_partition_serves_wide_rows(d, tuning):
    """Beyond 64 features the partition kernels hold fewer rows per block and stream the pivots
    through LDS in tiles; the table stays the oracle's and the plain caller-order stream's."""
    Q, _ = _clustered(21, 2500, d, 9)
    R, _ = _clustered(22, 18000, d, 9)
    R[50:80] = R[3]
    assert _hip.knn_grid_plan(2500, 18000, d, 30)["cells"] > 0
    st = _check(Q, R, 30)
    tuning(no_locality=True)
    assert _hip.knn_grid_plan(2500, 18000, d, 30)["cells"] == 0
    plain = _hip.knn_search(_dev(Q), _dev(R), 30)
    assert torch.equal(plain.idx, st.idx) and torch.equal(plain.dist2, st.dist2)
