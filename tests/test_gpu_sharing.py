"""Stream sharing of the dense search (fused_knn_share_kernel): when the main region of the grid
is a single round, blocks that have run out of work join the streams of blocks that have not and
fill shortlists of their own.  Whoever scores which stage, the table must be the oracle's and
identical to the table of the search that does not share."""
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
    X = centres[rng.integers(0, C, n)] + spread * rng.normal(size=(n, d))
    if unit:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(np.float32)


@pytest.fixture
def knobs():
    try:
        yield _hip.set_tuning, _hip.set_sharing
    finally:
        _hip.set_tuning()
        _hip.set_sharing()


def _check(Q, R, k, **kw):
    st = _hip.knn_search(_dev(Q), _dev(R), k, **kw)
    torch.cuda.synchronize()
    idx_o, d2_o = oracle.knn_dist2(Q, R, k)
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)
    return st


@pytest.mark.parametrize("d,unit", [(50, False), (30, True), (64, True), (16, False)])
def test_shared_streams_give_the_oracle_table(d, unit, knobs):
    """A pretend 8-block GPU (main region + tail), runs of two stages, streams joined as soon as one
    stage is left: with the partition and without, chunks, merged blocks, duplicates."""
    tuning, sharing = knobs
    Q = _clustered(31, 5000, d, 14, unit=unit)
    R = _clustered(32, 70000, d, 14, unit=unit)
    R[300:350] = R[9]
    Q[:6] = R[9]
    tuning(slots=8, tail_min_rows=1024)
    sharing(lists=2, run=2, min_left=1)
    info = _hip.knn_share_info(5000, 70000, d, 30)
    assert info["lists"] == 2 and info["run"] == 2 and info["streams"] >= 2
    assert _hip.knn_kernel_name(5000, 70000, d, 30).startswith("fused_knn_share_kernel<")
    st = _check(Q, R, 30)
    _check(Q[:3072], R, 31)                 # eight query blocks exactly: no tail
    _check(Q, R, 30, splits=3)
    sharing(lists=1, run=5, min_left=3, order=1)
    _check(Q[:4000], R, 30)
    sharing(lists=3, run=2, min_left=1, order=2)
    _check(Q[:4000], R, 31)
    sharing(lists=-1)                       # the same search without sharing: the same table
    assert _hip.knn_share_info(5000, 70000, d, 30)["lists"] == 0
    plain = _hip.knn_search(_dev(Q), _dev(R), 30)
    assert torch.equal(plain.idx, st.idx) and torch.equal(plain.dist2, st.dist2)
    # reference blocks merged one after the other
    sharing(lists=2, run=2, min_left=1)
    idx_o, d2_o = oracle.knn_dist2(Q, R, 30)
    q, st = _dev(Q), None
    cuts = (0, 30000, 70000)
    for lo, hi in zip(cuts, cuts[1:]):
        st = _hip.knn_search(q, _dev(R[lo:hi]), 30, col_offset=lo, state=st)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st.idx.cpu().numpy(), idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy(), d2_o)


def test_a_slow_stream_is_joined(knobs):
    """Caller-order streams (no partition) of which one is far slower than the rest: its queries
    sit at the origin and the references come nearer and nearer, so every row is a candidate for
    them, while the other blocks' queries meet theirs at the start.  The others finish early and
    must take over part of the slow stream; the table stays the oracle's."""
    tuning, sharing = knobs
    rng = np.random.default_rng(41)
    d, m = 50, 60000
    R = rng.normal(size=(m, d)).astype(np.float32)
    R *= np.linspace(6.0, 0.5, m, dtype=np.float32)[:, None]          # shrinking norms: nearer and nearer the origin
    Q = (R[rng.integers(0, 2000, 8 * 384)] + 0.01 * rng.normal(size=(8 * 384, d))).astype(np.float32)
    Q[:384] = 1e-3 * rng.normal(size=(384, d)).astype(np.float32)    # block 0: at the origin
    tuning(slots=8, tail_min_rows=1024, no_locality=True)
    sharing(lists=2, run=4, min_left=4)
    assert _hip.knn_grid_plan(len(Q), m, d, 30)["cells"] == 0
    _check(Q, R, 30, splits=1)
    stats = _hip.knn_share_info(len(Q), m, d, 30, splits=1, read_stats=True)
    assert stats["streams"] == 8
    assert stats["joined"] >= 1 and stats["helper_stages"] >= 4, stats


@pytest.mark.parametrize("order", [0, 1, 2])
def test_library_choice_at_a_single_round(order, knobs):
    """The library's own plan at 100 000 x 100 000: 256 whole streams and 75 tail pieces on the 256
    compute units of an MI355X -- more blocks than the GPU runs at once, so blocks end while
    others have not started yet -- shared by default, with the tail pieces first (the default) or
    last in the grid; rows of a seeded sample against the oracle, the whole table against the
    search that does not share."""
    tuning, sharing = knobs
    n, m, d = 100000, 100000, 50
    Q = _clustered(51, n, d, 30)
    R = _clustered(52, m, d, 30)
    sharing(order=order)
    info = _hip.knn_share_info(n, m, d, 30)
    assert info["lists"] == 2 and info["streams"] >= 256
    q, r = _dev(Q), _dev(R)
    st = _hip.knn_search(q, r, 30)
    torch.cuda.synchronize()
    stats = _hip.knn_share_info(n, m, d, 30, read_stats=True)
    assert stats["joined"] >= 1                      # 331 blocks of uneven length: somebody ran out of work early
    rows = np.random.default_rng(5).choice(n, 1500, replace=False)
    idx_o, d2_o = oracle.knn_dist2(Q[rows], R, 30)
    np.testing.assert_array_equal(st.idx.cpu().numpy()[rows], idx_o)
    np.testing.assert_array_equal(st.dist2.cpu().numpy()[rows], d2_o)
    again = _hip.knn_search(q, r, 30)               # who joins what differs from run to run; the table does not
    assert torch.equal(again.idx, st.idx) and torch.equal(again.dist2, st.dist2)
    sharing(lists=-1)
    plain = _hip.knn_search(q, r, 30)
    assert torch.equal(plain.idx, st.idx) and torch.equal(plain.dist2, st.dist2)
