"""Host-side checks that need no GPU: the C-ABI library loads and exports every
symbol the header declares, and the drop-in API validates arguments like the
reference does."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "scmiac_hip.h").read_text()
    return sorted(set(re.findall(r"SCMIAC_API\s+[\w\s\*]+?\b(scmiac_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from scmiac_amd import _hip

    declared = _declared_symbols()
    assert len(declared) >= 10
    lib = ctypes.CDLL(str(_hip.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/scmiac_hip.h but not exported"
    assert sorted(_hip.EXPORTED_SYMBOLS) == declared
    assert _hip.load_library().scmiac_version() >= 1000


def test_size_queries_and_argument_checks_without_gpu():
    from scmiac_amd import _hip

    lib = _hip.load_library()
    n = ctypes.c_size_t(0)
    assert lib.scmiac_knn_workspace_size(1000, 2000, 50, 30, 0, 0, ctypes.byref(n)) == 0
    assert n.value > 1000 * 64 * 8
    assert lib.scmiac_knn_workspace_size(10, 10, 500, 30, 0, 0, ctypes.byref(n)) == 1
    assert b"dimension" in lib.scmiac_last_error()
    assert lib.scmiac_knn_workspace_size(10, 10, 50, 65, 0, 0, ctypes.byref(n)) == 1
    # the search mode is an explicit argument: dense, pruned, nothing else
    assert lib.scmiac_knn_workspace_size(1000, 2000, 50, 30, 0, 2, ctypes.byref(n)) == 1
    assert b"SCMIAC_KNN_PRUNED" in lib.scmiac_last_error()
    dense, pruned = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.scmiac_knn_workspace_size(50000, 200000, 50, 30, 0, 0, ctypes.byref(dense)) == 0
    assert lib.scmiac_knn_workspace_size(50000, 200000, 50, 30, 0, 1, ctypes.byref(pruned)) == 0
    assert pruned.value > 50000 * 195 * 4 and pruned.value != dense.value   # (the query-to-pivot bounds of 195 cells)
    plan = _hip.knn_grid_plan(50000, 200000, 50, 30, mode="pruned")
    assert plan["S"] == 1 and plan["St"] == 1 and plan["grid_blocks"] == plan["query_blocks"]   # one block per query block
    assert _hip.knn_grid_plan(50000, 200000, 50, 30, mode="pruned")["prunes"] == 1
    assert _hip.knn_grid_plan(50000, 200000, 50, 30, mode="dense")["prunes"] == 0
    for d in (100, 200, 256):   # wide rows prune too (kernels of their own)
        assert _hip.knn_grid_plan(50000, 200000, d, 30, mode="pruned")["prunes"] == 1
        assert _hip.knn_kernel_name(50000, 200000, d, 30, mode="pruned").endswith(", 2>")
    assert _hip.knn_grid_plan(1000, 200000, 50, 30, mode="pruned")["prunes"] == 0      # too few queries for a partition: dense
    assert _hip.knn_grid_plan(50000, 200000, 100, 30)["cells"] == 195                # ... but partitioned
    assert _hip.knn_kernel_name(50000, 200000, 50, 30, mode="pruned") == "fused_knn_kernel<25, 12, 320, 64, true, 2>"
    # 131 query blocks on 256 compute units: a single round, whose streams are shared ...
    assert _hip.knn_kernel_name(50000, 200000, 50, 30) == "fused_knn_share_kernel<25, 12, 320, 64>"
    assert _hip.knn_share_info(50000, 200000, 50, 30)["lists"] == 2
    # ... several rounds: the plain dense kernel
    assert _hip.knn_kernel_name(500000, 200000, 50, 30) == "fused_knn_kernel<25, 12, 320, 64, true, 0>"
    assert _hip.knn_share_info(500000, 200000, 50, 30)["lists"] == 0
    assert lib.scmiac_knn_set_sharing(0, 1, 0, 0) == 1 and b"run of 1" in lib.scmiac_last_error()
    _hip.set_sharing(lists=-1)
    try:
        assert _hip.knn_kernel_name(50000, 200000, 50, 30) == "fused_knn_kernel<25, 12, 320, 64, true, 0>"
    finally:
        _hip.set_sharing()
    off = ctypes.c_size_t(0)
    assert lib.scmiac_knn_prune_stats_offset(50000, 200000, 50, 30, 0, ctypes.byref(off)) == 0 and 0 < off.value < pruned.value
    assert lib.scmiac_knn_prune_stats_offset(500, 2000, 50, 30, 0, ctypes.byref(off)) == 1
    assert lib.scmiac_compact_workspace_size(12345, ctypes.byref(n)) == 0 and n.value >= 2 * 12345
    # the widened entry points: sizes and argument errors need no device either
    assert lib.scmiac_knn_workspace_size(10, 10, 256, 30, 0, 0, ctypes.byref(n)) == 0
    assert lib.scmiac_knn_workspace_size(10, 10, 257, 30, 0, 0, ctypes.byref(n)) == 1
    assert lib.scmiac_closer_count_workspace_size(5000, 7000, 30, ctypes.byref(n)) == 0 and n.value > 7000 * 30 * 4
    assert lib.scmiac_closer_count(None, 10, None, 11, 30, None, None, None, None, 0, None) == 1
    assert b"equal row counts" in lib.scmiac_last_error()
    assert lib.scmiac_ntxent2_parts(256) == 1 and lib.scmiac_ntxent2_parts(257) == 2
    assert lib.scmiac_ntxent2_workspace_size(256, 30, ctypes.byref(n)) == 0 and n.value == 16 * 2 * 256 * 32 * 4
    assert lib.scmiac_ntxent2_forward(None, None, 8, 65, 0.5, None, None, None, None, 0, None) == 1
    assert b"d=65" in lib.scmiac_last_error()
    assert lib.scmiac_ntxent2_backward(None, None, 8, 30, 0.0, None, None, None, None, None, None, 0, None) == 1
    assert b"temperature" in lib.scmiac_last_error()


def test_no_cpu_fallback_without_device():
    import torch
    from scmiac_amd import _hip
    from scmiac_amd.preprocessing import anchors as A

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(_hip.ScmiacHipError):
        A.find_neighbor(np.zeros((10, 4), np.float32), np.zeros((10, 4), np.float32), 3)


def test_find_anchor_rejects_unknown_dim_red():
    from scmiac_amd.preprocessing.anchors import SeuratIntegration

    with pytest.raises(ValueError, match="is not supported"):
        SeuratIntegration().find_anchor([], dim_red="umap")


def test_find_anchors_mode_v():
    import pandas as pd
    from scmiac_amd import find_anchors

    class AD:
        def __init__(self, n):
            self.shape = (n, 5)
            self.obs = pd.DataFrame({"cell_type": [f"t{i % 3}" for i in range(n)]})

    df = find_anchors(AD(7), AD(7), mode="v")
    assert list(df.columns) == ["x1", "x2", "x1_ct"] and len(df) == 7
    with pytest.raises(ValueError, match="must be equal when mode='v'"):
        find_anchors(AD(7), AD(8), mode="v")


def test_min_max_is_the_reference_formula(golden):
    from scmiac_amd.preprocessing.anchors import min_max

    f = golden["f2_score_anchor"]
    np.testing.assert_array_equal(min_max([int(v) for v in f["minmax_in"]]), f["minmax_out"])


def test_label_codes_follow_string_equality():
    from scmiac_amd.anchor_search import label_codes

    a = np.array(["B", "T", 1, None], dtype=object)
    b = np.array(["T", "1", "None", "x"], dtype=object)
    c1, c2 = label_codes(a, b)
    eq = a.astype(str)[:, None] == b.astype(str)[None, :]
    np.testing.assert_array_equal(c1[:, None] == c2[None, :], eq)


def test_grid_plan(monkeypatch):
    """Main query blocks fill whole rounds of concurrent blocks; the rest (the tail)
    are cut into more reference chunks so that the last round is short and full."""
    from scmiac_amd import _hip

    # 256 compute units, one 12-wave block (384 queries) each: what the library assumes
    # when no GPU answers
    p = _hip.knn_grid_plan(1_000_000, 1_000_000, 50, 30)
    assert p["queries_per_block"] == 384 and p["query_blocks"] == 2605
    assert p["S"] == 1 and p["main_query_blocks"] == 2560 and p["St"] > 1
    assert p["grid_blocks"] == 2560 + 45 * p["St"]
    assert p["tail_chunk_rows"] * p["St"] >= 1_000_000
    # a whole number of rounds: nothing to cut
    p = _hip.knn_grid_plan(393216, 1_000_000, 50, 30)
    assert p["S"] == 1 and p["main_query_blocks"] == 1024 and p["grid_blocks"] == 1024
    # few query blocks: all of them are cut, so that the grid still fills the GPU
    p = _hip.knn_grid_plan(10_000, 1_000_000, 50, 30)
    assert p["main_query_blocks"] == 0 and p["grid_blocks"] == 27 * p["St"] >= 200
    # an explicit number of chunks is taken as given
    p = _hip.knn_grid_plan(3000, 20000, 50, 30, splits=7)
    assert p["S"] == 7 and p["main_query_blocks"] == 8 and p["grid_blocks"] == 56
    # test aid: a pretend 8-slot GPU makes small problems grow a tail
    _hip.set_tuning(slots=8, tail_min_rows=1024)
    try:
        p = _hip.knn_grid_plan(3300, 20000, 50, 30)   # 9 query blocks: 8 + a tail of 1
        assert p["main_query_blocks"] == 8 and p["St"] > 1 and p["grid_blocks"] == 8 + p["St"]
    finally:
        _hip.set_tuning()
    # the library reads no environment variables: a variable of that name changes nothing
    monkeypatch.setenv("SCMIAC_FUSED_SLOTS", "8")
    assert _hip.knn_grid_plan(3300, 20000, 50, 30)["main_query_blocks"] != 8
    # a kernel variant that is not compiled is rejected, not silently launched
    with pytest.raises(ValueError):
        _hip.set_tuning(variant="7x100x32x1")
    with pytest.raises(ValueError):
        _hip.set_tuning(slots=-1)


def test_host_tensors_are_rejected_before_any_launch(monkeypatch):
    """The kernels take device pointers: a tensor in host memory must raise in Python, with
    no call into the library (the reference's losses and metrics accept CPU tensors)."""
    import torch

    from scmiac_amd import _hip

    calls = []

    class _Recorder:
        def __getattr__(self, name):
            def fn(*a):
                calls.append(name)
                return 0
            return fn

    monkeypatch.setattr(_hip, "require_gpu", lambda: _Recorder())
    z = torch.randn(8, 4)
    i32 = torch.zeros((8, 4), dtype=torch.int32)
    pairs = torch.zeros((8, 2), dtype=torch.int64)
    one = torch.zeros(1, dtype=torch.int64)
    cases = [
        lambda: _hip.ntxent2_forward(z, z, 0.5),
        lambda: _hip.ntxent2_backward(z, z, 0.5, z[:, 0], z[:, 0], torch.ones(1)),
        lambda: _hip.closer_count(z, z),
        lambda: _hip.find_mnn(i32, i32, 2),
        lambda: _hip.snn_score(pairs, one, i32, i32, i32, i32, 3),
        lambda: _hip.anchor_filter(pairs, one, i32[:, 0], 0.0, 1.0, i32[:, 0], i32[:, 0]),
    ]
    for case in cases:
        with pytest.raises(ValueError, match="GPU device"):
            case()
    assert calls == []
    from scmiac_amd.modeling.loss import NTXentLoss2
    with pytest.raises(ValueError, match="GPU device"):
        NTXentLoss2(0.5)(z, z)
    assert calls == []


def test_process_group_wiring_is_opt_in():
    """The drop-in API only turns collective after ``use_process_group``, which needs an
    initialised process group; a plain process is untouched."""
    from scmiac_amd import sharded

    assert sharded.active_group() is None
    with pytest.raises(RuntimeError):
        sharded.use_process_group()
    sharded.use_process_group(enable=False)
    assert sharded.active_group() is None


def test_stream_sharing_plan_invariants_without_gpu():
    """Planning only: where a dense search shares its streams, every block of the grid is a stream,
    the kernel is the sharing one, the workspace grows by the helpers' shortlists (64 keys, 64
    pending entries and a threshold per query, `lists` times per stream) plus the queues, and
    switching sharing off gives the plain plan back.  Several rounds, pruned searches and more than
    64 features never share."""
    from scmiac_amd import _hip

    lib = _hip.load_library()
    shared = 0
    try:
        for n, m, d in ((100_000, 100_000, 50), (50_000, 200_000, 50), (40_000, 60_000, 30), (20_000, 1_000_000, 64),
                        (3_000, 40_000, 16), (1_000_000, 1_000_000, 50), (125_000, 1_000_000, 50), (10_412, 10_412, 30),
                        (100_000, 100_000, 100), (2_000, 500, 8)):
            _hip.set_sharing()
            plan, info = _hip.knn_grid_plan(n, m, d, 30), _hip.knn_share_info(n, m, d, 30)
            assert plan["main_query_blocks"] * plan["S"] + (plan["query_blocks"] - plan["main_query_blocks"]) * plan["St"] \
                == plan["grid_blocks"]
            on, off = ctypes.c_size_t(0), ctypes.c_size_t(0)
            assert lib.scmiac_knn_workspace_size(n, m, d, 30, 0, 0, ctypes.byref(on)) == 0
            _hip.set_sharing(lists=-1)
            assert lib.scmiac_knn_workspace_size(n, m, d, 30, 0, 0, ctypes.byref(off)) == 0
            assert _hip.knn_share_info(n, m, d, 30)["lists"] == 0
            assert _hip.knn_grid_plan(n, m, d, 30) == plan          # the grid itself does not depend on sharing
            assert _hip.knn_kernel_name(n, m, d, 30).startswith("fused_knn_kernel<")
            _hip.set_sharing()
            if info["lists"] == 0:
                assert on.value == off.value and info["streams"] == 0
                assert _hip.knn_kernel_name(n, m, d, 30).startswith("fused_knn_kernel<")
                continue
            shared += 1
            assert d <= 64 and plan["main_query_blocks"] * plan["S"] <= 256 and plan["grid_blocks"] <= 512
            assert info["streams"] == plan["grid_blocks"] and info["lists"] == 2 and info["run"] >= 2
            assert _hip.knn_kernel_name(n, m, d, 30).startswith("fused_knn_share_kernel<")
            rows = info["lists"] * info["streams"] * plan["queries_per_block"]
            extra = on.value - off.value
            assert rows * (64 * 8 + 64 * 8 + 4) <= extra <= rows * (64 * 8 + 64 * 8 + 4) + info["streams"] * (3 * 4 + 4 * 384 + 16 * 8) + 8192
            assert _hip.knn_share_info(n, m, d, 30, splits=2)["lists"] in (0, 2)
            assert _hip.knn_grid_plan(n, m, d, 30, mode="pruned")["prunes"] in (0, 1)
        assert shared >= 4
        assert lib.scmiac_knn_set_sharing(9, 0, 0, 0) == 1 and lib.scmiac_knn_set_sharing(-2, 0, 0, 0) == 1
        assert lib.scmiac_knn_set_sharing(0, 0, 0, 3) == 1
    finally:
        _hip.set_sharing()


def test_search_modes_without_gpu():
    """The mode names the host layer takes; "auto" is planned as the dense search (what it falls
    back to), and sizes at which a probe cannot pay never reach the GPU."""
    import torch
    from scmiac_amd import _hip
    assert _hip.SEARCH_MODES == ("dense", "pruned", "auto")
    previous = _hip.set_default_mode("auto")
    try:
        assert _hip._mode_index(None) == 0 and _hip._mode_index("pruned") == 1 and _hip._mode_index("dense") == 0
        assert _hip.knn_grid_plan(100000, 100000, 50, 30) == _hip.knn_grid_plan(100000, 100000, 50, 30, mode="dense")
    finally:
        assert _hip.set_default_mode(previous) == "auto"
    with pytest.raises(ValueError):
        _hip.set_default_mode("fast")
    for n, m, d, k in ((2000, 10 ** 6, 50, 30), (10 ** 6, 20000, 50, 30), (10 ** 5, 10 ** 5, 50, 30),
                       (10 ** 6, 10 ** 6, 50, 70)):
        assert _hip.choose_mode(torch.empty((n, d), device="meta"), torch.empty((m, d), device="meta"), k) == "dense"
