"""ctypes binding of ``libscmiac_hip.so`` (C ABI in ``include/scmiac_hip.h``).

PyTorch supplies device memory and the stream; every computation on the anchor
path goes through the HIP kernels.  There is no CPU or eager-PyTorch fallback:
if the library is missing or no MI355X is visible, the calls raise.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SCMIAC_HIP_LIB", _HERE / "csrc" / "libscmiac_hip.so"))  # override: experiment builds

KNN_MAX_K = 64
KNN_MAX_D = 256
KNN_MODES = {"dense": 0, "pruned": 1}  # SCMIAC_KNN_DENSE / SCMIAC_KNN_PRUNED

_lib = None

_c = ctypes
_vp, _i64, _i32, _sz, _dbl = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t, _c.c_double

_SIGNATURES = {
    "scmiac_version": ([], _c.c_int),
    "scmiac_last_error": ([], _c.c_char_p),
    "scmiac_knn_workspace_size": ([_i64, _i64, _i32, _i32, _i32, _i32, _c.POINTER(_sz)], _c.c_int),
    "scmiac_knn_search": ([_vp, _i64, _vp, _i64, _i32, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp,
                           _vp, _i32, _vp, _sz, _vp], _c.c_int),
    "scmiac_knn_exact_rows": ([_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
                              _c.c_int),
    "scmiac_find_mnn": ([_vp, _i64, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp], _c.c_int),
    "scmiac_find_mnn_noalloc": ([_vp, _i64, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp], _c.c_int),
    "scmiac_compact_workspace_size": ([_i64, _c.POINTER(_sz)], _c.c_int),
    "scmiac_snn_score": ([_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp],
                         _c.c_int),
    "scmiac_anchor_filter": ([_vp, _vp, _i64, _vp, _dbl, _dbl, _dbl, _vp, _i64, _vp, _i64, _i32, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp, _sz, _vp], _c.c_int),
    "scmiac_knn_diag_region": ([_i64, _i64, _i32, _i32, _i32, _i32, _c.POINTER(_sz), _c.POINTER(_sz),
                                _c.POINTER(_c.c_int)], _c.c_int),
    "scmiac_knn_prune_stats_offset": ([_i64, _i64, _i32, _i32, _i32, _c.POINTER(_sz)], _c.c_int),
    "scmiac_knn_set_sharing": ([_i32] * 4, _c.c_int),
    "scmiac_knn_share_info": ([_i64, _i64, _i32, _i32, _i32, _c.POINTER(_i64), _c.POINTER(_sz)], _c.c_int),
    "scmiac_knn_grid_plan": ([_i64, _i64, _i32, _i32, _i32, _i32, _c.POINTER(_i64)], _c.c_int),
    "scmiac_knn_kernel_name": ([_i64, _i64, _i32, _i32, _i32, _i32, _c.c_char_p, _sz], _c.c_int),
    "scmiac_lu_row_perm": ([_vp, _i32, _i64, _vp, _vp, _vp], _c.c_int),
    "scmiac_knn_set_timing": ([_i32], _c.c_int),
    "scmiac_knn_set_tuning": ([_i32] * 13, _c.c_int),
    "scmiac_knn_fused_ms": ([_c.POINTER(_c.c_float), _c.POINTER(_c.c_int)], _c.c_int),
    "scmiac_closer_count_workspace_size": ([_i64, _i64, _i32, _c.POINTER(_sz)], _c.c_int),
    "scmiac_closer_count": ([_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp], _c.c_int),
    "scmiac_ntxent2_parts": ([_i64], _c.c_int),
    "scmiac_ntxent2_workspace_size": ([_i64, _i32, _c.POINTER(_sz)], _c.c_int),
    "scmiac_ntxent2_forward": ([_vp, _vp, _i64, _i32, _c.c_float, _vp, _vp, _vp, _vp, _sz, _vp], _c.c_int),
    "scmiac_ntxent2_backward": ([_vp, _vp, _i64, _i32, _c.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
                                _c.c_int),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class ScmiacHipError(RuntimeError):
    pass


def load_library():
    """Load the shared library (no GPU needed for loading itself)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ScmiacHipError(
                f"{LIB_PATH} is missing: build it with `make -C {LIB_PATH.parent}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = lib
    return _lib


def _check(rc: int):
    if rc != 0:
        msg = load_library().scmiac_last_error().decode("utf-8", "replace")
        raise (ValueError if rc == 1 else ScmiacHipError)(f"libscmiac_hip status {rc}: {msg}")


def require_gpu():
    if not torch.cuda.is_available():
        raise ScmiacHipError("no HIP device visible: the anchor search runs only on the GPU (no CPU fallback)")
    return load_library()


def _cuda_device(t, what):
    """Device of ``t``, which must be a tensor in GPU memory: the kernels take device
    pointers, and a host pointer handed to them is a GPU memory fault, not an exception."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        where = t.device if isinstance(t, torch.Tensor) else type(t).__name__
        raise ValueError(f"{what} must be a tensor on a GPU device (got {where}); there is no CPU path")
    return t.device


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(x, device):
    t = torch.as_tensor(x)
    if t.dtype != torch.float32 or t.device != device or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.float32).contiguous()
    return t


def _dev_i32(x, device):
    t = torch.as_tensor(x)
    if t.dtype != torch.int32 or t.device != device or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.int32).contiguous()
    return t


class KnnState:
    """Running result of a search over one or more reference blocks."""

    def __init__(self, n, k, device):
        self.idx = torch.empty((n, k), dtype=torch.int32, device=device)
        self.dist2 = torch.empty((n, k), dtype=torch.float32, device=device)
        self.dist = torch.empty((n, k), dtype=torch.float32, device=device)
        self.lower_bound = torch.empty((n,), dtype=torch.float64, device=device)
        self.uncertified_rows = torch.empty((max(n, 1),), dtype=torch.int32, device=device)
        self.n_uncertified = torch.zeros((1,), dtype=torch.int32, device=device)
        self.blocks = 0
        self.auto_mode = None  # what mode="auto" picked for the first reference block


_ws_cache = {}


def _workspace(nbytes, device):
    """One grow-only scratch tensor per device (the C ABI allocates nothing)."""
    ws = _ws_cache.get(device)
    if ws is None or ws.numel() < nbytes:
        _ws_cache[device] = None
        ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=device)
        _ws_cache[device] = ws
    return ws


_default_mode = "dense"
SEARCH_MODES = ("dense", "pruned", "auto")

# "auto": a probe search decides.  Sizes below which the probe (3 - 12 ms) is not worth its time or
# its clumps would not look like query blocks (the dense search is taken: at most ~35 ms), the
# probe's shape, and the share of the dense stream above which the pruned search no longer pays
# (its stages are all candidate-rich: ~1.4x the time of a dense stage; measured shares are
# below 0.15 on clustered data and above 0.95 on data without structure).
AUTO_MIN_QUERIES = 16384
AUTO_MIN_REFERENCES = 32768
AUTO_MIN_PAIRS = 4e10
AUTO_PROBE_GROUPS = 8
AUTO_PROBE_POOL = 65536
AUTO_KEPT_LIMIT = 0.6


def set_default_mode(mode):
    """``"dense"``, ``"pruned"`` or ``"auto"``: how searches that do not say so themselves go
    through the reference rows (see ``SCMIAC_KNN_PRUNED`` in the header; all give the same, exact
    tables; ``"auto"`` see :func:`choose_mode`).  Returns the previous setting."""
    global _default_mode
    if mode not in SEARCH_MODES:
        raise ValueError(f"mode must be one of {sorted(SEARCH_MODES)}")
    previous, _default_mode = _default_mode, mode
    return previous


def _mode_index(mode):
    mode = mode or _default_mode
    return KNN_MODES["dense" if mode == "auto" else mode]    # planning calls: auto is sized as the dense search


def knn_kept_fraction(q, r, k, *, groups=AUTO_PROBE_GROUPS, group_rows=None):
    """Share of the dense stream that the pruned search of ``q`` in ``r`` would go through,
    measured on a probe: ``groups`` clumps of queries (evenly spaced query rows, each with the
    ``group_rows`` - 1 queries nearest to it: what a query block of the search looks like after
    the locality partition; ``group_rows`` is the number of queries in a block of the pruned
    kernel for this width -- a clump per query block of the probe, or blocks would mix far-apart
    queries and their thresholds say nothing) searched in pruned mode against all of ``r``, the
    reference rows cut into enough chunks to fill the GPU so that a probe on data that do not
    prune is over quickly.  Waits for the GPU."""
    n, m = q.shape[0], r.shape[0]
    groups = max(1, min(groups, n))
    if group_rows is None:
        group_rows = knn_grid_plan(n, m, q.shape[1], k, mode="pruned")["queries_per_block"]
    group_rows = max(1, min(group_rows, n))
    centres = q[torch.arange(groups, device=q.device) * (n // groups) + n // (2 * groups)]
    score = (q * q).sum(1)[None, :] - 2.0 * (centres @ q.T)                # |q - c|^2 - |c|^2, a row per centre
    # a radius that holds the clump with room to spare, read off at most 65 536 evenly spaced
    # queries, so that the exact selection runs over a few thousand rows instead of all
    pool = score[:, ::max(1, n // AUTO_PROBE_POOL)] if n > AUTO_PROBE_POOL else score
    expect = group_rows * pool.shape[1] / n
    radius = torch.kthvalue(pool, max(1, min(pool.shape[1], round(3 * expect + 48))), dim=1).values
    picks = []
    for g in range(groups):
        near = torch.nonzero(score[g] <= radius[g]).squeeze(1)
        if near.numel() < group_rows:
            near = torch.arange(n, device=q.device)
        picks.append(near[torch.topk(score[g][near], group_rows, largest=False, sorted=False).indices])
    probe = q[torch.cat(picks)].contiguous()
    splits = max(1, min(256 // groups, m // 8192))
    if not knn_grid_plan(probe.shape[0], m, q.shape[1], k, splits=splits, mode="pruned")["prunes"]:
        return 1.0  # a shape the pruned kernels do not serve: the search would be the dense one anyway
    knn_search(probe, r, k, mode="pruned", exact_fallback=False, splits=splits)
    kept, dense = knn_prune_stats(probe.shape[0], m, q.shape[1], k, splits=splits, device=q.device)
    return kept / dense if dense > 0 else 1.0


def choose_mode(q, r, k):
    """``"dense"`` or ``"pruned"`` for this search (device tensors): the dense search for sizes at
    which a probe would cost more than it can save (and for more neighbours than one search
    returns), else the pruned search if :func:`knn_kept_fraction` says that it goes through less than
    ``AUTO_KEPT_LIMIT`` of the stream.  Either answer gives the same table."""
    n, d = q.shape
    m = r.shape[0]
    if n < AUTO_MIN_QUERIES or m < AUTO_MIN_REFERENCES or float(n) * m < AUTO_MIN_PAIRS or k + 1 > 64:
        return "dense"
    return "pruned" if knn_kept_fraction(q, r, k) < AUTO_KEPT_LIMIT else "dense"


def knn_search(q, r, k, *, col_offset=0, state=None, splits=0, exact_fallback=True, mode=None):
    """Exact k-NN of the rows of ``q`` among the rows of ``r`` (device tensors).

    Returns a :class:`KnnState`; pass it back as ``state`` with another
    reference block (and that block's ``col_offset``) to extend the search.
    ``mode``: ``"dense"`` / ``"pruned"`` / ``"auto"`` (None: :func:`set_default_mode`'s, initially
    dense; ``"auto"`` runs a probe first and waits for it, see :func:`choose_mode`; when a search is
    extended block by block, the first block's choice holds for the rest).
    """
    lib = require_gpu()
    mode = mode or _default_mode
    if mode not in SEARCH_MODES:
        raise ValueError(f"mode must be one of {sorted(SEARCH_MODES)}")
    device = q.device if isinstance(q, torch.Tensor) and q.is_cuda else torch.device("cuda", torch.cuda.current_device())
    q = _dev_f32(q, device)
    r = _dev_f32(r, device)
    if q.dim() != 2 or r.dim() != 2 or q.shape[1] != r.shape[1]:
        raise ValueError("q and r must be 2-D with the same number of columns")
    if mode == "auto":  # a search over several reference blocks keeps what its first block was given
        mode = getattr(state, "auto_mode", None) or choose_mode(q, r, k)
        chosen = mode
    else:
        chosen = None
    mode_i = KNN_MODES[mode]
    n, d = q.shape
    m = r.shape[0]
    merge = 0 if state is None else 1
    if state is None:
        state = KnnState(n, k, device)
    elif state.idx.shape != (n, k):
        raise ValueError("state does not match (n, k)")
    nbytes = ctypes.c_size_t(0)
    _check(lib.scmiac_knn_workspace_size(n, m, d, k, splits, mode_i, ctypes.byref(nbytes)))
    ws = _workspace(nbytes.value, device)
    with torch.cuda.device(device):
        _check(lib.scmiac_knn_search(_ptr(q), n, _ptr(r), m, d, k, col_offset, merge, splits, mode_i, _ptr(state.idx),
                                     _ptr(state.dist2), _ptr(state.dist), _ptr(state.lower_bound),
                                     _ptr(state.uncertified_rows), _ptr(state.n_uncertified),
                                     1 if exact_fallback else 0, _ptr(ws), ws.numel(), _stream()))
    state.blocks += 1
    if chosen is not None:
        state.auto_mode = chosen
    return state


def knn_exact_rows(q, r, k, rows, n_rows, state, *, col_offset=0, scratch_rows=64):
    """Direct-evaluation merge of reference block ``r`` into ``state`` for the listed rows.
    ``scratch_rows`` listed rows get 128 KiB of scratch each and are scanned GPU-wide."""
    lib = require_gpu()
    device = _cuda_device(state.idx, "state")
    q = _dev_f32(q, device)
    r = _dev_f32(r, device)
    nbytes = int(scratch_rows) * 128 * 1024
    ws = _workspace(nbytes, device) if nbytes else None
    with torch.cuda.device(device):
        _check(lib.scmiac_knn_exact_rows(_ptr(q), q.shape[0], _ptr(r), r.shape[0], q.shape[1], k, col_offset,
                                         _ptr(rows), _ptr(n_rows), _ptr(state.idx), _ptr(state.dist2),
                                         _ptr(state.dist), _ptr(ws) if ws is not None else None, nbytes,
                                         _stream()))
    return state


def find_mnn(g12, g21, ka):
    """Device MNN search; returns ``(pairs int64 [n1*ka, 2], n_pairs int64 [1], status int32 [1])``."""
    lib = require_gpu()
    device = _cuda_device(g12, "g12")
    g12 = _dev_i32(g12, device)
    g21 = _dev_i32(g21, device)
    n1, ld12 = g12.shape
    n2, ld21 = g21.shape
    pairs = torch.empty((max(n1 * ka, 1), 2), dtype=torch.int64, device=device)
    n_pairs = torch.zeros((1,), dtype=torch.int64, device=device)
    status = torch.zeros((1,), dtype=torch.int32, device=device)
    nbytes = ctypes.c_size_t(0)
    _check(lib.scmiac_compact_workspace_size(n1 * ka, ctypes.byref(nbytes)))
    ws = _workspace(nbytes.value, device)
    with torch.cuda.device(device):
        _check(lib.scmiac_find_mnn(_ptr(g12), n1, ld12, _ptr(g21), n2, ld21, ka, _ptr(pairs), _ptr(n_pairs),
                                   _ptr(status), _ptr(ws), ws.numel(), _stream()))
    return pairs, n_pairs, status


def snn_score(pairs, n_pairs, g11, g12, g21, g22, ks, *, n_pairs_max=None, want_hist=False):
    """Raw shared-neighbour scores (int32 [n_pairs_max]) of the first ``n_pairs`` rows of ``pairs``."""
    lib = require_gpu()
    device = _cuda_device(pairs, "pairs")
    _cuda_device(n_pairs, "n_pairs")
    widths = {int(g.shape[1]) for g in (g11, g12, g21, g22)}
    if len(widths) != 1:  # differing widths: cut every table to its first ks columns
        g11, g12, g21, g22 = (g[:, :ks] for g in (g11, g12, g21, g22))
    tabs = [_dev_i32(g, device) for g in (g11, g12, g21, g22)]
    ld = tabs[0].shape[1]
    if ks > ld:
        raise IndexError("k_score exceeds the number of neighbour columns")
    if n_pairs_max is None:
        n_pairs_max = pairs.shape[0]
    score = torch.empty((max(n_pairs_max, 1),), dtype=torch.int32, device=device)
    hist = torch.zeros((2 * ks + 1,), dtype=torch.int64, device=device) if want_hist else None
    with torch.cuda.device(device):
        _check(lib.scmiac_snn_score(_ptr(pairs), _ptr(n_pairs), n_pairs_max, _ptr(tabs[0]), _ptr(tabs[1]),
                                    _ptr(tabs[2]), _ptr(tabs[3]), tabs[0].shape[0], tabs[2].shape[0], ld, ks,
                                    _ptr(score), _ptr(hist), _stream()))
    return score, hist


def anchor_filter(pairs, n_pairs, raw_score, p_lo, p_hi, codes1, codes2, *, threshold=0.2, ct_filter=True,
                  n_pairs_max=None, want_all_scores=False):
    """Normalised score + label filter with ordered compaction (device)."""
    lib = require_gpu()
    device = _cuda_device(pairs, "pairs")
    _cuda_device(n_pairs, "n_pairs")
    _cuda_device(raw_score, "raw_score")
    codes1 = _dev_i32(codes1, device)
    codes2 = _dev_i32(codes2, device)
    if n_pairs_max is None:
        n_pairs_max = pairs.shape[0]
    cap = max(n_pairs_max, 1)
    out_pairs = torch.empty((cap, 2), dtype=torch.int64, device=device)
    out_score = torch.empty((cap,), dtype=torch.float64, device=device)
    out_same = torch.empty((cap,), dtype=torch.uint8, device=device)
    out_index = torch.empty((cap,), dtype=torch.int64, device=device)
    counts = torch.zeros((2,), dtype=torch.int64, device=device)
    all_score = torch.empty((cap,), dtype=torch.float64, device=device) if want_all_scores else None
    nbytes = ctypes.c_size_t(0)
    _check(lib.scmiac_compact_workspace_size(n_pairs_max, ctypes.byref(nbytes)))
    ws = _workspace(nbytes.value, device)
    with torch.cuda.device(device):
        _check(lib.scmiac_anchor_filter(_ptr(pairs), _ptr(n_pairs), n_pairs_max, _ptr(raw_score), float(p_lo),
                                        float(p_hi), float(threshold), _ptr(codes1), codes1.shape[0],
                                        _ptr(codes2), codes2.shape[0], 1 if ct_filter else 0, _ptr(out_pairs),
                                        _ptr(out_score), _ptr(out_same), _ptr(out_index), _ptr(counts),
                                        _ptr(all_score), _ptr(ws), ws.numel(), _stream()))
    return out_pairs, out_score, out_same, out_index, counts, all_score


def knn_diag(n, m, d, k, splits=0, device=None, mode=None):
    """Per-wave counters of the last fused kernel (needs ``set_tuning(diag=True)`` at search time):
    int64 array [waves, 16]."""
    off, nbytes, waves = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_int(0)
    _check(load_library().scmiac_knn_diag_region(n, m, d, k, splits, _mode_index(mode),
                                                 ctypes.byref(off), ctypes.byref(nbytes), ctypes.byref(waves)))
    device = device or torch.device("cuda", torch.cuda.current_device())
    ws = _ws_cache[device]
    return ws[off.value:off.value + nbytes.value].view(torch.int64).view(waves.value, 16).cpu().numpy()


def knn_prune_stats(n, m, d, k, splits=0, device=None):
    """``(stages streamed, stages of the dense stream)`` of the last pruned search of this shape on
    ``device`` (read from the workspace it used)."""
    off = ctypes.c_size_t(0)
    _check(load_library().scmiac_knn_prune_stats_offset(n, m, d, k, splits, ctypes.byref(off)))
    device = device or torch.device("cuda", torch.cuda.current_device())
    kept, dense = _ws_cache[device][off.value:off.value + 16].view(torch.int64).cpu().tolist()
    return kept, dense


def set_sharing(*, lists=0, run=0, min_left=0, order=0):
    """Stream-sharing knobs of the dense search (``set_sharing()`` resets them): ``lists`` helper
    shortlists per stream (0 the library's choice, -1 never), ``run`` stages handed out at a time,
    ``min_left`` stages a stream must have left to be joined, ``order`` 1 / 2 the tail pieces first / last in the grid."""
    _check(load_library().scmiac_knn_set_sharing(int(lists), int(run), int(min_left), int(order)))


def knn_share_info(n, m, d, k, splits=0, device=None, read_stats=False):
    """How a dense search of this shape shares its streams: dict with ``lists`` (0: it does not),
    ``run``, ``min_left``, ``streams`` and, with ``read_stats``, ``joined`` / ``helper_stages`` of the
    last such search on ``device`` (read from the workspace it used)."""
    info = (_i64 * 4)()
    off = ctypes.c_size_t(0)
    _check(load_library().scmiac_knn_share_info(n, m, d, k, splits, info, ctypes.byref(off)))
    out = dict(zip(("lists", "run", "min_left", "streams"), (int(v) for v in info)))
    if read_stats and out["lists"] > 0:
        device = device or torch.device("cuda", torch.cuda.current_device())
        words = _ws_cache[device][off.value:off.value + 8 * (2 + 16 * out["streams"])].view(torch.int64).cpu()
        out["joined"], out["helper_stages"] = int(words[0]), int(words[1])
        # per grid block: start, end of its own stream, end (ticks of the 100 MHz clock), stages scored for
        # others, stages of its own stream it scored itself, first stream joined + 1, when, streams joined
        out["trace"] = words[2:].view(-1, 16).numpy()  # (+ the first four streams joined + 1, stages scored in each)
    return out


def _share_stats_offset(n, m, d, k, splits=0):
    info = (_i64 * 4)()
    off = ctypes.c_size_t(0)
    _check(load_library().scmiac_knn_share_info(n, m, d, k, splits, info, ctypes.byref(off)))
    return off.value


def knn_grid_plan(n, m, d, k, splits=0, mode=None):
    """Grid layout of the fused kernel for this problem (planning only: no kernel runs)."""
    plan = (_i64 * 10)()
    _check(load_library().scmiac_knn_grid_plan(n, m, d, k, splits, _mode_index(mode), plan))
    names = ("query_blocks", "queries_per_block", "S", "chunk_rows", "main_query_blocks", "St",
             "tail_chunk_rows", "grid_blocks", "cells", "prunes")
    return dict(zip(names, (int(v) for v in plan)))


def knn_kernel_name(n, m, d, k, splits=0, mode=None):
    """Instantiation of the fused kernel a search of this shape launches (planning only)."""
    buf = ctypes.create_string_buffer(128)
    _check(load_library().scmiac_knn_kernel_name(n, m, d, k, splits, _mode_index(mode), buf, len(buf)))
    return buf.value.decode()


def set_tuning(*, slots=0, tail_min_rows=0, no_tail=False, ablate=0, diag=False, variant=None, no_locality=False,
               home_rows=0, seed_rows=0, cell_rows=0):
    """Diagnostic / tuning knobs of the fused kernel's grid plan (``set_tuning()`` resets them).
    ``variant`` is ``"NWxRTxPCAPxGP"`` or a 4-tuple; an uncompiled one raises ``ValueError``."""
    nw = rt = pcap = gp = 0
    if variant:
        parts = [int(v) for v in variant.split("x")] if isinstance(variant, str) else [int(v) for v in variant]
        nw, rt = parts[0], parts[1]
        pcap = parts[2] if len(parts) > 2 else 32
        gp = parts[3] if len(parts) > 3 else 0
    _check(load_library().scmiac_knn_set_tuning(int(slots), int(tail_min_rows), int(bool(no_tail)), int(ablate),
                                                int(bool(diag)), nw, rt, pcap, gp, int(bool(no_locality)), int(home_rows),
                                                int(seed_rows), int(cell_rows)))


def set_timing(enabled: bool):
    _check(load_library().scmiac_knn_set_timing(1 if enabled else 0))


def fused_ms():
    """(summed fused-kernel milliseconds, launches) since timing was enabled."""
    ms = ctypes.c_float(0.0)
    cnt = ctypes.c_int(0)
    _check(load_library().scmiac_knn_fused_ms(ctypes.byref(ms), ctypes.byref(cnt)))
    return float(ms.value), int(cnt.value)


_ntxent_bytes = {}


def _ntxent_workspace(lib, batch, d, device):
    need = _ntxent_bytes.get((batch, d))
    if need is None:  # the loss runs every training step: ask once per shape
        nbytes = ctypes.c_size_t(0)
        _check(lib.scmiac_ntxent2_workspace_size(batch, d, ctypes.byref(nbytes)))
        need = _ntxent_bytes[(batch, d)] = max(nbytes.value, 256)
    return _workspace(need, device)


class _on_device:
    """``torch.cuda.device(device)`` without the switch when ``device`` is already current."""

    def __init__(self, device):
        self._ctx = None if device.index in (None, torch.cuda.current_device()) else torch.cuda.device(device)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)


def ntxent2_forward(zi, zj, temperature):
    """Symmetric NT-Xent loss of two float32 ``[batch, d]`` device tensors.

    Returns ``(loss_parts, lse_row, lse_col)``: the loss is ``loss_parts.sum()``; the
    two log-sum-exp vectors are what :func:`ntxent2_backward` needs."""
    lib = require_gpu()
    device = _cuda_device(zi, "zi")
    _cuda_device(zj, "zj")
    zi, zj = _dev_f32(zi, device), _dev_f32(zj, device)
    if zi.ndim != 2 or zi.shape != zj.shape:
        raise ValueError("zi and zj must be 2-D tensors of the same shape")
    batch, d = zi.shape
    parts = torch.empty((max(lib.scmiac_ntxent2_parts(batch), 1),), dtype=torch.float32, device=device)
    lse_row = torch.empty((batch,), dtype=torch.float32, device=device)
    lse_col = torch.empty((batch,), dtype=torch.float32, device=device)
    ws = _ntxent_workspace(lib, batch, d, device)
    with _on_device(device):
        _check(lib.scmiac_ntxent2_forward(_ptr(zi), _ptr(zj), batch, d, float(temperature), _ptr(lse_row),
                                          _ptr(lse_col), _ptr(parts), _ptr(ws), ws.numel(), _stream()))
    return parts, lse_row, lse_col


def ntxent2_backward(zi, zj, temperature, lse_row, lse_col, grad_out):
    """Gradients of the loss with respect to ``zi`` and ``zj`` (``grad_out``: one float32 on the device)."""
    lib = require_gpu()
    device = _cuda_device(zi, "zi")
    for name, t in (("zj", zj), ("lse_row", lse_row), ("lse_col", lse_col)):
        _cuda_device(t, name)
    zi, zj = _dev_f32(zi, device), _dev_f32(zj, device)
    grad_out = _dev_f32(grad_out, device).reshape(1)
    batch, d = zi.shape
    dzi, dzj = torch.empty_like(zi), torch.empty_like(zj)
    ws = _ntxent_workspace(lib, batch, d, device)
    with _on_device(device):
        _check(lib.scmiac_ntxent2_backward(_ptr(zi), _ptr(zj), batch, d, float(temperature), _ptr(lse_row),
                                           _ptr(lse_col), _ptr(grad_out), _ptr(dzi), _ptr(dzj), _ptr(ws),
                                           ws.numel(), _stream()))
    return dzi, dzj


def lu_row_perm(pivots, n, status=None):
    """Row permutation of a pivoted LU (``torch.linalg.lu_factor``'s 1-based int32 pivots of an
    ``n``-row block) as an int64 device tensor, without a trip to the host.  ``status`` (one int32
    on the device, zeroed by the caller) becomes 1 when an entry is no pivot of an ``n``-row block."""
    lib = require_gpu()
    device = _cuda_device(pivots, "pivots")
    pivots = _dev_i32(pivots, device)
    if status is not None:
        _cuda_device(status, "status")
    perm = torch.empty((n,), dtype=torch.int64, device=device)
    with _on_device(device):
        _check(lib.scmiac_lu_row_perm(_ptr(pivots), pivots.shape[0], n, _ptr(perm), _ptr(status), _stream()))
    return perm


def closer_count(q, r, match=None):
    """``counts[i] = #{j : D(q_i, r_j) < D(q_i, r_match(i))}`` as an int64 device tensor
    (``match=None``: row ``i`` of ``r``).  Raises ``IndexError`` for a match index outside ``r``."""
    lib = require_gpu()
    device = _cuda_device(q, "q")
    q, r = _dev_f32(q, device), _dev_f32(r, device)
    if q.ndim != 2 or r.ndim != 2 or q.shape[1] != r.shape[1]:
        raise ValueError("q and r must be 2-D with the same number of columns")
    n, d = q.shape
    m = r.shape[0]
    if match is None and n != m:
        raise ValueError("Shapes do not match!")
    if match is not None:
        match = _dev_i32(match, device)
        if match.shape != (n,):
            raise ValueError("match needs one entry per query row")
    counts = torch.zeros((max(n, 1),), dtype=torch.int32, device=device)
    status = torch.zeros((1,), dtype=torch.int32, device=device)
    if n:
        nbytes = ctypes.c_size_t(0)
        _check(lib.scmiac_closer_count_workspace_size(n, m, d, ctypes.byref(nbytes)))
        ws = _workspace(nbytes.value, device)
        with torch.cuda.device(device):
            _check(lib.scmiac_closer_count(_ptr(q), n, _ptr(r), m, d, _ptr(match), _ptr(counts), _ptr(status),
                                           _ptr(ws), ws.numel(), _stream()))
        if int(status.item()):
            raise IndexError("match index out of bounds")
    return counts[:n].to(torch.int64)
