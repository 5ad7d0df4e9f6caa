"""Exact k-NN for more than 64 neighbours per row, composed from 64-neighbour searches.

``filter_anchor`` and ``find_nearest_anchor`` of the reference ask their index for 200 and
100 neighbours (``scmiac/preprocessing/anchors.py:616-624, 1222-1233``); the fused kernel
keeps 64 per row.  The reference rows are dealt round-robin into ``G`` groups, every group
is searched for its 64 nearest rows, and the group results are merged by ``(distance,
index)``.  The merge is exact for a query row if no group's list is cut short: every row a
group did not list comes after that group's 64th key, so the merged top-k is final when
each full group list ends at or after the k-th merged key.  Rows where that fails (more than
64 of the k nearest fell into one group) are searched again with twice as many groups;
groups of at most 64 rows are always complete, so this ends.
"""
from __future__ import annotations

import math

import torch

from . import _hip

_SENTINEL = torch.iinfo(torch.int64).max


def _group_keys(q, r_group, g, n_groups, backend):
    """int64 keys (float32 distance bits << 32 | global row) of q's 64 nearest rows in one group."""
    kk = min(_hip.KNN_MAX_K, r_group.shape[0])
    st = backend.knn_search(q, r_group, kk)
    rows = st.idx.to(torch.int64) * n_groups + g
    keys = (st.dist2.view(torch.int32).to(torch.int64) << 32) | rows
    keys = torch.where(st.idx >= 0, keys, torch.full_like(keys, _SENTINEL))
    if kk < _hip.KNN_MAX_K:  # a short group is listed completely: pad with "nothing further"
        pad = torch.full((q.shape[0], _hip.KNN_MAX_K - kk), _SENTINEL, dtype=torch.int64, device=q.device)
        keys = torch.cat([keys, pad], dim=1)
    return keys


def knn_search_wide(q, r, k, *, n_groups=None, query_chunk=32768, backend=_hip):
    """``(idx int32 [n, k], dist2 float32 [n, k], dist float32 [n, k])`` of the k nearest rows
    of ``r`` for every row of ``q``, ordered by ``(distance, index)``; any ``k <= len(r)``.
    ``backend`` supplies ``knn_search`` (the HIP library; a CPU stand-in in the host-logic tests)."""
    if backend is _hip:
        _hip.require_gpu()
    device = q.device
    n, m = q.shape[0], r.shape[0]
    if k > m:
        raise ValueError(f"k={k} exceeds the number of reference rows ({m})")
    if n_groups is None:
        n_groups = max(2, 4 * math.ceil(k / _hip.KNN_MAX_K))
    n_groups = min(n_groups, m)
    groups = [r[g::n_groups].contiguous() for g in range(n_groups)]
    idx = torch.empty((n, k), dtype=torch.int32, device=device)
    d2 = torch.empty((n, k), dtype=torch.float32, device=device)
    for c0 in range(0, n, query_chunk):
        qc = q[c0:c0 + query_chunk]
        keys = torch.cat([_group_keys(qc, rg, g, n_groups, backend) for g, rg in enumerate(groups)], dim=1)
        top = torch.sort(keys, dim=1).values[:, :k]
        kth = top[:, k - 1:k]
        last = keys[:, _hip.KNN_MAX_K - 1::_hip.KNN_MAX_K]     # 64th key of every group
        # only a group of more than 64 rows can have rows it did not list: a group of exactly
        # 64 also ends in a real key, but its list is the whole group
        may_cut = torch.tensor([g.shape[0] > _hip.KNN_MAX_K for g in groups], device=device)
        cut_short = ((last != _SENTINEL) & (last < kth) & may_cut).any(dim=1)
        idx[c0:c0 + qc.shape[0]] = (top & 0xFFFFFFFF).to(torch.int32)
        d2[c0:c0 + qc.shape[0]] = (top >> 32).to(torch.int32).view(torch.float32)
        bad = torch.nonzero(cut_short).flatten()
        if bad.numel():
            if max(g.shape[0] for g in groups) <= _hip.KNN_MAX_K:
                raise AssertionError("complete group lists cannot be cut short")
            i2, d22, _ = knn_search_wide(qc[bad], r, k, n_groups=2 * n_groups, query_chunk=query_chunk,
                                         backend=backend)
            idx[c0 + bad] = i2
            d2[c0 + bad] = d22
    dist = d2.to(torch.float64).sqrt().to(torch.float32)
    return idx, d2, dist


class WideState:
    """Result of :func:`knn_any_k` beyond 64 neighbours, with the members of ``_hip.KnnState`` the
    callers of a finished search read (the merged result is exact: nothing is left uncertified)."""

    def __init__(self, idx, dist2, dist):
        self.idx, self.dist2, self.dist = idx, dist2, dist
        self.n_uncertified = torch.zeros((1,), dtype=torch.int32, device=idx.device)
        self.uncertified_rows = torch.empty((0,), dtype=torch.int32, device=idx.device)
        self.blocks = 1


def knn_any_k(q, r, k, *, backend=_hip, **search_args):
    """One exact search for any ``k``: the fused kernel's own 64-neighbour search where that is
    enough, the group-wise merge beyond (``find_neighbor``'s ``_k = max(k_anchor, k_score)`` is the
    caller's choice, scmiac/preprocessing/anchors.py:885-886).  A reference set of fewer than ``k``
    rows pads with -1 / +inf like the narrow search does.  ``search_args`` (``splits``) reach the
    narrow search only."""
    if k <= _hip.KNN_MAX_K:
        return backend.knn_search(q, r, k, **search_args)
    m = r.shape[0]
    idx, d2, dist = knn_search_wide(q, r, min(k, m), backend=backend) if m > 0 and q.shape[0] > 0 else (
        torch.empty((q.shape[0], 0), dtype=torch.int32, device=q.device),
        torch.empty((q.shape[0], 0), dtype=torch.float32, device=q.device),
        torch.empty((q.shape[0], 0), dtype=torch.float32, device=q.device))
    if idx.shape[1] < k:
        pad = k - idx.shape[1]
        idx = torch.cat([idx, idx.new_full((idx.shape[0], pad), -1)], dim=1)
        d2 = torch.cat([d2, d2.new_full((d2.shape[0], pad), float("inf"))], dim=1)
        dist = torch.cat([dist, dist.new_full((dist.shape[0], pad), float("inf"))], dim=1)
    return WideState(idx.contiguous(), d2.contiguous(), dist.contiguous())
