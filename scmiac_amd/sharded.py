"""Anchor search over several GPUs of one node: one process per GPU, query rows sharded.

The k-NN tables of ``find_neighbor`` (``scmiac/preprocessing/anchors.py:525-566``)
are independent per query row, so every rank searches a contiguous block of query
rows and the per-shard tables are put together with one all-gather.  Two ways to
hold the reference set:

* replicated (:func:`knn_replicated`): every rank has all reference rows; no
  communication while searching;
* ring (:func:`knn_ring`): every rank holds one block of reference rows; the
  blocks travel round the ring of ranks (one send + one receive per step, on
  xGMI between neighbouring GPUs) while the block at hand is searched and merged
  into the running top-k (``merge`` / ``col_offset`` of ``scmiac_knn_search``).

``torch.distributed`` does the transport (backend ``nccl`` = RCCL on GPUs; the host
logic is backend-agnostic and is exercised with ``gloo`` in the CPU tests, where
``backend`` is a stand-in with the same ``knn_search`` / ``knn_exact_rows`` members
as :mod:`scmiac_amd._hip`).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _hip
from .anchor_search import NeighborTables


def shard_bounds(n, world, rank):
    """Rows ``[lo, hi)`` of rank ``rank``: contiguous blocks of ``ceil(n / world)`` rows."""
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def gather_rows(local, n, group=None):
    """All-gather of per-shard row blocks laid out by :func:`shard_bounds` -> ``[n, ...]``."""
    world = dist.get_world_size(group)
    per = (n + world - 1) // world
    if local.shape[0] < per:  # short or empty last shards: every rank contributes equally
        pad = local.new_full((per - local.shape[0],) + tuple(local.shape[1:]), -1)
        local = torch.cat([local, pad])
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n]


def knn_replicated(q_local, r_full, k, *, backend=_hip):
    """This rank's query rows against the whole (replicated) reference set."""
    return backend.knn_search(q_local, r_full, k)


def _uncertified(state):
    n = int(state.n_uncertified.item())
    return state.uncertified_rows[:n]


def knn_ring(q_local, r_local, k, *, group=None, backend=_hip):
    """Exact k-NN of this rank's query rows among the reference rows of ALL ranks.

    ``r_local`` is this rank's block of the reference set; global reference row
    numbers follow rank order.  ``world`` steps: search the block at hand while it
    is being passed on to the next rank and the previous rank's block arrives.
    Rows whose result could not be proved exact from the fused kernel's error bound
    (near-ties at the shortlist edge) are then searched by direct evaluation in a
    second trip round the ring, which is skipped when no rank has such rows.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = r_local.device
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([r_local.shape[0]], dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts = [int(c) for c in counts.cpu()]
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + c)
    cap, d = max(max(counts), 1), r_local.shape[1]
    nxt_rank, prv_rank = (rank + 1) % world, (rank - 1) % world
    if group is not None:
        nxt_rank, prv_rank = dist.get_global_rank(group, nxt_rank), dist.get_global_rank(group, prv_rank)

    def trip(visit):
        """One trip of the reference blocks round the ring; ``visit(block, owner)`` per step."""
        bufs = [torch.empty((cap, d), dtype=r_local.dtype, device=device) for _ in range(2)]
        bufs[0][:counts[rank]].copy_(r_local)
        for s in range(world):
            owner = (rank - s) % world
            cur, nxt = bufs[s & 1], bufs[(s + 1) & 1]
            reqs = []
            if s + 1 < world:  # pass the block on while it is searched
                reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, cur, nxt_rank, group),
                                               dist.P2POp(dist.irecv, nxt, prv_rank, group)])
            visit(cur[:counts[owner]], owner)
            for r in reqs:
                r.wait()

    state = None

    def search(block, owner):
        nonlocal state
        state = backend.knn_search(q_local, block, k, col_offset=offsets[owner], state=state,
                                   exact_fallback=False)

    trip(search)
    # the certificate of the last step covers every block seen: its list is the final one
    rows = _uncertified(state).clone()
    n_rows = torch.tensor([rows.shape[0]], dtype=torch.int32, device=rows.device)
    worst = n_rows.to(torch.int64).clone()
    dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group)
    if int(worst.item()) > 0:  # every rank must keep the ring turning, with or without rows of its own
        if rows.shape[0] == 0:
            rows = torch.zeros(1, dtype=torch.int32, device=rows.device)

        def direct(block, owner):
            backend.knn_exact_rows(q_local, block, k, rows, n_rows, state, col_offset=offsets[owner])

        trip(direct)
    return state


def find_neighbor_sharded(cc1, cc2, k, *, group=None, backend=_hip):
    """``find_neighbor`` with each table's query rows sharded over the ranks of
    ``group`` and the reference sets replicated; every rank returns the full tables."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n1, n2 = cc1.shape[0], cc2.shape[0]
    lo1, hi1 = shard_bounds(n1, world, rank)
    lo2, hi2 = shard_bounds(n2, world, rank)
    kk = k + 1  # the self tables come from an index built with n_neighbors = k + 1 (anchors.py:549, 559)
    s11 = knn_replicated(cc1[lo1:hi1], cc1, kk, backend=backend)
    g11 = gather_rows(s11.idx, n1, group)
    s21 = knn_replicated(cc2[lo2:hi2], cc1, k, backend=backend)
    g21 = gather_rows(s21.idx, n2, group)
    s22 = knn_replicated(cc2[lo2:hi2], cc2, kk, backend=backend)
    g22 = gather_rows(s22.idx, n2, group)
    s12 = knn_replicated(cc1[lo1:hi1], cc2, k, backend=backend)
    g12 = gather_rows(s12.idx, n1, group)
    unc = tuple(int(s.n_uncertified.item()) for s in (s11, s12, s21, s22))
    return NeighborTables(G11=g11[:, 1:kk].contiguous(), G12=g12, G21=g21, G22=g22[:, 1:kk].contiguous(),
                          uncertified=unc)


def closer_count_sharded(q, r, *, match=None, group=None, backend=_hip):
    """``closer_count`` (the integer behind FOSCTTM, ``evaluation/benchmark.py:99-106``) with
    the query rows sharded over the ranks and the references replicated: no collective while
    counting, one all-gather of the counts.  Every rank passes the full ``q`` / ``r`` /
    ``match`` and receives all counts (int64 ``[n]``)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = q.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    if match is None:
        if r.shape[0] != n:
            raise ValueError("Shapes do not match!")
        match = torch.arange(n, dtype=torch.int32, device=q.device)
    local = backend.closer_count(q[lo:hi], r, match=match[lo:hi])
    return gather_rows(local.reshape(-1, 1), n, group).reshape(-1)


def foscttm_sharded(x, y, *, group=None, backend=_hip):
    """Both directions of FOSCTTM over the ranks of ``group``: float32 numpy arrays
    ``(foscttm_x, foscttm_y)`` like the reference's (the division runs on the host)."""
    import numpy as np

    n = np.float32(x.shape[0])
    fx = closer_count_sharded(x, y, group=group, backend=backend).cpu().numpy().astype(np.float32) / n
    fy = closer_count_sharded(y, x, group=group, backend=backend).cpu().numpy().astype(np.float32) / n
    return fx, fy
