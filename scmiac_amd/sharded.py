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


class _ActiveGroup:
    def __init__(self, group):
        self.group = group  # None = the default group


_active = None


def use_process_group(group=None, enable=True):
    """From now on the drop-in API (``find_neighbor``, ``SeuratIntegration.find_anchor``,
    ``find_anchors``) shares the query rows of its k-NN tables out over the ranks of ``group``
    (default: the whole initialised process group).  Opt-in on purpose: the searches become
    collective calls, so EVERY rank of the group has to make them, with the same inputs; a
    process group that exists for other reasons is left alone.  ``enable=False`` switches back."""
    global _active
    if not enable:
        _active = None
        return
    if not dist.is_initialized():
        raise RuntimeError("use_process_group needs an initialised torch.distributed process group")
    _active = _ActiveGroup(group) if dist.get_world_size(group) > 1 else None


def active_group():
    """The group installed by :func:`use_process_group` (an object with a ``group`` member), or None."""
    return _active


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


class PendingGather:
    """An all-gather of per-shard row blocks that has been issued and may still be running
    (on RCCL's own stream, or gloo's worker threads): searches queued after it overlap with it.
    :meth:`wait` makes the current stream wait and returns the ``[n, ...]`` tensor."""

    def __init__(self, out, work, n):
        self._out, self._work, self._n = out, work, n

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._out[:self._n]


def gather_rows_async(local, n, group=None):
    """:func:`gather_rows`, issued without waiting for it."""
    world = dist.get_world_size(group)
    per = (n + world - 1) // world
    if local.shape[0] < per:
        pad = local.new_full((per - local.shape[0],) + tuple(local.shape[1:]), -1)
        local = torch.cat([local, pad])
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=True)
    return PendingGather(out, work, n)


def knn_replicated(q_local, r_full, k, *, backend=_hip):
    """This rank's query rows against the whole (replicated) reference set (any ``k``: beyond the
    fused kernel's 64 neighbours per row the search is the group-wise merge of ``wide_knn``)."""
    from .wide_knn import knn_any_k
    return knn_any_k(q_local, r_full, k, backend=backend)


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
    if int(worst.item()) > 0:
        # A second trip for the rows some rank could not certify.  Every rank keeps the ring
        # turning (the same sends and receives everywhere), but only ranks that have such rows
        # evaluate anything: a rank without them passes the blocks on untouched.
        mine = rows.shape[0] > 0

        def direct(block, owner):
            if mine:
                backend.knn_exact_rows(q_local, block, k, rows, n_rows, state, col_offset=offsets[owner])

        trip(direct)
    return state


def find_neighbor_sharded(cc1, cc2, k, *, group=None, backend=_hip):
    """``find_neighbor`` with each table's query rows sharded over the ranks of
    ``group`` and the reference sets replicated; every rank returns the full tables.
    Each table's all-gather is issued as soon as its search is queued and waited for only at
    the end, so it travels while the next table is being searched."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n1, n2 = cc1.shape[0], cc2.shape[0]
    lo1, hi1 = shard_bounds(n1, world, rank)
    lo2, hi2 = shard_bounds(n2, world, rank)
    kk = k + 1  # the self tables come from an index built with n_neighbors = k + 1 (anchors.py:549, 559)
    s11 = knn_replicated(cc1[lo1:hi1], cc1, kk, backend=backend)
    p11 = gather_rows_async(s11.idx, n1, group)
    s21 = knn_replicated(cc2[lo2:hi2], cc1, k, backend=backend)
    p21 = gather_rows_async(s21.idx, n2, group)
    s22 = knn_replicated(cc2[lo2:hi2], cc2, kk, backend=backend)
    p22 = gather_rows_async(s22.idx, n2, group)
    s12 = knn_replicated(cc1[lo1:hi1], cc2, k, backend=backend)
    p12 = gather_rows_async(s12.idx, n1, group)
    g11, g21, g22, g12 = p11.wait(), p21.wait(), p22.wait(), p12.wait()
    unc = tuple(int(s.n_uncertified.item()) for s in (s11, s12, s21, s22))
    return NeighborTables(G11=g11[:, 1:kk].contiguous(), G12=g12, G21=g21, G22=g22[:, 1:kk].contiguous(),
                          uncertified=unc)


def anchors_sharded(cc1, cc2, k, *, k_anchor=5, k_score=30, group=None, backend=_hip):
    """The whole search -- four tables, mutual nearest neighbours, shared-neighbour counts
    (``anchors.py:525-572, 660-664``) -- with everything sharded by the rows of ``cc1``:

    * rank r searches its rows of ``cc1`` (tables G11, G12) and its rows of ``cc2`` (G21, G22);
    * only G21 and G22 are all-gathered (an anchor ``(x, y)`` needs rows ``y`` of both, and
      ``y`` can be any row); G11 and G12 never leave the rank that made them;
    * rank r finds and scores the anchors whose ``x`` lies in its shard; since shards are
      contiguous row ranges, the anchors of all ranks in rank order ARE the reference's
      ``(i, j)`` loop order;
    * one small all-gather of the per-rank anchor lists (pairs + integer counts).

    Returns ``(pairs int64 [n_anchor, 2], raw int32 [n_anchor])`` on every rank, in global
    row numbers; ``min_max`` and the label filter follow on whichever rank wants them."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n1, n2 = cc1.shape[0], cc2.shape[0]
    lo1, hi1 = shard_bounds(n1, world, rank)
    lo2, hi2 = shard_bounds(n2, world, rank)
    kk = k + 1
    s21 = knn_replicated(cc2[lo2:hi2], cc1, k, backend=backend)
    p21 = gather_rows_async(s21.idx, n2, group)
    s22 = knn_replicated(cc2[lo2:hi2], cc2, kk, backend=backend)
    p22 = gather_rows_async(s22.idx[:, 1:kk].contiguous(), n2, group)
    g11 = knn_replicated(cc1[lo1:hi1], cc1, kk, backend=backend).idx[:, 1:kk].contiguous()
    g12 = knn_replicated(cc1[lo1:hi1], cc2, k, backend=backend).idx
    g21, g22 = p21.wait(), p22.wait()
    ks = min(k_score, k)
    n_local = hi1 - lo1
    device = g12.device
    if n_local > 0:
        # the kernels number the rows of cc1 from 0: shift the entries of G21 (rows of cc1) into
        # this shard's numbering; entries of other shards fall outside [0, n_local) and match nothing
        pairs, n_pairs, status = backend.find_mnn(g12, (g21[:, :k_anchor] - lo1).contiguous(), k_anchor)
        if int(status.item()) != 0:
            raise IndexError("neighbour index out of bounds")
        cnt = int(n_pairs.item())
        # (the counts compare row NUMBERS of cc1 between G11 and G21: no shift there)
        raw, _ = backend.snn_score(pairs, n_pairs, g11[:, :ks], g12[:, :ks], g21[:, :ks], g22[:, :ks], ks,
                                   n_pairs_max=max(cnt, 1))
        pairs = pairs[:cnt].clone()
        pairs[:, 0] += lo1
        raw = raw[:cnt]
    else:
        cnt = 0
        pairs = torch.empty((0, 2), dtype=torch.int64, device=device)
        raw = torch.empty((0,), dtype=torch.int32, device=device)
    # per-rank counts, then one padded gather of (x, y, count) triples
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, torch.tensor([cnt], dtype=torch.int64, device=device), group=group)
    counts = [int(c) for c in counts.cpu()]
    cap = max(max(counts), 1)
    mine = torch.full((cap, 3), -1, dtype=torch.int64, device=device)
    mine[:cnt, :2] = pairs
    mine[:cnt, 2] = raw.to(torch.int64)
    everyone = torch.empty((world * cap, 3), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    parts = [everyone[r * cap:r * cap + c] for r, c in enumerate(counts)]
    table = torch.cat(parts) if parts else everyone[:0]
    return table[:, :2].contiguous(), table[:, 2].to(torch.int32).contiguous()


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
