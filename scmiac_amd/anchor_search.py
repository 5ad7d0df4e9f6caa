"""Device-resident anchor search: k-NN tables -> MNN -> SNN score -> label filter.

Host-side orchestration of the HIP kernels for the path
``find_neighbor -> find_mnn -> score_anchor -> label filter`` of the reference
(``scmiac/preprocessing/anchors.py:525-679, 779-789`` and
``scmiac/modeling/scmiac.py:89-103``).  Inputs and every intermediate stay in HBM;
only the integer score vector crosses to the host, because the two percentiles of
``min_max`` are taken with ``numpy.percentile`` exactly as the reference takes
them (anchors.py:577) so that ``score`` is bit-identical.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import _hip


@dataclass
class NeighborTables:
    """The four k-NN tables of ``find_neighbor`` as int32 device tensors."""
    G11: torch.Tensor
    G12: torch.Tensor
    G21: torch.Tensor
    G22: torch.Tensor
    uncertified: tuple = (0, 0, 0, 0)  # rows per table that needed the direct-evaluation pass

    def numpy(self):
        return tuple(g.cpu().numpy() for g in (self.G11, self.G12, self.G21, self.G22))


def knn_table(query, ref, k, *, want_dist=False, splits=0):
    """``index.query(query, k)`` / ``index.neighbor_graph`` for an exact index over ``ref``."""
    from .wide_knn import knn_any_k  # (imports this package's binding)
    st = knn_any_k(query, ref, k, splits=splits)
    return (st.idx, st.dist, st) if want_dist else (st.idx, None, st)


def find_neighbor_device(cc1, cc2, k):
    """anchors.py:525-566 on device tensors (float32, row-major).  After
    :func:`scmiac_amd.sharded.use_process_group` the query rows of every table are shared out over
    the ranks of that group (every rank passes the same ``cc1`` / ``cc2`` and gets the whole tables)."""
    n1, n2 = cc1.shape[0], cc2.shape[0]
    kk = k + 1  # neighbor_graph is built with n_neighbors = k + 1, column 0 dropped
    if k < 1:
        raise ValueError(f"k={k} must be at least 1")
    from . import sharded  # (imports this module)
    if sharded.active_group() is not None:
        return sharded.find_neighbor_sharded(cc1, cc2, k, group=sharded.active_group().group)
    g11, _, s11 = knn_table(cc1, cc1, kk)
    g21, _, s21 = knn_table(cc2, cc1, k)
    g22, _, s22 = knn_table(cc2, cc2, kk)
    g12, _, s12 = knn_table(cc1, cc2, k)
    unc = tuple(int(s.n_uncertified.item()) for s in (s11, s12, s21, s22))
    return NeighborTables(G11=g11[:, 1:kk].contiguous(), G12=g12, G21=g21, G22=g22[:, 1:kk].contiguous(),
                          uncertified=unc)


@dataclass
class AnchorScores:
    pairs: torch.Tensor      # int64 [cap, 2] device, first n rows valid
    n: int                   # number of raw anchors
    raw: torch.Tensor        # int32 [cap] device shared-neighbour counts
    p_lo: float
    p_hi: float
    score: np.ndarray        # float64 [n] host, min_max-normalised


def mnn_and_score(tables: NeighborTables, k_anchor=5, k_score=30):
    """find_mnn + the integer part of score_anchor on the device, min_max on the host."""
    pairs, n_pairs, status = _hip.find_mnn(tables.G12, tables.G21, k_anchor)
    n = int(n_pairs.item())  # also the launch bound for the per-anchor kernels
    if int(status.item()) != 0:
        raise IndexError("neighbour index out of bounds")
    raw, _ = _hip.snn_score(pairs, n_pairs, tables.G11, tables.G12, tables.G21, tables.G22, k_score,
                            n_pairs_max=n)
    raw_host = raw[:n].cpu().numpy()
    if n == 0:
        return AnchorScores(pairs, 0, raw, float("nan"), float("nan"), np.empty(0)), n_pairs
    tmp = raw_host.astype(np.int64)
    p_lo, p_hi = np.percentile(tmp, [1, 90])  # anchors.py:577
    with np.errstate(all="ignore"):
        score = (tmp - p_lo) / (p_hi - p_lo)
        score[score > 1] = 1
        score[score < 0] = 0
    return AnchorScores(pairs, n, raw, float(p_lo), float(p_hi), score), n_pairs


@dataclass
class FilteredAnchors:
    x1: np.ndarray
    x2: np.ndarray
    score: np.ndarray
    is_same: np.ndarray
    index: np.ndarray
    n_pass_score: int


def filter_anchors(scores: AnchorScores, n_pairs, codes1, codes2, *, ct_filter=True, threshold=0.2):
    """scmiac.py:91-99 on the device: score threshold, label equality, ordered compaction."""
    if scores.n == 0:
        e = np.empty(0, dtype=np.int64)
        return FilteredAnchors(e, e.copy(), np.empty(0), np.empty(0, dtype=bool), e.copy(), 0)
    out_pairs, out_score, out_same, out_index, counts, _ = _hip.anchor_filter(
        scores.pairs, n_pairs, scores.raw, scores.p_lo, scores.p_hi, codes1, codes2,
        threshold=threshold, ct_filter=ct_filter, n_pairs_max=scores.n)
    n_pass, n_keep = (int(v) for v in counts.cpu().numpy())
    op = out_pairs[:n_keep].cpu().numpy()
    return FilteredAnchors(op[:, 0].copy(), op[:, 1].copy(), out_score[:n_keep].cpu().numpy(),
                           out_same[:n_keep].cpu().numpy().astype(bool), out_index[:n_keep].cpu().numpy(),
                           n_pass)


def label_codes(labels1, labels2):
    """Integer codes such that codes are equal iff ``astype(str)`` labels are equal."""
    s1 = np.asarray(labels1).astype(str)
    s2 = np.asarray(labels2).astype(str)
    uniq, inv = np.unique(np.concatenate([s1, s2]), return_inverse=True)
    inv = inv.astype(np.int32)
    return inv[:len(s1)], inv[len(s1):]
