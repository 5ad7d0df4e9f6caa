"""Drop-ins for the distance-based metrics of ``scmiac/evaluation/benchmark.py``.

* ``foscttm`` (75-112) and ``partially_matched_foscttm`` (115-159): the reference forms
  the dense cells x cells distance matrix (N^2 memory; 4 TB at a million cells).  Here a
  fused MFMA kernel counts, per cell, the cells of the other modality that are closer
  than the true match, without storing the matrix (``scmiac_closer_count``).  Distances
  are the float32 direct-difference ones of the k-NN path, so results differ from the
  reference's ``x_sq + y_sq - 2 x @ y.T`` only where two distances agree to rounding.
* ``neighbor_conservation`` (35-68) and ``knn_matching`` (342-374): exact k-NN searches
  on the GPU in place of sklearn's ``NearestNeighbors`` / a dense cosine matrix + argsort.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _hip


def _dev(x, device=None):
    _hip.require_gpu()
    dev = torch.device(device) if device not in (None, "cpu") else torch.device("cuda", torch.cuda.current_device())
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    return t.to(device=dev, dtype=torch.float32).contiguous()


def foscttm(x, y, device=None):
    """Fraction of samples closer than the true match, per cell, in both directions.

    Returns ``(foscttm_x, foscttm_y)`` as float32 numpy arrays like the reference
    (``benchmark.py:105-112``); ``device`` is accepted for signature compatibility (the
    computation always runs on the GPU)."""
    if tuple(x.shape) != tuple(y.shape):
        raise ValueError("Shapes do not match!")
    xd, yd = _dev(x, device), _dev(y, device)
    n = np.float32(xd.shape[0])
    # the division runs on the host: IEEE float32 division, as in the reference's CPU mean
    fx = _hip.closer_count(xd, yd).cpu().numpy().astype(np.float32) / n
    fy = _hip.closer_count(yd, xd).cpu().numpy().astype(np.float32) / n
    return fx, fy


def partially_matched_foscttm(x, y, x_barcodes, y_barcodes):
    """FOSCTTM over the cells present in both data sets (``benchmark.py:115-159``):
    float64 arrays over the sorted common barcodes."""
    x_barcodes = np.asarray(x_barcodes)
    y_barcodes = np.asarray(y_barcodes)
    x_pos = {b: i for i, b in enumerate(x_barcodes)}
    y_pos = {b: i for i, b in enumerate(y_barcodes)}
    matched = np.intersect1d(x_barcodes, y_barcodes)
    if matched.size == 0:
        raise ValueError("No matching barcodes found between x_barcodes and y_barcodes.")
    ix = np.array([x_pos[b] for b in matched])
    iy = np.array([y_pos[b] for b in matched])
    xd, yd = _dev(x), _dev(y)
    ixd = torch.from_numpy(ix).to(xd.device)
    iyd = torch.from_numpy(iy).to(xd.device)
    print(f"D_x.shape:{(len(matched), yd.shape[0])}")
    print(f"D_y.shape:{(len(matched), xd.shape[0])}")
    fx = _hip.closer_count(xd[ixd], yd, match=iyd).cpu().numpy().astype(np.float64) / yd.shape[0]
    fy = _hip.closer_count(yd[iyd], xd, match=ixd).cpu().numpy().astype(np.float64) / xd.shape[0]
    return fx, fy


def _self_knn(data, k):
    d = _dev(data)
    if k > _hip.KNN_MAX_K:
        from ..wide_knn import knn_search_wide
        return knn_search_wide(d, d, k)[0]
    return _hip.knn_search(d, d, k).idx


def neighbor_conservation(orig_data, latent_data, n_neighbors=10, exclude_top=0):
    """Mean Jaccard index between each cell's neighbour set in the original and in the
    latent space (``benchmark.py:35-68``).  Up to 256 features."""
    k = n_neighbors + exclude_top
    a = _self_knn(orig_data, k)[:, exclude_top:]
    b = _self_knn(latent_data, k)[:, exclude_top:]
    inter = (a[:, :, None] == b[:, None, :]).any(dim=2).sum(dim=1).to(torch.float64)
    width = a.shape[1]
    jaccard = inter / (2 * width - inter)  # rows of a k-NN table hold distinct indices
    return np.mean(jaccard.cpu().numpy())


def knn_matching(rna_latent, atac_latent, k=1):
    """Cross-modal matching accuracy (``benchmark.py:342-374``): is cell i of the other
    modality among the k most cosine-similar ones?  Rows are L2-normalised on the device;
    on unit vectors the cosine order is the Euclidean order."""
    a = torch.nn.functional.normalize(_dev(rna_latent), dim=1)
    b = torch.nn.functional.normalize(_dev(atac_latent), dim=1)
    own_a = torch.arange(a.shape[0], device=a.device, dtype=torch.int32)[:, None]
    own_b = torch.arange(b.shape[0], device=a.device, dtype=torch.int32)[:, None]
    hit_a = int((_hip.knn_search(a, b, k).idx == own_a).any(dim=1).sum().item())
    hit_b = int((_hip.knn_search(b, a, k).idx == own_b).any(dim=1).sum().item())
    return {"rna_to_atac_accuracy": np.round(hit_a / a.shape[0], 5),
            "atac_to_rna_accuracy": np.round(hit_b / b.shape[0], 5),
            "overall_accuracy": np.round((hit_a + hit_b) / (a.shape[0] + b.shape[0]), 5)}
