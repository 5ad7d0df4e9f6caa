"""Canonical correlation vectors on the GPU without forming the cell x cell matrix.

The reference (``scmiac/preprocessing/anchors.py:171-242``) standardises two
cells x genes matrices, multiplies them into a dense ``n1 x n2`` float64 matrix
(after cutting both down to ``max_cc_cell`` cells, ``anchors.py:480-507``), runs
scikit-learn's randomized ``TruncatedSVD`` on it and projects the cells that were
left out.  Every product that algorithm takes with ``A = X1 @ X2.T`` has a thin
right-hand side, so here ``A @ Y`` is evaluated as ``X1 @ (X2.T @ Y)``: the work
per product drops from ``n1 * n2 * r`` to ``(n1 + n2) * genes * r`` and the
``n1 x n2`` matrix never exists.  All arithmetic is float64 on the device (rocBLAS
GEMMs, hipSOLVER LU / QR of the thin ``n x r`` blocks); only ``r x r`` matrices and
pivot vectors come back to the host.

The algorithm is the one ``TruncatedSVD(algorithm="randomized")`` runs (Halko et
al. range finder with ``n_iter=5`` LU-normalised power iterations, 10 oversamples,
Gaussian start drawn from ``numpy.random.RandomState(random_state)``, sign
convention of ``svd_flip(u_based_decision=False)``, ``X @ components_.T`` as the
left vectors), so the result agrees with the reference's to float64 rounding; it is
not bit-identical because sums are taken in another order.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _hip

N_ITER = 5          # TruncatedSVD default
N_OVERSAMPLES = 10  # TruncatedSVD default


def _device():
    _hip.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _dense_rows(data, rows=None):
    """Host rows of a dense or scipy-sparse matrix as a C-ordered float64 array."""
    from scipy.sparse import issparse

    part = data if rows is None else data[rows]
    if issparse(part):
        part = part.toarray()
    return np.ascontiguousarray(np.asarray(part), dtype=np.float64)


class DeviceScaler:
    """Per-gene mean / standard deviation with ``sklearn.preprocessing.StandardScaler``'s
    conventions (population variance, constant genes get scale 1), fitted on the device."""

    def __init__(self, x):
        n = x.shape[0]
        total = x.sum(dim=0)
        self.mean_ = total / n
        centred = x - self.mean_
        correction = centred.sum(dim=0)
        var = ((centred * centred).sum(dim=0) - correction * correction / n) / n
        eps = float(np.finfo(np.float64).eps)
        constant = var <= n * eps * var + (n * self.mean_ * eps) ** 2
        scale = torch.sqrt(var)
        scale[constant] = 1.0
        self.var_ = var
        self.scale_ = scale

    def transform(self, x):
        return (x - self.mean_) / self.scale_


def _lu_rows(y, status=None):
    """``P @ L`` of the partially pivoted LU factorisation of a tall block
    (what ``scipy.linalg.lu(y, permute_l=True)[0]`` returns).  ``status``: see ``_hip.lu_row_perm``."""
    n, r = y.shape
    m = min(n, r)
    lu, piv = torch.linalg.lu_factor(y)
    low = torch.tril(lu[:, :m], diagonal=-1)
    low[:m].diagonal().fill_(1.0)
    perm = _hip.lu_row_perm(piv[:m], n, status)  # row swaps in the order LAPACK applied them, applied on the device
    out = torch.empty_like(low)
    out[perm] = low
    return out


def randomized_svd_of_product(t1, t2, n_components, random_state=0, n_iter=N_ITER,
                              n_oversamples=N_OVERSAMPLES):
    """Truncated SVD of ``A = t1 @ t2.T`` (``t1`` n1 x g, ``t2`` n2 x g, float64 on the
    device) the way ``TruncatedSVD.fit_transform`` does it, without forming ``A``.

    Returns ``(x_transformed [n1, c], singular_values [c], components [c, n2])``:
    ``TruncatedSVD``'s return value, ``singular_values_`` and ``components_``."""
    n1, n2 = t1.shape[0], t2.shape[0]
    if n_components > n2:
        raise ValueError(f"n_components({n_components}) must be <= n_features({n2}).")
    transpose = n1 < n2
    a, b = (t2, t1) if transpose else (t1, t2)  # M = a @ b.T, rows(M) >= cols(M)
    rng = np.random.RandomState(random_state) if not isinstance(random_state, np.random.RandomState) \
        else random_state
    r = n_components + n_oversamples
    q = torch.from_numpy(rng.normal(size=(b.shape[0], r))).to(t1.device)

    def m_dot(y):      # M @ y
        return a @ (b.T @ y)

    def mt_dot(y):     # M.T @ y
        return b @ (a.T @ y)

    bad_pivot = torch.zeros((1,), dtype=torch.int32, device=t1.device)
    for _ in range(n_iter):
        q = _lu_rows(m_dot(q), bad_pivot)
        q = _lu_rows(mt_dot(q), bad_pivot)
    q, _ = torch.linalg.qr(m_dot(q), mode="reduced")

    # B = Q.T @ M is r x cols(M); its SVD through a QR of B.T keeps the long side on the device
    bt = mt_dot(q)                                   # cols(M) x r
    q2, r2 = torch.linalg.qr(bt, mode="reduced")     # B = r2.T @ q2.T
    r2_host = r2.T.cpu().numpy()                     # (the one visit to the host: everything above has run)
    if int(bad_pivot.item()):
        raise ValueError("the LU factorisations of the range finder returned pivots outside their blocks "
                         "(non-finite values in the input matrices?)")
    uh, s, wt = np.linalg.svd(r2_host, full_matrices=False)
    uh = torch.from_numpy(np.ascontiguousarray(uh)).to(t1.device)
    wt = torch.from_numpy(np.ascontiguousarray(wt)).to(t1.device)
    u_m = (q @ uh)[:, :n_components]                 # left vectors of M
    vt_m = (wt @ q2.T)[:n_components]                # right vectors of M
    s = torch.from_numpy(s[:n_components].copy()).to(t1.device)
    vt = u_m.T.contiguous() if transpose else vt_m.contiguous()   # right vectors of A: c x n2

    # sign convention: the largest |entry| of every row of vt is positive
    pos = vt.abs().argmax(dim=1, keepdim=True)
    vt = vt * torch.sign(vt.gather(1, pos))
    x_transformed = t1 @ (t2.T @ vt.T)
    return x_transformed, s, vt


def cca_device(data1, data2, scale1=True, scale2=True, n_components=50, max_cc_cell=20000,
               chunk_size=50000, random_state=0, k_filter=None, n_features=200, as_numpy=True):
    """``cca`` of the reference (``anchors.py:171-242``) for ``svd_algorithm="randomized"``.

    Same arguments and return value ``(U, V, high_dim_feature)``; the cells used for
    the decomposition are the ones the reference draws (``numpy.random.seed`` +
    ``numpy.random.choice``).  ``as_numpy=False`` leaves ``U`` and ``V`` on the device."""
    from .preprocessing.anchors import top_features_idx

    dev = _device()
    np.random.seed(random_state)
    np.random.seed(random_state)  # the reference seeds once in cca and again in downsample
    n1, n2 = data1.shape[0], data2.shape[0]
    sel1 = np.random.choice(np.arange(n1), min(max_cc_cell, n1), False) if n1 > max_cc_cell else None
    sel2 = np.random.choice(np.arange(n2), min(max_cc_cell, n2), False) if n2 > max_cc_cell else None
    t1 = torch.from_numpy(_dense_rows(data1, sel1)).to(dev)
    t2 = torch.from_numpy(_dense_rows(data2, sel2)).to(dev)
    scaler1 = scaler2 = None
    if scale1:
        scaler1 = DeviceScaler(t1)
        t1 = scaler1.transform(t1)
    if scale2:
        scaler2 = DeviceScaler(t2)
        t2 = scaler2.transform(t2)

    xt, s, vt = randomized_svd_of_product(t1, t2, n_components, random_state=random_state)
    sel_dim = s != 0
    print("non zero dims", int(sel_dim.sum().item()))
    s_sel = s[sel_dim]
    V = vt[sel_dim].T.contiguous()
    U = xt[:, sel_dim] / s_sel

    high_dim_feature = None
    if k_filter:
        loadings = U.T @ t1 + V.T @ t2
        high_dim_feature = top_features_idx(loadings.cpu().numpy(), n_features=n_features)

    def project(data, scaler, right):
        parts = []
        for start in range(0, data.shape[0], chunk_size):
            tmp = torch.from_numpy(_dense_rows(data[start:start + chunk_size])).to(dev)
            if scaler is not None:
                tmp = scaler.transform(tmp)
            parts.append(tmp @ right)
        return torch.cat(parts) / s_sel

    if n2 > max_cc_cell:
        V_all = project(data2, scaler2, t1.T @ U)               # rows: (U.T @ X1) @ cell
    if n1 > max_cc_cell:
        U = project(data1, scaler1, (vt[sel_dim] @ t2).T)       # cell @ (components @ X2).T
    if n2 > max_cc_cell:
        V = V_all
    if as_numpy:
        return U.cpu().numpy(), V.cpu().numpy(), high_dim_feature
    return U, V, high_dim_feature
