"""Anchor alignment loss of the contrastive VAE training (``scmiac/modeling/loss.py``).

``NTXentLoss2`` has the reference's constructor and ``forward(z_i, z_j)``
(``loss.py:85-133``); forward and backward each run as one HIP kernel
(``scmiac_ntxent2_forward`` / ``scmiac_ntxent2_backward``) instead of the ~40 small
launches of the eager formulation, and the batch x batch similarity matrix is never
stored.  float32, ``d <= 64``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _hip


class _NTXent2Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_i, z_j, temperature):
        z_i = z_i.contiguous()
        z_j = z_j.contiguous()
        parts, lse_row, lse_col = _hip.ntxent2_forward(z_i, z_j, temperature)
        ctx.save_for_backward(z_i, z_j, lse_row, lse_col)
        ctx.temperature = temperature
        return parts.sum()

    @staticmethod
    def backward(ctx, grad_out):
        z_i, z_j, lse_row, lse_col = ctx.saved_tensors
        dzi, dzj = _hip.ntxent2_backward(z_i, z_j, ctx.temperature, lse_row, lse_col, grad_out)
        return dzi, dzj, None


def ntxent2(z_i, z_j, temperature=0.5):
    """0.5 * (cross-entropy of rows + cross-entropy of columns) of the cosine-similarity
    matrix ``normalize(z_i) @ normalize(z_j).T / temperature`` with the diagonal as targets."""
    if z_i.dtype != torch.float32 or z_j.dtype != torch.float32:
        raise TypeError("ntxent2 expects float32 embeddings")
    return _NTXent2Function.apply(z_i, z_j, float(temperature))


class NTXentLoss2(nn.Module):
    """Drop-in for the reference's ``NTXentLoss2`` (``loss.py:85-133``)."""

    def __init__(self, temperature=0.5):
        super().__init__()
        self.temperature = temperature

    def forward(self, z_i, z_j):
        return ntxent2(z_i, z_j, self.temperature)
