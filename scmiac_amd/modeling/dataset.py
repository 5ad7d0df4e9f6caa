"""Device-resident batch sources for the contrastive training loop.

The reference builds two ``torch.utils.data.DataLoader`` objects
(``scmiac/modeling/scmiac.py:143-161``): all cells (``MultiOmicDataset``,
``dataset.py:37-64``) and anchor pairs (``AnchorCellsDataset``, ``dataset.py:70-96``,
one ``DataFrame.iloc`` per sample in 4 worker processes, optionally drawn by a
``WeightedRandomSampler`` that balances the cell types).  Here the standardised
matrices and the anchor table live on the GPU and a batch is one index draw plus two
row gathers, so the loaders keep up with a training step of a few hundred
microseconds.  They are iterated exactly like the reference's loaders by
``train_model`` (``scmiac.py:239-243``: ``zip(all_cells_loader, anchor_cells_loader)``,
``len(all_cells_loader)``) and ``model_inference`` (``scmiac.py:323-324``:
``all_cells_loader.dataset.rna_data``).

Differences from the reference loaders: batches are already on the device (the loop's
``.to(device)`` is a no-op); the random streams are torch's device generators, not
the DataLoader's worker seeds; ``batch["ct"]`` holds integer codes into
``loader.ct_categories`` instead of a list of strings.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _hip
from ..cca_device import DeviceScaler


def _device(device):
    _hip.require_gpu()
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def _standardise(arr, device):
    """``StandardScaler().fit_transform(arr)`` in float64 on the device, then float32
    (what ``torch.tensor(sample, dtype=torch.float32)`` yields per sample in the reference)."""
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(arr), dtype=np.float64)).to(device)
    return DeviceScaler(x).transform(x).to(torch.float32).contiguous()


class _Resident:
    """Stand-in for ``loader.dataset``: the standardised matrices as numpy on request."""

    def __init__(self, rna, atac):
        self._rna, self._atac = rna, atac

    @property
    def rna_data(self):
        return self._rna.cpu().numpy()

    @property
    def atac_data(self):
        return self._atac.cpu().numpy()


class DeviceMultiOmicLoader:
    """All cells of both modalities (``MultiOmicDataset`` + shuffling DataLoader with
    ``drop_last=True``): sample ``i`` pairs RNA row ``i % n_rna`` with ATAC row ``i % n_atac``."""

    def __init__(self, rna_arr, atac_arr, batch_size=256, shuffle=True, drop_last=True, device=None, seed=None):
        self.device = _device(device)
        self.rna = _standardise(rna_arr, self.device)
        self.atac = _standardise(atac_arr, self.device)
        self.size = max(self.rna.shape[0], self.atac.shape[0])
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.dataset = _Resident(self.rna, self.atac)
        self.generator = torch.Generator(device=self.device)
        if seed is not None:
            self.generator.manual_seed(int(seed))

    def __len__(self):
        return self.size // self.batch_size if self.drop_last else -(-self.size // self.batch_size)

    def __iter__(self):
        order = (torch.randperm(self.size, device=self.device, generator=self.generator) if self.shuffle
                 else torch.arange(self.size, device=self.device))
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield {"rna": self.rna[idx % self.rna.shape[0]], "atac": self.atac[idx % self.atac.shape[0]]}


class DeviceAnchorLoader:
    """Anchor pairs (``AnchorCellsDataset``): a batch holds the RNA rows ``x1`` and the
    ATAC rows ``x2`` of the drawn anchors.  ``balanced=True`` draws ``n_anchor`` anchors per
    epoch with replacement and probability proportional to ``1 / count(cell type)``, like
    the reference's ``WeightedRandomSampler`` (``scmiac.py:152-156``); otherwise every
    anchor once, shuffled.

    ``anchor_df`` needs the columns ``x1``, ``x2`` and (for balancing) ``x1_ct``: the frame
    ``find_anchors`` returns."""

    def __init__(self, rna_arr, atac_arr, anchor_df, batch_size=256, balanced=True, drop_last=True,
                 device=None, seed=None, resident=None):
        self.device = _device(device)
        if resident is not None:  # share the matrices with the all-cells loader
            self.rna, self.atac = resident.rna, resident.atac
        else:
            self.rna = _standardise(rna_arr, self.device)
            self.atac = _standardise(atac_arr, self.device)
        x1 = np.asarray(anchor_df["x1"].values, dtype=np.int64)
        x2 = np.asarray(anchor_df["x2"].values, dtype=np.int64)
        if len(x1) and (x1.min() < 0 or x1.max() >= self.rna.shape[0] or x2.min() < 0
                        or x2.max() >= self.atac.shape[0]):
            raise IndexError("anchor index out of bounds")
        self.x1 = torch.from_numpy(x1).to(self.device)
        self.x2 = torch.from_numpy(x2).to(self.device)
        self.n_anchor = len(x1)
        self.batch_size, self.balanced, self.drop_last = int(batch_size), balanced, drop_last
        if "x1_ct" in anchor_df:
            cats, codes = np.unique(np.asarray(anchor_df["x1_ct"].values).astype(str), return_inverse=True)
            self.ct_categories = cats
            self.ct_codes = torch.from_numpy(codes.astype(np.int64)).to(self.device)
        else:
            if balanced:
                raise KeyError("balanced sampling needs the 'x1_ct' column")
            self.ct_categories, self.ct_codes = np.array([], dtype=str), None
        if balanced:
            counts = torch.bincount(self.ct_codes, minlength=len(self.ct_categories)).to(torch.float64)
            self.weights = (1.0 / counts)[self.ct_codes]
        self.dataset = _Resident(self.rna, self.atac)
        self.generator = torch.Generator(device=self.device)
        if seed is not None:
            self.generator.manual_seed(int(seed))

    def __len__(self):
        return self.n_anchor // self.batch_size if self.drop_last else -(-self.n_anchor // self.batch_size)

    def draw(self):
        """Anchor rows of one epoch, in batch order."""
        if self.balanced:
            return torch.multinomial(self.weights, self.n_anchor, replacement=True, generator=self.generator)
        return torch.randperm(self.n_anchor, device=self.device, generator=self.generator)

    def __iter__(self):
        order = self.draw()
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            batch = {"rna_anchor": self.rna[self.x1[idx]], "atac_anchor": self.atac[self.x2[idx]]}
            if self.ct_codes is not None:
                batch["ct"] = self.ct_codes[idx]
            yield batch


def anchor_frame(x1, x2, labels=None):
    """Minimal anchor table for :class:`DeviceAnchorLoader` from index arrays (numpy or
    device tensors, e.g. the kept pairs of ``anchor_search.filter_anchors``) and, for balanced
    sampling, the RNA-side labels of all cells: no CSV or score columns needed."""
    import pandas as pd

    def host(v):
        return v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)

    x1, x2 = host(x1).astype(np.int64), host(x2).astype(np.int64)
    frame = pd.DataFrame({"x1": x1, "x2": x2})
    if labels is not None:
        frame["x1_ct"] = np.asarray(labels)[x1]
    return frame


def make_loaders(adata_rna, adata_atac, anchor_df, rna_latent_key="X_pca", atac_latent_key="lsi49",
                 batch_size=256, balanced_sampler=True, device=None, seed=None):
    """The two loaders ``preprocess`` builds (``scmiac.py:140-161``), resident on the device."""
    rna_arr = adata_rna.obsm[rna_latent_key]
    atac_arr = adata_atac.obsm[atac_latent_key]
    all_cells = DeviceMultiOmicLoader(rna_arr, atac_arr, batch_size=batch_size, device=device, seed=seed)
    anchors = DeviceAnchorLoader(None, None, anchor_df, batch_size=batch_size, balanced=balanced_sampler,
                                 device=device, seed=None if seed is None else seed + 1, resident=all_cells)
    return all_cells, anchors
