"""``find_anchors`` with the reference signature (``scmiac/modeling/scmiac.py:14-111``).

The feature selection runs on the host as in the reference; the CCA, the k-NN
tables, mutual-nearest-neighbour pairs, shared-neighbour counts, score threshold
and cell-type filter run on the GPU.  The returned frame has the reference's
columns, dtypes, row order and index labels.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from .. import anchor_search as _as
from ..preprocessing.anchors import SeuratIntegration, get_cca_adata_list


def label_filter(integrator, rna_labels, atac_labels, ct_filter=True, threshold=0.2, pair=(0, 1)):
    """Rows of ``integrator.anchor[pair]`` with ``score > threshold`` (and equal
    labels when ``ct_filter``), as scmiac.py:90-99 builds them; the masks and the
    compaction are computed on the device from the integrator's cached result."""
    rna_labels = np.asarray(rna_labels)
    atac_labels = np.asarray(atac_labels)
    cached = integrator.device_scores.get(pair)
    if cached is None:  # anchors were rescored on the host (k_filter / k_local): pandas path
        anchor_df = integrator.anchor[pair]
        anchor_df["x1_ct"] = rna_labels[anchor_df["x1"].values]
        anchor_df["x2_ct"] = atac_labels[anchor_df["x2"].values]
        anchor_df["is_same"] = anchor_df["x1_ct"].astype(str) == anchor_df["x2_ct"].astype(str)
        anchor_df = anchor_df[anchor_df["score"] > threshold].reset_index(drop=True)
        n_pass = anchor_df.shape[0]
        if ct_filter:
            anchor_df = anchor_df[anchor_df["is_same"]]
        return anchor_df, n_pass
    scores, n_pairs = cached
    codes1, codes2 = _as.label_codes(rna_labels, atac_labels)
    kept = _as.filter_anchors(scores, n_pairs, codes1, codes2, ct_filter=ct_filter, threshold=threshold)
    anchor_df = pd.DataFrame({"x1": kept.x1, "x2": kept.x2, "score": kept.score}, index=kept.index)
    anchor_df["x1_ct"] = rna_labels[kept.x1]
    anchor_df["x2_ct"] = atac_labels[kept.x2]
    anchor_df["is_same"] = kept.is_same
    return anchor_df, kept.n_pass_score


def find_anchors(adata_rna, adata_atac, all_nfeatures=3000, single_nfeatures=2000, k_anchor=5,
                 n_components=30, ct_filter=True, mode="d", rna_celltype_key="cell_type",
                 atac_celltype_key="pred"):
    """Find anchors between RNA and ATAC cells (same parameters and return value
    as the reference's ``find_anchors``; ``mode='v'`` pairs cells by position)."""
    if mode == "v":
        if adata_rna.shape[0] != adata_atac.shape[0]:
            raise ValueError(
                f"The number of RNA cells ({adata_rna.shape[0]}) "
                f"and ATAC cells ({adata_atac.shape[0]}) must be equal when mode='v'.")
        n = min(adata_rna.shape[0], adata_atac.shape[0])
        anchor_df = pd.DataFrame({"x1": np.arange(adata_rna.shape[0]), "x2": np.arange(adata_atac.shape[0]),
                                  "x1_ct": adata_rna.obs[rna_celltype_key].values[:n]})
        print(f"Directly using {anchor_df.shape[0]} pairing cells as anchors")
        return anchor_df

    adata_list = get_cca_adata_list([adata_rna, adata_atac], all_nfeatures=all_nfeatures,
                                    single_nfeatures=single_nfeatures)
    integrator = SeuratIntegration()
    integrator.find_anchor(adata_list, k_local=None, key_local="X_pca", k_anchor=k_anchor, key_anchor="X",
                           dim_red="cca", max_cc_cells=50000, k_score=30, scale1=True, scale2=True,
                           n_components=n_components, k_filter=None, n_features=200,
                           alignments=[[[0], [1]]])

    anchor_df, n_pass = label_filter(integrator, adata_rna.obs[rna_celltype_key].values,
                                     adata_atac.obs[atac_celltype_key].values, ct_filter=ct_filter)
    print(f"Number of anchor pairs: {n_pass}")
    if ct_filter:
        print(f"Number of anchors pairs after cell type filtering: {anchor_df.shape[0]}")
    rna_is_anchor_num = len(np.unique(anchor_df["x1"].values))
    atac_is_anchor_num = len(np.unique(anchor_df["x2"].values))
    print(f"Number of RNA anchors: {rna_is_anchor_num}; Number of ATAC anchors: {atac_is_anchor_num}")

    adata_rna.obsm["cca"] = integrator.U
    adata_atac.obsm["cca"] = integrator.V
    return anchor_df
