"""Anchor engine (drop-in for ``scmiac.preprocessing.anchors``)."""
from .anchors import SeuratIntegration, get_cca_adata_list, find_neighbor, find_mnn, score_anchor, min_max

__all__ = ["SeuratIntegration", "get_cca_adata_list", "find_neighbor", "find_mnn", "score_anchor", "min_max"]
