"""Evaluation metrics that are k-NN or pairwise-distance work (``scmiac/evaluation/benchmark.py``)."""
from .benchmark import foscttm, partially_matched_foscttm, neighbor_conservation, knn_matching

__all__ = ["foscttm", "partially_matched_foscttm", "neighbor_conservation", "knn_matching"]
