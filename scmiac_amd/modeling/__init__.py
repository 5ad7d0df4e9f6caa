"""Core API (drop-in for the anchor part of ``scmiac.modeling.scmiac``)."""
from .scmiac import find_anchors

__all__ = ["find_anchors"]
