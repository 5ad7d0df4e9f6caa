"""scmiac_amd: MI355X-native anchor search for scMIAC (k-NN / mutual-nearest-
neighbour anchors, shared-neighbour scoring, cell-type filter).

``find_anchors`` and ``scmiac_amd.preprocessing.anchors`` keep the call
signatures of the reference package's ``scmiac.find_anchors`` and
``scmiac.preprocessing.anchors``.
"""
from .modeling.scmiac import find_anchors

__all__ = ("find_anchors",)
__version__ = "0.1.0"
