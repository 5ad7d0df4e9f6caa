"""Core API (drop-in for the anchor part of ``scmiac.modeling.scmiac``) and the
device-side consumer of the anchor table (loaders, contrastive loss)."""
from .scmiac import find_anchors
from .loss import NTXentLoss2
from .dataset import DeviceAnchorLoader, DeviceMultiOmicLoader, anchor_frame, make_loaders

__all__ = ["find_anchors", "NTXentLoss2", "DeviceAnchorLoader", "DeviceMultiOmicLoader", "anchor_frame", "make_loaders"]
