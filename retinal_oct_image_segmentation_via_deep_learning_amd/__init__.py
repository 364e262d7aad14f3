"""MI355X-native U-Net training path for retinal OCT segmentation (drop-in for the hot path of
ZhangHH233/Retinal_OCT_Image_Segmentation_via_Deep_Learning: SOTAS/{Layers,Lesions}_Segment U-Net
forward/backward, loss head, Metrics/{Region,ConfusionMatrix}_based reductions).

Python here is host plumbing; the arithmetic lives in liboct_hip.so (csrc/, include/oct_hip.h).
"""
from ._lib import OctError, build, lib  # noqa: F401
from .unet import UNet, get_model  # noqa: F401

__all__ = ["UNet", "get_model", "OctError", "build", "lib"]
