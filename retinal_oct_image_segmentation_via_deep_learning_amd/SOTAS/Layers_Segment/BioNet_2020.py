"""Drop-in for the `UNet` of the reference's SOTAS/Layers_Segment/BioNet_2020.py:24-75.

`BioRegularization` / `BioNet` (a torchvision ResNet-18 regulariser around two of these U-Nets,
BioNet_2020.py:77-130) are outside the hot path: torchvision is not in this image and SURVEY.md
§8 scopes the path to the U-Net itself.
"""
from ...unet import BioUNet as UNet

__all__ = ["UNet"]
