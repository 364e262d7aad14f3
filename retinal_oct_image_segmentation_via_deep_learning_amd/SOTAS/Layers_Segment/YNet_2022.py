"""Drop-in for SOTAS/Layers_Segment/YNet_2022 (reference file has no .py suffix; :33-139):
`UNet`, `get_model` -- the "Layers_Segment U-Net" of BASELINE configs 2/3."""
from ...unet import UNet, get_model  # noqa: F401

__all__ = ["UNet", "get_model"]
