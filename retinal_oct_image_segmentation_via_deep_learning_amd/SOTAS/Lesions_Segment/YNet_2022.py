"""Drop-in for SOTAS/Lesions_Segment/YNet_2022.py (reference :496-602): `UNet`, `get_model`.
Same constructor signatures, state_dict and output semantics; runs on the gfx950 HIP kernels."""
from ...unet import UNet, get_model  # noqa: F401

__all__ = ["UNet", "get_model"]
