"""Same classes as SOTAS/Layers_Segment/SD_Layer_Net/unet.py (BASELINE looks for AttU_Net here)."""
from ....blocks import AttU_Net, AttU_Net4, U_Net  # noqa: F401

__all__ = ["U_Net", "AttU_Net", "AttU_Net4"]
