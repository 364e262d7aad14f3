"""Drop-in for SOTAS/Layers_Segment/SD_Layer_Net/unet.py: `U_Net` (:8-74), `AttU_Net` (:76-150, BASELINE cfg4) and `AttU_Net4` (:153-214)."""
from ....blocks import AttU_Net, AttU_Net4, U_Net  # noqa: F401

__all__ = ["U_Net", "AttU_Net", "AttU_Net4"]
