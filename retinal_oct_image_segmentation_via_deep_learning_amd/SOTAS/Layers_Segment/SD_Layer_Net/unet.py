"""Drop-in for SOTAS/Layers_Segment/SD_Layer_Net/unet.py: `U_Net` (:8-74) and `AttU_Net` (:76-150, BASELINE cfg4)."""
from ....blocks import AttU_Net, U_Net  # noqa: F401

__all__ = ["U_Net", "AttU_Net"]
