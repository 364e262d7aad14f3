"""BASELINE configs[3] names the attention-gated U-Net under SOTAS/Lesions_Segment; the reference ships it under
SOTAS/Layers_Segment/SD_Layer_Net (unet.py:76-150, common.py:6-91).  Exported under both packages."""
from ....blocks import Attention_block, AttU_Net, AttU_Net4, U_Net, conv_block, up_conv  # noqa: F401

__all__ = ["AttU_Net", "AttU_Net4", "U_Net", "Attention_block", "conv_block", "up_conv"]
