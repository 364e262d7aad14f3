"""Drop-in for SOTAS/Layers_Segment/SD_Layer_Net/common.py:6-41,64-91."""
from ....blocks import Attention_block, conv_block, up_conv  # noqa: F401

__all__ = ["conv_block", "up_conv", "Attention_block"]
