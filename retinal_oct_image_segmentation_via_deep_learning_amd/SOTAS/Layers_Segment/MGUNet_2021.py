"""Drop-in for the block family of the reference's SOTAS/Layers_Segment/MGUNet_2021.py:
`UnetConv` (:42-70), `UnetUp` (:72-89), `UnetUp4` (:91-108), `init_weights` and the three
`weights_init_*` (:314-352).  The full MGUNet / MGUNet_2 networks add the graph-reasoning module
(`MGR_Module`, `GloRe_Unit`, :110-196), which is outside the accelerated path."""
from ...blocks import (UnetConv, UnetUp, UnetUp4, init_weights, weights_init_kaiming,  # noqa: F401
                       weights_init_normal, weights_init_xavier)

__all__ = ["UnetConv", "UnetUp", "UnetUp4", "init_weights", "weights_init_normal", "weights_init_xavier",
           "weights_init_kaiming"]
