"""Drop-in for the reference's SOTAS/Layers_Segment/MGUNet_2021.py (SURVEY.md §8 a9 and the §8(b) constructor list):
`Basconv` (:29-39), `UnetConv` (:42-70), `UnetUp` (:72-89), `UnetUp4` (:91-108), `GloRe_Unit` (:110-148), `MGR_Module`
(:150-194), `MGUNet` (:197-252), `MGUNet_2` (:255-309), `init_weights` and the three `weights_init_*` (:314-352)."""
from ...blocks import (UnetConv, UnetUp, UnetUp4, init_weights, weights_init_kaiming,  # noqa: F401
                       weights_init_normal, weights_init_xavier)
from ...mgunet import Basconv, GloRe_Unit, MGR_Module, MGUNet, MGUNet_2  # noqa: F401

__all__ = ["Basconv", "UnetConv", "UnetUp", "UnetUp4", "GloRe_Unit", "MGR_Module", "MGUNet", "MGUNet_2", "init_weights",
           "weights_init_normal", "weights_init_xavier", "weights_init_kaiming"]
