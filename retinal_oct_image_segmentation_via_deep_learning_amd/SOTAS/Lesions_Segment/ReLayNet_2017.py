"""Drop-in for SOTAS/Lesions_Segment/ReLayNet_2017.py: `ReLayNet` (:21-126) and its blocks (:133-203)."""
from ...relaynet import BasicBlock, ClassifierBlock, DecoderBlock, EncoderBlock, ReLayNet  # noqa: F401

__all__ = ["ReLayNet", "BasicBlock", "EncoderBlock", "DecoderBlock", "ClassifierBlock"]
