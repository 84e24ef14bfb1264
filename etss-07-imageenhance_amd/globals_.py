"""Enums mirrored from src/mon/globals.py:649-699 (only the members the hot path uses)."""
from enum import Enum


class Task(Enum):
    LLIE = "llie"
    DENOISE = "denoise"
    DERAIN = "derain"
    DEBLUR = "deblur"
    DESNOW = "desnow"
    LES = "les"
    SEGMENT = "segment"
    DEHAZE = "dehaze"
    UIE = "uie"


class Scheme(Enum):
    INFERENCE = "inference"
    INSTANCE = "instance"
    SUPERVISED = "supervised"
    TRADITIONAL = "traditional"
    UNSUPERVISED = "unsupervised"
    ZERO_REFERENCE = "zero_reference"
    ZERO_SHOT = "zero_shot"


WEIGHTS_FILE_FORMATS = [".pt", ".pth", ".onnx"]  # src/mon/globals.py:709
