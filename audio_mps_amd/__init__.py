"""audio_mps_amd: MI355X (gfx950) implementation of audio-mps's cMPS log-likelihood scan.

Python host code mirrors the reference's model / trainer surface (model.py, train.py,
training_estimators.py) and drives hand-written HIP kernels (csrc/) through the C ABI in include/cmps.h.
Importing the package does not touch the GPU; the HIP library is loaded on first use and there is no
CPU fallback.
"""
from .model import HParams, CMPS, PsiCMPS, RhoCMPS, AudioMPS, LegacyAudioMPS  # noqa: F401

__all__ = ["HParams", "CMPS", "PsiCMPS", "RhoCMPS", "AudioMPS", "LegacyAudioMPS"]
