#!/usr/bin/env python3
"""Diagnostic: distance of the HIP gradients and of the float32 C restatement from the float64 restatement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from audio_mps_amd import HParams, PsiCMPS
from audio_mps_amd.scan import unpack_grad
from oracle import c_oracle as C
from _util import make_audio, c_oracle_run, rel_inf
for T in (4000, 16000, 65536):
    hp = HParams(minibatch_size=6, bond_dim=32)
    audio = make_audio(6, T, hp.delta_t, 5)
    m = PsiCMPS(hp, data_iterator=audio, seed=1)
    per = m.loss_per_clip()
    flat, _ = m.grad_sums(); g = unpack_grad(flat.cpu().numpy(), 32)
    r32 = c_oracle_run(m, audio, "f32", nthreads=6); g32 = C.unpack_grad(r32["grad"], 32)
    r64 = c_oracle_run(m, audio, "f64", nthreads=6); g64 = C.unpack_grad(r64["grad"], 32)
    den = np.maximum(np.abs(r64["loss_per_clip"]), 1)
    print(f"T={T}: loss  hip-f64 {np.max(np.abs(per - r64['loss_per_clip']) / den):.2e}  f32-f64 {np.max(np.abs(r32['loss_per_clip'] - r64['loss_per_clip']) / den):.2e}")
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        print(f"   {k:8s} hip-f64 {rel_inf(g[k], g64[k]):.2e}   f32oracle-f64 {rel_inf(g32[k], g64[k]):.2e}   hip-f32oracle {rel_inf(g[k], g32[k]):.2e}")
