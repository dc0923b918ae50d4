#!/usr/bin/env python3
"""Development check of the MFMA pair kernels (bond dimension 128) against the bf16-emulating and the float32 oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from audio_mps_amd import HParams, PsiCMPS
from audio_mps_amd.scan import HipScan, unpack_grad
from oracle import cmps_oracle as O, c_oracle as C
from _util import make_audio, oracle_hparams, oracle_variables, c_oracle_run

for (D, T, B) in [(40, 60, 3), (64, 300, 5), (80, 100, 2), (96, 150, 4), (100, 70, 3), (128, 300, 5)]:
    hp = HParams(minibatch_size=B, bond_dim=D)
    audio = make_audio(B, T, hp.delta_t, 7)
    m = PsiCMPS(hp, data_iterator=audio, seed=3, backend=HipScan(D, variant=3))
    per = m.loss_per_clip()
    flat, _ = m.grad_sums()
    g = unpack_grad(flat.cpu().numpy(), D)
    em = O.psi_bf16_scan(oracle_hparams(hp), oracle_variables(m), audio, want_grad=True)
    ref = c_oracle_run(m, audio, "f32", want_grad=True)
    gr = C.unpack_grad(ref["grad"], D)
    den = np.maximum(np.abs(ref["loss_per_clip"]), 1.0)
    print(f"D={D} T={T} B={B}: loss hip vs bf16-emulation {np.max(np.abs(per - em['loss_per_clip']) / den):.3e}   hip vs f32 {np.max(np.abs(per - ref['loss_per_clip']) / den):.3e}")
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        a, e_, r_ = np.asarray(g[k]), np.asarray(em[k]), np.asarray(gr[k])
        print(f"   {k:8s} hip vs emulation {np.max(np.abs(a - e_)) / max(np.max(np.abs(e_)), 1e-30):.3e}   hip vs f32 {np.max(np.abs(a - r_)) / max(np.max(np.abs(r_)), 1e-30):.3e}   emul vs f32 {np.max(np.abs(e_ - r_)) / max(np.max(np.abs(r_)), 1e-30):.3e}")
    print("   loss_sum", g["loss_sum"], gr["loss_sum"])
