#!/usr/bin/env python3
"""Runs the forward / reverse scans several times on the same input and checks that every run gives the same bits
(usage: check_determinism.py D T B variant [runs])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from audio_mps_amd import HParams, PsiCMPS
from audio_mps_amd.scan import HipScan
from audio_mps_amd.data import damped_sine

D, T, B, variant = (int(a) for a in sys.argv[1:5])
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 4
hp = HParams(minibatch_size=B, bond_dim=D)
rng = np.random.default_rng(7)
x = damped_sine(B, T, hp.delta_t, seed=11)
audio = torch.from_numpy((x + 0.02 * rng.standard_normal(x.shape)).astype(np.float32)).cuda()
be = HipScan(D, variant=variant)
m = PsiCMPS(hp, seed=0, backend=be)
be.set_params(m.effective_params(), B, T, train=True)
ref = None
for r in range(runs):
    be.forward(audio, save_for_bwd=True)
    be.backward()
    torch.cuda.synchronize()
    cur = (be._loss.clone().cpu().numpy(), be._grad.clone().cpu().numpy())
    if ref is None:
        ref = cur
    else:
        same_l = np.array_equal(ref[0], cur[0]); same_g = np.array_equal(ref[1], cur[1])
        print(f"run {r}: loss bits equal {same_l}, gradient bits equal {same_g}",
              "" if same_l and same_g else f" max |dloss| {np.max(np.abs(ref[0]-cur[0])):.3e} max |dgrad| {np.max(np.abs(ref[1]-cur[1])):.3e}")
print("mean loss", float(ref[0].mean()))
