#!/usr/bin/env python3
"""Times RhoCMPS forward / backward (HIP events) at a stated shape: time_rho.py [rank] [T] [B] [variant] [D]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from audio_mps_amd import HParams, RhoCMPS
from audio_mps_amd.scan import HipScan
from _util import make_audio
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
D = int(sys.argv[5]) if len(sys.argv) > 5 else 32
hp = HParams(minibatch_size=B, bond_dim=D, initial_rank=rank)
a = make_audio(B, T, hp.delta_t, 2)
rm = RhoCMPS(hp, data_iterator=a, seed=2, backend=HipScan(D, variant=variant))
if D > 32:
    rm.variables["Rx"] *= np.float32(0.5); rm.variables["Ry"] *= np.float32(0.5)
d_a = rm._to_device(a)
rb = rm._prepare(B, T, train=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
f, bw = [], []
for r in range(4):
    ev[0].record(); rb.rho_forward(d_a, save_for_bwd=True); ev[1].record(); rb.rho_backward(); ev[2].record(); torch.cuda.synchronize()
    if r: f.append(ev[0].elapsed_time(ev[1])); bw.append(ev[1].elapsed_time(ev[2]))
print(f"D {D} rank {rank} T {T} B {B} variant {variant}: fwd {np.median(f):.3f} ms  bwd {np.median(bw):.3f} ms  loss {float(rb._loss.mean()):.6f}")
