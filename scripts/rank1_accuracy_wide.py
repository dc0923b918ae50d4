#!/usr/bin/env python3
"""Gate of the fp16 split (CMPS_RANK1_F16X2) in the wide kernels' gradient GEMM: distance of each rank-1 mode's R / Q sums
to BF16X3's and to the float64 restatement, over clip lengths (short clips: few accumulations per sum, so the product arithmetic
shows; long clips: float32 accumulation noise, common to all modes, dominates).
usage: python scripts/rank1_accuracy_wide.py [D]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from _util import c_oracle_run, make_audio, rel_inf      # noqa: E402
from oracle import c_oracle as C                          # noqa: E402
from audio_mps_amd import HParams, PsiCMPS                # noqa: E402
from audio_mps_amd.scan import HipScan, unpack_grad       # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
print(f"D = {D}; rel_inf = max |x - ref| / max |ref| on the R gradient sums (k_grad_gemm + k_finalize)")
print(f"{'T':>6} {'clips':>5} {'amp':>6} | {'x2 - f64':>9} {'x3 - f64':>9} {'f16 - f64':>9} | {'x2 - x3':>9} {'f16 - x3':>9}")
for T, B, amp in ((5, 2, 1.0), (9, 2, 1.0), (33, 2, 1.0), (129, 4, 1.0), (513, 4, 1.0), (2000, 6, 1.0), (8000, 4, 1.0),
                  (33, 2, 1e-3), (2000, 6, 1e-3), (33, 2, 4.0), (2000, 6, 4.0)):
    hp = HParams(minibatch_size=B, bond_dim=D)
    audio = (make_audio(B, T, hp.delta_t, T) * np.float32(amp)).astype(np.float32)
    m = PsiCMPS(hp, data_iterator=audio, seed=T, backend=HipScan(D, variant=5))
    m.variables["Rx"] *= np.float32(0.35 if D > 64 else 1.0)
    m.variables["Ry"] *= np.float32(0.35 if D > 64 else 1.0)
    ref = C.unpack_grad(c_oracle_run(m, audio, "f64", nthreads=8)["grad"], D)["Rbar"]
    out = {}
    for mode in (1, 2, 3):
        m._get_backend().set_rank1(mode)
        out[mode] = unpack_grad(m.grad_sums(audio)[0].cpu().numpy(), D)["Rbar"].astype(np.complex128)
    e = {k: rel_inf(v, ref) for k, v in out.items()}
    print(f"{T:6d} {B:5d} {amp:6g} | {e[1]:9.2e} {e[2]:9.2e} {e[3]:9.2e} | {rel_inf(out[1], out[2]):9.2e} {rel_inf(out[3], out[2]):9.2e}")
