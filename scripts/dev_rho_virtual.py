#!/usr/bin/env python3
"""dev: RhoCMPS reverse sweep on virtual clips vs the GEMM reverse sweep vs the oracle, all gradient tensors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from audio_mps_amd import _capi
from oracle import cmps_oracle as O
from test_gpu_rho import _rho_model, _oracle_side
from _util import rel_inf
for (D, T, B, rank, sigma, rscale) in [(32, 96, 3, 32, 1e-4, None), (32, 150, 3, 17, 0.3, 0.4), (20, 129, 2, 9, 0.2, 0.5), (32, 64, 1, 9, 1e-4, None)]:
    m, audio = _rho_model(D, T, B, rank=rank, sigma=sigma, seed=D + T, rscale=rscale)
    be = m._get_backend()
    ohp, ov, Wx, Wy = _oracle_side(m)
    ref64 = O.rho_loss_and_grads(ohp, ov.astype(np.float64), Wx.astype(np.float64), Wy.astype(np.float64), audio, "f64")
    for opt in (0, 1):
        _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_RHO_BWD, opt))
        _, g = m.loss_and_grads()
        print((D, T, B, rank, sigma), "virtual" if opt == 0 else "gemm   ", {k: f"{rel_inf(g[k], ref64[k]):.2e}" for k in ("A", "Rx", "Ry", "freqs", "Wx", "Wy")},
              "A", float(g["A"]), float(ref64["A"]))
