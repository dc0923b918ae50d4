#!/usr/bin/env python3
"""Diagnostic (round 5): the three RhoCMPS draws of scripts/random_sweep.py seeds 31 / 32 whose frequency gradient sits 1.1e-4 ... 2.0e-4
from the float64 oracle -- replayed with the two-wave, the one-wave and the GEMM reverse sweeps, next to the float32 oracle's own distance
(the same numbers: float32 conditioning at tiny R and long clips, not a kernel).  Run on a GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import _sweep as S
from _sweep import O, rel_inf, make_audio
from audio_mps_amd import HParams, RhoCMPS, _capi
cfgs = [(27, 19, 370, 1, 0.006, 0.0052, 0.33717), (30, 26, 384, 2, 0.0047, 0.1709, 0.37508), (11, 11, 375, 4, 0.026, 0.0019, 0.00057)]
# replay the sweep's generator to hit the same draws
for seed in (31, 32):
    rng = np.random.default_rng(seed * 1000 + list(S.FAMILIES).index("rho"))
    n = S.DEFAULT_COUNTS["rho"]
    for it in range(n):
        D = int(rng.integers(2, 41)); r = int(rng.integers(1, min(D, 32) + 1)); T = int(rng.integers(2, 150)); B = int(rng.integers(1, 6))
        rs, amp = 0.4, 1.0
        kind = it % 12 if n < 34 else (3 if it in (3, 7, 11) else 11 if it >= 31 else 5 if it >= 14 else 0)
        if kind == 3:
            D = int(rng.integers(33, 129)); r = int(rng.integers(1, min(D, 40) + 1)); T = int(rng.integers(2, 40)); B = int(rng.integers(1, 4))
        if kind == 11:
            D = int(rng.integers(72, 129)); r = D; T = int(rng.integers(2, 12)); B = int(rng.integers(1, 3))
        elif kind >= 5:
            D = int(rng.integers(9, 33)); r = int(rng.integers(9, D + 1)); T = int(rng.integers(2, 400))
            rs = float(10 ** rng.uniform(-2.5, 0.2)); amp = float(10 ** rng.uniform(-3, 0.7))
        hp = HParams(minibatch_size=B, bond_dim=D, initial_rank=r, sigma=float(10 ** rng.uniform(-4, -0.3)))
        cfg = (D, r, T, B, round(rs, 4), round(amp, 4), round(hp.sigma, 5))
        if cfg not in cfgs:
            continue
        audio = (make_audio(B, T, hp.delta_t, 200 + it) * np.float32(amp)).astype(np.float32)
        if kind >= 5 and it % 4 == 0 and T > 30:
            audio[:, T // 2:] = audio[:, T // 2: T // 2 + 1]
        m = RhoCMPS(hp, data_iterator=audio, seed=it)
        m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
        ov = O.Variables(np.asarray(m.variables["A"], np.float32), m.variables["Rx"], m.variables["Ry"], m.variables["freqs"],
                         np.zeros(D, np.float32), np.zeros(D, np.float32), scaled_R=True, scaled_freqs=True)
        ref = O.rho_loss_and_grads(O.HParams(**hp.values()), ov.astype(np.float64), m.variables["Wx"].astype(np.float64), m.variables["Wy"].astype(np.float64), audio, "f64")
        ref32 = O.rho_loss_and_grads(O.HParams(**hp.values()), ov, m.variables["Wx"], m.variables["Wy"], audio, "f32")
        print("cfg", cfg, "it", it)
        be = m._get_backend()
        for waves, rbwd in ((2, 0), (1, 0), (1, 1)):
            _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_BWD_WAVES, waves))
            _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_RHO_BWD, rbwd))
            loss, g = m.loss_and_grads()
            print(f"  waves {waves} rho_bwd {'gemm' if rbwd else 'virtual'}:", {k: f"{rel_inf(g[k], ref[k]):.2e}" for k in ("Rx", "Ry", "freqs", "Wx", "Wy", "A")})
        print("  f32 oracle vs f64:        ", {k: f"{rel_inf(ref32[k], ref[k]):.2e}" for k in ("Rx", "Ry", "freqs", "Wx", "Wy", "A")})
