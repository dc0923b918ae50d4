#!/usr/bin/env python3
"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference (TensorFlow 1.x) cannot run here, so the vectors come from the float32 op-for-op numpy
restatement oracle/cmps_oracle.py ("parity unpinned", see its header): inputs (raw variables, audio,
hyper-parameters) and expected outputs (per-clip loss, gradients w.r.t. effective parameters and raw
variables, float32 and float64-twin values).  Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cmps_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

TEST_HP = dict(minibatch_size=8, bond_dim=7, delta_t=1 / 16000, sigma=0.0001, A=100.0,
               h_reg=2 / (math.pi * 16000) ** 2, r_reg=2 / (math.pi * 16000))   # tests/test_model.py:13-14

CASES = {
    # name: (hparams overrides, T, seed, noise, R scale (None = reference init))
    "c1_d4_t256_b8": (dict(minibatch_size=8, bond_dim=4), 256, 1, 0.02, None),               # BASELINE configs[0] shape
    "reftest_d7_t256_b8": (TEST_HP, 256, 2, 0.0, None),                                      # reference test hparams
    "c2slice_d16_t1024_b8": (dict(minibatch_size=8, bond_dim=16), 1024, 3, 0.02, None),      # configs[1] slice
    "c3slice_d32_t2000_b4": (dict(minibatch_size=4, bond_dim=32), 2000, 4, 0.02, None),      # configs[2] slice
    "sigma1_d5_t200_b3": (dict(minibatch_size=3, bond_dim=5, sigma=0.7, A=3.0), 200, 5, 0.05, 0.3),
    "qubit_d2_t128_b2": (dict(minibatch_size=2, bond_dim=2, sigma=1.0, A=1.0,
                              h_reg=2 / (math.pi * 16000) ** 2, r_reg=2 / (math.pi * 16000) ** 2), 128, 6, 0.01, "qubit"),
}


def make_case(name):
    hpo, T, seed, noise, rscale = CASES[name]
    hp = O.HParams(**hpo)
    if rscale == "qubit":      # tests/test_model.py:147-151: R = [[0,1],[0,0]], freqs = [10,-10]
        var = O.init_variables(hp, seed=seed, R_in=np.array([[0, 1], [0, 0]], dtype=np.complex64),
                               freqs_in=np.array([10.0, -10.0], dtype=np.float32))
    else:
        var = O.init_variables(hp, seed=seed)
        if rscale is not None:
            var.Rx *= np.float32(rscale)
            var.Ry *= np.float32(rscale)
    rng = np.random.default_rng(seed + 12345)
    data = O.damped_sine(hp.minibatch_size, T, hp.delta_t, seed=seed)
    data = (data + noise * rng.standard_normal(data.shape)).astype(np.float32)
    g32 = O.psi_loss_and_grads(hp, var, data, "f32")
    g64 = O.psi_loss_and_grads(hp, var.astype(np.float64), data, "f64")
    out = {f"hp_{k}": np.asarray(v if v is not None else -1) for k, v in vars(hp).items()}
    out.update({f"var_{k}": np.asarray(getattr(var, k)) for k in O.Variables.NAMES})
    out["scaled_R"] = np.asarray(var.scaled_R)
    out["scaled_freqs"] = np.asarray(var.scaled_freqs)
    out["data"] = data
    for tag, g in (("f32", g32), ("f64", g64)):
        out[f"loss_per_clip_{tag}"] = g.per_clip
        out[f"loss_{tag}"] = np.asarray(g.loss)
        for k in O.Variables.NAMES:
            out[f"grad_{k}_{tag}"] = np.asarray(getattr(g, k))
        out[f"eff_Rbar_{tag}"] = g.eff["Rbar"]
        out[f"eff_fbar_{tag}"] = g.eff["fbar"]
        out[f"eff_psi0bar_{tag}"] = g.eff["psi0bar"]
        out[f"eff_Abar_{tag}"] = np.asarray(g.eff["Abar"])
    return out


def main():
    for name in CASES:
        out = make_case(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: loss {float(out['loss_f32']):.6f} (f64 {float(out['loss_f64']):.6f})  -> {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
