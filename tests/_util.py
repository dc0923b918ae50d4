"""Shared helpers for the parity tests: build the same model in the product (audio_mps_amd) and in the
oracle (oracle/), from one set of raw variables."""
from __future__ import annotations

import numpy as np

from oracle import cmps_oracle as O
from oracle import c_oracle as C


def oracle_hparams(hp) -> O.HParams:
    return O.HParams(**hp.values())


def oracle_variables(model) -> O.Variables:
    v = model.variables
    return O.Variables(np.asarray(v["A"], dtype=np.float32), v["Rx"].copy(), v["Ry"].copy(), v["freqs"].copy(),
                       v["psi_x"].copy(), v["psi_y"].copy(),
                       scaled_R=float(model._c_r) != 1.0, scaled_freqs=float(model._c_h) != 1.0)


def make_audio(B, T, delta_t, seed, noise=0.02):
    """damped sine (data.py:8-22) plus a little white noise so that every step carries signal."""
    rng = np.random.default_rng(seed + 12345)
    x = O.damped_sine(B, T, delta_t, seed=seed)
    return (x + noise * rng.standard_normal(x.shape)).astype(np.float32)


def c_oracle_run(model, audio, dtype="f32", want_grad=True, nthreads=0):
    """Run the C oracle on the model's effective parameters (as computed by the ORACLE's own a1/a2)."""
    ohp = oracle_hparams(model.hparams)
    ov = oracle_variables(model)
    dt = dtype
    R, f, _, _ = O.effective_params(ohp, ov if dt == "f32" else ov.astype(np.float64), dt)
    p0 = O.psi_0(ov if dt == "f32" else ov.astype(np.float64), dt)
    return C.psi_scan(audio, R, f, p0, float(ov.A), ohp.delta_t, ohp.sigma, dt, want_grad=want_grad,
                      nthreads=nthreads)


def rel_inf(a, b):
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64)
    b = np.asarray(b, dtype=a.dtype)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
