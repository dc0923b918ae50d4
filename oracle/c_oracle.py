"""ctypes front end of oracle/cmps_oracle.c (plain-C restatement; TEST INFRASTRUCTURE ONLY --
see the header of cmps_oracle.c; parity unpinned, as stated there).

``psi_scan(...)`` takes EFFECTIVE parameters (R after model.py:42, scaled freqs, normalised psi_0),
i.e. what the numpy oracle's ``effective_params`` / ``psi_0`` return, and runs rows a3-a9 of
SURVEY.md section 8 one clip at a time, optionally on several OpenMP threads.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/cmps_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cmps_oracle.c"))):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for name, real in (("cmps_oracle_psi_f32", ctypes.c_float), ("cmps_oracle_psi_f64", ctypes.c_double)):
            fn = getattr(_lib, name)
            p = ctypes.c_void_p
            fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, p, p, p, p, p, p, real,
                           ctypes.c_double, ctypes.c_double, p, p, p, ctypes.c_int]
            fn.restype = ctypes.c_int
    return _lib


def grad_size(D: int) -> int:
    return 2 * D * D + 3 * D + 2


def psi_scan(data, R, freqs, psi0, A, delta_t, sigma, dtype="f32", want_grad=False,
             want_states=False, nthreads=0):
    """Returns dict(loss_per_clip [B], grad (flat sums, see cmps_oracle.c) or None, states or None)."""
    real = np.float32 if dtype == "f32" else np.float64
    lib = _load()
    fn = lib.cmps_oracle_psi_f32 if dtype == "f32" else lib.cmps_oracle_psi_f64
    data = np.ascontiguousarray(data, dtype=real)
    B, T = data.shape
    R = np.asarray(R)
    D = R.shape[0]
    Rre = np.ascontiguousarray(R.real, dtype=real)
    Rim = np.ascontiguousarray(R.imag, dtype=real)
    f = np.ascontiguousarray(freqs, dtype=real)
    psi0 = np.asarray(psi0)
    p0r = np.ascontiguousarray(psi0.real, dtype=real)
    p0i = np.ascontiguousarray(psi0.imag, dtype=real)
    loss = np.zeros(B, dtype=real)
    grad = np.zeros(grad_size(D), dtype=real) if want_grad else None
    states = np.zeros((B, T - 1, D, 2), dtype=real) if want_states else None
    ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    rc = fn(B, T, D, ptr(data), ptr(Rre), ptr(Rim), ptr(f), ptr(p0r), ptr(p0i), real(A),
            float(delta_t), float(sigma), ptr(loss), ptr(grad), ptr(states), int(nthreads))
    if rc != 0:
        raise MemoryError("cmps_oracle_psi failed (allocation)")
    out = {"loss_per_clip": loss, "grad": grad, "states": None}
    if states is not None:
        out["states"] = states[..., 0] + 1j * states[..., 1]
    return out


def unpack_grad(grad, D):
    """flat sums -> dict(Rbar complex [D,D], fbar [D], psi0bar complex [D], Abar, loss_sum)."""
    g = np.asarray(grad)
    DD = D * D
    return {"Rbar": (g[:DD] + 1j * g[DD:2 * DD]).reshape(D, D),
            "fbar": g[2 * DD:2 * DD + D],
            "psi0bar": g[2 * DD + D:2 * DD + 2 * D] + 1j * g[2 * DD + 2 * D:2 * DD + 3 * D],
            "Abar": g[2 * DD + 3 * D], "loss_sum": g[2 * DD + 3 * D + 1]}
