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


# ---------------------------------------------------------------------------------------------------
# golden fixtures
# ---------------------------------------------------------------------------------------------------
import glob
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def golden_hparams(g, cls):
    """hyper-parameters of a fixture as an instance of ``cls`` (the product's or the oracle's HParams)."""
    kw = {}
    for k in ("minibatch_size", "bond_dim", "delta_t", "sigma", "h_reg", "r_reg", "initial_rank", "A", "learning_rate"):
        v = g["hp_" + k].item()
        if k in ("minibatch_size", "bond_dim"):
            v = int(v)
        if k == "initial_rank":
            v = None if v == -1 else int(v)
        kw[k] = v
    return cls(**kw)


def golden_oracle_variables(g):
    return O.Variables(*(np.array(g["var_" + k]) for k in O.Variables.NAMES),
                       scaled_R=bool(g["scaled_R"]), scaled_freqs=bool(g["scaled_freqs"]))


def model_from_golden(g, backend=None):
    """The product's PsiCMPS holding exactly the fixture's raw variables."""
    from audio_mps_amd import HParams, PsiCMPS
    hp = golden_hparams(g, HParams)
    D = hp.bond_dim
    kw = {}
    if not bool(g["scaled_R"]):
        kw["R_in"] = (g["var_Rx"] + 1j * g["var_Ry"]).astype(np.complex64)
    if not bool(g["scaled_freqs"]):
        kw["freqs_in"] = g["var_freqs"].astype(np.float32)
    m = PsiCMPS(hp, data_iterator=g["data"], backend=backend, **kw)
    for k in O.Variables.NAMES:
        m.variables[k] = np.array(g["var_" + k], dtype=np.float32)
    assert m.variables["Rx"].shape == (D, D)
    return m


class OracleBackend:
    """A stand-in for HipScan built on the oracle, used ONLY by CPU tests of the host logic (chain rule, Adam,
    trainer, data-parallel reduction).  Same two methods, host tensors instead of device tensors."""
    name = "oracle"
    device = None

    def __init__(self, D, dtype="f32"):
        self.D, self.dtype = D, dtype

    def set_params(self, p, B, T, train=True):
        self.p = p

    def _run(self, audio, want_grad):
        import torch
        a = audio.numpy() if isinstance(audio, torch.Tensor) else np.asarray(audio)
        return C.psi_scan(a, self.p.R, self.p.freqs, self.p.psi0, self.p.A, self.p.delta_t, self.p.sigma,
                          self.dtype, want_grad=want_grad, nthreads=2)

    def forward(self, audio, save_for_bwd=False):
        import torch
        self._audio = audio
        return torch.from_numpy(self._run(audio, False)["loss_per_clip"].astype(np.float32))

    def loss_and_grad_sums(self, audio, check=False):
        import torch
        out = self._run(audio, True)
        return (torch.from_numpy(out["loss_per_clip"].astype(np.float32)),
                torch.from_numpy(out["grad"].astype(np.float32)))


    # ---- RhoCMPS stand-in: the numpy oracle's effective-parameter adjoint, repacked like cmps_rho_loss_bwd ----
    def rho_set_state(self, phi, B, T, train=True):
        self.phi = np.asarray(phi, dtype=np.complex128)

    def _rho_model(self, audio):
        import torch
        a = audio.numpy() if isinstance(audio, torch.Tensor) else np.asarray(audio)
        D = self.D
        hp = O.HParams(minibatch_size=a.shape[0], bond_dim=D, delta_t=self.p.delta_t, sigma=self.p.sigma, A=self.p.A,
                       initial_rank=self.phi.shape[0])
        R = np.asarray(self.p.R)
        # effective parameters go in untouched: R_in / freqs_in style variables (no scaling, R has a zero diagonal)
        var = O.Variables(np.float32(self.p.A), R.real.astype(np.float32), R.imag.astype(np.float32),
                          np.asarray(self.p.freqs, dtype=np.float32), np.zeros(D, np.float32), np.zeros(D, np.float32),
                          scaled_R=False, scaled_freqs=False)
        W = np.conj(self.phi)          # rho_0 = W^dagger W / tr with tr = 1
        return hp, var, W.real.astype(np.float32), W.imag.astype(np.float32), a

    def rho_forward(self, audio, save_for_bwd=False):
        import torch
        hp, var, Wx, Wy, a = self._rho_model(audio)
        return torch.from_numpy(O.rho_loss_per_clip(hp, var, Wx, Wy, a, self.dtype).astype(np.float32))

    def rho_loss_and_grad_sums(self, audio):
        import torch
        hp, var, Wx, Wy, a = self._rho_model(audio)
        dt = self.dtype
        g = O.rho_loss_and_grads(hp, var if dt == "f32" else var.astype(np.float64), Wx.astype(np.float64 if dt == "f64" else np.float32),
                                 Wy.astype(np.float64 if dt == "f64" else np.float32), a, dt)
        B, D, r = a.shape[0], self.D, self.phi.shape[0]
        eff = g["eff"]
        Rbar = eff["Rbar"] * B
        r0b = eff["rho0bar"] * B
        phibar = (self.phi @ (r0b + np.conj(r0b.T)).T)           # phibar_a = (rho0bar + rho0bar^dagger) phi_a
        flat = np.concatenate([Rbar.real.ravel(), Rbar.imag.ravel(), eff["fbar"] * B, np.zeros(2 * D),
                               [eff["Abar"] * B, float(np.sum(g["per_clip"]))], phibar.real.ravel(), phibar.imag.ravel()])
        return torch.from_numpy(g["per_clip"].astype(np.float32)), torch.from_numpy(flat.astype(np.float32))
