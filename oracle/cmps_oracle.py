"""CPU restatement (numpy) of audio-mps's PsiCMPS log-likelihood scan and its gradient.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker.  The product path (``audio_mps_amd``) never imports this module and has no CPU fallback.

PARITY UNPINNED.  The reference arithmetic lives in TensorFlow 1.x (``requirements.txt:1``, unpinned),
which is not importable here (ordinary ``ModuleNotFoundError``; no wheel, no network), and the
reference's own tests (``tests/test_model.py``) hold structural invariants only -- no golden vectors,
no known-answer values.  This file therefore restates ``model.py`` op for op from its text, in the
same dtypes (float32 / complex64), the same operation order, with sequential fp32 ``t += dt`` and
``loss += ...`` accumulation and plain ``log(1 + z)``.  What pins it instead: the reference's six
invariants (tests/test_oracle.py), a float64 twin (``dtype="f64"``) and central
finite-difference gradient checks in float64.

Every function cites the reference lines it follows (paths relative to /root/reference).

The backward pass has no reference source (it is produced by ``tf.train.AdamOptimizer.minimize``,
``train.py:89``); ``psi_loss_and_grads`` is the op-by-op reverse-mode adjoint of the forward below,
i.e. what a while-loop autodiff would compute, in the same dtype as the forward.
Cotangent convention for a complex value z: zbar = dL/dRe(z) + 1j * dL/dIm(z), so that
dL = Re(conj(zbar) * dz).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np


# --------------------------------------------------------------------------------------------
# hyper-parameters (train.py:41-43; tests/test_model.py:13-14)
# --------------------------------------------------------------------------------------------
@dataclass
class HParams:
    """Field names are the reference's (``tf.contrib.training.HParams`` at train.py:41-43)."""
    minibatch_size: int = 8
    bond_dim: int = 8
    delta_t: float = 1.0 / 16000
    sigma: float = 0.0001
    h_reg: float = 200.0 / (math.pi * 16000) ** 2
    r_reg: float = 0.1
    initial_rank: Optional[int] = None
    A: float = 100.0
    learning_rate: float = 0.001


def _dt(dtype):
    if dtype == "f32":
        return np.float32, np.complex64
    if dtype == "f64":
        return np.float64, np.complex128
    raise ValueError(dtype)


# --------------------------------------------------------------------------------------------
# raw (trainable) variables and their initialisation
# --------------------------------------------------------------------------------------------
@dataclass
class Variables:
    """The trainable set of PsiCMPS: A, Rx, Ry, freqs (model.py:19,32-39,46-49), psi_x, psi_y
    (model.py:218-219).  ``scaled_R`` / ``scaled_freqs`` record which initialisation branch was taken:
    on the random-init branch the effective value is rsqrt(reg) * variable (model.py:36-39,49)."""
    A: np.ndarray
    Rx: np.ndarray
    Ry: np.ndarray
    freqs: np.ndarray
    psi_x: np.ndarray
    psi_y: np.ndarray
    scaled_R: bool = True
    scaled_freqs: bool = True

    NAMES = ("A", "Rx", "Ry", "freqs", "psi_x", "psi_y")

    def copy(self):
        return Variables(*(np.array(getattr(self, n), copy=True) for n in self.NAMES),
                         scaled_R=self.scaled_R, scaled_freqs=self.scaled_freqs)

    def astype(self, real):
        return Variables(*(np.asarray(getattr(self, n), dtype=real) for n in self.NAMES),
                         scaled_R=self.scaled_R, scaled_freqs=self.scaled_freqs)


def init_variables(hp: HParams, seed: int = 0, R_in=None, freqs_in=None, psi_in=None) -> Variables:
    """Initial values by the reference's rules, from a numpy Generator (TF's own RNG streams are not
    reproducible outside TF).
    Rx, Ry, freqs: ``tf.random_normal_initializer`` (std 1)            model.py:36-39,49
    R_in / freqs_in: variables initialised to the given arrays          model.py:31-33,44-46
    psi_x, psi_y: ``initializer=None`` -> TF default glorot-uniform; for a 1-D [D] variable the limit
    is sqrt(6 / (D + D)) (legacy graph shows +-0.7746 at D=5)           model.py:218-219
    psi_in: the reference's branch is broken (undefined psi_x_in, model.py:214-216); treated here as
    complex D-vector -> real / imaginary parts.
    """
    D = hp.bond_dim
    rng = np.random.default_rng(seed)
    A = np.float32(hp.A)
    nRx = rng.standard_normal((D, D)).astype(np.float32)
    nRy = rng.standard_normal((D, D)).astype(np.float32)
    nf = rng.standard_normal(D).astype(np.float32)
    lim = math.sqrt(6.0 / (2 * D))
    px = rng.uniform(-lim, lim, D).astype(np.float32)
    py = rng.uniform(-lim, lim, D).astype(np.float32)
    scaled_R = R_in is None
    scaled_f = freqs_in is None
    if R_in is not None:
        R_in = np.asarray(R_in)
        nRx, nRy = R_in.real.astype(np.float32), R_in.imag.astype(np.float32)
    if freqs_in is not None:
        nf = np.asarray(freqs_in, dtype=np.float32)
    if psi_in is not None:
        psi_in = np.asarray(psi_in)
        px, py = psi_in.real.astype(np.float32), psi_in.imag.astype(np.float32)
    return Variables(np.asarray(A), nRx, nRy, nf, px, py, scaled_R=scaled_R, scaled_freqs=scaled_f)


# --------------------------------------------------------------------------------------------
# a1: CMPS.__init__ (model.py:9-52) -- effective parameters
# --------------------------------------------------------------------------------------------
def _rsqrt(x, real):
    """tf.rsqrt on a scalar constant: 1/sqrt(x) evaluated in the working dtype."""
    x = real(x)
    return real(1) / np.sqrt(x)


def effective_params(hp: HParams, var: Variables, dtype="f32"):
    """Returns (R [D,D] complex, freqs [D] real, c_r, c_h).   model.py:31-52
    R = complex(Rx, Ry); R -= matrix_diag_part(R) -- the subtraction broadcasts the diagonal as a ROW
    vector: R[i, j] = Z[i, j] - Z[j, j]  (model.py:41-42)."""
    real, cplx = _dt(dtype)
    c_r = _rsqrt(hp.r_reg, real) if var.scaled_R else real(1)
    c_h = _rsqrt(hp.h_reg, real) if var.scaled_freqs else real(1)
    Rx = c_r * var.Rx.astype(real) if var.scaled_R else var.Rx.astype(real)
    Ry = c_r * var.Ry.astype(real) if var.scaled_R else var.Ry.astype(real)
    Z = (Rx + 1j * Ry).astype(cplx)
    R = (Z - np.diagonal(Z)[None, :]).astype(cplx)
    freqs = c_h * var.freqs.astype(real) if var.scaled_freqs else var.freqs.astype(real)
    return R, freqs.astype(real), c_r, c_h


# --------------------------------------------------------------------------------------------
# a8: PsiCMPS._normalize_psi (model.py:327-334)
# --------------------------------------------------------------------------------------------
def normalize_psi(x, axis=None, epsilon=1e-12, dtype="f32"):
    real, cplx = _dt(dtype)
    a = np.abs(x).astype(real)                                   # tf.abs of complex -> hypot
    square_sum = np.sum(np.square(a), axis=axis, keepdims=True, dtype=real)
    x_inv_norm = real(1) / np.sqrt(np.maximum(square_sum, real(epsilon)))   # tf.rsqrt
    return (x * x_inv_norm.astype(cplx)).astype(cplx)


def psi_0(var: Variables, dtype="f32"):
    """a2: PsiCMPS.__init__ (model.py:221-222): normalise complex(psi_x, psi_y) over all of D."""
    real, cplx = _dt(dtype)
    p = (var.psi_x.astype(real) + 1j * var.psi_y.astype(real)).astype(cplx)
    return normalize_psi(p, axis=None, dtype=dtype).reshape(-1)


# --------------------------------------------------------------------------------------------
# a5 / a7 / a6: the step pieces (model.py:300-317, 319-325, 293-294)
# --------------------------------------------------------------------------------------------
def _phases(freqs, t, dtype):
    """phases = tf.exp(1j * freqsc * t_c) (model.py:304-305, 321-322): the imaginary argument is the
    working-precision product fl(freqs * t); exp of a purely imaginary complex number."""
    real, cplx = _dt(dtype)
    freqsc = freqs.astype(cplx)
    t_c = cplx(t)
    return np.exp(cplx(1j) * freqsc * t_c).astype(cplx)


def update_ancilla_psi(psi, signal, t, R, freqs, A, hp: HParams, dtype="f32"):
    """PsiCMPS._update_ancilla_psi (model.py:300-317).  psi [B,D] complex, signal [B] real, t scalar."""
    real, cplx = _dt(dtype)
    s = (signal.astype(real) / real(A)).astype(cplx)              # model.py:303
    phases = _phases(freqs, real(t), dtype)                       # :304-305
    Upsi = psi * np.conj(phases)                                  # :306
    Rdag = np.conj(R.T)                                           # :308
    RUpsi = (Upsi @ R.T).astype(cplx)                             # :309  einsum('bc,ac->ab')
    RdagRUpsi = (RUpsi @ Rdag.T).astype(cplx)                     # :310
    cc = cplx(-hp.delta_t * hp.sigma ** 2)                        # python float64 product, then cast
    delta_Upsi = cc * RdagRUpsi / cplx(2.0)                       # :312
    delta_Upsi = delta_Upsi + s[:, None] * RUpsi                  # :313
    delta_psi = phases * delta_Upsi                               # :315
    return (psi + delta_psi).astype(cplx)                         # :317


def expectation(psi, t, R, freqs, dtype="f32"):
    """PsiCMPS._expectation (model.py:319-325): 2 Re <U| R |U>, U = psi * conj(phases)."""
    real, cplx = _dt(dtype)
    phases = _phases(freqs, real(t), dtype)
    Upsi = psi * np.conj(phases)
    RU = (Upsi @ R.T).astype(cplx)
    ex = np.sum(np.conj(Upsi) * RU, axis=1)                       # einsum('ab,bc,ac->a')
    return (real(2) * ex.real).astype(real)


def inc_loss_psi(psi, signal, t, R, freqs, A, dtype="f32"):
    """PsiCMPS._inc_loss_psi (model.py:293-294): -log(1. + e * signal / A), plain log(1+z)."""
    real, _ = _dt(dtype)
    e = expectation(psi, t, R, freqs, dtype)
    z = (e * signal.astype(real)) / real(A)
    return (-np.log(real(1) + z)).astype(real)


# --------------------------------------------------------------------------------------------
# a3 / a4: PsiCMPS._build_loss_psi and the fold (model.py:257-267, 276-282)
# --------------------------------------------------------------------------------------------
def time_table(delta_t, N, dtype="f32"):
    """t_0 = 0, t_{k+1} = t_k + dt accumulated sequentially in the working dtype
    (model.py:16 ``self.dt = tf.constant(delta_t, tf.float32)``; :266 initial 0.; :281 ``t += self.dt``)."""
    real, _ = _dt(dtype)
    t = np.empty(N + 1, dtype=real)
    acc = real(0)
    dt = real(delta_t)
    for k in range(N + 1):
        t[k] = acc
        acc = real(acc + dt)
    return t


def psi_loss_per_clip(hp: HParams, var: Variables, data, dtype="f32", return_states=False):
    """Per-clip loss [B] (the fold carry of model.py:265-266); ``mean`` of it is PsiCMPS.loss (:267).
    With return_states=True also returns the normalised psi after each step, [B, N, D]
    (what PsiCMPS.psi_evolve_with_data / _psi_update, model.py:231-240, 269-274, produce)."""
    real, cplx = _dt(dtype)
    data = np.asarray(data, dtype=real)
    B, T = data.shape
    R, freqs, _, _ = effective_params(hp, var, dtype)
    A = real(var.A)
    incs = (data[:, 1:] - data[:, :-1]).astype(real)              # model.py:263
    psi = np.tile(psi_0(var, dtype)[None, :], (B, 1)).astype(cplx)  # :260
    loss = np.zeros(B, dtype=real)
    t = real(0)
    dt = real(hp.delta_t)
    states = [] if return_states else None
    for k in range(T - 1):                                        # tf.foldl, :265
        x = incs[:, k]
        psi = update_ancilla_psi(psi, x, t, R, freqs, A, hp, dtype)          # :278
        loss = (loss + inc_loss_psi(psi, x, t, R, freqs, A, dtype)).astype(real)  # :279
        psi = normalize_psi(psi, axis=1, dtype=dtype)                          # :280
        t = real(t + dt)                                                       # :281
        if return_states:
            states.append(psi)
    if return_states:
        return loss, np.stack(states, axis=1)
    return loss


def psi_loss(hp, var, data, dtype="f32"):
    """PsiCMPS.loss = reduce_mean over the batch (model.py:267)."""
    real, _ = _dt(dtype)
    per = psi_loss_per_clip(hp, var, data, dtype)
    return real(np.mean(per, dtype=real))


# --------------------------------------------------------------------------------------------
# a10: total loss of train.py:55-60
# --------------------------------------------------------------------------------------------
def regularisers(hp: HParams, var: Variables, dtype="f32"):
    real, _ = _dt(dtype)
    R, freqs, _, _ = effective_params(hp, var, dtype)
    h_l2sqnorm = np.sum(np.square(freqs), dtype=real)             # train.py:55
    r_l2sqnorm = np.sum((np.conj(R) * R).real, dtype=real)        # train.py:56
    return real(hp.h_reg) * h_l2sqnorm + real(hp.r_reg) * r_l2sqnorm


def total_loss(hp, var, data, dtype="f32"):
    real, _ = _dt(dtype)
    return real(psi_loss(hp, var, data, dtype) + regularisers(hp, var, dtype))


# --------------------------------------------------------------------------------------------
# a9: gradients (no reference source; reverse-mode adjoint of the forward above)
# --------------------------------------------------------------------------------------------
@dataclass
class Grads:
    """Gradients of mean_b loss_b (+ regularisers if requested) w.r.t. the raw variables, plus the
    intermediate gradients w.r.t. the effective parameters (what the HIP kernels emit)."""
    A: np.ndarray
    Rx: np.ndarray
    Ry: np.ndarray
    freqs: np.ndarray
    psi_x: np.ndarray
    psi_y: np.ndarray
    loss: float = 0.0
    per_clip: Optional[np.ndarray] = None
    eff: dict = field(default_factory=dict)   # Rbar [D,D] complex, fbar [D], psi0bar [D] complex, Abar

    def flat(self):
        return np.concatenate([np.ravel(getattr(self, n)) for n in Variables.NAMES])


def psi_loss_and_grads(hp: HParams, var: Variables, data, dtype="f32", with_reg=False) -> Grads:
    real, cplx = _dt(dtype)
    data = np.asarray(data, dtype=real)
    B, T = data.shape
    N = T - 1
    D = hp.bond_dim
    R, freqs, c_r, c_h = effective_params(hp, var, dtype)
    A = real(var.A)
    Rdag_T = np.conj(R)                       # adjoint(R) transposed
    cc = cplx(-hp.delta_t * hp.sigma ** 2)
    incs = (data[:, 1:] - data[:, :-1]).astype(real)
    p0 = psi_0(var, dtype)
    psi = np.tile(p0[None, :], (B, 1)).astype(cplx)
    loss = np.zeros(B, dtype=real)
    t = real(0)
    dt = real(hp.delta_t)
    tape_psi = np.empty((N, B, D), dtype=cplx)
    tape_t = np.empty(N, dtype=real)
    # ---------------- forward (identical op order to psi_loss_per_clip) ----------------
    for k in range(N):
        tape_psi[k] = psi
        tape_t[k] = t
        x = incs[:, k]
        psi = update_ancilla_psi(psi, x, t, R, freqs, A, hp, dtype)
        loss = (loss + inc_loss_psi(psi, x, t, R, freqs, A, dtype)).astype(real)
        psi = normalize_psi(psi, axis=1, dtype=dtype)
        t = real(t + dt)
    # ---------------- reverse ----------------
    g = np.zeros((B, D), dtype=cplx)          # cotangent of the carried (normalised) psi
    lbar = real(1) / real(B)                  # d mean / d loss_b
    Rbar = np.zeros((D, D), dtype=cplx)
    fbar = np.zeros(D, dtype=real)
    Abar = real(0)
    eps = real(1e-12)
    for k in range(N - 1, -1, -1):
        psi_k = tape_psi[k]
        tk = tape_t[k]
        x = incs[:, k]
        # recompute the step's intermediates (same ops as the forward)
        s_r = (x / A).astype(real)
        s = s_r.astype(cplx)
        ph = _phases(freqs, tk, dtype)
        U = psi_k * np.conj(ph)
        V = (U @ R.T).astype(cplx)
        W = (V @ Rdag_T).astype(cplx)
        delta = cc * W / cplx(2.0) + s[:, None] * V
        psi_p = (psi_k + ph * delta).astype(cplx)
        Up = psi_p * np.conj(ph)
        RVp = (Up @ R.T).astype(cplx)
        ex = np.sum(np.conj(Up) * RVp, axis=1)
        e = (real(2) * ex.real).astype(real)
        ex_ = (e * x).astype(real)
        z = (ex_ / A).astype(real)
        a = np.abs(psi_p).astype(real)
        ss = np.sum(np.square(a), axis=1, keepdims=True, dtype=real)
        m = np.maximum(ss, eps)
        inv = (real(1) / np.sqrt(m)).astype(real)
        # --- normalise:  psi_next = psi_p * inv
        psi_p_bar = g * inv.astype(cplx)
        inv_bar = np.sum((np.conj(g) * psi_p).real, axis=1, keepdims=True).astype(real)
        m_bar = inv_bar * (real(-0.5) * inv / m)
        ss_bar = np.where(ss > eps, m_bar, real(0)).astype(real)
        psi_p_bar = psi_p_bar + (real(2) * ss_bar).astype(cplx) * psi_p
        # --- loss increment: l = -log(1 + z), z = (e * x) / A
        z_bar = (-lbar / (real(1) + z)).astype(real)
        Abar = real(Abar + np.sum(z_bar * (-ex_ / (A * A)), dtype=real))
        e_bar = (z_bar * x / A).astype(real)
        ex_bar = (real(2) * e_bar).astype(cplx)
        Up_bar = ex_bar[:, None] * RVp
        RVp_bar = ex_bar[:, None] * Up
        Up_bar = Up_bar + RVp_bar @ np.conj(R)
        Rbar = Rbar + (np.conj(Up.T) @ RVp_bar).T
        psi_p_bar = psi_p_bar + Up_bar * ph
        ph_bar = np.conj(np.sum(Up_bar * np.conj(psi_p), axis=0))
        # --- psi_p = psi_k + ph * delta
        g_new = psi_p_bar.copy()
        ph_bar = ph_bar + np.sum(psi_p_bar * np.conj(delta), axis=0)
        delta_bar = psi_p_bar * np.conj(ph)
        # --- delta = cc * W / 2 + s * V
        s_bar = np.sum((delta_bar * np.conj(V)).real, axis=1).astype(real)
        V_bar = delta_bar * s[:, None]
        W_bar = delta_bar * (np.conj(cc) / cplx(2.0))
        # --- W = V @ conj(R)
        V_bar = V_bar + W_bar @ R.T
        Rbar = Rbar + np.conj(np.conj(V.T) @ W_bar)
        # --- V = U @ R.T
        U_bar = V_bar @ np.conj(R)
        Rbar = Rbar + (np.conj(U.T) @ V_bar).T
        # --- U = psi_k * conj(ph)
        g_new = g_new + U_bar * ph
        ph_bar = ph_bar + np.conj(np.sum(U_bar * np.conj(psi_k), axis=0))
        # --- s = x / A
        Abar = real(Abar + np.sum(s_bar * (-x / (A * A)), dtype=real))
        # --- ph = exp(1j * f * t)
        w_bar = ph_bar * np.conj(ph)
        fbar = (fbar + tk * w_bar.imag).astype(real)
        g = g_new.astype(cplx)
    # psi_0 tiled over the batch
    psi0_bar = np.sum(g, axis=0)
    eff = {"Rbar": Rbar.copy(), "fbar": fbar.copy(), "psi0bar": psi0_bar.copy(), "Abar": real(Abar)}
    if with_reg:                                                   # train.py:55-60
        fbar = fbar + real(2 * hp.h_reg) * freqs
        Rbar = Rbar + cplx(2 * hp.r_reg) * R
    # R[i,j] = Z[i,j] - Z[j,j]   (model.py:42)  ->  Zbar = Rbar - diag(colsum(Rbar))
    Zbar = Rbar - np.diag(np.sum(Rbar, axis=0))
    gRx = (c_r * Zbar.real).astype(real)
    gRy = (c_r * Zbar.imag).astype(real)
    gf = (c_h * fbar).astype(real)
    # psi_0 = p * rsqrt(max(sum |p|^2, eps))
    p = (var.psi_x.astype(real) + 1j * var.psi_y.astype(real)).astype(cplx)
    ss0 = np.sum(np.square(np.abs(p).astype(real)), dtype=real)
    m0 = max(ss0, eps)
    inv0 = real(1) / np.sqrt(m0)
    p_bar = psi0_bar * cplx(inv0)
    inv0_bar = real(np.sum((np.conj(psi0_bar) * p).real, dtype=real))
    m0_bar = inv0_bar * (real(-0.5) * inv0 / m0)
    if ss0 > eps:
        p_bar = p_bar + cplx(2 * m0_bar) * p
    total = real(np.mean(loss, dtype=real))
    if with_reg:
        total = real(total + regularisers(hp, var, dtype))
    return Grads(A=np.asarray(real(Abar)), Rx=gRx, Ry=gRy, freqs=gf,
                 psi_x=p_bar.real.astype(real), psi_y=p_bar.imag.astype(real),
                 loss=total, per_clip=loss, eff=eff)


# --------------------------------------------------------------------------------------------
# next row: PsiCMPS.sample  (model.py:242-251, 284-291)
# --------------------------------------------------------------------------------------------
def sample_noise(hp: HParams, num_samples, length, temp=1.0, seed=0):
    """noise = tf.random_normal([length, num_samples], stddev=sigma*sqrt(temp*delta_t)) (model.py:246); a seeded
    numpy Generator stands in for TF's stream.  Returned as [length, num_samples] float32 like the reference."""
    rng = np.random.default_rng(seed)
    std = hp.sigma * np.sqrt(temp * hp.delta_t)
    return (std * rng.standard_normal((length, num_samples))).astype(np.float32)


def psi_sample(hp: HParams, var: Variables, noise, dtype="f32", return_states=False):
    """PsiCMPS.sample given the pre-drawn noise [length, num_samples]: tf.scan of _psi_and_sample_update
    (model.py:284-291) from (psi_0 stacked, zeros, 0.), returning A * transpose(samples)  [num_samples, length]."""
    real, cplx = _dt(dtype)
    noise = np.asarray(noise, dtype=real)
    length, n = noise.shape
    R, freqs, _, _ = effective_params(hp, var, dtype)
    A = real(var.A)
    psi = np.tile(psi_0(var, dtype)[None, :], (n, 1)).astype(cplx)          # model.py:245
    sample = np.zeros(n, dtype=real)                                        # :244
    t = real(0)
    dt = real(hp.delta_t)
    out = np.empty((length, n), dtype=real)
    states = []
    for k in range(length):
        increment = (expectation(psi, t, R, freqs, dtype) * real(hp.delta_t) + noise[k]).astype(real)   # :286
        sample = (sample + increment).astype(real)                                                     # :287
        psi = update_ancilla_psi(psi, increment, t, R, freqs, A, hp, dtype)                            # :288
        psi = normalize_psi(psi, axis=1, dtype=dtype)                                                  # :289
        t = real(t + dt)                                                                               # :290
        out[k] = sample
        if return_states:
            states.append(psi)
    res = (A * out.T).astype(real)                                                                     # :251
    if return_states:
        return res, np.stack(states, axis=1)
    return res


# --------------------------------------------------------------------------------------------
# input fixture: the damped sine of data.py:8-22
# --------------------------------------------------------------------------------------------
def damped_sine(batch, input_length, delta_t, seed=0):
    """data.py:8-22: gamma(alpha=2, rate=2/delay_time)-delayed, exponentially damped 261.6 Hz sine,
    float32 [batch, input_length].  The reference draws the delays with tf.random_gamma; here a numpy
    Generator with an explicit seed (TF's stream is not reproducible outside TF)."""
    rng = np.random.default_rng(seed)
    freq = 261.6
    decay_time = 0.1
    delay_time = input_length / 100
    delays = rng.gamma(shape=2.0, scale=delay_time / 2.0, size=batch).astype(np.float32)
    input_range = np.arange(input_length, dtype=np.float32)[None, :]
    times = (input_range - delays[:, None]) * np.float32(delta_t)
    two_pi_f = np.float32(2 * np.pi * freq)
    wave = (np.float32(0.5) * (np.sign(times) + np.float32(1))
            * np.sin(two_pi_f * times) * np.exp(-times / np.float32(decay_time)))
    return wave.astype(np.float32)


# --------------------------------------------------------------------------------------------
# next row: the legacy AudioMPS arithmetic, reconstructed in SURVEY Appendix A from the
# serialized training graph logging/graph.pbtxt (the class body no longer exists in model.py).
#   variables H, R: real float32 [D,D], glorot-uniform                          graph.pbtxt:9032-9303, 9632-9903
#   H_s = band_part(H,-1,0) + band_part(H,-1,0)^T                               :9442-9605
#   psi_0 = one_hot(0, D) as complex64, per clip                                :10585-10742
#   step:  e = 2 Re(conj(psi) . R_c psi)   on the normalised pre-update psi      :11857-12661
#          loss += (x - e)^2 / 2                                                 :12685-12819
#          Q = dt (-i H_s - R^T R / 2);  psi' = psi + Q psi + dt x (R psi)       :12982-14323
#          psi = psi' rsqrt(max(sum |psi'|^2, 1e-12))                            :14350-14594
#   result mean_b(loss)                                                          :14811-14847
# --------------------------------------------------------------------------------------------
def legacy_init(D, seed=0):
    """H, R glorot-uniform (limit sqrt(6 / (2 D)))."""
    rng = np.random.default_rng(seed)
    lim = math.sqrt(6.0 / (2 * D))
    return (rng.uniform(-lim, lim, (D, D)).astype(np.float32), rng.uniform(-lim, lim, (D, D)).astype(np.float32))


def legacy_Q(H, R, dt, dtype="f32"):
    real, cplx = _dt(dtype)
    H = np.asarray(H, dtype=real)
    R = np.asarray(R, dtype=real)
    L = np.tril(H)
    Hs = (L + L.T).astype(real)
    RtR = (R.T @ R).astype(real)
    Q = (cplx(-1j) * Hs.astype(cplx) - (RtR / real(2)).astype(cplx)) * cplx(real(dt))
    return Q.astype(cplx), Hs


def legacy_loss_and_grads(H, R, dt, data, dtype="f32"):
    """Returns dict(loss, per_clip, gH, gR, Qbar, Rcbar): mean_b loss and its gradients (reverse-mode adjoint)."""
    real, cplx = _dt(dtype)
    data = np.asarray(data, dtype=real)
    B, T = data.shape
    N = T - 1
    H = np.asarray(H, dtype=real)
    R = np.asarray(R, dtype=real)
    D = R.shape[0]
    Q, _ = legacy_Q(H, R, dt, dtype)
    Rc = R.astype(cplx)
    incs = (data[:, 1:] - data[:, :-1]).astype(real)
    psi = np.zeros((B, D), dtype=cplx)
    psi[:, 0] = 1
    loss = np.zeros(B, dtype=real)
    tape = np.empty((N, B, D), dtype=cplx)
    dtr = real(dt)
    eps = real(1e-12)
    for k in range(N):
        tape[k] = psi
        x = incs[:, k]
        v = (psi @ Rc.T).astype(cplx)
        e = (real(2) * np.sum(np.conj(psi) * v, axis=1).real).astype(real)
        loss = (loss + np.square(x - e) / real(2)).astype(real)
        y = (psi + psi @ Q.T + (dtr * x)[:, None].astype(cplx) * v).astype(cplx)
        ss = np.sum(np.square(np.abs(y).astype(real)), axis=1, keepdims=True, dtype=real)
        psi = (y * (real(1) / np.sqrt(np.maximum(ss, eps))).astype(cplx)).astype(cplx)
    g = np.zeros((B, D), dtype=cplx)
    Qbar = np.zeros((D, D), dtype=cplx)
    Rcbar = np.zeros((D, D), dtype=cplx)
    lbar = real(1) / real(B)
    for k in range(N - 1, -1, -1):
        p = tape[k]
        x = incs[:, k]
        v = (p @ Rc.T).astype(cplx)
        e = (real(2) * np.sum(np.conj(p) * v, axis=1).real).astype(real)
        c = (dtr * x).astype(real)
        y = (p + p @ Q.T + c[:, None].astype(cplx) * v).astype(cplx)
        ss = np.sum(np.square(np.abs(y).astype(real)), axis=1, keepdims=True, dtype=real)
        m = np.maximum(ss, eps)
        inv = (real(1) / np.sqrt(m)).astype(real)
        ybar = g * inv.astype(cplx)
        inv_bar = np.sum((np.conj(g) * y).real, axis=1, keepdims=True).astype(real)
        ss_bar = np.where(ss > eps, inv_bar * (real(-0.5) * inv / m), real(0)).astype(real)
        ybar = ybar + (real(2) * ss_bar).astype(cplx) * y
        # y = p + p Q^T + c v
        pbar = ybar + ybar @ np.conj(Q)
        Qbar = Qbar + (np.conj(p.T) @ ybar).T
        vbar = ybar * c[:, None].astype(cplx)
        # loss term: (x - e)^2 / 2, e = 2 Re(conj(p) . v)
        e_bar = (lbar * (e - x)).astype(real)
        pbar = pbar + (real(2) * e_bar)[:, None].astype(cplx) * v          # d/dconj(p) side
        vbar = vbar + (real(2) * e_bar)[:, None].astype(cplx) * p
        # v = p Rc^T
        pbar = pbar + vbar @ np.conj(Rc)
        Rcbar = Rcbar + (np.conj(p.T) @ vbar).T
        g = pbar.astype(cplx)
    # Q = dt (-i Hs - R^T R / 2)
    Hs_bar = (-dtr * Qbar.imag).astype(real)
    M_bar = (-(dtr / real(2)) * Qbar.real).astype(real)
    gR = (Rcbar.real + R @ (M_bar + M_bar.T)).astype(real)
    gH = np.tril(Hs_bar + Hs_bar.T).astype(real)
    return {"loss": real(np.mean(loss, dtype=real)), "per_clip": loss, "gH": gH, "gR": gR, "Qbar": Qbar, "Rcbar": Rcbar}


# --------------------------------------------------------------------------------------------
# next row: RhoCMPS, the density-matrix scan  (model.py:55-203)
# --------------------------------------------------------------------------------------------
def rho_init_W(hp: HParams, seed=0):
    """Wx, Wy [rank, D]: initializer=None -> glorot-uniform, limit sqrt(6 / (rank + D))   (model.py:127-128)."""
    D = hp.bond_dim
    r = hp.initial_rank if hp.initial_rank is not None else D
    rng = np.random.default_rng(seed + 777)
    lim = math.sqrt(6.0 / (r + D))
    return (rng.uniform(-lim, lim, (r, D)).astype(np.float32), rng.uniform(-lim, lim, (r, D)).astype(np.float32))


def rho_0(Wx, Wy, dtype="f32"):
    """model.py:129-132: rho_0 = W^dagger W / trace(W^dagger W)."""
    real, cplx = _dt(dtype)
    W = (np.asarray(Wx, dtype=real) + 1j * np.asarray(Wy, dtype=real)).astype(cplx)
    r0 = (np.conj(W.T) @ W).astype(cplx)
    return (r0 / np.trace(r0)).astype(cplx)


def _rho_step(rho, x, t, R, freqs, A, hp, dtype):
    """One _rho_and_loss_update (model.py:152-158) in the reference's lab-frame matrix form; returns the pieces the
    adjoint needs."""
    real, cplx = _dt(dtype)
    s = (x.astype(real) / real(A)).astype(cplx)                                    # :175
    ph = _phases(freqs, real(t), dtype)                                            # :178
    Rt = (ph[:, None] * R * np.conj(ph)[None, :]).astype(cplx)                     # :179 einsum('a,ab,b->ab')
    RRd = (np.conj(Rt.T) @ Rt).astype(cplx)                                        # :180
    # :184  `- 0.5 * RR_dag * self.delta_t * self.sigma**2`: three successive complex64 products, left to right
    damp = (((cplx(-0.5) * RRd).astype(cplx) * cplx(hp.delta_t)).astype(cplx) * cplx(hp.sigma ** 2)).astype(cplx)
    c = cplx(cplx(-0.5) * cplx(hp.delta_t) * cplx(hp.sigma ** 2))                  # the same factor, for the adjoint
    U = (np.eye(R.shape[0], dtype=cplx)[None] + (damp[None] + s[:, None, None] * Rt[None])).astype(cplx)
    new_rho = np.matmul(np.matmul(U, rho).astype(cplx), np.conj(np.transpose(U, (0, 2, 1)))).astype(cplx)   # :186 U rho U^dagger
    X = (Rt + np.conj(Rt.T)).astype(cplx)                                          # :193-194
    e = np.einsum('ab,cba->c', X, new_rho).real.astype(real)                       # :195 trace(X rho)
    z = (e * x.astype(real)) / real(A)                                             # :166
    tr = np.trace(new_rho, axis1=1, axis2=2).real.astype(real)                     # :200
    m = np.maximum(tr, real(1e-12))
    return dict(s=s, ph=ph, Rt=Rt, RRd=RRd, c=c, U=U, new_rho=new_rho, X=X, e=e, z=z, tr=tr, m=m)


def rho_loss_per_clip(hp: HParams, var: Variables, Wx, Wy, data, dtype="f32", return_states=False):
    real, cplx = _dt(dtype)
    data = np.asarray(data, dtype=real)
    B, T = data.shape
    R, freqs, _, _ = effective_params(hp, var, dtype)
    A = real(var.A)
    rho = np.tile(rho_0(Wx, Wy, dtype)[None], (B, 1, 1))
    incs = (data[:, 1:] - data[:, :-1]).astype(real)
    loss = np.zeros(B, dtype=real)
    t = real(0)
    states = []
    for k in range(T - 1):
        st = _rho_step(rho, incs[:, k], t, R, freqs, A, hp, dtype)
        loss = (loss + (-np.log(real(1) + st["z"]))).astype(real)
        rho = (st["new_rho"] * (real(1) / st["m"]).astype(cplx)[:, None, None]).astype(cplx)    # :201-203
        t = real(t + real(hp.delta_t))
        if return_states:
            states.append(rho)
    return (loss, np.stack(states, axis=1)) if return_states else loss


def rho_loss_and_grads(hp: HParams, var: Variables, Wx, Wy, data, dtype="f32"):
    """mean_b loss_b of RhoCMPS and its gradients w.r.t. A, Rx, Ry, freqs, Wx, Wy (reverse-mode adjoint of the
    matrix recursion above; cotangent convention Mbar = dL/dRe M + i dL/dIm M, dL = Re tr(Mbar^dagger dM))."""
    real, cplx = _dt(dtype)
    data = np.asarray(data, dtype=real)
    B, T = data.shape
    N = T - 1
    D = hp.bond_dim
    R, freqs, c_r, c_h = effective_params(hp, var, dtype)
    A = real(var.A)
    W = (np.asarray(Wx, dtype=real) + 1j * np.asarray(Wy, dtype=real)).astype(cplx)
    r0 = (np.conj(W.T) @ W).astype(cplx)
    tr0 = np.trace(r0)
    rho = np.tile((r0 / tr0)[None], (B, 1, 1)).astype(cplx)
    incs = (data[:, 1:] - data[:, :-1]).astype(real)
    loss = np.zeros(B, dtype=real)
    t = real(0)
    tape = []
    for k in range(N):
        tape.append((rho, t))
        st = _rho_step(rho, incs[:, k], t, R, freqs, A, hp, dtype)
        loss = (loss + (-np.log(real(1) + st["z"]))).astype(real)
        rho = (st["new_rho"] * (real(1) / st["m"]).astype(cplx)[:, None, None]).astype(cplx)
        t = real(t + real(hp.delta_t))
    G = np.zeros((B, D, D), dtype=cplx)          # cotangent of the carried rho
    Rbar = np.zeros((D, D), dtype=cplx)
    fbar = np.zeros(D, dtype=real)
    Abar = real(0)
    lbar = real(1) / real(B)
    for k in range(N - 1, -1, -1):
        rho_k, tk = tape[k]
        x = incs[:, k]
        st = _rho_step(rho_k, x, tk, R, freqs, A, hp, dtype)
        U, nr, Rt, X, ph, s = st["U"], st["new_rho"], st["Rt"], st["X"], st["ph"], st["s"]
        # normalise: rho_next = nr / m
        inv = (real(1) / st["m"]).astype(real)
        nr_bar = G * inv.astype(cplx)[:, None, None]
        inv_bar = np.einsum('abc,abc->a', np.conj(G), nr).real.astype(real)
        tr_bar = np.where(st["tr"] > real(1e-12), inv_bar * (-inv * inv), real(0)).astype(real)
        nr_bar = nr_bar + tr_bar.astype(cplx)[:, None, None] * np.eye(D, dtype=cplx)[None]
        # loss: l = -log(1 + z), z = (e x)/A, e = Re tr(X nr)
        z_bar = (-lbar / (real(1) + st["z"])).astype(real)
        Abar = real(Abar + np.sum(z_bar * (-(st["e"] * x) / (A * A)), dtype=real))
        e_bar = (z_bar * x / A).astype(real)
        nr_bar = nr_bar + e_bar.astype(cplx)[:, None, None] * np.conj(X.T)[None]      # d tr(X nr)/d nr -> X^dagger
        X_bar = np.einsum('a,abc->bc', e_bar.astype(cplx), np.conj(np.transpose(nr, (0, 2, 1))))
        # nr = U rho U^dagger
        UR = np.matmul(U, rho_k)
        Ud = np.conj(np.transpose(U, (0, 2, 1)))
        U_bar = np.matmul(nr_bar + np.conj(np.transpose(nr_bar, (0, 2, 1))), UR)
        G = np.matmul(np.matmul(Ud, nr_bar), U).astype(cplx)                          # U^dagger nr_bar U
        # U = I + c RRd + s Rt
        RRd_bar = np.conj(st["c"]) * np.sum(U_bar, axis=0)
        Rt_bar = np.einsum('a,abc->bc', np.conj(s), U_bar)
        s_bar = np.einsum('abc,bc->a', U_bar, np.conj(Rt)).real.astype(real)
        Abar = real(Abar + np.sum(s_bar * (-x / (A * A)), dtype=real))
        # RRd = Rt^dagger Rt ;  X = Rt + Rt^dagger
        Rt_bar = Rt_bar + Rt @ (RRd_bar + np.conj(RRd_bar.T)) + X_bar + np.conj(X_bar.T)
        # Rt = ph R conj(ph)
        Rbar = Rbar + np.conj(ph)[:, None] * Rt_bar * ph[None, :]
        M = Rt_bar * np.conj(R)
        ph_bar = np.sum(M * ph[None, :], axis=1) + np.conj(np.sum(M * np.conj(ph)[:, None], axis=0))
        w_bar = ph_bar * np.conj(ph)
        fbar = (fbar + tk * w_bar.imag).astype(real)
    rho0_bar = np.sum(G, axis=0)
    # rho_0 = r0 / tr(r0), r0 = W^dagger W
    r0_bar = rho0_bar / np.conj(tr0) + np.eye(D, dtype=cplx) * (-(np.sum(np.conj(rho0_bar) * r0) / (tr0 * tr0))).conjugate()
    W_bar = W @ (r0_bar + np.conj(r0_bar.T))
    Zbar = Rbar - np.diag(np.sum(Rbar, axis=0))
    return {"loss": real(np.mean(loss, dtype=real)), "per_clip": loss,
            "A": real(Abar), "Rx": (c_r * Zbar.real).astype(real), "Ry": (c_r * Zbar.imag).astype(real),
            "freqs": (c_h * fbar).astype(real), "Wx": W_bar.real.astype(real), "Wy": W_bar.imag.astype(real),
            "eff": {"Rbar": Rbar, "fbar": fbar, "Abar": real(Abar), "rho0bar": rho0_bar}}


def rho_update_ancilla(hp: HParams, var: Variables, rho, signal, t, dtype="f32"):
    """RhoCMPS._update_ancilla_rho (model.py:172-187) for a batch rho [B, D, D]."""
    real, cplx = _dt(dtype)
    R, freqs, _, _ = effective_params(hp, var, dtype)
    st = _rho_step(np.asarray(rho, dtype=cplx), np.asarray(signal, dtype=real), real(t), R, freqs, real(var.A), hp, dtype)
    return st["new_rho"]


def rho_sample(hp: HParams, var: Variables, Wx, Wy, noise, dtype="f32"):
    """RhoCMPS.sample / rho_evolve_with_sampling / purity (model.py:86-116) for pre-drawn noise [length, n]:
    tf.scan of _rho_and_sample_update (:160-167).  Returns (waveforms [n, length], rho [n, length, D, D], purity [n, length])."""
    real, cplx = _dt(dtype)
    noise = np.asarray(noise, dtype=real)
    length, n = noise.shape
    R, freqs, _, _ = effective_params(hp, var, dtype)
    A = real(var.A)
    rho = np.tile(rho_0(Wx, Wy, dtype)[None], (n, 1, 1))
    sample = np.zeros(n, dtype=real)
    t = real(0)
    outs, rhos = [], []
    for k in range(length):
        ph = _phases(freqs, t, dtype)
        Rt = (ph[:, None] * R * np.conj(ph)[None, :]).astype(cplx)
        X = (Rt + np.conj(Rt.T)).astype(cplx)
        e = np.einsum('ab,cba->c', X, rho).real.astype(real)                      # :189-196 on the current rho
        inc = (e * real(hp.delta_t) + noise[k]).astype(real)                      # :162
        sample = (sample + inc).astype(real)                                      # :163
        st = _rho_step(rho, inc, t, R, freqs, A, hp, dtype)                       # :164
        rho = (st["new_rho"] * (real(1) / st["m"]).astype(cplx)[:, None, None]).astype(cplx)   # :165
        t = real(t + real(hp.delta_t))
        outs.append(sample)
        rhos.append(rho)
    rhos = np.stack(rhos, axis=1)
    purity = np.einsum('abcd,abdc->ab', rhos, rhos).real.astype(real)             # :101
    return (A * np.stack(outs, axis=1)).astype(real), rhos, purity


# --------------------------------------------------------------------------------------------
# BASELINE configs[4] (D = 128, "bf16 with fp32 accumulate"): emulation of the arithmetic the MFMA pair kernels execute
# (audio_mps_amd/csrc/cmps_pair.hip).  Same recurrence as above in the rotating frame, but every matrix-vector
# product takes bfloat16 operands (round-to-nearest-even) with float32 accumulation; everything else stays float32.
# The float32 restatement above remains the reference; this one pins down WHAT the reduced-precision kernels compute so
# that their parity test can be tight, and the distance between the two is the documented cost of bf16.
# --------------------------------------------------------------------------------------------
def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (v_cvt_pk_bf16_f32)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def _cbf(z):
    return (bf16_round(z.real) + 1j * bf16_round(z.imag)).astype(np.complex64)


def _cmatvec32(Mb, vb):
    """rows of vb [B, D] times Mb^T with float32 real arithmetic: (M v)_i = sum_j M_ij v_j."""
    Mr, Mi = Mb.real.astype(np.float32), Mb.imag.astype(np.float32)
    vr, vi = vb.real.astype(np.float32), vb.imag.astype(np.float32)
    return ((vr @ Mr.T - vi @ Mi.T) + 1j * (vi @ Mr.T + vr @ Mi.T)).astype(np.complex64)


def psi_bf16_scan(hp: HParams, var: Variables, data, want_grad=True):
    """Per-clip loss and effective-parameter gradient SUMS (layout of cmps_psi_loss_bwd / c_oracle.unpack_grad) with
    bfloat16 mat-vec operands.  Rotating frame: ut_k = rho_{k-1} y_{k-1} (un-normalised), y_k = inv_{k-1} (ut + Q ut + s R ut)."""
    f32, c64 = np.float32, np.complex64
    data = np.asarray(data, dtype=f32)
    B, T = data.shape
    N = T - 1
    D = hp.bond_dim
    R, freqs, _, _ = effective_params(hp, var, "f32")
    A = f32(var.A)
    p0 = psi_0(var, "f32")
    c_half = f32(f32(-hp.delta_t * hp.sigma ** 2) / f32(2))
    Q = (np.float64(c_half) * (np.conj(R.T).astype(np.complex128) @ R.astype(np.complex128))).astype(c64)
    H = (R + np.conj(R.T)).astype(c64)
    Rb, Qb, Hb = _cbf(R), _cbf(Q), _cbf(H)
    tt = time_table(hp.delta_t, N, "f32")
    th = (freqs[None, :].astype(f32) * tt[:, None].astype(f32)).astype(f32)              # fl32(f * t_k)
    dth = th[:-1].astype(np.float64) - th[1:].astype(np.float64)
    rho = (np.cos(dth) + 1j * np.sin(dth)).astype(c64)                                     # rho_k = phases_k conj(phases_{k+1})
    incs = (data[:, 1:] - data[:, :-1]).astype(f32)
    ut = np.tile(p0[None, :], (B, 1)).astype(c64)
    inv_prev = np.ones(B, dtype=f32)
    loss = np.zeros(B, dtype=f32)
    ys, hys, ns, es = [], [], [], []
    for k in range(N):
        x = incs[:, k]
        s = (x / A).astype(f32)
        ub = _cbf(ut)
        av, aq = _cmatvec32(Rb, ub), _cmatvec32(Qb, ub)
        y = (inv_prev[:, None] * (ut + (aq + s[:, None] * av))).astype(c64)
        hy = _cmatvec32(Hb, _cbf(y))
        e = np.sum(y.real * hy.real + y.imag * hy.imag, axis=1, dtype=f32)
        n = np.sum(y.real ** 2 + y.imag ** 2, axis=1, dtype=f32)
        loss = (loss + (-np.log(f32(1) + (e * x) / A))).astype(f32)
        ys.append(y); hys.append(hy); ns.append(n); es.append(e)
        inv_prev = (f32(1) / np.sqrt(np.maximum(n, f32(1e-12)))).astype(f32)
        ut = (rho[k][None, :] * y).astype(c64)
    out = {"loss_per_clip": loss}
    if not want_grad:
        return out
    g = np.zeros((B, D), dtype=c64)
    Rbar = np.zeros((D, D), dtype=np.complex128)
    Qbar = np.zeros((D, D), dtype=np.complex128)
    fbar = np.zeros(D, dtype=np.float64)
    Abar = 0.0
    dtk = (tt[:-1].astype(f32) - tt[1:].astype(f32)).astype(f32)
    for k in range(N - 1, -1, -1):
        x = incs[:, k]
        s = (x / A).astype(f32)
        y, hy, n, e = ys[k], hys[k], ns[k], es[k]
        inv = (f32(1) / np.sqrt(np.maximum(n, f32(1e-12)))).astype(f32)
        yhat = (inv[:, None] * y).astype(c64)
        un = (rho[k][None, :] * yhat).astype(c64)
        fbar += np.sum(dtk[k] * (g.imag * un.real - g.real * un.imag), axis=0, dtype=np.float64)
        yhb = (np.conj(rho[k])[None, :] * g).astype(c64)
        dot = np.sum(yhat.real * yhb.real + yhat.imag * yhb.imag, axis=1, dtype=f32)
        ok = n > f32(1e-12)
        ybar = np.where(ok[:, None], (yhb - yhat * dot[:, None]) * inv[:, None], yhb * inv[:, None]).astype(c64)
        ex = (e * x).astype(f32)
        z = (ex / A).astype(f32)
        zbar = (f32(-1) / (f32(1) + z)).astype(f32)
        te = (f32(2) * (zbar * x / A)).astype(f32)
        Abar += float(np.sum(zbar * (-ex / (A * A)), dtype=np.float64))
        ybar = (ybar + te[:, None] * hy).astype(c64)
        ybb = _cbf(ybar)
        bq = _cmatvec32(Qb, ybb)                                                           # Q Hermitian
        d = _cmatvec32(np.conj(Rb.T), ybb)                                                 # R^dagger ybar
        if k > 0:
            invp = (f32(1) / np.sqrt(np.maximum(ns[k - 1], f32(1e-12)))).astype(f32)
            uk = (rho[k - 1][None, :] * (invp[:, None] * ys[k - 1])).astype(c64)
        else:
            uk = np.tile(p0[None, :], (B, 1)).astype(c64)
        sbar = np.sum(d.real * uk.real + d.imag * uk.imag, axis=1, dtype=f32)
        Abar += float(np.sum(sbar * (-x / (A * A)), dtype=np.float64))
        g = (ybar + bq + s[:, None] * d).astype(c64)
        a1, b1 = _cbf((te[:, None] * y).astype(c64)), _cbf(y)
        a2, a3, b2 = ybb, _cbf((s[:, None] * ybar).astype(c64)), _cbf(uk)
        Rbar += np.einsum('bi,bj->ij', a1.astype(np.complex128), np.conj(b1).astype(np.complex128))
        Rbar += np.einsum('bi,bj->ij', a3.astype(np.complex128), np.conj(b2).astype(np.complex128))
        Qbar += np.einsum('bi,bj->ij', a2.astype(np.complex128), np.conj(b2).astype(np.complex128))
    Rbar = Rbar + np.float64(c_half) * (R.astype(np.complex128) @ (Qbar + np.conj(Qbar.T)))
    out.update({"Rbar": Rbar, "fbar": fbar, "psi0bar": np.sum(g, axis=0).astype(np.complex128), "Abar": Abar,
                "loss_sum": float(np.sum(loss, dtype=np.float64))})
    return out
