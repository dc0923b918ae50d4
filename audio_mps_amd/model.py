"""Host-side mirror of the reference's model surface for the pure-state path.

Same class names, constructor arguments, hyper-parameter field names and attributes as
/root/reference/model.py (CMPS :5-52, PsiCMPS :206-334) and the legacy ``AudioMPS`` surface that
training_estimators.py:43-45 / follow_vae.py:45-51 call.  The reference builds a TensorFlow graph whose
``.loss`` tensor is evaluated in a session and differentiated by ``AdamOptimizer.minimize``
(train.py:89); here ``.loss`` launches the HIP scan (audio_mps_amd/csrc) and ``loss_and_grads()`` runs
the forward + reverse kernels and applies the tiny host-side chain rule from the kernels' outputs
(gradients w.r.t. the effective R, freqs, psi_0, A) to the raw variables (Rx, Ry, freqs, psi_x, psi_y, A).

RhoCMPS (model.py:55-203, ``mixed=True``) is mirrored the same way on top of the cmps_rho_* entry points.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, fields
from typing import Callable, Dict, Optional

import numpy as np

VARIABLE_NAMES = ("A", "Rx", "Ry", "freqs", "psi_x", "psi_y")


# --------------------------------------------------------------------------------------------------
# hyper-parameters: field names of tf.contrib.training.HParams at train.py:41-43
# --------------------------------------------------------------------------------------------------
@dataclass
class HParams:
    minibatch_size: int = 8
    bond_dim: int = 8
    delta_t: float = 1.0 / 16000
    sigma: float = 0.0001
    h_reg: float = 200.0 / (math.pi * 16000) ** 2
    r_reg: float = 0.1
    initial_rank: Optional[int] = None
    A: float = 100.0
    learning_rate: float = 0.001

    def parse(self, spec: str) -> "HParams":
        """``--hparams=name=value,...`` override (train.py:31,44)."""
        if not spec:
            return self
        known = {f.name: f for f in fields(self)}
        for item in spec.split(","):
            if not item.strip():
                continue
            name, _, value = item.partition("=")
            name = name.strip()
            if name not in known:
                raise ValueError(f"Unknown hyperparameter: {name}")
            cur = getattr(self, name)
            if name == "initial_rank":
                setattr(self, name, None if value.strip() in ("None", "") else int(value))
            elif isinstance(cur, int) and not isinstance(cur, bool):
                setattr(self, name, int(value))
            else:
                setattr(self, name, float(value))
        return self

    def values(self) -> dict:
        return {f.name: getattr(self, f.name) for f in fields(self)}


def _f32(x):
    return np.float32(x)


def _rsqrt32(x) -> np.float32:
    """tf.rsqrt(python_float): evaluated in float32 (model.py:36,38,49)."""
    return np.float32(1) / np.sqrt(np.float32(x))


def _normalize_psi_host(p: np.ndarray) -> np.ndarray:
    """_normalize_psi with axis=None (model.py:327-334) for the D-vector psi_0 (tiny; host)."""
    a = np.abs(p).astype(np.float32)
    ss = np.sum(np.square(a), dtype=np.float32)
    inv = np.float32(1) / np.sqrt(np.maximum(ss, np.float32(1e-12)))
    return (p * np.complex64(inv)).astype(np.complex64)


# --------------------------------------------------------------------------------------------------
class _VarDict(dict):
    """model.variables: a plain dict whose item assignment invalidates a live device-resident copy of the variables (see CMPS.variables)."""

    def __init__(self, model):
        super().__init__()
        self._model = model

    def __setitem__(self, key, value):
        owner = self._model._owner() if hasattr(self._model, "_device_owner") else None
        if owner is not None and getattr(owner, "_dev", None) is not None and not getattr(owner, "_syncing_back", False):
            owner.drop_device_state()              # device -> host for everything else, then forget the device copy
        super().__setitem__(key, value)


class CMPS:
    """Continuous Matrix Product State: the trainable variables and the effective parameters
    (model.py:5-52).  ``self.R`` (complex64 [D,D], diagonal removed as at model.py:42), ``self.freqs``,
    ``self.A``, ``self.sigma`` are the attributes train.py:55-75 reads."""

    def __init__(self, hparams, data_iterator=None, freqs_in=None, R_in=None, seed: int = 0):
        self.hparams = hparams
        self.bond_d = int(hparams.bond_dim)
        self.batch_size = hparams.minibatch_size
        self.h_reg = hparams.h_reg
        self.r_reg = hparams.r_reg
        self.delta_t = hparams.delta_t
        self.dt = np.float32(hparams.delta_t)                     # model.py:16
        self.sigma = hparams.sigma                                # model.py:21 (not trainable)
        self.data_iterator = data_iterator
        D = self.bond_d
        rng = np.random.default_rng(seed)
        self._device_owner = None
        self._variables: Dict[str, np.ndarray] = _VarDict(self)
        self._device_owner = None      # a Trainer whose device-resident optimiser state is newer than self._variables (see `variables`)
        self.variables["A"] = np.asarray(np.float32(hparams.A))   # model.py:19
        # --- R (model.py:31-42)
        if R_in is not None:
            R_in = np.asarray(R_in)
            if R_in.shape != (D, D):
                raise ValueError(f"R_in must be [{D},{D}]")
            self.variables["Rx"] = R_in.real.astype(np.float32)
            self.variables["Ry"] = R_in.imag.astype(np.float32)
            self._c_r = np.float32(1)
        else:
            self.variables["Rx"] = rng.standard_normal((D, D)).astype(np.float32)   # random_normal_initializer
            self.variables["Ry"] = rng.standard_normal((D, D)).astype(np.float32)
            self._c_r = _rsqrt32(self.r_reg)
        # --- freqs (model.py:44-49)
        if freqs_in is not None:
            freqs_in = np.asarray(freqs_in, dtype=np.float32)
            if freqs_in.shape != (D,):
                raise ValueError(f"freqs_in must be [{D}]")
            self.variables["freqs"] = freqs_in.copy()
            self._c_h = np.float32(1)
        else:
            self.variables["freqs"] = rng.standard_normal(D).astype(np.float32)
            self._c_h = _rsqrt32(self.h_reg)
        self._rng = rng

    def _owner(self):
        ref = self._device_owner
        return ref() if ref is not None else None

    @property
    def variables(self) -> Dict[str, np.ndarray]:
        """The raw trainable variables (host copies).  While a Trainer runs the device-resident optimiser step the current values
        live on the GPU; reading this attribute then brings them back first (Trainer._lazy_sync: one device -> host copy, only after
        steps that changed them), so loss / sample / effective_params / summaries never see stale parameters.
        ASSIGNING an entry (``m.variables["Rx"] = ...``, ``m.variables["Rx"] *= c``) while a device-resident state is live brings that
        state back and drops it, so the Trainer's next step starts from the host values (such writes used to be ignored);
        the Trainer is held by a weak reference."""
        owner = self._owner()
        if owner is not None:
            owner._lazy_sync()
        return self._variables

    # effective parameters, recomputed from the current variables ---------------------------------
    @property
    def A(self) -> np.float32:
        return np.float32(self.variables["A"])

    @property
    def R(self) -> np.ndarray:
        Rx = (self._c_r * self.variables["Rx"]).astype(np.float32) if self._c_r != 1 else self.variables["Rx"]
        Ry = (self._c_r * self.variables["Ry"]).astype(np.float32) if self._c_r != 1 else self.variables["Ry"]
        Z = (Rx + 1j * Ry).astype(np.complex64)                   # model.py:41
        return (Z - np.diagonal(Z)[None, :]).astype(np.complex64)  # model.py:42 (row-vector broadcast)

    @property
    def freqs(self) -> np.ndarray:
        f = self.variables["freqs"]
        return (self._c_h * f).astype(np.float32) if self._c_h != 1 else f.astype(np.float32)

    @property
    def freqsc(self) -> np.ndarray:
        return self.freqs.astype(np.complex64)                    # model.py:52

    # ---- backend plumbing (shared by PsiCMPS and RhoCMPS) ----
    _backend = None

    def _get_backend(self):
        if self._backend is None:
            from .scan import HipScan   # raises if libcmps.so or the GPU is missing: no fallback
            self._backend = HipScan(self.bond_d)
        return self._backend

    def _batch(self, data=None):
        data = self.data_iterator if data is None else data
        if callable(data):
            data = data()
        return data

    def _to_device(self, data):
        import torch
        be = self._get_backend()
        dev = getattr(be, "device", None)
        if isinstance(data, torch.Tensor):
            t = data.to(dtype=torch.float32)
            if dev is not None:
                t = t.to(dev)
            return t.contiguous()
        t = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))
        return t.to(dev) if dev is not None else t

    def _chain_common(self, Rbar, fbar, Abar, loss, with_reg: bool):
        """Adjoint of model.py:36-42 (scaling + the row-broadcast diagonal removal) and :49 for batch-mean cotangents of
        the effective R, freqs, A; optionally the regularisers of train.py:55-60."""
        if with_reg:
            R = self.R.astype(np.complex128)
            f = self.freqs.astype(np.float64)
            loss = loss + self.h_reg * np.sum(f * f) + self.r_reg * np.sum((np.conj(R) * R).real)
            fbar = fbar + 2.0 * self.h_reg * f
            Rbar = Rbar + 2.0 * self.r_reg * R
        Zbar = Rbar - np.diag(np.sum(Rbar, axis=0))               # adjoint of model.py:42
        c_r, c_h = float(self._c_r), float(self._c_h)
        return loss, {
            "A": np.asarray(np.float32(Abar)),
            "Rx": (c_r * Zbar.real).astype(np.float32),
            "Ry": (c_r * Zbar.imag).astype(np.float32),
            "freqs": (c_h * fbar).astype(np.float32),
        }


# --------------------------------------------------------------------------------------------------
class PsiCMPS(CMPS):
    """Evolves the state (model.py:206-334).

    ``data_iterator`` is the batch the loss is evaluated on: a float32 [B, T] numpy array or CUDA tensor,
    or a zero-argument callable returning one (the eager stand-in for the TF input tensor that yields a
    new batch per session.run).  ``backend`` is the scan implementation; the product has exactly one
    (``HipScan``, the HIP kernels) and constructs it on first use -- tests may inject another object with
    the same two methods to exercise the host logic without a GPU.
    """

    def __init__(self, hparams, psi_in=None, *args, backend=None, **kwargs):
        super().__init__(hparams, *args, **kwargs)
        D = self.bond_d
        if psi_in is not None:
            # the reference's branch is broken (undefined psi_x_in, model.py:214-216); a complex D-vector
            psi_in = np.asarray(psi_in)
            if psi_in.shape != (D,):
                raise ValueError(f"psi_in must be [{D}]")
            self.variables["psi_x"] = psi_in.real.astype(np.float32)
            self.variables["psi_y"] = psi_in.imag.astype(np.float32)
        else:
            # initializer=None -> TF default glorot_uniform; 1-D [D]: limit sqrt(6 / (D + D))  (model.py:218-219)
            lim = math.sqrt(6.0 / (2 * D))
            self.variables["psi_x"] = self._rng.uniform(-lim, lim, D).astype(np.float32)
            self.variables["psi_y"] = self._rng.uniform(-lim, lim, D).astype(np.float32)
        self._backend = backend
        self._last = None

    # ---- attributes of the reference object ----
    @property
    def psi_0(self) -> np.ndarray:
        p = (self.variables["psi_x"] + 1j * self.variables["psi_y"]).astype(np.complex64)   # model.py:221
        return _normalize_psi_host(p)                                                        # model.py:222

    @property
    def loss(self) -> np.float32:
        """PsiCMPS.loss (model.py:224-225, 267): mean over the batch of the per-clip loss."""
        if self.data_iterator is None:
            raise AttributeError("loss: the model was built without a data_iterator (model.py:224)")
        per_clip = self.loss_per_clip()
        return np.float32(np.mean(per_clip, dtype=np.float32))

    # ---- backend plumbing ----
    def effective_params(self):
        from .scan import EffectiveParams
        return EffectiveParams(R=self.R, freqs=self.freqs, psi0=self.psi_0, A=float(self.A),
                               sigma=float(self.sigma), delta_t=float(self.delta_t))

    # ---- the hot path ----
    def loss_per_clip(self, data=None) -> np.ndarray:
        """The fold carry ``loss`` [B] of model.py:265-266 (forward only)."""
        audio = self._to_device(self._batch(data))
        B, T = audio.shape
        be = self._get_backend()
        be.set_params(self.effective_params(), B, T, train=False)
        return be.forward(audio, save_for_bwd=False).detach().cpu().numpy()

    def flat_size(self) -> int:
        """Length of the buffer grad_sums() returns (cmps_psi_loss_bwd's layout): what an empty shard adds to the all-reduce as zeros."""
        from .scan import grad_size
        return grad_size(self.bond_d)

    def grad_sums(self, data=None):
        """Forward + reverse scan on this process's clips.  Returns (flat device/host buffer of SUMS over
        clips as laid out by cmps_psi_loss_bwd, number of clips).  Used by loss_and_grads and by the
        data-parallel trainer, which all-reduces the buffer before the chain rule."""
        audio = self._to_device(self._batch(data))
        B, T = audio.shape
        be = self._get_backend()
        be.set_params(self.effective_params(), B, T, train=True)
        _, grad = be.loss_and_grad_sums(audio, check=True)       # fp16-range check + documented fallback (include/cmps.h)
        return grad, B

    def chain_rule(self, flat_sums: np.ndarray, global_batch: int, with_reg: bool = False):
        """Effective-parameter gradient sums -> (mean loss, gradients w.r.t. the raw variables).
        Adjoint of model.py:36-42 (scaling + the row-broadcast diagonal removal), :49, :221-222, and
        optionally the regularisers of train.py:55-60."""
        from .scan import unpack_grad
        D = self.bond_d
        g = unpack_grad(np.asarray(flat_sums, dtype=np.float64), D)
        invB = 1.0 / float(global_batch)
        Rbar = g["Rbar"] * invB
        fbar = g["fbar"] * invB
        p0bar = g["psi0bar"] * invB
        Abar = g["Abar"] * invB
        loss = g["loss_sum"] * invB
        loss, grads = self._chain_common(Rbar, fbar, Abar, loss, with_reg)
        p = (self.variables["psi_x"].astype(np.float64) + 1j * self.variables["psi_y"].astype(np.float64))
        ss = float(np.sum(np.abs(p) ** 2))
        m = max(ss, 1e-12)
        inv = 1.0 / math.sqrt(m)
        pbar = p0bar * inv
        if ss > 1e-12:
            inv_bar = float(np.sum((np.conj(p0bar) * p).real))
            pbar = pbar + 2.0 * (inv_bar * (-0.5 * inv / m)) * p
        grads["psi_x"] = pbar.real.astype(np.float32)
        grads["psi_y"] = pbar.imag.astype(np.float32)
        return np.float32(loss), grads

    def loss_and_grads(self, data=None, with_reg: bool = False):
        """(loss, {variable: gradient}) of mean_b loss_b (+ train.py's regularisers if with_reg): what
        ``AdamOptimizer.minimize`` obtains from TF (train.py:89)."""
        flat, B = self.grad_sums(data)
        host = flat.detach().cpu().numpy() if hasattr(flat, "detach") else np.asarray(flat)
        self._last = host
        return self.chain_rule(host, B, with_reg=with_reg)

    # ---- other reference methods on this class ----
    def _update_ancilla_psi(self, psi, signal, t):
        """model.py:300-317 for a batch of states (used by the reference's testTrivialUpdateOfAncilla)."""
        be = self._get_backend()
        psi = np.asarray(psi, dtype=np.complex64)
        be.set_params(self.effective_params(), psi.shape[0], 2, train=False)
        return be.update_ancilla(psi, np.asarray(signal, dtype=np.float32), float(t))

    def psi_evolve_with_data(self, data=None) -> np.ndarray:
        """model.py:231-240: the normalised state after every step, [B, T-1, D]."""
        audio = self._to_device(self._batch(data))
        B, T = audio.shape
        be = self._get_backend()
        be.set_params(self.effective_params(), B, T, train=True)
        be.forward(audio, save_for_bwd=True)
        return be.states()

    def sample(self, num_samples, length, temp=1, seed=None, noise=None):
        """model.py:242-251: waveforms [num_samples, length] = A * running sum of the sampled increments.
        The Gaussian noise (stddev sigma * sqrt(temp * delta_t), model.py:246) is drawn on the host with a numpy
        Generator (``seed``), or passed in as ``noise`` [length, num_samples] like the reference's tensor."""
        if noise is None:
            rng = np.random.default_rng(seed)
            std = float(self.sigma) * math.sqrt(temp * float(self.delta_t))
            noise = (std * rng.standard_normal((length, num_samples))).astype(np.float32)
        noise = np.asarray(noise, dtype=np.float32)
        if noise.shape != (length, num_samples):
            raise ValueError(f"noise must be [{length}, {num_samples}]")
        be = self._get_backend()
        be.set_params(self.effective_params(), num_samples, length + 1, train=False)
        return be.sample(noise)


# --------------------------------------------------------------------------------------------------
class RhoCMPS(CMPS):
    """Evolves the density matrix (model.py:55-203).

    Variables Wx, Wy [rank, D] (rank = hparams.initial_rank or bond_dim, :62-65), rho_0 = W^dagger W / trace (:127-132).
    The HIP scan carries rho as its ``rank`` columns phi_a = conj(W[a, :]) / sqrt(trace) (cmps_rho_* in include/cmps.h);
    everything the reference exposes on this class is mirrored: ``loss``, ``rho_0``, ``rho_evolve_with_data``,
    ``rho_evolve_with_sampling``, ``purity``, ``sample``, ``_update_ancilla_rho``."""

    VARIABLE_NAMES = ("A", "Rx", "Ry", "freqs", "Wx", "Wy")

    def __init__(self, hparams, W_in=None, *args, backend=None, **kwargs):
        super().__init__(hparams, *args, **kwargs)
        D = self.bond_d
        self.rank_rho_0 = int(hparams.initial_rank) if hparams.initial_rank is not None else D     # :62-65
        if W_in is not None:
            W_in = np.asarray(W_in)
            if W_in.ndim != 2 or W_in.shape[1] != D:
                raise ValueError(f"W_in must be [rank, {D}]")
            self.rank_rho_0 = W_in.shape[0]
            self.variables["Wx"] = W_in.real.astype(np.float32)                                    # :121-124
            self.variables["Wy"] = W_in.imag.astype(np.float32)
        else:
            # initializer=None -> glorot_uniform on [rank, D]: limit sqrt(6 / (rank + D))           (:126-127)
            lim = math.sqrt(6.0 / (self.rank_rho_0 + D))
            self.variables["Wx"] = self._rng.uniform(-lim, lim, (self.rank_rho_0, D)).astype(np.float32)
            self.variables["Wy"] = self._rng.uniform(-lim, lim, (self.rank_rho_0, D)).astype(np.float32)
        self._backend = backend
        self._last = None

    # ---- attributes of the reference object ----
    @property
    def W(self) -> np.ndarray:
        return (self.variables["Wx"] + 1j * self.variables["Wy"]).astype(np.complex64)            # :128

    @property
    def rho_0(self) -> np.ndarray:
        W = self.W
        r0 = (np.conj(W.T) @ W).astype(np.complex64)                                               # :129
        return (r0 / np.trace(r0)).astype(np.complex64)                                            # :130

    def columns(self) -> np.ndarray:
        """phi [rank, D] with rho_0 = sum_a phi_a phi_a^dagger: phi_a = conj(W[a, :]) / sqrt(tr W^dagger W)."""
        W = self.W.astype(np.complex128)
        t0 = float(np.sum(np.abs(W) ** 2))
        return (np.conj(W) / math.sqrt(t0)).astype(np.complex64)

    @property
    def loss(self) -> np.float32:
        """RhoCMPS.loss (model.py:69-70, 144): mean over the batch of the per-clip loss."""
        if self.data_iterator is None:
            raise AttributeError("loss: the model was built without a data_iterator (model.py:69)")
        return np.float32(np.mean(self.loss_per_clip(), dtype=np.float32))

    def effective_params(self):
        from .scan import EffectiveParams
        e0 = np.zeros(self.bond_d, dtype=np.complex64)
        e0[0] = 1                                                  # the pure-state psi_0 slot is unused on this path
        return EffectiveParams(R=self.R, freqs=self.freqs, psi0=e0, A=float(self.A),
                               sigma=float(self.sigma), delta_t=float(self.delta_t))

    def _prepare(self, B, T, train):
        be = self._get_backend()
        be.set_params(self.effective_params(), B, T, train=False)
        be.rho_set_state(self.columns(), B, T, train=train)
        return be

    # ---- the scan ----
    def loss_per_clip(self, data=None) -> np.ndarray:
        audio = self._to_device(self._batch(data))
        B, T = audio.shape
        be = self._prepare(B, T, train=False)
        return be.rho_forward(audio, save_for_bwd=False).detach().cpu().numpy()

    def flat_size(self) -> int:
        """Length of the buffer grad_sums() returns: the pure-state layout followed by the 2 rank D column cotangents
        (include/cmps.h: cmps_rho_loss_bwd)."""
        from .scan import grad_size
        return grad_size(self.bond_d) + 2 * self.rank_rho_0 * self.bond_d

    def grad_sums(self, data=None):
        """(flat buffer of SUMS over this process's clips as laid out by cmps_rho_loss_bwd, number of clips)."""
        audio = self._to_device(self._batch(data))
        B, T = audio.shape
        be = self._prepare(B, T, train=True)
        _, grad = be.rho_loss_and_grad_sums(audio)
        return grad, B

    def chain_rule(self, flat_sums: np.ndarray, global_batch: int, with_reg: bool = False):
        """Effective-parameter gradient sums -> (mean loss, gradients w.r.t. A, Rx, Ry, freqs, Wx, Wy)."""
        from .scan import unpack_grad, grad_size
        D, r = self.bond_d, self.rank_rho_0
        flat = np.asarray(flat_sums, dtype=np.float64)
        g = unpack_grad(flat, D)
        invB = 1.0 / float(global_batch)
        loss, grads = self._chain_common(g["Rbar"] * invB, g["fbar"] * invB, g["Abar"] * invB, g["loss_sum"] * invB,
                                         with_reg)
        tail = flat[grad_size(D):grad_size(D) + 2 * r * D]
        phibar = (tail[:r * D] + 1j * tail[r * D:]).reshape(r, D) * invB
        # phi = p / |p|, p = conj(W) (all rank * D entries as one vector), then W = conj(p)        adjoint of :128-130
        W = self.W.astype(np.complex128)
        nrm = math.sqrt(float(np.sum(np.abs(W) ** 2)))
        phi = np.conj(W) / nrm
        pbar = (phibar - phi * float(np.sum((np.conj(phi) * phibar).real))) / nrm
        grads["Wx"] = pbar.real.astype(np.float32)
        grads["Wy"] = (-pbar.imag).astype(np.float32)
        return np.float32(loss), grads

    def loss_and_grads(self, data=None, with_reg: bool = False):
        flat, B = self.grad_sums(data)
        host = flat.detach().cpu().numpy() if hasattr(flat, "detach") else np.asarray(flat)
        self._last = host
        return self.chain_rule(host, B, with_reg=with_reg)

    # ---- other reference methods on this class ----
    def _update_ancilla_rho(self, rho, signal, t):
        """model.py:172-187 for a batch of density matrices [B, D, D]."""
        be = self._get_backend()
        rho = np.asarray(rho, dtype=np.complex64)
        be.set_params(self.effective_params(), rho.shape[0], 2, train=False)
        return be.rho_update_ancilla(rho, np.asarray(signal, dtype=np.float32), float(t))

    def rho_evolve_with_data(self, data=None) -> np.ndarray:
        """model.py:76-84: the normalised rho after every step, [B, T-1, D, D]."""
        audio = self._to_device(self._batch(data))
        B, T = audio.shape
        be = self._prepare(B, T, train=True)
        be.rho_forward(audio, save_for_bwd=True)
        return be.rho_states(B, T - 1, want_rho=True)

    def _noise(self, num_samples, length, temp, seed, noise):
        if noise is None:
            rng = np.random.default_rng(seed)
            std = float(self.sigma) * math.sqrt(temp * float(self.delta_t))                        # :88, 96, 106
            noise = (std * rng.standard_normal((length, num_samples))).astype(np.float32)
        noise = np.asarray(noise, dtype=np.float32)
        if noise.shape != (length, num_samples):
            raise ValueError(f"noise must be [{length}, {num_samples}]")
        return noise

    def _sample_scan(self, num_samples, length, temp, seed, noise, save_states):
        noise = self._noise(num_samples, length, temp, seed, noise)
        be = self._prepare(num_samples, length + 1, train=save_states)
        return be, be.rho_sample(noise, save_states=save_states)

    def sample(self, num_samples, length, temp=1, seed=None, noise=None):
        """model.py:103-116: waveforms [num_samples, length] = A * running sum of the sampled increments."""
        return self._sample_scan(num_samples, length, temp, seed, noise, False)[1]

    def rho_evolve_with_sampling(self, num_samples, length, temp=1, seed=None, noise=None) -> np.ndarray:
        """model.py:86-92: rho after every sampled step, [num_samples, length, D, D]."""
        be, _ = self._sample_scan(num_samples, length, temp, seed, noise, True)
        return be.rho_states(num_samples, length, want_rho=True)

    def purity(self, num_samples, length, temp=1, seed=None, noise=None) -> np.ndarray:
        """model.py:94-101: tr rho^2 along sampled paths, [num_samples, length]."""
        be, _ = self._sample_scan(num_samples, length, temp, seed, noise, True)
        return be.rho_states(num_samples, length, want_rho=False, want_purity=True)


class LegacyAudioMPS:
    """The legacy ``AudioMPS`` arithmetic (reconstructed from logging/graph.pbtxt): real
    variables H, R [D,D] (glorot-uniform), H_s = tril(H) + tril(H)^T, psi_0 = e_0,
    loss = mean_b sum_k (x_k - 2 Re<psi|R|psi>)^2 / 2 evaluated BEFORE the update,
    psi' = psi + dt (-i H_s - R^T R / 2) psi + dt x R psi, normalised.  Trained with Adam(1e-3) on the bare loss
    (training_estimators.py:64-68)."""

    VARIABLE_NAMES = ("H", "R")

    def __init__(self, bond_d, dt, batch_size=8, data_iterator=None, seed: int = 0, backend=None):
        self.bond_d = int(bond_d)
        self.delta_t = float(dt)
        self.batch_size = batch_size
        self.data_iterator = data_iterator
        rng = np.random.default_rng(seed)
        lim = math.sqrt(6.0 / (2 * self.bond_d))            # glorot_uniform, graph.pbtxt:9076-9114
        self.variables: Dict[str, np.ndarray] = {
            "H": rng.uniform(-lim, lim, (self.bond_d, self.bond_d)).astype(np.float32),
            "R": rng.uniform(-lim, lim, (self.bond_d, self.bond_d)).astype(np.float32)}
        self._backend = backend

    def _get_backend(self):
        if self._backend is None:
            from .scan import HipScan
            self._backend = HipScan(self.bond_d)
        return self._backend

    @property
    def H_s(self) -> np.ndarray:
        L = np.tril(self.variables["H"])
        return (L + L.T).astype(np.float32)                  # graph.pbtxt:9442-9605

    @property
    def Q(self) -> np.ndarray:
        R = self.variables["R"]
        RtR = (R.T @ R).astype(np.float32)                   # MatMul transpose_a, :13041
        return ((np.complex64(-1j) * self.H_s.astype(np.complex64) - (RtR / np.float32(2)).astype(np.complex64))
                * np.complex64(np.float32(self.delta_t))).astype(np.complex64)

    def _audio(self, data):
        import torch
        data = self.data_iterator if data is None else data
        if callable(data):
            data = data()
        be = self._get_backend()
        t = data if isinstance(data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))
        return t.to(be.device, dtype=torch.float32).contiguous()

    def loss_per_clip(self, data=None) -> np.ndarray:
        audio = self._audio(data)
        be = self._get_backend()
        be.legacy_set_params(self.variables["R"], self.Q, self.delta_t, audio.shape[0], audio.shape[1], train=False)
        return be.legacy_forward(audio).detach().cpu().numpy()

    @property
    def loss(self) -> np.float32:
        return np.float32(np.mean(self.loss_per_clip(), dtype=np.float32))

    def flat_size(self) -> int:
        """Length of the buffer grad_sums() returns (cmps_legacy_loss_bwd's layout)."""
        return 3 * self.bond_d * self.bond_d + 1

    def grad_sums(self, data=None):
        audio = self._audio(data)
        be = self._get_backend()
        be.legacy_set_params(self.variables["R"], self.Q, self.delta_t, audio.shape[0], audio.shape[1], train=True)
        be.legacy_forward(audio, save_for_bwd=True)
        return be.legacy_backward(), audio.shape[0]

    def chain_rule(self, flat_sums, global_batch: int, with_reg: bool = False):
        """(dQ, dR_direct) sums -> gradients w.r.t. H and R  (adjoint of Q = dt (-i H_s - R^T R / 2), H_s = L + L^T)."""
        D = self.bond_d
        g = np.asarray(flat_sums, dtype=np.float64) / float(global_batch)
        DD = D * D
        Qbar = (g[:DD] + 1j * g[DD:2 * DD]).reshape(D, D)
        Rdir = g[2 * DD:3 * DD].reshape(D, D)
        loss = g[3 * DD]
        dt = float(np.float32(self.delta_t))
        R = self.variables["R"].astype(np.float64)
        Hs_bar = -dt * Qbar.imag
        M_bar = -(dt / 2.0) * Qbar.real
        gR = Rdir + R @ (M_bar + M_bar.T)
        gH = np.tril(Hs_bar + Hs_bar.T)
        return np.float32(loss), {"H": gH.astype(np.float32), "R": gR.astype(np.float32)}

    def loss_and_grads(self, data=None, with_reg: bool = False):
        flat, B = self.grad_sums(data)
        return self.chain_rule(flat.detach().cpu().numpy(), B)


class AudioMPS(PsiCMPS):
    """Legacy surface: ``AudioMPS(bond_d, dt, batch_size, data_iterator=..., mixed=...)``
    (training_estimators.py:43-45; ``AudioMPS(bond_d, delta_t=..., data_iterator=...)`` in
    notebooks/testing-AudioMPS.ipynb:268).  The class body no longer exists in the reference's model.py;
    PsiCMPS is its successor, so this wraps PsiCMPS with train.py's remaining hyper-parameters."""

    def __new__(cls, *args, arithmetic: str = "psi", **kwargs):
        # arithmetic="legacy": the model this surface originally named (LegacyAudioMPS)
        if arithmetic == "legacy":
            kwargs.pop("mixed", None)
            dt = kwargs.pop("delta_t", None)
            if dt is not None and len(args) < 2:
                kwargs["dt"] = dt
            kwargs.pop("hparams", None)
            return LegacyAudioMPS(*args, **kwargs)
        # mixed=True: the density-matrix model (RhoCMPS); a non-instance return skips AudioMPS.__init__
        names = ("bond_d", "dt", "batch_size", "data_iterator", "mixed")
        bound = dict(zip(names, args))
        bound.update(kwargs)
        if bound.get("mixed", False):
            hp = cls._hparams(bound["bond_d"], bound.get("dt"), bound.get("batch_size", 8), bound.get("delta_t"),
                              bound.get("hparams"))
            extra = {k: v for k, v in bound.items() if k not in names + ("delta_t", "hparams")}
            return RhoCMPS(hp, data_iterator=bound.get("data_iterator"), **extra)
        return super().__new__(cls)

    @staticmethod
    def _hparams(bond_d, dt, batch_size, delta_t, hparams):
        if dt is None:
            dt = delta_t
        if dt is None:
            raise TypeError("AudioMPS needs dt (or delta_t)")
        hp = hparams if hparams is not None else HParams()
        return HParams(**{**hp.values(), "bond_dim": int(bond_d), "delta_t": float(dt),
                          "minibatch_size": int(batch_size)})

    def __init__(self, bond_d, dt=None, batch_size=8, data_iterator=None, mixed=False, delta_t=None,
                 hparams: Optional[HParams] = None, arithmetic: str = "psi", **kwargs):
        hp = self._hparams(bond_d, dt, batch_size, delta_t, hparams)
        super().__init__(hp, data_iterator=data_iterator, **kwargs)
