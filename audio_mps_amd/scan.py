"""Host driver of the HIP scan: owns a libcmps handle and a device workspace (a torch uint8 tensor used
purely as an allocation), and exposes the two operations the model needs.

PyTorch appears here only as plumbing: device memory, the current HIP stream, host<->device copies.
All arithmetic of the hot path happens in libcmps.so (audio_mps_amd/csrc/*.hip).
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _capi


@dataclass
class EffectiveParams:
    """What CMPS.__init__ / PsiCMPS.__init__ hand to the scan (model.py:41-52, 221-222)."""
    R: np.ndarray        # [D, D] complex64, after the diagonal removal of model.py:42
    freqs: np.ndarray    # [D] float32
    psi0: np.ndarray     # [D] complex64, normalised
    A: float
    sigma: float
    delta_t: float


def grad_size(D: int) -> int:
    return 2 * D * D + 3 * D + 2


def unpack_grad(flat, D: int):
    """Flat buffer of cmps_psi_loss_bwd -> dict (sums over clips)."""
    g = np.asarray(flat)
    DD = D * D
    return {
        "Rbar": (g[:DD] + 1j * g[DD:2 * DD]).reshape(D, D),
        "fbar": g[2 * DD:2 * DD + D],
        "psi0bar": g[2 * DD + D:2 * DD + 2 * D] + 1j * g[2 * DD + 2 * D:2 * DD + 3 * D],
        "Abar": g[2 * DD + 3 * D],
        "loss_sum": g[2 * DD + 3 * D + 1],
    }


class HipScan:
    """The MI355X backend: per-clip loss and parameter-gradient sums for a batch resident on the GPU."""

    name = "hip"

    def __init__(self, D: int, device: Optional[torch.device] = None, variant: int = _capi.CMPS_VARIANT_AUTO,
                 rank1: Optional[int] = None):
        self._lib = _capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipScan needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.D = int(D)
        h = ctypes.c_void_p()
        code = self._lib.cmps_create(self.D, ctypes.byref(h))
        if code != _capi.CMPS_OK:
            raise _capi.CmpsError(code, f"cmps_create(D={D}) failed")
        self._h = h
        _capi.check(self._h, self._lib.cmps_set_variant(self._h, int(variant)))
        if rank1 is not None:
            self.set_rank1(rank1)
        if os.environ.get("CMPS_BWD_WAVES"):            # diagnostic (like CMPS_LIB): A/B of the one- and two-wave reverse scans without code changes
            _capi.check(self._h, self._lib.cmps_set_option(self._h, _capi.CMPS_OPT_BWD_WAVES, int(os.environ["CMPS_BWD_WAVES"])))
        self._ws = None
        self._ws_key = None
        self._param_buf = torch.empty(2 * D * D + 3 * D, dtype=torch.float32, device=self.device)
        self._param_host = torch.empty(2 * D * D + 3 * D, dtype=torch.float32).pin_memory()
        self._param_evt = None
        self._B = self._T = 0
        self._audio = None
        self._loss = None
        self._grad = torch.empty(grad_size(D), dtype=torch.float32, device=self.device)
        self.f16_fallbacks = 0        # loss_and_grad_sums(check=True) re-runs that cmps_psi_grad_status asked for
        self.timing = None            # a list: forward() / backward() then append HIP-event pairs around their launches (bench.py)

    # ------------------------------------------------------------------
    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.cmps_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def variant(self) -> int:
        return int(self._lib.cmps_get_variant(self._h))

    def set_rank1(self, mode: int):
        """cmps_set_option(CMPS_OPT_RANK1): arithmetic of the rank-1 gradient updates (wave kernels, D <= 32)."""
        _capi.check(self._h, self._lib.cmps_set_option(self._h, _capi.CMPS_OPT_RANK1, int(mode)))

    @property
    def rank1(self) -> int:
        return int(self._lib.cmps_get_option(self._h, _capi.CMPS_OPT_RANK1))

    def set_wide_chain(self, mode: int):
        """cmps_set_option(CMPS_OPT_WIDE_CHAIN): 0 = fp32 VALU chain, 1 = fp16 x 2 split operands on the matrix cores (wide kernels' training forward)."""
        _capi.check(self._h, self._lib.cmps_set_option(self._h, _capi.CMPS_OPT_WIDE_CHAIN, int(mode)))

    @property
    def wide_chain(self) -> int:
        return int(self._lib.cmps_get_option(self._h, _capi.CMPS_OPT_WIDE_CHAIN))

    @property
    def effective_rank1(self) -> int:
        """The arithmetic the selected kernels run for the current option value (include/cmps.h): the wide kernels' gradient GEMM
        knows two bf16 pieces, three bf16 pieces (also for EXACT_F32) and two fp16 pieces (also for DEFAULT); the wave reverse scan
        exact fp32, two bf16 pieces, two fp16 pieces (also for DEFAULT) and three bf16 pieces.  (The 16-row kernels of D <= 16 always
        use exact fp32 MFMAs and the legacy mode maps the fp16 form to three bf16 pieces: this property describes the 32-row kernel.)"""
        mode, wide = self.rank1, self.variant == _capi.CMPS_VARIANT_WIDE
        if getattr(self, "_legacy_buf", None) is not None and not wide:        # a handle in legacy mode (legacy_set_params): the wave
            # reverse scan's legacy instance has no fp16 form -- F16X2 / DEFAULT run three bf16 pieces (include/cmps.h's table)
            return mode if mode in (_capi.CMPS_RANK1_EXACT_F32, _capi.CMPS_RANK1_BF16X2) else _capi.CMPS_RANK1_BF16X3
        if wide:
            return {0: _capi.CMPS_RANK1_BF16X3, 4: _capi.CMPS_RANK1_F16X2}.get(mode, mode)
        if mode == _capi.CMPS_RANK1_DEFAULT:
            return _capi.CMPS_RANK1_F16X2
        return mode if mode in (_capi.CMPS_RANK1_EXACT_F32, _capi.CMPS_RANK1_BF16X2, _capi.CMPS_RANK1_F16X2) else _capi.CMPS_RANK1_BF16X3

    def kernel_events(self, on: bool):
        """cmps_set_option(CMPS_OPT_KERNEL_EVENTS): bracket every kernel of forward() / backward() with HIP events (a measurement aid,
        used by bench.py outside its timed region)."""
        _capi.check(self._h, self._lib.cmps_set_option(self._h, _capi.CMPS_OPT_KERNEL_EVENTS, 1 if on else 0))

    def kernel_times(self) -> dict:
        """cmps_kernel_times: {kernel name: (summed ms, launches)} since the last call, in first-launch order (synchronises)."""
        import ctypes
        cap = 32
        names = ctypes.create_string_buffer(2048)
        ms = (ctypes.c_float * cap)()
        calls = (ctypes.c_int * cap)()
        n = self._lib.cmps_kernel_times(self._h, names, 2048, ms, calls, cap)
        if n < 0:
            raise _capi.CmpsError(n, self._lib.cmps_last_error(self._h).decode())
        keys = names.value.decode().split("\n") if n else []
        return {k: (float(ms[i]), int(calls[i])) for i, k in enumerate(keys)}

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace_bytes(self, B: int, T: int, train: bool) -> int:
        return int(self._lib.cmps_workspace_bytes(self.D, B, T, _capi.CMPS_WS_TRAIN if train else _capi.CMPS_WS_FWD_ONLY))

    def _ensure_ws(self, B: int, T: int, train: bool):
        key = (B, T, train)
        if self._ws_key != key:
            nbytes = self.workspace_bytes(B, T, train)
            if nbytes == 0:
                raise ValueError(f"invalid shape for the scan: D={self.D}, B={B}, T={T}")
            self._ws = None
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self._ws_key = key
            self._ws_fresh = True         # the allocator may hand back the old address: the cached tables are gone
        base = self._ws.data_ptr()
        return (base + 255) // 256 * 256, self._ws.numel() - 256

    # ------------------------------------------------------------------
    def set_params(self, p: EffectiveParams, B: int, T: int, train: bool = True):
        """cmps_set_params: upload the effective parameters and rebuild the derived tables."""
        D = self.D
        R = np.asarray(p.R)
        if R.shape != (D, D):
            raise ValueError(f"R must be [{D},{D}]")
        if self._param_evt is not None:
            self._param_evt.synchronize()  # the previous upload must have left the pinned buffer
        host = self._param_host.numpy()
        DD = D * D
        host[:DD] = R.real.astype(np.float32).ravel()
        host[DD:2 * DD] = R.imag.astype(np.float32).ravel()
        host[2 * DD:2 * DD + D] = np.asarray(p.freqs, dtype=np.float32)
        psi0 = np.asarray(p.psi0)
        host[2 * DD + D:2 * DD + 2 * D] = psi0.real.astype(np.float32)
        host[2 * DD + 2 * D:2 * DD + 3 * D] = psi0.imag.astype(np.float32)
        self._param_buf.copy_(self._param_host, non_blocking=True)
        self._param_evt = torch.cuda.Event()
        self._param_evt.record(torch.cuda.current_stream(self.device))
        ws_ptr, ws_bytes = self._ensure_ws(B, T, train)
        base = self._param_buf.data_ptr()
        f4 = 4
        flags = _capi.CMPS_WS_TRAIN if train else _capi.CMPS_WS_FWD_ONLY
        if getattr(self, "_ws_fresh", False):
            self._ws_fresh = False              # (re)allocated since the last call: every table is rebuilt (the library's default)
        else:
            flags |= _capi.CMPS_WS_REUSE_TABLES  # this object owns the workspace and has not touched it: keep the time table
        _capi.check(self._h, self._lib.cmps_set_params(
            self._h, base, base + DD * f4, base + 2 * DD * f4, base + (2 * DD + D) * f4,
            base + (2 * DD + 2 * D) * f4, float(p.A), float(p.sigma), float(p.delta_t), int(T), int(B),
            flags, ws_ptr, ws_bytes, self._stream()))
        self._B, self._T, self._train = B, T, train

    def set_params_dev(self, params: torch.Tensor, sigma: float, delta_t: float, B: int, T: int, train: bool = True):
        """cmps_set_params_dev: the effective parameters (A included) are read from the device buffer `params`
        [2 D^2 + 3 D + 1] that cmps_psi_apply_step wrote -- no host copy, no synchronisation."""
        if not (params.is_cuda and params.dtype == torch.float32 and params.numel() == 2 * self.D * self.D + 3 * self.D + 1):
            raise ValueError("params must be a float32 CUDA tensor of 2 D^2 + 3 D + 1 elements")
        ws_ptr, ws_bytes = self._ensure_ws(B, T, train)
        flags = _capi.CMPS_WS_TRAIN if train else _capi.CMPS_WS_FWD_ONLY
        if getattr(self, "_ws_fresh", False):
            self._ws_fresh = False              # (re)allocated since the last call: every table is rebuilt (the library's default)
        else:
            flags |= _capi.CMPS_WS_REUSE_TABLES  # this object owns the workspace and has not touched it: keep the time table
        _capi.check(self._h, self._lib.cmps_set_params_dev(self._h, params.data_ptr(), float(sigma), float(delta_t), int(T), int(B),
                                                           flags, ws_ptr, ws_bytes, self._stream()))
        self._B, self._T, self._train = B, T, train

    def apply_step(self, vars_: torch.Tensor, m: torch.Tensor, v: torch.Tensor, grad_sums: Optional[torch.Tensor], global_batch: int,
                   lr_t: float, beta1: float, beta2: float, eps: float, h_reg: float, r_reg: float, c_r: float, c_h: float,
                   with_reg: bool, params: torch.Tensor, losses: torch.Tensor):
        """cmps_psi_apply_step: chain rule + regularisers + Adam + next effective parameters, on the device (grad_sums None: only
        the effective parameters of `vars_`)."""
        if getattr(self, "_opt_scratch", None) is None:
            n = int(self._lib.cmps_apply_step_scratch_bytes(self.D))
            self._opt_scratch = torch.empty((n + 7) // 8, dtype=torch.float64, device=self.device)
        _capi.check(self._h, self._lib.cmps_psi_apply_step(
            self._h, vars_.data_ptr(), m.data_ptr(), v.data_ptr(), grad_sums.data_ptr() if grad_sums is not None else None,
            float(max(global_batch, 1)), float(lr_t), float(beta1), float(beta2), float(eps), float(h_reg), float(r_reg), float(c_r),
            float(c_h), 1 if with_reg else 0, params.data_ptr(), losses.data_ptr(), self._opt_scratch.data_ptr(), self._stream()))

    def _check_audio(self, audio: torch.Tensor):
        if not (isinstance(audio, torch.Tensor) and audio.is_cuda and audio.dtype == torch.float32
                and audio.dim() == 2 and audio.is_contiguous()):
            raise ValueError("audio must be a contiguous float32 CUDA tensor [B, T]")
        B, T = audio.shape
        if T != self._T or B > self._B:
            raise ValueError(f"audio shape {tuple(audio.shape)} does not fit set_params(B={self._B}, T={self._T})")
        return B, T

    def forward(self, audio: torch.Tensor, save_for_bwd: bool = False) -> torch.Tensor:
        """Per-clip loss [B] (device tensor): cmps_psi_loss_fwd."""
        B, T = self._check_audio(audio)
        if self._loss is None or self._loss.numel() != B:
            self._loss = torch.empty(B, dtype=torch.float32, device=self.device)
        ev = self._event_pair()
        _capi.check(self._h, self._lib.cmps_psi_loss_fwd(
            self._h, audio.data_ptr(), B, T, self._loss.data_ptr(), 1 if save_for_bwd else 0, self._stream()))
        self._event_close("fwd", ev)
        self._audio = audio
        return self._loss

    # HIP events on the stream the kernels are launched on (torch's current stream is the one handed to the C ABI); they are read
    # only after the caller has synchronised (timing_ms), so recording them never stalls the host
    def _event_pair(self):
        if self.timing is None:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.device))
        return e0, e1

    def _event_close(self, what, ev):
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(self.device))
            self.timing.append((what, ev[0], ev[1]))

    def timing_ms(self):
        """{'fwd': [ms ...], 'bwd': [ms ...]} of the launches recorded since `timing` was set to a list (synchronises)."""
        torch.cuda.synchronize(self.device)
        out = {"fwd": [], "bwd": []}
        for what, e0, e1 in self.timing or []:
            out[what].append(e0.elapsed_time(e1))
        return out

    def backward(self) -> torch.Tensor:
        """Flat gradient sums [2D^2+3D+2] (device tensor): cmps_psi_loss_bwd after forward(save_for_bwd=True)."""
        audio = self._audio
        if audio is None:
            raise RuntimeError("backward() needs forward(save_for_bwd=True) first")
        B, T = audio.shape
        ev = self._event_pair()
        _capi.check(self._h, self._lib.cmps_psi_loss_bwd(
            self._h, audio.data_ptr(), B, T, self._grad.data_ptr(), self._stream()))
        self._event_close("bwd", ev)
        return self._grad

    def grad_status(self):
        """cmps_psi_grad_status: (code, sticky flags).  Waits for the stream.  code = CMPS_ERR_F16_RANGE when the last backward()
        left Inf / NaN in the gradient next to finite per-clip losses (an fp16-split operand out of its scaled range)."""
        sticky = ctypes.c_int(0)
        code = int(self._lib.cmps_psi_grad_status(self._h, ctypes.byref(sticky), self._stream()))
        if code not in (_capi.CMPS_OK, _capi.CMPS_ERR_F16_RANGE):
            _capi.check(self._h, code)
        return code, int(sticky.value)

    def loss_and_grad_sums(self, audio: torch.Tensor, check: bool = False):
        """forward(save) + backward().  check=True (the host trainer and the model's gradient accessors: they read the result
        on the host anyway) asks cmps_psi_grad_status afterwards and, on CMPS_ERR_F16_RANGE, takes the documented fallback once:
        CMPS_OPT_RANK1 = BF16X3, CMPS_OPT_WIDE_CHAIN = VALU (no fp16 piece anywhere), the same two calls again; the handle's
        options are restored.  A gradient that is non-finite in that arithmetic as well is returned as it is (it propagates, as in
        the reference)."""
        loss = self.forward(audio, save_for_bwd=True)
        grad = self.backward()
        if check:
            code, _ = self.grad_status()
            if code == _capi.CMPS_ERR_F16_RANGE and (self.effective_rank1 == _capi.CMPS_RANK1_F16X2 or
                                                     (self.variant == _capi.CMPS_VARIANT_WIDE and self.wide_chain != _capi.CMPS_WIDE_CHAIN_VALU)):
                import warnings
                warnings.warn("libcmps: " + self._lib.cmps_last_error(self._h).decode() + " -- re-running this batch with bf16x3 pieces")
                keep = (self.rank1, self.wide_chain)
                self.f16_fallbacks += 1
                try:
                    self.set_rank1(_capi.CMPS_RANK1_BF16X3)
                    self.set_wide_chain(_capi.CMPS_WIDE_CHAIN_VALU)
                    loss = self.forward(audio, save_for_bwd=True)
                    grad = self.backward()
                finally:
                    self.set_rank1(keep[0])
                    self.set_wide_chain(keep[1])
        return loss, grad

    # ------------------------------------------------------------------
    def update_ancilla(self, psi: np.ndarray, signal: np.ndarray, t: float) -> np.ndarray:
        """PsiCMPS._update_ancilla_psi for a batch of states (host arrays in, host array out)."""
        psi = np.asarray(psi, dtype=np.complex64)
        B, D = psi.shape
        if D != self.D:
            raise ValueError("psi has the wrong bond dimension")
        inter = np.stack([psi.real, psi.imag], axis=-1).astype(np.float32)
        d_in = torch.from_numpy(np.ascontiguousarray(inter)).to(self.device)
        d_sig = torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float32)).to(self.device)
        d_out = torch.empty_like(d_in)
        _capi.check(self._h, self._lib.cmps_psi_update_ancilla(
            self._h, d_in.data_ptr(), d_sig.data_ptr(), float(t), B, d_out.data_ptr(), self._stream()))
        o = d_out.cpu().numpy()
        return (o[..., 0] + 1j * o[..., 1]).astype(np.complex64)

    def states(self) -> np.ndarray:
        """Normalised lab-frame psi after every step, [B, T-1, D] complex64 (psi_evolve_with_data)."""
        audio = self._audio
        if audio is None:
            raise RuntimeError("states() needs forward(save_for_bwd=True) first")
        B, T = audio.shape
        out = torch.empty((B, T - 1, self.D, 2), dtype=torch.float32, device=self.device)
        _capi.check(self._h, self._lib.cmps_psi_states(self._h, B, T, out.data_ptr(), self._stream()))
        o = out.cpu().numpy()
        return (o[..., 0] + 1j * o[..., 1]).astype(np.complex64)

    def sample(self, noise: np.ndarray) -> np.ndarray:
        """PsiCMPS.sample for pre-drawn noise [length, n] (the reference's layout) -> waveforms [n, length]."""
        noise = np.asarray(noise, dtype=np.float32)
        length, n = noise.shape
        d_noise = torch.from_numpy(np.ascontiguousarray(noise.T)).to(self.device)
        d_out = torch.empty((n, length), dtype=torch.float32, device=self.device)
        _capi.check(self._h, self._lib.cmps_psi_sample(self._h, d_noise.data_ptr(), n, length, d_out.data_ptr(),
                                                       self._stream()))
        return d_out.cpu().numpy()

    # ------------------------------------------------------------------
    # legacy AudioMPS arithmetic
    # ------------------------------------------------------------------
    def legacy_set_params(self, R: np.ndarray, Q: np.ndarray, delta_t: float, B: int, T: int, train: bool = True):
        D = self.D
        DD = D * D
        buf = torch.from_numpy(np.concatenate([np.asarray(R, dtype=np.float32).ravel(),
                                               np.asarray(Q).real.astype(np.float32).ravel(),
                                               np.asarray(Q).imag.astype(np.float32).ravel()])).to(self.device)
        self._legacy_buf = buf
        ws_ptr, ws_bytes = self._ensure_ws(B, T, train)
        base = buf.data_ptr()
        _capi.check(self._h, self._lib.cmps_legacy_set_params(
            self._h, base, base + DD * 4, base + 2 * DD * 4, float(delta_t), int(T), int(B),
            _capi.CMPS_WS_TRAIN if train else _capi.CMPS_WS_FWD_ONLY, ws_ptr, ws_bytes, self._stream()))
        self._B, self._T, self._train = B, T, train

    def legacy_forward(self, audio: torch.Tensor, save_for_bwd: bool = False) -> torch.Tensor:
        B, T = self._check_audio(audio)
        if self._loss is None or self._loss.numel() != B:
            self._loss = torch.empty(B, dtype=torch.float32, device=self.device)
        _capi.check(self._h, self._lib.cmps_legacy_loss_fwd(
            self._h, audio.data_ptr(), B, T, self._loss.data_ptr(), 1 if save_for_bwd else 0, self._stream()))
        self._audio = audio
        return self._loss

    def legacy_backward(self) -> torch.Tensor:
        audio = self._audio
        B, T = audio.shape
        if getattr(self, "_legacy_grad", None) is None:
            self._legacy_grad = torch.empty(3 * self.D * self.D + 1, dtype=torch.float32, device=self.device)
        _capi.check(self._h, self._lib.cmps_legacy_loss_bwd(
            self._h, audio.data_ptr(), B, T, self._legacy_grad.data_ptr(), self._stream()))
        return self._legacy_grad

    # ------------------------------------------------------------------
    # RhoCMPS: the density matrix carried as its `rank` columns
    # ------------------------------------------------------------------
    def rho_grad_size(self, rank: int) -> int:
        return grad_size(self.D) + 2 * rank * self.D

    def rho_set_state(self, phi: np.ndarray, B: int, T: int, train: bool = True):
        """cmps_rho_set_state: phi [rank, D] complex, rho_0 = sum_a phi_a phi_a^dagger.  After set_params."""
        phi = np.asarray(phi)
        r, D = phi.shape
        if D != self.D:
            raise ValueError("phi has the wrong bond dimension")
        flags = _capi.CMPS_WS_TRAIN if train else _capi.CMPS_WS_FWD_ONLY
        nbytes = int(self._lib.cmps_rho_workspace_bytes(D, r, B, T, flags))
        if nbytes == 0:
            raise ValueError(f"invalid shape for the rho scan: D={D}, rank={r}, B={B}, T={T}")
        key = (r, B, T, train)
        if getattr(self, "_rho_ws_key", None) != key:
            self._rho_ws = None
            self._rho_ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self._rho_ws_key = key
        base = (self._rho_ws.data_ptr() + 255) // 256 * 256
        host = np.concatenate([phi.real.astype(np.float32).ravel(), phi.imag.astype(np.float32).ravel()])
        self._phi_buf = torch.from_numpy(host).to(self.device)
        pb = self._phi_buf.data_ptr()
        _capi.check(self._h, self._lib.cmps_rho_set_state(
            self._h, pb, pb + r * D * 4, r, int(T), int(B), flags, base, self._rho_ws.numel() - 256, self._stream()))
        self._rho_rank, self._rho_B, self._rho_T = r, B, T

    def rho_forward(self, audio: torch.Tensor, save_for_bwd: bool = False) -> torch.Tensor:
        B, T = self._check_audio(audio)
        if self._loss is None or self._loss.numel() != B:
            self._loss = torch.empty(B, dtype=torch.float32, device=self.device)
        _capi.check(self._h, self._lib.cmps_rho_loss_fwd(
            self._h, audio.data_ptr(), B, T, self._loss.data_ptr(), 1 if save_for_bwd else 0, self._stream()))
        self._audio = audio
        return self._loss

    def rho_backward(self) -> torch.Tensor:
        audio = self._audio
        if audio is None:
            raise RuntimeError("rho_backward() needs rho_forward(save_for_bwd=True) first")
        B, T = audio.shape
        n = self.rho_grad_size(self._rho_rank)
        if getattr(self, "_rho_grad", None) is None or self._rho_grad.numel() != n:
            self._rho_grad = torch.empty(n, dtype=torch.float32, device=self.device)
        _capi.check(self._h, self._lib.cmps_rho_loss_bwd(
            self._h, audio.data_ptr(), B, T, self._rho_grad.data_ptr(), self._stream()))
        return self._rho_grad

    def rho_loss_and_grad_sums(self, audio: torch.Tensor):
        loss = self.rho_forward(audio, save_for_bwd=True)
        return loss, self.rho_backward()

    def rho_update_ancilla(self, rho: np.ndarray, signal: np.ndarray, t: float) -> np.ndarray:
        """RhoCMPS._update_ancilla_rho for a batch of matrices [B, D, D] (host in, host out)."""
        rho = np.asarray(rho, dtype=np.complex64)
        B, D, D2 = rho.shape
        if D != self.D or D2 != self.D:
            raise ValueError("rho has the wrong bond dimension")
        inter = np.stack([rho.real, rho.imag], axis=-1).astype(np.float32)
        d_in = torch.from_numpy(np.ascontiguousarray(inter)).to(self.device)
        d_sig = torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float32)).to(self.device)
        d_out = torch.empty_like(d_in)
        _capi.check(self._h, self._lib.cmps_rho_update_ancilla(
            self._h, d_in.data_ptr(), d_sig.data_ptr(), float(t), B, d_out.data_ptr(), self._stream()))
        o = d_out.cpu().numpy()
        return (o[..., 0] + 1j * o[..., 1]).astype(np.complex64)

    def rho_sample(self, noise: np.ndarray, save_states: bool = False) -> np.ndarray:
        """RhoCMPS.sample for pre-drawn noise [length, n] -> waveforms [n, length]."""
        noise = np.asarray(noise, dtype=np.float32)
        length, n = noise.shape
        d_noise = torch.from_numpy(np.ascontiguousarray(noise.T)).to(self.device)
        d_out = torch.empty((n, length), dtype=torch.float32, device=self.device)
        _capi.check(self._h, self._lib.cmps_rho_sample(self._h, d_noise.data_ptr(), n, length, d_out.data_ptr(),
                                                       1 if save_states else 0, self._stream()))
        return d_out.cpu().numpy()

    def rho_states(self, B: int, steps: int, want_rho: bool = True, want_purity: bool = False):
        """Lab-frame rho [B, steps, D, D] and/or purity [B, steps] of the last saved scan."""
        D = self.D
        d_rho = torch.empty((B, steps, D, D, 2), dtype=torch.float32, device=self.device) if want_rho else None
        d_pur = torch.empty((B, steps), dtype=torch.float32, device=self.device) if want_purity else None
        _capi.check(self._h, self._lib.cmps_rho_states(
            self._h, B, steps, d_rho.data_ptr() if want_rho else None, d_pur.data_ptr() if want_purity else None,
            self._stream()))
        out = []
        if want_rho:
            o = d_rho.cpu().numpy()
            out.append((o[..., 0] + 1j * o[..., 1]).astype(np.complex64))
        if want_purity:
            out.append(d_pur.cpu().numpy())
        return out[0] if len(out) == 1 else tuple(out)
