"""Random robustness sweep: random shapes, parameter scales, amplitudes and sampling rates through every kernel family against
the oracles.  Shared by `tests/test_gpu_sweep.py` (fixed seeds, asserted bars: what the driver runs) and
`scripts/random_sweep.py` (any seed, prints the worst case per family).

Why it is a test: rounds 2-3 were green with a wrong `Qbar` in `k_bwd_wave` (a VALU write one wait state in front of
the MFMA that read it) because no fixed test made Q = -(dt sigma^2 / 2) R^dagger R (/root/reference/model.py:312)
visible in float32; this sweep drew sigma = 0.36 with a large R and found it -- but only ran by hand.

Every record is (family, quantity, error, bar, configuration).  Bars: the float32 families' parity bars of tests/test_gpu_parity.py
(loss 1e-5 of max(|loss_b|, 1), gradients 1e-4 of each tensor's max) against the float32 C restatement; the bf16 pair kernels' bars of
tests/test_gpu_pair.py against the bf16-emulating oracle; dA apart everywhere: it is ONE scalar made of two large cancelling sums, so
its bar is the float32 oracle's own distance to its float64 twin (never below the tensor bar)."""
from __future__ import annotations

import numpy as np

from oracle import cmps_oracle as O, c_oracle as C
from _util import make_audio, c_oracle_run, rel_inf, oracle_hparams, oracle_variables

LOSS_BAR, GRAD_BAR = 1e-5, 1e-4
PAIR_LOSS_BAR, PAIR_GRAD_BAR = 3e-4, 2e-3          # tests/test_gpu_pair.py, vs oracle.psi_bf16_scan
DEFAULT_COUNTS = {"psi": 60, "wide": 16, "step": 8, "pair": 14, "rho": 34, "legacy": 10}


def _loss_err(per, ref):
    return float(np.max(np.abs(per - ref) / np.maximum(np.abs(ref), 1)))


def _dA_bar(g32, g64, floor):
    """dA's bar: the float32 restatement's own distance from float64 on this draw (x 2), never below `floor`."""
    return max(floor, 2.0 * rel_inf(g32, g64))


def sweep_psi(rng, n, out):
    """Pure-state wave (16- and 32-row) / block kernels vs the C restatement (model.py:257-334)."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan, unpack_grad
    for it in range(n):
        D = int(rng.integers(1, 33)); T = int(rng.integers(2, 700)); B = int(rng.integers(1, 14))
        sigma = float(10 ** rng.uniform(-4, 0)); rs = float(10 ** rng.uniform(-1.5, 0)); variant = int(rng.choice([1, 2]))
        hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, A=float(10 ** rng.uniform(0, 2)))
        if it % 4 == 1:                                    # other sampling rates: 3 kHz ... 100 kHz (rotation table, dt_k, the scale of Q)
            hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, A=hp.A, delta_t=float(10 ** rng.uniform(-5, -3.5)))
        amp = float(10 ** rng.uniform(-3, 0.3))           # amplitudes over three decades, silent stretches: the fp16 scales
        audio = (make_audio(B, T, hp.delta_t, it) * np.float32(amp)).astype(np.float32)
        if T > 40 and it % 3 == 0:
            audio[:, : T // 3] = 0.0
        m = PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D, variant=variant, rank1=int(rng.choice([2, 3, 4]))))
        m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
        per = m.loss_per_clip(); flat, _ = m.grad_sums(); g = unpack_grad(flat.cpu().numpy(), D)
        ref = c_oracle_run(m, audio, "f32"); gr = C.unpack_grad(ref["grad"], D)
        if not np.all(np.isfinite(ref["loss_per_clip"])):
            continue
        g64 = C.unpack_grad(c_oracle_run(m, audio, "f64")["grad"], D)
        cfg = (D, T, B, round(sigma, 5), round(rs, 3), variant, round(amp, 4), m._get_backend().effective_rank1, float(f"{hp.delta_t:.3g}"))
        assert np.all(np.isfinite(per)) and np.all(np.isfinite(flat.cpu().numpy())), ("psi non-finite", cfg)
        out.append(("psi", "loss", _loss_err(per, ref["loss_per_clip"]), LOSS_BAR, cfg))
        out.append(("psi", "grad", max(rel_inf(g[k], gr[k]) for k in ("Rbar", "fbar", "psi0bar")), GRAD_BAR, cfg))
        out.append(("psi", "dA", rel_inf(g["Abar"], gr["Abar"]), _dA_bar(gr["Abar"], g64["Abar"], GRAD_BAR), cfg))


def sweep_wide(rng, n, out):
    """Wide kernels (float32, 32 < D <= 128; AUTO), all three CMPS_OPT_WIDE_CHAIN values and every GEMM arithmetic."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan, unpack_grad
    from audio_mps_amd.data import synthetic_audio
    for it in range(n):
        D = int(rng.integers(33, 129)); T = int(rng.integers(2, 500)); B = int(rng.integers(1, 8))
        sigma = float(10 ** rng.uniform(-4, 0)); rs = float(10 ** rng.uniform(-1.5, -0.3))
        inp = str(rng.choice(["damped_sine", "damped_sine_noise", "bandlimited"]))
        hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, A=float(10 ** rng.uniform(0.5, 2)))
        audio = synthetic_audio(inp, B, T, hp.delta_t, 400 + it)
        m = PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D, rank1=int(rng.choice([1, 2, 3, 4]))))
        assert m._get_backend().variant == 5
        m._get_backend().set_wide_chain(int(rng.choice([0, 1, 1, 2])))
        m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
        per = m.loss_per_clip(); flat, _ = m.grad_sums(); g = unpack_grad(flat.cpu().numpy(), D)
        ref = c_oracle_run(m, audio, "f32"); gr = C.unpack_grad(ref["grad"], D)
        g64 = C.unpack_grad(c_oracle_run(m, audio, "f64")["grad"], D)
        cfg = (D, T, B, round(sigma, 5), round(rs, 3), inp, m._get_backend().effective_rank1, m._get_backend().wide_chain)
        assert np.all(np.isfinite(per)) and np.all(np.isfinite(flat.cpu().numpy())), ("wide non-finite", cfg)
        bf16x2 = m._get_backend().effective_rank1 == 1       # two bf16 pieces carry 16 operand bits (test_gpu_wide.py's bar for that mode)
        out.append(("wide", "loss", _loss_err(per, ref["loss_per_clip"]), LOSS_BAR, cfg))
        out.append(("wide", "grad", max(rel_inf(g[k], gr[k]) for k in ("Rbar", "fbar", "psi0bar")), GRAD_BAR * (2 if bf16x2 else 1), cfg))
        out.append(("wide", "dA", rel_inf(g["Abar"], gr["Abar"]), _dA_bar(gr["Abar"], g64["Abar"], GRAD_BAR), cfg))


def sweep_step(rng, n, out):
    """Device-resident optimiser step (cmps_psi_apply_step) vs the host one, 10 steps (train.py:55-60, 88-89)."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan
    from audio_mps_amd.train import Trainer
    for it in range(n):
        D = int(rng.choice([3, 8, 16, 24, 32, 48])); T = int(rng.integers(20, 300)); B = int(rng.integers(1, 9))
        hp = HParams(minibatch_size=B, bond_dim=D, learning_rate=float(10 ** rng.uniform(-3, -1.7)))
        audio = make_audio(B, T, hp.delta_t, 500 + it)
        ms = [PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D)) for _ in range(2)]
        if D > 32:
            for mm in ms:
                mm.variables["Rx"] *= np.float32(0.5); mm.variables["Ry"] *= np.float32(0.5)
        td, th = Trainer(ms[0], hp, device_step=True), Trainer(ms[1], hp)
        dev = np.array([td.step()["total_loss"] for _ in range(10)]); host = np.array([th.step()["total_loss"] for _ in range(10)])
        out.append(("step", "trajectory", float(np.max(np.abs(dev - host) / np.maximum(np.abs(host), 1))), 1e-6, (D, T, B, round(hp.learning_rate, 4))))


def sweep_pair(rng, n, out):
    """bf16 pair kernels vs the bf16-emulating oracle; odd draws make Q visible in float32 (sigma up to 0.5, large R)."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan, unpack_grad
    for it in range(n):
        D = int(rng.integers(33, 129)); T = int(rng.integers(2, 400)); B = int(rng.integers(1, 7))
        sigma = float(10 ** rng.uniform(-4, -0.3)) if it % 2 else 1e-4
        hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma)
        audio = make_audio(B, T, hp.delta_t, 100 + it)
        m = PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D, variant=3))
        if it % 2:
            rs = float(10 ** rng.uniform(-1.0, -0.2))
            m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
        per = m.loss_per_clip(); flat, _ = m.grad_sums(); g = unpack_grad(flat.cpu().numpy(), D)
        em = O.psi_bf16_scan(oracle_hparams(hp), oracle_variables(m), audio)
        cfg = (D, T, B, round(sigma, 5))
        if not np.all(np.isfinite(em["loss_per_clip"])):
            continue
        assert np.all(np.isfinite(per)) and np.all(np.isfinite(flat.cpu().numpy())), ("pair non-finite", cfg)
        out.append(("pair", "loss", _loss_err(per, em["loss_per_clip"]), PAIR_LOSS_BAR, cfg))
        out.append(("pair", "grad", max(rel_inf(g[k], em[k]) for k in ("Rbar", "fbar", "psi0bar")), PAIR_GRAD_BAR, cfg))
        out.append(("pair", "dA", rel_inf(g["Abar"], em["Abar"]), 10 * PAIR_GRAD_BAR, cfg))


def sweep_rho(rng, n, out):
    """RhoCMPS (model.py:55-203): column kernels for rank <= 8, GEMM kernels above, general kernels for D > 32; vs the float64 matrix-form oracle."""
    from audio_mps_amd import HParams, RhoCMPS
    for it in range(n):
        D = int(rng.integers(2, 41)); r = int(rng.integers(1, min(D, 32) + 1)); T = int(rng.integers(2, 150)); B = int(rng.integers(1, 6))
        rs, amp = 0.4, 1.0
        kind = it % 12 if n < 34 else (3 if it in (3, 7, 11) else 11 if it >= 31 else 5 if it >= 14 else 0)
        if kind == 3:                                      # the general kernels' column groups: 32 < D <= 128, ragged ranks
            D = int(rng.integers(33, 129)); r = int(rng.integers(1, min(D, 40) + 1)); T = int(rng.integers(2, 40)); B = int(rng.integers(1, 4))
        if kind == 11:                                     # the reference's default rank = D above the old LDS limit (workspace columns)
            D = int(rng.integers(72, 129)); r = D; T = int(rng.integers(2, 12)); B = int(rng.integers(1, 3))
        elif kind >= 5:                                    # the GEMM kernels: rank > 8 at D <= 32, the fp16 x 2 forward's scales --
            D = int(rng.integers(9, 33)); r = int(rng.integers(9, D + 1)); T = int(rng.integers(2, 400))      # loud / quiet clips, large / tiny R
            rs = float(10 ** rng.uniform(-2.5, 0.2)); amp = float(10 ** rng.uniform(-3, 0.7))
        hp = HParams(minibatch_size=B, bond_dim=D, initial_rank=r, sigma=float(10 ** rng.uniform(-4, -0.3)))
        audio = (make_audio(B, T, hp.delta_t, 200 + it) * np.float32(amp)).astype(np.float32)
        if kind >= 5 and it % 4 == 0 and T > 30:
            audio[:, T // 2:] = audio[:, T // 2: T // 2 + 1]                # a silent tail (increments exactly zero)
        m = RhoCMPS(hp, data_iterator=audio, seed=it)
        m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
        ov = O.Variables(np.asarray(m.variables["A"], np.float32), m.variables["Rx"], m.variables["Ry"], m.variables["freqs"],
                         np.zeros(D, np.float32), np.zeros(D, np.float32), scaled_R=True, scaled_freqs=True)
        ref = O.rho_loss_and_grads(O.HParams(**hp.values()), ov.astype(np.float64), m.variables["Wx"].astype(np.float64),
                                   m.variables["Wy"].astype(np.float64), audio, "f64")
        if not np.all(np.isfinite(ref["per_clip"])):
            continue                                       # (1 + z <= 0 somewhere: the model itself diverges on this draw)
        per = m.loss_per_clip(); loss, grads = m.loss_and_grads()
        cfg = (D, r, T, B, round(rs, 4), round(amp, 4), round(hp.sigma, 5))
        assert np.all(np.isfinite(per)) and all(np.all(np.isfinite(grads[k])) for k in grads), ("rho non-finite", cfg)
        out.append(("rho", "loss", _loss_err(per, ref["per_clip"]), LOSS_BAR, cfg))
        ref32 = O.rho_loss_and_grads(O.HParams(**hp.values()), ov, m.variables["Wx"], m.variables["Wy"], audio, "f32")
        # per tensor against max(bar, 2 x the float32 oracle's own distance from float64): at tiny R / long clips the frequency gradient of
        # ANY float32 evaluation sits 1 - 2e-4 from float64 (seeds 31, 32: 1.1e-4 ... 2.0e-4 for the oracle and for every kernel setting
        # alike, scripts/dev_rho_sweep_cases.py); the record is the tensor closest to (or furthest above) its bar
        worst = max(((rel_inf(grads[k], ref[k]), max(GRAD_BAR, 2.0 * rel_inf(ref32[k], ref[k]))) for k in ("Rx", "Ry", "freqs", "Wx", "Wy")),
                    key=lambda eb: eb[0] / eb[1])
        out.append(("rho", "grad", worst[0], worst[1], cfg))
        out.append(("rho", "dA", rel_inf(grads["A"], ref["A"]), _dA_bar(ref32["A"], ref["A"], GRAD_BAR), cfg))


def sweep_legacy(rng, n, out):
    """Legacy AudioMPS arithmetic (training_estimators.py:43-45): wave kernels in legacy mode / general kernels above D = 32."""
    from audio_mps_amd import LegacyAudioMPS
    for it in range(n):
        D = int(rng.integers(2, 41)); T = int(rng.integers(2, 300)); B = int(rng.integers(1, 7)); dt = float(10 ** rng.uniform(-3, -2))
        audio = make_audio(B, T, dt, 300 + it, noise=0.05)
        m = LegacyAudioMPS(D, dt, B, data_iterator=audio, seed=it)
        ref = O.legacy_loss_and_grads(m.variables["H"], m.variables["R"], dt, audio, "f32")
        per = m.loss_per_clip(); loss, grads = m.loss_and_grads()
        out.append(("legacy", "loss", _loss_err(per, ref["per_clip"]), LOSS_BAR, (D, T, B)))
        out.append(("legacy", "grad", max(rel_inf(grads["R"], ref["gR"]), rel_inf(grads["H"], ref["gH"])), GRAD_BAR, (D, T, B)))


FAMILIES = {"psi": sweep_psi, "wide": sweep_wide, "step": sweep_step, "pair": sweep_pair, "rho": sweep_rho, "legacy": sweep_legacy}


def run_sweep(seed, counts=None, families=None):
    """All families with one generator (the order is fixed, so a seed names its draws).  Returns the list of records."""
    rng = np.random.default_rng(seed)
    counts = dict(DEFAULT_COUNTS, **(counts or {}))
    out = []
    for fam, fn in FAMILIES.items():
        if families is None or fam in families:
            fn(rng, counts[fam], out)
    return out


def worst_by_kind(records):
    worst = {}
    for fam, what, err, bar, cfg in records:
        key = f"{fam} {what}"
        if key not in worst or err / bar > worst[key][0] / worst[key][1]:
            worst[key] = (err, bar, cfg)
    return worst
