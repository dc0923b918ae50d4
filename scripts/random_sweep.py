#!/usr/bin/env python3
"""One-off robustness sweep on the GPU: random shapes and parameter scales through every kernel family against the oracles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from audio_mps_amd import HParams, PsiCMPS, RhoCMPS, LegacyAudioMPS
from audio_mps_amd.scan import HipScan, unpack_grad
from oracle import cmps_oracle as O, c_oracle as C
from _util import make_audio, c_oracle_run, rel_inf, oracle_hparams, oracle_variables

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {}
def note(kind, val, cfg):
    if val > worst.get(kind, (0, None))[0]:
        worst[kind] = (val, cfg)

for it in range(60):                                   # pure-state wave / block kernels vs the C restatement
    D = int(rng.integers(1, 33)); T = int(rng.integers(2, 700)); B = int(rng.integers(1, 14))
    sigma = float(10 ** rng.uniform(-4, 0)); rs = float(10 ** rng.uniform(-1.5, 0)); variant = int(rng.choice([1, 2]))
    hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, A=float(10 ** rng.uniform(0, 2)))
    if it % 4 == 1:                                    # other sampling rates: 3 kHz ... 100 kHz (rotation table, dt_k, the scale of Q)
        hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, A=hp.A, delta_t=float(10 ** rng.uniform(-5, -3.5)))
    amp = float(10 ** rng.uniform(-3, 0.3))           # (round 4: amplitudes over three decades, silent stretches: the fp16 scales)
    audio = (make_audio(B, T, hp.delta_t, it) * np.float32(amp)).astype(np.float32)
    if T > 40 and it % 3 == 0:
        audio[:, : T // 3] = 0.0
    m = PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D, variant=variant, rank1=int(rng.choice([2, 3, 4]))))
    m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
    per = m.loss_per_clip(); flat, _ = m.grad_sums(); g = unpack_grad(flat.cpu().numpy(), D)
    ref = c_oracle_run(m, audio, "f32"); gr = C.unpack_grad(ref["grad"], D)
    if not np.all(np.isfinite(ref["loss_per_clip"])):
        continue
    cfg = (D, T, B, round(sigma, 5), round(rs, 3), variant, round(amp, 4), m._get_backend().effective_rank1, float(f"{hp.delta_t:.3g}"))
    note("psi loss", float(np.max(np.abs(per - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1))), cfg)
    note("psi grad", max(rel_inf(g[k], gr[k]) for k in ("Rbar", "fbar", "psi0bar")), cfg)
    note("psi dA", rel_inf(g["Abar"], gr["Abar"]), cfg)   # one scalar, a cancelling sum: the float32 restatement itself sits up to 1e-2 from float64
for it in range(16):                                   # wide kernels (float32, 32 < D <= 128; AUTO) vs the C restatement
    D = int(rng.integers(33, 129)); T = int(rng.integers(2, 500)); B = int(rng.integers(1, 8))
    sigma = float(10 ** rng.uniform(-4, 0)); rs = float(10 ** rng.uniform(-1.5, -0.3)); inp = str(rng.choice(["damped_sine", "damped_sine_noise", "bandlimited"]))
    from audio_mps_amd.data import synthetic_audio
    hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, A=float(10 ** rng.uniform(0.5, 2)))
    audio = synthetic_audio(inp, B, T, hp.delta_t, 400 + it)
    m = PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D, rank1=int(rng.choice([1, 2, 3, 4]))))
    assert m._get_backend().variant == 5
    m._get_backend().set_wide_chain(int(rng.choice([0, 1, 1, 2])))
    m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
    per = m.loss_per_clip(); flat, _ = m.grad_sums(); g = unpack_grad(flat.cpu().numpy(), D)
    ref = c_oracle_run(m, audio, "f32"); gr = C.unpack_grad(ref["grad"], D)
    cfg = (D, T, B, round(sigma, 5), round(rs, 3), inp, m._get_backend().effective_rank1, m._get_backend().wide_chain)
    note("wide loss", float(np.max(np.abs(per - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1))), cfg)
    note("wide grad", max(rel_inf(g[k], gr[k]) for k in ("Rbar", "fbar", "psi0bar")), cfg)
    note("wide dA", rel_inf(g["Abar"], gr["Abar"]), cfg)
for it in range(8):                                    # device-resident optimiser step vs the host one, 10 steps
    from audio_mps_amd.train import Trainer
    D = int(rng.choice([3, 8, 16, 24, 32, 48])); T = int(rng.integers(20, 300)); B = int(rng.integers(1, 9))
    hp = HParams(minibatch_size=B, bond_dim=D, learning_rate=float(10 ** rng.uniform(-3, -1.7)))
    audio = make_audio(B, T, hp.delta_t, 500 + it)
    ms = [PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D)) for _ in range(2)]
    if D > 32:
        for mm in ms:
            mm.variables["Rx"] *= np.float32(0.5); mm.variables["Ry"] *= np.float32(0.5)
    td, th = Trainer(ms[0], hp, device_step=True), Trainer(ms[1], hp)
    dev = np.array([td.step()["total_loss"] for _ in range(10)]); host = np.array([th.step()["total_loss"] for _ in range(10)])
    note("device-step trajectory", float(np.max(np.abs(dev - host) / np.maximum(np.abs(host), 1))), (D, T, B, round(hp.learning_rate, 4)))
for it in range(14):                                   # pair kernels vs the bf16 emulation
    D = int(rng.integers(33, 129)); T = int(rng.integers(2, 400)); B = int(rng.integers(1, 7))
    sigma = float(10 ** rng.uniform(-4, -0.3)) if it % 2 else 1e-4      # (odd draws: Q = -(dt sigma^2 / 2) R^dagger R visible in float32)
    hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma)
    audio = make_audio(B, T, hp.delta_t, 100 + it)
    m = PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D, variant=3))
    if it % 2:
        rs = float(10 ** rng.uniform(-1.0, -0.2))
        m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
    per = m.loss_per_clip(); flat, _ = m.grad_sums(); g = unpack_grad(flat.cpu().numpy(), D)
    em = O.psi_bf16_scan(oracle_hparams(hp), oracle_variables(m), audio)
    note("pair loss", float(np.max(np.abs(per - em["loss_per_clip"]) / np.maximum(np.abs(em["loss_per_clip"]), 1))), (D, T, B, round(sigma, 5)))
    note("pair grad", max(rel_inf(g[k], em[k]) for k in ("Rbar", "fbar", "psi0bar")), (D, T, B, round(sigma, 5)))
    note("pair dA", rel_inf(g["Abar"], em["Abar"]), (D, T, B, round(sigma, 5)))
for it in range(34):                                   # RhoCMPS (column kernels for rank <= 8, GEMM kernels above, block kernels for D > 32)
    D = int(rng.integers(2, 41)); r = int(rng.integers(1, min(D, 32) + 1)); T = int(rng.integers(2, 150)); B = int(rng.integers(1, 6))
    rs, amp = 0.4, 1.0
    if it in (3, 7, 11):                               # (round 4) the general kernels' column groups: 32 < D <= 128, ragged ranks
        D = int(rng.integers(33, 129)); r = int(rng.integers(1, min(D, 40) + 1)); T = int(rng.integers(2, 40)); B = int(rng.integers(1, 4))
    if it >= 31:                                       # the reference's default rank = D above the old LDS limit (workspace columns)
        D = int(rng.integers(72, 129)); r = D; T = int(rng.integers(2, 12)); B = int(rng.integers(1, 3))
    elif it >= 14:                                     # (round 4) the GEMM kernels: rank > 8 at D <= 32, the fp16 x 2 forward's scales --
        D = int(rng.integers(9, 33)); r = int(rng.integers(9, D + 1)); T = int(rng.integers(2, 400))      # loud / quiet clips, large / tiny R
        rs = float(10 ** rng.uniform(-2.5, 0.2)); amp = float(10 ** rng.uniform(-3, 0.7))
    hp = HParams(minibatch_size=B, bond_dim=D, initial_rank=r, sigma=float(10 ** rng.uniform(-4, -0.3)))
    audio = (make_audio(B, T, hp.delta_t, 200 + it) * np.float32(amp)).astype(np.float32)
    if it >= 14 and it % 4 == 0 and T > 30:
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
    note("rho loss", float(np.max(np.abs(per - ref["per_clip"]) / np.maximum(np.abs(ref["per_clip"]), 1))), cfg)
    note("rho grad", max(rel_inf(grads[k], ref[k]) for k in ("Rx", "Ry", "freqs", "Wx", "Wy")), cfg)
    note("rho dA", rel_inf(grads["A"], ref["A"]), cfg)   # one scalar, a cancelling sum: the float32 restatement itself sits 1e-5 ... 1e-3 from float64
    if not (np.all(np.isfinite(per)) and all(np.all(np.isfinite(grads[k])) for k in grads)):
        note("rho NONFINITE", 1.0, cfg)
for it in range(10):                                   # legacy AudioMPS
    D = int(rng.integers(2, 41)); T = int(rng.integers(2, 300)); B = int(rng.integers(1, 7)); dt = float(10 ** rng.uniform(-3, -2))
    audio = make_audio(B, T, dt, 300 + it, noise=0.05)
    m = LegacyAudioMPS(D, dt, B, data_iterator=audio, seed=it)
    ref = O.legacy_loss_and_grads(m.variables["H"], m.variables["R"], dt, audio, "f32")
    per = m.loss_per_clip(); loss, grads = m.loss_and_grads()
    note("legacy loss", float(np.max(np.abs(per - ref["per_clip"]) / np.maximum(np.abs(ref["per_clip"]), 1))), (D, T, B))
    note("legacy grad", max(rel_inf(grads["R"], ref["gR"]), rel_inf(grads["H"], ref["gH"])), (D, T, B))
for k, (v, cfg) in worst.items():
    print(f"{k:12s} worst {v:.2e} at {cfg}")
