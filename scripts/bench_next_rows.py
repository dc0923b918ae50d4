#!/usr/bin/env python3
"""Measurement of the SURVEY 8(f) "next" rows on one MI355X (not the headline metric: bench.py stays on the PsiCMPS scan).

For every row: device time of one call at a stated shape (HIP events around the launches, median of a few rounds, inputs
resident in HBM), the throughput in audio samples/s, and the same arithmetic on the host CPU (the numpy oracle, one
process, a bounded sample) for scale.  Prints one JSON object; scripts/.. -> profiles/r4_next_rows.json.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from audio_mps_amd import HParams, PsiCMPS, RhoCMPS, LegacyAudioMPS
from audio_mps_amd import tfrecord
from oracle import cmps_oracle as O
from _util import make_audio


def dev_ms(fn, rounds=3):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        ev0.record()
        fn()
        ev1.record()
        torch.cuda.synchronize()
        out.append(ev0.elapsed_time(ev1))
    return float(np.median(out))


res = {}

# ---- rank 1: PsiCMPS.sample (model.py:242-251): 1024 paths x 16000 steps, D = 32 (wave-per-path kernel)
hp = HParams(minibatch_size=1024, bond_dim=32, sigma=0.05)
m = PsiCMPS(hp, seed=0)
n, length = 1024, 16000
noise = (0.05 * np.sqrt(hp.delta_t) * np.random.default_rng(0).standard_normal((length, n))).astype(np.float32)
be = m._get_backend()
be.set_params(m.effective_params(), n, length + 1, train=False)
d_noise = torch.from_numpy(np.ascontiguousarray(noise.T)).to(be.device)
d_out = torch.empty((n, length), dtype=torch.float32, device=be.device)
from audio_mps_amd import _capi
ms = dev_ms(lambda: _capi.check(be._h, be._lib.cmps_psi_sample(be._h, d_noise.data_ptr(), n, length, d_out.data_ptr(), be._stream())))
t0 = time.perf_counter()
O.psi_sample(O.HParams(**hp.values()), O.Variables(np.asarray(m.variables["A"]), m.variables["Rx"], m.variables["Ry"], m.variables["freqs"],
                                                m.variables["psi_x"], m.variables["psi_y"], scaled_R=True, scaled_freqs=True), noise[:2000, :16])
cpu = 16 * 2000 / (time.perf_counter() - t0)
res["sample_psi"] = {"shape": "D=32, 1024 paths x 16000 steps", "kernel": "k_sample_wave", "ms": ms, "samples_per_s": n * length / ms * 1e3,
                     "cpu_numpy_samples_per_s": cpu, "bound": "serial chain: two wave reductions per step (latency)"}

# ---- rank 1 above D = 32 (round 4): the wide chain's sampling mode against the block sampler, D = 128, 512 paths x 4000 steps
from audio_mps_amd.scan import HipScan
n, length = 512, 4000
hp = HParams(minibatch_size=n, bond_dim=128, sigma=0.05)
noise = (0.05 * np.sqrt(hp.delta_t) * np.random.default_rng(1).standard_normal((length, n))).astype(np.float32)
d_noise = torch.from_numpy(np.ascontiguousarray(noise.T)).cuda()
d_out = torch.empty((n, length), dtype=torch.float32, device="cuda")
row = {}
for name, variant in (("k_sample_wide", 5), ("k_sample_block", 1)):
    mw = PsiCMPS(hp, seed=0, backend=HipScan(128, variant=variant))
    mw.variables["Rx"] *= np.float32(0.35)
    mw.variables["Ry"] *= np.float32(0.35)
    bw = mw._get_backend()
    bw.set_params(mw.effective_params(), n, length + 1, train=False)
    row[name] = dev_ms(lambda: _capi.check(bw._h, bw._lib.cmps_psi_sample(bw._h, d_noise.data_ptr(), n, length, d_out.data_ptr(), bw._stream())))
    row[name + "_out"] = d_out.cpu().numpy().copy()
dev = float(np.max(np.abs(row["k_sample_wide_out"] - row["k_sample_block_out"])) / max(1.0, float(np.max(np.abs(row["k_sample_block_out"])))))
res["sample_psi_d128"] = {"shape": "D=128, 512 paths x 4000 steps", "kernel": "k_sample_wide (one workgroup per pair of paths; the forward chain's layout, R ut and Q ut unmerged, two LDS barriers per step)",
                          "ms": row["k_sample_wide"], "us_per_step": row["k_sample_wide"] * 1e3 / length,
                          "samples_per_s": n * length / row["k_sample_wide"] * 1e3,
                          "block_sampler_ms": row["k_sample_block"], "max_dev_from_block_sampler": dev,
                          "bound": "serial chain: 256 v_pk_fma_f32 per wave and step + two cross-wave exchanges (LDS latency)"}

# ---- rank 2: legacy AudioMPS forward + backward (block kernels), D = 32, T = 4000, 1024 clips
T, B = 4000, 1024
audio = torch.from_numpy(make_audio(B, T, 1e-3, 1)).cuda()
lm = LegacyAudioMPS(32, 1e-3, B, seed=1)
lb = lm._get_backend()
lb.legacy_set_params(lm.variables["R"], lm.Q, lm.delta_t, B, T, train=True)
ms = dev_ms(lambda: (lb.legacy_forward(audio, save_for_bwd=True), lb.legacy_backward()))
t0 = time.perf_counter()
O.legacy_loss_and_grads(lm.variables["H"], lm.variables["R"], lm.delta_t, audio[:8, :1000].cpu().numpy())
cpu = 8 * 1000 / (time.perf_counter() - t0)
res["legacy_audiomps"] = {"shape": f"D=32, T={T}, B={B}, fwd+bwd", "kernel": "k_fwd_wave2<LEGACY> + k_bwd_wave<LEGACY> (the pure-state wave kernels in legacy mode)", "ms": ms,
                          "samples_per_s": B * T / ms * 1e3, "cpu_numpy_samples_per_s": cpu, "bound": "instruction issue of one wave per SIMD, as the headline kernels (the pure-state wave kernels in legacy mode: merged mat-vec chain, loss wave and rank-1 sums on the matrix cores)"}

# ---- rank 3: RhoCMPS forward + backward (rank-r column kernels), D = 32, rank 4 and 32, T = 1000, 256 clips
for rank in (4, 32):
    T, B = 1000, 256
    hp = HParams(minibatch_size=B, bond_dim=32, initial_rank=rank)
    a = make_audio(B, T, hp.delta_t, 2)
    rm = RhoCMPS(hp, data_iterator=a, seed=2)
    d_a = rm._to_device(a)
    rb = rm._prepare(B, T, train=True)
    ms = dev_ms(lambda: rb.rho_loss_and_grad_sums(d_a), rounds=2)
    t0 = time.perf_counter()
    O.rho_loss_and_grads(O.HParams(**hp.values()), O.Variables(np.asarray(rm.variables["A"]), rm.variables["Rx"], rm.variables["Ry"],
                                                              rm.variables["freqs"], np.zeros(32, np.float32), np.zeros(32, np.float32),
                                                              scaled_R=True, scaled_freqs=True), rm.variables["Wx"], rm.variables["Wy"], a[:4, :200])
    cpu = 4 * 200 / (time.perf_counter() - t0)
    res[f"rho_cmps_rank{rank}"] = {"shape": f"D=32, rank={rank}, T={T}, B={B}, fwd+bwd", "kernel": ("k_fwd_rho_mfma<f16x2> (wave per clip, row-array GEMMs, fp16 x 2 operands) + round 5: k_bwd_wave on one virtual clip per column (CMPS_OPT_RHO_BWD)" if rank > 8 else
                                              "k_fwd_rho_wave + k_bwd_rho_wave (wave per clip, column by column)"), "ms": ms,
                                   "samples_per_s": B * T / ms * 1e3, "cpu_numpy_samples_per_s": cpu,
                                   "bound": ("forward: 256 waves on 1024 SIMDs, dependent trip registers -> LDS -> MFMA per step, the same cost at every rank <= 32; reverse: the pure-state wave scan on B x rank virtual clips, linear in the rank"
                                             if rank > 8 else
                                             "straight-line wave-per-clip kernels, 3 r D^2 complex MACs + 6 r exact fp32 MFMAs per step (issue / LDS latency)")}

# ---- the same two rows above D = 32, where they still run on the general one-workgroup-per-clip kernels (not built)
T, B = 1000, 256
audio64 = torch.from_numpy(make_audio(B, T, 1e-3, 1)).cuda()
lm64 = LegacyAudioMPS(64, 1e-3, B, seed=1)
lb64 = lm64._get_backend()
lb64.legacy_set_params(lm64.variables["R"], lm64.Q, lm64.delta_t, B, T, train=True)
ms = dev_ms(lambda: (lb64.legacy_forward(audio64, save_for_bwd=True), lb64.legacy_backward()), rounds=2)
res["legacy_audiomps_d64"] = {"shape": f"D=64, T={T}, B={B}, fwd+bwd", "kernel": "k_fwd_wide<LEGACY> + k_hy_wide + k_loss_wide<LEGACY> + k_bwd_wide<LEGACY> + k_grad_gemm<LEGACY> (round 5: the wide kernels in legacy mode)", "ms": ms,
                              "samples_per_s": B * T / ms * 1e3, "bound": "serial fp32 VALU chains of one workgroup per pair of clips (128 workgroups at B = 256) + the two GEMMs"}
from audio_mps_amd.scan import HipScan
lm64b = LegacyAudioMPS(64, 1e-3, B, seed=1, backend=HipScan(64, variant=1))
lb64b = lm64b._get_backend()
lb64b.legacy_set_params(lm64b.variables["R"], lm64b.Q, lm64b.delta_t, B, T, train=True)
ms = dev_ms(lambda: (lb64b.legacy_forward(audio64, save_for_bwd=True), lb64b.legacy_backward()), rounds=2)
res["legacy_audiomps_d64_general_kernels"] = {"shape": f"D=64, T={T}, B={B}, fwd+bwd", "kernel": "k_fwd_legacy + k_bwd_legacy (CMPS_VARIANT_BLOCK: one workgroup per clip; what D > 32 ran on until round 4)", "ms": ms,
                                              "samples_per_s": B * T / ms * 1e3}
hp = HParams(minibatch_size=B, bond_dim=64, initial_rank=16)
a = make_audio(B, T, hp.delta_t, 2)
rm = RhoCMPS(hp, data_iterator=a, seed=2)
rm.variables["Rx"] *= np.float32(0.5); rm.variables["Ry"] *= np.float32(0.5)
d_a = rm._to_device(a)
rb = rm._prepare(B, T, train=True)
ms = dev_ms(lambda: rb.rho_loss_and_grad_sums(d_a), rounds=2)
rb.kernel_events(True)
rb.rho_loss_and_grad_sums(d_a); torch.cuda.synchronize(); rb.kernel_times()
rb.rho_loss_and_grad_sums(d_a); torch.cuda.synchronize()
kt = {k: round(v[0] / max(v[1], 1), 3) for k, v in rb.kernel_times().items()}
rb.kernel_events(False)
res["rho_cmps_d64_rank16"] = {"shape": f"D=64, rank=16, T={T}, B={B}, fwd+bwd", "kernel": "round 5: the columns as virtual clips of the wide kernels (k_fwd_wide_rho + k_hy_wide + k_bwd_wide + k_grad_gemm on 4096 virtual clips)", "ms": ms,
                              "samples_per_s": B * T / ms * 1e3, "kernel_ms": kt, "bound": "forward chain: one workgroup per clip looping over 8 column pairs per step; reverse chain and GEMMs: 2048 independent virtual pairs"}
for rk in (4, 32):
    hp = HParams(minibatch_size=B, bond_dim=64, initial_rank=rk)
    rm2 = RhoCMPS(hp, data_iterator=a, seed=2)
    rm2.variables["Rx"] *= np.float32(0.5); rm2.variables["Ry"] *= np.float32(0.5)
    rb2 = rm2._prepare(B, T, train=True)
    ms2 = dev_ms(lambda: rb2.rho_loss_and_grad_sums(d_a), rounds=2)
    res[f"rho_cmps_d64_rank{rk}"] = {"shape": f"D=64, rank={rk}, T={T}, B={B}, fwd+bwd", "ms": ms2, "samples_per_s": B * T / ms2 * 1e3}
rmb = RhoCMPS(HParams(minibatch_size=B, bond_dim=64, initial_rank=16), data_iterator=a, seed=2, backend=HipScan(64, variant=1))
rmb.variables["Rx"] *= np.float32(0.5); rmb.variables["Ry"] *= np.float32(0.5)
rbb = rmb._prepare(B, T, train=True)
ms = dev_ms(lambda: rbb.rho_loss_and_grad_sums(d_a), rounds=1)
res["rho_cmps_d64_rank16_general_kernels"] = {"shape": f"D=64, rank=16, T={T}, B={B}, fwd+bwd", "kernel": "k_fwd_rho + k_bwd_rho (CMPS_VARIANT_BLOCK: what D > 32 ran on until round 4)", "ms": ms,
                                              "samples_per_s": B * T / ms * 1e3}

# ---- rank 3b: RhoCMPS.sample (row-array GEMM sampler, one wavefront per path), D = 32, rank 32 and 4: 64 paths x 4000 steps
for rank in (32, 4):
    hp = HParams(minibatch_size=64, bond_dim=32, initial_rank=rank, sigma=0.05)
    rm = RhoCMPS(hp, seed=2)
    n, length = 64, 4000
    noise = (0.05 * np.sqrt(hp.delta_t) * np.random.default_rng(0).standard_normal((length, n))).astype(np.float32)
    rm.sample(n, length, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rm.sample(n, length, noise=noise)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3                      # host-timed (includes the upload of the noise and the download of the waveform)
    res[f"sample_rho_rank{rank}"] = {"shape": f"D=32, rank={rank}, {n} paths x {length} steps", "kernel": "k_sample_rho_mfma<f16x2> (wave per path, two row-array GEMMs per step, fp16 x 2 operands with fixed scales)",
                                     "ms": ms, "samples_per_s": n * length / ms * 1e3, "us_per_step": ms * 1e3 / length,
                                     "bound": "serial: 96 32x32x16 bf16 MFMAs per step on one SIMD's matrix pipe (~1.5 us) + the operand splits; "
                                              "the workgroup-per-path kernel (CMPS_VARIANT_BLOCK) takes 93 us per step at rank 32, 12.7 us at rank 4"}

# ---- rank 4: TFRecord reader (host, pure Python): records of 65536 float32 samples
import tempfile
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "x.tfrecords")
    rng = np.random.default_rng(0)
    recs = [rng.standard_normal(65536).astype(np.float32) for _ in range(32)]
    tfrecord.write_audio_tfrecord(path, recs)
    for verify in (False, True):
        t0 = time.perf_counter()
        cnt = sum(1 for _ in tfrecord._audio_records(path, 65536, verify))
        dt = time.perf_counter() - t0
        res["tfrecord_reader" + ("_verify" if verify else "")] = {
            "shape": "32 records x 65536 float32", "records": cnt, "ms": dt * 1e3, "samples_per_s": cnt * 65536 / dt,
            "MB_per_s": cnt * 65536 * 4 / dt / 1e6,
            "bound": "host: Python framing + protobuf wire decode (numpy frombuffer)" + (" + CRC-32C via cmps_crc32c (SSE4.2)" if verify else "")}
print(json.dumps(res, indent=1))
