"""Parity of the HIP path (through the C ABI, include/cmps.h) against the oracle, on a real MI355X.

Tolerances (stated once, used everywhere below):
  * per-clip log-likelihood: |hip - oracle_f32| <= 1e-5 * max(|oracle_f32|, 1)   (BASELINE north_star: 1e-5 relative).
    The floor of 1 covers clips whose loss is a small difference of large partial sums; there two float32
    evaluations that differ only in summation order already differ by ~1e-4 relative.
  * gradients: max |hip - oracle_f32| <= 1e-4 * max |oracle_f32| per tensor (observed ~3e-6).
Both kernel variants are exercised: the wave-per-clip kernels (D <= 32, the hot path) and the block-per-clip
kernels (any D <= 128).
"""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import cmps_oracle as O
from oracle import c_oracle as C
from _util import (c_oracle_run, golden_names, load_golden, make_audio, model_from_golden, oracle_hparams,
                   oracle_variables, rel_inf)

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5
GRAD_RTOL = 1e-4
BLOCK, WAVE, WAVE32 = 1, 2, 4      # WAVE: the 16-row layout for D <= 16, the 32-row layout above; WAVE32: always 32 rows


def _scan(D, variant):
    from audio_mps_amd.scan import HipScan
    return HipScan(D, variant=variant)


def _model(D, T, B, variant, sigma=1e-4, seed=0, rscale=None, **hpkw):
    from audio_mps_amd import HParams, PsiCMPS
    hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, **hpkw)
    audio = make_audio(B, T, hp.delta_t, seed)
    m = PsiCMPS(hp, data_iterator=audio, seed=seed, backend=_scan(D, variant))
    if rscale is not None:
        m.variables["Rx"] *= np.float32(rscale)
        m.variables["Ry"] *= np.float32(rscale)
    return m, audio


def _check_against_oracle(m, audio, nthreads=0, loss_rtol=LOSS_RTOL, grad_rtol=GRAD_RTOL):
    from audio_mps_amd.scan import unpack_grad
    D = m.bond_d
    per = m.loss_per_clip()
    flat, B = m.grad_sums()
    flat = flat.cpu().numpy()
    ref = c_oracle_run(m, audio, "f32", nthreads=nthreads)
    assert np.all(np.isfinite(per))
    err = np.max(np.abs(per - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1.0))
    assert err <= loss_rtol, f"loss rel err {err}"
    g, gr = unpack_grad(flat, D), C.unpack_grad(ref["grad"], D)
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        e = rel_inf(g[k], gr[k])
        assert e <= grad_rtol, f"{k} rel err {e}"
    assert abs(g["loss_sum"] - gr["loss_sum"]) <= loss_rtol * max(abs(gr["loss_sum"]), B)
    return per, flat


# ---------------------------------------------------------------------------------------------------
# golden fixtures and oracle parity over shapes / variants
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [BLOCK, WAVE, WAVE32])
@pytest.mark.parametrize("name", golden_names())
def test_golden(name, variant):
    from audio_mps_amd.scan import unpack_grad
    g = load_golden(name)
    D = int(g["hp_bond_dim"])
    m = model_from_golden(g, backend=_scan(D, variant))
    per = m.loss_per_clip()
    scale = np.maximum(np.abs(g["loss_per_clip_f32"]), 1.0)
    assert np.max(np.abs(per - g["loss_per_clip_f32"]) / scale) <= LOSS_RTOL
    loss, grads = m.loss_and_grads()
    assert abs(float(loss) - float(g["loss_f32"])) <= LOSS_RTOL * max(1.0, abs(float(g["loss_f32"])))
    for k in O.Variables.NAMES:
        assert rel_inf(grads[k], g[f"grad_{k}_f32"]) <= GRAD_RTOL, k
    eff = unpack_grad(m._last, D)
    B = g["data"].shape[0]
    assert rel_inf(eff["Rbar"] / B, g["eff_Rbar_f32"]) <= GRAD_RTOL
    assert rel_inf(eff["fbar"] / B, g["eff_fbar_f32"]) <= GRAD_RTOL


@pytest.mark.parametrize("D,T,B,sigma,variant", [
    (4, 256, 8, 1e-4, WAVE), (4, 256, 8, 1e-4, BLOCK),            # BASELINE configs[0] shape
    (8, 300, 5, 1e-4, WAVE),                                      # train.py default bond_dim, ragged B
    (7, 256, 8, 1.0, WAVE), (7, 256, 8, 1.0, BLOCK),              # odd D, sigma = 1 (dissipator visible)
    (16, 1000, 16, 1e-4, WAVE), (16, 1000, 16, 1e-4, BLOCK),
    (32, 2000, 12, 1e-4, WAVE), (32, 700, 6, 1e-4, BLOCK),
    (32, 500, 4, 1.0, WAVE),
    (33, 200, 3, 1e-4, BLOCK), (64, 300, 4, 1e-4, BLOCK), (128, 100, 2, 1e-4, BLOCK),   # D > 32: block variant
])
def test_oracle_parity(D, T, B, sigma, variant):
    rscale = 0.1 if sigma == 1.0 else None
    m, audio = _model(D, T, B, variant, sigma=sigma, seed=D + T, rscale=rscale)
    _check_against_oracle(m, audio)


@pytest.mark.parametrize("kind", ["damped_sine", "damped_sine_noise", "bandlimited"])
@pytest.mark.parametrize("D,T,B", [(32, 4000, 8), (16, 2048, 8), (64, 1200, 4)])
def test_input_distributions(kind, D, T, B):
    """SURVEY 8(d)'s inputs (bench.py --input): the reference's pure damped sine (data.py:8-22: silent before the onset), the same
    with white noise, and the band-limited random walk 0.1 cumsum(N(0,1)) / sqrt(T), each against the oracle."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.data import synthetic_audio
    hp = HParams(minibatch_size=B, bond_dim=D)
    audio = synthetic_audio(kind, B, T, hp.delta_t, seed=D)
    assert audio.dtype == np.float32 and audio.shape == (B, T) and np.max(np.abs(audio)) <= 1.2
    m = PsiCMPS(hp, data_iterator=audio, seed=1, backend=_scan(D, 0))
    _check_against_oracle(m, audio)


@pytest.mark.parametrize("T", [2, 3, 64, 65, 66, 129, 193])
def test_chunk_boundaries(T):
    """T - 1 steps around the 64-step chunking of the wave kernels (1, 2, 63, 64, 65, 128, 192 steps)."""
    m, audio = _model(32, T, 5, WAVE, seed=T)
    _check_against_oracle(m, audio)


@pytest.mark.parametrize("T", [2, 3, 8, 9, 10, 17, 33, 34, 65, 66, 130, 257])
@pytest.mark.parametrize("D", [3, 16])
def test_wave16_chunk_and_octet_boundaries(D, T):
    """The 16-row layout (cmps_wave16.hip): T - 1 steps around its 32-step forward chunks, its 8-step reverse octets and the
    64-step scalar chunks (1, 2, 7, 8, 9, 16, 32, 33, 64, 65, 129, 256 steps), full and partly used rows."""
    m, audio = _model(D, T, 3, WAVE, seed=T + D)
    _check_against_oracle(m, audio)


def test_wave16_matches_wave32():
    """D <= 16 in both lane layouts: same arithmetic up to float32 summation order."""
    m1, audio = _model(16, 1500, 7, WAVE, seed=8)
    m2, _ = _model(16, 1500, 7, WAVE32, seed=8)
    assert m1._get_backend().variant == WAVE and m2._get_backend().variant == WAVE32
    p1, p2 = m1.loss_per_clip(), m2.loss_per_clip()
    assert np.max(np.abs(p1 - p2) / np.maximum(np.abs(p2), 1.0)) <= LOSS_RTOL
    f1 = m1.grad_sums()[0].cpu().numpy()
    f2 = m2.grad_sums()[0].cpu().numpy()
    assert rel_inf(f1[:2 * 16 * 16], f2[:2 * 16 * 16]) <= GRAD_RTOL
    assert rel_inf(f1[2 * 16 * 16:], f2[2 * 16 * 16:]) <= GRAD_RTOL
    s1, s2 = m1.psi_evolve_with_data(), m2.psi_evolve_with_data()
    assert np.max(np.abs(s1 - s2)) < 1e-5


def test_single_clip_and_non_multiple_of_four():
    for B in (1, 2, 3, 7):
        m, audio = _model(16, 130, B, WAVE, seed=B)
        _check_against_oracle(m, audio)


def test_config2_full_size():
    """BASELINE configs[1]: D=16, T=4096, batch=256, full size against the C oracle."""
    m, audio = _model(16, 4096, 256, WAVE, seed=2)
    _check_against_oracle(m, audio, nthreads=16)


def test_config3_reduced_batch():
    """BASELINE configs[2] at full T = 16000, D = 32, on a reduced batch the oracle finishes in seconds."""
    m, audio = _model(32, 16000, 32, WAVE, seed=3)
    _check_against_oracle(m, audio, nthreads=16)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_rank1_modes(mode):
    """cmps_set_option(CMPS_OPT_RANK1): exact fp32 MFMA, the bf16x2 and bf16x3 splits and the scaled fp16x2 split of the rank-1 gradient
    updates (k_bwd_wave<0|1|2|3>) all stay inside the gradient bar; a new handle's DEFAULT means F16X2 here (include/cmps.h's table; the
    forward's loss product follows the same option: two fp16 pieces for F16X2 / DEFAULT, three bf16 pieces otherwise).
    T = 3000 spans many aligned octets plus an unaligned top (2999 steps) and a chunk boundary."""
    from audio_mps_amd import _capi
    m, audio = _model(32, 3000, 10, WAVE, seed=17)
    be = m._get_backend()
    assert be.rank1 == _capi.CMPS_RANK1_DEFAULT and be.effective_rank1 == _capi.CMPS_RANK1_F16X2
    be.set_rank1(mode)
    assert be.rank1 == mode
    _check_against_oracle(m, audio, nthreads=10)
    with pytest.raises(_capi.CmpsError):
        be.set_rank1(7)


def test_rank1_modes_order_of_accuracy():
    """Distance of the three modes from each other on the rank-1 sums: x3 must sit closer to the exact-fp32 MFMA than x2
    does (all three share every other instruction, so the difference isolates the product arithmetic)."""
    m, audio = _model(32, 2000, 8, WAVE, seed=23)
    be = m._get_backend()
    flats = {}
    for mode in (0, 1, 2, 3):
        be.set_rank1(mode)
        flats[mode] = m.grad_sums()[0].cpu().numpy().astype(np.float64)[:2 * 32 * 32]
    e2 = rel_inf(flats[1], flats[0])
    e3 = rel_inf(flats[2], flats[0])
    e16 = rel_inf(flats[3], flats[0])
    print(f"rank-1 sums vs exact fp32 MFMA: bf16x2 {e2:.2e}, bf16x3 {e3:.2e}, f16x2 {e16:.2e}")
    assert e3 <= 1e-5, e3          # fp32 summation-order noise only (the MFMA adds 16 products per instruction)
    assert e2 <= 5e-5, e2
    assert e3 <= e2
    assert e16 <= 1e-5 and e16 <= 4 * e3 + 2e-6, (e16, e3)       # the fp16 split (round 4): bf16x3's class


@pytest.mark.parametrize("D,T", [(18, 60), (32, 60), (32, 501), (9, 47)])
def test_qbar_sums_visible_at_large_sigma(D, T):
    """With train.py's sigma = 1e-4 the term Q = -(dt sigma^2 / 2) R^dagger R is below float32 resolution and errors in Qbar = sum ybar u^dagger
    never reach the R gradient.  sigma = 0.36 with a large R makes them visible: every arithmetic of the rank-1 sums must then sit at
    the float32 oracle's own distance from float64.  (Round 4: hipcc put a v_pk_fma_f32 one instruction in front of the v_mfma_f32_32x32x2_f32
    that reads its result in the steps updated immediately -- every step of EXACT_F32, the steps above the first aligned octet otherwise --
    and the matrix core used the previous step's u_k: Rbar off by 3e-4 ... 3e-3 here, unnoticed at small sigma.  DESIGN 4.3e.)"""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan, unpack_grad
    hp = HParams(minibatch_size=5, bond_dim=D, sigma=0.36, A=66.0)
    audio = (make_audio(5, T, hp.delta_t, 3) * np.float32(0.09)).astype(np.float32)
    g64 = own = None
    for variant in (WAVE, 4):                               # 16-row layout below D = 17 / 32-row layout; the 32-row layout for every D
        for mode in (0, 1, 2, 3):
            m = PsiCMPS(hp, data_iterator=audio, seed=7, backend=HipScan(D, variant=variant, rank1=mode))
            m.variables["Rx"] *= np.float32(0.69)
            m.variables["Ry"] *= np.float32(0.69)
            g = unpack_grad(m.grad_sums()[0].cpu().numpy(), D)
            if g64 is None:
                g64 = C.unpack_grad(c_oracle_run(m, audio, "f64")["grad"], D)
                own = {k: rel_inf(C.unpack_grad(c_oracle_run(m, audio, "f32")["grad"], D)[k], g64[k]) for k in ("Rbar", "fbar", "psi0bar")}
            for k in ("Rbar", "fbar", "psi0bar"):
                bar = 3 * own[k] + (3e-5 if mode == 1 else 3e-6)         # BF16X2 carries 16 operand bits
                assert rel_inf(g[k], g64[k]) <= bar, (variant, mode, k, rel_inf(g[k], g64[k]), own[k])


@pytest.mark.parametrize("D", [17, 24, 32])
def test_rank1_f16x2_scale_jumps(D):
    """The fp16 split of the 32-row reverse scan takes its power-of-two scales per 64-step chunk from a bound of |ybar|: audio with
    silence -> signal, a 1e4 x amplitude step, isolated clicks and tiny noise (ybar moves by orders of magnitude inside and between
    chunks; a wrong bound would overflow the fp16 pieces: inf / NaN), every clip different, against the oracle; and against BF16X3."""
    from audio_mps_amd.scan import unpack_grad
    rng = np.random.default_rng(D)
    T = 777
    m, audio = _model(D, T, 5, WAVE, seed=D)
    audio = audio.copy()
    audio[0, :300] = 0.0
    audio[1, 400:] *= np.float32(1e-4)
    audio[2] = 0.0
    audio[2, 100] = 0.3; audio[2, 101] = -0.2; audio[2, 640] = 0.25
    audio[3] = (1e-3 * rng.standard_normal(T)).astype(np.float32)
    audio[3, 500:] *= np.float32(100.0)
    audio[4] *= np.float32(1e-3)
    be = m._get_backend()
    be.set_rank1(3)
    ref = c_oracle_run(m, audio, "f32")
    assert np.all(np.isfinite(ref["loss_per_clip"]))
    flat = m.grad_sums(audio)[0].cpu().numpy()
    assert np.all(np.isfinite(flat))
    g, gr = unpack_grad(flat, D), C.unpack_grad(ref["grad"], D)
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        assert rel_inf(g[k], gr[k]) <= GRAD_RTOL, k
    be.set_rank1(2)
    g3 = unpack_grad(m.grad_sums(audio)[0].cpu().numpy(), D)
    assert rel_inf(g["Rbar"], g3["Rbar"]) <= 2e-5


@pytest.mark.parametrize("D,T,B,sigma,rscale", [
    (32, 2, 3, 1e-4, None), (32, 3, 2, 1e-4, None), (32, 8, 5, 1e-4, None), (32, 9, 4, 1e-4, None), (32, 10, 4, 1e-4, None),
    (32, 16, 4, 1e-4, None), (32, 17, 1, 1e-4, None), (32, 33, 6, 1e-4, None), (32, 34, 5, 1e-4, None), (32, 65, 5, 1e-4, None),
    (32, 66, 7, 1e-4, None), (32, 129, 3, 1e-4, None), (32, 130, 9, 1e-4, None), (32, 700, 6, 1e-4, None), (24, 257, 5, 1e-4, None),
    (17, 131, 4, 1e-4, None), (32, 501, 5, 0.36, 0.69), (18, 60, 5, 0.36, 0.69), (32, 2000, 12, 1e-4, None), (32, 500, 4, 1.0, 0.1)])
def test_two_wave_reverse_scan_matches_oracle_and_one_wave(D, T, B, sigma, rscale):
    """CMPS_OPT_BWD_WAVES = 2 (round 5, cmps_wave_bwd2.hip; the default): the reverse scan as a chain wave + a gradient wave per clip.  T - 1 around
    every boundary of the hand-over (one step, a lone top octet, (N - 1) & 7 = 0 ... 7, 32-step chunks, 64-step scalar chunks), ragged
    batches (a partly filled last workgroup), Q visible (sigma = 0.36, large R): against the oracle at the usual bars and against the
    one-wave kernel (same chain, the sums in the same fp16 x 2 class)."""
    from audio_mps_amd import _capi
    from audio_mps_amd.scan import unpack_grad
    kw = {"A": 66.0} if sigma == 0.36 else {}
    m, audio = _model(D, T, B, WAVE, sigma=sigma, seed=D + T, rscale=rscale, **kw)
    if sigma == 0.36:
        audio = (audio * np.float32(0.09)).astype(np.float32)
        m.data_iterator = audio
    be = m._get_backend()
    assert be._lib.cmps_get_option(be._h, _capi.CMPS_OPT_BWD_WAVES) == 2          # the default since round 5
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_BWD_WAVES, 1))
    be.kernel_events(True)
    g1 = unpack_grad(m.grad_sums(audio)[0].cpu().numpy(), D)
    assert "k_bwd_wave" in set(be.kernel_times())
    be.kernel_events(False)
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_BWD_WAVES, 2))
    be.kernel_events(True)
    flat = m.grad_sums(audio)[0].cpu().numpy()
    names = set(be.kernel_times())
    be.kernel_events(False)
    assert "k_bwd_wave2w" in names, names
    assert np.all(np.isfinite(flat))
    g2 = unpack_grad(flat, D)
    ref = c_oracle_run(m, audio, "f32")
    gr = C.unpack_grad(ref["grad"], D)
    g64 = C.unpack_grad(c_oracle_run(m, audio, "f64")["grad"], D)
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        own = rel_inf(gr[k], g64[k])
        assert rel_inf(g2[k], gr[k]) <= max(GRAD_RTOL, 3 * own), (k, rel_inf(g2[k], gr[k]), own)
        assert rel_inf(g2[k], g1[k]) <= max(GRAD_RTOL, 3 * own), (k, rel_inf(g2[k], g1[k]))


def test_f16_range_tripwire_and_fallback():
    """The run-time check of the fp16-split arithmetic (include/cmps.h: cmps_psi_grad_status).  The scales follow
    guaranteed bounds, so no input provokes an overflow; the diagnostic option CMPS_OPT_F16_SCALE_SHIFT pushes every data-dependent
    scale of the wave reverse scan up by 2^20 instead: the pieces of ybar overflow fp16, the gradient comes out non-finite while every
    per-clip loss is finite -> CMPS_ERR_F16_RANGE (not a silent NaN), the sticky word remembers it, cmps_psi_apply_step skips the
    update, and HipScan.loss_and_grad_sums(check=True) -- what PsiCMPS.grad_sums calls -- returns the BF16X3 result instead."""
    import warnings
    import torch
    from audio_mps_amd import _capi
    from audio_mps_amd.scan import unpack_grad
    from audio_mps_amd.train import Trainer
    D, T, B = 32, 300, 6
    m, audio = _model(D, T, B, WAVE, seed=11)
    be = m._get_backend()
    assert be.effective_rank1 == _capi.CMPS_RANK1_F16X2
    ref = c_oracle_run(m, audio, "f32")
    gr = C.unpack_grad(ref["grad"], D)
    d_audio = torch.from_numpy(audio).to(be.device)
    # in range: status OK, nothing sticky
    be.set_params(m.effective_params(), B, T, train=True)
    be.loss_and_grad_sums(d_audio)
    assert be.grad_status() == (_capi.CMPS_OK, 0)
    # out of range on purpose
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_F16_SCALE_SHIFT, 20))
    loss, grad = be.loss_and_grad_sums(d_audio)
    assert np.all(np.isfinite(loss.cpu().numpy())) and not np.all(np.isfinite(grad.cpu().numpy()[:2 * D * D]))
    code, sticky = be.grad_status()
    assert code == _capi.CMPS_ERR_F16_RANGE and sticky == 1
    assert b"fp16" in be._lib.cmps_last_error(be._h)
    assert be.grad_status() == (_capi.CMPS_ERR_F16_RANGE, 0)          # the last pass is still the bad one; the sticky word restarted
    # the documented fallback, as the model's accessor takes it: same batch with bf16x3 pieces, options restored
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        flat, nb = m.grad_sums(audio)
    assert be.f16_fallbacks == 1 and any("bf16x3" in str(x.message) for x in w)
    assert be.rank1 == _capi.CMPS_RANK1_DEFAULT and be.effective_rank1 == _capi.CMPS_RANK1_F16X2
    g = unpack_grad(flat.cpu().numpy(), D)
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        assert rel_inf(g[k], gr[k]) <= GRAD_RTOL, k
    # the device-resident optimiser step skips such a step: variables untouched, total loss NaN as the marker
    tr = Trainer(m, m.hparams, device_step=True)
    st = tr._device_state()
    before = st["vars"].clone()
    out = tr.step(audio, sync=True)
    assert np.isfinite(out["model_loss"]) and np.isnan(out["total_loss"])
    assert torch.equal(st["vars"], before)
    # back in range: the next step moves the variables again
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_F16_SCALE_SHIFT, 0))
    out = tr.step(audio, sync=True)
    assert np.isfinite(out["total_loss"]) and not torch.equal(st["vars"], before)
    assert be.grad_status()[0] == _capi.CMPS_OK


def test_nan_input_is_not_an_f16_range_error():
    """NaN / Inf that comes from the DATA propagates as in the reference (model.py has no guard; tests/test_model.py:113 only looks):
    loss and gradient are both non-finite, and cmps_psi_grad_status says CMPS_OK -- the check is about the split arithmetic only."""
    import torch
    from audio_mps_amd import _capi
    m, audio = _model(24, 130, 3, WAVE, seed=2)
    audio = audio.copy()
    audio[1, 40] = np.nan
    be = m._get_backend()
    be.set_params(m.effective_params(), 3, 130, train=True)
    loss, grad = be.loss_and_grad_sums(torch.from_numpy(audio).to(be.device))
    assert np.isnan(loss.cpu().numpy()[1]) and np.all(np.isfinite(loss.cpu().numpy()[[0, 2]]))
    code, sticky = be.grad_status()
    assert code == _capi.CMPS_OK and (sticky & 2)


def test_variants_agree():
    m1, audio = _model(32, 1500, 9, WAVE, seed=5)
    m2, _ = _model(32, 1500, 9, BLOCK, seed=5)
    p1, p2 = m1.loss_per_clip(), m2.loss_per_clip()
    assert np.max(np.abs(p1 - p2) / np.maximum(np.abs(p2), 1.0)) <= LOSS_RTOL
    f1 = m1.grad_sums()[0].cpu().numpy()
    f2 = m2.grad_sums()[0].cpu().numpy()
    assert rel_inf(f1[:2 * 32 * 32], f2[:2 * 32 * 32]) <= GRAD_RTOL


# ---------------------------------------------------------------------------------------------------
# size-independent properties at BASELINE's full size (configs[2]: D=32, T=16000, B=1024)
# ---------------------------------------------------------------------------------------------------
def test_full_size_properties():
    from audio_mps_amd import HParams, PsiCMPS
    D, T, B = 32, 16000, 1024
    hp = HParams(minibatch_size=B, bond_dim=D)
    base = make_audio(64, T, hp.delta_t, 9)
    audio = np.concatenate([base] * (B // 64), axis=0)           # 16 copies of 64 distinct clips
    m = PsiCMPS(hp, seed=0, backend=_scan(D, WAVE))
    per = m.loss_per_clip(audio)
    assert per.shape == (B,) and np.all(np.isfinite(per))
    # (1) determinism / independence of clips: identical clips give bit-identical losses wherever they sit
    np.testing.assert_array_equal(per.reshape(B // 64, 64), np.tile(per[:64], (B // 64, 1)))
    # (2) the reduced loss and gradient SUMS are additive over a split of the batch
    full = m.grad_sums(audio)[0].cpu().numpy().astype(np.float64)
    h1 = m.grad_sums(audio[:512])[0].cpu().numpy().astype(np.float64)
    h2 = m.grad_sums(audio[512:])[0].cpu().numpy().astype(np.float64)
    assert rel_inf(h1 + h2, full) <= 2e-5
    assert abs(full[-1] - per.astype(np.float64).sum()) <= 1e-5 * abs(full[-1])
    # (3) against the oracle on the 64 distinct clips
    ref = c_oracle_run(m, base, "f32", want_grad=False, nthreads=16)
    assert np.max(np.abs(per[:64] - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1.0)) <= LOSS_RTOL


def test_gradient_is_directional_derivative():
    """d/deps loss(R + eps dR) from two forward scans == <grad, dR> from the reverse scan (fp32, loose)."""
    m, audio = _model(16, 400, 8, WAVE, seed=21)
    loss0, grads = m.loss_and_grads()
    rng = np.random.default_rng(0)
    d = {k: rng.standard_normal(np.shape(v)).astype(np.float32) for k, v in m.variables.items()}
    eps = 1e-3
    base = {k: v.copy() for k, v in m.variables.items()}
    vals = []
    for sgn in (+1, -1):
        for k in base:
            m.variables[k] = (base[k] + sgn * eps * d[k]).astype(np.float32)
        vals.append(float(np.mean(m.loss_per_clip().astype(np.float64))))
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = sum(float(np.sum(grads[k].astype(np.float64) * d[k])) for k in base)
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1e-3), (fd, an)


# ---------------------------------------------------------------------------------------------------
# the reference's own invariants (tests/test_model.py), now on the HIP path
# ---------------------------------------------------------------------------------------------------
REF_TEST_HP = dict(r_reg=2 / (math.pi * 16000), h_reg=2 / (math.pi * 16000) ** 2)   # tests/test_model.py:13-14


@pytest.mark.parametrize("variant", [BLOCK, WAVE])
def test_loss_not_nan(variant):
    """TestPsiCMPS.testLossNotNaN (tests/test_model.py:107-113)."""
    m, _ = _model(7, 2 ** 8, 8, variant, **REF_TEST_HP)
    assert not np.isnan(m.loss)


@pytest.mark.parametrize("variant", [BLOCK, WAVE])
def test_psi_evolved_with_data_remains_normalized(variant):
    """TestPsiCMPS.testPsiEvolvedWithDataRemainsNormalized (tests/test_model.py:115-122)."""
    m, audio = _model(7, 2 ** 8, 8, variant, **REF_TEST_HP)
    psi = m.psi_evolve_with_data()
    assert psi.shape == (8, 2 ** 8 - 1, 7)
    np.testing.assert_allclose(np.linalg.norm(psi, axis=-1), np.ones(psi.shape[:2]), rtol=1e-5)
    _, ref = O.psi_loss_per_clip(oracle_hparams(m.hparams), oracle_variables(m), audio, return_states=True)
    assert np.max(np.abs(psi - ref)) < 2e-4        # |R| ~ 1e3 with these hparams: states decorrelate slowly


def test_trivial_update_of_ancilla():
    """TestPsiCMPS.testTrivialUpdateOfAncilla (tests/test_model.py:124-138)."""
    from audio_mps_amd import HParams, PsiCMPS
    hp = HParams(minibatch_size=8, bond_dim=7, **REF_TEST_HP)
    m = PsiCMPS(hp, freqs_in=np.zeros(7, np.float32), R_in=np.zeros((7, 7), np.complex64), backend=_scan(7, 0))
    signal = np.random.rand(8).astype(np.float32)
    stack = np.stack(8 * [m.psi_0])
    out = m._update_ancilla_psi(stack, signal, 0.0)
    np.testing.assert_allclose(out, stack, rtol=1e-6)


def test_update_ancilla_matches_oracle():
    m, _ = _model(16, 8, 4, 0, sigma=1.0, seed=4, rscale=0.1)
    rng = np.random.default_rng(1)
    psi = (rng.standard_normal((4, 16)) + 1j * rng.standard_normal((4, 16))).astype(np.complex64)
    sig = rng.standard_normal(4).astype(np.float32)
    ohp, ov = oracle_hparams(m.hparams), oracle_variables(m)
    R, f, _, _ = O.effective_params(ohp, ov)
    ref = O.update_ancilla_psi(psi, sig, 0.37, R, f, ov.A, ohp)
    out = m._update_ancilla_psi(psi, sig, 0.37)
    assert np.max(np.abs(out - ref)) <= 1e-5 * np.max(np.abs(ref))


# ---------------------------------------------------------------------------------------------------
# error behaviour at the ABI
# ---------------------------------------------------------------------------------------------------
def test_error_codes_and_nan_propagation():
    from audio_mps_amd import _capi
    from audio_mps_amd.scan import HipScan
    lib = _capi.load()
    sc = HipScan(8)
    m, audio = _model(8, 64, 4, 0)
    # backward before forward
    with pytest.raises(_capi.CmpsError) as ei:
        sc.set_params(m.effective_params(), 4, 64, train=True)
        sc._audio = torch.zeros((4, 64), device=sc.device)
        sc.backward()
    assert ei.value.code == _capi.CMPS_ERR_STATE
    # save_for_bwd with a forward-only workspace
    sc.set_params(m.effective_params(), 4, 64, train=False)
    with pytest.raises(_capi.CmpsError) as ei:
        sc.forward(torch.zeros((4, 64), device=sc.device), save_for_bwd=True)
    assert ei.value.code == _capi.CMPS_ERR_WORKSPACE
    # workspace too small
    h = ctypes.c_void_p()
    assert lib.cmps_create(8, ctypes.byref(h)) == 0
    buf = torch.zeros(1024, dtype=torch.uint8, device=sc.device)
    p = sc._param_buf.data_ptr()
    rc = lib.cmps_set_params(h, p, p, p, p, p, 100.0, 1e-4, 1 / 16000, 64, 4, 1, buf.data_ptr(), 1024, None)
    assert rc == _capi.CMPS_ERR_WORKSPACE and b"workspace" in lib.cmps_last_error(h)
    lib.cmps_destroy(h)
    # NaN is not an error at the boundary: it propagates into the loss like in the reference
    bad = audio.copy()
    bad[1, 10] = np.nan
    per = m.loss_per_clip(bad)
    assert np.isnan(per[1]) and np.all(np.isfinite(per[[0, 2, 3]]))
    # 1 + z <= 0 gives NaN / inf (plain log(1 + z), model.py:294), again without an error
    big = (audio * 1e6).astype(np.float32)
    assert not np.all(np.isfinite(m.loss_per_clip(big)))


# ---------------------------------------------------------------------------------------------------
# next row: PsiCMPS.sample
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [BLOCK, WAVE])
def test_sampling_two_level_system(variant):
    """TestPsiCMPS.testSampling (tests/test_model.py:140-158): R = [[0,1],[0,0]], freqs = [w, -w], sigma = 1, A = 1."""
    from audio_mps_amd import HParams, PsiCMPS
    hp = HParams(minibatch_size=8, bond_dim=2, delta_t=1 / 16000, sigma=1, A=1.0,
                 h_reg=2 / (math.pi * 16000) ** 2, r_reg=2 / (math.pi * 16000) ** 2)
    R = np.array([[0, 1], [0, 0]], dtype=np.complex64)
    freqs = np.array([10.0, -10.0], dtype=np.float32)
    qubit = PsiCMPS(hp, R_in=R, freqs_in=freqs, backend=_scan(2, variant))
    waveform = qubit.sample(num_samples=2, length=512, seed=3)
    assert waveform.shape == (2, 512) and np.all(np.isfinite(waveform))
    noise = O.sample_noise(oracle_hparams(hp), 2, 512, seed=5)
    ref = O.psi_sample(oracle_hparams(hp), oracle_variables(qubit), noise)
    out = qubit.sample(2, 512, noise=noise)
    assert np.max(np.abs(out - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("D,length,n,variant", [(8, 300, 5, WAVE), (32, 700, 9, WAVE), (32, 130, 3, BLOCK), (48, 100, 2, BLOCK)])
def test_sampling_matches_oracle(D, length, n, variant):
    from audio_mps_amd import HParams, PsiCMPS
    hp = HParams(minibatch_size=n, bond_dim=D, sigma=1.0, A=10.0)
    m = PsiCMPS(hp, seed=D, backend=_scan(D, variant))
    m.variables["Rx"] *= np.float32(0.05)
    m.variables["Ry"] *= np.float32(0.05)
    noise = O.sample_noise(oracle_hparams(hp), n, length, temp=0.5, seed=D)
    ref = O.psi_sample(oracle_hparams(hp), oracle_variables(m), noise)
    out = m.sample(n, length, noise=noise)
    assert out.shape == (n, length)
    assert np.max(np.abs(out - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("variant", [BLOCK, WAVE])
def test_normalisation_floor_branch(variant):
    """`tf.maximum(square_sum, 1e-12)` (model.py:332): a step whose update annihilates the state (I + s R) u = 0.
    Two-level system, R = [[0, a], [a, 0]] (eigenvalues +-a), psi_0 = the +a eigenvector, first increment x = -A/a."""
    from audio_mps_amd import HParams, PsiCMPS
    a, A = 2.0, 4.0
    hp = HParams(minibatch_size=3, bond_dim=2, sigma=0.0, A=A)
    R = np.array([[0, a], [a, 0]], dtype=np.complex64)
    m = PsiCMPS(hp, R_in=R, freqs_in=np.array([3.0, -5.0], dtype=np.float32),
                psi_in=np.array([1, 1], dtype=np.complex64), backend=_scan(2, variant))
    T = 40
    audio = make_audio(3, T, hp.delta_t, 77, noise=0.05)
    audio[1, 0] = 0.0
    audio[1, 1] = -A / a            # clip 1: s R u = -u at step 0  ->  |y|^2 = 0 <= 1e-12
    ref = c_oracle_run(m, audio, "f32")
    per = m.loss_per_clip(audio)
    assert np.all(np.isfinite(ref["loss_per_clip"])) and np.all(np.isfinite(per))
    assert np.max(np.abs(per - ref["loss_per_clip"])) <= 1e-5 * max(1.0, np.max(np.abs(ref["loss_per_clip"])))
    from audio_mps_amd.scan import unpack_grad
    flat = m.grad_sums(audio)[0].cpu().numpy()
    g, gr = unpack_grad(flat, 2), C.unpack_grad(ref["grad"], 2)
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        assert rel_inf(g[k], gr[k]) <= 1e-3, k


def test_more_clips_than_simds():
    """B = 1030 (> 1024 waves, not a multiple of 4): several rounds of workgroups per CU."""
    m, audio = _model(32, 130, 1030, WAVE, seed=11)
    per = m.loss_per_clip()
    ref = c_oracle_run(m, audio, "f32", want_grad=True, nthreads=16)
    assert np.max(np.abs(per - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1.0)) <= LOSS_RTOL
    from audio_mps_amd.scan import unpack_grad
    g, gr = unpack_grad(m.grad_sums()[0].cpu().numpy(), 32), C.unpack_grad(ref["grad"], 32)
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        assert rel_inf(g[k], gr[k]) <= GRAD_RTOL, k


# ---------------------------------------------------------------------------------------------------
# next row: the legacy AudioMPS arithmetic
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,T,B,dt", [(5, 200, 8, 0.01), (10, 300, 4, 0.001), (32, 150, 3, 0.01), (40, 60, 2, 0.01),
                                      (7, 2, 3, 0.01), (32, 34, 5, 0.004), (16, 1000, 9, 0.002),       # one step; a one-step last chunk; many chunks
                                      # D > 32: the wide kernels in legacy mode (round 5; cmps_wide.hip): the three padded sizes, odd batches,
                                      # one step, T - 1 around the 64-step scalar chunks and the 4-step GEMM units
                                      (64, 300, 4, 0.004), (96, 130, 3, 0.004), (128, 200, 5, 0.002), (33, 2, 1, 0.01), (72, 66, 2, 0.004),
                                      (100, 65, 3, 0.002), (48, 129, 1, 0.01), (64, 1000, 6, 0.001)])
def test_legacy_audiomps_matches_oracle(D, T, B, dt):
    from audio_mps_amd import AudioMPS, LegacyAudioMPS
    audio = make_audio(B, T, dt, D, noise=0.05)
    m = AudioMPS(D, dt, B, data_iterator=audio, arithmetic="legacy", seed=D)
    assert isinstance(m, LegacyAudioMPS)
    ref = O.legacy_loss_and_grads(m.variables["H"], m.variables["R"], dt, audio, "f32")
    per = m.loss_per_clip()
    assert np.max(np.abs(per - ref["per_clip"]) / np.maximum(np.abs(ref["per_clip"]), 1.0)) <= LOSS_RTOL
    loss, grads = m.loss_and_grads()
    assert abs(float(loss) - float(ref["loss"])) <= LOSS_RTOL * max(1.0, abs(float(ref["loss"])))
    assert rel_inf(grads["R"], ref["gR"]) <= GRAD_RTOL
    assert rel_inf(grads["H"], ref["gH"]) <= GRAD_RTOL
    assert np.allclose(np.triu(grads["H"], 1), 0)                  # only the lower triangle of H is used


@pytest.mark.parametrize("D,T,dt,scale", [(18, 60, 0.01, 1.5), (32, 60, 0.004, 2.0), (32, 501, 0.002, 1.5), (9, 47, 0.01, 2.0), (40, 75, 0.004, 1.5)])
def test_legacy_qbar_is_the_H_gradient_at_large_R(D, T, dt, scale):
    """VERDICT r4 weak 2, legacy row: in the legacy arithmetic Q = dt (-i H_s - R^T R / 2) is never negligible and
    Qbar = sum ybar psi^dagger is a direct OUTPUT (dQ of cmps_legacy_loss_bwd; the H gradient is its imaginary part), so the hazard class of
    DESIGN 4.3e shows up in gH.  Large H and R (x 1.5 ... 2 of the initialisation) with T above, on and below the octet / chunk boundaries,
    every rank-1 arithmetic of k_bwd_wave<*, LEGACY> (and the general kernels at D = 40), against the float64 oracle with the float32
    oracle's own distance (x 3) as the bar."""
    from audio_mps_amd import LegacyAudioMPS
    from audio_mps_amd.scan import HipScan
    B = 5
    audio = make_audio(B, T, dt, 3, noise=0.05)
    ref64 = own = None
    for mode in ((0, 1, 2, 3) if D <= 32 else (4,)):
        m = LegacyAudioMPS(D, dt, B, data_iterator=audio, seed=7, backend=HipScan(D, rank1=mode))
        m.variables["H"] *= np.float32(scale)
        m.variables["R"] *= np.float32(scale)
        if ref64 is None:
            ref64 = O.legacy_loss_and_grads(m.variables["H"].astype(np.float64), m.variables["R"].astype(np.float64), dt, audio, "f64")
            ref32 = O.legacy_loss_and_grads(m.variables["H"], m.variables["R"], dt, audio, "f32")
            own = {k: rel_inf(ref32[k], ref64[k]) for k in ("gR", "gH")}
            assert np.all(np.isfinite(ref32["per_clip"]))
        per = m.loss_per_clip()
        assert np.max(np.abs(per - ref32["per_clip"]) / np.maximum(np.abs(ref32["per_clip"]), 1.0)) <= LOSS_RTOL
        _, grads = m.loss_and_grads()
        for k, name in (("R", "gR"), ("H", "gH")):
            bar = 3 * own[name] + (3e-5 if mode == 1 else 1e-5)
            assert rel_inf(grads[k], ref64[name]) <= bar, (mode, k, rel_inf(grads[k], ref64[name]), own[name])


@pytest.mark.parametrize("D,T", [(40, 75), (64, 260), (96, 131), (128, 70)])
@pytest.mark.parametrize("mode", [2, 4])
def test_legacy_wide_and_general_kernels_agree(D, T, mode):
    """32 < D <= 128 runs the wide kernels in legacy mode (k_fwd_wide / k_bwd_wide<LEGACY> + the H y and gradient GEMMs, two fp16 or three
    bf16 pieces); CMPS_VARIANT_BLOCK keeps the general one-workgroup-per-clip kernels (cmps_legacy.hip): two independent implementations
    of SURVEY Appendix A.  Also the forward-only path (the loss product inside the chain kernel)."""
    from audio_mps_amd import LegacyAudioMPS
    from audio_mps_amd.scan import HipScan
    B, dt = 5, 0.004
    audio = make_audio(B, T, dt, D, noise=0.05)
    a = LegacyAudioMPS(D, dt, B, data_iterator=audio, seed=4, backend=HipScan(D, rank1=mode))
    b = LegacyAudioMPS(D, dt, B, data_iterator=audio, seed=4, backend=HipScan(D, variant=BLOCK))
    pa, pb = a.loss_per_clip(), b.loss_per_clip()                 # forward only (no stash)
    assert np.max(np.abs(pa - pb) / np.maximum(np.abs(pb), 1.0)) <= LOSS_RTOL
    la, ga = a.loss_and_grads()                                   # training forward (chain + GEMM + loss kernel) and reverse
    lb, gb = b.loss_and_grads()
    assert abs(float(la) - float(lb)) <= LOSS_RTOL * max(1.0, abs(float(lb)))
    for k in ga:
        assert rel_inf(ga[k], gb[k]) <= GRAD_RTOL, k


@pytest.mark.parametrize("T", [65, 130, 400])
def test_legacy_wave_and_block_kernels_agree(T):
    """D <= 32 runs the wave-per-clip kernels in legacy mode (k_fwd_wave2<LEGACY>, k_bwd_wave<LEGACY>); CMPS_VARIANT_BLOCK forces the block kernels."""
    from audio_mps_amd import LegacyAudioMPS
    from audio_mps_amd.scan import HipScan
    audio = make_audio(6, T, 0.004, 3, noise=0.05)
    a = LegacyAudioMPS(24, 0.004, 6, data_iterator=audio, seed=4)
    b = LegacyAudioMPS(24, 0.004, 6, data_iterator=audio, seed=4, backend=HipScan(24, variant=BLOCK))
    pa, pb = a.loss_per_clip(), b.loss_per_clip()
    assert np.max(np.abs(pa - pb) / np.maximum(np.abs(pb), 1.0)) <= LOSS_RTOL
    la, ga = a.loss_and_grads()
    lb, gb = b.loss_and_grads()
    for k in ga:
        assert rel_inf(ga[k], gb[k]) <= GRAD_RTOL, k


def test_legacy_full_size_properties():
    """The legacy arithmetic at the shape of scripts/bench_next_rows.py (D=32, T=4000, 1024 clips = one clip per SIMD): finite
    results, the reported loss is the mean of the per-clip losses, and the clip order does not matter (per-clip losses permute
    exactly, gradients agree up to float32 summation order)."""
    from audio_mps_amd import LegacyAudioMPS
    B, T, dt = 1024, 4000, 0.001
    audio = make_audio(B, T, dt, 11, noise=0.05)
    m = LegacyAudioMPS(32, dt, B, data_iterator=audio, seed=7)
    per = m.loss_per_clip()
    loss, grads = m.loss_and_grads()
    assert np.all(np.isfinite(per)) and all(np.all(np.isfinite(v)) for v in grads.values())
    assert abs(float(loss) - float(np.mean(per, dtype=np.float64))) <= 1e-5 * max(abs(float(loss)), 1.0)
    perm = np.random.default_rng(1).permutation(B)
    m2 = LegacyAudioMPS(32, dt, B, data_iterator=audio[perm], seed=7)
    np.testing.assert_array_equal(m2.loss_per_clip(), per[perm])
    _, grads2 = m2.loss_and_grads()
    for k in grads:
        assert rel_inf(grads2[k], grads[k]) <= 1e-5, k


def test_reference_default_clip_length():
    """T = 2**16, the reference's default --sample_duration (train.py:27): four times the benchmark length.  The loss stays
    inside the 1e-5 bar; the gradient bar is doubled here (observed 5e-5 ... 9e-5: the float32 restatement's own distance
    from float64 grows with the number of steps as well)."""
    m, audio = _model(32, 65536, 6, WAVE, seed=5)
    _check_against_oracle(m, audio, nthreads=6, grad_rtol=2e-4)


@pytest.mark.parametrize("D,T,B,variant", [(32, 700, 9, WAVE), (12, 300, 5, WAVE), (48, 90, 3, BLOCK), (128, 130, 5, 3), (64, 77, 2, 3)])
def test_runs_are_bit_reproducible(D, T, B, variant):
    """Two forward + reverse passes on the same input give the same BITS (per-clip losses and the reduced gradient sums):
    every cross-wave / cross-clip reduction runs in a fixed order, nothing is accumulated with atomics, and no kernel reads
    what another wave is still writing (a race would show up here as run-to-run noise)."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan
    hp = HParams(minibatch_size=B, bond_dim=D)
    audio = torch.from_numpy(make_audio(B, T, hp.delta_t, 31)).cuda()
    be = HipScan(D, variant=variant)
    m = PsiCMPS(hp, seed=3, backend=be)
    be.set_params(m.effective_params(), B, T, train=True)
    ref = None
    for _ in range(3):
        be.forward(audio, save_for_bwd=True)
        be.backward()
        torch.cuda.synchronize()
        cur = (be._loss.clone().cpu().numpy(), be._grad.clone().cpu().numpy())
        assert np.all(np.isfinite(cur[0])) and np.all(np.isfinite(cur[1]))
        if ref is None:
            ref = cur
        else:
            np.testing.assert_array_equal(cur[0], ref[0])
            np.testing.assert_array_equal(cur[1], ref[1])
