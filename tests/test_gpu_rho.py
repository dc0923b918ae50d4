"""RhoCMPS on the MI355X (cmps_rho_* entry points) against the oracle's matrix-form restatement of model.py:55-203,
plus the reference's own TestRhoCMPS cases (tests/test_model.py:31-103).

The two sides use different formulations on purpose: the oracle carries the D x D matrix in the lab frame exactly as
the reference does; the kernels carry the `rank` columns of rho in the rotating frame.  Tolerances as in
test_gpu_parity.py: per-clip loss 1e-5 * max(|loss|, 1), gradients 1e-4 of each tensor's max.
"""
import numpy as np
import pytest

from oracle import cmps_oracle as O
from _util import make_audio, rel_inf

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5
GRAD_RTOL = 1e-4


def _rho_model(D, T, B, rank=None, sigma=1e-4, seed=0, rscale=None, A=100.0, data=True, **kw):
    from audio_mps_amd import HParams, RhoCMPS
    hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, initial_rank=rank, A=A)
    audio = make_audio(B, T, hp.delta_t, seed) if data else None
    m = RhoCMPS(hp, data_iterator=audio, seed=seed, **kw)
    if rscale is not None:
        m.variables["Rx"] *= np.float32(rscale)
        m.variables["Ry"] *= np.float32(rscale)
    return m, audio


def _oracle_side(m):
    ohp = O.HParams(**m.hparams.values())
    D = m.bond_d
    ov = O.Variables(np.asarray(m.variables["A"], dtype=np.float32), m.variables["Rx"].copy(), m.variables["Ry"].copy(),
                     m.variables["freqs"].copy(), np.zeros(D, np.float32), np.zeros(D, np.float32),
                     scaled_R=float(m._c_r) != 1.0, scaled_freqs=float(m._c_h) != 1.0)
    return ohp, ov, m.variables["Wx"], m.variables["Wy"]


# ---------------------------------------------------------------------------------------------------
# the reference's TestRhoCMPS (tests/test_model.py:31-103), hparams of :13-14
# ---------------------------------------------------------------------------------------------------
def _ref_hparams(**over):
    from audio_mps_amd import HParams
    sr = 16000
    kw = dict(minibatch_size=8, bond_dim=7, delta_t=1 / sr, sigma=0.0001, initial_rank=None, A=100.,
              h_reg=2 / (np.pi * sr) ** 2, r_reg=2 / (np.pi * sr))
    kw.update(over)
    return HParams(**kw)


def test_loss_not_nan():
    from audio_mps_amd import RhoCMPS
    from audio_mps_amd.data import get_audio
    hp = _ref_hparams()
    data = get_audio(None, "damped_sine", hp, 2 ** 8)
    model = RhoCMPS(hp, data_iterator=data)
    assert not np.isnan(model.loss)


def test_rho_evolved_with_data_remains_normalized():
    from audio_mps_amd import RhoCMPS
    from audio_mps_amd.data import get_audio
    hp = _ref_hparams()
    data = get_audio(None, "damped_sine", hp, 2 ** 8)
    model = RhoCMPS(hp, data_iterator=data)
    rho_out = model.rho_evolve_with_data()
    assert rho_out.shape == (8, 255, 7, 7)
    np.testing.assert_allclose(np.trace(rho_out, axis1=2, axis2=3), np.ones((8, 255)), rtol=1e-5)


def test_rho_evolved_sampling_remains_normalized():
    from audio_mps_amd import RhoCMPS
    model = RhoCMPS(_ref_hparams())
    rho_out = model.rho_evolve_with_sampling(num_samples=5, length=256, seed=1)
    assert rho_out.shape == (5, 256, 7, 7)
    np.testing.assert_allclose(np.trace(rho_out, axis1=2, axis2=3), np.ones((5, 256)), rtol=1e-4)


def test_trivial_update_of_ancilla():
    """Update with H = R = 0 leaves rho alone."""
    from audio_mps_amd import RhoCMPS
    hp = _ref_hparams()
    D = hp.bond_dim
    signal = np.random.default_rng(0).random(hp.minibatch_size).astype(np.float32)
    model = RhoCMPS(hp, freqs_in=np.zeros(D, np.float32), R_in=np.zeros((D, D), np.complex64))
    stack = np.stack(hp.minibatch_size * [model.rho_0])
    np.testing.assert_allclose(model._update_ancilla_rho(stack, signal, 0.), stack, rtol=1e-6, atol=1e-7)


def test_sampling_two_level_system():
    from audio_mps_amd import RhoCMPS
    hp = _ref_hparams(bond_dim=2, sigma=1, A=1.)
    w = 10
    qubit = RhoCMPS(hp, R_in=np.array([[0, 1], [0, 0]], dtype=np.complex64), freqs_in=np.array([w, -w], dtype=np.float32))
    waveform = qubit.sample(num_samples=2, length=512, seed=3)
    assert waveform.shape == (2, 512) and np.all(np.isfinite(waveform))


# ---------------------------------------------------------------------------------------------------
# parity with the oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,T,B,rank,sigma,rscale", [
    (4, 256, 8, None, 1e-4, None),       # BASELINE C1 shape
    (7, 256, 8, None, 1e-4, None),       # the reference tests' shape
    (7, 100, 3, 2, 0.5, 0.3),            # low rank, visible damping
    (16, 130, 4, 5, 1e-4, None),
    (32, 96, 3, 32, 1e-4, None),         # full rank at the wave kernels' D (row-array GEMM kernels, cmps_rho_mfma.hip)
    (32, 150, 3, 17, 0.3, 0.4),          # GEMM kernels: ragged rank, visible damping, three 64-step chunks
    (20, 129, 2, 9, 0.2, 0.5),           # GEMM kernels: padded D, the smallest rank they take, a one-step last chunk
    (40, 40, 2, 6, 1e-4, None),          # D > 32
    (72, 24, 2, 3, 1e-4, None),          # D > 64 (two-wave workgroups)
    (64, 12, 2, 40, 1e-4, None),         # more than 64 KB of LDS-resident columns in the reverse sweep
    (80, 10, 2, None, 1e-4, 0.5),        # the reference's default rank = D (model.py:62-65) beyond rank * D = 5000: workspace columns
    (96, 8, 2, None, 1e-4, 0.4),         # (96, 96)
    (128, 6, 2, None, 1e-4, 0.3),        # (128, 128): the largest bond dimension, full rank
])
def test_rho_loss_and_gradients_match_oracle(D, T, B, rank, sigma, rscale):
    m, audio = _rho_model(D, T, B, rank=rank, sigma=sigma, seed=D + T, rscale=rscale)
    ohp, ov, Wx, Wy = _oracle_side(m)
    ref = O.rho_loss_and_grads(ohp, ov, Wx, Wy, audio, "f32")
    ref64 = O.rho_loss_and_grads(ohp, ov.astype(np.float64), Wx.astype(np.float64), Wy.astype(np.float64), audio, "f64")
    per = m.loss_per_clip()
    err = np.max(np.abs(per - ref["per_clip"]) / np.maximum(np.abs(ref["per_clip"]), 1.0))
    assert err <= LOSS_RTOL, f"loss rel err {err}"
    loss, grads = m.loss_and_grads()
    assert abs(float(loss) - float(ref["loss"])) <= LOSS_RTOL * max(abs(float(ref["loss"])), 1.0)
    for k in ("A", "Rx", "Ry", "freqs", "Wx", "Wy"):
        # the float32 oracle's own gradient error (vs float64) bounds what can be asked of the kernels
        own = rel_inf(ref[k], ref64[k])
        e = rel_inf(grads[k], ref64[k])
        assert e <= max(GRAD_RTOL, 3 * own), f"{k}: rel err {e} (oracle f32 vs f64: {own})"


def test_rank_one_rho_reproduces_the_pure_state_path():
    """rho_0 = |psi_0><psi_0| evolves exactly like PsiCMPS: the two HIP paths must agree on the loss."""
    from audio_mps_amd import HParams, PsiCMPS, RhoCMPS
    hp = HParams(minibatch_size=6, bond_dim=32, initial_rank=1)
    audio = make_audio(6, 500, hp.delta_t, 11)
    psi = PsiCMPS(hp, data_iterator=audio, seed=5)
    rho = RhoCMPS(hp, data_iterator=audio, seed=5, W_in=np.conj(psi.psi_0)[None, :])
    for k in ("A", "Rx", "Ry", "freqs"):
        rho.variables[k] = psi.variables[k].copy()
    a, b = psi.loss_per_clip(), rho.loss_per_clip()
    assert np.max(np.abs(a - b) / np.maximum(np.abs(a), 1.0)) <= LOSS_RTOL


def test_rho_gradient_is_directional_derivative():
    """Central difference of the HIP loss along a random direction in (A, Rx, Ry, freqs, Wx, Wy) vs the HIP gradient."""
    m, audio = _rho_model(8, 64, 4, rank=3, sigma=0.3, seed=21, rscale=0.2, A=5.0)
    _, grads = m.loss_and_grads()
    rng = np.random.default_rng(0)
    names = ("A", "Rx", "Ry", "freqs", "Wx", "Wy")
    dirs = {k: rng.standard_normal(np.shape(m.variables[k])).astype(np.float32) for k in names}
    dd = sum(float(np.sum(grads[k].astype(np.float64) * dirs[k])) for k in names)
    base = {k: np.array(m.variables[k], copy=True) for k in names}
    eps = 2e-3
    vals = []
    for sgn in (+1, -1):
        for k in names:
            m.variables[k] = (base[k] + sgn * eps * dirs[k]).astype(np.float32)
        vals.append(float(np.mean(m.loss_per_clip().astype(np.float64))))
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert fd == pytest.approx(dd, rel=2e-2, abs=1e-4)


def test_rho_update_ancilla_matches_oracle():
    m, _ = _rho_model(7, 16, 5, rank=3, sigma=0.8, seed=2, rscale=0.3, data=False)
    ohp, ov, Wx, Wy = _oracle_side(m)
    rng = np.random.default_rng(1)
    Z = rng.standard_normal((5, 7, 7)) + 1j * rng.standard_normal((5, 7, 7))
    rho = np.einsum("bij,bkj->bik", Z, np.conj(Z)).astype(np.complex64)          # general Hermitian PSD input
    rho /= np.trace(rho, axis1=1, axis2=2)[:, None, None]
    signal = rng.standard_normal(5).astype(np.float32)
    t = 0.0123
    out = m._update_ancilla_rho(rho, signal, t)
    ref = O.rho_update_ancilla(ohp, ov, rho, signal, t)
    assert rel_inf(out, ref) <= 1e-5


@pytest.mark.parametrize("D,rank,length,n", [(7, None, 256, 5), (2, None, 512, 2), (32, 4, 200, 3), (40, 3, 64, 2), (96, None, 12, 2)])
def test_rho_sampling_matches_oracle(D, rank, length, n):
    sigma = 1.0 if D == 2 else 0.05
    m, _ = _rho_model(D, 8, 2, rank=rank, sigma=sigma, seed=D, rscale=0.2, A=1.0 if D == 2 else 10.0, data=False)
    ohp, ov, Wx, Wy = _oracle_side(m)
    rng = np.random.default_rng(4)
    noise = (sigma * np.sqrt(ohp.delta_t) * rng.standard_normal((length, n))).astype(np.float32)
    wav = m.sample(n, length, noise=noise)
    rhos = m.rho_evolve_with_sampling(n, length, noise=noise)
    pur = m.purity(n, length, noise=noise)
    rw, rr, rp = O.rho_sample(ohp, ov, Wx, Wy, noise)
    scale = max(float(np.max(np.abs(rw))), 1e-6)
    assert np.max(np.abs(wav - rw)) <= 2e-4 * scale
    assert rel_inf(rhos, rr) <= 2e-4
    np.testing.assert_allclose(pur, rp, rtol=2e-4, atol=1e-6)
    assert np.all(pur <= 1 + 1e-4) and np.all(pur >= 1.0 / D - 1e-4)


def test_rho_trainer_step_decreases_loss():
    """train.py's loop with --mps_model rho_mps: Adam on the total loss through the HIP RhoCMPS scan."""
    from audio_mps_amd import HParams, RhoCMPS
    from audio_mps_amd.train import Trainer
    hp = HParams(minibatch_size=8, bond_dim=8, initial_rank=3, learning_rate=1e-2)
    audio = make_audio(8, 256, hp.delta_t, 3)
    m = RhoCMPS(hp, data_iterator=audio, seed=1)
    tr = Trainer(m, hp)
    losses = [tr.step()["total_loss"] for _ in range(25)]
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_rho_error_codes():
    from audio_mps_amd import _capi
    m, audio = _rho_model(4, 32, 2, rank=2)
    be = m._get_backend()
    import torch
    d_audio = torch.from_numpy(audio).to(be.device)
    be.set_params(m.effective_params(), 2, 32, train=False)
    with pytest.raises(_capi.CmpsError) as ei:               # no cmps_rho_set_state yet
        be.rho_forward(d_audio)
    assert ei.value.code == _capi.CMPS_ERR_STATE
    be.rho_set_state(m.columns(), 2, 32, train=False)
    with pytest.raises(_capi.CmpsError) as ei:               # forward-only workspace cannot save
        be.rho_forward(d_audio, save_for_bwd=True)
    assert ei.value.code == _capi.CMPS_ERR_WORKSPACE
    be.rho_forward(d_audio)
    with pytest.raises(_capi.CmpsError) as ei:               # backward without a saved forward
        be.rho_backward()
    assert ei.value.code == _capi.CMPS_ERR_STATE
    from audio_mps_amd.scan import HipScan
    be64 = HipScan(64)
    from audio_mps_amd.scan import EffectiveParams
    be64.set_params(EffectiveParams(R=np.zeros((64, 64), np.complex64), freqs=np.zeros(64, np.float32),
                                    psi0=np.eye(64, dtype=np.complex64)[0], A=1.0, sigma=1.0, delta_t=1e-3), 1, 8, train=False)
    with pytest.raises(_capi.CmpsError) as ei:               # rank above 128
        be64.rho_set_state(np.zeros((129, 64), np.complex64), 1, 8, train=True)
    assert ei.value.code == _capi.CMPS_ERR_UNSUPPORTED_D
    be64.rho_set_state(np.zeros((128, 64), np.complex64), 1, 8, train=True)     # rank * D = 8192: columns in the workspace


@pytest.mark.parametrize("D,rank", [(32, 5), (12, 12)])
def test_rho_wave_and_block_kernels_agree(D, rank):
    """D <= 32 runs the wave-per-clip RhoCMPS kernels (cmps_rho_wave.hip); CMPS_VARIANT_BLOCK forces the general block
    kernels (cmps_rho.hip).  Two independent implementations of the same column recurrence must agree."""
    from audio_mps_amd import RhoCMPS
    from audio_mps_amd.scan import HipScan
    m, audio = _rho_model(D, 200, 5, rank=rank, sigma=0.2, seed=9, rscale=0.5)
    blk = RhoCMPS(m.hparams, data_iterator=audio, seed=9, backend=HipScan(D, variant=1))
    for k in m.variables:
        blk.variables[k] = m.variables[k].copy()
    a, b = m.loss_per_clip(), blk.loss_per_clip()
    assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) <= LOSS_RTOL
    la, ga = m.loss_and_grads()
    lb, gb = blk.loss_and_grads()
    for k in ga:
        assert rel_inf(ga[k], gb[k]) <= GRAD_RTOL, k
    ra, rb = m.rho_evolve_with_data(), blk.rho_evolve_with_data()
    assert rel_inf(ra, rb) <= 1e-5


@pytest.mark.parametrize("D,rank,T,B,sigma,rscale", [
    (64, 16, 260, 3, 0.2, 0.4),          # scripts/bench_next_rows.py's shape (reduced): visible damping
    (40, 7, 131, 2, 0.3, 0.5),           # odd rank (a zero column pads the last pair), zero-padded rows (40 -> 64), one-step last chunk
    (96, 9, 66, 3, 1e-4, 0.4),           # PD = 96
    (128, 24, 70, 2, 0.1, 0.3),          # PD = 128
    (48, 48, 40, 2, 1e-4, 0.5),          # the reference's default rank = D (model.py:62-65): 24 column pairs
    (64, 1, 130, 5, 0.2, 0.5),           # rank 1
])
def test_rho_wide_kernels_match_oracle_and_general_kernels(D, rank, T, B, sigma, rscale):
    """32 < D <= 128 (round 5): the training forward / reverse run the columns of rho as VIRTUAL clips of the pure-state wide kernels
    (cmps_wide.hip: k_fwd_wide_rho, then k_hy_wide / k_bwd_wide / k_grad_gemm unchanged on rank x as many clips).  Against the oracle's
    matrix form (/root/reference/model.py:133-203 restated) and against the general one-workgroup-per-clip kernels (CMPS_VARIANT_BLOCK):
    loss, all six gradient tensors, the states rho_k and the purity read from the wide stash layout."""
    from audio_mps_amd import RhoCMPS
    from audio_mps_amd.scan import HipScan
    m, audio = _rho_model(D, T, B, rank=rank, sigma=sigma, seed=D + rank, rscale=rscale)
    be = m._get_backend()
    be.kernel_events(True)
    loss, grads = m.loss_and_grads()
    names = set(be.kernel_times())
    be.kernel_events(False)
    assert {"k_fwd_wide_rho", "k_bwd_wide", "k_grad_gemm", "k_rho_merge_e"} <= names, names       # the wide path really ran
    ohp, ov, Wx, Wy = _oracle_side(m)
    ref = O.rho_loss_and_grads(ohp, ov, Wx, Wy, audio, "f32")
    ref64 = O.rho_loss_and_grads(ohp, ov.astype(np.float64), Wx.astype(np.float64), Wy.astype(np.float64), audio, "f64")
    assert abs(float(loss) - float(ref["loss"])) <= LOSS_RTOL * max(abs(float(ref["loss"])), 1.0)
    for k in ("A", "Rx", "Ry", "freqs", "Wx", "Wy"):
        own = rel_inf(ref[k], ref64[k])
        e = rel_inf(grads[k], ref64[k])
        assert e <= max(GRAD_RTOL, 3 * own), f"{k}: rel err {e} (oracle f32 vs f64: {own})"
    blk = RhoCMPS(m.hparams, data_iterator=audio, seed=1, backend=HipScan(D, variant=1))
    for k in m.variables:
        blk.variables[k] = m.variables[k].copy()
    lb, gb = blk.loss_and_grads()
    assert abs(float(loss) - float(lb)) <= LOSS_RTOL * max(abs(float(lb)), 1.0)
    for k in grads:
        assert rel_inf(grads[k], gb[k]) <= (GRAD_RTOL if k != "A" else 10 * GRAD_RTOL), k
    ra, rb = m.rho_evolve_with_data(), blk.rho_evolve_with_data()
    np.testing.assert_allclose(np.trace(ra, axis1=2, axis2=3).real, np.ones((B, T - 1)), rtol=1e-5)
    assert rel_inf(ra, rb) <= 2e-5


@pytest.mark.parametrize("D,rank,T,B,sigma,rscale", [(32, 32, 200, 5, 0.3, 0.4), (32, 11, 65, 3, 1e-4, None), (24, 24, 130, 9, 0.2, 0.5),
                                                    (20, 9, 257, 2, 0.4, 0.6), (32, 17, 40, 7, 0.1, 0.3)])
def test_rho_reverse_on_virtual_clips_matches_gemm_reverse_and_oracle(D, rank, T, B, sigma, rscale):
    """D <= 32, rank > 8 (round 5; CMPS_OPT_RHO_BWD): the reverse sweep that follows the row-array GEMM forward is, by default, the
    pure-state wave reverse scan (k_bwd_wave2w, or k_bwd_wave) on one VIRTUAL clip per column (cmps_rho_wave.hip) -- given the clip's per-step scalars the
    column cotangents do not couple.  Against k_bwd_rho_mfma (CMPS_RHO_BWD_GEMM, the round 1-4 reverse sweep: a different formulation,
    cotangent array as GEMMs) and against the matrix-form oracle; every rank-1 arithmetic of the wave scan."""
    from audio_mps_amd import _capi
    m, audio = _rho_model(D, T, B, rank=rank, sigma=sigma, seed=D + T, rscale=rscale)
    be = m._get_backend()
    ohp, ov, Wx, Wy = _oracle_side(m)
    ref = O.rho_loss_and_grads(ohp, ov, Wx, Wy, audio, "f32")
    ref64 = O.rho_loss_and_grads(ohp, ov.astype(np.float64), Wx.astype(np.float64), Wy.astype(np.float64), audio, "f64")
    assert be._lib.cmps_get_option(be._h, _capi.CMPS_OPT_RHO_BWD) == _capi.CMPS_RHO_BWD_VIRTUAL
    be.kernel_events(True)
    loss, gv = m.loss_and_grads()
    names = set(be.kernel_times())
    be.kernel_events(False)
    assert "k_bwd_wave2w" in names, names                       # F16X2 sums: the two-wave scan (CMPS_OPT_BWD_WAVES = 2, the default)
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_BWD_WAVES, 1))
    be.kernel_events(True)
    _, g1 = m.loss_and_grads()
    assert "k_bwd_wave" in set(be.kernel_times())
    be.kernel_events(False)
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_BWD_WAVES, 2))
    for k in ("A", "Rx", "Ry", "freqs", "Wx", "Wy"):
        assert rel_inf(g1[k], ref64[k]) <= max(GRAD_RTOL, 3 * rel_inf(ref[k], ref64[k])), ("one wave", k)
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_RHO_BWD, _capi.CMPS_RHO_BWD_GEMM))
    _, gg_ = m.loss_and_grads()
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_RHO_BWD, _capi.CMPS_RHO_BWD_VIRTUAL))
    for k in ("A", "Rx", "Ry", "freqs", "Wx", "Wy"):
        own = rel_inf(ref[k], ref64[k])
        assert rel_inf(gv[k], ref64[k]) <= max(GRAD_RTOL, 3 * own), (k, rel_inf(gv[k], ref64[k]), own)
        assert rel_inf(gv[k], gg_[k]) <= max(GRAD_RTOL, 3 * own), (k, rel_inf(gv[k], gg_[k]))
    for mode in (0, 1, 2):
        be.set_rank1(mode)
        _, gm = m.loss_and_grads()
        for k in ("Rx", "Ry", "Wx", "Wy"):
            assert rel_inf(gm[k], ref64[k]) <= max(GRAD_RTOL, 3 * rel_inf(ref[k], ref64[k])) + (3e-5 if mode == 1 else 0.0), (mode, k)


def test_rho_wide_rank_scaling_and_batch_order():
    """The cost of the wide path is linear in the rank (the GEMM kernels of D <= 32 cost the same at rank 4 and 32), and clips are
    independent: permuting the batch permutes the per-clip losses (training forward) and leaves the gradient sums unchanged."""
    import torch
    m, audio = _rho_model(64, 300, 6, rank=8, sigma=0.1, seed=3, rscale=0.4)
    perm = np.random.default_rng(0).permutation(6)
    be = m._prepare(6, 300, train=True)
    la = be.rho_forward(m._to_device(audio), save_for_bwd=True).cpu().numpy().copy()
    ga = be.rho_backward().cpu().numpy().copy()
    lb = be.rho_forward(m._to_device(audio[perm]), save_for_bwd=True).cpu().numpy().copy()
    gb = be.rho_backward().cpu().numpy().copy()
    np.testing.assert_array_equal(lb, la[perm])
    assert rel_inf(gb, ga) <= 1e-5
    assert np.all(np.isfinite(ga))


@pytest.mark.parametrize("D,rank,T,B", [(32, 32, 200, 5), (32, 11, 65, 3), (24, 24, 130, 9)])
def test_rho_gemm_and_column_kernels_agree(D, rank, T, B):
    """rank > 8 runs the scan as row-array GEMMs on the matrix cores (cmps_rho_mfma.hip, bf16 x 3 operand split);
    CMPS_VARIANT_WAVE32 keeps the column-by-column float32 wave kernels (cmps_rho_wave.hip).  B = 9 leaves a partly
    filled last workgroup.  Loss, gradients and the saved states must agree to float32 tolerance."""
    from audio_mps_amd import RhoCMPS, _capi
    from audio_mps_amd.scan import HipScan
    m, audio = _rho_model(D, T, B, rank=rank, sigma=0.2, seed=13, rscale=0.5)
    colk = RhoCMPS(m.hparams, data_iterator=audio, seed=13, backend=HipScan(D, variant=_capi.CMPS_VARIANT_WAVE32))
    for k in m.variables:
        colk.variables[k] = m.variables[k].copy()
    a, b = m.loss_per_clip(), colk.loss_per_clip()
    assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) <= LOSS_RTOL
    la, ga = m.loss_and_grads()
    lb, gb = colk.loss_and_grads()
    assert abs(float(la) - float(lb)) <= LOSS_RTOL * max(abs(float(lb)), 1.0)
    for k in ga:
        assert rel_inf(ga[k], gb[k]) <= GRAD_RTOL, k
    assert rel_inf(m.rho_evolve_with_data(), colk.rho_evolve_with_data()) <= 1e-5
    # a second backward on fresh forward state gives the same gradients (the forward's P1 buffer is rewritten, not accumulated)
    _, ga2 = m.loss_and_grads()
    for k in ga:
        np.testing.assert_array_equal(ga[k], ga2[k])


@pytest.mark.parametrize("amp,rscale,sigma", [(1.0, 0.5, 0.2), (15.0, 1.0, 0.5), (4.0, 2.0, 0.5), (1e-3, 1e-3, 1e-4)])
def test_rho_gemm_forward_fp16_and_bf16_operands_agree(amp, rscale, sigma):
    """The GEMM forward runs with fp16 x 2 operands by default (scales per clip from |W_Q|, max |s| |W_R|, tr rho = 1) and with bf16 x 3
    under CMPS_RANK1_BF16X3: same loss and gradients to float32 tolerance, also for loud clips with a large R (W_k far from the
    identity, scales 2^-4 ... 2^20) and for a nearly silent model (tiny operands: the scales clamp, the low pieces go subnormal)."""
    from audio_mps_amd import RhoCMPS, _capi
    from audio_mps_amd.scan import HipScan
    D, rank, T, B = 32, 19, 200, 5
    m, audio = _rho_model(D, T, B, rank=rank, sigma=sigma, seed=23, rscale=rscale)
    audio = (audio * np.float32(amp)).astype(np.float32)
    m = RhoCMPS(m.hparams, data_iterator=audio, seed=23)
    m.variables["Rx"] *= np.float32(rscale)
    m.variables["Ry"] *= np.float32(rscale)
    ref = RhoCMPS(m.hparams, data_iterator=audio, seed=23, backend=HipScan(D, rank1=_capi.CMPS_RANK1_BF16X3))
    for k in m.variables:
        ref.variables[k] = m.variables[k].copy()
    a, b = m.loss_per_clip(), ref.loss_per_clip()
    assert np.all(np.isfinite(b)), "the case itself diverges (1 + z < 0): pick a quieter one"
    assert np.all(np.isfinite(a)) and np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) <= LOSS_RTOL
    la, ga = m.loss_and_grads()
    lb, gb = ref.loss_and_grads()
    assert abs(float(la) - float(lb)) <= LOSS_RTOL * max(abs(float(lb)), 1.0)
    for k in ga:
        assert np.all(np.isfinite(ga[k])) and rel_inf(ga[k], gb[k]) <= GRAD_RTOL, k


@pytest.mark.parametrize("D,rank,length,n", [(32, 32, 150, 5), (20, 9, 130, 3), (32, 1, 70, 2)])
def test_rho_gemm_sampler_matches_block_sampler(D, rank, length, n):
    """D <= 32 samples with the row-array GEMM kernel (one wavefront per path, cmps_rho_mfma.hip); CMPS_VARIANT_BLOCK keeps the
    workgroup-per-path kernel.  Same noise in, same waveform, states and purity out (float32 tolerance; n = 5 leaves a partly
    filled workgroup, length 130 / 150 a partial 64-step chunk)."""
    from audio_mps_amd import RhoCMPS
    from audio_mps_amd.scan import HipScan
    m, _ = _rho_model(D, 8, 2, rank=rank, sigma=0.1, seed=17, rscale=0.3, A=5.0, data=False)
    blk = RhoCMPS(m.hparams, seed=17, backend=HipScan(D, variant=1))
    for k in m.variables:
        blk.variables[k] = m.variables[k].copy()
    rng = np.random.default_rng(8)
    noise = (0.1 * np.sqrt(m.hparams.delta_t) * rng.standard_normal((length, n))).astype(np.float32)
    wa, wb = m.sample(n, length, noise=noise), blk.sample(n, length, noise=noise)
    scale = max(float(np.max(np.abs(wb))), 1e-6)
    assert np.max(np.abs(wa - wb)) <= 1e-4 * scale
    # the GEMM sampler's two arithmetics: fp16 x 2 pieces with fixed scales (a new handle) and bf16 x 3 (CMPS_RANK1_BF16X3)
    m3 = RhoCMPS(m.hparams, seed=17, backend=HipScan(D, rank1=2))
    for k in m.variables:
        m3.variables[k] = m.variables[k].copy()
    assert np.max(np.abs(m3.sample(n, length, noise=noise) - wb)) <= 1e-4 * scale
    assert rel_inf(m.rho_evolve_with_sampling(n, length, noise=noise), blk.rho_evolve_with_sampling(n, length, noise=noise)) <= 1e-4
    np.testing.assert_allclose(m.purity(n, length, noise=noise), blk.purity(n, length, noise=noise), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("D,rank", [(8, 3), (24, 12), (40, 5)])      # column wave kernels, row-array GEMM kernels, block kernels
def test_rho_entries_after_set_params_dev(D, rank):
    """cmps_set_params_dev keeps A in device memory (Dev::Adev) and leaves Dev::A unset; the rho kernels used to read
    Dev::A and divided by zero.  Every rho kernel family must give the same losses and gradient sums after cmps_set_params_dev
    as after cmps_set_params with the same numbers."""
    import torch
    m, audio = _rho_model(D, 60, 3, rank=rank, rscale=0.3 if D > 32 else None)
    be = m._get_backend()
    p = m.effective_params()
    dev_audio = m._to_device(audio)
    be.set_params(p, 3, 60, train=True)
    be.rho_set_state(m.columns(), 3, 60, train=True)
    ref_loss, ref_grad = (t.cpu().numpy().copy() for t in be.rho_loss_and_grad_sums(dev_audio))
    R = np.asarray(p.R)
    flat = np.concatenate([R.real.ravel(), R.imag.ravel(), np.asarray(p.freqs), np.asarray(p.psi0).real, np.asarray(p.psi0).imag,
                           [float(p.A)]]).astype(np.float32)
    be.set_params_dev(torch.from_numpy(flat).to(be.device), p.sigma, p.delta_t, 3, 60, train=True)
    be.rho_set_state(m.columns(), 3, 60, train=True)
    loss, grad = (t.cpu().numpy() for t in be.rho_loss_and_grad_sums(dev_audio))
    assert np.all(np.isfinite(loss)) and np.all(np.isfinite(grad))
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(grad, ref_grad, rtol=1e-5, atol=1e-6 * np.max(np.abs(ref_grad)))
