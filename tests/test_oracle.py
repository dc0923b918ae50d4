"""CPU tests of the oracle itself: the reference's own invariants (tests/test_model.py, tests/test_data.py
of /root/reference) restated on the restatement, the two oracle implementations against each other, the
float64 twin, finite-difference gradients, and the committed golden fixtures."""
import math
import os

import numpy as np
import pytest

from oracle import cmps_oracle as O
from oracle import c_oracle as C
from _util import golden_names, load_golden, golden_hparams, golden_oracle_variables, rel_inf, make_audio

# tests/test_model.py:13-14
TEST_HP = O.HParams(minibatch_size=8, bond_dim=7, delta_t=1 / 16000, sigma=0.0001, initial_rank=None, A=100.0,
                    h_reg=2 / (math.pi * 16000) ** 2, r_reg=2 / (math.pi * 16000))
SAMPLE_DURATION = 2 ** 8    # tests/test_model.py:9


def test_R_has_no_diagonal_elements():
    """TestCMPS.testRHasNoDiagonalElements (tests/test_model.py:19-25)."""
    var = O.init_variables(TEST_HP, seed=0)
    R, _, _, _ = O.effective_params(TEST_HP, var)
    np.testing.assert_allclose(np.diagonal(R), np.zeros(TEST_HP.bond_dim), atol=1e-6)
    # the quirk of model.py:42: every column j is shifted by -Z[j, j]
    Z = (O._rsqrt(TEST_HP.r_reg, np.float32) * (var.Rx + 1j * var.Ry)).astype(np.complex64)
    np.testing.assert_allclose(R, Z - np.diagonal(Z)[None, :], rtol=1e-6)


def test_data_shape():
    """TestGetAudio.testCorrectShape (tests/test_data.py:12-16)."""
    d = O.damped_sine(TEST_HP.minibatch_size, SAMPLE_DURATION, TEST_HP.delta_t, seed=0)
    assert d.shape == (TEST_HP.minibatch_size, SAMPLE_DURATION) and d.dtype == np.float32


def test_loss_not_nan():
    """TestPsiCMPS.testLossNotNaN (tests/test_model.py:107-113)."""
    var = O.init_variables(TEST_HP, seed=0)
    data = O.damped_sine(TEST_HP.minibatch_size, SAMPLE_DURATION, TEST_HP.delta_t, seed=1)
    assert not np.isnan(O.psi_loss(TEST_HP, var, data))


def test_psi_evolved_with_data_remains_normalized():
    """TestPsiCMPS.testPsiEvolvedWithDataRemainsNormalized (tests/test_model.py:115-122)."""
    var = O.init_variables(TEST_HP, seed=0)
    data = O.damped_sine(TEST_HP.minibatch_size, SAMPLE_DURATION, TEST_HP.delta_t, seed=1)
    _, states = O.psi_loss_per_clip(TEST_HP, var, data, return_states=True)
    assert states.shape == (TEST_HP.minibatch_size, SAMPLE_DURATION - 1, TEST_HP.bond_dim)
    np.testing.assert_allclose(np.linalg.norm(states, axis=-1), np.ones(states.shape[:2]), rtol=1e-5)


def test_trivial_update_of_ancilla():
    """TestPsiCMPS.testTrivialUpdateOfAncilla (tests/test_model.py:124-138): H = R = 0 leaves psi unchanged."""
    D, B = TEST_HP.bond_dim, TEST_HP.minibatch_size
    var = O.init_variables(TEST_HP, seed=0, R_in=np.zeros((D, D), np.complex64), freqs_in=np.zeros(D, np.float32))
    R, f, _, _ = O.effective_params(TEST_HP, var)
    psi0 = np.tile(O.psi_0(var)[None], (B, 1))
    signal = np.random.default_rng(0).random(B).astype(np.float32)
    out = O.update_ancilla_psi(psi0, signal, 0.0, R, f, var.A, TEST_HP)
    np.testing.assert_allclose(out, psi0, rtol=1e-6)


@pytest.mark.parametrize("D,T,B,sigma", [(4, 64, 3, 1e-4), (7, 100, 4, 1.0), (16, 80, 2, 1e-4)])
def test_c_oracle_matches_numpy_oracle(D, T, B, sigma):
    hp = O.HParams(minibatch_size=B, bond_dim=D, sigma=sigma)
    var = O.init_variables(hp, seed=D)
    if sigma == 1.0:
        var.Rx *= 0.1
        var.Ry *= 0.1
    data = O.damped_sine(B, T, hp.delta_t, seed=D + 1)
    for dtype, tol in (("f32", 2e-5), ("f64", 1e-9)):
        v = var if dtype == "f32" else var.astype(np.float64)
        g = O.psi_loss_and_grads(hp, v, data, dtype)
        R, f, _, _ = O.effective_params(hp, v, dtype)
        c = C.psi_scan(data, R, f, O.psi_0(v, dtype), v.A, hp.delta_t, hp.sigma, dtype, want_grad=True, want_states=True)
        scale = max(1.0, np.abs(g.per_clip).max())
        assert np.abs(c["loss_per_clip"] - g.per_clip).max() / scale < tol
        cg = C.unpack_grad(c["grad"], D)
        for k in ("Rbar", "fbar", "psi0bar"):
            assert rel_inf(np.asarray(cg[k]) / B, g.eff[k]) < 20 * tol, k
        _, states = O.psi_loss_per_clip(hp, v, data, dtype, return_states=True)
        assert np.abs(c["states"] - states).max() < 50 * tol


def test_float64_gradients_match_finite_differences():
    hp = O.HParams(minibatch_size=2, bond_dim=3, sigma=0.7, A=3.0)
    var = O.init_variables(hp, seed=1)
    var.Rx *= 0.3
    var.Ry *= 0.3
    data = (O.damped_sine(2, 21, hp.delta_t, seed=2)
            + 0.05 * np.random.default_rng(5).standard_normal((2, 21))).astype(np.float32)
    v64 = var.astype(np.float64)
    g = O.psi_loss_and_grads(hp, v64, data, "f64", with_reg=True)
    f = lambda v: float(O.total_loss(hp, v, data, "f64"))
    for name in O.Variables.NAMES:
        arr = getattr(v64, name)
        ga = np.asarray(getattr(g, name))
        for idx in np.ndindex(arr.shape):
            h = 1e-6 * max(1.0, abs(float(arr[idx])))
            vp, vm = v64.copy(), v64.copy()
            getattr(vp, name)[idx] += h
            getattr(vm, name)[idx] -= h
            fd = (f(vp) - f(vm)) / (2 * h)
            assert abs(fd - float(ga[idx])) <= 1e-6 * max(1e-3, abs(fd)), (name, idx, fd, float(ga[idx]))


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_golden(name):
    """The committed vectors are what the oracle computes today (numpy / libm differences allowed for)."""
    g = load_golden(name)
    hp = golden_hparams(g, O.HParams)
    var = golden_oracle_variables(g)
    out = O.psi_loss_and_grads(hp, var, g["data"], "f32")
    scale = np.maximum(np.abs(g["loss_per_clip_f32"]), 1.0)
    assert np.max(np.abs(out.per_clip - g["loss_per_clip_f32"]) / scale) < 2e-6
    for k in O.Variables.NAMES:
        assert rel_inf(getattr(out, k), g[f"grad_{k}_f32"]) < 1e-4, k
    # the C restatement against the same vectors
    R, f, _, _ = O.effective_params(hp, var)
    c = C.psi_scan(g["data"], R, f, O.psi_0(var), var.A, hp.delta_t, hp.sigma, "f32", want_grad=True)
    assert np.max(np.abs(c["loss_per_clip"] - g["loss_per_clip_f32"]) / scale) < 2e-5
    cg = C.unpack_grad(c["grad"], hp.bond_dim)
    B = g["data"].shape[0]
    assert rel_inf(cg["Rbar"] / B, g["eff_Rbar_f32"]) < 1e-4
    assert rel_inf(cg["fbar"] / B, g["eff_fbar_f32"]) < 1e-4
    # float64 twin stays close (it is a sanity bound, not the parity target)
    assert np.max(np.abs(g["loss_per_clip_f64"] - g["loss_per_clip_f32"]) / scale) < 1e-3


def test_sampling_shape_and_twin():
    """TestPsiCMPS.testSampling (tests/test_model.py:140-158): the two-level system; shape (2, 512)."""
    hp = O.HParams(minibatch_size=8, bond_dim=2, delta_t=1 / 16000, sigma=1, A=1.0,
                   h_reg=2 / (math.pi * 16000) ** 2, r_reg=2 / (math.pi * 16000) ** 2)
    var = O.init_variables(hp, seed=0, R_in=np.array([[0, 1], [0, 0]], dtype=np.complex64),
                           freqs_in=np.array([10.0, -10.0], dtype=np.float32))
    noise = O.sample_noise(hp, 2, 512, seed=1)
    w, states = O.psi_sample(hp, var, noise, return_states=True)
    assert w.shape == (2, 512) and np.all(np.isfinite(w))
    np.testing.assert_allclose(np.linalg.norm(states, axis=-1), 1.0, rtol=1e-5)
    w64 = O.psi_sample(hp, var.astype(np.float64), noise, "f64")
    assert np.max(np.abs(w - w64)) < 1e-5


def test_legacy_audiomps_gradients_match_finite_differences():
    D, dt = 3, 0.01
    H, R = O.legacy_init(D, 1)
    data = (O.damped_sine(2, 25, dt, seed=2) + 0.05 * np.random.default_rng(3).standard_normal((2, 25))).astype(np.float32)
    H64, R64 = H.astype(np.float64), R.astype(np.float64)
    out = O.legacy_loss_and_grads(H64, R64, dt, data, "f64")
    for name in ("H", "R"):
        for idx in np.ndindex(D, D):
            hp_, hm_ = H64.copy(), H64.copy()
            rp_, rm_ = R64.copy(), R64.copy()
            (hp_ if name == "H" else rp_)[idx] += 1e-6
            (hm_ if name == "H" else rm_)[idx] -= 1e-6
            fd = (float(O.legacy_loss_and_grads(hp_, rp_, dt, data, "f64")["loss"])
                  - float(O.legacy_loss_and_grads(hm_, rm_, dt, data, "f64")["loss"])) / 2e-6
            an = out["gH" if name == "H" else "gR"][idx]
            assert abs(fd - an) <= 1e-6 * max(1.0, abs(fd)), (name, idx, fd, an)


def test_bf16_emulation_stays_close_to_float32():
    """oracle.psi_bf16_scan restates the arithmetic of the D = 128 MFMA kernels (bf16 mat-vec operands, float32
    accumulation, rotating frame, analytic adjoint).  It must stay within the documented bf16 distance of the float32
    restatement -- which also validates its hand-written reverse sweep."""
    from oracle import c_oracle as C
    hp = O.HParams(minibatch_size=3, bond_dim=24)
    var = O.init_variables(hp, seed=2)
    data = make_audio(3, 150, hp.delta_t, 3)
    em = O.psi_bf16_scan(hp, var, data)
    R, f, _, _ = O.effective_params(hp, var)
    ref = C.psi_scan(data, R, f, O.psi_0(var), float(var.A), hp.delta_t, hp.sigma, "f32", want_grad=True)
    gr = C.unpack_grad(ref["grad"], 24)
    den = np.maximum(np.abs(ref["loss_per_clip"]), 1.0)
    assert np.max(np.abs(em["loss_per_clip"] - ref["loss_per_clip"]) / den) <= 2e-3
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        assert rel_inf(em[k], gr[k]) <= 3e-2, k
    # and the rounding helper is round-to-nearest-even on the top 16 bits
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.0e-39], dtype=np.float32)
    np.testing.assert_array_equal(O.bf16_round(x), np.array([1.0, 1.0, 1.015625, -2.5, O.bf16_round(x[4:5])[0]], dtype=np.float32))


# ---------------------------------------------------------------------------------------------------
# the only reference-held NUMBERS on the path: the constants of logging/graph.pbtxt (legacy AudioMPS graph)
# ---------------------------------------------------------------------------------------------------
def _graph_constants():
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "legacy_graph_constants.json")
    with open(path) as fh:
        return {k: v["value"] for k, v in json.load(fh)["constants"].items()}


def test_legacy_oracle_uses_the_graphs_constants():
    """tests/golden/legacy_graph_constants.json is extracted from /root/reference/logging/graph.pbtxt by
    scripts/graph_constants.py (committed).  A two-step scan evaluated from those numbers ALONE (float64, written here, none
    of the oracle's code) must agree with oracle/cmps_oracle.py::legacy_loss_and_grads, whose own constants are literals."""
    c = _graph_constants()
    assert c["loop_bound"] == 4095                                   # T - 1 for the notebook's T = 4096
    assert c["hamiltonian_factor"] == {"re": -0.0, "im": -1.0}
    dt = c["delta_t_Q"]["re"]
    assert dt == c["delta_t_signal"]["re"] == float(np.float32(0.01))
    D, B, T = 5, 8, 3                                                # the graph's D and batch; two steps
    H, R = O.legacy_init(D, seed=3)
    data = make_audio(B, T, dt, 1, noise=0.3)
    ref = O.legacy_loss_and_grads(H, R, dt, data, "f64")
    H64, R64 = H.astype(np.float64), R.astype(np.float64)
    Hs = np.tril(H64) + np.tril(H64).T                               # Appendix A: band_part(H,-1,0) + its transpose
    fac = complex(c["hamiltonian_factor"]["re"], c["hamiltonian_factor"]["im"])
    Q = dt * (fac * Hs - (R64.T @ R64) / c["dissipator_divisor"]["re"])
    per = np.zeros(B)
    for b in range(B):
        psi = np.zeros(D, complex)
        psi[0] = 1.0
        for k in range(T - 1):
            x = float(data[b, k + 1]) - float(data[b, k])
            e = c["expectation_factor"] * np.real(np.vdot(psi, R64 @ psi))
            per[b] += (x - e) ** c["loss_exponent"] / c["loss_divisor"]
            y = psi + Q @ psi + c["delta_t_signal"]["re"] * x * (R64 @ psi)
            psi = y / np.sqrt(max(np.sum(np.abs(y) ** 2), c["norm_floor"]))
    np.testing.assert_allclose(ref["per_clip"], per, rtol=1e-12)
    # a wrong constant is visible at this tolerance: e.g. factor 2 -> 1 or divisor 2 -> 1 changes the loss by O(1)
    assert abs(np.mean(per) - float(ref["loss"])) < 1e-12


def test_adam_defaults_are_the_graphs():
    """tf.train.AdamOptimizer as serialized in the legacy graph (learning rate 1e-3, beta1 0.9, beta2 0.999, eps 1e-8)."""
    from audio_mps_amd.train import AdamOptimizer
    c = _graph_constants()
    opt = AdamOptimizer()
    for mine, key in ((opt.lr, "adam_learning_rate"), (opt.b1, "adam_beta1"), (opt.b2, "adam_beta2"), (opt.eps, "adam_epsilon")):
        assert np.float32(mine) == np.float32(c[key]), key
