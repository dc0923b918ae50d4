"""CPU tests of the host-side mirror (audio_mps_amd.model / train): hyper-parameters, effective parameters,
the chain rule from kernel outputs to raw-variable gradients, Adam, the trainer, checkpoint/resume.
The scan itself is supplied by the oracle-backed stand-in (tests/_util.OracleBackend): these tests exercise
the host logic only -- the HIP path has its own parity tests (-m gpu)."""
import math
import os

import numpy as np
import pytest

from audio_mps_amd import HParams, CMPS, PsiCMPS, AudioMPS, RhoCMPS
from audio_mps_amd.train import AdamOptimizer, Trainer
from audio_mps_amd.data import get_audio
from oracle import cmps_oracle as O
from _util import (OracleBackend, golden_names, load_golden, model_from_golden, oracle_hparams, oracle_variables,
                   rel_inf, make_audio)


def test_hparams_defaults_and_parse():
    hp = HParams()                                           # train.py:41-43
    assert hp.minibatch_size == 8 and hp.bond_dim == 8 and hp.A == 100.0 and hp.sigma == 1e-4
    assert hp.r_reg == 0.1 and hp.learning_rate == 1e-3 and hp.initial_rank is None
    assert hp.h_reg == pytest.approx(200 / (math.pi * 16000) ** 2)
    hp.parse("bond_dim=32,minibatch_size=64,learning_rate=0.3")   # train.py:31,44
    assert (hp.bond_dim, hp.minibatch_size, hp.learning_rate) == (32, 64, 0.3)
    with pytest.raises(ValueError):
        HParams().parse("no_such=1")


def test_effective_parameters_match_oracle():
    hp = HParams(bond_dim=7, r_reg=2 / (math.pi * 16000), h_reg=2 / (math.pi * 16000) ** 2)
    m = PsiCMPS(hp, seed=3, backend=OracleBackend(7))
    R, f, _, _ = O.effective_params(oracle_hparams(hp), oracle_variables(m))
    np.testing.assert_array_equal(m.R, R)
    np.testing.assert_array_equal(m.freqs, f)
    np.testing.assert_array_equal(m.psi_0, O.psi_0(oracle_variables(m)))
    np.testing.assert_allclose(np.diagonal(m.R), 0, atol=1e-6)      # tests/test_model.py:19-25
    assert abs(np.linalg.norm(m.psi_0) - 1) < 1e-6


def test_R_in_freqs_in_branches():
    D = 2
    hp = HParams(bond_dim=D, sigma=1.0, A=1.0)
    R = np.array([[0, 1], [0, 0]], dtype=np.complex64)              # tests/test_model.py:147-151
    f = np.array([10.0, -10.0], dtype=np.float32)
    m = PsiCMPS(hp, R_in=R, freqs_in=f, backend=OracleBackend(D))
    np.testing.assert_array_equal(m.R, R - np.diagonal(R)[None, :])
    np.testing.assert_array_equal(m.freqs, f)
    with pytest.raises(ValueError):
        CMPS(hp, R_in=np.zeros((3, 3)))


@pytest.mark.parametrize("name", golden_names())
def test_loss_and_chain_rule_match_golden(name):
    """PsiCMPS.loss / loss_and_grads through the host chain rule == the oracle's own end-to-end gradients."""
    g = load_golden(name)
    m = model_from_golden(g, backend=OracleBackend(int(g["hp_bond_dim"])))
    loss, grads = m.loss_and_grads()
    assert abs(float(loss) - float(g["loss_f32"])) <= 2e-5 * max(1.0, abs(float(g["loss_f32"])))
    for k in O.Variables.NAMES:
        assert rel_inf(grads[k], g[f"grad_{k}_f32"]) < 2e-4, k
    assert abs(float(m.loss) - float(g["loss_f32"])) <= 2e-5 * max(1.0, abs(float(g["loss_f32"])))


def test_regularised_gradients_match_oracle():
    hp = HParams(minibatch_size=3, bond_dim=5, sigma=0.7, A=3.0)
    m = PsiCMPS(hp, seed=1, backend=OracleBackend(5))
    m.variables["Rx"] *= np.float32(0.3)
    m.variables["Ry"] *= np.float32(0.3)
    data = make_audio(3, 60, hp.delta_t, 2, noise=0.05)
    total, grads = m.loss_and_grads(data, with_reg=True)
    ref = O.psi_loss_and_grads(oracle_hparams(hp), oracle_variables(m).astype(np.float64), data, "f64", with_reg=True)
    assert abs(float(total) - float(ref.loss)) < 1e-4 * abs(float(ref.loss))
    for k in O.Variables.NAMES:
        assert rel_inf(grads[k], getattr(ref, k)) < 1e-3, k


def test_adam_matches_formula():
    opt = AdamOptimizer(learning_rate=1e-3)
    v = {k: np.ones(3, np.float32) for k in ("A", "Rx", "Ry", "freqs", "psi_x", "psi_y")}
    g = {k: np.full(3, 0.5, np.float32) for k in v}
    opt.apply_gradients(v, g)
    # t = 1: m = 0.1 g, v = 0.001 g^2, lr_t = lr sqrt(1-b2)/(1-b1) -> step = lr * g / (|g| + eps*...) ~ lr
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    expect = 1 - lr_t * (0.1 * 0.5) / (math.sqrt(0.001 * 0.25) + 1e-8)
    np.testing.assert_allclose(v["Rx"], expect, rtol=1e-6)


def test_trainer_decreases_loss_and_checkpoints(tmp_path):
    hp = HParams(minibatch_size=4, bond_dim=4, learning_rate=0.01)
    data = make_audio(4, 80, hp.delta_t, 3)
    m = PsiCMPS(hp, data_iterator=data, seed=0, backend=OracleBackend(4))
    tr = Trainer(m, hp)
    first = tr.step()["total_loss"]
    for _ in range(5):
        last = tr.step()["total_loss"]
    assert np.isfinite(last) and last < first
    path = os.path.join(tmp_path, "ckpt", "model.ckpt.npz")
    tr.save(path)
    m2 = PsiCMPS(hp, data_iterator=data, seed=9, backend=OracleBackend(4))
    tr2 = Trainer(m2, hp)
    assert tr2.restore(path) and tr2.global_step == tr.global_step
    for k in m.variables:
        np.testing.assert_array_equal(m.variables[k], m2.variables[k])
    assert tr2.step()["total_loss"] == pytest.approx(tr.step()["total_loss"], rel=1e-6)


def test_audiomps_surface():
    """training_estimators.py:43-45: AudioMPS(bond_d, dt, batch_size, data_iterator=data, mixed=discr).loss"""
    data = make_audio(4, 64, 0.001, 1)
    a = AudioMPS(4, 0.001, 4, data_iterator=data, mixed=False, backend=OracleBackend(4))
    assert a.bond_d == 4 and a.delta_t == 0.001 and np.isfinite(a.loss)
    r = AudioMPS(4, 0.001, 4, data_iterator=data, mixed=True, backend=OracleBackend(4))
    assert isinstance(r, RhoCMPS) and r.bond_d == 4 and r.rank_rho_0 == 4 and np.isfinite(r.loss)


def test_rho0_is_a_density_matrix():
    """tests/test_model.py:41-48 (testRho0IsADensityMatrix) plus the column form handed to the scan."""
    m = RhoCMPS(HParams(bond_dim=7, initial_rank=3), seed=2)
    r0 = m.rho_0
    np.testing.assert_allclose(r0, r0 / np.trace(r0), rtol=1e-6)
    np.testing.assert_allclose(r0, np.conj(r0.T), atol=1e-7)
    phi = m.columns()
    assert phi.shape == (3, 7)
    np.testing.assert_allclose(np.einsum("ai,aj->ij", phi, np.conj(phi)), r0, atol=1e-6)


@pytest.mark.parametrize("rank", [None, 2])
def test_rho_chain_rule_matches_oracle(rank):
    """RhoCMPS.loss_and_grads (host chain rule over the cmps_rho_loss_bwd layout) against the oracle's direct
    variable gradients, with the oracle standing in for the kernels."""
    hp = HParams(minibatch_size=3, bond_dim=4, sigma=0.5, A=2.0, initial_rank=rank, r_reg=0.3, h_reg=1e-4)
    data = make_audio(3, 24, hp.delta_t, 5, noise=0.05)
    m = RhoCMPS(hp, data_iterator=data, seed=4, backend=OracleBackend(4, "f64"))
    loss, grads = m.loss_and_grads()
    ov = O.Variables(np.asarray(m.variables["A"]), m.variables["Rx"], m.variables["Ry"], m.variables["freqs"],
                     np.zeros(4, np.float32), np.zeros(4, np.float32), scaled_R=True, scaled_freqs=True)
    ohp = O.HParams(**hp.values())
    ref = O.rho_loss_and_grads(ohp, ov.astype(np.float64), m.variables["Wx"].astype(np.float64),
                               m.variables["Wy"].astype(np.float64), data, "f64")
    assert float(loss) == pytest.approx(float(ref["loss"]), rel=1e-5)
    for k in ("A", "Rx", "Ry", "freqs", "Wx", "Wy"):
        assert rel_inf(grads[k], ref[k]) < 2e-5, k


def test_get_audio():
    hp = HParams(minibatch_size=8)
    d = get_audio(None, "damped_sine", hp, sample_duration=2 ** 8)   # tests/test_data.py:12-16
    assert d.shape == (8, 256) and d.dtype == np.float32
    with pytest.raises(FileNotFoundError):
        get_audio("./data", "guitar", hp)


def test_tfrecord_pipeline(tmp_path):
    """data.py:25-43 / training_estimators.py:76-95 on a small file written like make-small-dataset.py:18-32."""
    from audio_mps_amd import tfrecord as tfr
    assert tfr.crc32c(b"123456789") == 0xE3069283                  # CRC-32C check value
    rng = np.random.default_rng(0)
    clips = rng.standard_normal((10, 64)).astype(np.float32)
    path = os.path.join(tmp_path, "guitar.tfrecords")
    tfr.write_audio_tfrecord(path, clips)
    recs = [tfr.parse_example(r)["audio"] for r in tfr.read_records(path, verify=True)]
    np.testing.assert_array_equal(np.stack(recs), clips)
    hp = HParams(minibatch_size=4)
    nxt = get_audio(str(tmp_path), "guitar", hp, sample_duration=64, seed=1)     # batch(4) -> shuffle(24) -> repeat
    seen = [nxt() for _ in range(6)]                               # two epochs of 3 batches (4 + 4 + 2 clips)
    assert sorted(b.shape[0] for b in seen[:3]) == [2, 4, 4]
    first = np.concatenate(seen[:3])
    assert sorted(map(tuple, first.round(4).tolist())) == sorted(map(tuple, clips.round(4).tolist()))
    est = tfr.audio_batches(path, 4, 64, seed=2, order="estimator")  # shuffle(24).repeat().batch(4)
    b = [est() for _ in range(5)]
    assert all(x.shape == (4, 64) for x in b)
    with pytest.raises(ValueError):                                # FixedLenFeature([T]) rejects other lengths (data.py:32)
        get_audio(str(tmp_path), "guitar", hp, sample_duration=32)()
    # corruption is detected when verification is on
    raw = bytearray(open(path, "rb").read())
    raw[40] ^= 0xFF
    bad = os.path.join(tmp_path, "bad.tfrecords")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        list(tfr.read_records(bad, verify=True))


def test_tfrecord_known_answer_bytes(tmp_path):
    """One framed tf.train.Example assembled BY HAND from the wire specification (not by tfrecord.write_audio_tfrecord):
    audio = [0.5, -1.25, 3.0] -> FloatList{1: packed} 0A 0C <12 bytes little-endian>; Feature{2: float_list} 12 0E ...;
    map entry {1: "audio", 2: Feature} 0A 05 'audio' 12 10 ...; Features{1: entry} 0A 19 ...; Example{1: features} 0A 1B ...
    (data.py:31-32, make-small-dataset.py:24-32); framing uint64 length 0x1D | masked CRC | payload | masked CRC with
    CRC-32C computed by a bit-at-a-time loop (polynomial 0x82F63B78; crc(length) = 0x224440AE, crc(payload) = 0x21628FEF)
    and mask ((crc >> 15 | crc << 17) + 0xA282EAD8)."""
    from audio_mps_amd import tfrecord as tfr
    rec = bytes.fromhex("1d00000000000000" "602fdf23"
                        "0a1b0a190a05617564696f1210120e0a0c0000003f0000a0bf00004040" "9d2d61c2")
    payload = rec[12:-4]
    assert tfr.crc32c(rec[:8]) == 0x224440AE and tfr.crc32c(payload) == 0x21628FEF
    assert tfr._crc32c_python(payload) == 0x21628FEF
    path = os.path.join(tmp_path, "one.tfrecords")
    open(path, "wb").write(rec + rec)
    got = [tfr.parse_example(r)["audio"] for r in tfr.read_records(path, verify=True)]
    assert len(got) == 2
    np.testing.assert_array_equal(got[0], np.array([0.5, -1.25, 3.0], np.float32))
    # the writer produces exactly these bytes
    out = os.path.join(tmp_path, "w.tfrecords")
    tfr.write_audio_tfrecord(out, [np.array([0.5, -1.25, 3.0], np.float32)])
    assert open(out, "rb").read() == rec
    # the reference's pipeline on it: FixedLenFeature([3]) -> batch(2)
    nxt = get_audio(str(tmp_path), "one", HParams(minibatch_size=2), sample_duration=3)
    np.testing.assert_array_equal(nxt(), np.tile(np.array([0.5, -1.25, 3.0], np.float32), (2, 1)))


def test_train_main_iterates_a_tfrecord_dataset(tmp_path):
    """train.py:46-47 builds the input ONCE and every session.run pulls the next batch.  main() on a TFRecord dataset (host
    logic only: the scan is the oracle stand-in) must see different batches in successive steps, cover the epoch, and
    write its checkpoint (it used to index the batch callable)."""
    from audio_mps_amd import tfrecord as tfr
    from audio_mps_amd import train
    clips = make_audio(8, 64, 1 / 16000, 3)
    tfr.write_audio_tfrecord(os.path.join(tmp_path, "organ.tfrecords"), clips)
    seen = []

    class Spy(OracleBackend):
        def loss_and_grad_sums(self, audio, check=False):
            seen.append(np.asarray(audio).copy())
            return super().loss_and_grad_sums(audio)

    tr = train.main(["--dataset", "organ", "--datadir", str(tmp_path), "--sample_duration", "64", "--max_steps", "4",
                     "--hparams", "bond_dim=4,minibatch_size=4", "--logdir", os.path.join(tmp_path, "log")], backend=Spy(4))
    assert tr.global_step == 4 and len(seen) == 4 and all(s.shape == (4, 64) for s in seen)
    epoch = np.concatenate(seen[:2])
    assert sorted(map(tuple, epoch.round(5).tolist())) == sorted(map(tuple, clips.round(5).tolist()))
    assert all(np.isfinite(h["model_loss"]) for h in tr.history)
    assert os.path.exists(os.path.join(tmp_path, "log", "organ", f"4_{tr.hparams.delta_t}_4", "model.ckpt.npz"))
    # rho at the reference's default rank = D beyond the old LDS limit (rank * D = 6400 > 5000) is accepted now (round 3)
    tr2 = train.main(["--mps_model", "rho_mps", "--hparams", "bond_dim=80,minibatch_size=2", "--max_steps", "1",
                      "--sample_duration", "12", "--logdir", os.path.join(tmp_path, "log2")], backend=OracleBackend(80))
    assert tr2.global_step == 1 and np.isfinite(tr2.history[-1]["model_loss"])


def test_product_has_no_cpu_fallback():
    """Without a GPU the product must refuse to compute rather than silently use something else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = PsiCMPS(HParams(bond_dim=4), data_iterator=make_audio(2, 16, 1 / 16000, 0))
    with pytest.raises(RuntimeError):
        _ = m.loss


def test_estimator_surface(tmp_path):
    """training_estimators.py: audiomps(...) -> model_fn (loss, Adam 1e-3) -> Estimator.train with checkpoints."""
    from audio_mps_amd import tfrecord as tfr
    from audio_mps_amd.training_estimators import Estimator, build_input_fns, build_parser
    args = build_parser().parse_args([])
    assert (args.bond_d, args.dt, args.batch_size, args.viz_steps, args.max_steps, args.discr) == (10, 0.001, 32, 1, 5001, False)
    clips = make_audio(12, 48, 0.001, 4)
    path = os.path.join(tmp_path, "pitch_30.tfrecords")
    tfr.write_audio_tfrecord(path, clips)
    input_fn = build_input_fns(path, 4, sample_duration=48, seed=0)
    params = dict(bond_d=4, dt=0.001, batch_size=4, discr=False)
    est = Estimator(params, model_dir=os.path.join(tmp_path, "m"), save_checkpoints_steps=2,
                    model_kw={"backend": OracleBackend(4), "seed": 0})
    est.train(input_fn, steps=4)
    assert est.global_step == 4 and np.isfinite(est.last_loss)
    est2 = Estimator(params, model_dir=os.path.join(tmp_path, "m"), model_kw={"backend": OracleBackend(4), "seed": 5})
    assert est2.global_step == 4
    for k in est.model.variables:
        np.testing.assert_array_equal(est.model.variables[k], est2.model.variables[k])


def test_crc32c_native_matches_python_and_known_answer():
    """TFRecord framing checksum: the C-ABI utility (cmps_crc32c) and the pure-Python fallback agree, and both give the
    CRC-32C check value of the standard test vector."""
    from audio_mps_amd import tfrecord as t
    assert t._crc32c_python(b"123456789") == 0xE3069283
    assert t.crc32c(b"123456789") == 0xE3069283
    data = np.random.default_rng(3).integers(0, 256, 4099, dtype=np.uint8).tobytes()
    assert t.crc32c(data) == t._crc32c_python(data)
    assert t.crc32c(b"") == 0


def test_bench_headline_is_compact(capsys, tmp_path, monkeypatch):
    """bench.py's LAST stdout line must fit the driver's record (BENCH_r04.json.parsed was null for a 23.5 KB line): the headline
    formatter keeps the contract's keys, drops the per-kernel tables, and never exceeds HEADLINE_LIMIT; details are separate
    {"detail": ...} lines without a top-level "metric"."""
    import json
    import bench
    monkeypatch.setattr(bench, "DETAIL_DIR", str(tmp_path / "detail"))
    ktimes = {"k_fwd_wave2": (13.5, 3), "k_bwd_wave": (18.6, 3), "k_reduce_slabs": (0.1, 3), "k_finalize": (0.02, 3)}
    roof = bench.roofline_record(32, 16000, 1024, bench.V_WAVE, 3, 4.5e-3, 6.2e-3, 10.8, ktimes)
    assert roof["bound"] == "valu-issue" and roof["kernels"]
    name = bench.emit_detail("roofline_detail", roof)
    printed = capsys.readouterr().out.strip().splitlines()
    assert name == "roofline_detail" and len(printed) == 1
    d = json.loads(printed[0])
    assert d["detail"] == "roofline_detail" and "metric" not in d and d["data"]["kernels"]
    assert (tmp_path / "detail" / "roofline_detail.json").exists()
    comp = bench.compact_roofline(roof)
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "algorithmic_bytes"):
        assert key in comp, key
    assert "kernels" not in comp and "executed" not in comp and "step_traffic" not in comp
    assert comp["frac_algorithmic"] == pytest.approx(56 * 32 * 32 * 1024 * 15999 / 6.2e-3 / 1e12 / 157.3, rel=1e-9)
    assert comp["frac"] == pytest.approx(12 * 32 * 32 * 1024 * 15999 / 6.2e-3 / 1e12 / 157.3, rel=1e-9)       # the executed VALU work: bounded by 1
    par = {"clips": 128, "max_rel_loss_err": 1e-6, "max_rel_loss_err_unfloored": 1e-6, "loss_err_note": "x" * 500, "max_rel_grad_err": 5e-5,
           "grad_err_by_tensor": {"Rbar": 1e-6}, "tolerance": {"loss": 1e-5, "grad": 1e-4}, "ok": True, "against": "oracle " * 40}
    row = {"config": "configs[4] in float32: D=128, T=16000, batch 512 (wide kernels)", "ms_per_step": 43.1234567, "value": 1.9e8,
           "dtype": bench.dtype_label(bench.V_WIDE, 128, 3, 1), "dominant_kernel": "k_bwd_chain16", "frac": 0.43, "kernels": [roof] * 5,
           "parity_in_bench": par}
    out = {"metric": "audio samples/sec (fwd+bwd) at D=32, T=16000", "value": 1.5e9, "unit": "samples/s", "n_gpus": 1, "steps": 20,
           "warmup": 5, "ms_per_step": 10.8, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": bench.dtype_label(bench.V_WAVE, 32, 3), "data": "synthetic",
           "config": {"workload": "w" * 250, "parallelism": "dp1", "arithmetic": bench.arithmetic_note(bench.V_WAVE, 32, 3)},
           "roofline": comp, "cpu_baseline": {"value": 1.7e6, "unit": "samples/s", "cores": 16, "kind": "port", "sample": "s" * 100},
           "parity_in_bench": bench.compact_parity(par),
           "other_configs": [bench.compact_other_config(row)] * 4 + [bench.compact_other_config({"config": "c", "error": "e" * 900})],
           "collective": {"settings": {"NCCL_PROTO": "LL"}}, "detail_lines": ["roofline_detail"] * 9}
    line = bench.headline_line(out)
    assert len(line) < 4096 and len(line) <= bench.HEADLINE_LIMIT
    back = json.loads(line)
    assert back["roofline"]["frac"] == pytest.approx(comp["frac"], rel=1e-5) and back["cpu_baseline"]["kind"] == "port" and len(back["other_configs"]) == 5
    # too long: optional keys go, the contract's keys stay; hopeless: an error, never a silent over-long line
    out["other_configs"] = [{"config": "c" * 300}] * 30
    back = json.loads(bench.headline_line(out))
    assert "other_configs" in back["dropped_for_size"] and "roofline" in back and "cpu_baseline" in back
    out["config"]["workload"] = "w" * 5000
    with pytest.raises(RuntimeError):
        bench.headline_line(out)
    # the wide family's labels follow CMPS_OPT_WIDE_CHAIN
    assert "matrix cores" in bench.dtype_label(bench.V_WIDE, 128, 3, 1) and bench.dtype_label(bench.V_WIDE, 128, 3, 0) == "f32"
    assert "VALU" in bench.arithmetic_note(bench.V_WIDE, 128, 3, 0) and "f16x2" in bench.arithmetic_note(bench.V_WIDE, 128, 3, 1)
    rw = bench.roofline_record(128, 16000, 512, bench.V_WIDE, 3, 18e-3, 25e-3, 43.0,
                               {"k_fwd_chain16": (27.0, 2), "k_bwd_chain16": (31.0, 2), "k_grad_gemm<f16x2>": (17.0, 2)}, 1)
    assert rw["kernel"] == "k_bwd_chain16" and rw["bound"] == "mfma" and "chain16" in bench.executed_split(bench.V_WIDE, 128, 3, 1)["bwd"]["what"]
