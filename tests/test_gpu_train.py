"""End-to-end on the GPU: the trainer (train.py mirror) driving the HIP scan, against the same trainer driven
by the oracle stand-in; checkpoint/resume; the smoke entry."""
import os

import numpy as np
import pytest

from _util import OracleBackend, make_audio

pytestmark = pytest.mark.gpu


def test_trainer_tracks_oracle_trainer(tmp_path):
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan
    from audio_mps_amd.train import Trainer
    hp = HParams(minibatch_size=8, bond_dim=16, learning_rate=0.01)
    data = make_audio(8, 400, hp.delta_t, 5)
    m_hip = PsiCMPS(hp, data_iterator=data, seed=0, backend=HipScan(16))
    m_ref = PsiCMPS(hp, data_iterator=data, seed=0, backend=OracleBackend(16))
    t_hip, t_ref = Trainer(m_hip, hp), Trainer(m_ref, hp)
    hist = []
    for _ in range(6):
        a, b = t_hip.step(), t_ref.step()
        hist.append((a["total_loss"], b["total_loss"]))
    hist = np.array(hist)
    assert np.all(np.isfinite(hist))
    assert hist[-1, 0] < hist[0, 0]                                   # Adam makes progress
    np.testing.assert_allclose(hist[:, 0], hist[:, 1], rtol=2e-4)     # same trajectory as the oracle-driven run
    path = os.path.join(tmp_path, "model.ckpt.npz")
    t_hip.save(path)
    m2 = PsiCMPS(hp, data_iterator=data, seed=1, backend=HipScan(16))
    t2 = Trainer(m2, hp)
    assert t2.restore(path)
    assert t2.step()["total_loss"] == pytest.approx(t_hip.step()["total_loss"], rel=1e-6)


@pytest.mark.parametrize("D,given", [(8, False), (16, False), (32, False), (12, True), (64, False)])
def test_device_step_matches_host_trainer(D, given):
    """cmps_psi_apply_step (chain rule of model.py:36-42, 49, 221-222 + regularisers train.py:55-60 + Adam train.py:89, all on the
    device, no D2H inside a step) against the host implementation of the same half (model.chain_rule + AdamOptimizer), both on the
    HIP scan: the same trajectory to 1e-6 over 20 steps, the same variables and Adam slots afterwards.  `given`: R_in / freqs_in
    supplied (no rsqrt(reg) scaling, model.py:31-33, 44-46)."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan
    from audio_mps_amd.train import Trainer
    hp = HParams(minibatch_size=6, bond_dim=D, learning_rate=0.01)
    data = make_audio(6, 260, hp.delta_t, 5)
    kw = {}
    if given:
        rng = np.random.default_rng(1)
        kw = {"R_in": (0.3 * (rng.standard_normal((D, D)) + 1j * rng.standard_normal((D, D)))).astype(np.complex64),
              "freqs_in": (100.0 * rng.standard_normal(D)).astype(np.float32)}
    m_dev = PsiCMPS(hp, data_iterator=data, seed=0, backend=HipScan(D), **kw)
    m_host = PsiCMPS(hp, data_iterator=data, seed=0, backend=HipScan(D), **kw)
    if D > 32:
        for m in (m_dev, m_host):
            m.variables["Rx"] *= np.float32(0.5)
            m.variables["Ry"] *= np.float32(0.5)
    t_dev, t_host = Trainer(m_dev, hp, device_step=True), Trainer(m_host, hp)
    hist = []
    for _ in range(20):
        a, b = t_dev.step(), t_host.step()
        hist.append([a["model_loss"], a["total_loss"], b["model_loss"], b["total_loss"]])
    hist = np.array(hist)
    assert np.all(np.isfinite(hist))
    np.testing.assert_allclose(hist[:, 0], hist[:, 2], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(hist[:, 1], hist[:, 3], rtol=1e-6, atol=1e-6)
    assert hist[-1, 1] < hist[0, 1]
    t_dev.sync_to_host()
    for k in ("A", "Rx", "Ry", "freqs", "psi_x", "psi_y"):
        np.testing.assert_allclose(m_dev.variables[k], m_host.variables[k], rtol=2e-5, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(t_dev.opt.m[k], t_host.opt.m[k], rtol=1e-3, atol=1e-9 + 1e-5 * np.max(np.abs(t_host.opt.m[k])), err_msg=k)
    # sync=False keeps everything on the device: the losses come back as a device tensor, the step count advances
    out = t_dev.step(sync=False)
    assert "losses_dev" in out and out["losses_dev"].is_cuda and out["global_step"] == 21
    # checkpoint of the device-resident state -> a host-path trainer continues the same trajectory
    assert float(out["losses_dev"].cpu()[1]) == pytest.approx(t_host.step()["total_loss"], rel=1e-6, abs=1e-6)


def test_empty_shard_contributes_zeros():
    """A rank whose shard of a short final batch is empty skips the scan and adds zeros to the all-reduce (both optimiser paths)."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan
    from audio_mps_amd.train import Trainer
    hp = HParams(minibatch_size=4, bond_dim=8, learning_rate=0.01)
    data = make_audio(4, 100, hp.delta_t, 1)
    for dev_step in (False, True):
        m = PsiCMPS(hp, data_iterator=data, seed=0, backend=HipScan(8))
        t = Trainer(m, hp, device_step=dev_step)
        t.step()
        before = {k: v.copy() for k, v in (t.sync_to_host() or m.variables).items()}
        out = t.step(data[:0], global_batch=4)                       # nothing local: the update comes from the regularisers alone
        t.sync_to_host()
        assert out["model_loss"] == 0.0 and np.isfinite(out["total_loss"])
        assert any(not np.array_equal(before[k], m.variables[k]) for k in before)


def test_train_main_on_tfrecords(tmp_path):
    """train.py on a TFRecord dataset (data.py:25-43: batch -> shuffle(24) -> repeat, built once, get_next per step):
    `python -m audio_mps_amd.train --dataset guitar` for 3 steps on the HIP scan, against the same main() driven by the
    oracle stand-in on the same file and seed (same batches in the same order)."""
    from audio_mps_amd import tfrecord as tfr
    from audio_mps_amd import train
    clips = make_audio(12, 300, 1 / 16000, 8)
    tfr.write_audio_tfrecord(os.path.join(tmp_path, "guitar.tfrecords"), clips)
    argv = ["--dataset", "guitar", "--datadir", str(tmp_path), "--sample_duration", "300", "--max_steps", "4",
            "--hparams", "bond_dim=8,minibatch_size=4,learning_rate=0.01", "--seed", "3"]
    t_hip = train.main(argv + ["--logdir", os.path.join(tmp_path, "hip")])
    t_ref = train.main(argv + ["--logdir", os.path.join(tmp_path, "ref")], backend=OracleBackend(8))
    a = np.array([h["model_loss"] for h in t_hip.history])
    b = np.array([h["model_loss"] for h in t_ref.history])
    assert len(a) == 4 and np.all(np.isfinite(a))
    assert len(set(np.round(b, 6))) > 1                                # the batches differ from step to step
    np.testing.assert_allclose(a, b, rtol=2e-4)
    assert os.path.exists(os.path.join(tmp_path, "hip", "guitar", f"8_{t_hip.hparams.delta_t}_4", "model.ckpt.npz"))


def test_trainer_model_loss_for_rho():
    """Trainer.step's model_loss is the mean of the per-clip losses for RhoCMPS too (its gradient buffer is longer)."""
    from audio_mps_amd import HParams, RhoCMPS
    from audio_mps_amd.scan import HipScan
    from audio_mps_amd.train import Trainer
    hp = HParams(minibatch_size=5, bond_dim=6, initial_rank=3)
    data = make_audio(5, 120, hp.delta_t, 2)
    m = RhoCMPS(hp, data_iterator=data, seed=1, backend=HipScan(6))
    per = m.loss_per_clip()
    out = Trainer(m, hp).step()
    assert out["model_loss"] == pytest.approx(float(np.mean(per.astype(np.float64))), rel=1e-5)


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_nccl_single_rank_collectives(tmp_path):
    """The RCCL code path of audio_mps_amd.parallel (backend "nccl" = RCCL on ROCm) with a one-rank group on
    this box's GPU: init with device_id, all-reduce of the flat gradient buffer, barrier, max-over-ranks."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import os, sys, socket
        sys.path.insert(0, %r)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch, torch.distributed as dist
        from audio_mps_amd.parallel import DataParallel
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        dp = DataParallel.__new__(DataParallel)
        dp.rank, dp.world_size, dp.local_rank, dp.device, dp.backend, dp._own_group = 0, 2, 0, dev, "nccl", False
        dp.world_size = 2          # take the collective branch although the group has one rank
        flat = torch.arange(10, dtype=torch.float32, device=dev)
        host, nb = dp.allreduce_sums(flat, 7)
        assert nb == 7 and host.tolist() == list(range(10)), (nb, host)
        dp.time_collective = True  # bench.py's all-reduce timing: HIP events around the collective
        for _ in range(3):
            host, nb = dp.allreduce_sums(flat, 7)
        assert len(dp.collective_ms) == 3 and 0.0 < dp.collective_us() < 1e6, dp.collective_ms
        assert dp.measured_world_size() == 1           # an all-reduce of ones over the (one-rank) RCCL group
        dp.barrier()
        assert dp.max_over_ranks(1.5) == 1.5
        dist.destroy_process_group()
        print("OK")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("args", [
    ["--bond-dim", "4", "--T", "96", "--batch-per-gpu", "3", "--cpu-clips", "8"],                 # the sample (8 clips) exceeds the batch
    ["--bond-dim", "12", "--T", "200", "--batch-per-gpu", "6", "--cpu-clips", "4"],               # 16-row kernels
    ["--bond-dim", "32", "--T", "300", "--batch-per-gpu", "8", "--cpu-clips", "8"],               # hot path, with precision_ab
    ["--bond-dim", "64", "--T", "120", "--batch-per-gpu", "4", "--cpu-clips", "4", "--variant", "3"],   # pair kernels
    ["--bond-dim", "40", "--T", "80", "--batch-per-gpu", "2", "--cpu-clips", "2"],                # block kernels (AUTO above 32)
])
def test_bench_runs_at_small_shapes(args):
    """bench.py end to end (child process) at shapes that take seconds: one JSON line with the contract's keys, the
    in-bench parity check green.  Guards the code outside the timed region (it once assumed sample <= batch)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1"] + args,
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = proc.stdout.strip().splitlines()
    # the LAST line is the contract's record and must fit the driver's record (BENCH_r04.parsed was null for a 23.5 KB line);
    # every earlier JSON line is a {"detail": ...} object without a top-level "metric"
    assert len(lines[-1]) < 4096, len(lines[-1])
    line = json.loads(lines[-1])
    earlier = [json.loads(l) for l in lines[:-1] if l.startswith("{")]
    assert earlier and all("detail" in d and "metric" not in d for d in earlier)
    assert sorted(line["detail_lines"]) == sorted(d["detail"] for d in earlier)
    assert "roofline_detail" in line["detail_lines"] and "parity_in_bench" in line["detail_lines"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "algorithmic_bytes"):
        assert key in line["roofline"], key
    assert "kernels" not in line["roofline"] and "workload" in line["config"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert line["steps"] == 2 and line["n_gpus"] == 1 and line["value"] > 0
    assert line["parity_in_bench"]["ok"], line["parity_in_bench"]


def test_model_accessors_follow_the_device_resident_state():
    """with the device-resident optimiser step the current variables live on the GPU; anything that reads the model between
    steps (variables, R, freqs, psi_0, loss) must see them, not the values of step 0.  The host copies are refreshed lazily: once per
    read after a step, not per step."""
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.scan import HipScan
    from audio_mps_amd.train import Trainer
    D = 8
    hp = HParams(minibatch_size=4, bond_dim=D, learning_rate=0.01)
    data = make_audio(4, 120, hp.delta_t, 3)
    m_dev = PsiCMPS(hp, data_iterator=data, seed=0, backend=HipScan(D))
    m_host = PsiCMPS(hp, data_iterator=data, seed=0, backend=HipScan(D))
    t_dev, t_host = Trainer(m_dev, hp, device_step=True), Trainer(m_host, hp)
    R0 = m_dev.R.copy()
    for _ in range(3):
        t_dev.step(sync=False)
        t_host.step()
    assert t_dev._dirty                                             # nothing has come back yet
    R3 = m_dev.R                                                    # first read: one sync
    assert not t_dev._dirty and np.max(np.abs(R3 - R0)) > 1e-4
    np.testing.assert_allclose(R3, m_host.R, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(m_dev.freqs, m_host.freqs, rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(m_dev.psi_0, m_host.psi_0, rtol=2e-5, atol=2e-6)
    assert float(m_dev.loss) == pytest.approx(float(m_host.loss), rel=1e-5, abs=1e-6)
    t_dev.step(sync=False)
    assert t_dev._dirty
    # ASSIGNING to model.variables while the device state is live is honoured -- the state comes back, is dropped, and the
    # next device step starts from the host values (such writes used to be silently overwritten by the next sync)
    t_host.step()
    for mm in (m_dev, m_host):
        mm.variables["Rx"] *= np.float32(0.5)
        mm.variables["freqs"] = (mm.variables["freqs"] + np.float32(0.25)).astype(np.float32)
    assert t_dev._dev is None and not t_dev._dirty
    for _ in range(2):
        t_dev.step(sync=False)
        t_host.step()
    np.testing.assert_allclose(m_dev.R, m_host.R, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(m_dev.freqs, m_host.freqs, rtol=2e-5, atol=1e-4)
    import gc, weakref
    ref = weakref.ref(t_dev)
    del t_dev
    gc.collect()
    assert ref() is None and m_dev._owner() is None                # the model does not keep its Trainer (and the GPU buffers) alive
    assert np.all(np.isfinite(m_dev.variables["Rx"]))
