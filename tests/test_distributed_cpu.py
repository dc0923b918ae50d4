"""world_size = 2 on CPU (gloo): the batch is sharded over ranks, each rank produces gradient SUMS for its
clips, one all-reduce of the flat buffer (+ clip count) gives every rank the global sums, and the host
chain rule + Adam then run redundantly and identically on every rank.  The scan is the oracle stand-in (no GPU
here); what is under test is audio_mps_amd.parallel + train, i.e. the N > 1 path of bench.py / train.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, T, D, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.parallel import DataParallel
    from audio_mps_amd.train import Trainer
    from _util import OracleBackend, make_audio
    hp = HParams(minibatch_size=B, bond_dim=D, learning_rate=0.01)
    dp = DataParallel(backend="gloo")
    assert dp.world_size == world and dp.rank == rank
    full = make_audio(B, T, hp.delta_t, 11)
    start, count = dp.shard(B)
    model = PsiCMPS(hp, seed=0, backend=OracleBackend(D))
    tr = Trainer(model, hp, dp)
    logs = [tr.step(full[start:start + count]) for _ in range(3)]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), total=np.array([l["total_loss"] for l in logs]),
             gb=np.array([l["global_batch"] for l in logs]), **{k: v for k, v in model.variables.items()})
    dp.barrier()
    dp.close()


@pytest.mark.timeout(300)
def test_two_ranks_match_single_process(tmp_path):
    B, T, D, world = 5, 48, 4, 2          # ragged: 3 + 2 clips
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, T, D, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(os.path.join(tmp_path, "rank0.npz"))
    r1 = np.load(os.path.join(tmp_path, "rank1.npz"))
    assert list(r0["gb"]) == [B] * 3
    for k in ("A", "Rx", "Ry", "freqs", "psi_x", "psi_y", "total"):
        np.testing.assert_array_equal(r0[k], r1[k])         # replicas stay bit-identical
    # single-process reference on the whole batch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from audio_mps_amd import HParams, PsiCMPS
    from audio_mps_amd.train import Trainer
    from _util import OracleBackend, make_audio
    hp = HParams(minibatch_size=B, bond_dim=D, learning_rate=0.01)
    model = PsiCMPS(hp, seed=0, backend=OracleBackend(D))
    tr = Trainer(model, hp)
    full = make_audio(B, T, hp.delta_t, 11)
    totals = [tr.step(full)["total_loss"] for _ in range(3)]
    np.testing.assert_allclose(r0["total"], totals, rtol=2e-5)
    for k in ("A", "Rx", "Ry", "freqs", "psi_x", "psi_y"):
        np.testing.assert_allclose(r0[k], model.variables[k], rtol=2e-4, atol=1e-6)


def test_shard_covers_batch():
    from audio_mps_amd.parallel import DataParallel
    for world in (1, 2, 3, 8):
        for B in (1, 5, 8, 1024, 8192):
            seen = []
            for r in range(world):
                dp = DataParallel.__new__(DataParallel)
                dp.world_size, dp.rank = world, r
                s, c = dp.shard(B)
                seen += list(range(s, s + c))
            assert seen == list(range(B))


@pytest.mark.timeout(300)
def test_bench_launcher_spawns_two_children():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start the ranks itself (child process of
    torch.distributed.run, before any GPU call in the parent), relay rank 0's JSON and exit 0.  The ranks here only
    set up the process group (gloo) and all-reduce once (--launcher-selftest): no GPU, no scan."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                          env=env, capture_output=True, text=True, timeout=280)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout
    rec = json.loads(lines[0])
    assert rec == {"launcher_selftest": True, "world_size": 2, "requested": 2, "allreduce_sum": 3.0, "clip_count": 21}


@pytest.mark.timeout(300)
def test_bench_launcher_propagates_child_failure():
    """A failing rank makes the launcher exit non-zero (here: WORLD_SIZE the children see != --gpus they were given is
    impossible by construction, so use an argument only the ranks reject)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    # the ranks have no GPU in this container: the real worker refuses to run -> non-zero exit through the launcher
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                           "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=280)
    if __import__("torch").cuda.is_available():
        pytest.skip("needs a box without GPUs")
    assert proc.returncode != 0
    assert not [l for l in proc.stdout.splitlines() if l.startswith("{")]


def _collectives_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from audio_mps_amd.parallel import DataParallel
    dp = DataParallel(backend="gloo")
    got = dp.gather_floats(10.0 + rank)
    n = dp.measured_world_size()
    mx = dp.max_over_ranks(float(rank))
    np.savez(os.path.join(out_dir, f"c{rank}.npz"), got=got, n=n, mx=mx, us=np.array(dp.collective_us() or -1.0))
    dp.close()


@pytest.mark.timeout(300)
def test_gather_and_measured_world_size(tmp_path):
    world = 2
    mp.spawn(_collectives_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"c{r}.npz"))
        np.testing.assert_array_equal(z["got"], [10.0, 11.0])
        assert int(z["n"]) == 2 and float(z["mx"]) == 1.0 and float(z["us"]) == -1.0


def _rho_empty_shard_worker(rank, world, port, T, D, r, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from audio_mps_amd import HParams, RhoCMPS
    from audio_mps_amd.parallel import DataParallel
    from audio_mps_amd.train import Trainer
    from _util import OracleBackend, make_audio
    B = 1                                   # a short final batch: fewer clips than ranks -> rank 1's shard is empty
    hp = HParams(minibatch_size=B, bond_dim=D, learning_rate=0.01, initial_rank=r)
    dp = DataParallel(backend="gloo")
    full = make_audio(B, T, hp.delta_t, 5)
    start, count = dp.shard(B)
    assert count == (1 if rank == 0 else 0)
    model = RhoCMPS(hp, seed=0, backend=OracleBackend(D))
    assert model.flat_size() == 2 * D * D + 3 * D + 2 + 2 * r * D
    tr = Trainer(model, hp, dp, device_step=False)
    logs = [tr.step(full[start:start + count]) for _ in range(2)]
    np.savez(os.path.join(out_dir, f"rho{rank}.npz"), total=np.array([l["total_loss"] for l in logs]),
             gb=np.array([l["global_batch"] for l in logs]), **{k: v for k, v in model.variables.items()})
    dp.barrier()
    dp.close()


@pytest.mark.timeout(300)
def test_rho_empty_shard_two_ranks(tmp_path):
    """the empty rank's zero buffer must have the MODEL's length (RhoCMPS appends 2 rank D column cotangents to the
    pure-state layout); with the pure-state length the two ranks would all-reduce tensors of different sizes."""
    T, D, r, world = 40, 4, 3, 2
    mp.spawn(_rho_empty_shard_worker, args=(world, _free_port(), T, D, r, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(os.path.join(tmp_path, "rho0.npz"))
    r1 = np.load(os.path.join(tmp_path, "rho1.npz"))
    assert list(r0["gb"]) == [1, 1] and np.all(np.isfinite(r0["total"]))
    for k in ("A", "Rx", "Ry", "freqs", "Wx", "Wy", "total"):
        np.testing.assert_array_equal(r0[k], r1[k])         # replicas stay bit-identical
