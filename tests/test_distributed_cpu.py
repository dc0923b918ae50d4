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
