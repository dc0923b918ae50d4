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


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()
