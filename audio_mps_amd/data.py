"""Input side of the reference: ``get_audio`` (data.py:6-45).  The synthetic ``damped_sine`` branch (data.py:8-22)
is the fixture the reference's model tests use; the TFRecord branch (data.py:25-43) is served by
``audio_mps_amd.tfrecord`` (pure Python reader; the reference's data blobs themselves are absent,
.MISSING_LARGE_BLOBS)."""
from __future__ import annotations

import numpy as np


def damped_sine(batch: int, input_length: int, delta_t: float, seed: int = 0) -> np.ndarray:
    """Gamma-delayed, exponentially damped 261.6 Hz sine, float32 [batch, input_length] (data.py:10-22).
    delays ~ Gamma(alpha=2, beta=2/delay_time) with delay_time = input_length / 100 (data.py:13,15);
    the reference draws them with tf.random_gamma, here a seeded numpy Generator."""
    rng = np.random.default_rng(seed)
    freq = 261.6            # Middle C (data.py:11)
    decay_time = 0.1        # data.py:12
    delay_time = input_length / 100
    delays = rng.gamma(shape=2.0, scale=delay_time / 2.0, size=batch).astype(np.float32)
    input_range = np.arange(input_length, dtype=np.float32)[None, :]
    times = (input_range - delays[:, None]) * np.float32(delta_t)
    wave = (np.float32(0.5) * (np.sign(times) + np.float32(1))
            * np.sin(np.float32(2 * np.pi * freq) * times) * np.exp(-times / np.float32(decay_time)))
    return wave.astype(np.float32)


INPUT_KINDS = ("damped_sine", "damped_sine_noise", "bandlimited")


def synthetic_audio(kind: str, batch: int, input_length: int, delta_t: float, seed: int = 0) -> np.ndarray:
    """The synthetic inputs of SURVEY.md 8(d), float32 [batch, input_length]:
       damped_sine        the reference's own generator (data.py:8-22), nothing added;
       damped_sine_noise  the same plus 0.02 N(0, 1) white noise (every step carries signal, also before the onset);
       bandlimited        x = 0.1 cumsum(N(0, 1)) / sqrt(T), clipped to [-1, 1] (a random walk: no silent stretch, no periodicity)."""
    if kind == "damped_sine":
        return damped_sine(batch, input_length, delta_t, seed=seed)
    rng = np.random.default_rng(seed + 12345)
    if kind == "damped_sine_noise":
        x = damped_sine(batch, input_length, delta_t, seed=seed)
        return (x + 0.02 * rng.standard_normal(x.shape)).astype(np.float32)
    if kind == "bandlimited":
        steps = rng.standard_normal((batch, input_length))
        return np.clip(0.1 * np.cumsum(steps, axis=1) / np.sqrt(input_length), -1.0, 1.0).astype(np.float32)
    raise ValueError(f"unknown synthetic input {kind!r}: one of {INPUT_KINDS}")


def get_audio(datadir, dataset, hps, sample_duration: int = 2 ** 16, seed: int = 0) -> np.ndarray:
    """``get_audio(datadir, dataset, hps)`` (data.py:6); ``sample_duration`` is the reference's global
    FLAGS.sample_duration (train.py:27, data.py:10)."""
    if dataset == "damped_sine":
        return damped_sine(hps.minibatch_size, sample_duration, hps.delta_t, seed=seed)
    # data.py:25-43: f"{datadir}/{dataset}.tfrecords" -> "audio" FixedLenFeature -> batch -> shuffle(24) -> repeat.
    # Returns a zero-argument callable yielding the next batch (the eager stand-in for iterator.get_next()).
    import os
    from .tfrecord import audio_batches
    path = os.path.join(str(datadir), f"{dataset}.tfrecords")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} (the reference's data blobs are not distributed: .MISSING_LARGE_BLOBS)")
    return audio_batches(path, hps.minibatch_size, sample_duration, seed=seed, order="data")
