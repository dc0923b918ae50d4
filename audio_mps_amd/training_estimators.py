"""Estimator-style trainer mirroring /root/reference/training_estimators.py (the surface BASELINE.json calls
"the AudioMPS / train.py Estimator surface").

  flags                     training_estimators.py:16-39   viz_steps=1, max_steps=5001, bond_d=10, dt=0.001, discr=False,
                                                            batch_size=32, model_dir, data_dir
  audiomps(...)             :43-45                          AudioMPS(bond_d, dt, batch_size, data_iterator=data, mixed=discr)
  model_fn                  :48-74                          loss = audiomps(...).loss; AdamOptimizer(1e-3).minimize(loss)
  static_nsynth_dataset     :76-85                          TFRecord -> "audio" FixedLenFeature([2**16]) float32
  build_input_fns           :87-95                          shuffle(24).repeat().batch(batch_size)
  main                      :97-117                         estimator.train(input_fn, steps=viz_steps), max_steps // viz_steps times,
                                                            checkpoint every viz_steps

tf.estimator itself is replaced by a small loop: "train_op" = forward + reverse HIP scans, chain rule, Adam.
Note the reference's model_fn minimises the bare model loss (no regularisers), unlike train.py.
"""
from __future__ import annotations

import argparse
import os
from typing import Callable, Optional

import numpy as np

from .model import AudioMPS, HParams
from .parallel import DataParallel
from .train import AdamOptimizer


def build_parser():
    p = argparse.ArgumentParser(description="Train AudioMPS (mirror of audio-mps training_estimators.py)")
    p.add_argument("--viz_steps", type=int, default=1)             # :16-17
    p.add_argument("--max_steps", type=int, default=5001)          # :18-19
    p.add_argument("--bond_d", type=int, default=10)               # :20-21
    p.add_argument("--dt", type=float, default=0.001)              # :22-23
    p.add_argument("--discr", action="store_true", default=False)  # :24-27
    p.add_argument("--batch_size", type=int, default=32)           # :28-31
    p.add_argument("--model_dir", default="../logging/loggingrrrrrrrrtt")   # :32-35
    p.add_argument("--data_dir", default="")                       # :36-39 (a .tfrecords file)
    p.add_argument("--sample_duration", type=int, default=2 ** 16)  # FixedLenFeature([2 ** 16]) at :81
    p.add_argument("--seed", type=int, default=0)
    return p


def audiomps(bond_d, dt, batch_size, data, discr, **kw):
    """training_estimators.py:43-45."""
    return AudioMPS(bond_d, dt, batch_size, data_iterator=data, mixed=discr, **kw)


def static_nsynth_dataset(directory: str, sample_duration: int = 2 ** 16):
    """training_estimators.py:76-85: the records' "audio" feature as float32 [sample_duration] arrays."""
    from .tfrecord import _audio_records
    return _audio_records(directory, sample_duration, verify=False)


def build_input_fns(data_dir: str, batch_size: int, sample_duration: int = 2 ** 16, seed: int = 0) -> Callable[[], np.ndarray]:
    """training_estimators.py:87-95: shuffle(buffer_size=24).repeat().batch(batch_size)."""
    from .tfrecord import audio_batches
    return audio_batches(data_dir, batch_size, sample_duration, seed=seed, order="estimator")


class Estimator:
    """The part of tf.estimator.Estimator this script uses: train(input_fn, steps) with checkpoints in model_dir."""

    def __init__(self, params: dict, model_dir: Optional[str] = None, save_checkpoints_steps: int = 1,
                 dp: Optional[DataParallel] = None, model_kw: Optional[dict] = None):
        self.params = params
        self.model_dir = model_dir
        self.save_checkpoints_steps = max(1, int(save_checkpoints_steps))
        self.dp = dp if dp is not None else DataParallel()
        self.model = audiomps(params["bond_d"], params["dt"], params["batch_size"], None, params["discr"],
                              **(model_kw or {}))
        self.opt = AdamOptimizer(1e-3)                               # :67
        self.global_step = 0
        self.last_loss = None
        if model_dir:
            self._restore()

    def _ckpt(self):
        return os.path.join(self.model_dir, "model.ckpt.npz")

    def _restore(self):
        path = self._ckpt()
        if os.path.exists(path):
            with np.load(path) as z:
                for k in self.model.variables:
                    self.model.variables[k] = np.asarray(z[f"model/{k}"], dtype=np.float32)
                self.opt.load_state_dict({k[5:]: z[k] for k in z.files if k.startswith("adam/")})
                self.global_step = int(z["global_step"])

    def _save(self):
        os.makedirs(self.model_dir, exist_ok=True)
        payload = {f"model/{k}": v for k, v in self.model.variables.items()}
        payload.update({f"adam/{k}": v for k, v in self.opt.state_dict().items()})
        payload["global_step"] = np.int64(self.global_step)
        tmp = self._ckpt() + ".tmp.npz"
        np.savez(tmp, **payload)
        os.replace(tmp, self._ckpt())

    def train(self, input_fn: Callable[[], np.ndarray], steps: int):
        """model_fn in TRAIN mode, `steps` times (training_estimators.py:48-74, 114-115)."""
        start, count = None, None
        for _ in range(steps):
            batch = input_fn()
            if start is None:
                start, count = self.dp.shard(batch.shape[0])
            flat, b_local = self.model.grad_sums(batch[start:start + count])
            host, b_global = self.dp.allreduce_sums(flat, b_local)
            loss, grads = self.model.chain_rule(host, b_global, with_reg=False)     # :64, :68 minimise(loss)
            self.opt.apply_gradients(self.model.variables, grads)
            self.global_step += 1
            self.last_loss = float(loss)
            if self.model_dir and self.dp.rank == 0 and self.global_step % self.save_checkpoints_steps == 0:
                self._save()
        return self


def main(argv=None):
    import torch
    args = build_parser().parse_args(argv)
    params = vars(args)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dp = DataParallel(device=torch.device("cuda", local_rank))
    if args.data_dir:
        train_input_fn = build_input_fns(args.data_dir, args.batch_size, args.sample_duration, seed=args.seed)
    else:   # no dataset given: the reference's synthetic fixture (data.py:8-22), so that the script runs anywhere
        from .data import damped_sine
        state = {"i": 0}

        def train_input_fn():
            state["i"] += 1
            return damped_sine(args.batch_size, args.sample_duration, args.dt, seed=args.seed + state["i"])
    estimator = Estimator(params, model_dir=args.model_dir, save_checkpoints_steps=args.viz_steps, dp=dp,
                          model_kw={"seed": args.seed})
    for _ in range(args.max_steps // args.viz_steps):                               # :114-115
        estimator.train(train_input_fn, steps=args.viz_steps)
        if dp.rank == 0:
            print(f"global_step {estimator.global_step}: loss_function {estimator.last_loss:.6f}")
    dp.close()


if __name__ == "__main__":
    main()
