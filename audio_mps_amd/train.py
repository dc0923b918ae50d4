"""Trainer mirroring /root/reference/train.py for the pure-state model.

  hparams            train.py:41-44      (same names, same defaults, ``--hparams=a=b,...`` override)
  data               train.py:46-47      get_audio(...)
  model              train.py:49-53      PsiCMPS(hparams, data_iterator=data)
  total loss         train.py:55-60      loss + h_reg * sum(freqs^2) + r_reg * sum(|R|^2)   (effective R, freqs)
  optimiser          train.py:88-89      tf.train.AdamOptimizer(learning_rate).minimize(total_loss, global_step)
  loop / checkpoints train.py:93-94      tf.contrib.training.train(..., save_checkpoint_secs=60, logdir=...)

The TensorBoard summaries (train.py:62-85: scalars, audio, histogram, tfplot images, model.sample) are
not part of the hot path; the scalar ones are returned by ``Trainer.step`` as a dict.
Run:  python -m audio_mps_amd.train --dataset=damped_sine --hparams=bond_dim=32,minibatch_size=64
(under torchrun for several GPUs: the batch is sharded over ranks, one RCCL all-reduce per step).
"""
from __future__ import annotations

import argparse
import math
import os
import time
from typing import Dict, Optional

import numpy as np

from .model import HParams, PsiCMPS, RhoCMPS
from .parallel import DataParallel


class AdamOptimizer:
    """tf.train.AdamOptimizer (train.py:89; legacy graph: beta1 0.9, beta2 0.999, eps 1e-8,
    logging/graph.pbtxt:32102-32192): lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);
    m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  var -= lr_t * m / (sqrt(v) + eps)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.t = 0
        self.m: Dict[str, np.ndarray] = {}
        self.v: Dict[str, np.ndarray] = {}

    def apply_gradients(self, variables: Dict[str, np.ndarray], grads: Dict[str, np.ndarray]):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for name in sorted(grads):
            g = np.asarray(grads[name], dtype=np.float32)
            if name not in self.m:
                self.m[name] = np.zeros_like(g)
                self.v[name] = np.zeros_like(g)
            m = self.m[name] = (self.b1 * self.m[name] + (1.0 - self.b1) * g).astype(np.float32)
            v = self.v[name] = (self.b2 * self.v[name] + (1.0 - self.b2) * g * g).astype(np.float32)
            variables[name] = (variables[name] - np.float32(lr_t) * m / (np.sqrt(v) + np.float32(self.eps))).astype(np.float32)

    def state_dict(self):
        return {"t": self.t, **{f"m/{k}": v for k, v in self.m.items()}, **{f"v/{k}": v for k, v in self.v.items()}}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        self.m = {k[2:]: np.asarray(v) for k, v in sd.items() if k.startswith("m/")}
        self.v = {k[2:]: np.asarray(v) for k, v in sd.items() if k.startswith("v/")}


class Trainer:
    """One optimiser step = upload parameters, forward scan, reverse scan, (all-reduce), chain rule, Adam."""

    def __init__(self, model: PsiCMPS, hparams: HParams, dp: Optional[DataParallel] = None):
        self.model = model
        self.hparams = hparams
        self.dp = dp if dp is not None else DataParallel()
        self.opt = AdamOptimizer(hparams.learning_rate)
        self.global_step = 0
        self.history = []             # the dict of every step (the scalar summaries of train.py:62-66)

    def step(self, data=None) -> dict:
        """``data``: this rank's shard [B_local, T] (or None to use model.data_iterator)."""
        flat, b_local = self.model.grad_sums(data)
        host, b_global = self.dp.allreduce_sums(flat, b_local)
        total, grads = self.model.chain_rule(host, b_global, with_reg=True)       # train.py:55-60
        # sum_b loss_b sits at the end of the pure-state layout (2 D^2 + 3 D + 2 floats); RhoCMPS appends the column
        # cotangents behind it (include/cmps.h: cmps_rho_loss_bwd)
        D = self.model.bond_d
        model_loss = host[2 * D * D + 3 * D + 1] / b_global
        self.opt.apply_gradients(self.model.variables, grads)                     # train.py:89
        self.global_step += 1
        out = {"model_loss": float(model_loss), "total_loss": float(total), "global_step": self.global_step,
               "global_batch": b_global}
        self.history.append(out)
        return out

    # -- checkpoint / resume (train.py:93: save_checkpoint_secs=60, automatic restore from logdir) --
    def save(self, path: str):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        payload = {f"model/{k}": v for k, v in self.model.variables.items()}
        payload.update({f"adam/{k}": v for k, v in self.opt.state_dict().items()})
        payload["global_step"] = np.int64(self.global_step)
        tmp = path + ".tmp.npz"
        np.savez(tmp, **payload)
        os.replace(tmp, path)

    def restore(self, path: str) -> bool:
        if not os.path.exists(path):
            return False
        with np.load(path) as z:
            for k in list(self.model.variables):
                self.model.variables[k] = np.asarray(z[f"model/{k}"], dtype=np.float32)
            self.opt.load_state_dict({k[5:]: z[k] for k in z.files if k.startswith("adam/")})
            self.global_step = int(z["global_step"])
        return True


def build_parser():
    p = argparse.ArgumentParser(description="Train PsiCMPS on MI355X (mirror of audio-mps train.py)")
    p.add_argument("--mps_model", default="psi_mps", choices=["rho_mps", "psi_mps"])           # train.py:18-20
    p.add_argument("--dataset", default="damped_sine",
                   choices=["damped_sine", "guitar", "organ", "nsynth"])                       # train.py:23-25
    p.add_argument("--sample_duration", type=int, default=2 ** 16)                              # train.py:27
    p.add_argument("--sample_rate", type=int, default=16000)                                    # train.py:28
    p.add_argument("--hparams", default="")                                                     # train.py:31
    p.add_argument("--datadir", default="./data")                                               # train.py:32
    p.add_argument("--logdir", default="../logging/audio_mps")                                  # train.py:33
    p.add_argument("--max_steps", type=int, default=100)
    p.add_argument("--save_checkpoint_secs", type=float, default=60.0)                          # train.py:93
    p.add_argument("--seed", type=int, default=0)                                               # train.py:13
    p.add_argument("--kernel_variant", type=int, default=0,
                   help="0 auto (float32: wave-per-clip kernels for D <= 32, block kernels above), 1 block, 2 wave, "
                        "3 MFMA pair kernels (32 < D <= 128, bf16 mat-vec operands)")
    return p


def main(argv=None, backend=None):
    """``backend``: a scan backend to use instead of HipScan (CPU tests of the host logic inject one; the product
    always builds a HipScan and fails loudly without a GPU)."""
    from .data import get_audio
    args = build_parser().parse_args(argv)
    hp = HParams(delta_t=1.0 / args.sample_rate, h_reg=200.0 / (math.pi * args.sample_rate) ** 2)  # train.py:41-43
    hp.parse(args.hparams)
    if args.mps_model == "rho_mps":
        rank = hp.initial_rank if hp.initial_rank is not None else hp.bond_dim
        if rank * hp.bond_dim > 5000:
            raise SystemExit(f"rho_mps: initial_rank * bond_dim = {rank * hp.bond_dim} exceeds 5000 (the columns of rho are "
                             "LDS-resident in the HIP kernels); lower --hparams=initial_rank=... or bond_dim")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        import torch
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dp = DataParallel(device=dev)
        from .scan import HipScan
        backend = HipScan(hp.bond_dim, device=dev, variant=args.kernel_variant)
    else:
        dp = DataParallel(backend="gloo")
    start, count = dp.shard(hp.minibatch_size)
    cls = RhoCMPS if args.mps_model == "rho_mps" else PsiCMPS                                   # train.py:50-53
    model = cls(hp, seed=args.seed, backend=backend)
    trainer = Trainer(model, hp, dp)
    logdir = f"{args.logdir}/{args.dataset}/{hp.bond_dim}_{hp.delta_t}_{hp.minibatch_size}"    # train.py:94
    ckpt = os.path.join(logdir, "model.ckpt.npz")
    if trainer.restore(ckpt) and dp.rank == 0:
        print(f"Restoring parameters from {ckpt} (global_step {trainer.global_step})")
    last_save = time.time()
    # train.py:46-47: the input pipeline is built ONCE; for TFRecord datasets get_audio returns the one-shot iterator's
    # get_next (batch -> shuffle(24) -> repeat, data.py:37-40), for damped_sine every step draws fresh delays (data.py:15)
    source = get_audio(args.datadir, args.dataset, hp, args.sample_duration, seed=args.seed) \
        if args.dataset != "damped_sine" else None
    for it in range(args.max_steps):
        if source is not None:
            full = source()
        else:
            full = get_audio(args.datadir, args.dataset, hp, args.sample_duration, seed=args.seed + trainer.global_step)
        # every rank reads the same global batch and keeps its shard (a short final batch of an epoch is sharded as it is)
        s0, c0 = dp.shard(full.shape[0]) if full.shape[0] != hp.minibatch_size else (start, count)
        out = trainer.step(full[s0:s0 + c0])
        if dp.rank == 0:
            print(f"step {out['global_step']}: model_loss {out['model_loss']:.6f} total_loss {out['total_loss']:.6f}")
            if time.time() - last_save > args.save_checkpoint_secs or it == args.max_steps - 1:
                trainer.save(ckpt)
                last_save = time.time()
    dp.close()
    return trainer


if __name__ == "__main__":
    main()
