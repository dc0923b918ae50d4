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


VAR_ORDER = ("A", "Rx", "Ry", "freqs", "psi_x", "psi_y")      # layout of the device-resident variable / Adam-slot buffers (include/cmps.h)


class Trainer:
    """One optimiser step = parameter tables, forward scan, reverse scan, (all-reduce), chain rule, Adam.

    ``device_step=False``: the gradient sums come to the host, chain rule (model.chain_rule) and Adam (AdamOptimizer) run in
    numpy, the parameters are uploaded again -- the tested reference implementation of the optimiser half.
    ``device_step=True`` (PsiCMPS on the HIP backend): variables, Adam slots and effective parameters live on the GPU and
    cmps_psi_apply_step does the optimiser half there; a step issues no device -> host copy.  ``step`` then returns the
    losses as a 2-element device tensor under "losses_dev" unless ``sync=True``; ``sync_to_host()`` brings variables and
    Adam state back (checkpoints, sampling, inspection)."""

    HISTORY = 4096                # step records kept (a long run must not grow without bound)

    def __init__(self, model: PsiCMPS, hparams: HParams, dp: Optional[DataParallel] = None, device_step: bool = False):
        from collections import deque
        self.model = model
        self.hparams = hparams
        self.dp = dp if dp is not None else DataParallel()
        self.opt = AdamOptimizer(hparams.learning_rate)
        self.global_step = 0
        self.history = deque(maxlen=self.HISTORY)   # the dict of every step (the scalar summaries of train.py:62-66)
        self.device_step = bool(device_step)
        self._dev = None
        if self.device_step and not isinstance(model, PsiCMPS):
            raise ValueError("device_step: the device-resident optimiser step exists for PsiCMPS (the hot path) only")

    # -- host reference path ----------------------------------------------------------------------
    def step(self, data=None, sync: bool = True, global_batch: Optional[int] = None) -> dict:
        """``data``: this rank's shard [B_local, T] (or None to use model.data_iterator).  An empty shard (a short final batch
        with fewer clips than ranks) skips the scan and contributes zeros to the all-reduce."""
        if self.device_step:
            return self._step_device(data, sync, global_batch)
        D = self.model.bond_d
        if data is not None and not callable(data) and len(data) == 0:
            import torch
            # sized from the MODEL (RhoCMPS appends its column cotangents: a shorter buffer here would be a size mismatch in the
            # all-reduce -- ADVICE r3), not from what an earlier call happened to leave behind
            dev = getattr(self.model._get_backend(), "device", None)
            flat = torch.zeros(self.model.flat_size(), dtype=torch.float32, device=dev if dev is not None else "cpu")
            b_local = 0
        else:
            flat, b_local = self.model.grad_sums(data)
        host, b_global = self.dp.allreduce_sums(flat, b_local)
        if global_batch is not None:
            b_global = int(global_batch)
        total, grads = self.model.chain_rule(host, max(b_global, 1), with_reg=True)       # train.py:55-60
        # sum_b loss_b sits at the end of the pure-state layout (2 D^2 + 3 D + 2 floats); RhoCMPS appends the column
        # cotangents behind it (include/cmps.h: cmps_rho_loss_bwd)
        model_loss = host[2 * D * D + 3 * D + 1] / max(b_global, 1)
        self.opt.apply_gradients(self.model.variables, grads)                     # train.py:89
        self.global_step += 1
        out = {"model_loss": float(model_loss), "total_loss": float(total), "global_step": self.global_step,
               "global_batch": b_global}
        self.history.append(out)
        return out

    # -- device-resident path ---------------------------------------------------------------------
    def _device_state(self):
        if self._dev is None:
            import torch
            be = self.model._get_backend()
            D = self.model.bond_d
            V = 2 * D * D + 3 * D + 1
            host = np.concatenate([np.asarray(self.model.variables[k], dtype=np.float32).ravel() for k in VAR_ORDER])
            st = {"vars": torch.from_numpy(host).to(be.device)}
            for slot, src in (("m", self.opt.m), ("v", self.opt.v)):
                h = np.concatenate([np.asarray(src.get(k, np.zeros_like(self.model.variables[k])), dtype=np.float32).ravel()
                                    for k in VAR_ORDER])
                st[slot] = torch.from_numpy(h).to(be.device)
            st["params"] = torch.empty(V, dtype=torch.float32, device=be.device)
            st["losses"] = torch.zeros(2, dtype=torch.float32, device=be.device)
            self._dev = st
            self._dirty = False
            import weakref
            self.model._device_owner = weakref.ref(self)           # model.variables now syncs lazily from the device state
            self._apply(None, 1)                                   # effective parameters of the current variables
        return self._dev

    def _lazy_sync(self):
        """Called by model.variables: device state -> host copies if a device step has run since the last sync."""
        if self._dev is not None and getattr(self, "_dirty", False) and not getattr(self, "_syncing", False):
            self._syncing = True
            try:
                self.sync_to_host()
            finally:
                self._syncing = False

    def _apply(self, grad_sums, global_batch):
        st, m, o = self._dev, self.model, self.opt
        lr_t = o.lr * math.sqrt(1.0 - o.b2 ** max(o.t, 1)) / (1.0 - o.b1 ** max(o.t, 1))
        m._get_backend().apply_step(st["vars"], st["m"], st["v"], grad_sums, global_batch, lr_t, o.b1, o.b2, o.eps,
                                    m.h_reg, m.r_reg, float(m._c_r), float(m._c_h), True, st["params"], st["losses"])

    def _step_device(self, data, sync, global_batch):
        import torch
        m = self.model
        st = self._device_state()
        be = m._get_backend()
        D = m.bond_d
        audio = m._to_device(m._batch(data))
        b_local, T = audio.shape
        if b_local > 0:
            be.set_params_dev(st["params"], m.sigma, m.delta_t, b_local, T, train=True)
            be.forward(audio, save_for_bwd=True)
            flat = be.backward()
        else:
            flat = torch.zeros(2 * D * D + 3 * D + 2, dtype=torch.float32, device=be.device)
        if global_batch is None:                                   # known without communication: the reference's batch is the global one
            global_batch = b_local if self.dp.world_size == 1 else int(self.hparams.minibatch_size)
        self.dp.allreduce_device(flat)
        self.opt.t += 1
        self._apply(flat, global_batch)
        self._dirty = True                                          # the host copies (model.variables, Adam slots) are stale now
        self.global_step += 1
        out = {"global_step": self.global_step, "global_batch": int(global_batch), "losses_dev": st["losses"]}
        if sync:
            lo = st["losses"].cpu().numpy()
            out.update(model_loss=float(lo[0]), total_loss=float(lo[1]))
            del out["losses_dev"]
            self.history.append(out)
        return out

    def drop_device_state(self):
        """Bring the device-resident state back and forget it: the next device step re-uploads variables and Adam slots from the host
        (called when model.variables is assigned to while the state is live)."""
        if self._dev is not None:
            self.sync_to_host()
            self._dev = None
            self._dirty = False

    def sync_to_host(self):
        """Device-resident state -> model.variables and the host AdamOptimizer's slots (checkpoints, sampling, inspection)."""
        if self._dev is None:
            return
        self._syncing_back = True              # (the assignments below must not drop the state they are reading)
        try:
            self._sync_to_host()
        finally:
            self._syncing_back = False

    def _sync_to_host(self):
        D = self.model.bond_d
        shapes = {"A": (), "Rx": (D, D), "Ry": (D, D), "freqs": (D,), "psi_x": (D,), "psi_y": (D,)}
        self._dirty = False
        for name, dst in (("vars", self.model._variables), ("m", self.opt.m), ("v", self.opt.v)):
            flat = self._dev[name].cpu().numpy()
            o = 0
            for k in VAR_ORDER:
                n = int(np.prod(shapes[k])) if shapes[k] else 1
                dst[k] = np.asarray(flat[o:o + n].reshape(shapes[k]), dtype=np.float32).copy()
                o += n

    # -- checkpoint / resume (train.py:93: save_checkpoint_secs=60, automatic restore from logdir) --
    def save(self, path: str):
        self.sync_to_host()
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
        self._dev = None                                            # the device-resident copy is rebuilt from the restored state
        self._dirty = False
        return True


def build_parser():
    p = argparse.ArgumentParser(description="Train PsiCMPS on MI355X (mirror of audio-mps train.py)")
    p.add_argument("--log_every", type=int, default=1, help="print the losses every n-th step (the device-resident step copies them "
                   "to the host only then)")
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
                   help="0 auto (float32: wave-per-clip kernels for D <= 32, wide kernels above), 1 block, 2 wave, "
                        "3 MFMA pair kernels (32 < D <= 128, bf16 mat-vec operands), 5 float32 wide kernels (32 < D <= 128)")
    p.add_argument("--host_optimizer", action="store_true",
                   help="chain rule and Adam in numpy on the host (the reference implementation of the optimiser half) instead of "
                        "the device-resident step (cmps_psi_apply_step)")
    return p


def main(argv=None, backend=None):
    """``backend``: a scan backend to use instead of HipScan (CPU tests of the host logic inject one; the product
    always builds a HipScan and fails loudly without a GPU)."""
    from .data import get_audio
    args = build_parser().parse_args(argv)
    hp = HParams(delta_t=1.0 / args.sample_rate, h_reg=200.0 / (math.pi * args.sample_rate) ** 2)  # train.py:41-43
    hp.parse(args.hparams)
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
    device_step = (cls is PsiCMPS and not args.host_optimizer and type(backend).__name__ == "HipScan")
    trainer = Trainer(model, hp, dp, device_step=device_step)
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
        # the losses come to the host only on logging steps (the device-resident step needs no device -> host copy otherwise)
        log_now = (it % max(args.log_every, 1) == 0) or it == args.max_steps - 1
        out = trainer.step(full[s0:s0 + c0], sync=log_now, global_batch=int(full.shape[0]))
        if dp.rank == 0:
            if log_now:
                print(f"step {out['global_step']}: model_loss {out['model_loss']:.6f} total_loss {out['total_loss']:.6f}")
            if time.time() - last_save > args.save_checkpoint_secs or it == args.max_steps - 1:
                trainer.save(ckpt)
                last_save = time.time()
    dp.close()
    return trainer


if __name__ == "__main__":
    main()
