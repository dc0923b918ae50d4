"""Data parallelism over the batch axis: one process per GPU, one collective per optimiser step.

The reference is single-process (no distributed code at all).  The scan shards naturally
because clips are independent given the parameters (model.py:260; the only cross-clip op is the final
reduce_mean, model.py:267): rank r owns clips [r*B/W, (r+1)*B/W), parameters are replicated, and the
only exchange is ONE all-reduce(sum) of the flat buffer the reverse kernel emits
(2 D^2 + 3 D + 2 floats: dR, dfreqs, dpsi0, dA, sum loss) plus the local clip count -- 8.6 KB at D=32.
On ROCm the "nccl" backend of torch.distributed IS RCCL (over xGMI inside a node); "gloo" is used for the
CPU tests.  The message is latency-bound, so nothing is bucketed or overlapped: it is a single call.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


# The one collective is 2 D^2 + 3 D + 2 floats (8.6 KB at D = 32, 133 KB at D = 128) against a step of >= 10 ms: latency-bound.  On a
# fully connected 8-GPU xGMI node a ring pays 2 (N - 1) = 14 hops whatever the size, and what a hop costs at this size is the
# protocol's synchronisation, not the bytes: RCCL's LL protocol (8-byte {data, flag} granules, no separate fence per chunk) is the
# low-latency one; LL128 / Simple only win from ~100s of KB.  RCCL's tuner picks LL at this size by itself; setting it makes the
# choice explicit and keeps a tuner change from moving the step.  NCCL_ALGO is left to RCCL: inside
# one node Tree degenerates to a chain of the same hop count, and the one-shot all-to-all form SURVEY describes is not an RCCL
# algorithm one can select by name.  Both are only DEFAULTS: a value in the environment wins.
LOW_LATENCY_ENV = {"NCCL_PROTO": "LL"}


def collective_settings() -> dict:
    """What is in force for the collective (for the bench line)."""
    return {k: os.environ.get(k) for k in ("NCCL_PROTO", "NCCL_ALGO", "RCCL_MSCCL_ENABLE", "RCCL_MSCCLPP_ENABLE", "NCCL_MIN_NCHANNELS")}


class DataParallel:
    time_collective = False       # bench.py sets it: HIP events around the all-reduce ("print the all-reduce us")
    collective_ms = None          # list of timed collectives (ms), created on first use

    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None, force_group: bool = False):
        """force_group: build the process group and run the collectives even when WORLD_SIZE is 1 (a one-rank RCCL
        communicator: `bench.py --gpus 1 --spawn` uses it so that the N = 1 line crosses the same code as N > 1)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self._own_group = False
        self.collective_ms = []
        self.collective = self.world_size > 1 or force_group      # take the collective branches
        if self.collective:
            if backend is None:
                backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
            self.backend = backend
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                kwargs = {}
                if backend == "nccl":
                    for k, v in LOW_LATENCY_ENV.items():          # before the communicator exists
                        os.environ.setdefault(k, v)
                if backend == "nccl" and device is not None:
                    kwargs["device_id"] = torch.device(device)
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size, **kwargs)
                self._own_group = True
        else:
            self.backend = None

    def _collective(self) -> bool:
        # (objects built with __new__ in tests carry no `collective` attribute: world_size decides)
        return getattr(self, "collective", self.world_size > 1)

    # -- sharding ---------------------------------------------------------------------------------
    def shard(self, global_batch: int) -> Tuple[int, int]:
        """(first clip, number of clips) of this rank; the remainder goes to the lowest ranks."""
        W, r = self.world_size, self.rank
        base, rem = divmod(int(global_batch), W)
        count = base + (1 if r < rem else 0)
        start = r * base + min(r, rem)
        return start, count

    # -- the one collective -----------------------------------------------------------------------
    def allreduce_sums(self, flat: torch.Tensor, local_clips: int) -> Tuple[np.ndarray, int]:
        """Sum the per-rank gradient/loss sums and clip counts over all ranks.
        flat: [G] float32 tensor on this rank's device (or CPU for gloo).  Returns (host float64 [G], B_global)."""
        if not self._collective():
            return flat.detach().cpu().numpy().astype(np.float64), int(local_clips)
        # gloo has no device collectives on this build: its (rehearsal / CPU-test) path reduces on the host
        red_dev = flat.device if self.backend == "nccl" else torch.device("cpu")
        buf = torch.empty(flat.numel() + 1, dtype=torch.float32, device=red_dev)
        buf[:-1] = flat
        buf[-1] = float(local_clips)
        if self.time_collective and buf.is_cuda:
            # the collective is enqueued behind the current stream and the current stream waits for it, so two events
            # on the current stream bracket exactly the collective (plus its stream hand-over)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            e1.record()
            host = buf.detach().cpu().numpy().astype(np.float64)     # synchronises the stream
            if self.collective_ms is None:
                self.collective_ms = []
            self.collective_ms.append(e0.elapsed_time(e1))
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            host = buf.detach().cpu().numpy().astype(np.float64)
        return host[:-1], int(round(host[-1]))

    def allreduce_device(self, flat: torch.Tensor) -> torch.Tensor:
        """Sum `flat` over all ranks IN PLACE on the device (the device-resident optimiser step consumes it there); the global
        clip count is known without communication (every rank knows the global batch and its shard).  RCCL reduces the device
        buffer directly; the gloo rehearsal goes through the host."""
        if not self._collective():
            return flat
        if self.backend == "nccl":
            if self.time_collective:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                e1.record()
                self._pending_events = getattr(self, "_pending_events", []) + [(e0, e1)]
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        else:
            host = flat.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        return flat

    def collective_us(self) -> Optional[float]:
        """Mean HIP-event time of the timed all-reduces in microseconds (None when nothing was timed)."""
        for e0, e1 in getattr(self, "_pending_events", []):      # device-side all-reduces: their events are read here, after the run
            e1.synchronize()
            self.collective_ms.append(e0.elapsed_time(e1))
        self._pending_events = []
        return 1e3 * float(np.mean(self.collective_ms)) if self.collective_ms else None

    def gather_floats(self, value: float) -> np.ndarray:
        """One float from every rank -> array [world_size] (same on all ranks)."""
        if not self._collective():
            return np.array([float(value)])
        dev = self.device if self.backend == "nccl" else "cpu"
        mine = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        parts = [torch.zeros_like(mine) for _ in range(self.world_size)]
        dist.all_gather(parts, mine)
        return np.array([float(p.item()) for p in parts])

    def measured_world_size(self) -> int:
        """The number of ranks an actual all-reduce of ones adds up to, checked against dist.get_world_size()."""
        if not self._collective():
            return 1
        dev = self.device if self.backend == "nccl" else "cpu"
        one = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        n = int(round(float(one.item())))
        if n != dist.get_world_size():
            raise RuntimeError(f"all-reduce of ones gave {n}, process group says {dist.get_world_size()}")
        return n

    def barrier(self):
        if self._collective():
            if self.backend == "nccl" and self.device is not None:
                dist.barrier(device_ids=[torch.device(self.device).index])
            else:
                dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self._collective():
            return float(value)
        dev = self.device if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def replicas_identical(self, t: torch.Tensor) -> bool:
        """True when every rank holds bit-for-bit the same tensor (MAX and MIN over ranks of its int32 view coincide): the
        data-parallel invariant after an optimiser step (same all-reduced sums -> same Adam update on every rank)."""
        if not self._collective():
            return True
        bits = t.detach().contiguous().view(torch.int32)
        if self.backend != "nccl":
            bits = bits.cpu()
        hi, lo = bits.clone(), bits.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        return bool(torch.equal(hi, lo))

    def close(self):
        if self._own_group and dist.is_initialized():
            dist.destroy_process_group()
            self._own_group = False
