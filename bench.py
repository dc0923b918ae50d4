#!/usr/bin/env python3
"""Benchmark of the hot path: audio samples/s for forward + backward of the PsiCMPS scan.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 with WORLD_SIZE unset makes this process a LAUNCHER: before anything touches
the GPU it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py`
as a child process (one rank per GPU over RCCL), relays rank 0's JSON line and exits with the child's code
(`--spawn` forces the same path at N = 1).  Under torch.distributed.run (WORLD_SIZE set) it is a rank.

A "step" is one complete optimiser step on one batch of synthetic audio already resident in HBM:
parameter upload + table rebuild, forward scan, reverse scan, slab reduction, (one RCCL all-reduce of the
gradient sums when N > 1), chain rule and Adam on the host.  Default workload at every N: BASELINE.json configs[2]
per GPU -- D=32, T=16000, batch 1024 per GPU (configs[3] = the same per GPU on 8 GPUs), so scaling is weak.
Inputs: the reference's damped sine (data.py:8-22) plus white noise, parameters by the reference's
initialisation rules (model.py:36-39, 49, 218-219) with train.py:41-43 hyper-parameters, seed 0.

Rank 0 prints ONE JSON line; `roofline` describes the dominant kernel, `cpu_baseline` is the plain-C restatement
(oracle/cmps_oracle.c) timed on this host's cores on a bounded sample, `parity_in_bench` compares the GPU results on
that same sample with the oracle's (outside the timed region), `precision_ab` times the three arithmetic modes of the
rank-1 gradient updates side by side.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 vector peak == fp32-input MFMA peak (dense)
BF16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (never the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak
RANK1_MODES = {"exact_f32": 0, "bf16x2": 1, "bf16x3": 2}
RANK1_LABEL = {0: "exact fp32 MFMA", 1: "bf16x2 split (16 operand bits), fp32 accumulate",
               2: "bf16x3 split (24 operand bits, fp32-faithful products), fp32 accumulate"}


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--bond-dim", type=int, default=32)
    p.add_argument("--T", type=int, default=16000)
    p.add_argument("--batch-per-gpu", type=int, default=1024)
    p.add_argument("--variant", type=int, default=0,
                   help="0 auto, 1 block-per-clip, 2 wave-per-clip, 3 MFMA pair kernels (needs --bond-dim 64 or 128; bf16 operands)")
    p.add_argument("--rank1", choices=sorted(RANK1_MODES), default="bf16x3",
                   help="arithmetic of the rank-1 gradient updates in the wave reverse scan (cmps_set_option)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-precision-ab", action="store_true")
    p.add_argument("--cpu-clips", type=int, default=0, help="clips in the CPU sample (0 = 8 per thread)")
    p.add_argument("--spawn", action="store_true", help="go through the child-process launcher even at --gpus 1")
    p.add_argument("--launcher-selftest", action="store_true",
                   help="ranks only set up the process group (gloo on CPU) and do one all-reduce: tests the launcher")
    p.add_argument("--force-collective", action="store_true",
                   help="build the process group and run the all-reduce / barrier / gather collectives even at WORLD_SIZE 1 (a "
                        "one-rank RCCL communicator); the launcher adds it for `--gpus 1 --spawn`")
    p.add_argument("--master-port", type=int, default=0)
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="N > 1 ranks that all use cuda:0 and reduce over gloo on the host: exercises every line of the multi-rank "
                        "path except RCCL itself on a one-GPU box (the line is marked and is NOT a scaling measurement)")
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: runs BEFORE any GPU call of this process and never execs
# ---------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_children(args, argv) -> int:
    """Start N ranks as a child process tree, relay rank 0's JSON line, return the child's exit code."""
    port = args.master_port or _free_port()
    child_argv = [a for a in argv if a != "--spawn"]
    if args.gpus == 1 and "--force-collective" not in child_argv:
        child_argv.append("--force-collective")                  # N = 1 through the launcher crosses the RCCL code too
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child_argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out, err = proc.communicate()
    relayed = False
    for line in out.splitlines():
        s = line.strip()
        if s.startswith("{") and s.endswith("}"):
            try:
                json.loads(s)
            except ValueError:
                continue
            print(s, flush=True)
            relayed = True
    if err:
        sys.stderr.write(err[-8000:])
    if proc.returncode != 0:
        sys.stderr.write(f"bench.py launcher: child exited with {proc.returncode}\n")
        return proc.returncode
    if not relayed:
        sys.stderr.write("bench.py launcher: no JSON line from rank 0\n" + out[-4000:])
        return 1
    return 0


def launcher_selftest(args):
    """Rank body of --launcher-selftest: process-group setup + one all-reduce on the CPU (gloo), no scan."""
    import torch
    from audio_mps_amd.parallel import DataParallel
    dp = DataParallel(backend="gloo")
    flat = torch.full((5,), float(dp.rank + 1))
    host, count = dp.allreduce_sums(flat, dp.rank + 10)
    dp.barrier()
    if dp.rank == 0:
        print(json.dumps({"launcher_selftest": True, "world_size": dp.world_size, "requested": args.gpus,
                          "allreduce_sum": float(host[0]), "clip_count": count}), flush=True)
    dp.close()


# ---------------------------------------------------------------------------------------------------
# CPU side (checker and reported baseline)
# ---------------------------------------------------------------------------------------------------
def host_cores() -> int:
    # the GPU box gives one GPU's share of the host: use the affinity mask, capped at 16 threads
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))


def oracle_model(model):
    """The model's raw variables restated for the oracle, whose OWN a1/a2 (effective parameters, psi_0) are then used."""
    from oracle import cmps_oracle as O
    v = model.variables
    ohp = O.HParams(**model.hparams.values())
    ov = O.Variables(np.asarray(v["A"], dtype=np.float32), v["Rx"].copy(), v["Ry"].copy(), v["freqs"].copy(),
                     v["psi_x"].copy(), v["psi_y"].copy(),
                     scaled_R=float(model._c_r) != 1.0, scaled_freqs=float(model._c_h) != 1.0)
    return ohp, ov


def cpu_baseline(model, D, T, seed, clips_arg):
    """Times oracle/cmps_oracle.c (kind 'port': the reference itself is TensorFlow 1.x and cannot run here)
    on a bounded sample of the same workload: forward + backward, float32, OpenMP over clips.  Returns the
    baseline record, the sample and the oracle's outputs on it (for parity_in_bench / precision_ab)."""
    from oracle import cmps_oracle as O, c_oracle as C
    cores = host_cores()
    ohp, ov = oracle_model(model)
    R, f, _, _ = O.effective_params(ohp, ov)
    p0 = O.psi_0(ov)
    clips = clips_arg if clips_arg > 0 else 8 * cores
    data = make_audio_host(clips, T, ohp.delta_t, seed)
    C.psi_scan(data[:cores], R, f, p0, float(ov.A), ohp.delta_t, ohp.sigma, "f32", want_grad=True, nthreads=cores)  # warm
    t0 = time.perf_counter()
    out = C.psi_scan(data, R, f, p0, float(ov.A), ohp.delta_t, ohp.sigma, "f32", want_grad=True, nthreads=cores)
    dt = time.perf_counter() - t0
    assert np.all(np.isfinite(out["loss_per_clip"]))
    rec = {"value": clips * T / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"{clips} clips of T={T}, D={D}, fwd+bwd, float32, {cores} OpenMP threads, {dt:.2f} s"}
    return rec, data, out


def cpu_reference_style():
    """SURVEY 8(d) baseline A: the batched numpy restatement (oracle/cmps_oracle.py: one small op group per scan step over
    the whole batch, the execution style of tf.foldl at model.py:265), forward + backward, float32, at BASELINE C1 and C2
    in full and C3 on 64 clips (linear in B).  Next to it the tight C port at C1 / C2."""
    from oracle import cmps_oracle as O, c_oracle as C
    cores = host_cores()
    rows = []
    for name, D, T, B, Bfull in (("C1", 4, 256, 8, 8), ("C2", 16, 4096, 256, 256), ("C3", 32, 16000, 64, 1024)):
        hp = O.HParams(minibatch_size=B, bond_dim=D)
        var = O.init_variables(hp, seed=0)
        data = make_audio_host(B, T, hp.delta_t, seed=7)
        t0 = time.perf_counter()
        g = O.psi_loss_and_grads(hp, var, data, "f32")
        dt = time.perf_counter() - t0
        assert np.isfinite(float(g.loss))
        row = {"config": name, "kind": "port (numpy, batched per scan step like the TF graph)", "value": B * T / dt,
               "unit": "samples/s", "sample": f"D={D}, T={T}, {B} of {Bfull} clips, fwd+bwd, float32, {dt:.2f} s",
               "threads": "numpy default (small-matrix einsum: effectively 1)"}
        rows.append(row)
        if name != "C3":
            R, f, _, _ = O.effective_params(hp, var)
            p0 = O.psi_0(var)
            t0 = time.perf_counter()
            C.psi_scan(data, R, f, p0, float(var.A), hp.delta_t, hp.sigma, "f32", want_grad=True, nthreads=cores)
            dtc = time.perf_counter() - t0
            rows.append({"config": name, "kind": "port (C, OpenMP over clips)", "value": B * T / dtc, "unit": "samples/s",
                         "sample": f"D={D}, T={T}, {B} clips, fwd+bwd, float32, {cores} threads, {dtc:.3f} s"})
    return rows


def profiled_traffic(kernel, D, T, B):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary OF THIS WORKLOAD
    (profiles/*pmc_summary.json: FETCH_SIZE and WRITE_SIZE collected in separate passes of this same command, FETCH doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950).  None if no such profile is committed."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), key=os.path.getmtime)
    for path in reversed(files):
        try:
            with open(path) as fh:
                doc = json.load(fh)
            if f"D={D}, T={T}, B={B} " not in doc.get("workload", "") + " ":
                continue
            for name, c in doc["kernels"].items():
                if re.search(r"\b" + re.escape(kernel) + r"\b", name):
                    dv = c["derived"]
                    return {"bytes": dv["hbm_read_bytes_per_launch_corrected"] + dv["hbm_write_bytes_per_launch"],
                            "source": os.path.relpath(path, ROOT)}
        except Exception:
            continue
    return None


def make_audio_host(B, T, delta_t, seed, noise=0.02):
    from audio_mps_amd.data import damped_sine
    rng = np.random.default_rng(seed + 12345)
    x = damped_sine(B, T, delta_t, seed=seed)
    return (x + noise * rng.standard_normal(x.shape)).astype(np.float32)


def rel_inf(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


BASELINE_CONFIGS = {(4, 256, 8): "BASELINE configs[0] shape", (16, 4096, 256): "BASELINE configs[1]",
                    (32, 16000, 1024): "BASELINE configs[2]", (128, 16000, 512): "BASELINE configs[4]"}


# ---------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------
def worker(ARGS):
    import torch
    from audio_mps_amd import HParams, PsiCMPS, _capi
    from audio_mps_amd.parallel import DataParallel
    from audio_mps_amd.scan import HipScan, unpack_grad
    from audio_mps_amd.train import Trainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != ARGS.gpus:
        raise SystemExit(f"bench.py: --gpus {ARGS.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if ARGS.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp = DataParallel(backend="gloo" if ARGS.rehearse_on_one_gpu else None, device=dev, force_group=ARGS.force_collective)
    dp.time_collective = True

    D, T, B = ARGS.bond_dim, ARGS.T, ARGS.batch_per_gpu
    hp = HParams(minibatch_size=B * world, bond_dim=D)          # train.py:41-43 defaults otherwise
    config_id = 3
    audio_host = make_audio_host(B, T, hp.delta_t, seed=1000 * config_id + rank)
    audio = torch.from_numpy(audio_host).to(dev)                 # resident in HBM before the timed region
    backend = HipScan(D, device=dev, variant=ARGS.variant, rank1=RANK1_MODES[ARGS.rank1])
    model = PsiCMPS(hp, seed=0, backend=backend)
    trainer = Trainer(model, hp, dp)
    wave = backend.variant in (_capi.CMPS_VARIANT_WAVE, _capi.CMPS_VARIANT_WAVE32)
    pair = backend.variant == _capi.CMPS_VARIANT_PAIR

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def step(fwd_ms=None, bwd_ms=None):
        # same sequence as Trainer.step, with HIP events around the two scan launches (on the stream they are launched on:
        # torch's current stream is the one handed to the C ABI)
        be = model._get_backend()
        be.set_params(model.effective_params(), B, T, train=True)
        ev[0].record()
        be.forward(audio, save_for_bwd=True)
        ev[1].record()
        flat = be.backward()
        ev[2].record()
        host, b_global = dp.allreduce_sums(flat, B)              # D2H copy synchronises the stream
        total, grads = model.chain_rule(host, b_global, with_reg=True)
        trainer.opt.apply_gradients(model.variables, grads)
        if fwd_ms is not None:
            fwd_ms.append(ev[0].elapsed_time(ev[1]))
            bwd_ms.append(ev[1].elapsed_time(ev[2]))
        return host[-1] / b_global

    def timed_run(steps, warmup):
        fwd_ms, bwd_ms = [], []
        for _ in range(warmup):
            step()
        dp.collective_ms.clear()
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = 0.0
        for _ in range(steps):
            last = step(fwd_ms, bwd_ms)
        torch.cuda.synchronize()
        local = time.perf_counter() - t0
        dp.barrier()
        elapsed = dp.max_over_ranks(time.perf_counter() - t0)
        return elapsed, local, last, float(np.mean(fwd_ms)) * 1e-3, float(np.mean(bwd_ms)) * 1e-3

    elapsed, local_elapsed, last, t_fwd, t_bwd = timed_run(ARGS.steps, ARGS.warmup)
    if not np.isfinite(last):
        raise SystemExit(f"non-finite loss {last}")
    allreduce_us = dp.collective_us()                            # mean HIP-event time of the collective (None at N = 1)
    per_rank_ms = dp.gather_floats(1e3 * local_elapsed / ARGS.steps)
    rccl_world = dp.measured_world_size()                        # dist.get_world_size() after a GPU all-reduce of ones

    out = None
    if rank == 0:
        N = T - 1
        ms_per_step = 1e3 * elapsed / ARGS.steps
        value = world * B * T * ARGS.steps / elapsed
        flops_bwd = 56.0 * D * D * B * N                         # 56 D^2 per (clip, sample)
        flops_fwd = 24.0 * D * D * B * N
        bytes_alg = 8.0 * B * T                                  # 4 B read forward + 4 B read in the reverse sweep
        wave16 = wave and D <= 16 and backend.variant == _capi.CMPS_VARIANT_WAVE
        kern = {"fwd": {"name": "k_fwd_wave2 (forward scan: chain wave + loss wave on the matrix cores)" if wave else "k_fwd_block",
                        "t": t_fwd, "flops": flops_fwd, "pmc": "k_fwd_wave2" if wave else "k_fwd_block"},
                "bwd": {"name": "k_bwd_wave (reverse scan)" if wave else "k_bwd_block", "t": t_bwd, "flops": flops_bwd,
                        "pmc": "k_bwd_wave" if wave else "k_bwd_block"}}
        if wave16:
            kern["fwd"].update(name="k_fwd_wave16 (forward scan, 16-row layout: chain wave + loss wave)", pmc="k_fwd_wave16")
            kern["bwd"].update(name="k_bwd_wave16 (reverse scan, 16-row layout: chain wave + gradient wave)", pmc="k_bwd_wave16")
        if pair:                                                 # D = 128: MFMA pair kernels (bf16 operands, fp32 accumulate)
            kern["fwd"].update(name="k_fwd_pair (forward scan: 4x4x4 bf16 MFMA chain waves + 32x32x16 loss waves, eight steps per tile)", pmc="k_fwd_pair")
            kern["bwd"].update(name="k_bwd_pair + k_grad_pair (reverse scan + streaming gradient GEMM)", pmc="k_bwd_pair")
        dom = "fwd" if t_fwd >= t_bwd else "bwd"                 # the dominant kernel = the longer launch
        oth = "bwd" if dom == "fwd" else "fwd"
        traffic = profiled_traffic(kern[dom]["pmc"], D, T, B)
        ach = kern[dom]["flops"] / kern[dom]["t"] / 1e12
        peak = BF16_PEAK_TFLOPS if pair else FP32_PEAK_TFLOPS
        whole = (flops_fwd + flops_bwd) / (1e-3 * ms_per_step) / 1e12
        rank1 = backend.rank1
        roofline = {
            "bound": "compute: fp32 VALU issue (instruction-issue / latency bound scan)" if not pair
                     else "compute: bf16 MFMA + per-step VALU/LDS tail",
            "bound_class": "mfma",                               # the contract's class for a compute-bound kernel (not 'hbm')
            "kernel": kern[dom]["name"], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
            "frac": ach / peak,
            "traffic": traffic["bytes"] if traffic else None, "traffic_source": traffic["source"] if traffic else None,
            "algorithmic_bytes": 4.0 * B * T, "launch_ms": kern[dom]["t"] * 1e3,
            "flops_per_launch": kern[dom]["flops"],
            "whole_step": {"flops": flops_fwd + flops_bwd, "achieved": whole, "frac": whole / peak,
                           "note": "80 D^2 algorithmic flop per (clip, sample) over the full optimiser step (host work included)"},
            "note": "fp32 path: peak = 157.3 TFLOP/s (fp32 vector peak = f32-input MFMA peak); achieved = SURVEY 8(d) "
                    "ALGORITHMIC flops (24 D^2 forward, 56 D^2 backward per clip-sample) / launch time, not executed flops: the "
                    "reverse scan executes two mat-vecs on the VALU (H y is stashed by the forward) and its rank-1 updates on "
                    f"the matrix pipe ({RANK1_LABEL[rank1]}); the scan is instruction-issue/latency bound, not HBM bound "
                    "(10 D^2 flop per algorithmic byte)",
            "other_kernel": {"kernel": kern[oth]["name"], "achieved": kern[oth]["flops"] / kern[oth]["t"] / 1e12,
                             "frac": kern[oth]["flops"] / kern[oth]["t"] / 1e12 / peak,
                             "launch_ms": kern[oth]["t"] * 1e3},
            "hbm": {"achieved": bytes_alg / (t_fwd + t_bwd) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": bytes_alg / (t_fwd + t_bwd) / 1e9 / HBM_PEAK_GBS}}
        if pair:
            roofline["note"] = ("bf16-operand path (BASELINE configs[4]): peak = 2500 TFLOP/s dense bf16 MFMA; the mat-vecs use "
                                "v_mfma_f32_4x4x4_16b_bf16 (two clips x {re, im} fill its four B columns), whose own ceiling is "
                                "256 flop/cycle/SIMD = 629 TFLOP/s; the scans are per-step latency / issue bound")
        cfg_name = BASELINE_CONFIGS.get((D, T, B), "custom shape")
        if world > 1 and (D, T, B) == (32, 16000, 1024):
            cfg_name = "BASELINE configs[3] (configs[2] per GPU)" if world == 8 else f"BASELINE configs[2] per GPU x {world}"
        if pair:
            dtype = "bf16 (mat-vec operands; fp32 state and accumulate)"
        elif wave:
            dtype = "f32" if (rank1 != 1 or wave16) else "f32 (rank-1 gradient updates: bf16x2 split, 16 operand bits)"
        else:
            dtype = "f32"
        out = {
            "metric": f"audio samples/sec (fwd+bwd) at D={D}, T={T}",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": ARGS.steps, "warmup": ARGS.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{cfg_name}: PsiCMPS fwd+bwd scan, D={D}, T={T}, batch {B} per GPU"
                                   f" (global {B * world}), damped sine + noise, full optimiser step",
                       "parallelism": f"dp{world}", "kernel_variant": int(backend.variant),
                       "rank1_updates": ("exact fp32 MFMA (16-row layout)" if wave16 else RANK1_LABEL[rank1]) if wave else None},
            "roofline": roofline,
            "final_loss": float(last),
            "per_rank_ms_per_step": {"min": float(np.min(per_rank_ms)), "max": float(np.max(per_rank_ms))},
            "rccl_world_size": rccl_world,
            "allreduce_us": allreduce_us,
            "collective_backend": dp.backend,                    # "nccl" (= RCCL), "gloo" (rehearsal) or None (N = 1 in process)
        }
        if ARGS.rehearse_on_one_gpu:
            out["rehearsal"] = "all ranks share cuda:0 and reduce over gloo on the host: NOT a scaling measurement"

    # ---- outside the timed region: CPU baseline, parity of this very workload, precision A/B (rank 0, N = 1 only) ----
    if rank == 0 and world == 1 and not ARGS.no_cpu_baseline:
        cb, sample, ref = cpu_baseline(model, D, T, seed=1000 * config_id, clips_arg=ARGS.cpu_clips)
        out["cpu_baseline"] = cb
        from oracle import c_oracle as C
        gr = C.unpack_grad(ref["grad"], D)
        d_sample = torch.from_numpy(sample).to(dev)
        Bs = sample.shape[0]

        def gpu_on_sample():
            # the sample may hold more clips than the workload's batch (BASELINE C1 has 8): chunks of at most B clips, losses
            # concatenated, gradient sums added
            backend.set_params(model.effective_params(), B, T, train=True)
            pers, flat = [], None
            for s0 in range(0, Bs, B):
                chunk = d_sample[s0:s0 + B].contiguous()
                pers.append(backend.forward(chunk, save_for_bwd=True).cpu().numpy().copy())
                gch = backend.backward().cpu().numpy().astype(np.float64)
                flat = gch if flat is None else flat + gch
            per = np.concatenate(pers)
            g = unpack_grad(flat.astype(np.float32), D)
            loss_err = float(np.max(np.abs(per - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1.0)))
            gerr = {k: rel_inf(g[k], gr[k]) for k in ("Rbar", "fbar", "psi0bar", "Abar")}
            return loss_err, gerr

        loss_err, gerr = gpu_on_sample()
        tol_l, tol_g = (2e-3, 3e-2) if pair else (1e-5, 1e-4)
        out["parity_in_bench"] = {"clips": Bs, "max_rel_loss_err": loss_err, "max_rel_grad_err": max(gerr.values()),
                                  "grad_err_by_tensor": gerr, "tolerance": {"loss": tol_l, "grad": tol_g},
                                  "ok": bool(loss_err <= tol_l and max(gerr.values()) <= tol_g),
                                  "against": "oracle/cmps_oracle.c float32 (parity UNPINNED: no reference-held vectors exist)"}
        if wave and not wave16 and not ARGS.no_precision_ab:
            ab = {}
            for name, mode in RANK1_MODES.items():               # accuracy first: the timed steps below move the parameters
                backend.set_rank1(mode)
                _, ge = gpu_on_sample()
                ab[name] = {"max_rel_grad_err_vs_f32_oracle": max(ge.values()), "products": RANK1_LABEL[mode]}
            for name, mode in RANK1_MODES.items():
                backend.set_rank1(mode)
                el, _, _, tf_, tb_ = timed_run(5, 1)
                ab[name].update({"ms_per_step": 1e3 * el / 5, "samples_per_s": B * T * 5 / el, "bwd_ms": tb_ * 1e3, "fwd_ms": tf_ * 1e3})
            backend.set_rank1(RANK1_MODES[ARGS.rank1])
            out["precision_ab"] = {"what": "rank-1 gradient updates of k_bwd_wave; everything else is identical fp32 code",
                                   "headline_mode": ARGS.rank1, "modes": ab}
        if (D, T, B) == (32, 16000, 1024):
            out["cpu_baseline"]["reference_style"] = cpu_reference_style()
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    dp.barrier()
    dp.close()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(launch_children(args, argv))                    # parent: no GPU call has been made, none will be
    if args.launcher_selftest:
        launcher_selftest(args)
        return
    worker(args)


if __name__ == "__main__":
    main()
