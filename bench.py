#!/usr/bin/env python3
"""Benchmark of the hot path: audio samples/s for forward + backward of the PsiCMPS scan.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 with WORLD_SIZE unset makes this process a LAUNCHER: before anything touches
the GPU it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py`
as a child process (one rank per GPU over RCCL), relays rank 0's JSON line and exits with the child's code
(`--spawn` forces the same path at N = 1, with a one-rank RCCL communicator).  Under torch.distributed.run (WORLD_SIZE set)
it is a rank.

A "step" is one complete optimiser step on one batch of synthetic audio already resident in HBM, exactly
`audio_mps_amd.train.Trainer.step`: parameter tables, forward scan, reverse scan, slab reduction, (one RCCL all-reduce of the
gradient sums when N > 1), chain rule + regularisers + Adam.  By default the optimiser half runs on the device
(cmps_psi_apply_step: no device -> host copy inside a step); `--host-optimizer` runs it in numpy on the host (the tested
reference implementation of that half).  Default workload at every N: BASELINE.json configs[2] per GPU -- D=32, T=16000,
batch 1024 per GPU (configs[3] = the same per GPU on 8 GPUs), so scaling is weak.
Inputs (`--input`): the reference's damped sine (data.py:8-22; SURVEY 8(d)'s generator, the default), the same plus white noise,
or SURVEY 8(d)'s band-limited random walk; parameters by the reference's initialisation rules (model.py:36-39, 49, 218-219) with
train.py:41-43 hyper-parameters, seed 0 (D > 64: SURVEY 8(d)'s scaled R_in, which keeps 1 + e x / A positive).

Output (rank 0).  The LAST stdout line is the contract's record: ONE compact JSON object (< 4 KB; `HEADLINE_LIMIT`) with
metric / value / config, `roofline` for the dominant kernel (the longer of the two scan launches: `achieved` / `frac` are the
contract's ALGORITHMIC flops -- SURVEY 8(d): 24 D^2 forward, 56 D^2 backward per clip-sample -- over the launch time measured with
HIP events on the launch stream), `cpu_baseline` (the plain-C restatement oracle/cmps_oracle.c timed on this host's cores on a
bounded sample) and `parity_in_bench` (the GPU results on that same sample against the oracle's, outside the timed region).
Everything else goes out EARLIER, one JSON object per line, each of the form {"detail": name, "data": ...} (never a top-level
"metric" key), and as sidecar files under gpurun_out/bench_detail/ when that directory can be written: `roofline_detail` (per-kernel
/ per-pipe records, executed split, step traffic), `precision_ab` (the arithmetic modes of the rank-1 gradient sums side by side),
`reference_style` (numpy restatement timings), `other_config:<name>` (BASELINE configs[0] shape, [1], [4] in float32 and as named,
a few steps each with their own parity check; the headline keeps a one-line summary of each).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
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
RANK1_MODES = {"exact_f32": 0, "bf16x2": 1, "bf16x3": 2, "f16x2": 3, "default": 4}
RANK1_LABEL = {0: "exact fp32 MFMA", 1: "bf16x2 split (16 operand bits), fp32 accumulate",
               2: "bf16x3 split (24 operand bits, fp32-faithful products), fp32 accumulate",
               3: "f16x2 split (power-of-two scaled operands, 11+1+11+1 operand bits: bf16x3's accuracy class), fp32 accumulate"}
V_BLOCK, V_WAVE, V_PAIR, V_WAVE32, V_WIDE = 1, 2, 3, 4, 5


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--bond-dim", type=int, default=32)
    p.add_argument("--T", type=int, default=16000)
    p.add_argument("--batch-per-gpu", type=int, default=1024)
    p.add_argument("--variant", type=int, default=0,
                   help="0 auto (float32: wave-per-clip kernels for D <= 32, wide kernels above), 1 block-per-clip, 2 wave-per-clip, "
                        "3 MFMA pair kernels (32 < D <= 128; bf16 operands), 5 float32 wide kernels (32 < D <= 128)")
    p.add_argument("--rank1", choices=sorted(RANK1_MODES), default="default",
                   help="arithmetic of the rank-1 gradient sums (cmps_set_option): wave reverse scan of 17 <= D <= 32, gradient GEMM of "
                        "the wide kernels")
    p.add_argument("--input", choices=["damped_sine", "damped_sine_noise", "bandlimited"], default="damped_sine",
                   help="synthetic input (SURVEY 8(d)): the reference's damped sine (data.py:8-22), the same + 0.02 white noise, or "
                        "0.1 cumsum(N(0,1)) / sqrt(T) clipped")
    p.add_argument("--host-optimizer", action="store_true",
                   help="chain rule + Adam in numpy on the host (D2H of the gradient sums every step) instead of cmps_psi_apply_step")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-precision-ab", action="store_true")
    p.add_argument("--no-other-configs", action="store_true")
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


def oracle_on(model, data, nthreads):
    from oracle import cmps_oracle as O, c_oracle as C
    ohp, ov = oracle_model(model)
    R, f, _, _ = O.effective_params(ohp, ov)
    p0 = O.psi_0(ov)
    return C.psi_scan(data, R, f, p0, float(ov.A), ohp.delta_t, ohp.sigma, "f32", want_grad=True, nthreads=nthreads)


def cpu_baseline(model, D, T, kind, seed, clips_arg):
    """Times oracle/cmps_oracle.c (kind 'port': the reference itself is TensorFlow 1.x and cannot run here)
    on a bounded sample of the same workload: forward + backward, float32, OpenMP over clips.  Returns the
    baseline record, the sample and the oracle's outputs on it (for parity_in_bench / precision_ab)."""
    cores = host_cores()
    clips = clips_arg if clips_arg > 0 else 8 * cores
    data = make_audio_host(kind, clips, T, model.hparams.delta_t, seed)
    oracle_on(model, data[:cores], cores)                        # warm
    t0 = time.perf_counter()
    out = oracle_on(model, data, cores)
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
        data = make_audio_host("damped_sine_noise", B, T, hp.delta_t, seed=7)
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


def make_audio_host(kind, B, T, delta_t, seed):
    from audio_mps_amd.data import synthetic_audio
    return synthetic_audio(kind, B, T, delta_t, seed)


def rel_inf(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


# ---------------------------------------------------------------------------------------------------
# output: detail lines first, ONE compact headline object as the last stdout line
#   (round 4's single 23.5 KB line did not fit the driver's record: BENCH_r04.json.parsed = null)
# ---------------------------------------------------------------------------------------------------
HEADLINE_LIMIT = 4000          # bytes of the last line, asserted by tests/test_host.py and tests/test_gpu_train.py
DETAIL_DIR = os.environ.get("CMPS_BENCH_DETAIL_DIR", os.path.join(ROOT, "gpurun_out", "bench_detail"))
ROOFLINE_HEADLINE_KEYS = ("bound", "kernel", "pipe", "achieved", "peak", "unit", "frac", "frac_is", "traffic", "traffic_source",
                          "algorithmic_bytes", "launch_ms", "flops_per_launch", "achieved_algorithmic", "frac_algorithmic",
                          "whole_step_frac_algorithmic")
HEADLINE_DROP_ORDER = ("detail_lines", "other_configs", "per_rank_ms_per_step", "collective")     # if the line were ever too long


def _jsonable(o):
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    if isinstance(o, np.ndarray):
        return o.tolist()
    raise TypeError(type(o).__name__)


def emit_detail(name, data, stream=None):
    """One earlier stdout line {"detail": name, "data": data} + a sidecar file (best effort).  Returns the name."""
    line = json.dumps({"detail": name, "data": data}, default=_jsonable)
    assert not line.startswith('{"metric"')
    print(line, file=stream or sys.stdout, flush=True)
    try:
        os.makedirs(DETAIL_DIR, exist_ok=True)
        with open(os.path.join(DETAIL_DIR, re.sub(r"[^A-Za-z0-9_.-]+", "_", name) + ".json"), "w") as fh:
            fh.write(line + "\n")
    except OSError:
        pass
    return name


def short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 1] + "~"


def compact_roofline(full):
    """The contract's roofline object without the per-kernel tables (those are the `roofline_detail` line)."""
    if full is None:
        return None
    rec = {k: full[k] for k in ROOFLINE_HEADLINE_KEYS if k in full}
    for k, n in (("frac_is", 150), ("kernel", 60), ("pipe", 60)):
        if rec.get(k) is not None:
            rec[k] = short(rec[k], n)
    return rec


def compact_parity(par):
    if par is None:
        return None
    return {"clips": par["clips"], "max_rel_loss_err": par["max_rel_loss_err"], "max_rel_grad_err": par["max_rel_grad_err"],
            "tolerance": par["tolerance"], "ok": par["ok"], "against": short(par["against"], 90)}


def compact_other_config(row):
    if "error" in row:
        return {"key": row.get("key"), "config": short(row["config"], 40), "error": short(row["error"], 120)}
    par = row.get("parity_in_bench") or {}
    return {"key": row.get("key"), "config": short(row["config"], 40), "ms_per_step": round(row["ms_per_step"], 4), "value": float(f"{row['value']:.5g}"),
            "dtype": short(row["dtype"], 12), "dominant_kernel": short(row["dominant_kernel"], 20),
            "frac": None if row.get("frac") is None else round(row["frac"], 4),
            "parity_ok": par.get("ok"), "loss_err": par.get("max_rel_loss_err"), "grad_err": par.get("max_rel_grad_err")}


def _round_floats(o, digits):
    if isinstance(o, (float, np.floating)):
        return float(f"{float(o):.{digits}g}") if np.isfinite(o) else float(o)
    if isinstance(o, dict):
        return {k: _round_floats(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round_floats(v, digits) for v in o]
    return o


def headline_line(out):
    """Serialise the headline object (top-level numbers to 10, nested ones to 6 significant digits); should it ever exceed
    HEADLINE_LIMIT, optional keys go (in HEADLINE_DROP_ORDER) -- the contract's keys never do."""
    out = {k: _round_floats(v, 6 if isinstance(v, (dict, list, tuple)) else 10) for k, v in out.items()}
    line = json.dumps(out, default=_jsonable)
    for key in HEADLINE_DROP_ORDER:
        if len(line) <= HEADLINE_LIMIT:
            break
        if key in out:
            out.pop(key)
            out["dropped_for_size"] = out.get("dropped_for_size", []) + [key]
            line = json.dumps(out, default=_jsonable)
    if len(line) > HEADLINE_LIMIT:
        raise RuntimeError(f"bench.py: headline line is {len(line)} bytes (> {HEADLINE_LIMIT})")
    return line


# ---------------------------------------------------------------------------------------------------
# committed rocprofv3 summaries (profiles/*pmc_summary.json: scripts/profile_scan.sh + scripts/summarize_prof.py)
# ---------------------------------------------------------------------------------------------------
def profile_summary(D, T, B, variant, wide_chain=1):
    """The newest committed PMC summary of THIS workload and kernel family (None if there is none)."""
    want = {V_WAVE: "k_bwd_wave", V_WAVE32: "k_bwd_wave", V_PAIR: "k_bwd_pair", V_WIDE: "k_bwd_wide" if wide_chain == 0 else "k_bwd_chain16",
            V_BLOCK: "k_bwd_block"}[variant]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")))
    for path in reversed(files):                                 # names carry the round: the last one is the newest
        try:
            with open(path) as fh:
                doc = json.load(fh)
            if f"D={D}, T={T}, B={B} " not in doc.get("workload", "") + " ":
                continue
            if not any(want in name for name in doc["kernels"]):
                continue
            doc["_source"] = os.path.relpath(path, ROOT)
            return doc
        except Exception:
            continue
    return None


def kernel_counters(doc, kernel):
    if doc is None:
        return None
    for name, c in doc["kernels"].items():
        if re.search(r"\b" + re.escape(kernel) + r"\b", name):
            return c
    return None


# ---------------------------------------------------------------------------------------------------
# what a kernel family executes per (clip, sample), in units of D^2 flop (fp32-equivalent), by pipe
#   algorithmic: forward 24 = Q u (8) + s R u (8) + R y (8); backward 56 = R^dagger y, R ybar, R^dagger R ybar,
#   R^dagger ybar (4 x 8) + three rank-1 sums (3 x 8)
# ---------------------------------------------------------------------------------------------------
def executed_split(variant, D, rank1, wide_chain=1):
    if variant in (V_WAVE, V_WAVE32) and not (variant == V_WAVE and D <= 16):
        prod = {0: 1, 1: 3, 2: 6, 3: 3}[rank1]
        return {"fwd": {"valu_fp32": 12, "mfma_fp32_equiv": 8, "mfma_products": 3 if rank1 == 3 else 6, "eliminated": 4,
                        "what": "merged (Q + s R) u mat-vec 8 + forming it 4 on the VALU; H y as a split-operand GEMM over 32-step chunks "
                                "(f16x2: 3 products; bf16x3: 6)"},
                "bwd": {"valu_fp32": 12, "mfma_fp32_equiv": 24, "mfma_products": prod, "eliminated": 20,
                        "what": "merged (Q + s R^dagger) ybar 8 + forming it 4 on the VALU; H y read from the forward's stash (16 "
                                "eliminated), the merge (4 eliminated); three rank-1 sums on the matrix cores"}}
    if variant == V_WAVE:                                        # 16-row layout
        return {"fwd": {"valu_fp32": 20, "mfma_fp32_equiv": 0, "mfma_products": 0, "eliminated": 4, "what": "chain wave 12, loss wave 8 (VALU)"},
                "bwd": {"valu_fp32": 12, "mfma_fp32_equiv": 24, "mfma_products": 1, "eliminated": 20,
                        "what": "merged mat-vec on the VALU; rank-1 sums as exact fp32 MFMAs (16x16x4) on the gradient wave"}}
    if variant == V_WIDE:
        prod = 6 if rank1 == 2 else 3
        hyp = 6 if rank1 in (1, 2) else 3
        fwd_mfma, bwd_mfma = wide_chain in (1, 2), wide_chain == 1
        return {"fwd": {"valu_fp32": 0 if fwd_mfma else 12, "mfma_fp32_equiv": 24 if fwd_mfma else 8, "mfma_products": 3 if fwd_mfma else hyp,
                        "eliminated": 0 if fwd_mfma else 4,
                        "what": ("k_fwd_chain16: Q u and s R u as f16x2-split operands on the matrix cores (three products; the identity part and "
                                 "everything behind the mat-vec in float32)" if fwd_mfma else
                                 "k_fwd_wide: merged (Q + s R) u as fp32 v_pk_fma chains, 8 + 4 forming it") +
                                "; k_hy_wide: H y for all (clip, step) pairs as a split-operand GEMM; k_loss_wide: the sequential float32 loss sums"},
                "bwd": {"valu_fp32": 0 if bwd_mfma else 12, "mfma_fp32_equiv": 40 if bwd_mfma else 24, "mfma_products": prod,
                        "eliminated": 16 if bwd_mfma else 20,
                        "what": ("k_bwd_chain16: Q ybar and s R^dagger ybar as f16x2-split operands on the matrix cores" if bwd_mfma else
                                 "k_bwd_wide: merged (Q + s R^dagger) ybar on the VALU") +
                                "; k_grad_gemm: rank-1 sums as split-operand GEMMs (three bf16, two bf16 or two fp16 pieces); H y from the stash"}}
    if variant == V_PAIR:
        return {"fwd": {"valu_fp32": 0, "mfma_fp32_equiv": 24, "mfma_products": 1, "eliminated": 0, "what": "R u, Q u (16x16x32 bf16), H y (32x32x16 bf16)"},
                "bwd": {"valu_fp32": 0, "mfma_fp32_equiv": 40, "mfma_products": 1, "eliminated": 16,
                        "what": "Q ybar, R^dagger ybar (16x16x32 bf16) + the gradient GEMM; H y from the stash"}}
    return {"fwd": {"valu_fp32": 24, "mfma_fp32_equiv": 0, "mfma_products": 0, "eliminated": 0, "what": "three fp32 mat-vecs"},
            "bwd": {"valu_fp32": 56, "mfma_fp32_equiv": 0, "mfma_products": 0, "eliminated": 0, "what": "four fp32 mat-vecs + three rank-1 updates"}}


KERNEL_NAMES = {
    "wave": ("k_fwd_wave2", "k_fwd_wave2 (forward scan: chain wave + loss wave on the matrix cores)", "k_bwd_wave", "k_bwd_wave (reverse scan)"),
    "wave2w": ("k_fwd_wave2", "k_fwd_wave2 (forward scan: chain wave + loss wave on the matrix cores)", "k_bwd_wave2w",
               "k_bwd_wave2w (reverse scan: chain wave + gradient wave per clip)"),
    "wave16": ("k_fwd_wave16", "k_fwd_wave16 (forward scan, 16-row layout: chain wave + loss wave)", "k_bwd_wave16",
               "k_bwd_wave16 (reverse scan, 16-row layout: chain wave + gradient wave)"),
    "pair": ("k_fwd_pair", "k_fwd_pair (forward scan: 16x16x32 bf16 MFMA chain waves + 32x32x16 loss waves, eight steps per tile)", "k_bwd_pair",
             "k_bwd_pair + k_grad_gemm<1 piece> (reverse scan + streaming gradient GEMM)"),
    "wide": ("k_fwd_wide", "k_fwd_wide + k_hy_wide + k_loss_wide (float32 forward chain on the VALU, R / Q register resident; H y as a split-operand GEMM)",
             "k_bwd_wide", "k_bwd_wide + k_grad_gemm (float32 reverse scan on the VALU + split-operand gradient GEMM)"),
    "wide_mfma": ("k_fwd_chain16", "k_fwd_chain16 + k_hy_wide + k_loss_wide (forward chain: f16x2-split operands on the matrix cores; H y as a split-operand GEMM)",
                  "k_bwd_chain16", "k_bwd_chain16 + k_grad_gemm (reverse chain: f16x2-split operands on the matrix cores + split-operand gradient GEMM)"),
    "wide_mfma_fwd": ("k_fwd_chain16", "k_fwd_chain16 + k_hy_wide + k_loss_wide (forward chain on the matrix cores)",
                      "k_bwd_wide", "k_bwd_wide + k_grad_gemm (reverse chain on the VALU: the A/B setting)"),
    "block": ("k_fwd_block", "k_fwd_block", "k_bwd_block", "k_bwd_block"),
}


def family_of(variant, D):
    if variant == V_WAVE and D <= 16:
        return "wave16"
    return {V_WAVE: "wave", V_WAVE32: "wave", V_PAIR: "pair", V_WIDE: "wide", V_BLOCK: "block"}[variant]


# ---------------------------------------------------------------------------------------------------
# one record per kernel and pipe: what the kernel EXECUTES on that pipe (per (clip, sample), in units of D^2 flop, or bytes), its
# live duration (HIP events around each launch: cmps_set_option(CMPS_OPT_KERNEL_EVENTS), outside the timed region) and the peak of
# THAT pipe -- so no fraction can exceed 1 (56 D^2 over the fp32 vector peak gave 1.32 for a pair of kernels that runs
# most of it on the matrix pipe).  Products = bf16 piece products issued per float32 product (x6 / x3 split, x1 plain bf16).
# ---------------------------------------------------------------------------------------------------
PIPE_PEAK = {"valu_fp32": (FP32_PEAK_TFLOPS, "TFLOP/s"), "mfma_f32": (FP32_PEAK_TFLOPS, "TFLOP/s"),
             "mfma_bf16": (BF16_PEAK_TFLOPS, "TFLOP/s"), "hbm": (HBM_PEAK_GBS, "GB/s")}


def kernel_work_model(fam, D, DP, rank1):
    """{kernel name as cmps_kernel_times reports it: [(pipe, D^2-flop per (clip, sample) | bytes per (clip, sample), what)]}"""
    prod = {0: 1, 1: 3, 2: 6, 3: 3}[rank1]
    rp = "mfma_f32" if rank1 == 0 else "mfma_bf16"
    if fam == "wave":
        hyp = 3 if rank1 == 3 else 6
        return {"k_fwd_wave2": [("valu_fp32", 12 * D * D, "merged (Q + s R) u mat-vec 8 + forming it 4"),
                                ("mfma_bf16", 8 * hyp * D * D, f"H y as a split-operand GEMM over 32-step chunks: {hyp} piece products (f16x2 for "
                                                               "CMPS_RANK1_F16X2 / DEFAULT, bf16x3 otherwise; fp16 and bf16 MFMAs have the same dense peak)"),
                                ("hbm", 512.0, "stash rows written: 512 B per (clip, step)")],
                "k_bwd_wave": [("valu_fp32", 12 * D * D, "merged (Q + s R^dagger) ybar mat-vec 8 + forming it 4"),
                               (rp, 24 * prod * D * D, f"three rank-1 sums, {prod} product(s) per float32 product"),
                               ("hbm", 512.0, "stash rows read")],
                "k_bwd_wave2w": [("valu_fp32", 12 * D * D, "chain wave: merged (Q + s R^dagger) ybar mat-vec 8 + forming it 4"),
                                 ("mfma_bf16", 24 * 3 * D * D, "gradient wave: three rank-1 sums, f16x2-split operands, 3 products per float32 product"),
                                 ("hbm", 512.0, "stash rows read")]}
    if fam == "wave16":
        return {"k_fwd_wave16": [("valu_fp32", 20 * D * D, "chain wave 12 + loss wave 8"), ("hbm", 512.0, "stash rows written")],
                "k_bwd_wave16": [("valu_fp32", 12 * D * D, "merged mat-vec"), ("mfma_f32", 24 * D * D, "rank-1 sums, exact fp32 16x16x4 MFMAs"),
                                 ("hbm", 512.0, "stash rows read")]}
    if fam == "wide":
        gp = 6 if rank1 == 2 else 3
        gname = {1: "k_grad_gemm<2>", 2: "k_grad_gemm<3>", 3: "k_grad_gemm<f16x2>"}[rank1]
        return {"k_fwd_wide": [("valu_fp32", 12 * D * D, "merged (Q + s R) u, v_pk_fma_f32"), ("hbm", 8.0 * DP, "y rows written: 16 DP B per pair-step")],
                "k_fwd_chain16": [("mfma_bf16", 96 * DP * DP, "(Q + s R) u as f16x2-split operands on v_mfma_f32_16x16x32_f16: DP / 32 waves x DP / 16 K-steps x 6 "
                                                                "MFMAs per pair and step (round 5: both pieces of the vector stacked in the 16 A rows, so v R_hi + v R_lo are "
                                                                "the three products and all rows carry data; Q one: the |Q|_F <= 2^-19 instance; 8 otherwise), as issued"),
                                  ("hbm", 8.0 * DP, "y rows written: 16 DP B per pair-step")],
                ("k_hy_wide<3>" if rank1 in (1, 2) else "k_hy_wide<f16x2>"):
                    [("mfma_bf16", 8 * (6 if rank1 in (1, 2) else 3) * D * D,
                      "H y for all (clip, step) pairs: bf16x3 split, 6 piece products (BF16X2 / BF16X3) or f16x2 split, 3 products"),
                     ("hbm", 16.0 * DP, "y rows read + H y rows written")],
                "k_loss_wide": [("hbm", 8.0, "e_k, |y_k|^2 scalars")],
                "k_bwd_wide": [("valu_fp32", 12 * D * D, "merged (Q + s R^dagger) ybar"), ("hbm", 24.0 * DP, "y, H y rows read, ybar rows written")],
                "k_bwd_chain16": [("mfma_bf16", 96 * DP * DP, "(Q + s R^dagger) ybar as f16x2-split operands on v_mfma_f32_16x16x32_f16, as issued (see k_fwd_chain16)"),
                                  ("hbm", 24.0 * DP, "y, H y rows read, ybar rows written")],
                gname: [("mfma_bf16", 24 * gp * D * D, f"three rank-1 sums as GEMMs, {gp} piece products (fp16 and bf16 MFMAs have the same dense peak)"),
                        ("hbm", 16.0 * DP, "y and ybar rows read: 32 DP B per pair-step")]}
    if fam == "pair":
        return {"k_fwd_pair": [("mfma_bf16", 24 * D * D, "R u, Q u (16x16x32, 4 of 16 A rows useful: issued 4 x) + H y (32x32x16)"),
                               ("hbm", 16.0 * DP, "y and H y rows written")],
                "k_bwd_pair": [("mfma_bf16", 16 * D * D, "Q ybar, R^dagger ybar (16x16x32, 4 of 16 A rows useful: issued 4 x)"),
                               ("hbm", 24.0 * DP, "y, H y rows read, ybar rows written")],
                "k_grad_gemm<1>": [("mfma_bf16", 24 * D * D, "three rank-1 sums as bf16 GEMMs"), ("hbm", 16.0 * DP, "y and ybar rows read")]}
    return {"k_fwd_block": [("valu_fp32", 24 * D * D, "three fp32 mat-vecs")], "k_bwd_block": [("valu_fp32", 56 * D * D, "all of it on the VALU")]}


def kernel_records(fam, D, T, B, rank1, ktimes):
    """ktimes: {name: (summed ms, launches)} of backend.kernel_times()."""
    DP = (D + 31) // 32 * 32
    units = float(B) * (T - 1)
    model = kernel_work_model(fam, D, DP, rank1)
    recs = []
    for name, (ms_sum, calls) in ktimes.items():
        ms = ms_sum / max(calls, 1)
        if name not in model:
            recs.append({"kernel": name, "duration_ms": ms, "launches_timed": calls})
            continue
        for pipe, per_unit, what in model[name]:
            peak, unit = PIPE_PEAK[pipe]
            work = per_unit * units
            rate = work / (ms * 1e-3) / (1e9 if pipe == "hbm" else 1e12)
            recs.append({"kernel": name, "pipe": pipe, ("executed_bytes" if pipe == "hbm" else "executed_flop"): work,
                         "duration_ms": ms, "achieved": rate, "peak": peak, "unit": unit, "frac": rate / peak, "what": what,
                         "launches_timed": calls})
    return recs


def roofline_record(D, T, B, variant, rank1, t_fwd, t_bwd, ms_per_step, ktimes=None, wide_chain=1):
    """The contract's roofline object for the dominant kernel + the per-pipe split of what is executed + the step's traffic."""
    N = T - 1
    units = float(B) * N
    fam = family_of(variant, D)
    pmc_f, name_f, pmc_b, name_b = KERNEL_NAMES[fam if fam != "wide" else {0: "wide", 1: "wide_mfma", 2: "wide_mfma_fwd"}[wide_chain]]
    if fam == "wave" and ktimes and "k_bwd_wave2w" in ktimes:      # CMPS_OPT_BWD_WAVES = 2 (the default for the F16X2 sums)
        pmc_f, name_f, pmc_b, name_b = KERNEL_NAMES["wave2w"]
    pair = variant == V_PAIR
    peak = BF16_PEAK_TFLOPS if pair else FP32_PEAK_TFLOPS
    split = executed_split(variant, D, rank1, wide_chain)
    alg = {"fwd": 24.0 * D * D * units, "bwd": 56.0 * D * D * units}
    tt = {"fwd": t_fwd, "bwd": t_bwd}
    dom = "fwd" if t_fwd >= t_bwd else "bwd"
    oth = "bwd" if dom == "fwd" else "fwd"
    doc = profile_summary(D, T, B, variant, wide_chain)

    def pipes(which):
        sp = split[which]
        t = tt[which]
        valu = sp["valu_fp32"] * D * D * units / t / 1e12
        mf_eq = sp["mfma_fp32_equiv"] * D * D * units / t / 1e12
        mf_raw = mf_eq * max(sp["mfma_products"], 1)
        f32_matrix = fam == "wave16" or (fam == "wave" and rank1 == 0 and which == "bwd")
        rec = {"valu_fp32_tflops": valu, "valu_frac_of_fp32_peak": valu / FP32_PEAK_TFLOPS,
               "matrix_pipe_tflops_fp32_equiv": mf_eq, "matrix_pipe_tflops_issued": mf_raw,
               "matrix_pipe_products_per_fp32_product": sp["mfma_products"],
               "matrix_pipe_frac_of_its_peak": mf_raw / (FP32_PEAK_TFLOPS if f32_matrix else BF16_PEAK_TFLOPS),
               "eliminated_frac_of_algorithmic": sp["eliminated"] / (24.0 if which == "fwd" else 56.0),
               "what": sp["what"]}
        c = kernel_counters(doc, pmc_f if which == "fwd" else pmc_b)
        if c is not None and "derived" in c:
            dv = c["derived"]
            wc = dv.get("wave_cycles_per_clip_step")
            if wc:
                rec["pmc"] = {"issue_busy_frac": dv.get("active_inst_any_cycles_per_clip_step", 0.0) / wc,
                              "wait_frac": dv.get("wait_any_cycles_per_clip_step", 0.0) / wc,
                              "issue_stall_frac": dv.get("wait_inst_any_cycles_per_clip_step", 0.0) / wc,
                              "valu_insts_per_clip_step": dv.get("valu_insts_per_clip_step"),
                              "mfma_insts_per_clip_step": dv.get("mfma_insts_per_clip_step"),
                              "mfma_busy_cycles_per_clip_step": dv.get("mfma_busy_cycles_per_clip_step"),
                              "wave_cycles_per_clip_step": wc, "source": doc["_source"]}
        return rec

    traffic = None
    step_traffic = None
    if doc is not None:
        c = kernel_counters(doc, pmc_f if dom == "fwd" else pmc_b)
        if c is not None:
            dv = c["derived"]
            traffic = dv.get("hbm_read_bytes_per_launch_corrected", 0.0) + dv.get("hbm_write_bytes_per_launch", 0.0)
        tot = 0.0
        per = {}
        for name, c in doc["kernels"].items():
            dv = c.get("derived", {})
            b = dv.get("hbm_read_bytes_per_launch_corrected", 0.0) + dv.get("hbm_write_bytes_per_launch", 0.0)
            short = name.split("::")[-1].split("<")[0]
            per[short] = per.get(short, 0.0) + b
            tot += b
        step_traffic = {"bytes_per_step": tot, "algorithmic_bytes_per_step": 8.0 * B * T, "ratio": tot / (8.0 * B * T),
                        "by_kernel": per, "achieved_GBs": tot / (1e-3 * ms_per_step) / 1e9,
                        "frac_of_hbm_peak": tot / (1e-3 * ms_per_step) / 1e9 / HBM_PEAK_GBS,
                        "what": "FETCH_SIZE x 2 + WRITE_SIZE of the scan launches of one step (rocprofv3 --pmc, separate passes; "
                                "MI355X_MICROARCH.md corrections): implementation traffic = the per-step state stash, written by the "
                                "forward and read by the reverse sweep", "source": doc["_source"]}
    bytes_alg = 8.0 * B * T
    krecs = kernel_records(fam, D, T, B, rank1, ktimes) if ktimes else None
    if fam in ("wide", "pair"):
        # Several kernels on different pipes per entry point: SURVEY's algorithmic flops over ONE pipe's peak is not a fraction of
        # anything (round 3: 1.32).  The dominant kernel is the one with the longest launch; its record against ITS OWN pipe is the
        # object's achieved / peak / frac; the algorithmic figure stays as a rate without a denominator.
        comp = [r for r in (krecs or []) if r.get("pipe") in ("valu_fp32", "mfma_f32", "mfma_bf16")]
        top = max(comp, key=lambda r: r["duration_ms"]) if comp else None
        whole_alg = (alg["fwd"] + alg["bwd"]) / (1e-3 * ms_per_step) / 1e12
        return {
            "bound": "mfma",
            "binding": ("one LDS round trip (store latency + barrier + operand burst) + 512 matrix-pipe cycles + the dependent tail per step of "
                        "ONE wave per SIMD") if pair else
                       ("instruction issue of the float32 VALU chain kernels (two waves per SIMD); the GEMM kernels sit at what hides behind an MFMA"
                        if wide_chain == 0 else
                        "serial chain of ONE wave per SIMD: the matrix pipe is busy ~60 % of a step (f16x2-split products, half the A rows "
                        "padding), the rest is the LDS round trip + barrier + dependent tail; the GEMM kernels sit at their operand builds"),
            "kernel": top["kernel"] if top else (name_f if dom == "fwd" else name_b),
            "pipe": top["pipe"] if top else None,
            "achieved": top["achieved"] if top else None, "peak": top["peak"] if top else peak, "unit": "TFLOP/s",
            "frac": top["frac"] if top else None,
            "frac_is": "EXECUTED (as issued) flops of the longest kernel on its pipe / live launch time / that pipe's peak (roofline_detail.kernels: "
                       "every kernel and pipe, HBM included)",
            "launch_ms": top["duration_ms"] if top else tt[dom] * 1e3,
            "traffic": traffic, "traffic_source": doc["_source"] if doc is not None else None,
            "algorithmic_bytes": 4.0 * B * T,
            "whole_step_frac_algorithmic": None,
            "kernels": krecs,
            "algorithmic": {"flops_per_step": alg["fwd"] + alg["bwd"], "whole_step_tflops": whole_alg,
                            "fwd_entry_tflops": alg["fwd"] / t_fwd / 1e12, "bwd_entry_tflops": alg["bwd"] / t_bwd / 1e12,
                            "note": "SURVEY 8(d): 80 D^2 flop per (clip, sample) over the entry points' times -- a rate, deliberately not divided "
                                    "by a peak: the work is spread over the fp32 VALU and the bf16 matrix pipe"},
            "entry_ms": {"cmps_psi_loss_fwd": t_fwd * 1e3, "cmps_psi_loss_bwd": t_bwd * 1e3},
            "step_traffic": step_traffic,
            "hbm": {"achieved_algorithmic": bytes_alg / (t_fwd + t_bwd) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac_algorithmic": bytes_alg / (t_fwd + t_bwd) / 1e9 / HBM_PEAK_GBS},
        }
    ach = alg[dom] / tt[dom] / 1e12
    whole = (alg["fwd"] + alg["bwd"]) / (1e-3 * ms_per_step) / 1e12
    return {
        # compute bound, but not by the matrix cores: what binds is the instruction issue of one wave per SIMD (`binding`), so the
        # contract's "mfma" class would mislabel it; the peak is the fp32 vector rate (= the f32-input MFMA rate)
        "bound": "valu-issue",
        "binding": ("instruction issue of ONE wave per SIMD: fp32 VALU mat-vecs on the serial chain, rank-1 / loss products beside them on "
                    "the matrix cores; 10 D^2 algorithmic flop per algorithmic byte, so never HBM"),
        "kernel": name_f if dom == "fwd" else name_b, "pipe": "valu_fp32 (+ matrix cores for the off-chain products)",
        # with 24 of SURVEY's 56 D^2 on the 2.5 PF matrix pipe and 20 removed by the stash, algorithmic flops over the fp32
        # vector peak are not bounded by 1 (1.07 with the two-wave reverse scan) -- the fraction is the EXECUTED fp32 VALU work of the
        # binding pipe; the algorithmic rate stays beside it
        "achieved": ach * split[dom]["valu_fp32"] / (24.0 if dom == "fwd" else 56.0), "peak": peak, "unit": "TFLOP/s",
        "frac": ach * split[dom]["valu_fp32"] / (24.0 if dom == "fwd" else 56.0) / peak,
        "frac_is": "EXECUTED fp32 VALU flop (12 D^2 per clip-sample) / live launch time / fp32 vector peak; *_algorithmic: SURVEY 8d's 56 D^2 bwd "
                   "(24 D^2 fwd) / the same time -- counts work the stash removes or the matrix cores run, so it may pass the vector peak",
        "achieved_algorithmic": ach, "frac_algorithmic": ach / peak,
        "whole_step_frac_algorithmic": whole / peak,
        "traffic": traffic, "traffic_source": doc["_source"] if doc is not None else None,
        "algorithmic_bytes": 4.0 * B * T, "launch_ms": tt[dom] * 1e3, "flops_per_launch": alg[dom],
        "executed": pipes(dom),
        "kernels": krecs,
        "other_kernel": {"kernel": name_b if dom == "fwd" else name_f, "achieved_algorithmic": alg[oth] / tt[oth] / 1e12,
                         "frac_algorithmic": alg[oth] / tt[oth] / 1e12 / peak,
                         "frac": alg[oth] / tt[oth] / 1e12 / peak * split[oth]["valu_fp32"] / (24.0 if oth == "fwd" else 56.0),
                         "launch_ms": tt[oth] * 1e3, "executed": pipes(oth)},
        "whole_step": {"flops": alg["fwd"] + alg["bwd"], "achieved_algorithmic": whole, "frac_algorithmic": whole / peak,
                       "note": "80 D^2 algorithmic flop per (clip, sample) over the full optimiser step"},
        "step_traffic": step_traffic,
        "hbm": {"achieved_algorithmic": bytes_alg / (t_fwd + t_bwd) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac_algorithmic": bytes_alg / (t_fwd + t_bwd) / 1e9 / HBM_PEAK_GBS,
                "note": "the audio itself, 8 B per (clip, sample): the scan is compute bound (BASELINE's 40 % of the HBM roofline is "
                        "unreachable as stated: 100 % of fp32 peak would be 0.2 % of it)"},
    }


BASELINE_CONFIGS = {(4, 256, 8): "BASELINE configs[0] shape", (16, 4096, 256): "BASELINE configs[1]",
                    (32, 16000, 1024): "BASELINE configs[2]", (128, 16000, 512): "BASELINE configs[4]"}


# ---------------------------------------------------------------------------------------------------
# one configuration on this rank's GPU
# ---------------------------------------------------------------------------------------------------
class Run:
    def __init__(self, D, T, B, variant, rank1, kind, dp, dev, world, rank, host_optimizer, config_id=3):
        import torch
        from audio_mps_amd import HParams, PsiCMPS
        from audio_mps_amd.scan import HipScan
        from audio_mps_amd.train import Trainer
        self.D, self.T, self.B, self.kind, self.dp, self.dev, self.world = D, T, B, kind, dp, dev, world
        self.hp = HParams(minibatch_size=B * world, bond_dim=D)      # train.py:41-43 defaults otherwise
        self.seed = 1000 * config_id + rank
        # resident in HBM before the timed region
        self.audio = torch.from_numpy(make_audio_host(kind, B, T, self.hp.delta_t, self.seed)).to(dev)
        self.backend = HipScan(D, device=dev, variant=variant, rank1=RANK1_MODES[rank1])
        kw = {}
        if D > 64:      # R_in = 0.5 / sqrt(D) (N(0,1) + i N(0,1)), zero diagonal, keeps 1 + e x / A > 0 (hard part viii)
            rng = np.random.default_rng(0)
            Rin = (0.5 / np.sqrt(D)) * (rng.standard_normal((D, D)) + 1j * rng.standard_normal((D, D)))
            np.fill_diagonal(Rin, 0.0)
            kw["R_in"] = Rin.astype(np.complex64)
        self.model = PsiCMPS(self.hp, seed=0, backend=self.backend, **kw)
        self.trainer = Trainer(self.model, self.hp, dp, device_step=not host_optimizer)
        self.host_optimizer = host_optimizer
        self.variant = int(self.backend.variant)
        self.wide_chain = int(self.backend.wide_chain)               # CMPS_OPT_WIDE_CHAIN of this handle (matters for V_WIDE only)

    def step(self):
        if self.host_optimizer:
            self.trainer.step(self.audio)
        else:
            self.trainer.step(self.audio, sync=False, global_batch=self.B * self.world)

    def last_loss(self):
        if self.host_optimizer:
            return float(self.trainer.history[-1]["model_loss"])
        return float(self.trainer._dev["losses"].cpu().numpy()[0])

    def timed(self, steps, warmup):
        """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        import torch
        dp = self.dp
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        dp.collective_us()                                           # drain the warm-up's collective events
        dp.collective_ms.clear()
        self.backend.timing = []
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        local = time.perf_counter() - t0
        dp.barrier()
        elapsed = dp.max_over_ranks(time.perf_counter() - t0)
        tm = self.backend.timing_ms()
        self.backend.timing = None
        return {"elapsed": elapsed, "local": local, "t_fwd": float(np.mean(tm["fwd"])) * 1e-3,
                "t_bwd": float(np.mean(tm["bwd"])) * 1e-3, "last": self.last_loss()}

    def kernel_pass(self, steps=3):
        """Per-kernel live durations of the same step (HIP events around every launch, cmps_kernel_times): a separate short pass, so
        the events never sit inside the timed region."""
        import torch
        be = self.backend
        be.kernel_events(True)
        try:
            self.step()                                              # first launch with events: event creation
            torch.cuda.synchronize()
            be.kernel_times()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            return be.kernel_times()
        finally:
            be.kernel_events(False)

    def parity(self, sample, ref, pair):
        """The GPU's per-clip loss and gradient sums on a host sample against the oracle's (chunks of at most B clips)."""
        import torch
        from audio_mps_amd.scan import unpack_grad
        from oracle import c_oracle as C
        be, D, B, T = self.backend, self.D, self.B, self.T
        self.trainer.sync_to_host()
        gr = C.unpack_grad(ref["grad"], D)
        d_sample = torch.from_numpy(sample).to(self.dev)
        Bs = sample.shape[0]
        be.set_params(self.model.effective_params(), B, T, train=True)
        pers, flat = [], None
        for s0 in range(0, Bs, B):
            chunk = d_sample[s0:s0 + B].contiguous()
            pers.append(be.forward(chunk, save_for_bwd=True).cpu().numpy().copy())
            gch = be.backward().cpu().numpy().astype(np.float64)
            flat = gch if flat is None else flat + gch
        per = np.concatenate(pers)
        g = unpack_grad(flat.astype(np.float32), D)
        loss_err = float(np.max(np.abs(per - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1.0)))
        loss_err_plain = float(np.max(np.abs(per - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1e-30)))
        gerr = {k: rel_inf(g[k], gr[k]) for k in ("Rbar", "fbar", "psi0bar", "Abar")}
        tol_l, tol_g = (2e-3, 3e-2) if pair else (1e-5, 1e-4)
        return {"clips": Bs, "max_rel_loss_err": loss_err, "max_rel_loss_err_unfloored": loss_err_plain,
                "loss_err_note": "max_rel_loss_err = |d| / max(|loss_b|, 1) (the tolerance's form); _unfloored = |d| / |loss_b|, with min |loss_b| = "
                                 f"{float(np.min(np.abs(ref['loss_per_clip']))):.3g} on this sample",
                "max_rel_grad_err": max(gerr.values()), "grad_err_by_tensor": gerr,
                "tolerance": {"loss": tol_l, "grad": tol_g}, "ok": bool(loss_err <= tol_l and max(gerr.values()) <= tol_g),
                "against": "oracle/cmps_oracle.c float32 (parity UNPINNED: no reference-held vectors exist)"}


def dtype_label(variant, D, rank1, wide_chain=1):
    """The arithmetic the path computes in (a label, not a precision claim); `arithmetic_note` has the long form."""
    if variant == V_PAIR:
        return "bf16 operands, f32 state and accumulate"
    if variant == V_WIDE and wide_chain != 0:
        return "f32 (mat-vecs as f16x2-split operands on the matrix cores, f32 accumulate)"
    return "f32"


def arithmetic_note(variant, D, rank1, wide_chain=1):
    if variant == V_PAIR:
        return "bf16 mat-vec operands on v_mfma_f32_16x16x32_bf16; fp32 state, identity part and accumulate"
    split = {0: "exact fp32 MFMA", 1: "bf16x2 split (16 operand bits)", 2: "bf16x3 split (24 operand bits)",
             3: "f16x2 split (scaled, 24 operand bits)"}[rank1]
    if variant in (V_WAVE, V_WAVE32) and not (variant == V_WAVE and D <= 16):
        hy = "f16x2 split" if rank1 == 3 else "bf16x3 split"
        return "fp32 FMA chains on the serial path; off-chain products on the matrix cores, fp32 accumulate: H y " + hy + ", rank-1 sums " + split
    if variant == V_WAVE:
        return "fp32 FMA chains; rank-1 gradient sums as exact fp32 MFMAs (16-row layout)"
    if variant == V_WIDE:
        chain = {0: "serial chains: fp32 v_pk_fma (VALU)", 1: "serial chains (forward and reverse): scaled f16x2-split mat-vec operands on the "
                 "matrix cores, fp32 identity part and accumulate", 2: "forward chain: scaled f16x2-split operands on the matrix cores; reverse chain: fp32 VALU"}[wide_chain]
        return chain + "; H y and rank-1 gradient GEMMs: " + split + ", fp32 accumulate"
    return "plain fp32 FMA code"


def other_config_rows(ARGS, dp, dev):
    """BASELINE configs[0] (shape), [1], [4] in float32 and [4] as named (bf16): a few steps each + parity on <= 16 CPU clips,
    outside the headline's timed region -- driver-witnessed numbers for every single-GPU config."""
    import torch
    rows = []
    cores = host_cores()
    for key, name, D, T, B, variant, steps, cid in (
            ("c0", "configs[0] shape: D=4, T=256, batch 8", 4, 256, 8, 0, 20, 1),
            ("c1", "configs[1]: D=16, T=4096, batch 256", 16, 4096, 256, 0, 10, 2),
            ("c4_f32", "configs[4] in float32: D=128, T=16000, batch 512 (wide kernels)", 128, 16000, 512, 0, 3, 5),
            ("c4_bf16", "configs[4] as named (bf16 operands, fp32 accumulate): pair kernels", 128, 16000, 512, V_PAIR, 3, 5)):
        run = None
        try:
            run = Run(D, T, B, variant, ARGS.rank1, ARGS.input, dp, dev, 1, 0, ARGS.host_optimizer, config_id=cid)
            r = run.timed(steps, 2)
            ms = 1e3 * r["elapsed"] / steps
            roof = roofline_record(D, T, B, run.variant, run.backend.effective_rank1, r["t_fwd"], r["t_bwd"], ms, run.kernel_pass(2),
                                   run.wide_chain)
            clips = min(B, 16)
            sample = make_audio_host(ARGS.input, clips, T, run.hp.delta_t, run.seed)
            run.trainer.sync_to_host()
            t0 = time.perf_counter()
            ref = oracle_on(run.model, sample, cores)
            cpu_s = time.perf_counter() - t0
            par = run.parity(sample, ref, run.variant == V_PAIR)
            rows.append({"key": key, "config": name, "ms_per_step": ms, "value": B * T * steps / r["elapsed"], "unit": "samples/s", "steps": steps,
                         "kernel_variant": run.variant, "dtype": dtype_label(run.variant, D, run.backend.effective_rank1, run.wide_chain),
                         "arithmetic": arithmetic_note(run.variant, D, run.backend.effective_rank1, run.wide_chain),
                         "wide_chain": run.wide_chain if run.variant == V_WIDE else None,
                         "fwd_ms": r["t_fwd"] * 1e3, "bwd_ms": r["t_bwd"] * 1e3, "final_loss": r["last"],
                         "dominant_kernel": roof["kernel"], "dominant_pipe": roof.get("pipe", "valu_fp32 (executed flops, see roofline.frac_is)"),
                         "frac": roof["frac"], "peak": roof["peak"], "kernels": roof.get("kernels"),
                         "algorithmic_whole_step_tflops": (roof["algorithmic"]["whole_step_tflops"] if "algorithmic" in roof
                                                           else roof["whole_step"]["achieved_algorithmic"]),
                         "cpu_port": {"value": clips * T / cpu_s, "unit": "samples/s", "cores": cores, "sample": f"{clips} clips, {cpu_s:.2f} s"},
                         "parity_in_bench": par})
        except Exception as exc:                                      # a failing side configuration must not take the headline down
            rows.append({"key": key, "config": name, "error": f"{type(exc).__name__}: {exc}"})
        emit_detail("other_config:" + key, rows[-1])                 # one line per configuration, as soon as it is known
        del run
        torch.cuda.empty_cache()
    return rows


# ---------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------
def worker(ARGS):
    import torch
    from audio_mps_amd.parallel import DataParallel, collective_settings

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
    run = Run(D, T, B, ARGS.variant, ARGS.rank1, ARGS.input, dp, dev, world, rank, ARGS.host_optimizer)
    variant, rank1, wide_chain = run.variant, run.backend.effective_rank1, run.wide_chain
    r = run.timed(ARGS.steps, ARGS.warmup)
    if not np.isfinite(r["last"]):
        raise SystemExit(f"non-finite loss {r['last']}")
    allreduce_us = dp.collective_us()                            # mean HIP-event time of the collective (None at N = 1)
    ktimes = run.kernel_pass(3) if (rank == 0 and world == 1) else None      # per-kernel durations, outside the timed region
    per_rank_ms = dp.gather_floats(1e3 * r["local"] / ARGS.steps)
    # the data-parallel invariant: after the same all-reduced sums every rank holds bit-identical variables and Adam slots
    replicas_ok = None
    if world > 1 and not ARGS.host_optimizer:
        st = run.trainer._dev
        replicas_ok = all(dp.replicas_identical(st[k]) for k in ("vars", "m", "v"))
    rccl_world = dp.measured_world_size()                        # dist.get_world_size() after a GPU all-reduce of ones

    out = None
    details = []
    pair = variant == V_PAIR
    fam = family_of(variant, D)
    if rank == 0:
        ms_per_step = 1e3 * r["elapsed"] / ARGS.steps
        value = world * B * T * ARGS.steps / r["elapsed"]
        cfg_name = BASELINE_CONFIGS.get((D, T, B), "custom shape")
        if world > 1 and (D, T, B) == (32, 16000, 1024):
            cfg_name = "BASELINE configs[3] (configs[2] per GPU)" if world == 8 else f"BASELINE configs[2] per GPU x {world}"
        roof = roofline_record(D, T, B, variant, rank1, r["t_fwd"], r["t_bwd"], ms_per_step, ktimes, wide_chain)
        details.append(emit_detail("roofline_detail", roof))
        settings = collective_settings()
        out = {
            "metric": f"audio samples/sec (fwd+bwd) at D={D}, T={T}",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": ARGS.steps, "warmup": ARGS.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_label(variant, D, rank1, wide_chain), "data": "synthetic",
            "config": {"workload": f"{cfg_name}: PsiCMPS fwd+bwd scan, D={D}, T={T}, batch {B} per GPU"
                                   f" (global {B * world}), input {ARGS.input}, full optimiser step "
                                   f"({'host numpy' if ARGS.host_optimizer else 'device-resident'} chain rule + Adam)",
                       "parallelism": f"dp{world}", "kernel_variant": variant, "kernel_family": fam, "input": ARGS.input,
                       "optimizer_step": "host" if ARGS.host_optimizer else "device (cmps_psi_apply_step)",
                       "arithmetic": arithmetic_note(variant, D, rank1, wide_chain),
                       "rank1_updates": (("exact_f32 (16-row layout)" if fam == "wave16" else {v: k for k, v in RANK1_MODES.items()}[rank1])
                                         if fam in ("wave", "wave16", "wide") else None),
                       "wide_chain": ({0: "valu", 1: "mfma", 2: "mfma_fwd"}[wide_chain] if fam == "wide" else None)},
            "roofline": compact_roofline(roof),
            "fwd_ms": r["t_fwd"] * 1e3, "bwd_ms": r["t_bwd"] * 1e3,
            "final_loss": float(r["last"]),
            "per_rank_ms_per_step": {"min": float(np.min(per_rank_ms)), "max": float(np.max(per_rank_ms))},
            "rccl_world_size": rccl_world,
            "allreduce_us": allreduce_us,
            "collective_backend": dp.backend,                    # "nccl" (= RCCL), "gloo" (rehearsal) or None (N = 1 in process)
            "collective": {"message_bytes": 4 * (2 * D * D + 3 * D + 2), "calls_per_step": 1, "op": "all_reduce(sum), in place on the device buffer",
                           "settings": {k: v for k, v in sorted(settings.items()) if v is not None},
                           "replicas_bit_identical_after_run": replicas_ok},
        }
        if ARGS.rehearse_on_one_gpu:
            out["rehearsal"] = "all ranks share cuda:0 and reduce over gloo on the host: NOT a scaling measurement"

    # ---- outside the timed region: CPU baseline, parity of this very workload, precision A/B, the other configs (rank 0, N = 1) ----
    if rank == 0 and world == 1 and not ARGS.no_cpu_baseline:
        run.trainer.sync_to_host()                               # the oracle sees the variables the timed steps left behind
        cb, sample, ref = cpu_baseline(run.model, D, T, ARGS.input, seed=run.seed, clips_arg=ARGS.cpu_clips)
        out["cpu_baseline"] = cb
        par = run.parity(sample, ref, pair)
        details.append(emit_detail("parity_in_bench", par))
        out["parity_in_bench"] = compact_parity(par)
        if fam in ("wave", "wide") and not ARGS.no_precision_ab:
            ab = {}
            modes = {"exact_f32": 0, "bf16x2": 1, "bf16x3": 2, "f16x2": 3} if fam == "wave" else {"bf16x2": 1, "bf16x3": 2, "f16x2": 3}
            for name, mode in modes.items():                     # accuracy first: the timed steps below move the parameters
                run.backend.set_rank1(mode)
                pr = run.parity(sample, ref, False)
                ab[name] = {"max_rel_grad_err_vs_f32_oracle": pr["max_rel_grad_err"], "products": RANK1_LABEL[mode]}
            for name, mode in modes.items():
                run.backend.set_rank1(mode)
                rr = run.timed(5, 1)
                ab[name].update({"ms_per_step": 1e3 * rr["elapsed"] / 5, "samples_per_s": B * T * 5 / rr["elapsed"],
                                 "bwd_ms": rr["t_bwd"] * 1e3, "fwd_ms": rr["t_fwd"] * 1e3})
            run.backend.set_rank1(RANK1_MODES[ARGS.rank1])
            head_mode = {v: k for k, v in RANK1_MODES.items()}[rank1]
            details.append(emit_detail("precision_ab", {
                "what": "rank-1 gradient sums (k_bwd_wave / k_grad_gemm); everything else is identical fp32 code.  All "
                        "modes sit in float32 reorder noise of the oracle: the label 'fp32-faithful' of bf16x3 rests on its "
                        "operand-bit argument (24 bits kept), not on a difference this comparison can resolve",
                "headline_mode": head_mode, "modes": ab}))
            out["precision_ab"] = {"headline_mode": head_mode,
                                   "grad_err_vs_f32_oracle": {k: float(f"{v['max_rel_grad_err_vs_f32_oracle']:.3g}") for k, v in ab.items()},
                                   "ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in ab.items()}}
        if (D, T, B) == (32, 16000, 1024):
            details.append(emit_detail("reference_style", cpu_reference_style()))
            if not ARGS.no_other_configs:
                del run
                torch.cuda.empty_cache()
                rows = other_config_rows(ARGS, dp, dev)
                details += ["other_config:" + row["key"] for row in rows]
                out["other_configs"] = [compact_other_config(row) for row in rows]
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        out["detail_lines"] = details                            # names of the {"detail": ...} lines printed above this one
        print(headline_line(out), flush=True)                    # the LAST stdout line: the contract's record
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
