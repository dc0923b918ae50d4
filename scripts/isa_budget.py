#!/usr/bin/env python3
"""Instruction budget of a kernel's inner loop from the ISA hipcc emits ("show the instruction budget").

usage: isa_budget.py SOURCE.hip 'MANGLED_KERNEL_PREFIX' [--steps 8] [--loop n | --ranges LO:HI,LO:HI,...] [--branches] [--flags ...]
  --branches   print the kernel's branch map (line numbers relative to the kernel label) and exit: how to find the hot path of a loop
               that holds a cold block (the F16X2 reverse scan's rescale block has its own small loops, so the step loop is not "innermost")
  --ranges     count these line ranges (inclusive, relative to the kernel label) instead of an automatically chosen loop
Compiles SOURCE for gfx950 (device only, -S), finds the kernel, takes its LONGEST innermost loop (the unrolled steps) and prints the
instructions per step by mnemonic and by class.  The classes are mnemonic heuristics for the wave-per-clip kernels:
  mat-vec FMAs   v_pk_fma_f32 with op_sel (the CM chains of cmps_wave_util.h)
  forming M_k    v_pk_fma_f32 / v_fma_f32 without op_sel on the chain registers (Q + s R)
  splits         v_and_b32, v_sub_f32, v_perm_b32, v_cvt_pk_bf16_f32, v_pk_add_f32 (bf16 piece arithmetic of the rank-1 operands)
  exchange       v_permlane*, DPP forms, v_readlane, v_cndmask (lane exchanges, reductions, selects)
  moves          v_mov_b32, v_accvgpr_*
  chain scalar   the remaining VALU (rotation, ybar assembly, df / dA sums)
"""
import collections, os, re, subprocess, sys, tempfile

def main():
    src, sym = sys.argv[1], sys.argv[2]
    steps = 8
    flags = []
    which = 0
    ranges, branches = None, False
    a = sys.argv[3:]
    while a:
        if a[0] == "--steps": steps = int(a[1]); a = a[2:]
        elif a[0] == "--loop": which = int(a[1]); a = a[2:]      # 0: the longest innermost loop, 1: the second longest, ...
        elif a[0] == "--ranges": ranges = [tuple(int(x) for x in r.split(":")) for r in a[1].split(",")]; a = a[2:]
        elif a[0] == "--branches": branches = True; a = a[1:]
        elif a[0] == "--flags": flags = a[1:]; a = []
        else: a = a[1:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src] + flags,
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # innermost loops: a label with "Depth=N" comments up to the backward branch to it; pick the longest
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    if branches:
        for i, l in enumerate(body):
            m = re.match(r"\s+s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels:
                t = labels[m.group(1)]
                print(f"{i:6d} {l.strip():40s} -> {t:6d} {'BACK' if t < i else 'fwd'}")
        return
    inner = [(lo, hi) for lo, hi in loops if not any(lo < l2 and h2 < hi for l2, h2 in loops)]     # no other loop inside
    inner.sort(key=lambda t: t[0] - t[1])
    print("innermost loops (lines):", [h - l + 1 for l, h in inner][:8])
    lo, hi = inner[which]
    picked = body[lo:hi + 1]
    if ranges:
        picked = [l for a0, b0 in ranges for l in body[a0:b0 + 1]]
        lo, hi = ranges[0][0], ranges[0][0] + len(picked) - 1
        print("counting the given line ranges:", ranges)
    mn = collections.Counter()
    cls = collections.Counter()
    for l in picked:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        op = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", l.split()[0])
        mn[op] += 1
        if op.startswith("v_mfma"): c = "MFMA"
        elif op.startswith("ds_"): c = "LDS"
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c = "VMEM"
        elif op == "s_nop": c = "s_nop"
        elif op == "s_waitcnt": c = "s_waitcnt"
        elif op.startswith("s_"): c = "SALU / branch"
        elif op == "v_pk_fma_f32" and "op_sel:" in l: c = "VALU mat-vec FMAs (v_pk_fma_f32, op_sel chains)"
        elif op in ("v_pk_fma_f32", "v_fma_f32") : c = "VALU forming M_k = Q + s R(+) and other plain FMAs"
        elif op in ("v_and_b32", "v_sub_f32", "v_perm_b32", "v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_lshlrev_b32", "v_cvt_pk_f16_f32",
                    "v_fma_mix_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_pk_mul_f32", "v_cvt_f32_f16", "v_pack_b32_f16"): c = "VALU operand splits + packing (bf16 / fp16 pieces)"
        elif op.startswith("v_permlane") or "row_" in l or "quad_perm" in l or op in ("v_readlane_b32", "v_cndmask_b32", "v_readfirstlane_b32"): c = "VALU lane exchange / reductions / selects"
        elif op in ("v_mov_b32", "v_mov_b64") or op.startswith("v_accvgpr"): c = "VALU moves"
        else: c = "VALU chain scalar math (rotation, ybar, df / dA sums)"
        cls[c] += 1
    tot = sum(mn.values())
    print(f"{os.path.basename(src)} :: {sym}  inner loop of {hi - lo + 1} lines = {tot} instructions = {tot / steps:.1f} per step ({steps} steps per iteration)")
    print("by class (per step):")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
        print(f"  {k:70s} {v / steps:7.2f}")
    valu = sum(v for k, v in cls.items() if k.startswith("VALU"))
    print(f"  {'all VALU (without MFMA)':70s} {valu / steps:7.2f}")
    print("by mnemonic (per step):")
    for k, v in mn.most_common(30):
        print(f"  {k:34s} {v / steps:7.2f}")

if __name__ == "__main__":
    main()
