#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile_scan.sh into the two files committed under profiles/:
   <tag>_kernel_stats.csv  (the --kernel-trace --stats table, verbatim)
   <tag>_pmc_summary.json  (per-kernel averages of every collected counter + derived per-clip-step figures).
usage: summarize_prof.py SRC_DIR TAG D T B
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: this image's rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
(x 1024 B); gfx950 correction for wide coalesced reads: FETCH_SIZE x 2.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles (x 4)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r2_c3"
tag = sys.argv[2] if len(sys.argv) > 2 else "r2_c3"
D, T, B = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (32, 16000, 1024)
N = T - 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
SCAN = ("k_fwd_wave", "k_bwd_wave", "k_fwd_pair", "k_bwd_pair", "k_grad_gemm", "k_fwd_block", "k_bwd_block", "k_fwd_wide", "k_bwd_wide",
        "k_fwd_chain16", "k_bwd_chain16", "k_sample_wide", "k_sample_wave",
        "k_hy_wide", "k_loss_wide", "k_apply_step", "k_reduce_slabs", "k_reduce_parts", "k_finalize", "k_pack", "k_rho_raw", "k_rho_fix")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    per_dispatch = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0]
            if any(k in name for k in SCAN):
                per_dispatch[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (name, _, cname), v in per_dispatch.items():
        acc[name][cname].append(v)
cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "python3 bench.py"
out = {"command": cmd + "  (rocprofv3 --kernel-trace [--pmc ...] -- <that>, one pass per counter group; scripts/profile_scan.sh, "
                        "condensed by scripts/summarize_prof.py)",
       "workload": f"D={D}, T={T}, B={B} ({B * N / 1e6:.2f}M clip-steps per launch)", "kernels": {}}
for name, counters in acc.items():
    k = {c: sum(v) / len(v) for c, v in counters.items()}
    steps = B * N
    d = {}
    for c, key in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_LDS", "lds"), ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_MFMA", "mfma"),
                   ("SQ_INSTS_VMEM_WR", "vmem_wr"), ("SQ_INSTS_VMEM_RD", "vmem_rd")):
        if c in k:
            d[f"{key}_insts_per_clip_step"] = k[c] / steps
    if "SQ_WAVE_CYCLES" in k:
        d["wave_cycles_per_clip_step"] = k["SQ_WAVE_CYCLES"] * 4 / steps      # all waves of a clip together
    for c, key in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                   ("SQ_ACTIVE_INST_VALU", "active_inst_valu")):
        if c in k:
            d[f"{key}_cycles_per_clip_step"] = k[c] * 4 / steps
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k:
        d["mfma_busy_cycles_per_clip_step"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / steps
    if "SQ_LDS_BANK_CONFLICT" in k and "SQ_LDS_IDX_ACTIVE" in k:
        d["lds_bank_conflict_frac"] = k["SQ_LDS_BANK_CONFLICT"] / max(k["SQ_LDS_IDX_ACTIVE"], 1.0)
    if "FETCH_SIZE" in k:
        d["hbm_read_bytes_per_launch_corrected"] = k["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in k:
        d["hbm_write_bytes_per_launch"] = k["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in k and "TCC_MISS_sum" in k:
        d["l2_hit_rate"] = k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k["TCC_MISS_sum"], 1.0)
    k["derived"] = d
    out["kernels"][name] = k
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({n: v["derived"] for n, v in out["kernels"].items()}, indent=1))
if stats:
    print(open(stats[0]).read()[:3000])
