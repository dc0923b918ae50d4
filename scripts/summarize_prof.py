#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile_r1.sh into the two files committed under profiles/:
   <tag>_kernel_stats.csv  (the --kernel-trace --stats table, verbatim)
   <tag>_pmc_summary.json  (per-kernel averages of every collected counter + derived per-wave-step figures).
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in 64-byte units... this image's
rocprofv3 reports them in KB (x 1024 B); gfx950 correction for reads: FETCH_SIZE x 2."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r1"
tag = sys.argv[2] if len(sys.argv) > 2 else "r1_v6"
note = sys.argv[3] if len(sys.argv) > 3 else ""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    per_dispatch = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0]
            if "cmps::k_fwd_wave" in name or "cmps::k_bwd_wave" in name:
                per_dispatch[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (name, _, cname), v in per_dispatch.items():
        acc[name][cname].append(v)
B, N = 1024, 15999
out = {"command": "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline (rocprofv3 --kernel-trace [--pmc ...], one pass per "
                  "counter group; scripts/profile_r1.sh, condensed by scripts/summarize_prof.py)",
       "workload": f"D=32, T=16000, B={B} ({B * N / 1e6:.2f}M clip-steps per launch)", "version": note, "kernels": {}}
for name, counters in acc.items():
    k = {c: sum(v) / len(v) for c, v in counters.items()}
    steps = B * N
    d = {}
    for c, key in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_LDS", "lds"), ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_MFMA", "mfma"),
                   ("SQ_INSTS_VMEM_WR", "vmem_wr"), ("SQ_INSTS_VMEM_RD", "vmem_rd")):
        if c in k:
            d[f"{key}_insts_per_clip_step"] = k[c] / steps
    if "SQ_WAVE_CYCLES" in k:
        d["wave_cycles_per_clip_step"] = k["SQ_WAVE_CYCLES"] * 4 / steps      # counter is in units of 4 cycles
    if "SQ_BUSY_CYCLES" in k and "SQ_WAIT_ANY" in k and "SQ_WAVE_CYCLES" in k:
        d["frac_wait_any"] = k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]
        d["frac_active_inst_any"] = k.get("SQ_ACTIVE_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
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
