#!/usr/bin/env python3
"""Per-kernel launch durations of one bench configuration (HIP events around every launch, bench.py's roofline_detail line), one row per
kernel: the A/B tool for a change to ONE kernel of a multi-kernel step.  usage: kernel_ms.py [bench.py flags ...]   (CMPS_LIB selects
a diagnostic library build)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-other-configs", "--no-precision-ab", "--steps", "5", "--warmup", "2"] + sys.argv[1:]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout
seen = {}
for line in out.splitlines():
    try:
        d = json.loads(line)
    except Exception:
        continue
    if d.get("detail") == "roofline_detail":
        for r in d["data"].get("kernels") or []:
            seen.setdefault(r["kernel"], r["duration_ms"])
    elif "metric" in d:
        print(f"{'step':40s} {d['ms_per_step']:9.3f} ms")
for k, ms in seen.items():
    print(f"{k:40s} {ms:9.3f} ms")
