#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python scripts/kernel_resources.py audio_mps_amd/csrc/cmps_wide.hip [extra hipcc flags]"""
import re
import subprocess
import sys
import os

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
       "-I", os.path.join(root, "audio_mps_amd", "csrc"), src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line) or re.search(r"Name: (\S+) \[", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\])?(?: \[waves/SIMD\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    print(f"{r['name'][:60]:60s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):4d} SGPR {r.get('TotalSGPRs', -1):4d} "
          f"spill {r.get('VGPRs Spill', -1):4d} scratch {r.get('ScratchSize', -1):5d} occ {r.get('Occupancy', -1)} LDS {r.get('LDS Size', -1)}")
