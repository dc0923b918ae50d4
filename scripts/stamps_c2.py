#!/usr/bin/env python3
"""Where a step of the D <= 16 chain waves goes: runs BASELINE configs[1] (D=16, T=4096, B=256) once on the
stamped diagnostic build (-DCMPS_DIAG -DW16_TIMING: s_memtime at the phase boundaries of k_fwd_wave16 / k_bwd_wave16's chain
waves, block 0 prints the per-phase averages) and on the product build for the un-stamped launch times.
    python scripts/build_variant.py w16stamp -DCMPS_DIAG -DW16_TIMING && python scripts/stamps_c2.py > profiles/r3_c2_stamps.log"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = ["16", "4096", "256", "5"]
lib = os.path.join(ROOT, "audio_mps_amd", "lib", "libcmps_w16stamp.so")
print("# BASELINE configs[1]: D=16, T=4096, B=256 (one clip per CU; chain wave + helper wave on two SIMDs of the CU)")
print("## product build (scripts/time_kernels.py, HIP events)")
sys.stdout.flush()
out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_kernels.py")] + args, capture_output=True, text=True)
print(out.stdout.strip())
print("## stamped build (s_memtime; the stamps themselves cost ~10 % of the step, so the phases sum to more than the product's step)")
sys.stdout.flush()
out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_kernels.py")] + args[:3] + ["1"], capture_output=True, text=True,
                     env=dict(os.environ, CMPS_LIB=lib))
lines = [l for l in out.stdout.splitlines() if "cycles per step" in l]
seen = set()
for l in lines:
    key = l.split(",")[0]
    if key not in seen or True:
        print(l)
    seen.add(key)
print(out.stdout.strip().splitlines()[-4] if out.stdout.strip() else out.stderr[-2000:])
