#!/usr/bin/env python3
"""Diagnostic: build libcmps variants with extra -D flags (results may be wrong, timing is informative) and time the scans.

usage: ablate.py [--shape D T B ROUNDS VARIANT] name=-DFLAG[,-DFLAG...] ...
  e.g. ablate.py base= no_mfma=-DPABL_NO_MFMA --shape 128 16000 512 2 3
Every build gets -DCMPS_DIAG; known switches (all inert without it): CMPS_DIAG_NO_LOSS / CMPS_DIAG_NO_CHAIN (cmps_wave2.hip:
only the chain wave / only the loss wave of the forward runs), PABL_NO_MFMA / PABL_NO_BARRIER / PABL_TIMING (s_memtime stamps
inside the pair scans, printed per launch) (cmps_pair.hip; round 4 dropped NO_REDUCE, NO_EXPORT*, NO_STASHREAD and HALF_READS with
the code they switched).  tests/test_capi_load.py compiles each of them so that they cannot rot.
Runs on the GPU box (hipcc is available there); libraries go to gpurun_out/.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_mps_amd import build

args = sys.argv[1:]
if "--only" in args:
    i = args.index("--only")
    del args[i:i + 2]
shape = ["32", "16000", "1024", "4"]
if "--shape" in args:
    i = args.index("--shape")
    shape = args[i + 1:]                  # D T B ROUNDS VARIANT [RANK1 [WIDE_CHAIN]]: everything scripts/time_kernels.py takes
    args = args[:i]
variants = {}
for a in args or ["base="]:
    name, _, flags = a.partition("=")
    variants[name] = [f for f in flags.split(",") if f]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
only = None
if "--only" in sys.argv:          # --only cmps_pair.hip[,cmps_wide.hip]: recompile just these sources, link the other objects of the normal build
    only = sys.argv[sys.argv.index("--only") + 1].split(",")
for name, flags in variants.items():
    if name in ("--only",) or (only and name in only):
        continue
    lib = os.path.join(ROOT, "gpurun_out", f"libcmps_{name}.so")
    common = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xarch_host", "-msse4.2", "-DCMPS_DIAG"] + flags
    if only:
        build.build()
        objs = []
        for s in build.SOURCES:
            o = os.path.join(build.OBJ_DIR, s.replace(".hip", ".o"))
            if s in only:
                o = os.path.join(ROOT, "gpurun_out", f"{name}_{s}.o")
                subprocess.run(common + build.EXTRA_FLAGS.get(s, []) + ["-c", os.path.join(build.CSRC, s), "-o", o], check=True)
            objs.append(o)
        subprocess.run([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    else:
        subprocess.run(common + ["-shared", "-o", lib] + [os.path.join(build.CSRC, s) for s in build.SOURCES], check=True)
    env = dict(os.environ, CMPS_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_kernels.py")] + shape, env=env,
                         capture_output=True, text=True).stdout
    print("==", name, " ".join(flags))
    print("\n".join(l for l in out.splitlines() if "median" in l or "cycles per step" in l))
