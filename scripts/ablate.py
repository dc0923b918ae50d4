#!/usr/bin/env python3
"""Diagnostic: build libcmps variants with ablation flags (results wrong, timing informative) and time the scans."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_mps_amd import build
variants = {"base": [], "no_stagger": ["-DFWD2_NO_STAGGER"], "loss_prio1": ["-DFWD2_LOSS_PRIO=1"], "loss_prio3": ["-DFWD2_LOSS_PRIO=3"], "loss_stagger": ["-DFWD2_LOSS_STAGGER"]}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name, flags in variants.items():
    lib = os.path.join(ROOT, "gpurun_out", f"libcmps_{name}.so")
    subprocess.run([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", lib] + flags
                   + [os.path.join(build.CSRC, s) for s in build.SOURCES], check=True)
    env = dict(os.environ, CMPS_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_kernels.py"), "32", "16000", "1024", "4"], env=env, capture_output=True, text=True).stdout
    print("==", name); print("\n".join(l for l in out.splitlines() if "median" in l))
