#!/usr/bin/env python3
"""Build a variant of libcmps.so for A/B timing: python scripts/build_variant.py NAME [-DFLAG ...] -> audio_mps_amd/lib/libcmps_NAME.so
(select it with CMPS_LIB=<path>; scripts/ab_libs.sh interleaves rounds of several libraries on one device)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_mps_amd import build as B
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(B.LIB_DIR, f"libcmps_{name}.so")
cmd = [B._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xarch_host", "-msse4.2", "-Wno-unused-function",
       "-o", out] + flags + [os.path.join(B.CSRC, s) for s in B.SOURCES]
subprocess.run(cmd, check=True)
print(out)
