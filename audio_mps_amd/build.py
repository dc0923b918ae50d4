"""Builds libcmps.so (HIP, gfx950 only) in-tree with hipcc.  No CPU fallback exists: if this library is
missing the product raises at import of ``audio_mps_amd._capi``."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcmps.so")
SOURCES = ["cmps_capi.hip", "cmps_prep.hip", "cmps_opt.hip", "cmps_block.hip", "cmps_wave.hip", "cmps_wave2.hip", "cmps_wave16.hip", "cmps_pair.hip", "cmps_wide.hip", "cmps_legacy.hip", "cmps_rho.hip", "cmps_rho_wave.hip", "cmps_rho_mfma.hip"]
HEADERS = ["cmps_internal.h", "cmps_wave_util.h", "cmps_grad_gemm.h", os.path.join("..", "..", "include", "cmps.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libcmps.so cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xarch_host", "-msse4.2",
           "-Wall", "-Wno-unused-function",
           "-o", LIB_PATH] + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout)
    if verbose and proc.stdout:
        print(proc.stdout, file=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
