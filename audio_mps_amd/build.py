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
OBJ_DIR = os.path.join(_HERE, "obj")
SOURCES = ["cmps_capi.hip", "cmps_prep.hip", "cmps_opt.hip", "cmps_block.hip", "cmps_wave.hip", "cmps_wave2.hip", "cmps_wave16.hip", "cmps_pair.hip", "cmps_wide.hip", "cmps_legacy.hip", "cmps_rho.hip", "cmps_rho_wave.hip", "cmps_rho_mfma.hip"]
# per-source flags: the pair kernels place plain VALU between MFMAs themselves (cmps_pair.hip::mfma_valu_pipeline); the compiler's
# SLP packing into v_pk_* (expensive beside MFMAs, and a v_mov shuffle per operand) is switched off there
EXTRA_FLAGS = {"cmps_pair.hip": ["-fno-slp-vectorize"]}
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


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xarch_host", "-msse4.2", "-Wall", "-Wno-unused-function"]


def build(force: bool = False, verbose: bool = False, extra_flags=(), jobs: int | None = None) -> str:
    """One object per source, compiled in parallel (a source is recompiled when it or any header is newer than its
    object, or when the flags changed), then one link.  ``force`` recompiles everything."""
    if not force and not needs_build() and not extra_flags:
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    flags = _flags() + list(extra_flags)
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    todo, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not same_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        return src, subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 2)) as pool:
        for src, proc in pool.map(compile_one, todo):
            if proc.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n" + proc.stdout)
            if verbose and proc.stdout:
                print(proc.stdout, file=sys.stderr)
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    proc = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + proc.stdout)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
