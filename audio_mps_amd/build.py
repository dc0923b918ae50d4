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
SOURCES = ["cmps_capi.hip", "cmps_prep.hip", "cmps_opt.hip", "cmps_block.hip", "cmps_wave.hip", "cmps_wave_bwd2.hip", "cmps_wave2.hip", "cmps_wave16.hip", "cmps_pair.hip", "cmps_wide.hip", "cmps_legacy.hip", "cmps_rho.hip", "cmps_rho_wave.hip", "cmps_rho_mfma.hip"]
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


def _stamp_text(flags) -> str:
    """Everything a compile depends on besides the sources: the common flags AND the per-source ones."""
    return " ".join(flags) + " | " + " ; ".join(f"{k}: {' '.join(v)}" for k, v in sorted(EXTRA_FLAGS.items()))


def build(force: bool = False, verbose: bool = False, extra_flags=(), jobs: int | None = None) -> str:
    """One object per source, compiled in parallel (a source is recompiled when it or any header is newer than its
    object, or when the flags changed), then one link.  ``force`` recompiles everything.

    ``extra_flags`` (diagnostic -DCMPS_DIAG ... builds, whose results are documented as wrong) never touch the shipped
    library: they compile into their own object directory and link ``lib/libcmps_diag_<hash>.so``, whose path is returned
    (select it with CMPS_LIB).  The shipped library is rebuilt whenever its stamped flags differ from the current ones."""
    import hashlib
    flags = _flags() + list(extra_flags)
    obj_dir, lib_path = OBJ_DIR, LIB_PATH
    if extra_flags:
        tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:10]
        obj_dir, lib_path = OBJ_DIR + "_diag_" + tag, os.path.join(LIB_DIR, f"libcmps_diag_{tag}.so")
    stamp = os.path.join(obj_dir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == _stamp_text(flags)
    if not force and not extra_flags and same_flags and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    todo, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not same_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        tmp = f"{obj}.{os.getpid()}.tmp"                      # concurrent ranks never write the same file; the rename is atomic
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode == 0:
            os.replace(tmp, obj)
        return src, proc

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 2)) as pool:
        for src, proc in pool.map(compile_one, todo):
            if proc.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n" + proc.stdout)
            if verbose and proc.stdout:
                print(proc.stdout, file=sys.stderr)
    with open(stamp, "w") as f:
        f.write(_stamp_text(flags))
    tmp_lib = f"{lib_path}.{os.getpid()}.tmp"
    proc = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + objs,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + proc.stdout)
    os.replace(tmp_lib, lib_path)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
