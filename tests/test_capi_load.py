"""The C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every symbol that
include/cmps.h declares.  No compute call is made here (that needs a GPU: tests marked gpu)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cmps.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmps_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(hip_lib):
    from audio_mps_amd import _capi
    names = declared_symbols()
    assert len(names) >= 12
    assert sorted(_capi.SYMBOLS) == names
    for n in names:
        assert hasattr(hip_lib, n), n


def test_host_only_entry_points(hip_lib):
    from audio_mps_amd import _capi
    assert hip_lib.cmps_version() >= 100
    h = ctypes.c_void_p()
    assert hip_lib.cmps_create(0, ctypes.byref(h)) == _capi.CMPS_ERR_UNSUPPORTED_D
    assert hip_lib.cmps_create(129, ctypes.byref(h)) == _capi.CMPS_ERR_UNSUPPORTED_D
    assert hip_lib.cmps_create(32, ctypes.byref(h)) == _capi.CMPS_OK and h.value
    assert hip_lib.cmps_get_variant(h) == _capi.CMPS_VARIANT_WAVE
    assert hip_lib.cmps_set_variant(h, _capi.CMPS_VARIANT_BLOCK) == _capi.CMPS_OK
    assert hip_lib.cmps_get_variant(h) == _capi.CMPS_VARIANT_BLOCK
    assert hip_lib.cmps_set_variant(h, _capi.CMPS_VARIANT_WIDE) == _capi.CMPS_ERR_UNSUPPORTED_D     # 32 < D only
    assert hip_lib.cmps_set_variant(h, 7) == _capi.CMPS_ERR_BAD_ARG
    assert b"unknown variant" in hip_lib.cmps_last_error(h)
    # call order is enforced before anything touches the device
    assert hip_lib.cmps_psi_loss_fwd(h, None, 1, 2, None, 0, None) == _capi.CMPS_ERR_STATE
    assert hip_lib.cmps_psi_loss_bwd(h, None, 1, 2, None, None) == _capi.CMPS_ERR_STATE
    assert hip_lib.cmps_destroy(h) == _capi.CMPS_OK
    h64 = ctypes.c_void_p()
    assert hip_lib.cmps_create(64, ctypes.byref(h64)) == _capi.CMPS_OK
    assert hip_lib.cmps_get_variant(h64) == _capi.CMPS_VARIANT_WIDE            # AUTO above 32: the float32 wide kernels
    assert hip_lib.cmps_set_variant(h64, _capi.CMPS_VARIANT_WAVE) == _capi.CMPS_ERR_UNSUPPORTED_D
    assert hip_lib.cmps_set_variant(h64, _capi.CMPS_VARIANT_PAIR) == _capi.CMPS_OK
    assert hip_lib.cmps_get_variant(h64) == _capi.CMPS_VARIANT_PAIR
    hip_lib.cmps_destroy(h64)


def test_apply_step_argument_checks(hip_lib):
    """cmps_psi_apply_step / cmps_set_params_dev / cmps_apply_step_scratch_bytes reject bad arguments before touching the device."""
    from audio_mps_amd import _capi
    assert hip_lib.cmps_apply_step_scratch_bytes(0) == 0 and hip_lib.cmps_apply_step_scratch_bytes(129) == 0
    assert hip_lib.cmps_apply_step_scratch_bytes(32) == 2 * 32 * 8 + 2 * 32 * 32 * 4
    h = ctypes.c_void_p()
    assert hip_lib.cmps_create(8, ctypes.byref(h)) == _capi.CMPS_OK
    z = [0.0] * 9
    assert hip_lib.cmps_psi_apply_step(h, None, None, None, None, *z, 1, None, None, None, None) == _capi.CMPS_ERR_BAD_ARG
    assert b"null variable" in hip_lib.cmps_last_error(h)
    # an update (grad_sums given) needs the Adam slots, the loss outputs, the scratch buffer and a positive batch
    assert hip_lib.cmps_psi_apply_step(h, 256, None, None, 256, 0.0, *z[1:], 1, 256, None, None, None) == _capi.CMPS_ERR_BAD_ARG
    assert hip_lib.cmps_psi_apply_step(h, 256, 256, 256, 256, 4.0, *z[1:], 1, 256, 256, 260, None) == _capi.CMPS_ERR_BAD_ARG   # misaligned scratch
    assert hip_lib.cmps_set_params_dev(h, None, 1e-4, 1 / 16000, 16, 1, 0, None, 0, None) == _capi.CMPS_ERR_BAD_ARG
    assert hip_lib.cmps_set_params_dev(h, 256, 1e-4, 1 / 16000, 16, 1, 0, None, 0, None) == _capi.CMPS_ERR_WORKSPACE
    hip_lib.cmps_destroy(h)


def test_workspace_bytes(hip_lib):
    from audio_mps_amd import _capi
    assert hip_lib.cmps_workspace_bytes(0, 1, 16, 0) == 0
    assert hip_lib.cmps_workspace_bytes(32, 1, 1, 0) == 0
    fwd = hip_lib.cmps_workspace_bytes(32, 1024, 16000, _capi.CMPS_WS_FWD_ONLY)
    trn = hip_lib.cmps_workspace_bytes(32, 1024, 16000, _capi.CMPS_WS_TRAIN)
    assert 4_000_000 < fwd < 8_000_000                     # rotation table 4.1 MB + small tables
    stash = 1024 * 15999 * 64 * 8                          # one 512-B row of (y, H y) per clip-step
    assert stash < trn < stash * 1.05                     # the per-step state stash dominates (8.4 GB at C3)


def test_workspace_bytes_pair_variant(hip_lib):
    """32 < D <= 128 with CMPS_WS_TRAIN: the stash of (y, H y) in float32 plus the reverse scan's ybar rows (float32), from which the
    gradient GEMM builds its operands -- include/cmps.h / cmps_internal.h::make_layout (round 2: five bf16 operand arrays, 21 GB)."""
    from audio_mps_amd import _capi
    D, B, T = 128, 512, 16000
    N, pairs = T - 1, B // 2
    stash = pairs * N * 2 * 2 * 2 * D * 4                   # [pair][step][y | H y][clip][re | im][D] float32
    gops = pairs * N * 2 * 2 * D * 4                        # [pair][step][clip][re | im][D] float32
    trn = hip_lib.cmps_workspace_bytes(D, B, T, _capi.CMPS_WS_TRAIN)
    assert stash + gops < trn < (stash + gops) * 1.02       # 16.8 GB + 8.4 GB at BASELINE configs[4]
    assert hip_lib.cmps_workspace_bytes(D, B, T, _capi.CMPS_WS_FWD_ONLY) < 64 * 1024 * 1024


def test_options_default_and_errors(hip_lib):
    """cmps_set_option / cmps_get_option (no device work): the rank-1 arithmetic defaults to DEFAULT (include/cmps.h's table: F16X2
    in the 32-row wave reverse scan, the wide kernels' GEMMs and the RhoCMPS GEMM forward / sampler; BF16X3 in legacy mode)."""
    from audio_mps_amd import _capi
    h = ctypes.c_void_p()
    assert hip_lib.cmps_create(32, ctypes.byref(h)) == _capi.CMPS_OK
    assert hip_lib.cmps_get_option(h, _capi.CMPS_OPT_RANK1) == _capi.CMPS_RANK1_DEFAULT == 4
    for v in (_capi.CMPS_RANK1_EXACT_F32, _capi.CMPS_RANK1_BF16X2, _capi.CMPS_RANK1_BF16X3, _capi.CMPS_RANK1_F16X2, _capi.CMPS_RANK1_DEFAULT):
        assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_RANK1, v) == _capi.CMPS_OK
        assert hip_lib.cmps_get_option(h, _capi.CMPS_OPT_RANK1) == v
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_RANK1, 5) == _capi.CMPS_ERR_BAD_ARG
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_RANK1, -1) == _capi.CMPS_ERR_BAD_ARG
    # CMPS_OPT_WIDE_CHAIN: a new handle runs the wide kernels' chains on the matrix cores; VALU and MFMA_FWD stay selectable
    assert hip_lib.cmps_get_option(h, _capi.CMPS_OPT_WIDE_CHAIN) == _capi.CMPS_WIDE_CHAIN_MFMA == 1
    for v in (_capi.CMPS_WIDE_CHAIN_VALU, _capi.CMPS_WIDE_CHAIN_MFMA_FWD, _capi.CMPS_WIDE_CHAIN_MFMA):
        assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_WIDE_CHAIN, v) == _capi.CMPS_OK
        assert hip_lib.cmps_get_option(h, _capi.CMPS_OPT_WIDE_CHAIN) == v
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_WIDE_CHAIN, 3) == _capi.CMPS_ERR_BAD_ARG
    # CMPS_OPT_RHO_BWD: the virtual-clip reverse sweep is a new handle's setting
    assert hip_lib.cmps_get_option(h, _capi.CMPS_OPT_RHO_BWD) == _capi.CMPS_RHO_BWD_VIRTUAL == 0
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_RHO_BWD, _capi.CMPS_RHO_BWD_GEMM) == _capi.CMPS_OK
    assert hip_lib.cmps_get_option(h, _capi.CMPS_OPT_RHO_BWD) == _capi.CMPS_RHO_BWD_GEMM
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_RHO_BWD, 2) == _capi.CMPS_ERR_BAD_ARG
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_RHO_BWD, _capi.CMPS_RHO_BWD_VIRTUAL) == _capi.CMPS_OK
    # CMPS_OPT_F16_SCALE_SHIFT (diagnostic: provokes CMPS_ERR_F16_RANGE in tests/test_gpu_parity.py): 0 on a new handle, -40 .. 40
    assert hip_lib.cmps_get_option(h, _capi.CMPS_OPT_F16_SCALE_SHIFT) == 0
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_F16_SCALE_SHIFT, 7) == _capi.CMPS_OK
    assert hip_lib.cmps_get_option(h, _capi.CMPS_OPT_F16_SCALE_SHIFT) == 7
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_F16_SCALE_SHIFT, 41) == _capi.CMPS_ERR_BAD_ARG
    assert hip_lib.cmps_set_option(h, _capi.CMPS_OPT_F16_SCALE_SHIFT, 0) == _capi.CMPS_OK
    # cmps_psi_grad_status before any cmps_set_params: a call-order error, never a crash; the null handle is a bad argument
    sticky = ctypes.c_int(5)
    assert hip_lib.cmps_psi_grad_status(h, ctypes.byref(sticky), None) == _capi.CMPS_ERR_STATE and sticky.value == 0
    assert b"cmps_set_params" in hip_lib.cmps_last_error(h)
    assert hip_lib.cmps_psi_grad_status(None, None, None) == _capi.CMPS_ERR_BAD_ARG
    assert _capi.CMPS_ERR_F16_RANGE == 6
    assert hip_lib.cmps_set_option(h, 99, 0) == _capi.CMPS_ERR_BAD_ARG
    assert hip_lib.cmps_get_option(h, 99) == -1 and hip_lib.cmps_get_option(None, _capi.CMPS_OPT_RANK1) == -1
    # CMPS_WS_FRESH / CMPS_WS_REUSE_TABLES are requests, not layouts: they do not change the size
    assert hip_lib.cmps_workspace_bytes(8, 4, 64, _capi.CMPS_WS_TRAIN | _capi.CMPS_WS_REUSE_TABLES) == \
        hip_lib.cmps_workspace_bytes(8, 4, 64, _capi.CMPS_WS_TRAIN)
    assert hip_lib.cmps_workspace_bytes(8, 4, 64, _capi.CMPS_WS_TRAIN | _capi.CMPS_WS_FRESH) == \
        hip_lib.cmps_workspace_bytes(8, 4, 64, _capi.CMPS_WS_TRAIN)
    hip_lib.cmps_destroy(h)


@pytest.mark.parametrize("source,flags", [
    ("cmps_wave2.hip", ["-DCMPS_DIAG_NO_LOSS"]), ("cmps_wave2.hip", ["-DCMPS_DIAG_NO_CHAIN"]),
    ("cmps_pair.hip", ["-DPABL_NO_MFMA", "-DPABL_NO_BARRIER"]), ("cmps_pair.hip", ["-DPABL_TIMING"]),
    ("cmps_wave16.hip", ["-DW16_TIMING"]), ("cmps_wide.hip", ["-DWABL_NO_LOSSMV", "-DWABL_GRAD_NO_LOADS", "-DWABL_GRAD_NO_STORE"]),
    ("cmps_pair.hip", ["-DWABL_GRAD_NO_SLICES"]), ("cmps_pair.hip", ["-DC16_TIMING", "-DPABL_READ_BURST"]),
])
def test_diagnostic_switches_compile(source, flags, tmp_path):
    """The timing-only switches of scripts/ablate.py live behind -DCMPS_DIAG; each must keep compiling for gfx950."""
    import subprocess
    from audio_mps_amd import build
    out = os.path.join(tmp_path, "diag.o")
    cmd = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-DCMPS_DIAG"] + flags + \
          [os.path.join(build.CSRC, source), "-o", out]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]


def test_no_valu_write_next_to_mfma_read(hip_lib):
    """Static check of the built code objects (scripts/check_mfma_hazards.py): no v_mfma reads a VGPR that a VALU instruction wrote
    fewer than two wait states earlier.  hipcc keeps that distance for its own instructions but not around inline asm, of which
    the kernels have plenty; the matrix core then reads the register's previous contents (round 4, k_fwd_rho_mfma)."""
    import subprocess
    import sys
    from audio_mps_amd import build
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_mfma_hazards.py"), build.LIB_PATH],
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert " 0 closer than" in proc.stdout and "MFMA instructions checked" in proc.stdout
