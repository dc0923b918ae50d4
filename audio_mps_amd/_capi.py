"""ctypes binding of libcmps.so (include/cmps.h).  There is NO fallback: if the HIP library is missing
or a symbol is absent this module raises, and everything built on it fails loudly."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMPS_LIB") or os.path.join(_HERE, "lib", "libcmps.so")   # CMPS_LIB: diagnostic builds

CMPS_OK = 0
CMPS_ERR_BAD_ARG = 1
CMPS_ERR_UNSUPPORTED_D = 2
CMPS_ERR_WORKSPACE = 3
CMPS_ERR_HIP = 4
CMPS_ERR_STATE = 5
CMPS_ERR_F16_RANGE = 6

CMPS_WS_FWD_ONLY = 0
CMPS_WS_TRAIN = 1
CMPS_WS_FRESH = 2
CMPS_WS_REUSE_TABLES = 4

CMPS_VARIANT_AUTO = 0
CMPS_VARIANT_BLOCK = 1
CMPS_VARIANT_WAVE = 2
CMPS_VARIANT_PAIR = 3
CMPS_VARIANT_WAVE32 = 4
CMPS_VARIANT_WIDE = 5

CMPS_OPT_RANK1 = 1
CMPS_OPT_KERNEL_EVENTS = 2
CMPS_RANK1_EXACT_F32 = 0
CMPS_RANK1_BF16X2 = 1
CMPS_RANK1_BF16X3 = 2
CMPS_RANK1_F16X2 = 3
CMPS_RANK1_DEFAULT = 4
CMPS_OPT_WIDE_CHAIN = 3
CMPS_OPT_F16_SCALE_SHIFT = 4
CMPS_OPT_RHO_BWD = 5
CMPS_OPT_BWD_WAVES = 6
CMPS_RHO_BWD_VIRTUAL = 0
CMPS_RHO_BWD_GEMM = 1
CMPS_WIDE_CHAIN_VALU = 0
CMPS_WIDE_CHAIN_MFMA = 1
CMPS_WIDE_CHAIN_MFMA_FWD = 2
RANK1_NAMES = {0: "exact_f32", 1: "bf16x2", 2: "bf16x3", 3: "f16x2", 4: "default"}

# every symbol include/cmps.h declares
SYMBOLS = (
    "cmps_version", "cmps_create", "cmps_destroy", "cmps_last_error", "cmps_set_variant",
    "cmps_get_variant", "cmps_set_option", "cmps_get_option", "cmps_kernel_times", "cmps_workspace_bytes", "cmps_set_params", "cmps_set_params_dev",
    "cmps_apply_step_scratch_bytes", "cmps_psi_apply_step", "cmps_psi_loss_fwd",
    "cmps_psi_loss_bwd", "cmps_psi_grad_status", "cmps_psi_update_ancilla", "cmps_psi_states", "cmps_psi_sample",
    "cmps_legacy_set_params", "cmps_legacy_loss_fwd", "cmps_legacy_loss_bwd",
    "cmps_rho_workspace_bytes", "cmps_rho_set_state", "cmps_rho_loss_fwd", "cmps_rho_loss_bwd",
    "cmps_rho_update_ancilla", "cmps_rho_sample", "cmps_rho_states", "cmps_crc32c",
)


class CmpsError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libcmps error {code}: {message}")
        self.code = code


def _declare(lib):
    c_int, c_float, c_double, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t
    vp = ctypes.c_void_p
    lib.cmps_version.argtypes = []
    lib.cmps_version.restype = c_int
    lib.cmps_create.argtypes = [c_int, ctypes.POINTER(vp)]
    lib.cmps_create.restype = c_int
    lib.cmps_destroy.argtypes = [vp]
    lib.cmps_destroy.restype = c_int
    lib.cmps_last_error.argtypes = [vp]
    lib.cmps_last_error.restype = ctypes.c_char_p
    lib.cmps_set_variant.argtypes = [vp, c_int]
    lib.cmps_set_variant.restype = c_int
    lib.cmps_get_variant.argtypes = [vp]
    lib.cmps_get_variant.restype = c_int
    lib.cmps_set_option.argtypes = [vp, c_int, c_int]
    lib.cmps_set_option.restype = c_int
    lib.cmps_get_option.argtypes = [vp, c_int]
    lib.cmps_get_option.restype = c_int
    lib.cmps_kernel_times.argtypes = [vp, ctypes.c_char_p, c_size_t, ctypes.POINTER(c_float), ctypes.POINTER(c_int), c_int]
    lib.cmps_kernel_times.restype = c_int
    lib.cmps_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.cmps_workspace_bytes.restype = c_size_t
    lib.cmps_set_params.argtypes = [vp, vp, vp, vp, vp, vp, c_float, c_double, c_double, c_int, c_int,
                                    c_int, vp, c_size_t, vp]
    lib.cmps_set_params.restype = c_int
    lib.cmps_set_params_dev.argtypes = [vp, vp, c_double, c_double, c_int, c_int, c_int, vp, c_size_t, vp]
    lib.cmps_set_params_dev.restype = c_int
    lib.cmps_apply_step_scratch_bytes.argtypes = [c_int]
    lib.cmps_apply_step_scratch_bytes.restype = c_size_t
    lib.cmps_psi_apply_step.argtypes = [vp, vp, vp, vp, vp, c_double, c_double, c_double, c_double, c_double, c_double, c_double,
                                        c_double, c_double, c_int, vp, vp, vp, vp]
    lib.cmps_psi_apply_step.restype = c_int
    lib.cmps_psi_loss_fwd.argtypes = [vp, vp, c_int, c_int, vp, c_int, vp]
    lib.cmps_psi_loss_fwd.restype = c_int
    lib.cmps_psi_loss_bwd.argtypes = [vp, vp, c_int, c_int, vp, vp]
    lib.cmps_psi_loss_bwd.restype = c_int
    lib.cmps_psi_grad_status.argtypes = [vp, ctypes.POINTER(c_int), vp]
    lib.cmps_psi_grad_status.restype = c_int
    lib.cmps_psi_update_ancilla.argtypes = [vp, vp, vp, c_float, c_int, vp, vp]
    lib.cmps_psi_update_ancilla.restype = c_int
    lib.cmps_psi_states.argtypes = [vp, c_int, c_int, vp, vp]
    lib.cmps_psi_states.restype = c_int
    lib.cmps_psi_sample.argtypes = [vp, vp, c_int, c_int, vp, vp]
    lib.cmps_psi_sample.restype = c_int
    lib.cmps_legacy_set_params.argtypes = [vp, vp, vp, vp, c_double, c_int, c_int, c_int, vp, c_size_t, vp]
    lib.cmps_legacy_set_params.restype = c_int
    lib.cmps_legacy_loss_fwd.argtypes = [vp, vp, c_int, c_int, vp, c_int, vp]
    lib.cmps_legacy_loss_fwd.restype = c_int
    lib.cmps_legacy_loss_bwd.argtypes = [vp, vp, c_int, c_int, vp, vp]
    lib.cmps_legacy_loss_bwd.restype = c_int
    lib.cmps_rho_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.cmps_rho_workspace_bytes.restype = c_size_t
    lib.cmps_rho_set_state.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, vp, c_size_t, vp]
    lib.cmps_rho_set_state.restype = c_int
    lib.cmps_rho_loss_fwd.argtypes = [vp, vp, c_int, c_int, vp, c_int, vp]
    lib.cmps_rho_loss_fwd.restype = c_int
    lib.cmps_rho_loss_bwd.argtypes = [vp, vp, c_int, c_int, vp, vp]
    lib.cmps_rho_loss_bwd.restype = c_int
    lib.cmps_rho_update_ancilla.argtypes = [vp, vp, vp, c_float, c_int, vp, vp]
    lib.cmps_rho_update_ancilla.restype = c_int
    lib.cmps_rho_sample.argtypes = [vp, vp, c_int, c_int, vp, c_int, vp]
    lib.cmps_rho_sample.restype = c_int
    lib.cmps_rho_states.argtypes = [vp, c_int, c_int, vp, vp, vp]
    lib.cmps_rho_states.restype = c_int
    lib.cmps_crc32c.argtypes = [ctypes.c_char_p, c_size_t, ctypes.c_uint]
    lib.cmps_crc32c.restype = ctypes.c_uint


_lib = None


def load():
    """Load libcmps.so (built in-tree by audio_mps_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library has not been built "
            "(run `python -m audio_mps_amd.build` or `__graft_entry__.build()`); there is no CPU fallback")
    # The device buffers and streams handed to libcmps come from PyTorch's HIP runtime, so libcmps must bind to
    # that same libamdhip64 instance: import torch first so its runtime is the one already loaded.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} lacks symbols {missing}")
    _declare(lib)
    _lib = lib
    return lib


def check(handle, code: int):
    if code != CMPS_OK:
        msg = load().cmps_last_error(handle)
        raise CmpsError(code, msg.decode() if msg else "?")
