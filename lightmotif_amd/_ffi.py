"""ctypes binding of ``liblightmotif_hip.so`` (the C ABI of include/lightmotif_hip.h).

There is no CPU fallback: if the library is missing this module raises at import
of the symbol table, and every call that needs a device fails with the library's
own error (``LM_HIP_ERR_NO_DEVICE`` without a gfx950 GPU).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "liblightmotif_hip.so"

OK, ERR_BAD_ARGS, ERR_WRAP, ERR_HIP, ERR_OOM, ERR_NO_DEVICE, ERR_INVALID_SYMBOL, ERR_CAPACITY, ERR_COMM = range(9)


class Coords(C.Structure):
    _fields_ = [("row", C.c_size_t), ("col", C.c_size_t)]


class Hit(C.Structure):
    _fields_ = [("position", C.c_size_t), ("score", C.c_float)]


class LightmotifHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"lightmotif_hip error {status}: {message}")
        self.status = status
        self.message = message


class UnsupportedBackend(LightmotifHipError):
    """No gfx950 device (err.rs:34 ``UnsupportedBackend``)."""


class InvalidSymbol(ValueError):
    """err.rs:10 ``InvalidSymbol``."""


_sz = C.c_size_t
_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_szp = C.POINTER(C.c_size_t)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)
_cp = C.POINTER(Coords)

# name -> (restype, argtypes); also the list the CPU test-suite checks exports against
SIGNATURES = {
    "lm_hip_abi_version": (C.c_int, []),
    "lm_hip_last_error": (C.c_char_p, []),
    "lm_hip_device_count": (C.c_int, [_ip]),
    "lm_hip_device_ordinal": (C.c_int, [C.c_int, _ip]),
    "lm_hip_free": (None, [_vp]),
    "lm_hip_result_pool_info": (C.c_int, [_szp, _szp, _szp]),
    "lm_hip_device_clock_mhz": (C.c_int, [C.c_int, C.c_uint, C.POINTER(C.c_double)]),
    "lm_hip_stride": (_sz, [_sz, _sz]),
    "lm_hip_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "lm_hip_ctx_create_on_stream": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "lm_hip_ctx_destroy": (C.c_int, [_vp]),
    "lm_hip_ctx_sync": (C.c_int, [_vp]),
    "lm_hip_ctx_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "lm_hip_ctx_set_rows_per_stream": (C.c_int, [_vp, _sz]),
    "lm_hip_ctx_set_prefilter": (C.c_int, [_vp, C.c_int]),
    "lm_hip_ctx_set_track_argmax": (C.c_int, [_vp, C.c_int]),
    "lm_hip_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "lm_hip_ctx_set_xcd_remap": (C.c_int, [_vp, C.c_int]),
    "lm_hip_ctx_last_kernel": (C.c_char_p, [_vp]),
    "lm_hip_ctx_last_scan_counts": (C.c_int, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "lm_hip_ctx_last_scan_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "lm_hip_ctx_last_phases_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "lm_hip_ctx_last_scan_info": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lm_hip_pssm_create": (C.c_int, [_vp, _vp, _sz, _sz, _sz, C.POINTER(_vp)]),
    "lm_hip_pssm_reverse_complement": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "lm_hip_pssm_destroy": (C.c_int, [_vp]),
    "lm_hip_pssm_len": (_sz, [_vp]),
    "lm_hip_score_f32_dptr": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _vp, _sz,
                                       _szp, _szp]),
    "lm_hip_argmax_f32_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _ip, _cp, _fp]),
    "lm_hip_max_f32_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _ip, _fp]),
    "lm_hip_argmax_shard_f32_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, C.c_int, _ip, _cp, _fp]),
    "lm_hip_threshold_f32_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, C.c_float, C.POINTER(_cp), _szp]),
    "lm_hip_score_u8_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz,
                                      _vp, _sz, C.c_int, _szp, _szp]),
    "lm_hip_score_u8": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _vp, _sz, _sz, C.c_int, _vp, _sz, _szp, _szp]),
    "lm_hip_argmax_u8_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _ip, _cp, C.POINTER(C.c_uint8)]),
    "lm_hip_threshold_u8_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, C.c_uint8, C.POINTER(_cp), _szp]),
    "lm_hip_score_argmax_f32_dptr": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz,
                                              _ip, _cp, _fp]),
    "lm_hip_score_argmax_shard_f32_dptr": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz,
                                                    _sz, C.c_int, _ip, _cp, _fp]),
    "lm_hip_score_threshold_f32_dptr": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz,
                                                 C.c_float, C.POINTER(_cp), C.POINTER(_fp), _szp]),
    "lm_hip_scan_argmax_batch": (C.c_int, [_vp, C.POINTER(_vp), _sz, _vp, _ip, _cp, _fp]),
    "lm_hip_scan_threshold_batch": (C.c_int, [_vp, C.POINTER(_vp), _fp, _sz, _vp, _szp,
                                              C.POINTER(_cp), C.POINTER(_fp)]),
    "lm_hip_scan_f32": (C.c_int, [_vp, _vp, _vp, C.c_float, C.POINTER(C.POINTER(Hit)), _szp]),
    "lm_hip_scan_max_f32": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.c_int, C.c_uint, C.c_int, _sz, C.c_float, _sz, _ip,
                                      C.POINTER(Hit)]),
    "lm_hip_encode_dptr": (C.c_int, [_vp, C.c_char, _vp, _sz, C.c_int, _vp, _szp]),
    "lm_hip_stripe_dptr": (C.c_int, [_vp, _vp, _sz, _sz, C.c_uint8, _sz, _vp, _sz]),
    "lm_hip_configure_wrap_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _sz, C.c_uint8]),
    "lm_hip_seq_upload": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, C.POINTER(_vp)]),
    "lm_hip_seq_from_encoded": (C.c_int, [_vp, _vp, _sz, _sz, _sz, C.POINTER(_vp)]),
    "lm_hip_seq_from_ascii": (C.c_int, [_vp, C.c_char, _vp, _sz, _sz, C.c_int, C.POINTER(_vp), _szp]),
    "lm_hip_seq_from_2bit": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, C.POINTER(_vp)]),
    "lm_hip_seq_configure_wrap": (C.c_int, [_vp, _vp, _sz]),
    "lm_hip_seq_info": (C.c_int, [_vp, _szp, _szp, _szp, _szp, _szp, C.POINTER(_vp)]),
    "lm_hip_seq_download": (C.c_int, [_vp, _vp, _vp]),
    "lm_hip_seq_destroy": (C.c_int, [_vp]),
    "lm_hip_scores_create": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "lm_hip_scores_info": (C.c_int, [_vp, _szp, _szp, _szp, _szp, C.POINTER(_vp)]),
    "lm_hip_scores_download": (C.c_int, [_vp, _vp, _vp]),
    "lm_hip_scores_download_rows": (C.c_int, [_vp, _vp, _sz, _sz, _vp]),
    "lm_hip_scores_destroy": (C.c_int, [_vp]),
    "lm_hip_score_rows_into": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "lm_hip_score_into": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lm_hip_argmax": (C.c_int, [_vp, _vp, _ip, _cp, _fp]),
    "lm_hip_max": (C.c_int, [_vp, _vp, _ip, _fp]),
    "lm_hip_threshold": (C.c_int, [_vp, _vp, C.c_float, C.POINTER(_cp), _szp]),
    # row-sharded jobs
    "lm_hip_seq_adopt_dptr": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, C.POINTER(_vp)]),
    "lm_hip_scores_set_first_cell_rule": (C.c_int, [_vp, C.c_int]),
    "lm_hip_combine_argmax": (C.c_int, [_ip, _cp, _fp, _sz, _ip, _cp, _fp]),
    "lm_hip_comm_unique_id": (C.c_int, [_vp]),
    "lm_hip_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lm_hip_comm_destroy": (C.c_int, [_vp]),
    "lm_hip_comm_info": (C.c_int, [_vp, _ip, _ip]),
    "lm_hip_exchange_halo_dptr": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint8]),
    "lm_hip_merge_argmax": (C.c_int, [_vp, _vp, C.c_int, _cp, C.c_float, _sz, _ip, _cp, _fp]),
    "lm_hip_argmax_sharded": (C.c_int, [_vp, _vp, _vp, _sz, _ip, _cp, _fp]),
    "lm_hip_argmax_sharded_begin": (C.c_int, [_vp, _vp, _vp, _sz, _ip]),
    "lm_hip_argmax_sharded_end": (C.c_int, [_vp, _vp, C.c_int, _ip, _cp, _fp]),
    "lm_hip_merge_max": (C.c_int, [_vp, _vp, C.c_int, C.c_float, _ip, _fp]),
    "lm_hip_merge_threshold": (C.c_int, [_vp, _vp, _cp, _sz, _sz, C.POINTER(_cp), _szp]),
    "lm_hip_score_f32": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _sz, _vp, _sz, _sz, _sz, _sz, _sz, _vp,
                                  _sz, _szp, _szp]),
    "lm_hip_score_u8_host": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _sz, _vp, _sz, _sz, _sz, _sz, _sz, C.c_int, _vp, _sz, _szp, _szp]),
    "lm_hip_argmax_f32": (C.c_int, [_vp, _sz, _sz, _sz, _ip, _cp, _fp]),
    "lm_hip_max_f32": (C.c_int, [_vp, _sz, _sz, _sz, _ip, _fp]),
    "lm_hip_threshold_f32": (C.c_int, [_vp, _sz, _sz, _sz, C.c_float, C.POINTER(_cp), _szp]),
    "lm_hip_scan_f32_host": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _sz, _vp, _sz, _sz, _sz, C.c_float, C.POINTER(C.POINTER(Hit)), _szp]),
    "lm_hip_scan_max_f32_host": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _sz, _vp, _sz, _sz, _sz, _vp, _sz, C.c_int, C.c_uint, C.c_int,
                                           _sz, C.c_float, _sz, _ip, C.POINTER(Hit)]),
    "lm_hip_host_crossover": (C.c_int, [C.c_int, _sz, _sz, _szp]),
    "lm_hip_host_calibrate": (C.c_int, [C.c_double, C.c_int]),
    "lm_hip_host_set_cpu_cost": (C.c_int, [C.c_int, C.c_double, C.c_double]),
    "lm_hip_host_set_crossover": (C.c_int, [C.c_int, C.c_size_t]),
    "lm_hip_host_cost_model": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    "lm_hip_host_trim": (C.c_int, []),
    "lm_hip_host_bind_thread": (C.c_int, [C.c_int]),
    "lm_hip_host_spread_lanes": (C.c_int, [C.c_int]),
    "lm_hip_host_reuse_scores": (C.c_int, [C.c_int]),
    "lm_hip_host_reuse_count": (C.c_int, [C.POINTER(C.c_size_t)]),
    "lm_hip_host_lane_info": (C.c_int, [_ip, _ip, _ip]),
}

_lib = None


def lib() -> C.CDLL:
    """Loads the shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        # LM_HIP_LIBRARY: another build of the same ABI (A/B runs of two kernel versions)
        path = Path(os.environ.get("LM_HIP_LIBRARY", LIB_PATH))
        if not path.exists():
            raise ImportError(
                f"{path} is missing: build it with `python -m lightmotif_amd.build` "
                "(there is no CPU fallback)")
        L = C.CDLL(str(path))
        # (an OLDER build under LM_HIP_LIBRARY, for bisecting: entry points it lacks are skipped when LM_HIP_LIBRARY_OLDER is set)
        lenient = "LM_HIP_LIBRARY" in os.environ and os.environ.get("LM_HIP_LIBRARY_OLDER")
        for name, (res, args) in SIGNATURES.items():
            if lenient and not hasattr(L, name):
                continue
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.lm_hip_abi_version() != 1:
            raise ImportError("lightmotif_hip ABI version mismatch")
        _lib = L
    return _lib


def last_error() -> str:
    return lib().lm_hip_last_error().decode("utf-8", "replace")


def check(status: int) -> None:
    if status == OK:
        return
    msg = lib().lm_hip_last_error().decode("utf-8", "replace")
    if status == ERR_NO_DEVICE:
        raise UnsupportedBackend(status, msg)
    if status == ERR_INVALID_SYMBOL:
        raise InvalidSymbol(msg)
    raise LightmotifHipError(status, msg)
