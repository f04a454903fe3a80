"""ctypes binding of libdta_mi355x.so (include/dta.h).  There is no fallback: every entry point
raises if the library is missing or a call returns a non-zero status."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB

_lib = None
_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

_PROTOS = {
    "dta_version": ([], C.c_int),
    "dta_take_pending_error": ([C.c_char_p, _i32], C.c_int),
    "dta_lcp_adjacent": ([_vp, _vp, _vp, _i32, _vp, _vp, _vp], C.c_int),
    "dta_leafize": ([_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp], C.c_int),
    "dta_preorder_meta": ([_vp] * 8 + [_i32, _i32] + [_vp] * 4 + [_vp], C.c_int),
    "dta_tree_attn_fwd": ([_vp] * 8 + [_i32] * 6 + [_i64] * 3 + [_f32, _i32, _vp], C.c_int),
    "dta_tree_attn_bwd": ([_vp] * 14 + [_i32] * 6 + [_i64] * 5 + [_f32, _i32, _i32, _vp], C.c_int),
    "dta_tree_attn_fwd_ex": ([_vp] * 8 + [_i32] * 6 + [_i64] * 8 + [_f32, _i32, _vp], C.c_int),
    "dta_tree_attn_bwd_ex": ([_vp] * 14 + [_i32] * 6 + [_i64] * 12 + [_f32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp], C.c_int),
    "dta_logprob_entropy_fwd": ([_vp] * 8 + [_i32, _i32, _i64, _f32, _i32, _vp], C.c_int),
    "dta_logprob_entropy_shard_stats": ([_vp] * 6 + [_i32, _i32, _i64, _f32, _i32, _vp], C.c_int),
    "dta_logprob_entropy_bwd": ([_vp] * 10 + [_i32, _i32, _i64, _i64, _f32, _i32, _vp], C.c_int),
    "dta_rmsnorm_fwd": ([_vp] * 6 + [_i32, _i32, _f32, _i32, _vp], C.c_int),
    "dta_rmsnorm_bwd_blocks": ([_i32], C.c_int),
    "dta_rmsnorm_bwd": ([_vp] * 7 + [_i32, _i32, _i32, _vp], C.c_int),
    "dta_qk_norm_rope_fwd": ([_vp] * 5 + [_i32, _i32, _i32, _i64, _f32, _i32, _vp], C.c_int),
    "dta_qk_norm_rope_bwd_blocks": ([_i64], C.c_int),
    "dta_qk_norm_rope_bwd": ([_vp] * 7 + [_i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _vp], C.c_int),
    "dta_swiglu_fwd": ([_vp] * 3 + [_i64, _i32, _i64, _i32, _vp], C.c_int),
    "dta_swiglu_bwd": ([_vp] * 5 + [_i64, _i32, _i64, _i64, _i32, _vp], C.c_int),
    "dta_transpose": ([_vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp], C.c_int),
    "dta_sum_slabs": ([_vp, _i64, _i64, _i64, _vp, _vp, _i32, _vp], C.c_int),
}
EXPORTS = tuple(_PROTOS)
_ERR = {-1: "DTA_EINVAL", -2: "DTA_EUNSUPPORTED", -3: "DTA_EALIGN", -4: "DTA_ELAUNCH", -5: "DTA_EPRIOR"}


def _hip_runtimes_mapped():
    """Paths of every libamdhip64 mapped into this process (Linux)."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    except OSError:
        return []


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} is missing: run `python -m dynamictreeattn_amd.build` (hipcc, gfx950). "
                               "There is no CPU fallback for the hot path.")
        # torch ships its own libamdhip64; it must be resident BEFORE this library so that both resolve
        # to ONE HIP runtime (otherwise torch's streams/pointers are foreign to our kernels: launches fail).
        import torch  # noqa: F401
        l = C.CDLL(LIB)
        rts = _hip_runtimes_mapped()
        if len(rts) > 1:
            # two HIP runtimes = two sets of streams/contexts: torch's device pointers are foreign to our kernels and
            # every launch fails (round 1: "dta_lcp_adjacent failed: DTA_ELAUNCH").  Refuse to run like that.
            raise RuntimeError("more than one libamdhip64 is mapped into this process: " + ", ".join(rts) +
                               " — libdta_mi355x.so must resolve to the HIP runtime torch ships")
        for name, (args, res) in _PROTOS.items():
            fn = getattr(l, name)
            fn.argtypes, fn.restype = args, res
        _lib = l
    return _lib


def check(status: int, what: str):
    if status == 0:
        return
    if status == -5:
        buf = C.create_string_buffer(256)
        code = lib().dta_take_pending_error(buf, 256)
        raise RuntimeError(f"{what}: not launched, a HIP error was already pending on this thread: {buf.value.decode() or code} "
                           f"(hipError {code}; now cleared — after an asynchronous kernel fault the GPU context stays unusable)")
    raise RuntimeError(f"{what} failed: {_ERR.get(status, status)}")


def ptr(t):
    return None if t is None else t.data_ptr()
