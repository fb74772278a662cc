"""ctypes binding of libp2i_hip.so (C ABI declared in include/p2i_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C p2i-gan-benchmark_amd/csrc``.
There is NO fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libp2i_hip.so")
_lib = None

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH = 0, 1, 2, 3


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("B", "Cin", "Cout", "Ti", "Hi", "Wi", "To", "Ho", "Wo", "kt", "kh", "kw",
                 "st", "sh", "sw", "pt", "ph", "pw")]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_int64
_D = C.POINTER(ConvDesc)

# name -> argtypes (every entry point returns int); must list EVERY symbol of include/p2i_hip.h
SIGNATURES = {
    "p2i_conv_fwd": [_D, _P, _P, _P, _P, _P, _I, _P],
    "p2i_conv_dgrad": [_D, _P, _P, _I, _P, _P, _P, _I, _P, _P],
    "p2i_conv_wgrad": [_D, _P, _P, _P, _I, _P, _P, _P],
    "p2i_conv_wgrad_ws": [_D, _P, _P, _P, _I, _P, _P, _P, _L, _P],
    "p2i_conv_fwd_x6": [_D, _P, _P, _P, _P, _P, _P, _I, _P],
    "p2i_conv_dgrad_x6": [_D, _P, _P, _P, _P, _P, _I, _P, _P],
    "p2i_x6_split": [_P, _P, _I, _I, _I, _P],
    "p2i_x6_split_batched": [_P, _P, _P, _P, _P, _I, _P],
    "p2i_x6c_would_take": [_D, _I, _I],
    "p2i_conv_fwd_x6s": [_D, _P, _P, _P, _I, _P, _P, _P, _I, _P],
    "p2i_conv_dgrad_x6s": [_D, _P, _P, _P, _I, _P, _P, _I, _P, _P],
    "p2i_conv_last_plan": [C.POINTER(C.c_int)],
    "p2i_wgrad_last_plan": [C.POINTER(C.c_int)],
    "p2i_doconv_fold_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "p2i_doconv_fold_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "p2i_doconv_fold_fwd_batched": [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    "p2i_doconv_fold_bwd_batched": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "p2i_weight_pack": [_P, _I, _I, _I, _P, _P, _P, _P],
    "p2i_weight_pack_batched": [_P, _P, _P, _P, _P, _P, _P, _I, _P],
    "p2i_weight_unpack_grad_batched": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "p2i_weight_unpack_grad_batched_acc": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "p2i_weight_unpack_grad": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "p2i_spectral_norm": [_P, _I, _I, _P, _P, _I, _P, _P, _P],
    "p2i_spectral_norm_batched": [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "p2i_attn_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "p2i_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "p2i_idw_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "p2i_idw_fwd_ws": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "p2i_idw_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "p2i_pooldup_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "p2i_pooldup_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "p2i_upmod_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "p2i_window_gather": [_P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P],
    "p2i_window_mean": [_P, _P, _I, _L, _I, _I, _I, _F, _P],
    "p2i_upmod_fwd_ba": [_P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "p2i_upmod_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "p2i_dtail_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "p2i_dtail_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "p2i_recloss": [_P, _P, _F, _P, _P, _P, _I, _I, _I, _P],
    "p2i_gan_loss": [_P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P],
    "p2i_adam": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _P],
    "p2i_adam_dev": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P, _P],
    "p2i_metrics_pointwise": [_P, _P, _L, C.POINTER(C.c_float), _I, _I, _P, _P, _P, _P],
    "p2i_metrics_fss": [_P, _I, _I, _I, _I, C.POINTER(C.c_int), _I, _P, _P, _P],
    "p2i_assemble_batch": [_P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _P],
    "p2i_axpy": [_P, _P, _F, _L, _P],
    "p2i_add2": [_P, _P, _P, _L, _P],
    "p2i_zero": [_P, _L, _P],
    "p2i_act_bwd": [_P, _P, _I, _P, _L, _P],
    "p2i_bias_grad": [_P, _P, _I, _P, _I, _I, _L, _P],
    "p2i_act_bwd_bias": [_P, _P, _I, _P, _P, _I, _I, _L, _P],
    "p2i_det_workspace": [_P, _L, _P, _I],
    "p2i_x6_split_planes": [_P, _P, _I, _I, _L, _P],
    "p2i_x6_next_source_planes": [_P],
    "p2i_event_record": [_I, _P],
    "p2i_event_wait": [_I, _P],
    "p2i_tape_begin": [_P],
    "p2i_tape_end": [C.POINTER(C.c_void_p)],
    "p2i_tape_info": [_P, C.POINTER(C.c_int)],
    "p2i_tape_replay": [_P, _P],
    "p2i_tape_free": [_P],
}


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load the HIP library (once).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"libp2i_hip.so not found at {_LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C p2i-gan-benchmark_amd/csrc`.  There is no CPU fallback for the product path.")
    # P2I_HIP_LIB: another build of the SAME library (kernel A/B experiments, tools/); never a different backend
    lib = C.CDLL(os.environ.get("P2I_HIP_LIB") or _LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.p2i_abi_version.restype = C.c_int
    lib.p2i_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().p2i_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
