"""ctypes binding of libganffn.so (the C ABI in include/ganffn.h).

The HIP library is the product path: if it is missing or a call fails, this module
raises — there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GANFFN_LIB") or os.path.join(_HERE, "lib", "libganffn.so")   # GANFFN_LIB: tuning-lab builds only


class EncCfg(C.Structure):
    _fields_ = [("S", C.c_int32), ("B", C.c_int32), ("E", C.c_int32), ("H", C.c_int32), ("F", C.c_int32),
                ("L", C.c_int32), ("p_pe", C.c_float), ("p_enc", C.c_float), ("ln_eps", C.c_float),
                ("train", C.c_int32)]


class LstmCfg(C.Structure):
    """ganffn_lstm_cfg"""
    _fields_ = [("S", C.c_int32), ("B", C.c_int32), ("In", C.c_int32), ("H", C.c_int32)]


class HeadCfg(C.Structure):
    _fields_ = [("T", C.c_int32), ("E", C.c_int32), ("D1", C.c_int32), ("D2", C.c_int32), ("kind", C.c_int32),
                ("p", C.c_float), ("train", C.c_int32)]


class DrnnCfg(C.Structure):
    _fields_ = [("S", C.c_int32), ("B", C.c_int32), ("Dm", C.c_int32), ("H", C.c_int32), ("He", C.c_int32),
                ("p", C.c_float), ("train", C.c_int32)]


DRNN_PARAM_FIELDS = ["g_wih", "g_whh", "g_bih", "g_bhh", "p_wih", "p_whh", "p_bih", "p_bhh", "e_wih", "e_whh", "e_bih",
                     "e_bhh", "att_w"]


class DrnnPtrs(C.Structure):        # ganffn_drnn_params / ganffn_drnn_grads: 13 pointers
    _fields_ = [(n, C.c_void_p) for n in DRNN_PARAM_FIELDS]


_P = C.c_void_p
_I, _L, _F, _U32, _U64 = C.c_int, C.c_int64, C.c_float, C.c_uint32, C.c_uint64
_PE, _PH = C.POINTER(EncCfg), C.POINTER(HeadCfg)

# name -> (restype, argtypes); must cover every symbol include/ganffn.h declares
SIGNATURES = {
    "ganffn_version": (_I, []),
    "ganffn_last_error": (C.c_char_p, []),
    "ganffn_layer_param_count": (_L, [_I, _I]),
    "ganffn_layer_param_offsets": (_I, [_I, _I, C.POINTER(C.c_int64)]),
    "ganffn_encoder_saved_floats": (_L, [_PE]),
    "ganffn_encoder_saved_hidden_offset": (_L, [_PE, _I]),
    "ganffn_encoder_workspace_floats": (_L, [_PE]),
    "ganffn_head_saved_floats": (_L, [_PH]),
    "ganffn_head_workspace_floats": (_L, [_PH]),
    "ganffn_rng_advance": (_I, [_P, _U64, _P]),
    "ganffn_pe_table": (_I, [_P, _I, _I, _P]),
    "ganffn_encoder_fwd": (_I, [_PE, _P, _P, _P, _P, _P, _P, _P, _U64, _P]),
    "ganffn_encoder_bwd": (_I, [_PE, _I, _I, _P, _P, _P, _P, _P, _P, _U64, _P]),
    "ganffn_encoder_bwd2": (_I, [_PE, _I, _I, _P, _P, _P, _P, _P, _P, _U64, _I, _P]),
    "ganffn_head_fwd": (_I, [_PH, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U64, _P]),
    "ganffn_head_bwd": (_I, [_PH, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U64, _P]),
    "ganffn_linear_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ganffn_linear_bwd_workspace_floats": (_L, [_I, _I, _I]),
    "ganffn_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P]),
    "ganffn_bce_fwd": (_I, [_P, _F, _I, _F, _P, _I, _P]),
    "ganffn_bce_bwd": (_I, [_P, _F, _I, _F, _P, _P]),
    "ganffn_bce2_fwd": (_I, [_P, _F, _F, _I, _I, _I, _F, _P, _I, _P]),
    "ganffn_bce2_bwd": (_I, [_P, _F, _F, _I, _I, _I, _F, _P, _P]),
    "ganffn_adam_step": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P]),
    "ganffn_lstm_saved_floats": (_L, [C.POINTER(LstmCfg)]),
    "ganffn_lstm_workspace_floats": (_L, [C.POINTER(LstmCfg)]),
    "ganffn_lstm_layer_fwd": (_I, [C.POINTER(LstmCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ganffn_lstm_layer_bwd": (_I, [C.POINTER(LstmCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ganffn_adam_step_parts": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P, _L, _I, _L, _L, _L, _P]),
    "ganffn_encoder_bwd_parts_supported": (_I, [_PE]),
    "ganffn_encoder_bwd_parts_covered": (_L, [_I, _I]),
    "ganffn_encoder_bwd_parts": (_I, [_PE, _P, _P, _P, _P, _P, _P, _U64, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int), _P]),
    "ganffn_adam_update": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P]),
    "ganffn_adam_bump": (_I, [_P, _P]),
    "ganffn_add3": (_I, [_P, _P, _P, _P, _L, _P]),
    "ganffn_logsoftmax_nll": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ganffn_gemm_nt": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ganffn_gemm_nn": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ganffn_gemm_tn_acc": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ganffn_ffn_linear1_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _U32, _P, _U64, _I, _P]),
    "ganffn_gemm_tn_grouped_workspace_floats": (_L, []),
    "ganffn_gemm_tn_grouped": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "ganffn_attention_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _U32, _P, _U64, _P]),
    "ganffn_attention_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _U32, _P, _U64, _P]),
    "ganffn_gemm_hook": (_I, [_I, _I, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _U32, _P, _U64, _I, _I, _P, _P]),
    "ganffn_ffn_k100_hook": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _F, _U32, _P, _U64, _I, _P]),
    "ganffn_attention_keep_words": (_L, [_I, _I]),
    "ganffn_attention_fwd_keep": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _U32, _P, _U64, _P]),
    "ganffn_attention_bwd_keep": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _U32, _P, _U64, _P]),
    "ganffn_add_dropout_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _U32, _P, _U64, _P]),
    "ganffn_add_dropout_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U32, _P, _U64, _P]),
    "ganffn_general2_attention_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ganffn_general2_attention_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ganffn_gemm_n100": (_I, [_P, _P, _I, _P, _P, _L, _I, _I, _I, C.POINTER(C.c_int), _P]),
    "ganffn_debug_set_ffn_mode": (_I, [_I]),
    "ganffn_drnn_skinny": (_I, [_I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "ganffn_drnn_saved_floats": (_L, [C.POINTER(DrnnCfg)]),
    "ganffn_drnn_workspace_floats": (_L, [C.POINTER(DrnnCfg)]),
    "ganffn_drnn_fwd": (_I, [C.POINTER(DrnnCfg), _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U64, _P]),
    "ganffn_drnn_bwd": (_I, [C.POINTER(DrnnCfg), _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U64, _P]),
    "ganffn_dropout": (_I, [_P, _P, _I, _I, _F, _U32, _P, _U64, _P]),
    "ganffn_seq_reverse": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ganffn_drnn_join_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _U32, _U32, _P, _U64, _I, _P]),
    "ganffn_drnn_join_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _U32, _U32, _P, _U64, _I, _P]),
    "ganffn_mask_pos_inplace": (_I, [_P, _P, _F, _L, _P]),
}

_lib = None


class GanffnError(RuntimeError):
    pass


def load():
    """dlopen libganffn.so and bind every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GanffnError(
            "libganffn.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C gan_ffn_amd/csrc`. There is no CPU fallback for the GAN-FFN hot path." % LIB_PATH)
    # torch first: its wheel carries its own HIP runtime, and device memory and streams come from torch.  Loaded in the other
    # order, libganffn.so binds /opt/rocm's libamdhip64 and the process ends up with two runtimes — every launch of this
    # library then fails with "no ROCm-capable device is detected" (seen with build() + smoke() in one process).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("GANFFN_LIB") and not hasattr(lib, name):
            continue             # a lab / older build named through GANFFN_LIB may lack newer entry points (A/B measurement only)
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ganffn_last_error()
        raise GanffnError("%s failed (rc=%d): %s" % (what or "libganffn call", rc, msg.decode() if msg else "?"))


def call(name, *args):
    """call an int-returning entry point and raise on a non-zero code"""
    check(getattr(load(), name)(*args), name)
