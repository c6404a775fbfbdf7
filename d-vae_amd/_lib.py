"""ctypes binding of libdvae_hip.so (C-ABI declared in include/dvae_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared
object is missing, or a compute entry point is called without a HIP device, the
call fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DVAE_EXP_LIB=<path>: load an alternative build of the library (A/B and ablation runs, tools/*_ablate.sh); diagnostics only
LIB_PATH = os.path.abspath(os.environ["DVAE_EXP_LIB"]) if os.environ.get("DVAE_EXP_LIB") else os.path.join(HERE, "csrc", "libdvae_hip.so")
NUM_PARAMS = 36


class ModelDesc(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("max_n", "nvt", "hs", "nz", "bidir", "vid", "is_bn", "start_type", "end_type", "mfma_bf16")]


class BatchCounts(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("B", "n", "T", "is_bn", "nB", "E", "PT", "PL", "NA", "NB", "NP", "CA", "CH", "CC",
                 "R", "RL", "NCE", "HGF", "HGB", "level_major")] + [("total_ints", C.c_int64)]


class BatchCaps(C.Structure):
    """dvae_batch_caps (include/dvae_hip.h): the edge-dependent counts a shape-stable export is padded to"""
    _fields_ = [(k, C.c_int32) for k in ("E", "PT", "PL", "NB", "CH")]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("a_rows", C.c_void_p), ("b_rows", C.c_void_p), ("c_rows", C.c_void_p),
                ("bias", C.c_void_p),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("a_kmajor", C.c_int32), ("b_kmajor", C.c_int32),
                ("act", C.c_int32), ("accumulate", C.c_int32), ("relu_mask", C.c_void_p),
                ("b_rows_bound", C.c_int32), ("a_bf16", C.c_int32), ("b_bf16", C.c_int32), ("c_bf16", C.c_int32)]


class DvaeHipError(RuntimeError):
    pass


class LevelDesc(C.Structure):
    """dvae_level_desc (include/dvae_hip.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("a", "w", "gi_tab", "types", "bhn", "gates", "ghn", "hout", "gv_row", "gm",
                                          "msg", "next_agg", "next_off", "idx", "dx", "hprev", "dgh", "dgin", "dhprev",
                                          "gather_src", "gather_src2", "off2", "idx2", "dgm_next")] + \
               [(n, C.c_int32) for n in ("lda", "K", "ld_hout", "ld_msg", "cur_row0", "ld_agg", "ld_dx", "ld_hprev",
                                         "ld_dhprev", "ldw_bf16")] + [("w_bf16", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float
_MD = C.POINTER(ModelDesc)
_BC = C.POINTER(BatchCounts)

# name -> (restype, argtypes); every symbol include/dvae_hip.h declares
SIGNATURES = {
    "dvae_last_error": (C.c_char_p, []),
    "dvae_version": (_I, []),
    "dvae_gemm_f32": (_I, [C.POINTER(GemmDesc), _P]),
    "dvae_gemm_bf16": (_I, [C.POINTER(GemmDesc), _P]),
    "dvae_gemm_skinny_f32": (_I, [C.POINTER(GemmDesc), _I, _P]),
    "dvae_debug_set_gemm_skinny_mmac": (_I, [_I]),
    "dvae_f32_to_bf16": (_I, [_P, _P, C.c_int64, _P]),
    "dvae_debug_set_kmajor_span_limit": (_I, [C.c_int64]),
    "dvae_gather_sum_f32": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P]),
    "dvae_scatter_rows_add_f32": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _P]),
    "dvae_gru_pointwise_fwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _P]),
    "dvae_gru_pointwise_bwd": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "dvae_gated_pointwise_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dvae_gated_pointwise_bwd": (_I, [_P, _I, _P, _P, _I, _I, _P]),
    "dvae_gated_fwd_gather": (_I, [_P, _P, _P, _I, _I, _P, _I, _P, _P, _I, _I, _P]),
    "dvae_gated_bwd_gather": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dvae_gru_level_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dvae_gru_level_bwd": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "dvae_level_dgrad_acc": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _P]),
    "dvae_chain_level_batch": (_I, [_I, C.POINTER(LevelDesc), _I, _I, _I, _P]),
    "dvae_gated_level_fwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _I, _I, _P]),
    "dvae_type_csr": (_I, [_P, _I, _P, _I, _I, _P, _P, _P]),
    "dvae_type_colsum": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "dvae_segment_colsum_f32": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "dvae_pair_mask_words": (C.c_int64, [_I, _I]),
    "dvae_pair_bce_fwd": (_I, [_P] * 14 + [_I, _I, _I, _P]),
    "dvae_pair_bce_bwd_grouped": (_I, [_P] * 8 + [_I, _I, _P]),
    "dvae_vertex_ce": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dvae_packed_floats": (C.c_int64, [_MD]),
    "dvae_pack_weights": (_I, [_MD, C.POINTER(_P), _P, _P]),
    "dvae_unpack_grads": (_I, [_MD, _P, _P, C.POINTER(_P), C.POINTER(C.c_uint8), _I, _P]),
    "dvae_adam_step_f32": (_I, [_P, _P, _P, _P, C.c_int64, _F, _F, _F, _F, _I, _I, _P]),
    "dvae_adam_step_dev_f32": (_I, [_P, _P, _P, _P, C.c_int64, _P, _F, _F, _F, _P, _I, _P]),
    "dvae_repack_weights": (_I, [_MD, C.POINTER(_P), _P, _P]),
    "dvae_batch_create": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, C.POINTER(_P), _BC]),
    "dvae_batch_export": (_I, [_P, _P, C.c_int64]),
    "dvae_batch_export_padded": (_I, [_P, C.POINTER(BatchCaps), _P, C.c_int64, _BC]),
    "dvae_batch_destroy": (None, [_P]),
    "dvae_decode_ws_floats": (C.c_int64, [_MD, _I]),
    "dvae_decode": (_I, [_MD, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "dvae_decode_static_max_floats": (C.c_int64, []),
    "dvae_decode_static": (_I, [_MD, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "dvae_encode_ws_floats": (C.c_int64, [_MD, _BC]),
    "dvae_loss_ws_floats": (C.c_int64, [_MD, _BC]),
    "dvae_debug_side_stream": (_I, [_P, C.POINTER(_P)]),
    "dvae_debug_launch_count": (C.c_int64, []),
    "dvae_encode_forward": (_I, [_MD, _BC, _P, _P, _P, _P, _P, _P]),
    "dvae_encode_backward": (_I, [_MD, _BC, _P, _P, _P, _P, _P, _P, _P]),
    "dvae_loss_forward": (_I, [_MD, _BC, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P, _P]),
    "dvae_loss_backward": (_I, [_MD, _BC, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load libdvae_hip.so (once).  Raises DvaeHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DvaeHipError(
            "HIP library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C d-vae_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise DvaeHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dvae_last_error()
        raise DvaeHipError("%s failed (%d): %s" % (what or "dvae call", rc, msg.decode() if msg else "?"))


def ptr(t):
    """device/host pointer of a torch tensor or numpy array, or None"""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
