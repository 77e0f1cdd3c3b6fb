"""ctypes binding of libxvec_hip.so (include/xvec_hip.h).  Plumbing only: every function
takes plain pointers/ints; torch is imported first so that the HIP runtime the library
binds to is the one torch already loaded (same SONAME libamdhip64.so.7), which is what
makes torch tensors' data_ptr() and stream handles valid inside the library.

There is NO fallback: if the shared library is missing or a symbol is absent, loading
raises and the product path is unusable (build with `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxvec_hip.so")

XV_MAX_ATT_LAYERS = 4
XV_OK = 0
XV_ERR_TOO_SHORT = -7
XV_PREC_F32 = 0
XV_PREC_BF16X3 = 1
XV_PREC_F16X3 = 2
XV_PREC_F16F6 = 3
XV_POOL_STATISTICS = 0
XV_POOL_SELF_ATTENTION = 1
XV_ACT_RELU, XV_ACT_LRELU, XV_ACT_PRELU = 0, 1, 2

EXPORTS = ["xv_version", "xv_create", "xv_set_tensor", "xv_finalize", "xv_set_option", "xv_check_overflow", "xv_flags_async", "xv_flags_decode", "xv_node_id", "xv_node_context",
           "xv_plan_create", "xv_plan_query", "xv_plan_destroy", "xv_forward", "xv_profile_begin", "xv_profile_end",
           "xv_destroy", "xv_last_error",
           "xv_frontend_cmn_select", "xv_length_normalize", "xv_speaker_mean",
           "xv_ark_open", "xv_ark_open_scp", "xv_ark_scp_count", "xv_ark_scp_shapes", "xv_ark_next_batch", "xv_ark_pending_shape", "xv_ark_skipped", "xv_ark_set_copy_threads", "xv_ark_error", "xv_ark_close", "xv_ark_format_vectors", "xv_crc32c", "xv_pack_rows"]


class ModelDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("network_type", C.c_int32), ("feat_dim", C.c_int32),
        ("channels", C.c_int32), ("pooling_type", C.c_int32), ("relu_type", C.c_int32),
        ("num_nodes_pooling_layer", C.c_int32), ("num_nodes_last_layer", C.c_int32),
        ("last_layer_no_bn", C.c_int32), ("last_layer_linear", C.c_int32),
        ("feature_norm", C.c_int32), ("feature_scaling_factor", C.c_float),
        ("att_key_input", C.c_int32), ("att_value_input", C.c_int32),
        ("att_num_key_layers", C.c_int32), ("att_key_num_nodes", C.c_int32 * XV_MAX_ATT_LAYERS),
        ("att_key_network_type", C.c_int32),
        ("att_num_value_layers", C.c_int32), ("att_value_num_nodes", C.c_int32 * XV_MAX_ATT_LAYERS),
        ("att_value_network_type", C.c_int32),
        ("att_apply_nonlinear", C.c_int32), ("att_use_scale", C.c_int32), ("att_num_heads", C.c_int32),
        ("att_split_value", C.c_int32), ("att_split_key", C.c_int32), ("precision", C.c_int32),
        ("resnet_blocks", C.c_int32 * 4), ("resnet_maxpooling", C.c_int32), ("resnet_time_stride", C.c_int32),
    ]


class PlanInfo(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("node_id", C.c_int32), ("batch", C.c_int32), ("frame_level", C.c_int32),
        ("in_frames", C.c_int64), ("out_rows", C.c_int64), ("out_cols", C.c_int64),
        ("workspace_bytes", C.c_int64), ("flops", C.c_int64),
    ]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("launches", C.c_int32),
                ("flops", C.c_int64), ("bytes", C.c_int64)]


_lib = None


def load():
    """Load libxvec_hip.so once and declare the signatures.  Raises on any problem."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "HIP extension %s is missing: the x-vector path has no CPU fallback. "
            "Build it with __graft_entry__.build() (hipcc --offload-arch=gfx950)." % LIB_PATH)
    import torch  # noqa: F401  (loads torch's libamdhip64 first; see module docstring)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    missing = [n for n in EXPORTS if not hasattr(lib, n)]
    if missing:
        raise RuntimeError("libxvec_hip.so lacks symbols: %s" % ", ".join(missing))
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.xv_version.restype = C.c_char_p
    lib.xv_version.argtypes = []
    lib.xv_last_error.restype = C.c_char_p
    lib.xv_last_error.argtypes = [vp]
    lib.xv_create.argtypes = [C.POINTER(ModelDesc), i32, C.POINTER(vp)]
    lib.xv_set_tensor.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32]
    lib.xv_finalize.argtypes = [vp]
    lib.xv_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.xv_check_overflow.argtypes = [vp, i32]
    lib.xv_flags_async.argtypes = [vp, vp, vp]
    lib.xv_flags_decode.argtypes = [vp]
    lib.xv_node_id.argtypes = [vp, C.c_char_p]
    lib.xv_node_context.argtypes = [vp, i32]
    lib.xv_plan_create.argtypes = [vp, vp, i32, i32, vp, C.POINTER(vp)]
    lib.xv_plan_query.argtypes = [vp, C.POINTER(PlanInfo)]
    lib.xv_plan_destroy.argtypes = [vp]
    lib.xv_plan_destroy.restype = None
    lib.xv_forward.argtypes = [vp, vp, vp, i32, vp, i64, vp, i64, vp]
    lib.xv_profile_begin.argtypes = [vp, i32]
    lib.xv_profile_end.argtypes = [vp, C.POINTER(KernelTime), i32, C.POINTER(i32)]
    lib.xv_destroy.argtypes = [vp]
    lib.xv_destroy.restype = None
    lib.xv_frontend_cmn_select.argtypes = [i32, vp, i32, i32, vp, i32, vp, i64, i32, i32, i32, vp, vp, vp]
    lib.xv_length_normalize.argtypes = [i32, vp, i64, i64, i32, i32, vp, i64, vp]
    lib.xv_speaker_mean.argtypes = [i32, vp, i64, i32, vp, vp, i64, vp, i64, vp]
    lib.xv_ark_open.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    lib.xv_ark_open_scp.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.xv_ark_scp_count.argtypes = [vp]
    lib.xv_ark_scp_count.restype = i64
    lib.xv_ark_scp_shapes.argtypes = [vp, vp, vp, i64]
    lib.xv_ark_next_batch.argtypes = [vp, i64, i32, i32, vp, i64, vp, vp, i64, C.POINTER(i32), C.POINTER(i32)]
    lib.xv_ark_pending_shape.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.xv_ark_skipped.argtypes = [vp]
    lib.xv_ark_set_copy_threads.argtypes = [vp, i32]
    lib.xv_ark_skipped.restype = i64
    lib.xv_ark_error.argtypes = [vp]
    lib.xv_ark_error.restype = C.c_char_p
    lib.xv_ark_close.argtypes = [vp]
    lib.xv_ark_close.restype = None
    lib.xv_ark_format_vectors.argtypes = [vp, i32, vp, i32, i64, vp, i64]
    lib.xv_ark_format_vectors.restype = i64
    lib.xv_crc32c.argtypes = [C.c_uint32, vp, i64]
    lib.xv_pack_rows.argtypes = [vp, vp, i32, vp, i32]
    lib.xv_pack_rows.restype = i64
    lib.xv_crc32c.restype = C.c_uint32
    for n in EXPORTS:
        if n not in ("xv_version", "xv_last_error", "xv_plan_destroy", "xv_destroy", "xv_ark_skipped", "xv_ark_error",
                     "xv_ark_close", "xv_ark_format_vectors", "xv_ark_scp_count", "xv_crc32c", "xv_pack_rows"):
            getattr(lib, n).restype = i32
    _lib = lib
    return lib


class XvError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "xvec_hip error %d: %s" % (code, msg))
        self.code = code


def check(rc, handle=None):
    if rc < 0:
        msg = load().xv_last_error(handle)
        raise XvError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
