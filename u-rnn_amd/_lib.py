"""ctypes binding of ``liburnn_hip.so`` (the C ABI in include/urnn_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing any op raises.
The product path never touches ``oracle/``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("URNN_LIB") or os.path.join(_HERE, "liburnn_hip.so")   # URNN_LIB: tuning variants

_lib = None

c_float_p = ctypes.c_void_p  # raw device pointers (tensor.data_ptr())
_i, _f, _p, _sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/urnn_hip.h exactly
SIGNATURES = {
    "urnn_abi_version": (_i, []),
    "urnn_last_error": (ctypes.c_char_p, []),
    "urnn_set_matrix_mode": (_i, [_i]),
    "urnn_get_matrix_mode": (_i, []),
    "urnn_max_abs_f32": (_i, [_p, ctypes.c_long, _p, _p]),
    "urnn_packed_conv_floats": (_sz, [_i, _i]),
    "urnn_pack_conv_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "urnn_packed_gru_floats": (_sz, [_i, _i, _i]),
    "urnn_pack_gru_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "urnn_packed_deconv_floats": (_sz, [_i, _i]),
    "urnn_pack_deconv_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "urnn_stage_conv_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "urnn_gru_cell_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "urnn_gru_cell_f32": (_i, [_p] * 10 + [_sz, _i, _i, _i, _i, _i, _f, _p]),
    "urnn_gru_cell_phases_f32": (_i, [_p] * 10 + [_sz, _i, _i, _i, _i, _i, _f, _i, _p]),
    "urnn_gru_cell_fused_reset_gate_applies": (_i, [_i] * 6),
    "urnn_gru_cell_coop_blocks": (_i, [_i] * 7),
    "urnn_gru_cell_tail_applies": (_i, [_i] * 6),
    "urnn_head_tail_partial_floats": (_sz, [_i, _i, _i]),
    "urnn_gru_cell_tail_f32": (_i, [_p] * 10 + [_sz, _i, _i, _i, _i, _i, _f, _i, _p, _i, _i, _f, _p, _p, _p, _p]),
    "urnn_head_coop_blocks_f32": (_i, [_i, _i, _i]),
    "urnn_head_coop_f32": (_i, [_p] * 13 + [_sz, _i, _i, _i, _i, _f, _f, _f, _p]),
    "urnn_head_after_tail_f32": (_i, [_p] * 13 + [_sz, _i, _i, _i, _i, _f, _f, _f, _p, _p]),
    "urnn_deconv2x2_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "urnn_head_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "urnn_head_f32": (_i, [_p] * 13 + [_sz, _i, _i, _i, _i, _f, _f, _f, _p]),
    "urnn_preprocess_f32": (_i, [_p] * 5 + [_f, _f, _p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _f, _p]),
    "urnn_stage1_static_f32": (_i, [_p, _p, _p, _f, _f, _p, _p, _i, _i, _i, _i, _i, _p]),
    "urnn_stage1_scalar_rain_f32": (_i, [_p] * 6 + [_i, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _p]),
    "urnn_gru_cell_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "urnn_gru_cell_backward_packed_floats": (_sz, [_i, _i, _i]),
    "urnn_gru_cell_backward_f32": (_i, [_p] * 25 + [_i, _p, _sz, _i, _i, _i, _i, _i, _i, _p]),
    "urnn_weight_gradient_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "urnn_weight_gradient_f32": (_i, [_p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _sz, _i, _i, _i, _i, _i, _p]),
    "urnn_stage_conv_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "urnn_stage_conv_backward_packed_floats": (_sz, [_i, _i]),
    "urnn_stage_conv_backward_f32": (_i, [_p] * 8 + [_i, _p, _sz, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "urnn_deconv2x2_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "urnn_deconv2x2_backward_packed_floats": (_sz, [_i, _i]),
    "urnn_deconv2x2_backward_f32": (_i, [_p] * 8 + [_i, _p, _sz, _i, _i, _i, _i, _i, _f, _i, _p]),
    "urnn_head_backward_workspace_bytes": (_sz, [_i, _i, _i]),
    "urnn_head_backward_f32": (_i, [_p] * 16 + [_sz, _i, _i, _i, _i, _f, _f, _i, _p]),
    "urnn_loss_workspace_bytes": (_sz, [ctypes.c_long]),
    "urnn_loss_f32": (_i, [_p, _p, _f, _p, _p, _p, _sz, ctypes.c_long, _p]),
    "urnn_adam_workspace_bytes": (_sz, [ctypes.c_long]),
    "urnn_adam_step_f32": (_i, [_p, _p, _p, _p, ctypes.c_long, _f, _f, _f, _f, _i, _p, _f, _p, _p, _sz, _p]),
    "urnn_advance_counter": (_i, [_p, _i, _p]),
    "urnn_head_rollout_f32": (_i, [_p] * 13 + [_sz, _i, _i, _i, _i, _f, _f, _f, _i, _p, _p, _p]),
    "urnn_preprocess_rollout_f32": (_i, [_p] * 5 + [_f, _f, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _p]),
    "urnn_stage1_scalar_rain_rollout_f32": (_i, [_p] * 6 + [_p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _p]),
    "urnn_gru_cell_strip_f32": (_i, [_p] * 10 + [_sz, _i, _i, _i, _i, _i, _f, _i, ctypes.c_long, _p]),
    "urnn_gru_cell_strip_stats_f32": (_i, [_p, _sz, _i, _i, _i, _i, _i, _i, _p, _p]),
    "urnn_head_strip_f32": (_i, [_p] * 13 + [_sz, _i, _i, _i, _i, _f, _f, _f, _i, ctypes.c_long, _p]),
    "urnn_head_strip_stats_f32": (_i, [_p, _sz, _i, _i, _i, _i, _i, _i, _p, _p]),
    "urnn_stage_conv_stem_applies": (_i, [_i, _i, _i, _i, _i]),
    "urnn_stage_conv_stem_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p]),
    "urnn_step_workspace_bytes": (_sz, [_p, _i, _i, _i]),
    "urnn_step_workspace_init": (_i, [_p, _p, _sz, _i, _i, _i, _p, _p, _p]),
    "urnn_step_f32": (_i, [_p] * 8 + [_sz, _i, _i, _i, _f, _f, _f, _p]),
}


class UrnnError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises loudly when the HIP extension is absent."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise UrnnError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for the U-RNN hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.urnn_abi_version() != 3:
            raise UrnnError("liburnn_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().urnn_last_error()
        raise UrnnError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
