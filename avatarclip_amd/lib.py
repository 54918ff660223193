"""ctypes binding of libavc.so (the C ABI declared in include/avc.h).  No fallback: if the library is missing
or a call fails, an exception is raised."""
import ctypes
import os

import torch

from . import build as _build

_lib = None
ABI_VERSION = 2          # AVC_ABI_VERSION of include/avc.h this binding was written against

c_int, c_long, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p
P = c_void_p

_SIGS = {
    "avc_version": (c_int, []),
    "avc_num_offsets": (c_int, []),
    "avc_sdf_forward": (c_int, [c_int, P, P, P, P, c_int, c_int, c_long, P, P, P, P, P, c_int, P]),
    "avc_upsample_step": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, P, P, P, P, P]),
    "avc_upsample_step_lanes": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, P, P, P, P, c_int, P]),
    "avc_render_points_fwd": (c_int, [c_int, P, P, P, P, c_int, c_int, c_float, c_long, P, P, P, P, P, P, c_long, P, P]),
    "avc_fwd_scratch_bytes_per_wave": (c_long, [c_int]),
    "avc_composite_fwd": (c_int, [P, P, P, P, P, P, c_int, c_int, P, c_float, c_float, P, c_int, P, P, P, P, P, P, P, P, P, P]),
    "avc_composite_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, P, c_float, c_float, P, c_int, P, P, P, P, P, P, P,
                                  P, P, P, P, P]),
    "avc_render_points_fwd_train": (c_int, [c_int, P, P, P, P, c_int, c_int, c_float, c_long, P, P, P, P, P, P, c_long, P, P, P]),
    "avc_fwd_panel_tiles": (c_int, [c_int]),
    "avc_grad_panel_tiles": (c_int, [c_int]),
    "avc_mask_u16_per_block": (c_int, [c_int]),
    "avc_render_points_bwd": (c_int, [c_int, P, P, P, P, c_int, c_int, c_float, c_long, P, P, P, P, P, P, P, P, P, P, P, c_long, P]),
    "avc_bwd_colsum_floats": (c_int, [c_int]),
    "avc_bwd_colsum_rows": (c_long, [c_long, c_long]),
    "avc_mc_classify": (c_int, [P, c_int, c_int, c_int, c_float, P, P, P, P]),
    "avc_mc_emit": (c_int, [P, c_int, c_int, c_int, c_float, P, P, P, P, P, P, P, P, P]),
    "avc_text_attention_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "avc_vit_linear": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "avc_vit_linear_bwd_gelu": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P]),
    "avc_vit_workspace_bytes": (c_long, [c_int, c_int]),
    "avc_vit_ln_pack": (c_int, [P, P, P, c_float, c_int, c_int, P, P]),
    "avc_vit_linear_packed": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "avc_vit_pack": (c_int, [P, P, P, c_int, c_int, P]),
    "avc_vit_attention_bwd_packed": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "avc_vit_linear_small": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "avc_vit_ln_bwd": (c_int, [P, P, P, c_float, P, P, P, c_int, c_int, P]),
    "avc_vit_attention_fwd_packed": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "avc_vit_attention_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "avc_vit_attention_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "avc_probe_mfma": (c_int, [P, P, P, P, P, P, P]),
    "avc_rasterize_faces": (c_int, [P, P, c_int, c_int, c_float, c_float, P, P, P]),
    "avc_rasterize_scratch_bytes": (c_long, [c_int, c_int]),
    "avc_rasterize_mesh": (c_int, [P, c_int, P, c_int, P, c_float, P, c_int, c_float, c_float, P, P, c_int, c_int, P, P]),
    "avc_dense_params_fwd": (c_int, [c_int, P, P, P, P, P, P, P, P, P]),
    "avc_dense_params_bwd": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, P]),
    "avc_weight_grad_all": (c_int, [P, c_int, P, c_int, c_int, P, c_long, P, P, c_int, c_int, c_int, P]),
    "avc_weight_grad_reduce": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    "avc_weight_grad_unpack": (c_int, [P, P, P, P, c_int, P, P]),
    "avc_shade_loss_blocks": (c_int, [c_int]),
    "avc_shade_loss_fwd": (c_int, [P, P, P, P, P, P, P, P, c_float, P, c_int, c_int, P, P, P, P, P]),
    "avc_colsum": (c_int, [P, c_long, c_int, c_int, P, P, P]),
    "avc_colsum_scratch_bytes": (c_long, []),
    "avc_inv_s": (c_int, [P, P, P, P]),
    "avc_pack_params": (c_int, [P, c_int, P, P, c_int, P, P, c_int, P, P, P, P]),
    "avc_coarse_z": (c_int, [P, P, P, c_int, c_int, P, P]),
    "avc_loss_tail_fwd": (c_int, [P, P, c_int, c_int, c_int, P, P, c_float, c_float, c_float, c_float, P, P, P, P]),
    "avc_loss_tail_bwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P, c_float, c_float, c_float, c_float, P, P, P]),
    "avc_shade_loss_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, P, P, P, P, P, P, P, P]),
    "avc_resize_norm_fwd": (c_int, [P, c_int, c_int, c_int, P, P, P, P]),
    "avc_resize_norm_bwd": (c_int, [P, c_int, c_int, c_int, P, P, P, P]),
    "avc_gen_rays": (c_int, [P, P, P, c_int, c_int, c_float, c_float, c_float, c_int, c_int, c_int, P, P, P, P, P, P, P]),
    "avc_chess_background": (c_int, [P, c_int, c_int, c_int, P, P]),
}
_OPTIONAL = {}
# experimental entry points of libavc_ring.so (include/avc_ring.h): bound when the loaded library has them
_RING_SIGS = {
    "avc_bwd_ring_ctl_bytes": (c_long, []),
    "avc_bwd_ring_payload_bytes": (c_long, [c_int, c_int, c_int]),
    "avc_bwd_ring_types": (c_int, [c_int]),
    "avc_render_points_bwd_ring": (c_int, [c_int, P, P, P, P, c_int, c_int, c_float, c_long, P, P, P, P, P, P, P, P, P, P, P,
                                           P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
}


def lib_path():
    return _build.LIB


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("libavc.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "there is no CPU fallback for the HIP hot path")
    lib = ctypes.CDLL(path)
    lib.avc_last_error.restype = ctypes.c_char_p
    lib.avc_last_error.argtypes = []
    for name, (res, args) in list(_SIGS.items()) + list(_OPTIONAL.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _RING_SIGS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    got = lib.avc_version()
    if got != ABI_VERSION:
        raise RuntimeError("%s implements revision %d of include/avc.h, this binding expects %d: rebuild the library "
                           "(python -m avatarclip_amd.build --force)" % (path, got, ABI_VERSION))
    _lib = lib
    return lib


def has_ring():
    """the loaded library is libavc_ring.so (csrc/avc_bwd_ring.hip linked in)"""
    return hasattr(load(), "avc_render_points_bwd_ring")


def register_optional(sigs):
    _OPTIONAL.update(sigs)


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "HIP kernels need contiguous device tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libavc %s failed: %s" % (what, load().avc_last_error().decode()))
