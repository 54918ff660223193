"""The per-pixel glue between the renderer and CLIP as two fused HIP kernels each way (csrc/avc_glue.hip; main.py:426-534):
`ShadeLossFn` = random-light Lambert shading + scatter of the silhouette rays into full images + the per-pixel terms of the colour L1
and mask BCE losses; `ResizeNormFn` = CLIP's preprocessing of the images.  Runner.shade_and_scatter / assemble_loss remain the readable
torch statement of the same lines (pinned against the reference's own lines by tests/test_glue_golden.py); tests/test_gpu_glue.py
holds the fused path to them, values and gradients.  No CPU fallback: the callers use the torch statement off the GPU."""
import ctypes

import numpy as np
import torch

from . import lib as L
from .clip_vit import CLIP_MEAN, CLIP_STD

_MEAN = (ctypes.c_float * 3)(*CLIP_MEAN)
_STD = (ctypes.c_float * 3)(*CLIP_STD)


class ShadeLossFn(torch.autograd.Function):
    """(color [R,3], extra [R,3], wsum [R], nsum [R,3] | None) -> images [2,P,3] (texture_shading | extra, rand_shading_rgb),
    sum |color - true| mask, sum mask, sum BCE terms, sum (color - true)^2 mask (for the logged psnr; not differentiable)"""

    @staticmethod
    def forward(ctx, color, extra, wsum, nsum, true_rgb, mask, ray_of_pixel, bg, bg_const, light, img0_is_extra):
        lib = L.load()
        P = mask.numel()
        f32 = lambda t: None if t is None else t.contiguous().float()
        color, extra, wsum, nsum, true_rgb, mask, bg = (f32(t) for t in (color, extra, wsum, nsum, true_rgb, mask, bg))
        images = torch.empty(2, P, 3, device=color.device, dtype=torch.float32)
        partial = torch.empty(lib.avc_shade_loss_blocks(P), 4, device=color.device, dtype=torch.float32)
        L.check(lib.avc_shade_loss_fwd(L.ptr(color), L.ptr(extra), L.ptr(wsum), L.ptr(nsum), L.ptr(true_rgb), L.ptr(mask), L.ptr(ray_of_pixel),
                                       L.ptr(bg), float(bg_const), L.ptr(light), P, int(img0_is_extra), L.ptr(images), L.ptr(partial), L.stream()),
                "avc_shade_loss_fwd")
        l1, ms, bce, sq = partial.sum(0).unbind(0)
        ctx.save_for_backward(color, extra, wsum, nsum if nsum is not None else color.new_zeros(0), true_rgb, mask,
                              ray_of_pixel if ray_of_pixel is not None else color.new_zeros(0), light if light is not None else color.new_zeros(0))
        ctx.flags = (nsum is not None, ray_of_pixel is not None, light is not None, int(img0_is_extra), P)
        ctx.mark_non_differentiable(ms, sq)
        return images, l1, ms, bce, sq

    @staticmethod
    def backward(ctx, dimages, dl1, dms, dbce, dsq):
        color, extra, wsum, nsum, true_rgb, mask, rop, light = ctx.saved_tensors
        has_n, has_rop, has_light, img0_is_extra, P = ctx.flags
        dev = color.device
        z = lambda: torch.zeros((), device=dev, dtype=torch.float32)
        gs = torch.stack([dl1.float() if dl1 is not None else z(), dbce.float() if dbce is not None else z()]).contiguous()
        dimages = dimages.contiguous().float() if dimages is not None else torch.zeros(2, P, 3, device=dev)
        dcolor, dextra, dwsum = torch.zeros_like(color), torch.zeros_like(extra), torch.zeros_like(wsum)
        dnsum = torch.zeros_like(nsum) if has_n else None
        L.check(L.load().avc_shade_loss_bwd(L.ptr(color), L.ptr(extra), L.ptr(wsum), L.ptr(nsum) if has_n else None, L.ptr(true_rgb), L.ptr(mask),
                                            L.ptr(rop) if has_rop else None, L.ptr(light) if has_light else None, P, img0_is_extra,
                                            dimages[0].data_ptr(), dimages[1].data_ptr(), L.ptr(gs), L.ptr(dcolor), L.ptr(dextra), L.ptr(dwsum),
                                            L.ptr(dnsum) if has_n else None, L.stream()), "avc_shade_loss_bwd")
        return dcolor, dextra, dwsum, dnsum, None, None, None, None, None, None, None


class ResizeNormFn(torch.autograd.Function):
    """images [B,H,W,3] in [0,1] -> [B,3,224,224] CLIP input (bilinear, align_corners=False; Normalize)"""

    @staticmethod
    def forward(ctx, images):
        images = images.contiguous().float()
        B, H, W, _ = images.shape
        out = torch.empty(B, 3, 224, 224, device=images.device, dtype=torch.float32)
        L.check(L.load().avc_resize_norm_fwd(L.ptr(images), B, H, W, _MEAN, _STD, L.ptr(out), L.stream()), "avc_resize_norm_fwd")
        ctx.shape = (B, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, H, W = ctx.shape
        dout = dout.contiguous().float()
        dimg = torch.empty(B, H, W, 3, device=dout.device, dtype=torch.float32)
        L.check(L.load().avc_resize_norm_bwd(L.ptr(dout), B, H, W, _MEAN, _STD, L.ptr(dimg), L.stream()), "avc_resize_norm_bwd")
        return dimg


def unit_light(light_dir, ambience):
    """host side of main.py:434-441: the light direction normalised like `rand_light_d / (norm + 1e-7)` in float32, + the ambience"""
    l = np.asarray(light_dir, np.float32)
    l = l / (np.float32(np.sqrt((l * l).sum(dtype=np.float32))) + np.float32(1e-7))
    return np.array([l[0], l[1], l[2], ambience], np.float32)
