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


_tickets = {}


def _ticket(device):
    """the zero-initialised word avc_shade_loss_fwd counts its finished blocks in (one per device and stream; the kernel resets it)"""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _tickets.get(key)
    if t is None:
        t = _tickets[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


class ShadeLossFn(torch.autograd.Function):
    """(color [R,3], extra [R,3], wsum [R], nsum [R,3] | None) -> images [2,P,3] (texture_shading | extra, rand_shading_rgb) and
    sums [4] = sum |color - true| mask, sum mask, sum BCE terms, sum (color - true)^2 mask (the last for the logged psnr; the gradient
    flows through [0] and [2])"""

    @staticmethod
    def forward(ctx, color, extra, wsum, nsum, true_rgb, mask, ray_of_pixel, bg, bg_const, light, img0_is_extra):
        lib = L.load()
        P = mask.numel()
        f32 = lambda t: None if t is None else t.contiguous().float()
        color, extra, wsum, nsum, true_rgb, mask, bg = (f32(t) for t in (color, extra, wsum, nsum, true_rgb, mask, bg))
        dev = color.device
        images = torch.empty(2, P, 3, device=dev, dtype=torch.float32)
        partial = torch.empty(lib.avc_shade_loss_blocks(P), 4, device=dev, dtype=torch.float32)
        sums = torch.empty(4, device=dev, dtype=torch.float32)
        L.check(lib.avc_shade_loss_fwd(L.ptr(color), L.ptr(extra), L.ptr(wsum), L.ptr(nsum), L.ptr(true_rgb), L.ptr(mask), L.ptr(ray_of_pixel),
                                       L.ptr(bg), float(bg_const), L.ptr(light), P, int(img0_is_extra), L.ptr(images), L.ptr(partial), L.ptr(sums),
                                       L.ptr(_ticket(dev)), L.stream()), "avc_shade_loss_fwd")
        ctx.save_for_backward(color, extra, wsum, nsum if nsum is not None else color.new_zeros(0), true_rgb, mask,
                              ray_of_pixel if ray_of_pixel is not None else color.new_zeros(0), light if light is not None else color.new_zeros(0))
        ctx.flags = (nsum is not None, ray_of_pixel is not None, light is not None, int(img0_is_extra), P)
        return images, sums

    @staticmethod
    def backward(ctx, dimages, dsums):
        color, extra, wsum, nsum, true_rgb, mask, rop, light = ctx.saved_tensors
        has_n, has_rop, has_light, img0_is_extra, P = ctx.flags
        dev = color.device
        gs = dsums.contiguous().float() if dsums is not None else torch.zeros(4, device=dev)
        dimages = dimages.contiguous().float() if dimages is not None else torch.zeros(2, P, 3, device=dev)
        R = wsum.numel()
        z = torch.zeros(R * (10 if has_n else 7), device=dev, dtype=torch.float32)      # (rays that own no pixel keep a zero gradient)
        dcolor, dextra, dwsum = z[:3 * R].view_as(color), z[3 * R:6 * R].view_as(extra), z[6 * R:7 * R].view_as(wsum)
        dnsum = z[7 * R:].view_as(nsum) if has_n else None
        L.check(L.load().avc_shade_loss_bwd(L.ptr(color), L.ptr(extra), L.ptr(wsum), L.ptr(nsum) if has_n else None, L.ptr(true_rgb), L.ptr(mask),
                                            L.ptr(rop) if has_rop else None, L.ptr(light) if has_light else None, P, img0_is_extra,
                                            dimages[0].data_ptr(), dimages[1].data_ptr(), L.ptr(gs), L.ptr(dcolor), L.ptr(dextra), L.ptr(dwsum),
                                            L.ptr(dnsum) if has_n else None, L.stream()), "avc_shade_loss_bwd")
        return dcolor, dextra, dwsum, dnsum, None, None, None, None, None, None, None


class LossTailFn(torch.autograd.Function):
    """(enc [B,512], text [T,512], sums [4] of ShadeLossFn, eikonal scalar) -> loss (scalar), stats [8] = loss, colour loss, mask loss,
    cos_0, cos_1: main.py:491-534 from the embeddings on -- the two cosine similarities, the normalisations of the colour / mask sums
    and the weighted sum -- in one launch each way (csrc/avc_glue.hip loss_tail_*; ~35 + ~40 torch launches otherwise)."""

    @staticmethod
    def forward(ctx, enc, text, sums, eik, igr_w, mask_w, clip_w, P):
        enc, text, sums = enc.contiguous().float(), text.contiguous().float(), sums.contiguous().float()
        e1 = eik.float().reshape(1)
        B, D = enc.shape
        dev = enc.device
        loss = torch.empty((), device=dev, dtype=torch.float32)
        out = torch.empty(8, device=dev, dtype=torch.float32)
        saved = torch.empty(4 * B + 1, device=dev, dtype=torch.float32)
        w = (float(igr_w), float(mask_w), float(clip_w), float(P))
        L.check(L.load().avc_loss_tail_fwd(L.ptr(enc), L.ptr(text), B, text.shape[0], D, L.ptr(sums), L.ptr(e1), *w, L.ptr(loss), L.ptr(out),
                                           L.ptr(saved), L.stream()), "avc_loss_tail_fwd")
        ctx.save_for_backward(enc, text, sums, saved)
        ctx.w, ctx.eik_shape = w, eik.shape
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, gloss, gout):
        enc, text, sums, saved = ctx.saved_tensors
        B, D = enc.shape
        g = gloss.contiguous().float()
        d_enc = torch.empty_like(enc)
        dd = torch.empty(8, device=enc.device, dtype=torch.float32)
        L.check(L.load().avc_loss_tail_bwd(L.ptr(g), L.ptr(enc), L.ptr(text), B, text.shape[0], D, L.ptr(sums), L.ptr(saved), *ctx.w, L.ptr(d_enc),
                                           L.ptr(dd), L.stream()), "avc_loss_tail_bwd")
        return d_enc, None, dd[:4], dd[4].reshape(ctx.eik_shape), None, None, None, None


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
