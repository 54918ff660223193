"""CLIP ViT-B/32 image encoder on the gfx950 kernels -- drop-in for `perceptor.encode_image`
(AvatarGen/AppearanceGen/main.py:259-260,512,518,524; upstream: OpenAI clip/model.py VisionTransformer, un-vendored).

All GEMMs (patch embedding, in_proj, out_proj, c_fc, c_proj, visual.proj) run in avc_vit_linear (bf16 MFMA, fp32
accumulate; the reference runs CLIP in fp16 on GPU), attention in avc_vit_attention_*.  Weights are frozen
(`requires_grad_(False)`, main.py:260): the backward only propagates to the pixels, re-using the same GEMM kernel on
the packed transposes.  LayerNorm / residual bookkeeping are torch elementwise ops under autograd.
Accepts an OpenAI-format state dict (`visual.*` keys, e.g. torch.jit.load('ViT-B-32.pt').state_dict()).
"""
import math
import os
from typing import Dict

import torch
import torch.nn.functional as F

from . import lib as L
from . import h2d

WIDTH, LAYERS, HEADS, PATCH, RES, EMBED = 768, 12, 12, 32, 224, 512
TOKENS = (RES // PATCH) ** 2 + 1
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N,K] -> bf16 [N/32][K/16][64][8] in MFMA B-operand order (lane (n,h) holds W[32t+n][16s+8h+j])."""
    N, K = w.shape
    assert N % 32 == 0 and K % 16 == 0
    p = w.reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
    return p.to(torch.bfloat16).reshape(-1)


LIBRARY_GEMM = os.environ.get("AVC_VIT_LIBRARY_GEMM") == "1"   # see _linear_raw
PACKED_PIPELINE = os.environ.get("AVC_VIT_PACKED", "1") != "0"   # see ClipVisionB32._encode_image_batched
TRAIN_GRAPH = os.environ.get("AVC_CLIP_GRAPH", "1") != "0"        # see ClipVisionB32.encode_image
FUSED_BLOCKS = os.environ.get("AVC_CLIP_FUSED_BLOCKS", "1") != "0"   # see BlocksFn (0: one autograd node per kernel)


class _Lin:
    def __init__(self, w, b, dev):
        w = w.float().to(dev)
        self.N, self.K = w.shape
        self.wp = pack_weight(w)
        self.wtp = pack_weight(w.t().contiguous())
        self.wd = w.to(torch.bfloat16) if LIBRARY_GEMM else None   # dense copy for the library-GEMM comparison path only
        self.b = None if b is None else b.float().to(dev).contiguous()


_workspace = {}


def _ws(device, nbytes):
    """bf16 fragment copy of the activations of one avc_vit_linear call (stream-ordered reuse)"""
    key = str(device)
    if key not in _workspace or _workspace[key].numel() < nbytes:
        _workspace[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    return _workspace[key]


BIG_M = 192   # rows from which a call counts as "batched" (B >= 4 images): the kernel then takes groups of 4 row tiles per workgroup
# AVC_VIT_LIBRARY_GEMM=1 (LIBRARY_GEMM above): batched calls go to the library GEMM (torch.mm -> hipBLASLt) instead of the
# hand-written kernel -- the comparison point of the batched scoring path (ShapeGen codebook search, pose retrieval), not the default


def _gelu_grad(pre):
    s = torch.sigmoid(1.702 * pre)
    return s + 1.702 * pre * s * (1 - s)


def _linear_raw(x2d, lin, transposed, bias, residual, act, want_pre, gelu_pre=None):
    """y = x W^T (+ b) (QuickGELU) (+ residual) with bf16 operands and fp32 accumulate / output.  transposed: y = x W (backward);
    gelu_pre (backward only): x is multiplied by QuickGELU'(gelu_pre) first (folded into the packing pass of the kernel)."""
    lib = L.load()
    M = x2d.shape[0]
    N, K = (lin.K, lin.N) if transposed else (lin.N, lin.K)
    if LIBRARY_GEMM and M >= BIG_M:
        if gelu_pre is not None:
            x2d = x2d * _gelu_grad(gelu_pre)
        w = lin.wd if transposed else lin.wd.t()
        y = torch.mm(x2d.to(torch.bfloat16), w, out_dtype=torch.float32)
        if bias is not None:
            y += bias
        pre = None
        if act:
            pre = y.clone() if want_pre else None
            y = y * torch.sigmoid(1.702 * y)
        if residual is not None:
            y += residual
        return y, pre
    wp = lin.wtp if transposed else lin.wp
    y = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    pre = torch.empty_like(y) if (act and want_pre) else None
    ws = _ws(x2d.device, lib.avc_vit_workspace_bytes(M, K))
    if gelu_pre is not None:
        L.check(lib.avc_vit_linear_bwd_gelu(L.ptr(x2d), L.ptr(gelu_pre), L.ptr(wp), L.ptr(y), M, N, K, L.ptr(ws), L.stream()),
                "avc_vit_linear_bwd_gelu")
    else:
        L.check(lib.avc_vit_linear(L.ptr(x2d), L.ptr(wp), L.ptr(bias), L.ptr(residual), L.ptr(y), L.ptr(pre), M, N, K, act, L.ptr(ws),
                                   L.stream()), "avc_vit_linear")
    return y, pre


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lin, act, residual):
        shp = x.shape
        x2 = x.reshape(-1, lin.K).contiguous().float()
        r2 = None if residual is None else residual.reshape(-1, lin.N).contiguous().float()
        # the pre-activation of a QuickGELU layer is only kept for the backward pass (scoring calls run under no_grad: one 4-byte
        # store per output element less)
        y, pre = _linear_raw(x2, lin, False, lin.b, r2, act, bool(act) and ctx.needs_input_grad[0])
        ctx.lin, ctx.act, ctx.has_res = lin, act, residual is not None
        if pre is not None:          # (nothing to save for a linear without activation: a dummy tensor would cost a fill launch per call)
            ctx.save_for_backward(pre)
        return y.reshape(*shp[:-1], lin.N)

    @staticmethod
    def backward(ctx, dy):
        lin = ctx.lin
        shp = dy.shape
        d2 = dy.reshape(-1, lin.N).contiguous().float()
        pre = ctx.saved_tensors[0] if ctx.act else None      # QuickGELU'(pre) is applied inside the GEMM's packing pass
        dx, _ = _linear_raw(d2, lin, True, None, None, 0, False, gelu_pre=pre)
        dres = dy if ctx.has_res else None
        return dx.reshape(*shp[:-1], lin.K), None, None, dres


class AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv):
        lib = L.load()
        B, T, _ = qkv.shape
        qkv = qkv.contiguous().float()
        out = torch.empty(B, T, WIDTH, device=qkv.device, dtype=torch.float32)
        L.check(lib.avc_vit_attention_fwd(L.ptr(qkv), L.ptr(out), B, T, WIDTH, HEADS, L.stream()), "avc_vit_attention_fwd")
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        (qkv,) = ctx.saved_tensors
        B, T, _ = qkv.shape
        dout = dout.contiguous().float()
        dqkv = torch.empty_like(qkv)
        L.check(lib.avc_vit_attention_bwd(L.ptr(qkv), L.ptr(dout), L.ptr(dqkv), B, T, WIDTH, HEADS, L.stream()),
                "avc_vit_attention_bwd")
        return dqkv


class BlocksFn(torch.autograd.Function):
    """The 12 residual blocks (clip/model.py ResidualAttentionBlock) of a per-iteration call (M = B * 50 <= 128 rows) as ONE autograd
    node: activations go from kernel to kernel as packed bf16 operands, forward and backward -- LayerNorm writes the operand of the
    linear behind it, attention that of the out-projection, c_fc + QuickGELU that of c_proj; in the backward the proj^T product leaves
    already multiplied by QuickGELU' and packed, and the LayerNorm backward adds the residual branch's gradient and packs its result
    for the transposed linear below.  7 + 8 launches per block instead of 11 + ~14 (a packing launch per linear, torch LayerNorm
    kernels, AccumulateGrad adds).  Arithmetic as in LinearFn / AttentionFn / F.layer_norm: bf16 operands, fp32 accumulation, fp32
    residual stream and LayerNorm statistics."""

    @staticmethod
    def forward(ctx, x, model):
        lib, st = L.load(), L.stream()
        B = x.shape[0]
        M, W = B * TOKENS, WIDTH
        x0 = x.reshape(M, W).contiguous().float()
        nb = len(model.blocks)
        dev = x0.device
        keep = ctx.needs_input_grad[0]
        xs = torch.empty(nb, M, W, device=dev, dtype=torch.float32)            # block outputs (= the next block's input)
        x2s = torch.empty(nb if keep else 1, M, W, device=dev, dtype=torch.float32)
        qkvs = torch.empty(nb if keep else 1, M, 3 * W, device=dev, dtype=torch.float32)
        pres = torch.empty(nb, M, 4 * W, device=dev, dtype=torch.float32) if keep else None
        a_ln, a_at, a_fc = model._train_buf("ln", M, W), model._train_buf("attn", M, W), model._train_buf("fc", M, 4 * W)
        cur = x0
        for i, blk in enumerate(model.blocks):
            k = i if keep else 0
            x2, qkv, out = x2s[k], qkvs[k], xs[i]
            L.check(lib.avc_vit_ln_pack(L.ptr(cur), L.ptr(blk["ln1"][0]), L.ptr(blk["ln1"][1]), 1e-5, M, W, L.ptr(a_ln), st), "avc_vit_ln_pack")
            L.check(lib.avc_vit_linear_small(L.ptr(a_ln), L.ptr(blk["qkv"].wp), L.ptr(blk["qkv"].b), None, None, L.ptr(qkv), None, None,
                                             M, 3 * W, W, 0, st), "avc_vit_linear_small")
            L.check(lib.avc_vit_attention_fwd_packed(L.ptr(qkv), L.ptr(a_at), B, TOKENS, W, HEADS, st), "avc_vit_attention_fwd_packed")
            L.check(lib.avc_vit_linear_small(L.ptr(a_at), L.ptr(blk["out"].wp), L.ptr(blk["out"].b), L.ptr(cur), None, L.ptr(x2), None, None,
                                             M, W, W, 0, st), "avc_vit_linear_small")
            L.check(lib.avc_vit_ln_pack(L.ptr(x2), L.ptr(blk["ln2"][0]), L.ptr(blk["ln2"][1]), 1e-5, M, W, L.ptr(a_ln), st), "avc_vit_ln_pack")
            L.check(lib.avc_vit_linear_small(L.ptr(a_ln), L.ptr(blk["fc"].wp), L.ptr(blk["fc"].b), None, None, None,
                                             L.ptr(pres[i]) if keep else None, L.ptr(a_fc), M, 4 * W, W, 1, st), "avc_vit_linear_small")
            L.check(lib.avc_vit_linear_small(L.ptr(a_fc), L.ptr(blk["proj"].wp), L.ptr(blk["proj"].b), L.ptr(x2), None, L.ptr(out), None, None,
                                             M, W, 4 * W, 0, st), "avc_vit_linear_small")
            cur = out
        if keep:
            ctx.model, ctx.B = model, B
            ctx.save_for_backward(x0, xs, x2s, qkvs, pres)
        return xs[nb - 1].reshape(B, TOKENS, W)

    @staticmethod
    def backward(ctx, g):
        lib, st = L.load(), L.stream()
        model, B = ctx.model, ctx.B
        x0, xs, x2s, qkvs, pres = ctx.saved_tensors
        M, W = B * TOKENS, WIDTH
        dev = x0.device
        g = g.reshape(M, W).contiguous().float()
        p_g, p_dh = model._train_buf("g", M, W), model._train_buf("dh", M, 4 * W)
        p_g2, p_dqkv = model._train_buf("g2", M, W), model._train_buf("dqkv", M, 3 * W)
        t768 = torch.empty(5, M, W, device=dev, dtype=torch.float32)
        dy2, da, dy1, bufa, bufb = t768[0], t768[1], t768[2], t768[3], t768[4]
        L.check(lib.avc_vit_pack(L.ptr(g), None, L.ptr(p_g), M, W, st), "avc_vit_pack")
        for i in range(len(model.blocks) - 1, -1, -1):
            blk = model.blocks[i]
            xin = x0 if i == 0 else xs[i - 1]
            # c_proj^T, times QuickGELU'(pre), packed for c_fc^T
            L.check(lib.avc_vit_linear_small(L.ptr(p_g), L.ptr(blk["proj"].wtp), None, None, L.ptr(pres[i]), None, None, L.ptr(p_dh),
                                             M, 4 * W, W, 2, st), "avc_vit_linear_small")
            L.check(lib.avc_vit_linear_small(L.ptr(p_dh), L.ptr(blk["fc"].wtp), None, None, None, L.ptr(dy2), None, None,
                                             M, W, 4 * W, 0, st), "avc_vit_linear_small")
            # ln_2 backward + the residual branch's gradient: fp32 (the next residual sum) and packed (out_proj^T)
            L.check(lib.avc_vit_ln_bwd(L.ptr(dy2), L.ptr(x2s[i]), L.ptr(blk["ln2"][0]), 1e-5, L.ptr(g), L.ptr(bufa), L.ptr(p_g2), M, W, st),
                    "avc_vit_ln_bwd")
            L.check(lib.avc_vit_linear_small(L.ptr(p_g2), L.ptr(blk["out"].wtp), None, None, None, L.ptr(da), None, None,
                                             M, W, W, 0, st), "avc_vit_linear_small")
            L.check(lib.avc_vit_attention_bwd_packed(L.ptr(qkvs[i]), L.ptr(da), L.ptr(p_dqkv), B, TOKENS, W, HEADS, st),
                    "avc_vit_attention_bwd_packed")
            L.check(lib.avc_vit_linear_small(L.ptr(p_dqkv), L.ptr(blk["qkv"].wtp), None, None, None, L.ptr(dy1), None, None,
                                             M, W, 3 * W, 0, st), "avc_vit_linear_small")
            # ln_1 backward + residual: the gradient of the block's input, packed for the block below
            L.check(lib.avc_vit_ln_bwd(L.ptr(dy1), L.ptr(xin), L.ptr(blk["ln1"][0]), 1e-5, L.ptr(bufa), L.ptr(bufb),
                                       L.ptr(p_g) if i else None, M, W, st), "avc_vit_ln_bwd")
            g = bufb       # (bufa / bufb are re-used by the block below: each is dead by the time it is written again)
        return g.reshape(B, TOKENS, W), None


class ClipVisionB32:
    """`perceptor` stand-in: .encode_image(x[B,3,224,224]) -> [B,512] (differentiable wrt x only)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("ClipVisionB32 runs on the MI355X kernels only (no CPU fallback)")
        sd = {k: v for k, v in state_dict.items() if k.startswith("visual.")}
        f = lambda k: sd[k].float().to(dev).contiguous()
        self.device = dev
        self.conv = _Lin(sd["visual.conv1.weight"].reshape(WIDTH, -1), None, dev)
        self.cls = f("visual.class_embedding")
        self.pos = f("visual.positional_embedding")
        self.ln_pre = (f("visual.ln_pre.weight"), f("visual.ln_pre.bias"))
        self.ln_post = (f("visual.ln_post.weight"), f("visual.ln_post.bias"))
        self.blocks = []
        for i in range(LAYERS):
            p = "visual.transformer.resblocks.%d." % i
            self.blocks.append(dict(
                ln1=(f(p + "ln_1.weight"), f(p + "ln_1.bias")), ln2=(f(p + "ln_2.weight"), f(p + "ln_2.bias")),
                qkv=_Lin(sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], dev),
                out=_Lin(sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], dev),
                fc=_Lin(sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"], dev),
                proj=_Lin(sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"], dev)))
        self.proj = _Lin(sd["visual.proj"].t().contiguous(), None, dev)
        # the text tower is built on demand from the same state dict (clip_text.ClipTextB32) when it carries one
        self._text_sd = state_dict if "token_embedding.weight" in state_dict else None
        self._text = None
        self._packed = {}
        self._graphed = {}
        self._graph_busy = {}         # batch size -> a replayed forward is waiting for its backward
        self._graph_gen = {}          # batch size -> number of replays so far (a backward checks it still owns the static activations)
        self._warned_busy = False
        self._graph_ws = []           # packing workspaces the captured launches point at

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def cuda(self):
        return self

    def _packed_buf(self, tag, M, K):
        """a packed bf16 activation buffer of the batched pipeline (one per role, grown on demand, stream-ordered reuse)"""
        need = L.load().avc_vit_workspace_bytes(M, K)
        buf = self._packed.get(tag)
        if buf is None or buf.numel() < need:
            buf = self._packed[tag] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return buf

    def _train_buf(self, tag, M, K):
        """a packed operand of the per-iteration pipeline (BlocksFn): one per (role, shape), never re-allocated -- captured graphs
        point at it; zero-filled once (the attention kernel only writes the rows below M)"""
        key = ("train", tag, M, K)
        buf = self._packed.get(key)
        if buf is None:
            buf = self._packed[key] = torch.zeros(L.load().avc_vit_workspace_bytes(M, K), dtype=torch.uint8, device=self.device)
        return buf

    @torch.no_grad()
    def _encode_image_batched(self, image: torch.Tensor) -> torch.Tensor:
        """encode_image for scoring calls (no gradient, 3+ images: ShapeGen/main.py:104-128, pose_generation.py:79-110): the same
        arithmetic as the autograd path below (bf16 GEMM operands, fp32 accumulation, fp32 residual stream and LayerNorm
        statistics), with the activations handed from kernel to kernel as packed bf16 operands -- LayerNorm writes the operand of
        the linear behind it, attention that of the out-projection, c_fc + QuickGELU that of c_proj -- instead of fp32 rows and a
        packing pass in front of every linear."""
        lib, st = L.load(), L.stream()
        B = image.shape[0]
        x = image.float().reshape(B, 3, RES // PATCH, PATCH, RES // PATCH, PATCH).permute(0, 2, 4, 1, 3, 5)
        x = x.reshape(B, (RES // PATCH) ** 2, 3 * PATCH * PATCH)
        x = LinearFn.apply(x, self.conv, 0, None)
        x = torch.cat([self.cls.expand(B, 1, WIDTH), x], dim=1) + self.pos
        x = F.layer_norm(x, (WIDTH,), self.ln_pre[0], self.ln_pre[1], 1e-5).reshape(B * TOKENS, WIDTH).contiguous()
        M = B * TOKENS
        a768, b768, a3072 = self._packed_buf("ln", M, WIDTH), self._packed_buf("attn", M, WIDTH), self._packed_buf("fc", M, 4 * WIDTH)
        qkv = torch.empty(M, 3 * WIDTH, device=x.device, dtype=torch.float32)
        x2 = torch.empty_like(x)
        for blk in self.blocks:
            L.check(lib.avc_vit_ln_pack(L.ptr(x), L.ptr(blk["ln1"][0]), L.ptr(blk["ln1"][1]), 1e-5, M, WIDTH, L.ptr(a768), st), "avc_vit_ln_pack")
            L.check(lib.avc_vit_linear_packed(L.ptr(a768), L.ptr(blk["qkv"].wp), L.ptr(blk["qkv"].b), None, L.ptr(qkv), None,
                                              M, 3 * WIDTH, WIDTH, 0, st), "avc_vit_linear_packed")
            L.check(lib.avc_vit_attention_fwd_packed(L.ptr(qkv), L.ptr(b768), B, TOKENS, WIDTH, HEADS, st), "avc_vit_attention_fwd_packed")
            L.check(lib.avc_vit_linear_packed(L.ptr(b768), L.ptr(blk["out"].wp), L.ptr(blk["out"].b), L.ptr(x), L.ptr(x2), None,
                                              M, WIDTH, WIDTH, 0, st), "avc_vit_linear_packed")
            L.check(lib.avc_vit_ln_pack(L.ptr(x2), L.ptr(blk["ln2"][0]), L.ptr(blk["ln2"][1]), 1e-5, M, WIDTH, L.ptr(a768), st), "avc_vit_ln_pack")
            L.check(lib.avc_vit_linear_packed(L.ptr(a768), L.ptr(blk["fc"].wp), L.ptr(blk["fc"].b), None, None, L.ptr(a3072),
                                              M, 4 * WIDTH, WIDTH, 1, st), "avc_vit_linear_packed")
            L.check(lib.avc_vit_linear_packed(L.ptr(a3072), L.ptr(blk["proj"].wp), L.ptr(blk["proj"].b), L.ptr(x2), L.ptr(x), None,
                                              M, WIDTH, 4 * WIDTH, 0, st), "avc_vit_linear_packed")
        x = F.layer_norm(x.reshape(B, TOKENS, WIDTH)[:, 0, :], (WIDTH,), self.ln_post[0], self.ln_post[1], 1e-5)
        return LinearFn.apply(x, self.proj, 0, None)

    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        """main.py:512,524.  Three routes with the same arithmetic: scoring calls (no gradient, 3+ images) -> the packed pipeline;
        the per-iteration call (1-2 images WITH a gradient to the pixels) -> the encoder's ~250 forward and ~250 backward launches
        replayed as two HIP graphs (torch.cuda.make_graphed_callables: captured once per batch size after three warm-up passes;
        bit-identical results, 3.8 -> 0.34 ms of host time per forward + backward) -- what matters when the ray set is small and
        the iteration is bound by the host's launch rate (the reference's default silhouette mode: 7 000 rays); everything else eager."""
        B = image.shape[0]
        if PACKED_PIPELINE and B * TOKENS > 128 and not LIBRARY_GEMM and not (torch.is_grad_enabled() and image.requires_grad):
            return self._encode_image_batched(image)
        if (TRAIN_GRAPH and B <= 2 and image.is_cuda and torch.is_grad_enabled() and image.requires_grad
                and tuple(image.shape[1:]) == (3, RES, RES) and not torch.cuda.is_current_stream_capturing()):
            g = self._graphed.get(B)
            if g is None:
                sample = torch.zeros(B, 3, RES, RES, device=self.device, dtype=torch.float32, requires_grad=True)
                try:
                    with torch.cuda.device(self.device):     # (capture streams are created on the CURRENT device)
                        # the captured launches bake in the pointer of the linears' packing workspace: size it for the largest linear of
                        # this pass BEFORE the capture (no growth, i.e. no re-allocation, while launches are being recorded) ...
                        _ws(self.device, L.load().avc_vit_workspace_bytes(B * TOKENS, 4 * WIDTH))
                        g = torch.cuda.make_graphed_callables(self._encode_image_eager, (sample,))
                        # ... and keep that buffer alive for as long as the graphs exist: a later, larger eager call (batched scoring,
                        # M > 128) makes _ws() allocate a new one, and dropping the old one would leave the replays writing into freed memory
                        self._graph_ws.append(_workspace[str(self.device)])
                except Exception as e:      # a capture that the runtime refuses must not take the training run down: eager launches
                    import logging
                    logging.warning("CLIP encode_image: HIP graph capture failed (%s: %s); launching eagerly", type(e).__name__, str(e)[:200])
                    g = False
                self._graphed[B] = g
            # One captured instance per batch size = ONE set of static activations: a second call before the first one's backward would
            # overwrite them (and its embedding, which aliases the graph's static output).  The reference does exactly that when
            # add_no_texture is set (main.py:512 then :524, one image each); Runner merges the two into one B = 2 pass, other
            # callers get the eager launches for the overlapping call.  The instance is free again once its backward has run.
            if g is not False and self._graph_busy.get(B, False) and not self._warned_busy:
                self._warned_busy = True
                import warnings
                warnings.warn("ClipVisionB32.encode_image: the captured graphs for batch %d are still waiting for the backward of an "
                              "earlier call; this call runs as ~500 eager launches (Runner releases the graphs at the start of every "
                              "iteration; other callers: release_graphs())" % B, RuntimeWarning, stacklevel=2)
            if g is not False and not self._graph_busy.get(B, False):
                self._graph_busy[B] = True
                gen = self._graph_gen[B] = self._graph_gen.get(B, 0) + 1
                out = g(image.float())

                def _release(grad, B=B, gen=gen):
                    # the instance's static activations belong to its LATEST replay: a backward that arrives for an earlier one (its
                    # forward was declared finished by release_graphs() and the graph replayed since) would differentiate the newer
                    # call's activations -- silently wrong pixel gradients.  Refuse it.
                    if self._graph_gen[B] != gen:
                        raise RuntimeError("ClipVisionB32: backward of a graph-replayed encode_image (batch %d, replay %d) after release_graphs() "
                                           "and a later replay (%d) overwrote its captured activations; run that backward before the next "
                                           "iteration, or take the loss with TRAIN_GRAPH off" % (B, gen, self._graph_gen[B]))
                    self._graph_busy[B] = False
                    return grad
                out.register_hook(_release)
                return out.clone()       # (its own storage: the static output buffer is rewritten by the next replay)
        return self._encode_image_eager(image)

    def release_graphs(self):
        """Declare every graph-replayed forward of earlier calls finished.  The busy flag of a captured instance is cleared by a hook
        in its backward; a grad-enabled call that never reaches backward (an exception between forward and loss.backward(), a skipped
        non-finite loss, a probe call) would otherwise leave it set for good and every later call would silently take the eager path.
        Callers with a step structure call this where no earlier graph can still be wanted -- Runner.train_clip_iteration at its top."""
        for B in self._graph_busy:
            self._graph_busy[B] = False

    def _encode_image_eager(self, image: torch.Tensor) -> torch.Tensor:
        B = image.shape[0]
        # conv1 (32x32, stride 32, no bias) == GEMM over flattened patches in (c, ky, kx) order
        x = image.float().reshape(B, 3, RES // PATCH, PATCH, RES // PATCH, PATCH).permute(0, 2, 4, 1, 3, 5)
        x = x.reshape(B, (RES // PATCH) ** 2, 3 * PATCH * PATCH)
        x = LinearFn.apply(x, self.conv, 0, None)
        x = torch.cat([self.cls.expand(B, 1, WIDTH), x], dim=1) + self.pos
        x = F.layer_norm(x, (WIDTH,), self.ln_pre[0], self.ln_pre[1], 1e-5)
        if FUSED_BLOCKS and B * TOKENS <= 128 and not LIBRARY_GEMM:
            x = BlocksFn.apply(x, self)
            x = F.layer_norm(x[:, 0, :], (WIDTH,), self.ln_post[0], self.ln_post[1], 1e-5)
            return LinearFn.apply(x, self.proj, 0, None)
        for blk in self.blocks:
            y = F.layer_norm(x, (WIDTH,), blk["ln1"][0], blk["ln1"][1], 1e-5)
            a = AttentionFn.apply(LinearFn.apply(y, blk["qkv"], 0, None))
            x = LinearFn.apply(a, blk["out"], 0, x)                      # residual fused in the epilogue
            y = F.layer_norm(x, (WIDTH,), blk["ln2"][0], blk["ln2"][1], 1e-5)
            y = LinearFn.apply(y, blk["fc"], 1, None)                    # QuickGELU fused
            x = LinearFn.apply(y, blk["proj"], 0, x)
        x = F.layer_norm(x[:, 0, :], (WIDTH,), self.ln_post[0], self.ln_post[1], 1e-5)
        return LinearFn.apply(x, self.proj, 0, None)

    def encode_text(self, tokens):
        """main.py:276,282,288 (start-up only): needs a state dict that carries the text tower (a full OpenAI ViT-B/32
        checkpoint does; the seeded vision-only weights of the benchmark do not)"""
        if self._text is None:
            if self._text_sd is None:
                raise RuntimeError("this perceptor was built from vision-only weights: no text tower to encode prompts with "
                                   "(supply a full CLIP state dict, or cached prompt embeddings to Runner.init_clip)")
            from .clip_text import ClipTextB32
            self._text = ClipTextB32(self._text_sd, self.device)
        return self._text.encode_text(tokens)


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """An OpenAI CLIP ViT-B/32 checkpoint: the TorchScript archive `ViT-B-32.pt` that `clip.load` downloads, a plain
    `torch.save(model.state_dict())` file, or a .safetensors file with the same key names."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)     # a plain state dict: no code is executed
    except Exception:
        try:
            obj = torch.jit.load(path, map_location="cpu")                # OpenAI's ViT-B-32.pt is a TorchScript archive
        except RuntimeError:
            # a pickled module or other arbitrary pickle: executes code from the file, as the reference's clip.load does --
            # only with the explicit opt-in
            if os.environ.get("AVC_ALLOW_UNSAFE_PICKLE") != "1":
                raise RuntimeError("%s is neither a tensors-only checkpoint, a .safetensors file nor a TorchScript archive; loading it "
                                   "would execute pickled code (set AVC_ALLOW_UNSAFE_PICKLE=1 to allow that)" % path)
            obj = torch.load(path, map_location="cpu", weights_only=False)
    sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def clip_preprocess(img_hw3: torch.Tensor) -> torch.Tensor:
    """main.py:261-267,510-511: RandomResizedCrop(224, scale=(1,1)) of a square image == bilinear resize
    (align_corners=False, no antialias); RandomPerspective(p=0) == identity; then Normalize."""
    x = img_hw3.permute(2, 0, 1).unsqueeze(0)
    if x.shape[-1] != RES or x.shape[-2] != RES:
        x = F.interpolate(x, size=(RES, RES), mode="bilinear", align_corners=False)
    mean = h2d.const(CLIP_MEAN, x.device, x.dtype).view(1, 3, 1, 1)
    std = h2d.const(CLIP_STD, x.device, x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std
