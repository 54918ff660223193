"""CLIP ViT-B/32 text tower on the MI355X kernels: `encode_text(tokens)` of the reference's perceptor
(AvatarGen/AppearanceGen/main.py:276,282,288; OpenAI clip/model.py `CLIP.encode_text`).  Same bf16-MFMA linear kernel as the
image tower (`avc_vit_linear`), causal attention over 77 tokens (`avc_text_attention_fwd`).  Prompts are encoded once per run
and detached, so there is no backward."""
from typing import Dict

import torch
import torch.nn.functional as F

from . import lib as L
from .clip_vit import _Lin, _linear_raw

WIDTH, LAYERS, HEADS, CTX = 512, 12, 8, 77


def has_text_tower(state_dict) -> bool:
    return "token_embedding.weight" in state_dict and "text_projection" in state_dict


class ClipTextB32:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("ClipTextB32 runs on the MI355X kernels only (no CPU fallback)")
        f = lambda k: state_dict[k].float().to(dev).contiguous()
        self.device = dev
        self.tok = f("token_embedding.weight")
        self.pos = f("positional_embedding")
        self.ln_final = (f("ln_final.weight"), f("ln_final.bias"))
        self.proj = _Lin(state_dict["text_projection"].float().t().contiguous(), None, dev)
        self.blocks = []
        for i in range(LAYERS):
            p = "transformer.resblocks.%d." % i
            self.blocks.append(dict(
                ln1=(f(p + "ln_1.weight"), f(p + "ln_1.bias")), ln2=(f(p + "ln_2.weight"), f(p + "ln_2.bias")),
                qkv=_Lin(state_dict[p + "attn.in_proj_weight"], state_dict[p + "attn.in_proj_bias"], dev),
                out=_Lin(state_dict[p + "attn.out_proj.weight"], state_dict[p + "attn.out_proj.bias"], dev),
                fc=_Lin(state_dict[p + "mlp.c_fc.weight"], state_dict[p + "mlp.c_fc.bias"], dev),
                proj=_Lin(state_dict[p + "mlp.c_proj.weight"], state_dict[p + "mlp.c_proj.bias"], dev)))

    @staticmethod
    def _lin(x2d, lin, act=0, residual=None):
        y, _ = _linear_raw(x2d, lin, False, lin.b, residual, act, False)
        return y

    @torch.no_grad()
    def encode_text(self, tokens: torch.Tensor) -> torch.Tensor:
        tokens = tokens.to(self.device).long()
        B, T = tokens.shape
        lib = L.load()
        outs = []
        for b in range(B):      # one sequence (77 rows) per pass: the linear kernel takes up to 128 rows
            x = (self.tok[tokens[b]] + self.pos[:T]).contiguous()
            for blk in self.blocks:
                y = F.layer_norm(x, (WIDTH,), blk["ln1"][0], blk["ln1"][1], 1e-5)
                qkv = self._lin(y, blk["qkv"]).contiguous()
                a = torch.empty(T, WIDTH, device=self.device, dtype=torch.float32)
                L.check(lib.avc_text_attention_fwd(L.ptr(qkv), L.ptr(a), 1, T, WIDTH, HEADS, 1, L.stream()), "avc_text_attention_fwd")
                x = self._lin(a, blk["out"], 0, x)
                y = F.layer_norm(x, (WIDTH,), blk["ln2"][0], blk["ln2"][1], 1e-5)
                y = self._lin(y, blk["fc"], 1)
                x = self._lin(y, blk["proj"], 0, x)
            x = F.layer_norm(x, (WIDTH,), self.ln_final[0], self.ln_final[1], 1e-5)
            eot = int(tokens[b].argmax())
            outs.append(self._lin(x[eot:eot + 1].contiguous(), self.proj))
        return torch.cat(outs, 0)
