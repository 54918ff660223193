"""CPU fp32 restatement of the CLIP ViT-B/32 TEXT tower (TEST INFRASTRUCTURE: imported only by tests/).

Reference call sites: `self.perceptor.encode_text(clip.tokenize([prompt]).cuda())` (AvatarGen/AppearanceGen/main.py:273-288);
the arithmetic lives in the third-party package `clip @ git+https://github.com/openai/CLIP.git` (requirements.txt:12,
unpinned, not vendored; weights are downloaded at run time).  Published algorithm (clip/model.py `CLIP.encode_text`):

    x = token_embedding[tokens] + positional_embedding            [B,77,512]
    12 x ResidualAttentionBlock(width 512, 8 heads, causal mask, QuickGELU MLP 2048)
    x = ln_final(x);  x = x[arange(B), tokens.argmax(-1)] @ text_projection      (the EOT token has the largest id)

State-dict keys are OpenAI's: token_embedding.weight, positional_embedding, transformer.resblocks.{i}.*, ln_final.*,
text_projection.  Pinned architecturally against transformers.CLIPTextModelWithProjection with mapped seeded weights
(tests/test_clip_text.py).  **Parity against the real OpenAI weights: unpinned** (no weights / test vectors offline).
"""
from typing import Dict

import torch
import torch.nn.functional as F

WIDTH, LAYERS, HEADS, CTX, VOCAB, EMBED = 512, 12, 8, 77, 49408, 512


def random_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    sd = {"token_embedding.weight": rn(VOCAB, WIDTH, std=0.02), "positional_embedding": rn(CTX, WIDTH, std=0.01),
          "ln_final.weight": 1 + rn(WIDTH, std=0.05), "ln_final.bias": rn(WIDTH, std=0.05),
          "text_projection": rn(WIDTH, EMBED, std=WIDTH ** -0.5)}
    for i in range(LAYERS):
        p = "transformer.resblocks.%d." % i
        sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"] = rn(3 * WIDTH, WIDTH, std=WIDTH ** -0.5), rn(3 * WIDTH, std=0.02)
        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = rn(WIDTH, WIDTH, std=WIDTH ** -0.5 * (2 * LAYERS) ** -0.5), rn(WIDTH, std=0.02)
        for n in ("ln_1", "ln_2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = 1 + rn(WIDTH, std=0.05), rn(WIDTH, std=0.05)
        sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = rn(4 * WIDTH, WIDTH, std=(2 * WIDTH) ** -0.5), rn(4 * WIDTH, std=0.02)
        sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = rn(WIDTH, 4 * WIDTH, std=WIDTH ** -0.5 * (2 * LAYERS) ** -0.5), rn(WIDTH, std=0.02)
    return sd


def encode_text(sd: Dict[str, torch.Tensor], tokens: torch.Tensor) -> torch.Tensor:
    B, T = tokens.shape
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"][:T]
    mask = torch.full((T, T), float("-inf")).triu_(1)          # clip/model.py build_attention_mask
    hd = WIDTH // HEADS
    for i in range(LAYERS):
        p = "transformer.resblocks.%d." % i
        y = F.layer_norm(x, (WIDTH,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
        qkv = y @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
        q, k, v = [t.reshape(B, T, HEADS, hd).transpose(1, 2) for t in qkv.split(WIDTH, dim=-1)]
        att = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5 + mask, dim=-1)
        a = (att @ v).transpose(1, 2).reshape(B, T, WIDTH)
        x = x + a @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]
        y = F.layer_norm(x, (WIDTH,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
        y = y @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"]
        y = y * torch.sigmoid(1.702 * y)
        x = x + y @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"]
    x = F.layer_norm(x, (WIDTH,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    return x[torch.arange(B), tokens.argmax(dim=-1)] @ sd["text_projection"]


def to_hf_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """OpenAI key layout -> transformers.CLIPTextModelWithProjection"""
    out = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
           "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
           "text_model.final_layer_norm.weight": sd["ln_final.weight"], "text_model.final_layer_norm.bias": sd["ln_final.bias"],
           "text_projection.weight": sd["text_projection"].t().contiguous()}
    for i in range(LAYERS):
        p, q = "transformer.resblocks.%d." % i, "text_model.encoder.layers.%d." % i
        w, b = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            out[q + "self_attn.%s.weight" % n] = w[j * WIDTH:(j + 1) * WIDTH]
            out[q + "self_attn.%s.bias" % n] = b[j * WIDTH:(j + 1) * WIDTH]
        out[q + "self_attn.out_proj.weight"], out[q + "self_attn.out_proj.bias"] = sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"]
        out[q + "layer_norm1.weight"], out[q + "layer_norm1.bias"] = sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]
        out[q + "layer_norm2.weight"], out[q + "layer_norm2.bias"] = sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]
        out[q + "mlp.fc1.weight"], out[q + "mlp.fc1.bias"] = sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]
        out[q + "mlp.fc2.weight"], out[q + "mlp.fc2.bias"] = sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"]
    return out
