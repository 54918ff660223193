"""CPU oracle for the CLIP ViT-B/32 image encoder (the `perceptor.encode_image` of main.py:259,512).

TEST INFRASTRUCTURE ONLY.  The arithmetic lives in a third-party, un-vendored dependency of the reference:
`clip @ git+https://github.com/openai/CLIP.git` (requirements.txt:12, no commit pin; weights ViT-B-32.pt are
downloaded by clip.load at run time and are absent offline).  This file restates the published architecture
(clip/model.py: VisionTransformer, ResidualAttentionBlock, QuickGELU, LayerNorm(eps=1e-5)) with OpenAI's state-dict
key names, so a real ViT-B-32 state dict loads unmodified.  PARITY STATUS: pinned *architecturally* against
transformers.CLIPVisionModelWithProjection (same weights mapped, tests/test_clip_oracle.py); parity against the
real OpenAI weights is UNPINNED because the reference holds no test vector for this boundary and the weights
cannot be fetched here.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

WIDTH, LAYERS, HEADS, PATCH, RES, EMBED = 768, 12, 12, 32, 224, 512
TOKENS = (RES // PATCH) ** 2 + 1


def random_state_dict(seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded weights with OpenAI's init scales (clip/model.py: initialize_parameters) and key names."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g, dtype=dtype) * std
    sd = {}
    scale = WIDTH ** -0.5
    sd["visual.conv1.weight"] = rn(WIDTH, 3, PATCH, PATCH, std=0.02)
    sd["visual.class_embedding"] = rn(WIDTH, std=scale)
    sd["visual.positional_embedding"] = rn(TOKENS, WIDTH, std=scale)
    for n in ("ln_pre", "ln_post"):
        sd["visual.%s.weight" % n] = 1.0 + rn(WIDTH, std=0.05)
        sd["visual.%s.bias" % n] = rn(WIDTH, std=0.05)
    proj_std = (WIDTH ** -0.5) * ((2 * LAYERS) ** -0.5)
    attn_std = WIDTH ** -0.5
    fc_std = (2 * WIDTH) ** -0.5
    for i in range(LAYERS):
        p = "visual.transformer.resblocks.%d." % i
        sd[p + "attn.in_proj_weight"] = rn(3 * WIDTH, WIDTH, std=attn_std)
        sd[p + "attn.in_proj_bias"] = rn(3 * WIDTH, std=0.02)
        sd[p + "attn.out_proj.weight"] = rn(WIDTH, WIDTH, std=proj_std)
        sd[p + "attn.out_proj.bias"] = rn(WIDTH, std=0.02)
        sd[p + "ln_1.weight"] = 1.0 + rn(WIDTH, std=0.05)
        sd[p + "ln_1.bias"] = rn(WIDTH, std=0.05)
        sd[p + "ln_2.weight"] = 1.0 + rn(WIDTH, std=0.05)
        sd[p + "ln_2.bias"] = rn(WIDTH, std=0.05)
        sd[p + "mlp.c_fc.weight"] = rn(4 * WIDTH, WIDTH, std=fc_std)
        sd[p + "mlp.c_fc.bias"] = rn(4 * WIDTH, std=0.02)
        sd[p + "mlp.c_proj.weight"] = rn(WIDTH, 4 * WIDTH, std=proj_std)
        sd[p + "mlp.c_proj.bias"] = rn(WIDTH, std=0.02)
    sd["visual.proj"] = rn(WIDTH, EMBED, std=scale)
    return sd


def outlier_state_dict(seed: int = 0, n_channels: int = 8, gain: float = 30.0, n_fc_rows: int = 4, fc_gain: float = 20.0,
                       ln_blocks=(), ln_gain=None):
    """An OFFLINE PROXY for the dynamic range of trained CLIP weights (the real ViT-B-32.pt cannot be fetched here): the seeded
    state dict with the signature of trained ViTs' "massive activation" channels.
      * default: `n_channels` residual channels are inflated `gain` times in the class / CLS-position embedding and in the ln_pre
        gain, so the CLS token enters -- and, through the residual connections, stays in -- every block with a handful of channels
        30x above the median; `n_fc_rows` rows of every mlp.c_fc are `fc_gain` times larger.
      * ln_blocks = range(12): additionally the ln_1 / ln_2 gains of those channels are inflated (`ln_gain` times, default = `gain`)
        in the listed blocks.  With all 24
        gains at 30x the seeded (untrained) network becomes chaotic in its INPUT: merely rounding the linear operands to bf16 on the
        CPU (fp32 accumulation) changes d cos / d pixels by O(1) while the embedding moves by 3e-4 (tests/test_gpu_clip.py prints it).
    Not a substitute for the real weights (test_real_openai_weights_when_supplied stays the real check); it shows which tolerance
    of the bf16-operand kernels survives outlier channels far above the median."""
    sd = dict(random_state_dict(seed))
    g = torch.Generator().manual_seed(1000 + seed)
    ch = torch.randperm(WIDTH, generator=g)[:n_channels]
    sd["visual.class_embedding"] = sd["visual.class_embedding"].clone()
    sd["visual.class_embedding"][ch] *= gain
    sd["visual.positional_embedding"] = sd["visual.positional_embedding"].clone()
    sd["visual.positional_embedding"][0, ch] *= gain
    sd["visual.ln_pre.weight"] = sd["visual.ln_pre.weight"].clone()
    sd["visual.ln_pre.weight"][ch] *= gain
    for i in range(LAYERS):
        p = "visual.transformer.resblocks.%d." % i
        if i in ln_blocks:
            for n in ("ln_1.weight", "ln_2.weight"):
                sd[p + n] = sd[p + n].clone()
                sd[p + n][ch] *= gain if ln_gain is None else ln_gain
        rows = torch.randperm(4 * WIDTH, generator=g)[:n_fc_rows]
        sd[p + "mlp.c_fc.weight"] = sd[p + "mlp.c_fc.weight"].clone()
        sd[p + "mlp.c_fc.weight"][rows] *= fc_gain
    return sd


def encode_image_bf16_operands(sd, image):
    """encode_image with the operands of every linear rounded to bf16 and fp32 accumulation -- the arithmetic class of the HIP
    encoder (and of the reference's fp16 CLIP on a GPU), as a CPU statement; used to separate "the kernel is wrong" from "bf16
    operands move this network" in the outlier tests"""
    orig = F.linear

    def qlin(x, w, b=None):
        return orig(x.bfloat16().float(), w.bfloat16().float(), b)
    F.linear = qlin
    try:
        return encode_image(sd, image)
    finally:
        F.linear = orig


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def encode_image(sd: Dict[str, torch.Tensor], image: torch.Tensor) -> torch.Tensor:
    """VisionTransformer.forward: image [B,3,224,224] (already CLIP-normalised) -> [B,512]."""
    B = image.shape[0]
    x = F.conv2d(image, sd["visual.conv1.weight"], stride=PATCH)            # [B,768,7,7]
    x = x.reshape(B, WIDTH, -1).permute(0, 2, 1)                              # [B,49,768]
    cls = sd["visual.class_embedding"].to(x.dtype) + torch.zeros(B, 1, WIDTH, dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    x = F.layer_norm(x, (WIDTH,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    hd = WIDTH // HEADS
    for i in range(LAYERS):
        p = "visual.transformer.resblocks.%d." % i
        y = F.layer_norm(x, (WIDTH,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
        qkv = F.linear(y, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
        q, k, v = qkv.split(WIDTH, dim=-1)
        sh = lambda t: t.reshape(B, TOKENS, HEADS, hd).permute(0, 2, 1, 3)
        q, k, v = sh(q), sh(k), sh(v)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        o = (att @ v).permute(0, 2, 1, 3).reshape(B, TOKENS, WIDTH)
        x = x + F.linear(o, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        y = F.layer_norm(x, (WIDTH,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
        y = quick_gelu(F.linear(y, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
        x = x + F.linear(y, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    x = F.layer_norm(x[:, 0, :], (WIDTH,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    return x @ sd["visual.proj"]


def to_hf_state_dict(sd):
    """OpenAI key layout -> transformers.CLIPVisionModelWithProjection (used only to cross-check this oracle)."""
    out = {}
    vm = "vision_model."
    out[vm + "embeddings.class_embedding"] = sd["visual.class_embedding"]
    out[vm + "embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    out[vm + "embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    out[vm + "pre_layrnorm.weight"] = sd["visual.ln_pre.weight"]
    out[vm + "pre_layrnorm.bias"] = sd["visual.ln_pre.bias"]
    out[vm + "post_layernorm.weight"] = sd["visual.ln_post.weight"]
    out[vm + "post_layernorm.bias"] = sd["visual.ln_post.bias"]
    for i in range(LAYERS):
        p = "visual.transformer.resblocks.%d." % i
        q = vm + "encoder.layers.%d." % i
        w, b = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            out[q + "self_attn.%s.weight" % n] = w[j * WIDTH:(j + 1) * WIDTH]
            out[q + "self_attn.%s.bias" % n] = b[j * WIDTH:(j + 1) * WIDTH]
        out[q + "self_attn.out_proj.weight"] = sd[p + "attn.out_proj.weight"]
        out[q + "self_attn.out_proj.bias"] = sd[p + "attn.out_proj.bias"]
        out[q + "layer_norm1.weight"] = sd[p + "ln_1.weight"]
        out[q + "layer_norm1.bias"] = sd[p + "ln_1.bias"]
        out[q + "layer_norm2.weight"] = sd[p + "ln_2.weight"]
        out[q + "layer_norm2.bias"] = sd[p + "ln_2.bias"]
        out[q + "mlp.fc1.weight"] = sd[p + "mlp.c_fc.weight"]
        out[q + "mlp.fc1.bias"] = sd[p + "mlp.c_fc.bias"]
        out[q + "mlp.fc2.weight"] = sd[p + "mlp.c_proj.weight"]
        out[q + "mlp.fc2.bias"] = sd[p + "mlp.c_proj.bias"]
    out["visual_projection.weight"] = sd["visual.proj"].t().contiguous()
    return out
