"""Pins oracle/clip_vit_oracle.py architecturally against transformers' CLIP vision tower (same seeded weights)."""
import pytest
import torch

from oracle import clip_vit_oracle as C


def test_oracle_matches_transformers_clip_vision():
    tr = pytest.importorskip("transformers")
    sd = C.random_state_dict(0)
    cfg = tr.CLIPVisionConfig()  # defaults == ViT-B/32: hidden 768, 12 layers, 12 heads, patch 32, quick_gelu, proj 512
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.patch_size, cfg.projection_dim, cfg.hidden_act) == \
        (768, 12, 32, 512, "quick_gelu")
    model = tr.CLIPVisionModelWithProjection(cfg).eval()
    missing, unexpected = model.load_state_dict(C.to_hf_state_dict(sd), strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 224, 224, generator=g)
    with torch.no_grad():
        ref = model(pixel_values=img).image_embeds
        out = C.encode_image(sd, img)
    assert torch.allclose(out, ref, atol=2e-4, rtol=1e-4), (out - ref).abs().max()
