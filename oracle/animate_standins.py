"""Seeded stand-ins for the third-party blobs AvatarAnimate loads (TEST INFRASTRUCTURE ONLY): a VPoser-shaped body prior (decode: 32-d latent ->
{'pose_body': [B,21,3]}, encode(pose[B,63]).mean -> [B,32]; the real one is human_body_prior's VPoser with data/vposer weights) and a text
encoder.  They have the interfaces the reference calls (pose_generation.py:322-326, motion_generation.py:112-120), nothing of their content."""
import types

import torch


class StandInVPoser(torch.nn.Module):
    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("dec_w", torch.randn(32, 63, generator=g) * 0.35)
        self.register_buffer("dec_b", torch.randn(63, generator=g) * 0.1)
        self.register_buffer("enc_w", torch.randn(63, 32, generator=g) * 0.2)

    def decode(self, z):
        return {"pose_body": torch.tanh(z @ self.dec_w + self.dec_b).reshape(z.shape[0], 21, 3)}

    def encode(self, pose):
        return types.SimpleNamespace(mean=pose @ self.enc_w)


def text_feature_of(text, seed=0):
    """a deterministic unit vector per prompt (the real one: CLIP's text tower on clip.tokenize(text))"""
    g = torch.Generator().manual_seed(seed + sum(ord(c) * (i + 1) for i, c in enumerate(text)) % 100003)
    return torch.nn.functional.normalize(torch.randn(512, generator=g), dim=0)
