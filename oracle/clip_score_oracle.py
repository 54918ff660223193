"""numpy restatement of the CLIP scoring arithmetic of ShapeGen / AvatarAnimate -- TEST INFRASTRUCTURE ONLY.
Follows AvatarGen/ShapeGen/main.py:104-114, AvatarAnimate/models/pose_generation.py:79-100 and
AvatarAnimate/models/motion_generation.py:335-344 line by line (loops instead of broadcasting where the reference loops)."""
import numpy as np

MEAN = np.array([0.48145466, 0.4578275, 0.40821073])
STD = np.array([0.26862954, 0.26130258, 0.27577711])


def preprocess(images):
    """F.interpolate(images, size=224) (nearest: src = floor(dst * in / out)), then (x - mean) / std."""
    B, C, H, W = images.shape
    iy = np.floor(np.arange(224) * (H / 224.0)).astype(np.int64).clip(0, H - 1)
    ix = np.floor(np.arange(224) * (W / 224.0)).astype(np.int64).clip(0, W - 1)
    x = images[:, :, iy][:, :, :, ix].astype(np.float64)
    return (x - MEAN.reshape(1, 3, 1, 1)) / STD.reshape(1, 3, 1, 1)


def _normalize(v, axis):
    return v / np.maximum(np.linalg.norm(v, axis=axis, keepdims=True), 1e-12)


def shape_codebook_search(clip_codebook, neutral_image_embed, nembed, tembed):
    delta = (tembed - nembed).reshape(-1)
    cos = (_normalize(clip_codebook - neutral_image_embed.reshape(1, -1), 1) * _normalize(delta, 0)).sum(-1)
    return int(np.argmax(cos)), cos


def pose_feature(embeds, num_camera):
    return embeds.reshape(num_camera, -1, embeds.shape[-1]).mean(0)


def cosine(a, b):
    return (a * b).sum(-1) / np.maximum(np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1), 1e-8)


def motion_clip_loss(pose_feats, text_feature, st_idx, clip_num_part, num_frame):
    per = 1 - cosine(pose_feats, text_feature.reshape(1, -1))
    loss = 0.0
    for j in range(per.shape[0]):
        loss = loss + (st_idx + j * clip_num_part) / num_frame * per[j]
    return loss
