"""CLIP scoring of rendered bodies: the encode_image consumers outside AppearanceGen.

  * ShapeGen codebook search (AvatarGen/ShapeGen/main.py:93-118): the shape whose pre-computed CLIP embedding moves from
    the neutral body's embedding in the direction "target text - neutral text".
  * AvatarAnimate pose scoring (AvatarAnimate/models/pose_generation.py:63-100): mean CLIP embedding of a pose rendered from
    several cameras, cosine against the text; top-k ranking; the per-frame CLIP term of the motion optimiser
    (AvatarAnimate/models/motion_generation.py:335-344).

The renders themselves (smplx + neural_renderer with the UV texture, VPoser) are inputs: `images` are what
`render_one_batch` returns ([B,3,H,W] in [0,1]).  Everything here runs on the HIP ViT kernels through `perceptor`
(avatarclip_amd.clip_vit.ClipVisionB32, batches of any size are processed in 128-row GEMM launches)."""
import torch
import torch.nn.functional as F

from .clip_vit import CLIP_MEAN, CLIP_STD
from . import h2d


def preprocess_renders(images: torch.Tensor) -> torch.Tensor:
    """ShapeGen/main.py:104-106 and pose_generation.py:79-83: F.interpolate(images, size=224) (default mode: NEAREST) and
    the CLIP normalisation."""
    x = F.interpolate(images.float(), size=224)
    mean = h2d.const(CLIP_MEAN, x.device, x.dtype).view(1, 3, 1, 1)
    std = h2d.const(CLIP_STD, x.device, x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def render_embedding(perceptor, images: torch.Tensor) -> torch.Tensor:
    """encode_image of a batch of renders -> [B,512] fp32"""
    return perceptor.encode_image(preprocess_renders(images)).float()


def shape_codebook_search(clip_codebook: torch.Tensor, neutral_image_embed: torch.Tensor, neutral_text_embed: torch.Tensor,
                          target_text_embed: torch.Tensor):
    """ShapeGen/main.py:97-114 -> (index of the best code, the per-code cosine).  clip_codebook [N,512]; the image
    embedding is the mean over the neutral body's renders; text embeddings [1,512] or [512]."""
    delta = (target_text_embed.float() - neutral_text_embed.float()).reshape(-1)
    cos = (F.normalize(clip_codebook.float() - neutral_image_embed.float().reshape(1, -1), dim=1) * F.normalize(delta, dim=0)).sum(-1).reshape(-1)
    return int(cos.argmax()), cos


def pose_feature(perceptor, images: torch.Tensor, num_camera: int) -> torch.Tensor:
    """pose_generation.py:84-88: `images` = [num_camera * bs, 3, H, W] (camera-major, as render_one_batch concatenates
    them) -> [bs,512] mean over the cameras"""
    emb = render_embedding(perceptor, images)
    return emb.view(num_camera, -1, emb.shape[-1]).mean(0)


def pose_score(text_feature: torch.Tensor, pose_feat: torch.Tensor) -> torch.Tensor:
    """pose_generation.py:90-94 (cosine_similarity of [1,512] against [bs,512])"""
    return F.cosine_similarity(text_feature.float().reshape(1, -1), pose_feat.float())


def rank_poses(text_feature: torch.Tensor, pose_feats: torch.Tensor, topk: int):
    """pose_generation.py:96-98 / the codebook generators: poses sorted by score, best first -> (indices, scores)"""
    s = pose_score(text_feature, pose_feats)
    order = torch.argsort(s, descending=True, stable=True)[:topk]
    return order, s[order]


def motion_clip_loss(pose_feats: torch.Tensor, text_feature: torch.Tensor, st_idx: int, clip_num_part: int, num_frame: int):
    """motion_generation.py:335-344: frames st_idx, st_idx + P, ... of the motion, weighted by their position in time"""
    per_pose = 1 - F.cosine_similarity(pose_feats.float(), text_feature.float().reshape(1, -1))
    coef = (st_idx + torch.arange(per_pose.shape[0], device=per_pose.device, dtype=per_pose.dtype) * clip_num_part) / num_frame
    return (coef * per_pose).sum()
