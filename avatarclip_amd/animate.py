"""AvatarAnimate's candidate-pose and motion generators (reference: AvatarAnimate/models/pose_generation.py,
motion_generation.py, builder.py, main.py) on this repository's kernels: every render goes through the HIP rasteriser (`smpl_prior.MeshPrior`,
camera_mode 'look_at'), every CLIP embedding through the HIP ViT / text towers (`clip_vit.ClipVisionB32`, batched scoring path), SMPL posing
through `smpl_lbs`.  The reference's third-party blobs are INPUTS, not part of the tree: the VPoser body prior (`human_body_prior`), the pose
codebook / conditional RealNVP / motion-VAE checkpoints, the SMPL model, the UV texture.  `AnimateContext` takes them as objects / state dicts with
the reference's names, so the reference's files load unmodified where somebody holds them; the tests drive everything with seeded stand-ins and pin
the arithmetic to the reference's OWN classes and methods (`oracle/gen_golden_animate.py` extracts them with `ast`).

What is built                                                                  reference
  AnimateContext.get_text_feature / get_pose_feature / calculate_pose_score     pose_generation.py:56-99, motion_generation.py:64-97
  VPoserCodebook   (codebook retrieval + duplicate suppression: the default)    pose_generation.py:288-329
  VPoserRealNVP    (conditional flow: decode / encode / sample / get_pose)      pose_generation.py:176-286
  MotionInterpolation (linear walk through VPoser's latent space)               motion_generation.py:100-137
  MotionOptimizer  (transformer motion-VAE decoder + reconstruction / delta     motion_generation.py:140-358
                    losses; `clip_coef` must be 0, see below)
  build_pose_generator / build_motion_generator / main (conf-driven CLI)        builder.py, main.py

What is NOT built, and why: PoseOptimizer, VPoserOptimizer and MotionOptimizer's CLIP term (clip_coef > 0) differentiate the CLIP score with
respect to the POSE through neural_renderer's backward pass -- a hand-designed pseudo-gradient of the rasteriser that cannot be restated without
its source.  Those entry points raise NotImplementedError naming exactly that; the reference's own `motion_ablation/baseline`
(clip_coef = 0) and `motion_ablation/interpolation` confs, and both pose confs that need no renderer gradient, run.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import clip_score

DEFAULT_ANGLES = (120, 150, 180, 210, 240)          # pose_generation.py:76-77
CAMERA_DISTANCE = 2.0                               # models/render.py:13


def pose_padding(pose):
    """63 body-pose values -> SMPL's 69 (the two hand joints stay at rest); 69 pass through  (pose_generation.py:19-24)"""
    if pose.shape[-1] == 69:
        return pose
    if pose.shape[-1] != 63:
        raise ValueError("a body pose has 63 or 69 values, got %d" % pose.shape[-1])
    return torch.cat([pose, pose.new_zeros(pose.shape[:-1] + (6,))], dim=-1)


# ----------------------------------------------------------------------------------------------------------------- rotations (models/utils.py)
def axis_angle_to_matrix(aa):
    """Rodrigues through the unit quaternion, as models/utils.py:143-221 composes it (angle -> 0 handled by the series of sin(x/2)/x)"""
    ang = aa.norm(dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - ang * ang / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    w, xyz = torch.cos(half), aa * k
    x, y, z = xyz.unbind(-1)
    w = w.squeeze(-1)
    s = 2.0 / (w * w + x * x + y * y + z * z)
    rows = [1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
            s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
            s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]
    return torch.stack(rows, -1).reshape(aa.shape[:-1] + (3, 3))


def matrix_to_rotation_6d(m):
    return m[..., :2, :].reshape(m.shape[:-2] + (6,))


def rotation_6d_to_matrix(d6):
    """Gram-Schmidt on the two stored rows (Zhou et al.; models/utils.py:125-141)"""
    b1 = F.normalize(d6[..., :3], dim=-1)
    a2 = d6[..., 3:]
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack([b1, b2, torch.linalg.cross(b1, b2, dim=-1)], dim=-2)


def matrix_to_axis_angle(m):
    """rotation matrix -> quaternion (the best-conditioned of the four candidates) -> axis-angle: the route of MotionOptimizer.decode
    (motion_generation.py:299-302 through models/utils.py:23-123)"""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.reshape(m.shape[:-2] + (9,)).unbind(-1)
    q2 = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    pos = q2 > 0                                            # sqrt(max(0, x)) with a ZERO subgradient at x <= 0 (utils.py:11-20): the optimiser differentiates through this
    q_abs = torch.where(pos, torch.sqrt(torch.where(pos, q2, torch.ones_like(q2))), torch.zeros_like(q2))
    cand = torch.stack([torch.stack([q2[..., 0].clamp(min=0), m21 - m12, m02 - m20, m10 - m01], -1),
                        torch.stack([m21 - m12, q2[..., 1].clamp(min=0), m10 + m01, m02 + m20], -1),
                        torch.stack([m02 - m20, m10 + m01, q2[..., 2].clamp(min=0), m12 + m21], -1),
                        torch.stack([m10 - m01, m20 + m02, m21 + m12, q2[..., 3].clamp(min=0)], -1)], -2)
    cand = cand / (2.0 * q_abs.clamp(min=0.1))[..., None]
    best = q_abs.argmax(-1)
    quat = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    n = quat[..., 1:].norm(dim=-1, keepdim=True)
    half = torch.atan2(n, quat[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - ang * ang / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return quat[..., 1:] / k


# ----------------------------------------------------------------------------------------------------------------- shared context
class AnimateContext:
    """What BasePoseGenerator / BaseMotionGenerator build in their constructors (pose_generation.py:31-49), handed in instead:

      perceptor     clip_vit.ClipVisionB32 (HIP) -- or anything with encode_image([B,3,224,224]) -> [B,512]
      text_feature  callable(text) -> [512] (ClipVisionB32.encode_text on the tokenised prompt, a cached embedding, ...)
      smpl          smpl_lbs.load_smpl_arrays(...) dict (v_template, posedirs, J_regressor, parents, lbs_weights, faces)
      vposer        VPoser-like object: decode(z[B,32]) -> {'pose_body': [B,21,3]}, encode(pose[B,63]).mean -> [B,32]
      render_fn     callable(vertices[bs,V,3] tensor, faces, angles) -> images [len(angles) * bs, 3, H, W] in [0,1], camera-major
                    (None: the HIP rasteriser below; the reference textures the body with data/smpl_uv.obj, which its repository does not hold)
    """

    def __init__(self, perceptor, text_feature, smpl, vposer, render_fn=None, device=None, image_size=256):
        self.perceptor, self.text_feature, self.smpl, self.vp = perceptor, text_feature, smpl, vposer
        self.device = torch.device(device) if device is not None else smpl["v_template"].device
        self.image_size = image_size
        self.render_fn = render_fn if render_fn is not None else self._render_hip

    def get_text_feature(self, text):
        with torch.no_grad():
            return self.text_feature(text).reshape(-1).float().to(self.device)

    def posed_vertices(self, pose):
        """SMPL vertices of body poses [bs, 63 | 69] with the root turned by pi/2 about x (pose_generation.py:70-75)"""
        from . import smpl_lbs
        pose = pose_padding(pose.reshape(-1, pose.shape[-1]).float().to(self.device))
        bs = pose.shape[0]
        root = pose.new_zeros(bs, 1, 3)
        root[:, 0, 0] = math.pi / 2
        full = torch.cat([root, pose.reshape(bs, 23, 3)], dim=1)
        rot = smpl_lbs.batch_rodrigues(full.reshape(-1, 3)).reshape(bs, 24, 3, 3)
        s = self.smpl
        v, _ = smpl_lbs.lbs(s["v_template"][None].expand(bs, -1, -1), rot, s["posedirs"], s["J_regressor"], s["parents"], s["lbs_weights"])
        return v

    def _render_hip(self, vertices, faces, angles):
        """models/render.py:10-39 on the HIP rasteriser: camera_mode 'look_at' at distance 2, azimuth = the angle, elevation drawn per angle
        from numpy's global generator (np.random.randn() * 0.3 degrees: the reference's draw, in its order).  White body under
        neural_renderer's light -- the UV texture is an input of `render_fn` replacements."""
        from .shapegen_render import get_points_from_angles
        from .smpl_prior import MeshPrior
        eyes = [get_points_from_angles(CAMERA_DISTANCE, np.random.randn() * 0.3, a) for a in angles]     # draw order: once per angle, before the batch loop
        priors = [MeshPrior(v.detach().cpu().numpy(), faces, device=self.device, image_size=self.image_size) for v in vertices]
        out = []
        for eye in eyes:
            for p in priors:
                g = p.render_grey(eye.astype(np.float32), (-eye / np.linalg.norm(eye)).astype(np.float32))
                out.append(g.unsqueeze(0).expand(3, -1, -1))
        return torch.stack(out)

    def get_pose_feature(self, pose, angles=None):
        """mean CLIP embedding of the posed body over the cameras -> [bs, 512]  (pose_generation.py:63-89)"""
        angles = DEFAULT_ANGLES if angles is None else tuple(angles)
        v = self.posed_vertices(pose)
        images = self.render_fn(v, self.smpl["faces"], angles)
        return clip_score.pose_feature(self.perceptor, images.to(self.device), len(angles))

    def calculate_pose_score(self, text, pose):
        return float(clip_score.pose_score(self.get_text_feature(text), self.get_pose_feature(pose)).reshape(-1)[0])

    def sort_poses_by_score(self, text, poses):
        """best first; every pose is scored ONCE (the reference's list.sort key renders each pose once as well)"""
        scores = [self.calculate_pose_score(text, p) for p in poses]
        return [poses[i] for i in sorted(range(len(poses)), key=lambda i: -scores[i])]


_NO_RENDERER_GRADIENT = ("%s optimises the CLIP score with respect to the pose THROUGH the rasteriser: it needs neural_renderer's backward pass (a hand-designed "
                         "pseudo-gradient), which is not part of this repository (DESIGN.md section 8).  Available without it: VPoserCodebook, VPoserRealNVP, "
                         "MotionInterpolation, MotionOptimizer with clip_coef = 0 (the reference's motion_ablation/baseline conf)")


class PoseOptimizer:
    def __init__(self, ctx=None, **conf):
        raise NotImplementedError(_NO_RENDERER_GRADIENT % "PoseOptimizer (pose_generation.py:102-135)")


class VPoserOptimizer:
    def __init__(self, ctx=None, **conf):
        raise NotImplementedError(_NO_RENDERER_GRADIENT % "VPoserOptimizer (pose_generation.py:138-173)")


# ----------------------------------------------------------------------------------------------------------------- candidate poses
class VPoserCodebook:
    """pose_generation.py:288-329: the `pre_topk` codebook entries whose stored CLIP embedding is closest to the text, decoded by VPoser,
    near-duplicates (mean |difference| <= filter_threshold to an already kept pose) dropped, the first `topk` kept.  `codebook` [N,32] latent
    codes, `codebook_embedding` [N,512] (the reference's data/codebook.pth: pass its path as `codebook_path`, or the two tensors)."""

    def __init__(self, ctx, codebook=None, codebook_embedding=None, codebook_path="data/codebook.pth", topk=5, pre_topk=40, filter_threshold=0.07,
                 name="VPoserCodebook", smpl_path=None, vposer_path=None):       # (smpl_path / vposer_path: the reference's base-class keys; the assets are in `ctx`)
        self.ctx, self.name, self.topk, self.pre_topk, self.filter_threshold = ctx, name, int(topk), int(pre_topk), float(filter_threshold)
        if codebook is None:
            data = torch.load(codebook_path, map_location="cpu", weights_only=True)
            codebook, codebook_embedding = data["codebook"], data["codebook_embedding"]
        self.codebook = codebook.float().to(ctx.device)
        self.codebook_embedding = codebook_embedding.float().to(ctx.device)

    @staticmethod
    def suppress_duplicated_poses(poses, threshold):
        """greedy, in score order: keep a pose iff its mean absolute distance to EVERY kept pose exceeds the threshold"""
        kept = [0]
        for i in range(1, poses.shape[0]):
            d = (poses[i][None] - poses[kept]).abs().mean(-1)
            if float(min(d.min(), torch.tensor(10.0, device=d.device))) > threshold:
                kept.append(i)
        return poses[kept]

    @torch.no_grad()
    def get_topk_poses(self, text):
        tf = self.ctx.get_text_feature(text)
        score = F.cosine_similarity(self.codebook_embedding, tf[None]).reshape(-1)
        idx = torch.topk(score, self.pre_topk).indices
        poses = self.ctx.vp.decode(self.codebook[idx])["pose_body"].reshape(self.pre_topk, -1)
        return self.suppress_duplicated_poses(poses, self.filter_threshold)[: self.topk]


class VPoserRealNVP(nn.Module):
    """pose_generation.py:176-286: a conditional RealNVP over VPoser's 32-d latent, conditioned on the CLIP text feature; `num_batch` rounds of
    `num_sample` draws, each scored by rendering the decoded pose, the best kept; `topk` such poses, sorted.  Parameter names are the reference's
    (`s.{i}.{0,2,4}`, `t.{i}.{0,2,4}`, buffer `mask`), so data/pose_realnvp.pth's `state_dict` loads as it is."""

    def __init__(self, ctx, dim=32, hdim=256, num_block=8, num_sample=10, num_batch=50, ckpt_path=None, state_dict=None, topk=5, name="VPoserRealNVP",
                 smpl_path=None, vposer_path=None):
        super().__init__()
        self.ctx, self.name, self.topk = ctx, name, int(topk)
        self.dim, self.num_block, self.num_sample, self.num_batch = dim, num_block, num_sample, num_batch
        mask = (torch.randn(num_block, 1, dim) > 0).float()
        self.register_buffer("mask", mask)
        mlp = lambda last: nn.Sequential(nn.Linear(dim + 512, hdim), nn.LeakyReLU(), nn.Linear(hdim, hdim), nn.LeakyReLU(), nn.Linear(hdim, dim), *last)
        self.s, self.t = nn.ModuleList(), nn.ModuleList()
        for _ in range(num_block):            # (s_i then t_i, block by block: the reference's construction order, so that a seed reproduces its initialisation)
            self.s.append(mlp([nn.Tanh()]))
            self.t.append(mlp([]))
        if state_dict is None and ckpt_path is not None:
            state_dict = torch.load(ckpt_path, map_location="cpu", weights_only=True)["state_dict"]
        if state_dict is not None:
            self.load_state_dict(state_dict, strict=False)
        self.to(ctx.device).eval()

    def _coupling(self, x_kept, features, i):
        h = torch.cat([x_kept, features], dim=-1)
        free = 1 - self.mask[i]
        return self.s[i](h) * free, self.t[i](h) * free, free

    def decode(self, z, features):
        x = z
        for i in range(self.num_block):
            kept = x * self.mask[i]
            s, t, free = self._coupling(kept, features, i)
            x = kept + free * (x * torch.exp(s) + t)
        return x

    def encode(self, x, features):
        """inverse flow + log-determinant (training only in the reference)"""
        z, log_det = x, x.new_zeros(x.shape[0])
        for i in reversed(range(self.num_block)):
            kept = z * self.mask[i]
            s, t, free = self._coupling(kept, features, i)
            z = free * (z - t) * torch.exp(-s) + kept
            log_det = log_det - s.sum(dim=1)
        return z, log_det

    def sample(self, bs, features):
        z = torch.randn(bs, self.dim, device=self.mask.device)        # N(0, I): distributions.MultivariateNormal(0, I).sample of the reference
        return self.decode(z, features.reshape(1, -1).expand(bs, -1))

    @torch.no_grad()
    def get_pose(self, text_feature):
        tf = text_feature.reshape(1, -1)
        best, best_score = None, 0.0
        for _ in range(self.num_batch):
            poses = self.ctx.vp.decode(self.sample(self.num_sample, tf))["pose_body"].reshape(self.num_sample, -1)
            score = F.cosine_similarity(self.ctx.get_pose_feature(poses), tf)
            i = int(score.argmax())
            if float(score[i]) > best_score:
                best, best_score = poses[i], float(score[i])
        return best

    def get_topk_poses(self, text):
        tf = self.ctx.get_text_feature(text)
        poses = self.ctx.sort_poses_by_score(text, [self.get_pose(tf) for _ in range(self.topk)])
        return torch.stack(poses, dim=0)


# ----------------------------------------------------------------------------------------------------------------- motions
class MotionInterpolation:
    """motion_generation.py:100-137: the candidate poses encoded by VPoser, placed at the anchor frames, the latent code walked linearly between
    consecutive anchors, every frame decoded -> [num_frame, 69]"""

    def __init__(self, ctx, num_frame=60, anchor_position=(0, 14, 29, 44, 59), name="MotionInterpolation", smpl_path=None, vposer_path=None):
        self.ctx, self.name, self.num_frame, self.anchor_position = ctx, name, int(num_frame), tuple(int(a) for a in anchor_position)
        if self.anchor_position[0] != 0 or self.anchor_position[-1] != self.num_frame - 1:
            raise ValueError("the anchors start at frame 0 and end at the last frame")

    @torch.no_grad()
    def get_motion(self, text, poses):
        codes = self.ctx.vp.encode(poses[:, :63]).mean                        # [n_anchor, 32]
        z = codes.new_zeros(self.num_frame, codes.shape[-1])
        z[0] = codes[0]
        for k in range(1, len(self.anchor_position)):
            a, b = self.anchor_position[k - 1], self.anchor_position[k]
            step = (codes[k] - codes[k - 1]) / (b - a)
            for j in range(a, b):                                             # (accumulated step by step: the reference's fp32 summation order)
                z[j + 1] = z[j] + step
        return pose_padding(self.ctx.vp.decode(z)["pose_body"].reshape(self.num_frame, 63))


class SinusoidalPositionalEncoding(nn.Module):
    """buffer `pe` [max_len, 1, d] (motion_generation.py:140-157); the decoder uses its first `seq_len` rows as the queries"""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * freq), torch.cos(pos * freq)
        self.register_buffer("pe", pe[:, None, :])

    def forward(self, x):
        return self.dropout(x + self.pe[: x.shape[0]])


class MotionXTransformerEncoder(nn.Module):
    """ACTOR-style motion encoder (motion_generation.py:160-200); kept for checkpoint compatibility -- get_motion only decodes"""

    def __init__(self, seq_len=16, latent_dim=256, output_dim=256, num_heads=4, ff_size=1024, num_layers=8, activation="gelu", dropout=0.1):
        super().__init__()
        self.input_feats, self.seq_len, self.latent_dim = 55 * 6, seq_len, latent_dim
        self.skelEmbedding = nn.Linear(self.input_feats, latent_dim)
        self.pos_encoder = SinusoidalPositionalEncoding(latent_dim)
        self.query = nn.Parameter(torch.randn(1, latent_dim))
        layer = nn.TransformerEncoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size, dropout=dropout, activation=activation)
        self.seqTransEncoder = nn.TransformerEncoder(layer, num_layers=num_layers)
        self.final = nn.Linear(latent_dim, output_dim)

    def forward(self, motion):
        B, T = motion.shape[:2]
        tok = torch.cat([self.query.reshape(1, 1, -1).expand(B, 1, -1), self.skelEmbedding(motion.reshape(B, T, -1))], dim=1)
        return self.final(self.seqTransEncoder(self.pos_encoder(tok.permute(1, 0, 2).contiguous()))[0])


class MotionXTransformerDecoder(nn.Module):
    """latent [B, input_dim] -> 6-d rotations [B, seq_len, 55, 6] (motion_generation.py:203-246): the positional encodings of the frames are the
    queries, the latent code is the single memory token"""

    def __init__(self, seq_len=16, input_dim=256, latent_dim=256, num_heads=4, ff_size=1024, num_layers=8, activation="gelu", dropout=0.1):
        super().__init__()
        self.linear = nn.Linear(input_dim, latent_dim) if input_dim != latent_dim else nn.Identity()
        self.input_feats, self.seq_len, self.latent_dim = 55 * 6, seq_len, latent_dim
        self.pos_encoder = SinusoidalPositionalEncoding(latent_dim)
        layer = nn.TransformerDecoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size, dropout=dropout, activation=activation)
        self.seqTransDecoder = nn.TransformerDecoder(layer, num_layers=num_layers)
        self.final = nn.Linear(latent_dim, self.input_feats)

    def forward(self, latent):
        B, T = latent.shape[0], self.seq_len
        memory = self.linear(latent).reshape(1, B, -1)
        queries = self.pos_encoder.pe[:T].reshape(T, 1, -1).expand(T, B, -1)
        out = self.final(self.seqTransDecoder(tgt=queries, memory=memory))
        return out.permute(1, 0, 2).reshape(B, T, 55, 6)


class MotionOptimizer(nn.Module):
    """motion_generation.py:249-358: a latent code of the pretrained motion VAE optimised (Adam, 5 000 iterations) so that the decoded motion
    passes through the candidate poses IN ORDER (for candidate j the best-matching frame's 6-d rotation error, weighted recon_coef[j]) while the
    frame-to-frame change is REWARDED (- delta_coef x mse).  The reference adds clip_coef x a CLIP term through the rasteriser; here clip_coef
    must be 0 (see the module docstring).  Decoder / encoder parameter names are the reference's: data/motion_vae.pth's `state_dict` loads as it is."""

    def __init__(self, ctx, num_frame=60, latent_dim=256, num_layers=4, num_heads=4, ckpt_path=None, state_dict=None, optim_name="Adam", optim_cfg=None,
                 num_iteration=5000, recon_coef=(1, 0.8, 0.6, 0.4, 0.2), clip_coef=0.001, delta_coef=0.01, clip_num_part=30, name="MotionOptimizer",
                 smpl_path=None, vposer_path=None):
        super().__init__()
        if float(clip_coef) > 0:
            raise NotImplementedError(_NO_RENDERER_GRADIENT % ("MotionOptimizer with clip_coef = %g (motion_generation.py:333-345)" % clip_coef))
        self.ctx, self.name, self.num_frame, self.latent_dim = ctx, name, int(num_frame), int(latent_dim)
        kw = dict(seq_len=self.num_frame, latent_dim=latent_dim, num_heads=num_heads, ff_size=latent_dim * 4, num_layers=num_layers)
        self.encoder = MotionXTransformerEncoder(output_dim=latent_dim, **kw)
        self.decoder = MotionXTransformerDecoder(input_dim=latent_dim, **kw)
        if state_dict is None and ckpt_path is not None:
            state_dict = torch.load(ckpt_path, map_location="cpu", weights_only=True)["state_dict"]
        if state_dict is not None:
            self.load_state_dict(state_dict, strict=False)
        self.optim_name, self.optim_cfg, self.num_iteration = optim_name, dict(optim_cfg or {"lr": 0.01}), int(num_iteration)
        self.recon_coef, self.delta_coef = tuple(float(c) for c in recon_coef), float(delta_coef)
        if self.num_iteration < 1:
            raise ValueError("num_iteration must be at least 1")
        self.to(ctx.device).eval()

    def decode(self, latent_code):
        """latent -> [num_frame, 63] axis-angle body pose (joints 1..21 of the 55 the VAE was trained on)"""
        rot6d = self.decoder(latent_code.reshape(-1, self.latent_dim)).reshape(-1, 6)
        aa = matrix_to_axis_angle(rotation_6d_to_matrix(rot6d)).reshape(-1, 165)
        return aa[:, 3:66].contiguous()

    def losses(self, motion, poses):
        """(reconstruction, delta) of a decoded motion [T,63] against candidate poses [k,63]"""
        r6 = lambda p: matrix_to_rotation_6d(axis_angle_to_matrix(p.reshape(p.shape[:-1] + (21, 3))))
        err = ((r6(motion)[None] - r6(poses)[:, None]) ** 2).mean(-1).mean(-1)          # [k, T]
        recon = (err.min(dim=1).values * torch.as_tensor(self.recon_coef[: poses.shape[0]], device=err.device, dtype=err.dtype)).sum()
        delta = F.mse_loss(motion[:-1], motion[1:])
        return recon, delta

    def get_motion(self, text, poses):
        poses = poses[..., :63].contiguous().to(self.ctx.device)
        latent = nn.Parameter(torch.randn(self.latent_dim))               # drawn on the host, as the reference does
        opt = getattr(torch.optim, self.optim_name)([latent], **self.optim_cfg)
        motion = None
        for _ in range(self.num_iteration):
            motion = self.decode(latent.to(self.ctx.device))
            recon, delta = self.losses(motion, poses)
            loss = recon - (delta * self.delta_coef if self.delta_coef > 0 else 0.0)
            opt.zero_grad()
            loss.backward()
            opt.step()
        return pose_padding(motion.detach()).to(self.ctx.device)


# ----------------------------------------------------------------------------------------------------------------- builder.py / main.py
POSE_GENERATORS = {"PoseOptimizer": PoseOptimizer, "VPoserOptimizer": VPoserOptimizer, "VPoserRealNVP": VPoserRealNVP, "VPoserCodebook": VPoserCodebook}
MOTION_GENERATORS = {"MotionInterpolation": MotionInterpolation, "MotionOptimizer": MotionOptimizer}


def build_pose_generator(conf, ctx, **assets):
    conf = dict(conf)
    name = conf.pop("type")
    return POSE_GENERATORS[name](ctx, name=name, **conf, **assets)


def build_motion_generator(conf, ctx, **assets):
    conf = dict(conf)
    name = conf.pop("type")
    return MOTION_GENERATORS[name](ctx, name=name, **conf, **assets)


def run(conf, ctx, pose_assets=None, motion_assets=None):
    """main.py:14-41: candidate poses (saved as candidate_<i>.npy), then -- unless general.mode == 'pose' -- the motion (motion.npy).  The
    reference also writes pyrender previews (visualize.py); those are not part of the generators and are left to the caller."""
    out_dir = conf.get_string("general.base_exp_dir")
    os.makedirs(out_dir, exist_ok=True)
    text = conf.get_string("general.text")
    poses = build_pose_generator(dict(conf["pose_generator"]), ctx, **(pose_assets or {})).get_topk_poses(text)
    for i in range(poses.shape[0]):
        np.save(os.path.join(out_dir, "candidate_%d.npy" % i), poses[i].detach().cpu().numpy())
    if conf.get_string("general.mode") == "pose":
        return poses, None
    motion = build_motion_generator(dict(conf["motion_generator"]), ctx, **(motion_assets or {})).get_motion(text, poses=poses)
    np.save(os.path.join(out_dir, "motion.npy"), motion.detach().cpu().numpy())
    return poses, motion


def main(argv=None):
    """python -m avatarclip_amd.animate --conf confs/base.conf --clip_weights ViT-B-32.pt --bpe bpe_simple_vocab_16e6.txt.gz --smpl SMPL_NEUTRAL.pkl
    --vposer data/vposer [--codebook data/codebook.pth] [--realnvp data/pose_realnvp.pth] [--motion_vae data/motion_vae.pth]
    (AvatarAnimate/main.py with the assets its constructors load named on the command line; VPoser itself comes from the `human_body_prior` package)"""
    import argparse
    from . import clip_vit, smpl_lbs, tokenizer
    from .conf import ConfigFactory
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--conf", default="./confs/base.conf")
    ap.add_argument("--gpu", type=int, default=0)
    for name in ("clip_weights", "bpe", "smpl", "vposer"):
        ap.add_argument("--" + name, required=True)
    for name in ("codebook", "realnvp", "motion_vae"):
        ap.add_argument("--" + name, default=None)
    args = ap.parse_args(argv)
    dev = torch.device("cuda", args.gpu)
    try:
        from human_body_prior.models.vposer_model import VPoser
        from human_body_prior.tools.model_loader import load_model
    except ImportError as e:
        raise SystemExit("the VPoser body prior comes from the `human_body_prior` package (pose_generation.py:15-16), which is not installed: %s" % e)
    vp, _ = load_model(args.vposer, model_code=VPoser, remove_words_in_model_weights="vp_model.", disable_grad=True)
    perceptor = clip_vit.ClipVisionB32({k: v.float() for k, v in clip_vit.load_state_dict(args.clip_weights).items()}, dev)
    tk = tokenizer.SimpleTokenizer(args.bpe)
    text_feature = lambda text: perceptor.encode_text(tokenizer.tokenize([text], tk))[0]
    ctx = AnimateContext(perceptor, text_feature, smpl_lbs.load_smpl_arrays(args.smpl, dev), vp.to(dev).eval(), device=dev)
    with open(args.conf) as fh:
        conf = ConfigFactory.parse_string(fh.read())
    pose_assets = {"codebook_path": args.codebook} if args.codebook else ({"ckpt_path": args.realnvp} if args.realnvp else {})
    run(conf, ctx, pose_assets=pose_assets, motion_assets={"ckpt_path": args.motion_vae} if args.motion_vae else {})


if __name__ == "__main__":
    main()

