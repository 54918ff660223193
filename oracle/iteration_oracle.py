"""One full iteration of Runner.train_clip (AvatarGen/AppearanceGen/main.py:345-566) assembled from the CPU oracle ONLY.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing here imports
avatarclip_amd's Runner / renderer / engine: the product's glue is therefore compared with an independent restatement, not
with itself.  Every stage cites the lines it follows; every random draw of the reference loop is an INPUT (`draws`), so the
product path and the oracle consume identical cameras, jitter, backgrounds, lights and ambience.

    state  = OracleState(sdf_params, color_params, variance, lr0, alpha, warm_up_end, end_iter)   # tensors + Adam
    stats  = train_clip_iteration(state, conf, draws, clip_sd, texts)

`conf` carries the reference's conf keys that the loop reads (main.py:50-127).
"""
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import clip_vit_oracle as C
from . import neus_oracle as O


def learning_rate_factor(iter_step, warm_up_end, end_iter, alpha):
    """main.py:577-586."""
    if iter_step < warm_up_end:
        return iter_step / warm_up_end
    progress = (iter_step - warm_up_end) / (end_iter - warm_up_end)
    return (np.cos(np.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


@dataclass
class OracleState:
    sdf: Dict[str, torch.Tensor]          # 'linK.weight_g' / 'weight_v' / 'bias' (requires_grad leaves)
    color: Dict[str, torch.Tensor]
    variance: torch.Tensor                # scalar leaf (fields.py:271-276)
    lr0: float = 5e-4
    alpha: float = 0.05
    warm_up_end: float = 0.0
    end_iter: int = 100000
    iter_step: int = 0
    opt: Optional[torch.optim.Optimizer] = field(default=None)

    def params(self):
        return list(self.sdf.values()) + [self.variance] + list(self.color.values())   # main.py:139-145 order

    def __post_init__(self):
        if self.opt is None:
            self.opt = torch.optim.Adam(self.params(), lr=self.lr0)                    # main.py:145
        self.update_learning_rate()

    def update_learning_rate(self):
        for g in self.opt.param_groups:
            g["lr"] = self.lr0 * learning_rate_factor(self.iter_step, self.warm_up_end, self.end_iter, self.alpha)


@dataclass
class Draws:
    """the random inputs of one iteration (main.py:348-359, 389-405, 318 of renderer.py, 433-440)"""
    eye: np.ndarray
    at: np.ndarray
    theta: float
    phi: float
    is_front: int
    prior_rgb: torch.Tensor               # [Hp,Wp,3] = render_one_batch(v, f, eye, at) (models/utils.py:108-125)
    jitter: Optional[torch.Tensor]        # [R,1] uniform (renderer.py:317-319)
    choice_i: int = 3                     # background augmentation branch (main.py:389-405)
    background_rgb: Optional[torch.Tensor] = None    # [1,3] | [H*W,1]
    light_dir: Optional[np.ndarray] = None
    ambience: float = 0.0
    dilated_mask: Optional[torch.Tensor] = None      # silhouette mode: [H,W] bool, with rays_o / rays_d [R,3]
    rays_o: Optional[torch.Tensor] = None
    rays_d: Optional[torch.Tensor] = None
    W: int = 0


def train_clip_iteration(st: OracleState, conf: dict, dr: Draws, clip_sd, texts, iter_i: int):
    """texts = dict(prompt=, face_prompt=, back_prompt=) of [1,512] embeddings.  Returns dict(loss, parts..., grads)."""
    H_img, W_img, focal = conf["H"], conf["W"], conf["focal"]
    n_samples, n_importance, up_steps = conf["n_samples"], conf["n_importance"], conf["up_sample_steps"]
    pose = torch.from_numpy(O.lookat(np.asarray(dr.eye, np.float64), np.asarray(dr.at, np.float64), np.array([0., 1, 0]))).float()
    if conf.get("use_silhouettes", False):       # main.py:365-369 (rays come with the draws: dataset.py:252-275 is pinned on its own)
        rays_o, rays_d, H, W = dr.rays_o, dr.rays_d, dr.W, dr.W
    else:                                        # main.py:370-374
        o, v = O.gen_rays_pose(pose, H_img, W_img, focal, conf.get("full_frame_resolution_level", 2.25))
        H, W = o.shape[0], o.shape[1]
        rays_o, rays_d = o.reshape(H * W, 3).float().contiguous(), v.reshape(H * W, 3).float().contiguous()
    Hp, Wp = dr.prior_rgb.shape[0], dr.prior_rgb.shape[1]
    true_rgb = F.interpolate(dr.prior_rgb.reshape(Hp, Wp, 3).permute(2, 0, 1).unsqueeze(0).float(), size=(H, W)) \
        .squeeze(0).permute(1, 2, 0).reshape(-1, 3)                                     # main.py:376-377
    mask = torch.zeros_like(true_rgb)
    mask[true_rgb != 0] = 1
    mask = mask[..., :1]
    near, far = O.near_far_from_sphere(rays_o, rays_d)                                   # main.py:382-385
    mask = (mask > 0.5).float() if conf["mask_weight"] > 0.0 else torch.ones_like(mask)  # main.py:407-410
    background_rgb = dr.background_rgb
    if conf.get("use_silhouettes", False) and dr.choice_i in (1, 2):                     # main.py:412-415
        masked_bg = background_rgb.reshape(H, W, 1)[dr.dilated_mask].reshape(-1, 1)
    else:
        masked_bg = background_rgb
    out = O.render(st.sdf, st.color, st.variance, rays_o, rays_d, near, far, n_samples, n_importance, up_steps, dr.jitter,
                   masked_bg, conf.get("cos_anneal_ratio", 1.0), extra_color=conf.get("extra_color", True))
    color_fine, extra, wsum = out["color_fine"], out["extra_color_fine"], out["weight_sum"]
    tex = shade = None
    if conf["add_no_texture"] or conf["texture_cast_light"]:                             # main.py:426-453
        tex, shade = O.cast_light(out, dr.light_dir, dr.ambience)
    if conf.get("use_silhouettes", False):                                               # main.py:461-487
        background = O.silhouette_background(H, W, dr.choice_i, background_rgb, dr.dilated_mask)
        if tex is not None:
            tex, shade = O.scatter_to_image(tex, dr.dilated_mask, background), O.scatter_to_image(shade, dr.dilated_mask, background)
        extra = O.scatter_to_image(extra, dr.dilated_mask, background)
        color_fine = O.scatter_to_image(color_fine, dr.dilated_mask, torch.zeros(H, W, 3))
        wsum = O.scatter_to_image(wsum, dr.dilated_mask, torch.zeros(H, W, 1))
    full = dict(out, color_fine=color_fine, weight_sum=wsum)
    loss, closs, eik, mloss = O.neus_losses(full, true_rgb, mask, conf["igr_weight"], conf["mask_weight"])   # main.py:489-497
    if conf["use_face_prompt"] and iter_i % 4 == 0:                                      # main.py:499-507
        text = texts["face_prompt"]
    elif conf["use_back_prompt"] and dr.is_front == 0:
        text = texts["back_prompt"]
    else:
        text = texts["prompt"]
    img = tex if conf["texture_cast_light"] else extra                                   # main.py:509-520
    enc = C.encode_image(clip_sd, O.clip_preprocess(img.reshape(H, W, 3)))
    cosine = O.clip_cosine(enc, text)
    loss = loss + (1.0 - cosine) * conf["clip_weight"]
    cosine_shading = None
    if conf["add_no_texture"]:                                                           # main.py:521-534
        enc2 = C.encode_image(clip_sd, O.clip_preprocess(shade.reshape(H, W, 3)))
        cosine_shading = O.clip_cosine(enc2, text)
        loss = loss + (1.0 - cosine_shading) * conf["clip_weight"]
    st.opt.zero_grad()                                                                   # main.py:536-538
    loss.backward()
    grads = [None if p.grad is None else p.grad.detach().clone() for p in st.params()]
    st.opt.step()
    st.iter_step += 1
    st.update_learning_rate()                                                            # main.py:563
    return dict(loss=loss.detach(), color=closs.detach(), eikonal=eik.detach(), mask=mloss.detach(), cosine=cosine.detach(),
                cosine_shading=None if cosine_shading is None else cosine_shading.detach(), grads=grads, H=H, W=W,
                color_fine=color_fine.detach(), extra_color_fine=extra.detach(), weight_sum=wsum.detach(),
                image=img.detach())
