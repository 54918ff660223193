"""CPU oracle for the AvatarCLIP AppearanceGen NeuS hot path.

TEST INFRASTRUCTURE ONLY -- this is the checker, never the product.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it; avatarclip_amd/ never does.

It is a functional restatement (plain torch on CPU, fp32 or fp64, autograd for the gradients exactly like
the reference) of the reference algorithm.  Every function cites the reference lines it follows; paths are
relative to /root/reference/AvatarGen/AppearanceGen/.  Parity is PINNED: tests/test_oracle_golden.py checks
this file against fixtures under tests/golden/ that oracle/gen_golden.py produced by *running the reference's
own modules* (models/fields.py, models/renderer.py, models/embedder.py) with the shipped small checkpoint
and with seeded full-size networks.

Parameters are passed as state-dicts with the reference's key names (`lin0.weight_g`, `lin0.weight_v`,
`lin0.bias`, ..., `extra_lin.*`, `variance`) so the shipped checkpoint loads unmodified.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# models/embedder.py:6-51
# ----------------------------------------------------------------------------------------------
def embed(x: Tensor, multires: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]  (embedder.py:14-36)."""
    if multires <= 0:
        return x
    outs = [x]
    freqs = 2.0 ** torch.linspace(0.0, multires - 1, multires, dtype=x.dtype)  # embedder.py:21-22
    for f in freqs:
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def wn_weight(sd: Dict[str, Tensor], name: str) -> Tensor:
    """nn.utils.weight_norm(lin) with dim=0: W = g * v / ||v||_row (fields.py:65-66)."""
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    g = sd[name + ".weight_g"]
    v = sd[name + ".weight_v"]
    return g * v / v.norm(dim=1, keepdim=True)


def n_linears(sd: Dict[str, Tensor]) -> int:
    n = 0
    while ("lin%d.bias" % n) in sd:
        n += 1
    return n


def softplus100(x: Tensor) -> Tensor:
    return F.softplus(x, beta=100)  # nn.Softplus(beta=100), fields.py:68


# ----------------------------------------------------------------------------------------------
# models/fields.py:72-107  SDFNetwork
# ----------------------------------------------------------------------------------------------
def sdf_forward(sd: Dict[str, Tensor], x: Tensor, multires: int = 6, scale: float = 1.0) -> Tensor:
    """SDFNetwork.forward (fields.py:72-88).  The skip layer is the last linear (skip_in = [n_lin-1])."""
    inputs = embed(x * scale, multires)
    nl = n_linears(sd)
    skip = nl - 1  # fields.py:36-39 with skip_in=[n_layers] in every shipped conf
    h = inputs
    for l in range(nl):
        if l == skip:
            h = torch.cat([h, inputs], 1) / math.sqrt(2)  # fields.py:81-82
        h = F.linear(h, wn_weight(sd, "lin%d" % l), sd["lin%d.bias" % l])
        if l < nl - 1:
            h = softplus100(h)  # fields.py:86-87
    return torch.cat([h[:, :1] / scale, h[:, 1:]], dim=-1)  # fields.py:88


def sdf_gradient(sd, x: Tensor, multires: int = 6, create_graph: bool = True) -> Tensor:
    """SDFNetwork.gradient (fields.py:96-107): d sdf / d x via autograd, graph kept."""
    x = x.detach().requires_grad_(True)
    y = sdf_forward(sd, x, multires)[:, :1]
    (g,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=create_graph, retain_graph=True)
    return g


# ----------------------------------------------------------------------------------------------
# models/fields.py:154-185  RenderingNetwork (mode == 'no_view_dir', squeeze_out, extra_color)
# ----------------------------------------------------------------------------------------------
def color_forward(sd: Dict[str, Tensor], points: Tensor, normals: Tensor, feature: Tensor,
                  extra_color: bool = True) -> Tensor:
    x = torch.cat([points, normals, feature], dim=-1)  # fields.py:164-165
    nl = n_linears(sd)
    extra = None
    for l in range(nl):
        x = F.linear(x, wn_weight(sd, "lin%d" % l), sd["lin%d.bias" % l])
        if l < nl - 1:
            x = F.relu(x)
        if extra_color and l == nl - 2:  # l == num_layers-3 with num_layers = nl+1 (fields.py:177-178)
            extra = F.linear(x, wn_weight(sd, "extra_lin"), sd["extra_lin.bias"])
    if extra_color:
        x = torch.cat([x, extra], -1)
    return torch.sigmoid(x)  # fields.py:183-184


def inv_s_from_variance(variance: Tensor) -> Tensor:
    """SingleVarianceNetwork.forward + clip (fields.py:275-276; renderer.py:234)."""
    return torch.exp(variance * 10.0).clip(1e-6, 1e6)


# ----------------------------------------------------------------------------------------------
# models/renderer.py:39-69  sample_pdf (det=True path is the only one on the hot path)
# ----------------------------------------------------------------------------------------------
def sample_pdf(bins: Tensor, weights: Tensor, n_samples: int, u: Optional[Tensor] = None) -> Tensor:
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:  # det=True (renderer.py:48-50)
        u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, dtype=bins.dtype)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b = torch.gather(cdf, 1, below)
    cdf_a = torch.gather(cdf, 1, above)
    bins_b = torch.gather(bins, 1, below)
    bins_a = torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bins_b + t * (bins_a - bins_b)


# renderer.py:133-177
def up_sample(rays_o, rays_d, z_vals, sdf, n_importance: int, inv_s: float) -> Tensor:
    R, n = z_vals.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
    radius = torch.linalg.norm(pts, ord=2, dim=-1)
    inside_sphere = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
    sdf = sdf.reshape(R, n)
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    prev_z, next_z = z_vals[:, :-1], z_vals[:, 1:]
    mid_sdf = (prev_sdf + next_sdf) * 0.5
    cos_val = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-5)
    prev_cos = torch.cat([torch.zeros([R, 1], dtype=z_vals.dtype), cos_val[:, :-1]], dim=-1)
    cos_val = torch.minimum(prev_cos, cos_val)
    cos_val = cos_val.clip(-1e3, 0.0) * inside_sphere
    dist = next_z - prev_z
    prev_esti = mid_sdf - cos_val * dist * 0.5
    next_esti = mid_sdf + cos_val * dist * 0.5
    prev_cdf = torch.sigmoid(prev_esti * inv_s)
    next_cdf = torch.sigmoid(next_esti * inv_s)
    alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
    weights = alpha * torch.cumprod(
        torch.cat([torch.ones([R, 1], dtype=z_vals.dtype), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    return sample_pdf(z_vals, weights, n_importance).detach()


# renderer.py:179-193
def cat_z_vals(sd_sdf, rays_o, rays_d, z_vals, new_z_vals, sdf, last: bool, multires: int = 6):
    R, n = z_vals.shape
    m = new_z_vals.shape[1]
    pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z_vals[..., :, None]
    z_all = torch.cat([z_vals, new_z_vals], dim=-1)
    z_sorted, index = torch.sort(z_all, dim=-1, stable=True)
    if not last:
        new_sdf = sdf_forward(sd_sdf, pts.reshape(-1, 3), multires)[:, :1].reshape(R, m)
        sdf = torch.cat([sdf, new_sdf], dim=-1)
        sdf = torch.gather(sdf, 1, index)
    return z_sorted, sdf


def coarse_z_vals(near, far, n_samples: int, jitter: Optional[Tensor]) -> Tensor:
    """renderer.py:304-319.  `jitter` is the injected torch.rand([R,1]) (None == perturb 0)."""
    z = torch.linspace(0.0, 1.0, n_samples, dtype=near.dtype)
    z = near + (far - near) * z[None, :]
    if jitter is not None:
        z = z + (jitter - 0.5) * 2.0 / n_samples
    return z


def hierarchical_z_vals(sd_sdf, rays_o, rays_d, near, far, n_samples, n_importance, up_sample_steps,
                        jitter=None, multires=6, return_steps=False):
    """The no_grad up-sampling block of NeuSRenderer.render (renderer.py:334-352)."""
    steps = []
    with torch.no_grad():
        z_vals = coarse_z_vals(near, far, n_samples, jitter)
        if n_importance > 0:
            R = rays_o.shape[0]
            pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
            sdf = sdf_forward(sd_sdf, pts.reshape(-1, 3), multires)[:, :1].reshape(R, n_samples)
            for i in range(up_sample_steps):
                new_z = up_sample(rays_o, rays_d, z_vals, sdf, n_importance // up_sample_steps, 64 * 2 ** i)
                steps.append(dict(z_in=z_vals, sdf_in=sdf, new_z=new_z))
                z_vals, sdf = cat_z_vals(sd_sdf, rays_o, rays_d, z_vals, new_z, sdf,
                                         last=(i + 1 == up_sample_steps), multires=multires)
    if return_steps:
        return z_vals, steps
    return z_vals


# renderer.py:195-300
def render_core(sd_sdf, sd_color, variance, rays_o, rays_d, z_vals, sample_dist, background_rgb=None,
                cos_anneal_ratio=0.0, extra_color=True, multires=6):
    R, S = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], sample_dist)], -1)
    mid_z_vals = z_vals + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z_vals[..., :, None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(R, S, 3).reshape(-1, 3)

    # reference: one forward for (sdf, feature) and a second one inside .gradient (renderer.py:221-225);
    # mathematically one forward with x.requires_grad is identical.
    pts_g = pts.detach().requires_grad_(True)
    out = sdf_forward(sd_sdf, pts_g, multires)
    sdf = out[:, :1]
    feature = out[:, 1:]
    (gradients,) = torch.autograd.grad(sdf, pts_g, torch.ones_like(sdf), create_graph=True, retain_graph=True)

    raw = color_forward(sd_color, pts_g.detach(), gradients, feature, extra_color)
    if extra_color:
        raw = raw.reshape(R, S, 6)
        sampled_color, extra_sampled = raw[..., :3], raw[..., 3:]
    else:
        sampled_color, extra_sampled = raw.reshape(R, S, 3), None

    inv_s = inv_s_from_variance(variance).reshape(1, 1)  # renderer.py:234
    true_cos = (dirs * gradients).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) +
                 F.relu(-true_cos) * cos_anneal_ratio)  # renderer.py:241-242
    est_next = sdf + iter_cos * dists.reshape(-1, 1) * 0.5
    est_prev = sdf - iter_cos * dists.reshape(-1, 1) * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    p = prev_cdf - next_cdf
    c = prev_cdf
    alpha = ((p + 1e-5) / (c + 1e-5)).reshape(R, S).clip(0.0, 1.0)  # renderer.py:254

    pts_norm = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).reshape(R, S)
    inside_sphere = (pts_norm < 1.0).to(z_vals.dtype).detach()
    relax_inside_sphere = (pts_norm < 1.2).to(z_vals.dtype).detach()

    weights = alpha * torch.cumprod(
        torch.cat([torch.ones([R, 1], dtype=z_vals.dtype), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    weights_sum = weights.sum(dim=-1, keepdim=True)
    color = (sampled_color * weights[:, :, None]).sum(dim=1)
    extra = (extra_sampled * weights[:, :, None]).sum(dim=1) if extra_color else None
    if background_rgb is not None:  # renderer.py:277-281
        if extra_color:
            extra = extra + background_rgb * (1.0 - weights_sum)
        else:
            color = color + background_rgb * (1.0 - weights_sum)

    grad_norm = torch.linalg.norm(gradients.reshape(R, S, 3), ord=2, dim=-1)
    gradient_error = (grad_norm - 1.0) ** 2
    gradient_error = (relax_inside_sphere * gradient_error).sum() / (relax_inside_sphere.sum() + 1e-5)

    return dict(color=color, extra_color=extra, sdf=sdf, dists=dists, gradients=gradients.reshape(R, S, 3),
                s_val=1.0 / inv_s, mid_z_vals=mid_z_vals, weights=weights, cdf=c.reshape(R, S),
                gradient_error=gradient_error, inside_sphere=inside_sphere,
                sampled_color=raw, alpha=alpha)


# renderer.py:302-397
def render(sd_sdf, sd_color, variance, rays_o, rays_d, near, far, n_samples=32, n_importance=32,
           up_sample_steps=4, jitter=None, background_rgb=None, cos_anneal_ratio=0.0, extra_color=True,
           multires=6, z_vals=None):
    sample_dist = 2.0 / n_samples
    if z_vals is None:
        z_vals = hierarchical_z_vals(sd_sdf, rays_o, rays_d, near, far, n_samples, n_importance,
                                     up_sample_steps, jitter, multires)
    ret = render_core(sd_sdf, sd_color, variance, rays_o, rays_d, z_vals, sample_dist, background_rgb,
                      cos_anneal_ratio, extra_color, multires)
    weights = ret["weights"]
    R, S = weights.shape
    return {
        "color_fine": ret["color"],
        "extra_color_fine": ret["extra_color"],
        "s_val": ret["s_val"].expand(R * S, 1).reshape(R, S).mean(dim=-1, keepdim=True),
        "cdf_fine": ret["cdf"],
        "weight_sum": weights.sum(dim=-1, keepdim=True),
        "weight_max": torch.max(weights, dim=-1, keepdim=True)[0],
        "gradients": ret["gradients"],
        "weights": weights,
        "mid_z_vals": ret["mid_z_vals"],
        "gradient_error": ret["gradient_error"],
        "inside_sphere": ret["inside_sphere"],
        "z_vals": z_vals,
        "sdf": ret["sdf"],
        "sampled_color": ret["sampled_color"],
    }


# ----------------------------------------------------------------------------------------------
# models/dataset.py:277-342 ray generation; models/utils.py:9-70 cameras
# ----------------------------------------------------------------------------------------------
def gen_rays_pose(pose: Tensor, H: int, W: int, focal: float, resolution_level: float = 1):
    """dataset.py:277-293 (K = [[f,0,W/2],[0,f,H/2]] from dataset.py:243-247)."""
    l = resolution_level
    tx = torch.linspace(0, W - 1, int(W // l))
    ty = torch.linspace(0, H - 1, int(H // l))
    px, py = torch.meshgrid(tx, ty, indexing="ij")
    px, py = px.t(), py.t()
    p = torch.stack([(px - 0.5 * W) / focal, -(py - 0.5 * H) / focal, -torch.ones_like(px)], -1).float()
    v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
    v = torch.sum(v[..., None, :] * pose[:3, :3], -1)
    o = pose[None, None, :3, 3].expand(v.shape)
    return o, v


def near_far_from_sphere(rays_o, rays_d):
    """dataset.py:331-342."""
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    near = (mid - 1).clamp(min=0)
    far = mid + 1
    return near, far


def lookat(eye, at, up):
    """utils.py:9-27 -- returns the camera-to-world matrix [x y z eye]."""
    def nrm(a):
        return a / np.linalg.norm(a)
    z = nrm(eye - at)
    x = nrm(np.cross(up, z))
    y = np.cross(z, x)
    return np.array([[x[0], y[0], z[0], eye[0]],
                     [x[1], y[1], z[1], eye[1]],
                     [x[2], y[2], z[2], eye[2]],
                     [0, 0, 0, 1]])


def sphere_coord(theta, phi, r=1.0):
    """utils.py:59-64."""
    return np.array([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)])


# ----------------------------------------------------------------------------------------------
# main.py:426-453 shading glue and main.py:489-534 loss assembly (train_clip), main.py:214-224 (train)
# ----------------------------------------------------------------------------------------------
def cast_light(render_out, light_dir: np.ndarray, ambience: float):
    """main.py:426-453.  light_dir / ambience are the numpy-RNG draws, injected."""
    normals = (render_out["gradients"] * render_out["weights"][:, :, None]).sum(dim=1)
    normals = normals / (torch.norm(normals, dim=-1, keepdim=True) + 1e-7)
    ld = torch.from_numpy(np.asarray(light_dir)).to(normals.dtype)
    rand_light_d = torch.zeros_like(normals) + ld
    rand_light_d = rand_light_d / (torch.norm(rand_light_d, dim=-1, keepdim=True) + 1e-7)
    diffuse_shading = (normals * rand_light_d).sum(-1, keepdim=True).clamp(min=0, max=1)
    diffuse_shading = torch.where(torch.isnan(diffuse_shading), torch.ones_like(diffuse_shading), diffuse_shading)
    rand_shading = ambience + (1 - ambience) * diffuse_shading
    extra = render_out["extra_color_fine"]
    weight_sum = render_out["weight_sum"].reshape(-1)
    bgmask = (weight_sum < 0.5)[:, None]
    rand_shading_rgb = torch.where(bgmask, extra, rand_shading.repeat(1, 3))
    rand_shading = torch.where(bgmask, torch.ones_like(rand_shading), rand_shading)
    texture_shading = (extra * rand_shading).clamp(min=0, max=1)
    return texture_shading, rand_shading_rgb


def scatter_to_image(vals, dilated_mask, background):
    """main.py:461-487: rays rendered only inside the dilated silhouette go back to their pixels of a full [H,W,C] image
    (`background` = the augmentation background for the CLIP images, zeros for colour / weight_sum)."""
    full = background.clone()
    full[dilated_mask] = vals
    return full.reshape(-1, vals.shape[-1])


def silhouette_background(H, W, choice_i, background_rgb, dilated_mask):
    """main.py:462-466: the full-image background the masked renders are scattered onto."""
    background = torch.zeros([H, W, 3])
    if choice_i == 0:
        background[:] = 1
    if choice_i in (1, 2):
        background[~dilated_mask] = background_rgb.reshape(H, W, 1).repeat(1, 1, 3)[~dilated_mask]
    return background


def random_resized_crop_params(height, width, rng, scale=(1.0, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision.transforms.RandomResizedCrop.get_params restated (third-party, absent offline; main.py:261 builds it with
    scale=(1,1)): ten tries of area = H W U(scale), log-uniform aspect ratio, accept when the box fits; else the central crop
    at the clamped ratio.  With scale=(1,1) on a square image every accepted box and the fallback are the FULL frame
    (tests/test_glue_golden.py checks it over many draws), so the transform reduces to the resize to 224."""
    area = height * width
    log_ratio = (np.log(ratio[0]), np.log(ratio[1]))
    for _ in range(10):
        target_area = area * rng.uniform(scale[0], scale[1])
        aspect_ratio = np.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(np.sqrt(target_area * aspect_ratio)))
        h = int(round(np.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = int(rng.randint(0, height - h + 1))
            j = int(rng.randint(0, width - w + 1))
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def neus_losses(render_out, true_rgb, mask, igr_weight, mask_weight):
    """main.py:214-224 / 489-497: masked L1 + eikonal + mask BCE."""
    mask_sum = mask.sum() + 1e-5
    color_error = (render_out["color_fine"] - true_rgb) * mask
    color_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
    eik = render_out["gradient_error"]
    mask_loss = F.binary_cross_entropy(render_out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    return color_loss + eik * igr_weight + mask_loss * mask_weight, color_loss, eik, mask_loss


def clip_preprocess(img_hw3: Tensor) -> Tensor:
    """main.py:261-267,510-511: RandomResizedCrop(224, scale=(1,1)) on a square image == bilinear resize
    (align_corners=False, no antialias); RandomPerspective(p=0) == identity; Normalize."""
    x = img_hw3.permute(2, 0, 1).unsqueeze(0)
    if x.shape[-1] != 224 or x.shape[-2] != 224:
        x = F.interpolate(x, size=(224, 224), mode="bilinear", align_corners=False)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def clip_cosine(encoded_renders: Tensor, text: Tensor) -> Tensor:
    """main.py:513-514."""
    return torch.cosine_similarity(torch.mean(encoded_renders, dim=0), torch.mean(text, dim=0), dim=0)


def to_dtype(sd: Dict[str, Tensor], dtype) -> Dict[str, Tensor]:
    return {k: v.detach().to(dtype) for k, v in sd.items()}
