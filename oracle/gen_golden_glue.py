"""Golden vectors for the shading / scatter / loss glue of Runner.train_clip (AvatarGen/AppearanceGen/main.py:387-534),
produced by EXECUTING THE REFERENCE'S OWN SOURCE LINES.  Run in the build container only (needs /root/reference):

    python oracle/gen_golden_glue.py   ->   tests/golden/glue.npz

TEST INFRASTRUCTURE ONLY.  main.py cannot be imported here (cv2, pyhocon, tensorboard, clip, smplx, neural_renderer are
absent), and the glue is inline in the loop body of `train_clip`, so the statement range from `background_rgb = None`
(main.py:387) up to `self.optimizer.zero_grad()` (main.py:536) is cut out of the source TEXT, dedented and exec'd once per
case in a namespace that supplies exactly the names those lines read:

  * from the reference itself: `sphere_coord` (models/utils.py:59-64, extracted with `ast`);
  * injected inputs: `self` (flags, weights, a `renderer.render` that returns the fixture's render_out, the prompt
    embeddings), the view (`rays_o`, `H`, `W`, `true_rgb`, `mask`, `dilated_mask`, `theta`, `phi`, `is_front`, `iter_i`);
  * stand-ins for what lives in absent third-party packages, each stated in the fixture:
      - `self.resize` = torchvision RandomResizedCrop(224, scale=(1,1)) -> bilinear resize (see
        oracle/neus_oracle.random_resized_crop_params for why the crop is always the full frame),
      - `self.random_perspective` (p=0) and `.cuda()` -> identity, `self.clip_normalizer` -> the Normalize arithmetic,
      - `self.perceptor.encode_image` -> a FIXED random linear map [3*224*224] -> [512] (the glue, not the ViT, is under test),
      - `transforms.GaussianBlur` -> torchvision's separable reflect-padded blur restated (only used by case `chess`).
Outputs recorded per case: texture_shading, rand_shading_rgb, extra_color_fine, color_fine, weight_sum (after the scatter),
color_fine_loss, eikonal_loss, mask_loss, cosine, cosine_shading, loss, and the numpy / torch RNG draws the lines consumed.
"""
import os
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gen_golden import extract_functions, GOLD  # noqa: E402

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def glue_source(main_py):
    src = open(main_py).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.strip() == "background_rgb = None" and "use_bg_aug" in src[i + 1])
    end = next(i for i in range(start, len(src)) if src[i].strip() == "self.optimizer.zero_grad()")
    return textwrap.dedent("\n".join(src[start:end])), (start + 1, end)


class _Blur:
    """torchvision.transforms.GaussianBlur(kernel_size=(kx, ky), sigma=(lo, hi)) restated (v0.8-0.15 semantics): sigma drawn
    with torch.empty(1).uniform_, 1-D kernels exp(-x^2 / 2 sigma^2) normalised, reflect padding, depthwise conv."""

    def __init__(self, kernel_size, sigma):
        self.k, self.sigma = kernel_size, sigma
        self.last_sigma = None

    def __call__(self, img):
        sigma = torch.empty(1).uniform_(self.sigma[0], self.sigma[1]).item()
        self.last_sigma = sigma

        def k1d(k):
            lim = (k - 1) * 0.5
            x = torch.linspace(-lim, lim, steps=k)
            pdf = torch.exp(-0.5 * (x / sigma).pow(2))
            return pdf / pdf.sum()
        kx, ky = k1d(self.k[0]), k1d(self.k[1])
        k2d = torch.mm(ky[:, None], kx[None, :])
        C = img.shape[-3]
        k2d = k2d.expand(C, 1, k2d.shape[0], k2d.shape[1])
        pad = [self.k[0] // 2, self.k[0] // 2, self.k[1] // 2, self.k[1] // 2]
        x = F.pad(img, pad, mode="reflect")
        return F.conv2d(x, k2d, groups=C)


def make_case(name, seed, R_side, S, silhouettes, choice, flags):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, generator=g)
    H = W = R_side
    if silhouettes:
        yy, xx = np.mgrid[0:H, 0:W]
        dil = torch.from_numpy((((xx - W / 2) / (0.33 * W)) ** 2 + ((yy - H / 2) / (0.45 * H)) ** 2) < 1)
        R = int(dil.sum())
    else:
        dil, R = None, H * W
    weights = rnd(R, S) ** 4
    weights = weights / weights.sum(-1, keepdim=True) * (rnd(R, 1) * 1.2).clamp(max=1.0)   # weight_sum on both sides of 0.5
    weights[: R // 7] = 0.0                                                                # rays that miss: zero normals -> NaN-free path
    grads = torch.randn(R, S, 3, generator=g)
    out = dict(color_fine=rnd(R, 3), extra_color_fine=rnd(R, 3) * 1.1 - 0.05, gradients=grads, weights=weights,
               weight_sum=weights.sum(-1, keepdim=True), weight_max=weights.max(-1, keepdim=True)[0],
               s_val=torch.full((R, 1), 0.05), cdf_fine=rnd(R, S), gradient_error=rnd(()) * 0.3)
    true_rgb = rnd(H * W, 3) * (rnd(H * W, 1) > 0.4).float()
    return dict(name=name, seed=seed, H=H, W=W, S=S, R=R, silhouettes=silhouettes, choice=choice, flags=flags,
                dilated_mask=dil, render_out=out, true_rgb=true_rgb, theta=float(rnd(()) * 1.0 - 0.5), phi=float(rnd(()) * 6.28),
                is_front=int(flags.get("is_front", 1)), iter_i=int(flags.get("iter_i", 1)))


def run_case(code, sphere_coord, case, enc_w, texts):
    fl = case["flags"]
    H, W = case["H"], case["W"]
    blur_holder = {}

    class _T:
        @staticmethod
        def GaussianBlur(kernel_size, sigma):
            b = _Blur(kernel_size, sigma)
            blur_holder["b"] = b
            return b

    def resize(x):
        return x if x.shape[-1] == 224 and x.shape[-2] == 224 else F.interpolate(x, size=(224, 224), mode="bilinear", align_corners=False)

    def normalizer(x):
        m = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
        s = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
        return (x - m) / s

    render_out = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in case["render_out"].items()}
    seen = {}

    def render(rays_o, rays_d, near, far, background_rgb=None, cos_anneal_ratio=0.0):
        seen["background_rgb"] = None if background_rgb is None else background_rgb.clone()
        return render_out

    self = types.SimpleNamespace(
        use_bg_aug=True, mask_weight=fl.get("mask_weight", 1.0), use_silhouettes=case["silhouettes"],
        add_no_texture=fl.get("add_no_texture", True), texture_cast_light=fl.get("texture_cast_light", True),
        use_face_prompt=fl.get("use_face_prompt", True), use_back_prompt=fl.get("use_back_prompt", True),
        igr_weight=0.1, clip_weight=1.0, renderer=types.SimpleNamespace(n_samples=case["S"] // 2, n_importance=case["S"] // 2, render=render),
        get_cos_anneal_ratio=lambda: 1.0, resize=resize, random_perspective=lambda x: x, clip_normalizer=normalizer,
        perceptor=types.SimpleNamespace(encode_image=lambda x: x.reshape(x.shape[0], -1) @ enc_w),
        encoded_text=texts[0], encoded_face_text=texts[1], encoded_back_text=texts[2])
    true_rgb = case["true_rgb"].clone()
    mask = torch.zeros_like(true_rgb)
    mask[true_rgb != 0] = 1
    mask = mask[..., :1]
    ns = dict(np=np, torch=torch, F=F, transforms=_T, sphere_coord=sphere_coord, self=self, H=H, W=W, iter_i=case["iter_i"],
              theta=case["theta"], phi=case["phi"], is_front=case["is_front"], true_rgb=true_rgb, mask=mask,
              dilated_mask=case["dilated_mask"], rays_o=torch.zeros(case["R"], 3), rays_d=torch.zeros(case["R"], 3),
              near=None, far=None)
    # the lines draw choice_i with np.random.choice(4): seed numpy so that the wanted branch comes up, and record the stream
    seed = next(s for s in range(10000) if np.random.RandomState(s).choice(4) == case["choice"])
    np.random.seed(seed)
    torch.manual_seed(case["seed"])
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda t, *a, **k: t           # `.cuda()` calls inside the lines (CPU-only container)
    try:
        exec(compile(code, "main.py[glue]", "exec"), ns)
    finally:
        torch.Tensor.cuda = saved_cuda
    # re-derive the numpy draws the lines consumed (same seed, same order: choice, [chess length], 2 light angles, ambience)
    rs = np.random.RandomState(seed)
    rs.choice(4)
    chess_n = int(rs.choice(np.arange(10, 20))) if case["choice"] == 2 else -1
    light = None
    ambience = -1.0
    if self.add_no_texture or self.texture_cast_light:
        light = sphere_coord(case["theta"] + rs.uniform(-np.pi / 4, np.pi / 4), case["phi"] + rs.uniform(-np.pi / 4, np.pi / 4))
        ambience = float(rs.uniform(0, 0.2))
    rec = {"np_seed": seed, "chess_n": chess_n, "light_dir": np.asarray(light if light is not None else np.zeros(3)),
           "ambience": ambience, "blur_sigma": blur_holder["b"].last_sigma if "b" in blur_holder else -1.0,
           "choice_i": int(ns["choice_i"])}
    for k in ("color_fine", "extra_color_fine", "weight_sum", "color_fine_loss", "eikonal_loss", "mask_loss", "cosine", "loss", "mask"):
        rec[k] = ns[k].detach().numpy()
    if self.add_no_texture or self.texture_cast_light:
        rec["texture_shading"] = ns["texture_shading"].detach().numpy()
        rec["rand_shading_rgb"] = ns["rand_shading_rgb"].detach().numpy()
    if self.add_no_texture:
        rec["cosine_shading"] = ns["cosine_shading"].detach().numpy()
    if ns.get("background_rgb") is not None:
        rec["background_rgb"] = ns["background_rgb"].detach().numpy()
    if seen.get("background_rgb") is not None:
        rec["render_background_rgb"] = seen["background_rgb"].numpy()
    return rec


CASES = [
    # name, seed, side, S, silhouettes, background choice, flags
    ("full_black", 1, 24, 8, False, 3, dict()),
    ("full_white", 2, 24, 8, False, 0, dict(iter_i=4)),                       # face prompt iteration
    ("full_gauss", 3, 24, 8, False, 1, dict(is_front=0)),                     # back prompt
    ("full_chess", 4, 40, 8, False, 2, dict(add_no_texture=False)),
    ("sil_black", 5, 32, 8, True, 3, dict()),
    ("sil_white", 6, 32, 8, True, 0, dict(texture_cast_light=False)),
    ("sil_gauss", 7, 32, 8, True, 1, dict(mask_weight=0.0)),
    ("sil_chess", 8, 48, 8, True, 2, dict()),
    ("plain", 9, 24, 8, False, 3, dict(add_no_texture=False, texture_cast_light=False, use_face_prompt=False, use_back_prompt=False)),
]


def main():
    R = ref_loader.load_reference()
    AG = R.ref_ag
    code, lines = glue_source(os.path.join(AG, "main.py"))
    sphere_coord = extract_functions(os.path.join(AG, "models", "utils.py"), ["sphere_coord"])["sphere_coord"]
    g = torch.Generator().manual_seed(1234)
    enc_w = torch.randn(3 * 224 * 224, 512, generator=g) * (3 * 224 * 224) ** -0.5
    texts = [F.normalize(torch.randn(1, 512, generator=g), dim=-1) for _ in range(3)]
    out = {"enc_seed": 1234, "source_lines": np.asarray(lines)}
    for spec in CASES:
        case = make_case(*spec)
        rec = run_case(code, sphere_coord, case, enc_w, texts)
        p = case["name"] + "/"
        for k, v in rec.items():
            out[p + k] = np.asarray(v)
        for k, v in case["render_out"].items():
            out[p + "in_" + k] = v.numpy()
        out[p + "in_true_rgb"] = case["true_rgb"].numpy()
        if case["dilated_mask"] is not None:
            out[p + "in_dilated_mask"] = case["dilated_mask"].numpy()
        out[p + "meta"] = np.asarray([case["H"], case["W"], case["S"], case["R"], case["silhouettes"], case["choice"], case["seed"],
                                      case["is_front"], case["iter_i"]], dtype=np.int64)
        out[p + "angles"] = np.asarray([case["theta"], case["phi"]])
        fl = case["flags"]
        out[p + "flags"] = np.asarray([fl.get("add_no_texture", True), fl.get("texture_cast_light", True), fl.get("use_face_prompt", True),
                                       fl.get("use_back_prompt", True)], dtype=np.int64)
        out[p + "mask_weight"] = np.asarray(fl.get("mask_weight", 1.0))
        print("%-12s choice %d  loss %.6f  cosine %.6f" % (case["name"], rec["choice_i"], float(rec["loss"]), float(rec["cosine"])))
    np.savez_compressed(os.path.join(GOLD, "glue.npz"), **out)
    print("wrote", os.path.join(GOLD, "glue.npz"), "from main.py lines %d-%d" % lines)


if __name__ == "__main__":
    main()
