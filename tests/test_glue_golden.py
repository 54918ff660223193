"""Rows a14 / a15 of SURVEY.md section 8: the shading / scatter / loss glue of train_clip (main.py:387-534).

tests/golden/glue.npz was produced by oracle/gen_golden_glue.py, which EXECUTES the reference's own source lines
(main.py:387-535, cut out of the file as text) on injected inputs.  Two things are pinned against it here, on the CPU:
  * the oracle restatement (oracle/neus_oracle.py: cast_light, scatter_to_image, neus_losses, clip_preprocess, clip_cosine),
  * the product glue (avatarclip_amd/runner.py: shade_and_scatter, assemble_loss, draw_background / chess_background) --
    the same torch code runs on the GPU in the iteration tests.
Tolerance: 1e-6 absolute on images, 1e-5 relative on the scalar losses (identical fp32 arithmetic up to op order).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue.npz")
CASES = ["full_black", "full_white", "full_gauss", "full_chess", "sil_black", "sil_white", "sil_gauss", "sil_chess", "plain"]


def _load(case):
    z = np.load(GOLD)
    p = case + "/"
    d = {k[len(p):]: z[k] for k in z.files if k.startswith(p)}
    g = torch.Generator().manual_seed(int(z["enc_seed"]))
    enc_w = torch.randn(3 * 224 * 224, 512, generator=g) * (3 * 224 * 224) ** -0.5
    texts = [F.normalize(torch.randn(1, 512, generator=g), dim=-1) for _ in range(3)]
    return d, enc_w, texts


def _render_out(d):
    return {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("in_") and k not in ("in_true_rgb", "in_dilated_mask")}


def _pick_text(texts, flags, iter_i, is_front):
    use_face, use_back = bool(flags[2]), bool(flags[3])
    if use_face and iter_i % 4 == 0:
        return texts[1]
    if use_back and is_front == 0:
        return texts[2]
    return texts[0]


@pytest.mark.parametrize("case", CASES)
def test_oracle_glue_matches_reference_lines(case):
    from oracle import neus_oracle as O
    d, enc_w, texts = _load(case)
    H, W, S, R, sil, choice, seed, is_front, iter_i = [int(v) for v in d["meta"]]
    flags = d["flags"]
    add_no_texture, cast = bool(flags[0]), bool(flags[1])
    out = _render_out(d)
    true_rgb = torch.from_numpy(d["in_true_rgb"])
    mask = (true_rgb != 0).float()[..., :1]
    mask = (mask > 0.5).float() if float(d["mask_weight"]) > 0 else torch.ones_like(mask)
    assert np.array_equal(mask.numpy(), d["mask"])
    tex = shade = None
    if add_no_texture or cast:
        tex, shade = O.cast_light(out, d["light_dir"], float(d["ambience"]))
    color, extra, wsum = out["color_fine"], out["extra_color_fine"], out["weight_sum"]
    if sil:
        dil = torch.from_numpy(d["in_dilated_mask"])
        bg_rgb = torch.from_numpy(d["background_rgb"]) if "background_rgb" in d else None
        background = O.silhouette_background(H, W, choice, bg_rgb, dil)
        if tex is not None:
            tex, shade = O.scatter_to_image(tex, dil, background), O.scatter_to_image(shade, dil, background)
        extra = O.scatter_to_image(extra, dil, background)
        color = O.scatter_to_image(color, dil, torch.zeros(H, W, 3))
        wsum = O.scatter_to_image(wsum, dil, torch.zeros(H, W, 1))
    for name, val in (("color_fine", color), ("extra_color_fine", extra), ("weight_sum", wsum), ("texture_shading", tex), ("rand_shading_rgb", shade)):
        if val is not None:
            assert np.abs(val.numpy() - d[name]).max() < 1e-6, name
    full = dict(out, color_fine=color, weight_sum=wsum)
    loss, closs, eik, mloss = O.neus_losses(full, true_rgb, mask, 0.1, float(d["mask_weight"]))
    text = _pick_text(texts, flags, iter_i, is_front)
    img = tex if cast else extra
    enc = O.clip_preprocess(img.reshape(H, W, 3)).reshape(1, -1) @ enc_w
    cosine = O.clip_cosine(enc, text)
    loss = loss + (1.0 - cosine)
    if add_no_texture:
        enc2 = O.clip_preprocess(shade.reshape(H, W, 3)).reshape(1, -1) @ enc_w
        cs = O.clip_cosine(enc2, text)
        loss = loss + (1.0 - cs)
        assert abs(float(cs) - float(d["cosine_shading"])) < 1e-5
    for name, val in (("color_fine_loss", closs), ("eikonal_loss", eik), ("mask_loss", mloss), ("cosine", cosine), ("loss", loss)):
        assert abs(float(val) - float(d[name])) < 1e-5 * max(1.0, abs(float(d[name]))), (name, float(val), float(d[name]))


@pytest.mark.parametrize("case", CASES)
def test_runner_glue_matches_reference_lines(case):
    import types
    import bench
    from avatarclip_amd.runner import Runner, chess_background
    d, enc_w, texts = _load(case)
    H, W, S, R, sil, choice, seed, is_front, iter_i = [int(v) for v in d["meta"]]
    flags = d["flags"]
    conf = bench.make_conf(H, S, small=True)
    for k, v in zip(("add_no_texture", "texture_cast_light", "use_face_prompt", "use_back_prompt"), flags):
        conf.put("train." + k, bool(v))
    conf.put("train.use_silhouettes", bool(sil))
    conf.put("train.mask_weight", float(d["mask_weight"]))
    r = Runner(None, mode="train_clip", conf=conf, device=torch.device("cpu"))

    class Fake:
        def encode_image(self, x):
            return x.reshape(x.shape[0], -1) @ enc_w
    r.init_clip(perceptor=Fake(), text_embeddings=dict(prompt=texts[0], face_prompt=texts[1], back_prompt=texts[2]))
    out = _render_out(d)
    true_rgb = torch.from_numpy(d["in_true_rgb"])
    view = types.SimpleNamespace(H=H, W=W, theta=float(d["angles"][0]), phi=float(d["angles"][1]), is_front=is_front,
                                 true_rgb=true_rgb, mask=(true_rgb != 0).float()[..., :1],
                                 dilated_mask=torch.from_numpy(d["in_dilated_mask"]) if sil else None)
    bg_rgb = torch.from_numpy(d["background_rgb"]) if "background_rgb" in d else None
    comp = r.shade_and_scatter(out, view, choice, bg_rgb, light=(d["light_dir"], float(d["ambience"])))
    for name in ("color_fine", "extra_color_fine", "weight_sum", "texture_shading", "rand_shading_rgb"):
        if name in d:
            assert np.abs(comp[name].numpy() - d[name]).max() < 1e-6, name
    if sil:   # the product's own views carry the row-major pixel positions of the mask (dataset.gen_rays_silhouettes) and scatter /
              # gather by index instead of by boolean mask (no stream synchronisation): the same images
        view.sel_idx = view.dilated_mask.reshape(-1).nonzero().squeeze(1)
        comp_i = r.shade_and_scatter(out, view, choice, bg_rgb, light=(d["light_dir"], float(d["ambience"])))
        for name, v in comp.items():
            assert (v is None and comp_i[name] is None) or torch.equal(v, comp_i[name]), name
    loss, parts = r.assemble_loss(out, comp, view, iter_i)
    for name, val in (("color_fine_loss", parts["color"]), ("eikonal_loss", parts["eikonal"]), ("mask_loss", parts["mask"]),
                      ("cosine", parts["cosine"]), ("loss", loss)):
        assert abs(float(val) - float(d[name])) < 1e-5 * max(1.0, abs(float(d[name]))), (name, float(val), float(d[name]))
    if "cosine_shading" in d:
        assert abs(float(parts["cosine_shading"]) - float(d["cosine_shading"])) < 1e-5
    # the numpy draw order of the lines (choice, [chess length], light angles, ambience) through the product's own draws
    np.random.seed(int(d["np_seed"]))
    torch.manual_seed(seed)
    ch, bg2, masked = r.draw_background(view)
    assert ch == choice == int(d["choice_i"])
    if choice == 0:
        assert torch.equal(bg2, torch.ones(1, 3))
    if choice == 1:   # same torch seed, same shapes: the gaussian background is the same draw
        assert np.abs(bg2.numpy() - d["background_rgb"]).max() < 1e-6
    if choice == 2:   # the blur sigma comes from torch's CPU generator (torchvision GaussianBlur.get_params)
        assert np.abs(bg2.numpy() - d["background_rgb"]).max() < 1e-5
        assert np.abs(chess_background(H, W, H // int(d["chess_n"]), float(d["blur_sigma"]), "cpu").numpy() - d["background_rgb"]).max() < 1e-5
    if "render_background_rgb" in d:
        assert np.abs(masked.numpy() - d["render_background_rgb"]).max() < 1e-5
    comp2 = r.shade_and_scatter(out, view, choice, bg2)     # light / ambience drawn from numpy in the reference's order
    if "texture_shading" in d:
        assert np.abs(comp2["texture_shading"].numpy() - d["texture_shading"]).max() < 2e-5


def test_random_resized_crop_scale_one_is_the_full_frame():
    """main.py:261: RandomResizedCrop(224, scale=(1,1)) on the square render always crops the whole frame."""
    from oracle import neus_oracle as O
    rng = np.random.RandomState(0)
    for side in (64, 113, 224, 256, 512):
        for _ in range(400):
            assert O.random_resized_crop_params(side, side, rng) == (0, 0, side, side)
