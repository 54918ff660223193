"""The fused glue kernels (csrc/avc_glue.hip, avatarclip_amd/glue.py: Lambert shading + silhouette scatter + per-pixel loss terms, and
CLIP's resize + normalise) against the torch statement of the same reference lines (Runner.shade_and_scatter / assemble_loss = main.py:
422-534, themselves pinned against the reference's own lines by tests/test_glue_golden.py): images, loss terms and the gradients with
respect to everything the renderer hands over, for every flag combination and every background choice, in both sampling modes."""
import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu


class _StubPerceptor:
    """a smooth fp32 stand-in for encode_image, so that the comparison tests the glue and not bf16 rounding flips inside the ViT"""

    def __init__(self, dev):
        g = torch.Generator().manual_seed(3)
        self.w = (torch.randn(3 * 224 * 224, 512, generator=g) / 400.0).to(dev)

    def encode_image(self, x):
        return x.reshape(x.shape[0], -1) @ self.w


def _runner(silhouettes, add_no_texture, texture_cast_light, dev):
    import bench
    from avatarclip_amd.runner import Runner
    conf = bench.make_conf(256 if silhouettes else 56, 32, small=True)
    conf.put("train.use_silhouettes", silhouettes)
    conf.put("train.max_ray_num", 2500)
    conf.put("train.add_no_texture", add_no_texture)
    conf.put("train.texture_cast_light", texture_cast_light)
    conf.put("train.warm_up_end", 0)
    torch.manual_seed(0)
    np.random.seed(5)
    r = Runner(None, mode="train_clip", conf=conf, device=dev)
    r.init_clip(perceptor=_StubPerceptor(dev))
    r.init_smpl()
    return r


@gpu
@pytest.mark.parametrize("silhouettes", [True, False])
@pytest.mark.parametrize("add_no_texture,texture_cast_light", [(True, True), (True, False), (False, True), (False, False)])
def test_fused_glue_equals_the_torch_statement(silhouettes, add_no_texture, texture_cast_light):
    from avatarclip_amd.renderer import RenderOut
    dev = torch.device("cuda")
    r = _runner(silhouettes, add_no_texture, texture_cast_light, dev)
    for it, choice in enumerate((0, 1, 2, 3) if silhouettes else (3,)):
        view = r.make_view(it)
        _, background_rgb, masked_bg = r.draw_background(view, choice_i=choice)
        with torch.no_grad():
            ro = r.renderer.render(view.rays_o, view.rays_d, view.near, view.far, background_rgb=masked_bg, cos_anneal_ratio=1.0)
        R = view.rays_o.shape[0]
        g = torch.Generator().manual_seed(11 + it)
        base = dict(color_fine=ro["color_fine"], extra_color_fine=ro["extra_color_fine"] * 1.3 - 0.1,     # (some values outside [0,1]: the clamps)
                    weight_sum=(ro["weight_sum"] + (torch.rand(R, 1, generator=g).to(dev) - 0.5) * 0.6).clamp(-0.05, 1.05),
                    weighted_normals=ro.weighted_normals + torch.randn(R, 3, generator=g).to(dev) * 0.05)
        light = (np.array([0.3, -0.5, 0.8]), 0.13)
        r.writer = object()          # (makes both statements compute the logged psnr of main.py:493)
        res = []
        for fused in (False, True):
            leaf = {k: v.detach().clone().requires_grad_(True) for k, v in base.items()}
            out = RenderOut({"color_fine": leaf["color_fine"], "extra_color_fine": leaf["extra_color_fine"], "weight_sum": leaf["weight_sum"],
                             "gradient_error": ro["gradient_error"].detach(), "s_val": ro["s_val"]})
            out.weighted_normals = leaf["weighted_normals"]
            if fused:
                loss, parts, images = r.fused_shade_loss(out, view, choice, background_rgb, it, light=light)
            else:
                comp = r.shade_and_scatter(out, view, choice, background_rgb, light=light)
                loss, parts = r.assemble_loss(out, comp, view, it)
                img0 = comp["texture_shading"] if texture_cast_light else comp["extra_color_fine"]
                img1 = comp["rand_shading_rgb"] if comp["rand_shading_rgb"] is not None else img0
                images = torch.stack([img0.reshape(-1, 3), img1.reshape(-1, 3)])
            loss.backward()
            res.append((loss.detach(), {k: v.detach() for k, v in parts.items() if v is not None and k in ("color", "mask", "cosine", "cosine_shading", "psnr")},
                        images.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}))
        (la, pa, ia, ga), (lb, pb, ib, gb) = res
        nimg = 2 if add_no_texture else 1
        assert (ia[:nimg] - ib[:nimg]).abs().max() < 1e-6, (choice, (ia[:nimg] - ib[:nimg]).abs().max())
        assert abs(la.item() - lb.item()) < 1e-5 * max(1.0, abs(la.item())), (choice, la.item(), lb.item())
        for k in pa:
            assert abs(pa[k].item() - pb[k].item()) < 1e-5 * max(1.0, abs(pa[k].item())), (choice, k, pa[k].item(), pb[k].item())
        for k in ga:
            if not (add_no_texture or texture_cast_light) and k == "weighted_normals":
                assert ga[k].abs().max() == 0 and gb[k].abs().max() == 0
                continue
            den = ga[k].norm().item()
            assert den > 0, k
            rel = (ga[k] - gb[k]).norm().item() / den
            assert rel < 2e-5, (choice, k, rel)


@gpu
@pytest.mark.parametrize("H", [224, 97, 300])
def test_resize_norm_equals_interpolate(H):
    from avatarclip_amd import glue
    from avatarclip_amd.clip_vit import clip_preprocess
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(2)
    img = torch.rand(2, H, H, 3, generator=g).to(dev)
    a = img.clone().requires_grad_(True)
    b = img.clone().requires_grad_(True)
    ya = torch.cat([clip_preprocess(a[i]) for i in range(2)], dim=0)
    yb = glue.ResizeNormFn.apply(b)
    assert ya.shape == yb.shape == (2, 3, 224, 224)
    assert (ya - yb).abs().max() < 2e-5
    w = torch.randn(2, 3, 224, 224, generator=g).to(dev)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    assert (a.grad - b.grad).abs().max() < 1e-4 * a.grad.abs().max()


@gpu
@pytest.mark.parametrize("silhouettes", [True, False])
def test_fused_head_equals_the_torch_statement(silhouettes, monkeypatch):
    """Runner.make_view through dataset.rays_fused (rays, near / far and the resampled prior in one launch) against the torch statement
    of dataset.py:252-293,331-342 and main.py:376-380 -- which tests/test_host_logic.py pins against the reference's own functions."""
    dev = torch.device("cuda")
    r = _runner(silhouettes, True, True, dev)
    cam = r.sample_camera(1)
    views = []
    for fused in ("1", "0"):
        monkeypatch.setenv("AVC_FUSED_HEAD", fused)
        views.append(r.make_view(1, camera=cam))
    a, b = views
    assert (a.H, a.W) == (b.H, b.W) and a.rays_o.shape == b.rays_o.shape
    for k in ("rays_o", "rays_d", "near", "far", "true_rgb", "mask"):
        x, y = getattr(a, k), getattr(b, k)
        assert x.shape == y.shape, k
        assert (x - y).abs().max() < 2e-6, (k, (x - y).abs().max())
    if silhouettes:
        assert torch.equal(a.sel_idx, b.sel_idx) and torch.equal(a.dilated_mask, b.dilated_mask) and torch.equal(a.ray_of_pixel, b.ray_of_pixel)


@gpu
def test_fused_chess_background_equals_the_torch_statement():
    from avatarclip_amd import runner as RN
    dev = torch.device("cuda")
    for H, W, L, sigma in ((190, 190, 13, 0.7), (224, 224, 11, 1.9), (64, 80, 5, 0.1)):
        a = RN.chess_background(H, W, L, sigma, dev)
        b = RN.chess_background_fused(H, W, L, sigma, dev)
        assert a.shape == b.shape == (H * W, 1)
        assert (a - b).abs().max() < 2e-6


@gpu
@pytest.mark.parametrize("B,T", [(1, 1), (2, 1), (2, 3)])
def test_loss_tail_equals_the_torch_statement(B, T):
    """glue.LossTailFn (cosines + colour / mask normalisation + weighted sum, main.py:491-534) against the same lines in torch: values
    of the loss and its parts, gradients to the embeddings, the shade-loss sums and the eikonal term."""
    from avatarclip_amd import glue
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(4 + B + T)
    enc0 = torch.randn(B, 512, generator=g).to(dev)
    text = torch.randn(T, 512, generator=g).to(dev)
    text = text / text.norm(dim=-1, keepdim=True)
    sums0 = torch.tensor([812.5, 30211.0, 9120.25, 77.0]).to(dev)
    eik0 = torch.tensor(0.0371).to(dev)
    w = dict(igr=0.1, mask=0.3, clip=1.7)
    P = 224.0 * 224.0
    res = []
    for fused in (False, True):
        enc, sums, eik = (t.clone().requires_grad_(True) for t in (enc0, sums0, eik0))
        if fused:
            loss, st = glue.LossTailFn.apply(enc, text, sums, eik, w["igr"], w["mask"], w["clip"], P)
            parts = [st[1], st[2]] + [st[3 + b] for b in range(B)]
        else:
            colour = sums[0] / (sums[1] + 1e-5)
            maskl = sums[2] / P
            cs = [torch.cosine_similarity(torch.mean(enc[b:b + 1], dim=0), torch.mean(text, dim=0), dim=0) for b in range(B)]
            loss = colour + eik * w["igr"] + maskl * w["mask"]
            for c in cs:
                loss = loss + (1.0 - c) * w["clip"]
            parts = [colour, maskl] + cs
        (loss * 1.9).backward()
        res.append((loss.detach(), [p.detach() for p in parts], enc.grad, sums.grad, eik.grad))
    (la, pa, ea, sa, ka), (lb, pb, eb, sb, kb) = res
    assert abs(la.item() - lb.item()) < 2e-6 * abs(la.item())
    for x, y in zip(pa, pb):
        assert abs(x.item() - y.item()) < 2e-6 * max(1.0, abs(x.item())), (x.item(), y.item())
    assert (ea - eb).norm() < 1e-5 * ea.norm()
    # (sums[1], the mask count, and sums[3], the psnr's squared error, carry no gradient: constants of the iteration)
    assert torch.allclose(sa[[0, 2]], sb[[0, 2]], rtol=1e-5, atol=1e-12) and sb[1].item() == 0.0 and sb[3].item() == 0.0
    assert abs(ka.item() - kb.item()) < 1e-6 * abs(ka.item())
