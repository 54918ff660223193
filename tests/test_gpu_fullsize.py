"""GPU tests at BASELINE.json's full sizes (512 x 512 rays, 64 and 128 samples per ray, full-size nets) through
size-independent properties, plus the edge cases of the hot path (empty / single / ragged inputs, the 128-spp
configuration against the oracle on a few rays).

The CPU oracle cannot run 16.8 M points in test time, so at full size we check what must hold for ANY input:
  * hierarchical sampling returns sorted depths inside [near, far + sample_dist], finite sdf          (renderer.py:133-193)
  * alpha compositing returns weights in [0,1] with sum <= 1, colours in [0,1]                        (renderer.py:234-286)
  * the parameter gradient is linear in the upstream gradient: scaling d_sdf, d_normal, d_rgb by 2 (exact in binary
    floating point) scales every dense gradient by 2 up to the fp32 summation order of the final scatter  (main.py:537)
  * the parameter gradient does not depend on how the points are chunked over launches (fp32 re-association only)
"""
import numpy as np
import pytest
import torch

from oracle import neus_oracle as O

gpu = pytest.mark.gpu


def _full_renderer(dev, n_samples=32, n_importance=32, seed=0):
    from avatarclip_amd import fields, renderer
    torch.manual_seed(seed)
    sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                            geometric_init=True, weight_norm=True)
    col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2,
                                  weight_norm=True, multires_view=0, squeeze_out=True, extra_color=True)
    var = fields.SingleVarianceNetwork(0.3)
    sdf, col, var = sdf.to(dev), col.to(dev), var.to(dev)
    ren = renderer.NeuSRenderer(None, sdf, var, col, n_samples=n_samples, n_importance=n_importance, n_outside=0,
                                up_sample_steps=4, perturb=1.0, extra_color=True)
    return sdf, col, var, ren


def _view(res, dev, eye=(0.3, 0.2, 1.5)):
    pose = torch.from_numpy(O.lookat(np.array(eye), np.zeros(3), np.array([0., 1, 0]))).float()
    o, v = O.gen_rays_pose(pose, res, res, 0.5 * res / np.tan(np.pi / 6))
    ro, rd = o.reshape(-1, 3).contiguous(), v.reshape(-1, 3).contiguous()
    near, far = O.near_far_from_sphere(ro, rd)
    return ro.to(dev), rd.to(dev), near.to(dev), far.to(dev)


@gpu
@pytest.mark.parametrize("spp", [64, 128])
def test_fullsize_sampling_and_compositing_properties(spp):
    dev = torch.device("cuda")
    sdf, col, var, ren = _full_renderer(dev, spp // 2, spp // 2)
    ro, rd, near, far = _view(512, dev)
    R = ro.shape[0]
    assert R == 512 * 512
    with torch.no_grad():
        out = ren.render(ro, rd, near, far, background_rgb=torch.zeros(1, 3, device=dev), cos_anneal_ratio=1.0)
    torch.cuda.synchronize()
    sample_dist = 2.0 / (spp // 2)
    mid = out["mid_z_vals"]
    assert mid.shape == (R, spp)
    assert torch.isfinite(mid).all()
    assert (mid[:, 1:] >= mid[:, :-1] - 1e-6).all(), "section mid-points must be sorted along every ray"
    assert (mid >= near - 1.0 / (spp // 2) - 1e-4).all() and (mid <= far + sample_dist + 1e-4).all()
    w = out["weights"]
    assert torch.isfinite(w).all() and (w >= 0).all() and (w <= 1 + 1e-6).all()
    ws = out["weight_sum"]
    assert (ws <= 1 + 1e-4).all()
    assert torch.allclose(ws, w.sum(-1, keepdim=True), atol=1e-5)
    for k in ("color_fine", "extra_color_fine"):
        c = out[k]
        assert c.shape == (R, 3) and torch.isfinite(c).all() and (c >= -1e-6).all() and (c <= 1 + 1e-4).all()
    assert torch.isfinite(out["gradients"]).all() and torch.isfinite(out["gradient_error"])
    # geometric init: the SDF is a sphere of radius ~0.5, so rays through the centre hit it and corner rays miss it
    # (inv_s = exp(3) at initialisation: the surface is blurred, a missing ray still collects some weight)
    centre = ws.reshape(512, 512)[256, 256].item()
    corner = ws.reshape(512, 512)[0, 0].item()
    assert centre > 0.9 and corner < 0.4, (centre, corner)


def _pick_256_rays(eng, R, spp=64):
    """the first and the last rays, the rays whose panel blocks straddle every multiple of 4 GiB of the forward panel region, and
    random ones"""
    fwd_bytes_per_block = eng.fwd_tiles * 2048
    pick = {0, 1, R - 2, R - 1}
    k = 1
    while k * (1 << 32) < (R * spp // 32) * fwd_bytes_per_block:
        blk = k * (1 << 32) // fwd_bytes_per_block
        for b in (blk - 1, blk, blk + 1):
            pick.add(min(R - 1, max(0, b * 32 // spp)))
        k += 1
    rs = np.random.RandomState(3)
    while len(pick) < 256:
        pick.add(int(rs.randint(0, R)))
    return torch.tensor(sorted(pick)[:256], dtype=torch.long)


@gpu
def test_fullsize_one_launch_render_matches_oracle_on_256_rays():
    """512 x 512 rays x 64 spp through the TRAINING path (one forward launch over the whole view: operand-panel offsets far
    beyond 2^32 bytes), then 256 of its rays -- the first and the last, the rays whose panel blocks straddle every multiple of
    4 GiB of the forward panel region, and random ones -- against the CPU oracle on the kernel's own depths
    (renderer.py:195-300, fields.py:72-107,154-185)."""
    dev = torch.device("cuda")
    sdf, col, var, ren = _full_renderer(dev)
    ro, rd, near, far = _view(512, dev)
    R = ro.shape[0]
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    with torch.no_grad():
        z = ren.sample_z(pk, ro, rd, near, far, 1.0, jitter=torch.rand(R, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(9)))
    bg = torch.tensor([[0.1, 0.6, 0.3]], device=dev)
    out = ren.render(ro, rd, near, far, background_rgb=bg, cos_anneal_ratio=0.7, z_vals=z)     # grad mode on: avc_render_points_fwd_train
    assert out["color_fine"].requires_grad
    assert eng.rays_per_chunk(R, 64) >= R, "the whole view is expected to be one launch on a 288 GB device"
    torch.cuda.synchronize()
    idx = _pick_256_rays(eng, R)
    sd_s = {k_: v.detach().cpu() for k_, v in sdf.named_parameters()}
    sd_c = {k_: v.detach().cpu() for k_, v in col.named_parameters()}
    di = idx.to(dev)
    ref = O.render(sd_s, sd_c, var.variance.detach().cpu(), ro[di].cpu(), rd[di].cpu(), near[di].cpu(), far[di].cpu(),
                   background_rgb=bg.cpu(), cos_anneal_ratio=0.7, z_vals=z[di].cpu())
    for key, tol in (("color_fine", 5e-3), ("extra_color_fine", 5e-3), ("weights", 5e-3)):
        e = (out[key].detach()[di].cpu() - ref[key].detach()).abs()
        print(key, "max", e.max().item(), "mean", e.mean().item(), "rays", len(idx))
        assert e.max() < tol, key


def _pick_rays(eng, R, n, spp=64):
    """_pick_256_rays' set (boundaries first) topped up with random rays to n"""
    pick = set(_pick_256_rays(eng, R, spp).tolist())
    rs = np.random.RandomState(17)
    while len(pick) < n:
        pick.add(int(rs.randint(0, R)))
    return torch.tensor(sorted(pick), dtype=torch.long)


def _dense_gradient_errors(dev, nsel, two_slabs, monkeypatch, coherent=False):
    """Dense gradient of ONE 512 x 512 x 64-spp training launch for a loss supported on `nsel` rays (zero cotangents elsewhere) against
    the oracle's autograd on those rays.  Returns (loss, oracle loss, [(name, rel, cos, tiny)]).  coherent = False: independent normal
    coefficients per ray (the gradient is itself a random sum over rays); True: the SAME coefficients on every ray, as in a real loss
    (the gradient grows with the number of rays, rounding noise only with its square root)."""
    import gc
    from avatarclip_amd.engine import Engine
    gc.collect()                 # an earlier case's engine (180 GiB of operand panels) must be gone before this one plans its buffers
    torch.cuda.empty_cache()
    if two_slabs:
        monkeypatch.setattr(Engine, "SLAB_BLOCKS", 256 * 1024)      # two slabs (the default is one slab for this ray set): the boundary is part of the test
        monkeypatch.setattr(Engine, "SDF_BIAS_FP32", False)         # ... and so is the hi + lo d_sdf tile: the sdf bias out of the products, no fp32 side path
    sdf, col, var, ren = _full_renderer(dev)
    with torch.no_grad():   # off the degenerate initialisation (the PE columns of layer 0 are zero there), as oracle/gen_golden.py does
        gp = torch.Generator().manual_seed(11)
        for p in list(sdf.parameters()) + list(col.parameters()):
            p.add_((torch.randn(p.shape, generator=gp) * 0.02 * float(p.abs().mean().clamp(min=0.05))).to(dev))
        var.variance.fill_(0.45)
    ro, rd, near, far = _view(512, dev)
    R = ro.shape[0]
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    with torch.no_grad():
        z = ren.sample_z(pk, ro, rd, near, far, 1.0, jitter=torch.rand(R, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(9)))
    bg = torch.tensor([[0.1, 0.6, 0.3]], device=dev)
    idx = _pick_256_rays(eng, R) if nsel == 256 else _pick_rays(eng, R, nsel)
    di = idx.to(dev)
    gc = torch.Generator().manual_seed(4)
    c1, c2, cw = torch.randn(nsel, 3, generator=gc), torch.randn(nsel, 3, generator=gc), torch.randn(nsel, 1, generator=gc)
    cn = torch.randn(nsel, 3, generator=gc) * 0.3
    if coherent:
        c1, c2 = torch.tensor([[0.7, -0.4, 0.5]]).expand(nsel, 3), torch.tensor([[-0.3, 0.8, 0.6]]).expand(nsel, 3)
        cw, cn = torch.full((nsel, 1), 0.9), torch.tensor([[0.2, -0.3, 0.25]]).expand(nsel, 3)

    def loss_of(out, sel, t):
        nsum = (out["gradients"][sel] * out["weights"][sel][..., None]).sum(1)
        return ((out["color_fine"][sel] * t(c1)).sum() + (out["extra_color_fine"][sel] * t(c2)).sum()
                + (out["weight_sum"][sel] * t(cw)).sum() + (nsum * t(cn)).sum())

    out = ren.render(ro, rd, near, far, background_rgb=bg, cos_anneal_ratio=0.7, z_vals=z)
    assert eng.rays_per_chunk(R, 64) >= R, "one forward launch"
    if two_slabs:
        assert eng.plan(R, 64)[1] < R, "more than one backward slab"
    loss = loss_of(out, di, lambda v: v.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    leaf = lambda net: {k_: v.detach().cpu().clone().requires_grad_(True) for k_, v in net.named_parameters()}
    sd_s, sd_c = leaf(sdf), leaf(col)
    variance = var.variance.detach().cpu().clone().requires_grad_(True)
    ref = O.render(sd_s, sd_c, variance, ro[di].cpu(), rd[di].cpu(), near[di].cpu(), far[di].cpu(), background_rgb=bg.cpu(),
                   cos_anneal_ratio=0.7, z_vals=z[di].cpu())
    loss_ref = loss_of(ref, slice(None), lambda v: v)
    loss_ref.backward()
    pairs = [("sdf." + k_, p.grad, sd_s[k_].grad) for k_, p in sdf.named_parameters()]
    pairs += [("col." + k_, p.grad, sd_c[k_].grad) for k_, p in col.named_parameters()]
    pairs += [("var.variance", var.variance.grad, variance.grad)]
    gnorm = float(np.sqrt(sum(float(r.double().pow(2).sum()) for _, _, r in pairs if r is not None)))
    rows = []
    for name, g, r in pairs:
        if r is None or r.abs().max() < 1e-9:
            continue
        g = g.detach().cpu() if g is not None else torch.zeros_like(r)
        rel = ((g.double() - r.double()).norm() / r.double().norm()).item()
        cos = torch.nn.functional.cosine_similarity(g.reshape(1, -1).double(), r.reshape(1, -1).double()).item()
        tiny = r.double().norm().item() < 1e-4 * gnorm
        print("  %5d rays  %-22s rel %.3e cos %.6f |ref| %.3e%s" % (nsel, name, rel, cos, r.norm().item(), "  (tiny)" if tiny else ""))
        rows.append((name, rel, cos, tiny))
    return loss.item(), loss_ref.item(), rows


@gpu
def test_fullsize_dense_gradient_of_256_rays_matches_the_oracles_autograd(monkeypatch):
    """The backward of the 512 x 512 x 64-spp training launch against the oracle (main.py:537 through renderer.py:195-300 and
    fields.py:72-107,154-185): the loss is supported on the same 256 rays as the forward test (first / last ray, the rays at every
    4-GiB boundary of the F panel region, random ones) and has zero cotangents everywhere else, so the dense gradient that comes out
    of the two-slab, 110-GiB backward pass (avc_render_points_bwd + avc_weight_grad_all over 16.8 M points, panel offsets far beyond
    2^32) must equal the autograd gradient of the oracle rendering just those 256 rays.  SURVEY 8d gate: 1e-2 per tensor; the colour
    tensors get 1.5e-2 HERE because this loss has independent random coefficients per ray: its gradient is a random sum that grows like
    sqrt(rays), exactly like the per-point rounding noise (ReLU sign flips of f16-operand pre-activations), so their ratio -- 1.0-1.2 % on
    colour layer 0 -- is a property of the probe and does not shrink with more rays (1.16 % at 256 rays, 1.07 % at 4 096).  Under a loss with
    one sign per term, test_dense_gradient_under_a_coherent_loss_at_16k_and_262k_points holds EVERY tensor to 5e-3.  The loss takes colours,
    the CLIP colours, the weight sums and the normals (sum_i w_i n_i, main.py:428) -- not the eikonal term, whose normaliser runs over
    all rays of the view."""
    dev = torch.device("cuda")
    loss, loss_ref, rows = _dense_gradient_errors(dev, 256, True, monkeypatch)
    print("loss", loss, "oracle", loss_ref)
    assert abs(loss - loss_ref) < 2e-3 * max(1.0, abs(loss_ref))
    bad = []
    for name, rel, cos, tiny in rows:
        # SDF tensors: SURVEY's 1e-2 (measured 0.01-0.31 %, the sdf bias -- a heavily cancelling sum -- 0.03 % out of the hi + lo d_sdf
        # tile).  Colour tensors: 1.5e-2 (layer 0 sits at 1.0-1.2 % against this random-sign loss -- zero-mean noise measured against a
        # zero-mean signal; 0.2 % against a coherent one, see the test below)
        gate = 2e-2 if tiny else (1.5e-2 if name.startswith("col.") else 1e-2)
        if not (rel < gate and cos > 0.9995):
            bad.append((name, rel, cos))
    assert not bad, bad


@gpu
def test_dense_gradient_under_a_coherent_loss_at_16k_and_262k_points(monkeypatch):
    """is the 1.0-1.2 % of the colour tensors in the random-coefficient fixtures sampling noise (ReLU sign flips of
    f16-operand pre-activations, rounding of cancelling sums) or a per-point bias?  Noise grows like sqrt(points), a bias like the
    points -- but that only shows against a SIGNAL that grows like the points.  With independent random coefficients per ray (the
    256-ray test above, the golden fixtures) the gradient is itself a random sum, signal and noise both grow like sqrt(rays), and the
    ratio cannot move: measured 1.16 % at 256 rays, 1.07 % at 4 096 on col.lin0.weight_v (profiles/r05_gradient_noise.md).  This test
    gives every ray the SAME coefficients, as every real loss does (main.py:489-534: one sign per term): the same launch, the same
    weights, 256 rays (16 K points) and 4 096 rays (262 K points).  A per-point bias of the ReLU masks would show here as >= 1 % at both
    sizes.  Measured: EVERY tensor, the colour tensors included, sits at 0.003-0.25 % at BOTH sizes -- what is left is the one error that
    is identical for all points, the f16 / bf16 rounding of the weights themselves (2^-9 = 0.2 % for bf16).  Gate: 5e-3 per tensor at both
    sizes, no carve-out (SURVEY 8d asks 1e-2; fields.py:154-185 under main.py:537)."""
    dev = torch.device("cuda")
    rows = {}
    for nsel in (256, 4096):
        loss, loss_ref, rows[nsel] = _dense_gradient_errors(dev, nsel, False, monkeypatch, coherent=True)
        assert abs(loss - loss_ref) < 2e-3 * max(1.0, abs(loss_ref))
    e256 = {n: r for n, r, _, _ in rows[256]}
    bad = []
    for name, rel, cos, tiny in rows[4096]:
        print("  %-22s 256 rays %.3e -> 4096 rays %.3e" % (name, e256[name], rel))
        if not (rel < 5e-3 and e256[name] < 5e-3 and cos > 0.9995):
            bad.append((name, e256[name], rel, cos))
    assert not bad, bad


@gpu
@pytest.mark.parametrize("spp", [64, 128])
def test_fullsize_gradient_linearity_and_chunk_invariance(spp):
    """spp = 128: BASELINE config 3's per-GPU share at FULL size through the training path -- 174 GiB of forward-type panels in one launch, the
    gradient-type slab sized from what is left (several slabs), block offsets far beyond 2^32 bytes in both regions (round 6)."""
    import gc
    from avatarclip_amd.engine import Engine
    gc.collect()
    torch.cuda.empty_cache()
    dev = torch.device("cuda")
    sdf, col, var, ren = _full_renderer(dev, spp // 2, spp // 2)
    ro, rd, near, far = _view(512, dev)
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    with torch.no_grad():
        z = ren.sample_z(pk, ro, rd, near, far, 1.0)
    R, S = z.shape
    assert (R, S) == (512 * 512, spp)
    g = torch.Generator(device=dev).manual_seed(5)
    d_sdf = torch.randn(R, S, device=dev, generator=g) * 1e-3
    d_n = torch.randn(R, S, 3, device=dev, generator=g) * 1e-3
    d_rgb = torch.randn(R, S, 6, device=dev, generator=g) * 1e-3
    # the training forward leaves the forward-type operand panels of the whole view (87 GiB) in the engine's F region; the first
    # backward uses them, the second one finds them still there (the backward only writes the per-slab G region)
    assert eng.plan(R, S)[0] >= R, "one forward launch for the whole view on a 288 GB device"
    _, _, rgb = eng.points_fwd_train(pk, ro, rd, z, 2.0 / (spp // 2))
    g1 = eng.points_bwd(pk, ro, rd, z, 2.0 / (spp // 2), d_sdf, d_n, d_rgb, rgb, panels_valid=True)
    g2 = eng.points_bwd(pk, ro, rd, z, 2.0 / (spp // 2), 2 * d_sdf, 2 * d_n, 2 * d_rgb, rgb, panels_valid=True)
    torch.cuda.synchronize()
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    # every kernel is exactly linear under power-of-two scaling; the final scatter of the (i)/(ii) products into the dense
    # gradient is an atomic index_add_ whose fp32 summation order varies from run to run (1 ulp on ~0.04 % of the entries)
    lin = ((g2 - 2 * g1).double().norm() / g2.double().norm()).item()
    print("linearity rel", lin)
    assert lin < 1e-7
    old = Engine.PANEL_BYTES_BUDGET
    try:
        Engine.PANEL_BYTES_BUDGET = 3 << 30      # many more, smaller launches: the backward re-runs the training forward per chunk
        chunk, slab = eng.plan(R, S)
        assert chunk < R and slab <= chunk and chunk % 32 == 0, (chunk, slab)   # the budget applies although larger buffers are at hand
        g3 = eng.points_bwd(pk, ro, rd, z, 2.0 / (spp // 2), d_sdf, d_n, d_rgb, rgb)
    finally:
        Engine.PANEL_BYTES_BUDGET = old
    torch.cuda.synchronize()
    rel = ((g3 - g1).double().norm() / g1.double().norm()).item()
    print("chunk invariance rel", rel)
    assert rel < 1e-5


@gpu
@pytest.mark.parametrize("spp", [48, 64])
def test_chunked_and_slabbed_backward_equals_one_pass(spp):
    """The gradient must not depend on how the ray set is cut: one chunk / one slab vs several chunks (the training forward
    re-run per chunk) of several slabs each, with ray counts that are no multiple of the chunk or slab size and a samples-per-ray
    count (48) that is no multiple of the 32-point block.  Small nets, 1000 rays."""
    from avatarclip_amd import fields, renderer
    from avatarclip_amd.engine import Engine
    dev = torch.device("cuda")
    torch.manual_seed(3)
    sdf = fields.SDFNetwork(d_out=129, d_in=3, d_hidden=128, n_layers=3, skip_in=[3], multires=6, bias=0.5, scale=1.0,
                            geometric_init=True, weight_norm=True).to(dev)
    col = fields.RenderingNetwork(d_feature=128, mode="no_view_dir", d_in=6, d_out=3, d_hidden=128, n_layers=1,
                                  weight_norm=True, multires_view=0, squeeze_out=True, extra_color=True).to(dev)
    var = fields.SingleVarianceNetwork(0.3).to(dev)
    ren = renderer.NeuSRenderer(None, sdf, var, col, spp // 2, spp // 2, 0, 4, 1.0, True)
    ro, rd, near, far = [t[:1000].contiguous() for t in _view(32, dev)]
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    with torch.no_grad():
        z = ren.sample_z(pk, ro, rd, near, far, 1.0)
    R, S = z.shape
    g = torch.Generator(device=dev).manual_seed(1)
    d_sdf = torch.randn(R, S, device=dev, generator=g) * 1e-2
    d_n = torch.randn(R, S, 3, device=dev, generator=g) * 1e-2
    d_rgb = torch.randn(R, S, 6, device=dev, generator=g) * 1e-2
    sd = 2.0 / (spp // 2)
    _, _, rgb = eng.points_fwd_train(pk, ro, rd, z, sd)
    assert eng.plan(R, S) == (R, R)
    g1 = eng.points_bwd(pk, ro, rd, z, sd, d_sdf, d_n, d_rgb, rgb, panels_valid=True)
    old = (Engine.PANEL_BYTES_BUDGET, Engine.SLAB_BLOCKS)
    try:
        Engine.SLAB_BLOCKS = 160                       # 96 rays per slab at 48 spp (3 x 32), 64 at 64 spp
        g2 = eng.points_bwd(pk, ro, rd, z, sd, d_sdf, d_n, d_rgb, rgb, panels_valid=True)        # one chunk, many slabs
        assert eng.plan(R, S)[0] == R and eng.plan(R, S)[1] < R // 8
        blk = eng.fwd_tiles * 2048 + eng.mask_u16 * 2 + eng.grad_tiles * 2048
        Engine.PANEL_BYTES_BUDGET = 1 << 26            # the floor of the budget (64 MiB): a few hundred rays per chunk
        chunk, slab = eng.plan(R, S)
        assert slab < chunk < R and chunk % 32 == 0 and slab % 32 == 0 and R % chunk != 0, (chunk, slab, blk)
        g3 = eng.points_bwd(pk, ro, rd, z, sd, d_sdf, d_n, d_rgb, rgb)                            # chunks x slabs, forward re-run per chunk
    finally:
        Engine.PANEL_BYTES_BUDGET, Engine.SLAB_BLOCKS = old
    torch.cuda.synchronize()
    for name, gx in (("slabs", g2), ("chunks x slabs", g3)):
        rel = ((gx - g1).double().norm() / g1.double().norm()).item()
        print(spp, name, "rel", rel)
        assert rel < 1e-5, (name, rel)


@gpu
def test_edge_cases_empty_single_and_ragged():
    dev = torch.device("cuda")
    sdf, col, var, ren = _full_renderer(dev)
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    # ragged point count (not a multiple of the 32-point block) in point mode vs ray mode on the same points
    N = 1000 + 7
    pts = (torch.rand(N, 3, device=dev) - 0.5) * 1.6
    s_pts = eng.sdf_pts(pk, pts)
    ro = pts.clone()
    rd = torch.zeros(N, 3, device=dev); rd[:, 2] = 1.0
    s_ray = eng.sdf_rays(pk, ro, rd, torch.zeros(N, 1, device=dev))
    assert s_pts.shape[0] == N and torch.isfinite(s_pts).all()
    assert torch.allclose(s_pts.reshape(-1), s_ray.reshape(-1), atol=5e-4)
    ref = O.sdf_forward({k: v.detach().cpu() for k, v in sdf.named_parameters()}, pts.cpu())[:, 0]
    assert (s_pts.reshape(-1).cpu() - ref).abs().max() < 3e-3
    # one ray and 33 rays through the whole differentiable render
    for R in (1, 33):
        o, d, near, far = [t[:R].contiguous() for t in _view(8, dev)]
        for p in list(sdf.parameters()) + list(col.parameters()) + list(var.parameters()):
            p.grad = None
        out = ren.render(o, d, near, far, background_rgb=torch.zeros(1, 3, device=dev), cos_anneal_ratio=1.0)
        assert out["color_fine"].shape == (R, 3) and out["weights"].shape == (R, 64)
        (out["color_fine"].sum() + out["gradient_error"]).backward()
        assert all(torch.isfinite(p.grad).all() for p in sdf.parameters())
    # no rays at all: empty outputs, no launch, no error
    e = torch.zeros(0, 3, device=dev)
    out = ren.render(e, e, torch.zeros(0, 1, device=dev), torch.zeros(0, 1, device=dev),
                     background_rgb=torch.zeros(1, 3, device=dev), cos_anneal_ratio=1.0)
    assert out["color_fine"].shape == (0, 3) and out["weights"].shape[0] == 0


@gpu
def test_128spp_render_matches_oracle_on_identical_depths():
    """BASELINE config 3 uses 128 samples per ray (64 + 64): same kernels, longer rows."""
    dev = torch.device("cuda")
    sdf, col, var, ren = _full_renderer(dev, 64, 64)
    ro, rd, near, far = [t.contiguous() for t in _view(12, dev)]
    R = ro.shape[0]
    sd_s = {k: v.detach().cpu() for k, v in sdf.named_parameters()}
    sd_c = {k: v.detach().cpu() for k, v in col.named_parameters()}
    jitter = torch.rand(R, 1, generator=torch.Generator().manual_seed(2))
    ref = O.render(sd_s, sd_c, var.variance.detach().cpu(), ro.cpu(), rd.cpu(), near.cpu(), far.cpu(), 64, 64, 4, jitter,
                   torch.zeros(1, 3), 1.0)
    z = ref["z_vals"] if "z_vals" in ref else None
    assert z is not None and z.shape == (R, 128)
    with torch.no_grad():
        out = ren.render(ro, rd, near, far, background_rgb=torch.zeros(1, 3, device=dev), cos_anneal_ratio=1.0,
                         z_vals=z.to(dev))
    for k in ("color_fine", "extra_color_fine", "weight_sum"):
        e = (out[k].cpu() - ref[k].detach()).abs()
        print(k, "max", e.max().item(), "mean", e.mean().item())
        assert e.max() < 2e-2 and e.mean() < 1.5e-3, k


@gpu
def test_render_without_extra_color_head_matches_oracle():
    """confs/base_models/*.conf (NeuS-init stage): RenderingNetwork without `extra_lin`, NeuSRenderer(extra_color=False):
    forward and parameter gradients vs the oracle on identical depths (small nets)."""
    from avatarclip_amd import fields, renderer
    dev = torch.device("cuda")
    torch.manual_seed(1)
    sdf = fields.SDFNetwork(d_out=129, d_in=3, d_hidden=128, n_layers=3, skip_in=[3], multires=6, bias=0.5, scale=1.0,
                            geometric_init=True, weight_norm=True).to(dev)
    col = fields.RenderingNetwork(d_feature=128, mode="no_view_dir", d_in=6, d_out=3, d_hidden=128, n_layers=1,
                                  weight_norm=True, multires_view=0, squeeze_out=True, extra_color=False).to(dev)
    var = fields.SingleVarianceNetwork(0.3).to(dev)
    ren = renderer.NeuSRenderer(None, sdf, var, col, 16, 16, 0, 4, 1.0, False)
    ro, rd, near, far = [t.contiguous() for t in _view(12, dev)]
    R = ro.shape[0]
    sd_s = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sdf.named_parameters()}
    sd_c = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in col.named_parameters()}
    vv = var.variance.detach().cpu().clone().requires_grad_(True)
    jitter = torch.rand(R, 1, generator=torch.Generator().manual_seed(4))
    ref = O.render(sd_s, sd_c, vv, ro.cpu(), rd.cpu(), near.cpu(), far.cpu(), 16, 16, 4, jitter, torch.ones(1, 3), 1.0,
                   extra_color=False)
    out = ren.render(ro, rd, near, far, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0,
                     z_vals=ref["z_vals"].detach().to(dev))
    e = (out["color_fine"].detach().cpu() - ref["color_fine"].detach()).abs()
    print("color max", e.max().item(), "mean", e.mean().item())
    assert e.max() < 2e-2 and e.mean() < 1.5e-3
    w = torch.randn(R, 3, generator=torch.Generator().manual_seed(6))
    (ref["color_fine"] * w).sum().add(ref["gradient_error"]).backward()
    (out["color_fine"] * w.to(dev)).sum().add(out["gradient_error"]).backward()
    for name, net, sd in (("sdf", sdf, sd_s), ("col", col, sd_c)):
        for n_, p in net.named_parameters():
            g_ref = sd[n_].grad
            if g_ref is None or g_ref.abs().max() < 1e-7:
                continue
            g = p.grad.detach().cpu()
            rel = ((g - g_ref).double().norm() / g_ref.double().norm()).item()
            assert rel < 3e-2, (name, n_, rel)
