"""Cases of tests/test_gpu_ring.py, which runs this file in a process of its own with AVC_LIB_NAME=libavc_ring.so (the library of a pytest
session is loaded once, and the product's libavc.so does not contain the experiment).
GPU test of the role-specialised backward (csrc/avc_bwd_ring.hip): the abar tiles of the middle SDF layers handed to consumer
workgroups through the L2-resident ring must give the SAME dense gradient as the path through the G region (main.py:537 via
avc_render_points_bwd + avc_weight_grad_all): both contract the identical bf16 tiles in fp32, only the order of the partial sums
differs.  The plain path is itself pinned against the reference's autograd (tests/test_gpu_kernels.py)."""
import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu


def _nets(small, dev, seed=0):
    from avatarclip_amd import fields, renderer
    torch.manual_seed(seed)
    if small:
        sdf = fields.SDFNetwork(d_out=129, d_in=3, d_hidden=128, n_layers=3, skip_in=[3], multires=6)
        col = fields.RenderingNetwork(d_feature=128, mode="no_view_dir", d_in=6, d_out=3, d_hidden=128, n_layers=1, extra_color=True)
    else:
        sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6)
        col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, extra_color=True)
    var = fields.SingleVarianceNetwork(0.3)
    sdf, col, var = sdf.to(dev), col.to(dev), var.to(dev)
    return renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)


def _inputs(R, S, dev, seed=1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    ro = (torch.randn(R, 3, generator=g) * 0.1).to(dev)
    rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    z = torch.sort(torch.rand(R, S, generator=g) * 2, dim=-1)[0].contiguous().to(dev)
    dsdf = torch.randn(R, S, generator=g).to(dev)
    dn = (torch.randn(R, S, 3, generator=g) * 0.1).to(dev)
    drgb = (torch.randn(R, S, 6, generator=g) * 0.1).to(dev)
    return ro, rd, z, dsdf, dn, drgb


@gpu
@pytest.mark.parametrize("small,R,S,cpt,slots", [(True, 257, 48, 1, 2), (True, 4096, 64, 2, 6), (False, 300, 64, 2, 6), (False, 16384, 64, 2, 6),
                                                 (False, 16384, 64, 3, 3)])
def test_ring_backward_equals_the_panel_path(small, R, S, cpt, slots):
    from avatarclip_amd.engine import Engine
    dev = torch.device("cuda")
    ren = _nets(small, dev)
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    ro, rd, z, dsdf, dn, drgb = _inputs(R, S, dev)
    old = (Engine.RING, Engine.RING_CPT, Engine.RING_SLOTS, Engine.RING_CHECK)
    try:
        Engine.RING = False
        _, _, rgbf = eng.points_fwd_train(pk, ro, rd, z, 2 / 32)
        g0 = eng.points_bwd(pk, ro, rd, z, 2 / 32, dsdf, dn, drgb, rgbf, panels_valid=True).clone()
        Engine.RING, Engine.RING_CPT, Engine.RING_SLOTS, Engine.RING_CHECK = True, cpt, slots, True
        for rep in range(2):      # twice: the second launch reuses ring slots whose lines are still in this CU's caches
            _, _, rgbf = eng.points_fwd_train(pk, ro, rd, z, 2 / 32)
            g1 = eng.points_bwd(pk, ro, rd, z, 2 / 32, dsdf, dn, drgb, rgbf, panels_valid=True).clone()
            torch.cuda.synchronize()
            st = eng.ring_stats
            nblk = (R * S + 31) // 32
            assert int(st[6]) == (nblk + 7) // 8, "every 256-point group was claimed exactly once"
            assert int(st[4]) == eng.spec.NMID * ((nblk + 7) // 8), "every hand-off unit was contracted exactly once"
            assert torch.isfinite(g1).all()
            lay = eng.dl.lay
            for name, shape in lay.shapes:
                n = int(np.prod(shape))
                a = g0[lay.pbase[name]:lay.pbase[name] + n]
                b = g1[lay.pbase[name]:lay.pbase[name] + n]
                rel = float((a - b).norm() / (a.norm() + 1e-30))
                assert rel < 2e-5, (name, rel, rep)       # same products, different summation order (fp32)
    finally:
        Engine.RING, Engine.RING_CPT, Engine.RING_SLOTS, Engine.RING_CHECK = old
