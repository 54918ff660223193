"""child of tests/test_gpu_kernels.py::test_sdf_kernel_with_64_points_per_wavefront_equals_the_default_kernel: avc_sdf_forward on seeded
points with the kernel variant $AVC_SDF_POINTS_PER_WAVE selects (read once per process by the launcher); writes the values to argv[1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.ring_cases import _nets  # noqa: E402

dev = torch.device("cuda")
out = {}
for small in (True, False):
    ren = _nets(small, dev, seed=3)
    with torch.no_grad():
        for p in ren.sdf_network.parameters():
            p.add_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).to(dev) * 0.02)
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    for R, S in ((4099, 32), (333, 7), (1, 1), (65, 64)):         # ragged: neither a multiple of 32 nor of 64 points; a single point
        g = torch.Generator().manual_seed(R * 100 + S)
        ro = (torch.randn(R, 3, generator=g) * 0.2).to(dev)
        rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
        z = torch.sort(torch.rand(R, S, generator=g) * 2, dim=-1)[0].contiguous().to(dev)
        out["rays_%d_%d_%d" % (small, R, S)] = eng.sdf_rays(pk, ro, rd, z).cpu()
        # the scatter form of cat_z_vals (slot != NULL) and the point form
        slot = torch.stack([torch.randperm(S + 3, generator=g)[:S] for _ in range(R)]).int().to(dev)
        dst = torch.zeros(R, S + 3, device=dev)
        eng.sdf_rays(pk, ro, rd, z, sdf_out=dst, slot=slot, ld_out=S + 3)
        out["slot_%d_%d_%d" % (small, R, S)] = dst.cpu()
        pts = (ro[:, None, :] + rd[:, None, :] * z[..., None]).reshape(-1, 3)
        out["pts_%d_%d_%d" % (small, R, S)] = eng.sdf_pts(pk, pts).cpu()
torch.cuda.synchronize()
torch.save(out, sys.argv[1])
