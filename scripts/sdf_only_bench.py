"""Run avc_sdf_forward (and optionally the render forward) a few times on NPTS points: target of the PMC passes."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import fields, renderer
dev = torch.device("cuda"); torch.manual_seed(0)
sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6).to(dev)
col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, extra_color=True).to(dev)
var = fields.SingleVarianceNetwork(0.3).to(dev)
ren = renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)
eng = ren.engine; pk = eng.pack(ren.flat_params())
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
what = sys.argv[2] if len(sys.argv) > 2 else "sdf"
R = npts // 64
ro = torch.randn(R, 3, device=dev) * 0.1; rd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(R, 64, device=dev) * 2, dim=-1)[0].contiguous()
for _ in range(3):
    if "sdf" in what: eng.sdf_rays(pk, ro, rd, z)
    if "fwd" in what: eng.points_fwd(pk, ro, rd, z, 2 / 32)
torch.cuda.synchronize()
