"""Micro-benchmark of single kernels vs problem size (development aid)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import fields, renderer
dev = torch.device("cuda"); torch.manual_seed(0)
sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6).to(dev)
col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, extra_color=True).to(dev)
var = fields.SingleVarianceNetwork(0.3).to(dev)
ren = renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)
eng = ren.engine; pk = eng.pack(ren.flat_params())
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
sizes = [int(a) for a in sys.argv[1:]] or [2**14, 2**16, 2**17, 2**18, 2**20, 2**22]
for npts in sizes:
    R = npts // 64
    ro = torch.randn(R, 3, device=dev) * 0.1; rd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    z = torch.sort(torch.rand(R, 64, device=dev) * 2, dim=-1)[0].contiguous()
    ms_sdf = t(lambda: eng.sdf_rays(pk, ro, rd, z))
    ms_fwd = t(lambda: eng.points_fwd(pk, ro, rd, z, 2 / 32))
    dsdf = torch.randn(R, 64, device=dev); dn = torch.randn(R, 64, 3, device=dev) * 0.1; drgb = torch.randn(R, 64, 6, device=dev) * 0.1
    ms_fwt = t(lambda: eng.points_fwd_train(pk, ro, rd, z, 2 / 32))
    _, _, rgbf = eng.points_fwd_train(pk, ro, rd, z, 2 / 32)
    ms_bwd = t(lambda: eng.points_bwd(pk, ro, rd, z, 2 / 32, dsdf, dn, drgb, rgbf, panels_valid=True), reps=3)
    print("npts %8d  points_fwd_train %8.3f ms (%.1f TF/s)   points_bwd+weight_grad %8.3f ms" % (npts, ms_fwt, npts * 1186816 / ms_fwt / 1e9, ms_bwd), flush=True)
    print("npts %8d  sdf-only %8.3f ms (%.1f TF/s)   points_fwd %8.3f ms (%.1f TF/s)" % (npts, ms_sdf, npts * 393728 / ms_sdf / 1e9, ms_fwd, npts * 1186816 / ms_fwd / 1e9), flush=True)
