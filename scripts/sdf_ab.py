"""avc_sdf_forward alone at the sampler's launch sizes (512^2 x 32 coarse points, 512^2 x 8 per up-sampling step) and at 4 Mi points:
average launch time, fraction of the dense f16 MFMA peak (393 728 FLOP per point, 2.5 PFLOP/s) and a checksum of the values, for the
kernel variant AVC_SDF_POINTS_PER_WAVE selects (32: mlp_sdf_kernel, 64: mlp_sdf2_kernel).  python scripts/sdf_ab.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import fields, renderer
dev = torch.device("cuda"); torch.manual_seed(0)
small = os.environ.get("SDF_AB_SMALL") == "1"
if small:
    sdf = fields.SDFNetwork(d_out=129, d_in=3, d_hidden=128, n_layers=3, skip_in=[3], multires=6).to(dev)
else:
    sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6).to(dev)
with torch.no_grad():
    for p in sdf.parameters():
        p.add_(torch.randn_like(p) * 0.02)
ren = renderer.NeuSRenderer(None, sdf, fields.SingleVarianceNetwork(0.3).to(dev), None, 32, 32, 0, 4, 1.0, True)
eng = ren.engine; pk = eng.pack(ren.flat_params())
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flop = 393728 if not small else None
out = []
for R, S in ((512 * 512, 32), (512 * 512, 8), (65536, 64), (1000, 7)):
    g = torch.Generator(device=dev).manual_seed(R + S)
    ro = torch.randn(R, 3, device=dev, generator=g) * 0.1
    rd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=-1)
    z = torch.sort(torch.rand(R, S, device=dev, generator=g) * 2, dim=-1)[0].contiguous()
    for _ in range(3):
        v = eng.sdf_rays(pk, ro, rd, z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        v = eng.sdf_rays(pk, ro, rd, z)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    frac = (R * S * flop / (ms * 1e-3) / 2.5e15) if flop else float("nan")
    out.append("%dx%d %.4f ms frac %.4f sum %.9e absmax %.6e" % (R, S, ms, frac, v.double().sum().item(), v.abs().max().item()))
    torch.save(v.cpu(), "/tmp/sdf_ab_%s_%d_%d.pt" % (os.environ.get("AVC_SDF_POINTS_PER_WAVE", "32"), R, S))
print("ppw", os.environ.get("AVC_SDF_POINTS_PER_WAVE", "32"), os.environ.get("AVC_LIB_NAME", "libavc.so"), " | ".join(out))
