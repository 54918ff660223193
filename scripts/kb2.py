"""per-kernel times of the training path on NPTS points (development aid): python scripts/kb2.py [npts]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import fields, renderer
from avatarclip_amd.engine import Engine
dev = torch.device("cuda"); torch.manual_seed(0)
if os.environ.get("KB_MAX_FWD_WAVES"):   # a variant built with -DFWD_WPB=w needs 256 x w (one workgroup per CU)
    Engine.MAX_FWD_WAVES = int(os.environ["KB_MAX_FWD_WAVES"])
if os.environ.get("KB_MAX_BWD_WAVES"):   # likewise -DBWD_WPB=w
    Engine.MAX_BWD_WAVES = int(os.environ["KB_MAX_BWD_WAVES"])
sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6).to(dev)
col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, extra_color=True).to(dev)
var = fields.SingleVarianceNetwork(0.3).to(dev)
ren = renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)
eng = ren.engine; pk = eng.pack(ren.flat_params())
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
R = npts // 64
ro = torch.randn(R, 3, device=dev) * 0.1; rd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(R, 64, device=dev) * 2, dim=-1)[0].contiguous()
dsdf = torch.randn(R, 64, device=dev); dn = torch.randn(R, 64, 3, device=dev) * 0.1; drgb = torch.randn(R, 64, 6, device=dev) * 0.1
for rep in range(2):
    Engine.PROFILE = rep == 1
    Engine.prof_events = []
    for _ in range(3):
        eng.sdf_rays(pk, ro, rd, z)
        eng.points_fwd(pk, ro, rd, z, 2 / 32)
        _, _, rgbf = eng.points_fwd_train(pk, ro, rd, z, 2 / 32)
        eng.points_bwd(pk, ro, rd, z, 2 / 32, dsdf, dn, drgb, rgbf, panels_valid=True)
    torch.cuda.synchronize()
acc = {}
for name, n, e0, e1 in Engine.prof_events:
    acc.setdefault(name, []).append(e0.elapsed_time(e1))
print(os.environ.get("AVC_LIB_NAME", "libavc.so"), "npts", npts, " ".join("%s %.3f" % (k.replace("avc_", ""), sum(v) / len(v)) for k, v in acc.items()))
