import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import fields, renderer
dev = torch.device("cuda"); torch.manual_seed(0)
sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6).to(dev)
col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, extra_color=True).to(dev)
var = fields.SingleVarianceNetwork(0.3).to(dev)
ren = renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)
eng = ren.engine; pk = eng.pack(ren.flat_params())
npts = 1 << 22; R = npts // 64
ro = torch.randn(R, 3, device=dev) * 0.1; rd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(R, 64, device=dev) * 2, dim=-1)[0].contiguous()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print(os.environ.get("AVC_LIB_NAME", "libavc.so"), "fwd %.3f ms  fwd_train %.3f ms" % (t(lambda: eng.points_fwd(pk, ro, rd, z, 2 / 32)), t(lambda: eng.points_fwd_train(pk, ro, rd, z, 2 / 32))))
