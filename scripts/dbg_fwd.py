import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_kernels import _setup
for name in ("neus_small.npz", "neus_full.npz"):
    rec, sd_sdf, sd_col, variance, sdf, col, var, ren, dev = _setup(name)
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    ro, rd, z = rec["rays_o"].to(dev), rec["rays_d"].to(dev), rec["z_final"].to(dev).contiguous()
    a = eng.points_fwd(pk, ro, rd, z, 2.0 / 32)
    b = eng.points_fwd_train(pk, ro, rd, z, 2.0 / 32)
    for nm, x, y in zip(("sdf", "normal", "rgb"), a, b):
        print(name, nm, "max |train - plain| =", (x - y).abs().max().item(), "equal:", torch.equal(x, y))
