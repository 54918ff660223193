"""Ad-hoc timing of the render fwd/bwd path on one GPU (development aid, not the bench)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import fields, renderer
from oracle import neus_oracle as O

def main(res=224, small=False):
    dev = torch.device("cuda")
    torch.manual_seed(0)
    if small:
        sdf = fields.SDFNetwork(d_out=129, d_in=3, d_hidden=128, n_layers=3, skip_in=[3], multires=6)
        col = fields.RenderingNetwork(d_feature=128, mode="no_view_dir", d_in=6, d_out=3, d_hidden=128, n_layers=1, extra_color=True)
    else:
        sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6)
        col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, extra_color=True)
    var = fields.SingleVarianceNetwork(0.3)
    sdf, col, var = sdf.to(dev), col.to(dev), var.to(dev)
    ren = renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)
    pose = torch.from_numpy(O.lookat(np.array([0., 0.2, 1.5]), np.zeros(3), np.array([0., 1, 0]))).float()
    o, v = O.gen_rays_pose(pose, res, res, 0.5 * res / np.tan(np.pi / 6))
    ro, rd = o.reshape(-1, 3).contiguous().to(dev), v.reshape(-1, 3).contiguous().to(dev)
    near, far = O.near_far_from_sphere(ro, rd)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    def step():
        t = [ev() for _ in range(6)]
        t[0].record()
        flat = ren.flat_params(); pk = ren.engine.pack(flat)
        t[1].record()
        z = ren.sample_z(pk, ro, rd, near, far, 1.0)
        t[2].record()
        ret = ren.render_core(ro, rd, z, 2.0 / 32, torch.zeros(1, 3, device=dev), 1.0, flat)
        t[3].record()
        loss = ret["color"].mean() + ret["extra_color"].mean() + 0.1 * ret["gradient_error"] + ret["weights"].sum(-1).mean()
        loss.backward()
        t[4].record()
        torch.cuda.synchronize()
        return [t[i].elapsed_time(t[i + 1]) for i in range(4)], loss.item()
    for i in range(4):
        ms, l = step()
        print("res %d small %s iter %d: pack %.2f ms, sample %.2f ms, fwd %.2f ms, bwd %.2f ms, loss %.4f" % (res, small, i, *ms, l), flush=True)
    R = res * res
    print("rays/s (render fwd+bwd only): %.3e" % (R / (sum(ms) * 1e-3)))

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 224, len(sys.argv) > 2 and sys.argv[2] == "small")
