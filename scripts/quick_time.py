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
    if os.environ.get("QT_NOISE"):
        with torch.no_grad():
            for p_ in list(sdf.parameters()) + list(col.parameters()):
                p_.add_(torch.randn_like(p_) * float(os.environ["QT_NOISE"]))
    ren = renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)
    eye = np.array([0., 0.2, 1.5]); at = np.zeros(3)
    if os.environ.get("QT_RANDCAM"):
        from avatarclip_amd import utils as U
        np.random.seed(int(os.environ["QT_RANDCAM"]))
        e, th, ph, fr = U.random_eye_normal(); at = U.random_at(); eye = e + at
        print("random camera eye", eye, "at", at)
    pose = torch.from_numpy(O.lookat(eye, at, np.array([0., 1, 0]))).float()
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
    clip = None
    if os.environ.get("QT_CLIP"):
        from avatarclip_amd import clip_vit
        from avatarclip_amd.runner import clip_vit_random_state_dict
        clip = clip_vit.ClipVisionB32(clip_vit_random_state_dict(0), dev)
    for i in range(int(os.environ.get("QT_ITERS", "4"))):
        if clip is not None:
            for _ in range(2):
                img = torch.rand(1, 3, 224, 224, device=dev, requires_grad=True)
                clip.encode_image(img).sum().backward()
        if os.environ.get("QT_SLEEP"):
            torch.cuda.synchronize(); time.sleep(float(os.environ["QT_SLEEP"]))
        ms, l = step()
        print("res %d small %s iter %d: pack %.2f ms, sample %.2f ms, fwd %.2f ms, bwd %.2f ms, loss %.4f" % (res, small, i, *ms, l), flush=True)
    R = res * res
    print("rays/s (render fwd+bwd only): %.3e" % (R / (sum(ms) * 1e-3)))

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 224, len(sys.argv) > 2 and sys.argv[2] == "small")
