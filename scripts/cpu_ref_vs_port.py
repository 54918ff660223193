"""Build-container measurement (needs /root/reference): the reference's OWN modules (models/fields.py + models/renderer.py,
imported unmodified through oracle/ref_loader.py) against the CPU port (oracle/neus_oracle.py) on the same rays, weights and
jitter: one NeuS render + loss + backward + Adam step of the full-size networks at 64 x 64 and 224 x 224 rays, 64 spp.
(The CLIP term is left out on both sides: the `clip` package is absent; it is <= 0.3 % of the work)
Round 6: a thread sweep at 64 x 64 (AVC_CPU_THREADS = comma list, default 8,16,32 clipped to the host's cores), the 224 x 224 view at the
best count.     python scripts/cpu_ref_vs_port.py > profiles/r06_cpu_reference_vs_port.md"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import neus_oracle as O, ref_loader

host = os.cpu_count() or 1
sweep = sorted({min(int(t), host) for t in os.environ.get("AVC_CPU_THREADS", "8,16,32").split(",")})
R = ref_loader.load_reference()


def build():
    torch.manual_seed(0)
    sdf = R.SDFNetwork(**ref_loader.FULL_SDF)
    col = R.RenderingNetwork(**ref_loader.FULL_COLOR)
    var = R.SingleVarianceNetwork(0.3)
    return sdf, col, var


def rays(side):
    pose = torch.from_numpy(O.lookat(np.array([0.3, 0.2, 1.5]), np.zeros(3), np.array([0., 1, 0]))).float()
    o, v = O.gen_rays_pose(pose, side, side, 0.5 * side / np.tan(np.pi / 6))
    ro, rd = o.reshape(-1, 3).contiguous(), v.reshape(-1, 3).contiguous()
    near, far = O.near_far_from_sphere(ro, rd)
    return ro, rd, near, far


def loss_of(out, Rn):
    mask = torch.ones(Rn, 1)
    color_error = out["color_fine"] * mask
    l = torch.nn.functional.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / (mask.sum() + 1e-5)
    l = l + 0.1 * out["gradient_error"] + torch.nn.functional.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1 - 1e-3), mask)
    return l + out["extra_color_fine"].mean()


print("host cores: %d; thread counts tried at 64 x 64: %s; torch %s\n" % (host, sweep, torch.__version__))
print("| rays | leg | s / iteration | rays/s | threads | loss |")
print("|---|---|---|---|---|---|")
best = {}
for side, iters, nthreads in [(64, 2, t) for t in sweep] + [(224, 1, None)]:
    ro, rd, near, far = rays(side)
    Rn = ro.shape[0]
    for leg in ("reference modules", "port (oracle/neus_oracle.py)"):
        nt = nthreads if nthreads is not None else min(best[leg], key=best[leg].get)
        torch.set_num_threads(nt)
        sdf, col, var = build()
        params = list(sdf.parameters()) + list(var.parameters()) + list(col.parameters())
        opt = torch.optim.Adam(params, lr=5e-4)
        ren = R.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, extra_color=True)
        ts = []
        # 224 x 224 x 64 spp does not fit the 62 GB of the build container in one autograd graph: 8 ray chunks with accumulated
        # gradients on both legs (the arithmetic per ray is unchanged)
        nchunk = 1 if side == 64 else 8
        for it in range(iters + (1 if side == 64 else 0)):
            torch.manual_seed(100 + it)                 # the reference draws its jitter with torch.rand inside render()
            t0 = time.time()
            opt.zero_grad()
            for c in range(nchunk):
                sl = slice(c * Rn // nchunk, (c + 1) * Rn // nchunk)
                n = sl.stop - sl.start
                if leg.startswith("reference"):
                    out = ren.render(ro[sl], rd[sl], near[sl], far[sl], background_rgb=torch.zeros(1, 3), cos_anneal_ratio=1.0)
                else:
                    jitter = torch.rand(n, 1)
                    out = O.render(dict(sdf.named_parameters()), dict(col.named_parameters()), var.variance, ro[sl], rd[sl], near[sl], far[sl],
                                   32, 32, 4, jitter, torch.zeros(1, 3), 1.0)
                loss = loss_of(out, n) / nchunk
                loss.backward()
            opt.step()
            ts.append(time.time() - t0)
        t = float(np.mean(ts[1:])) if len(ts) > 1 else ts[0]
        if side == 64:
            best.setdefault(leg, {})[nt] = t
        print("| %d (%dx%d) | %s | %.2f | %.0f | %d | %.6f |" % (Rn, side, side, leg, t, Rn / t, nt, loss.item()), flush=True)
