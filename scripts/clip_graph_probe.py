"""Probe: CLIP encode_image forward + backward of the per-iteration shape ([2,3,224,224]) as HIP graphs (torch.cuda.make_graphed_callables)
against eager launches: results and host time per call.   python scripts/clip_graph_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import clip_vit as V
from avatarclip_amd.runner import clip_vit_random_state_dict

m = V.ClipVisionB32(clip_vit_random_state_dict(0), "cuda")
x = torch.randn(2, 3, 224, 224, device="cuda", requires_grad=True)
text = torch.randn(1, 512, device="cuda")
def run(fn, x, n=50):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        e = fn(x)
        loss = 1 - torch.cosine_similarity(e.mean(0), text.mean(0), dim=0)
        g, = torch.autograd.grad(loss, x)
    th = time.time() - t0
    torch.cuda.synchronize()
    return e.detach().clone(), g.detach().clone(), th / n, (time.time() - t0) / n
e0, g0, h0, w0 = run(m.encode_image, x)
e0, g0, h0, w0 = run(m.encode_image, x)
print("eager : host %.2f ms  wall %.2f ms per fwd+bwd" % (h0 * 1e3, w0 * 1e3))
xs = torch.randn(2, 3, 224, 224, device="cuda", requires_grad=True)
graphed = torch.cuda.make_graphed_callables(m.encode_image, (xs,))
e1, g1, h1, w1 = run(graphed, x)
e1, g1, h1, w1 = run(graphed, x)
print("graphs: host %.2f ms  wall %.2f ms per fwd+bwd" % (h1 * 1e3, w1 * 1e3))
print("max |d emb| %.3e  max |d grad| %.3e (rel %.3e)" % ((e1 - e0).abs().max().item(), (g1 - g0).abs().max().item(), ((g1 - g0).norm() / g0.norm()).item()))
