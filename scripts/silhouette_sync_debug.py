"""Where does the host wait in silhouette mode?  Prints every host-device synchronisation (with its stack) of three iterations and a
host-side timing of the stages.   python scripts/silhouette_sync_debug.py [max_ray_num]"""
import os, sys, time, warnings, traceback
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from avatarclip_amd.runner import Runner

max_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
conf = bench.make_conf(512, 64, small=False)
conf.put("train.use_silhouettes", True); conf.put("train.max_ray_num", max_rays); conf.put("train.use_bg_aug", True)
torch.manual_seed(0); np.random.seed(0)
r = Runner(None, mode="train_clip", conf=conf, device=torch.device("cuda"))
r.init_clip(); r.init_smpl(); r.update_learning_rate()
for i in range(6):
    r.train_clip_iteration(i); r.update_learning_rate()
torch.cuda.synchronize()
seen = {}
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/root/repo/" in f.filename or "GRAFT" in f.filename or "avatarclip_amd" in f.filename]
    key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-4:]))
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = showwarning
torch.cuda.set_sync_debug_mode("warn")
for i in range(6, 9):
    r.train_clip_iteration(i); r.update_learning_rate()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("synchronisations in 3 iterations:")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("  %3d x  %s" % (v, k))
# host-side stage times (each stage followed by a synchronise: GPU time included, but it shows where the wall time sits)
def stage(fn):
    torch.cuda.synchronize(); t = time.time(); out = fn(); h = time.time() - t; torch.cuda.synchronize(); return out, h, time.time() - t
tot = {}
for i in range(9, 19):
    view, h, w = stage(lambda: r.make_view(i)); tot.setdefault("make_view", []).append((h, w))
    (c, bg, mbg), h, w = stage(lambda: r.draw_background(view)); tot.setdefault("draw_background", []).append((h, w))
    ro, h, w = stage(lambda: r.renderer.render(view.rays_o, view.rays_d, view.near, view.far, background_rgb=mbg, cos_anneal_ratio=r.get_cos_anneal_ratio())); tot.setdefault("render", []).append((h, w))
    comp, h, w = stage(lambda: r.shade_and_scatter(ro, view, c, bg)); tot.setdefault("shade_and_scatter", []).append((h, w))
    (loss, parts), h, w = stage(lambda: r.assemble_loss(ro, comp, view, i)); tot.setdefault("assemble_loss(+CLIP fwd)", []).append((h, w))
    r.optimizer.zero_grad(set_to_none=True)
    _, h, w = stage(lambda: loss.backward()); tot.setdefault("backward", []).append((h, w))
    _, h, w = stage(lambda: r.optimizer.step()); tot.setdefault("adam", []).append((h, w))
for k, v in tot.items():
    print("  %-26s host %.2f ms   host+GPU %.2f ms" % (k, 1e3 * np.mean([a for a, b in v]), 1e3 * np.mean([b for a, b in v])))
