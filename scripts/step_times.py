"""per-step wall times of the bench workload (synchronising after every step; development aid)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from avatarclip_amd.runner import Runner
from avatarclip_amd.smpl_prior import MeshPrior
dev = torch.device("cuda")
conf = bench.make_conf(512, 64, False)
r = Runner(None, mode="train_clip", conf=conf, device=dev)
r.init_clip()
z = np.load(os.path.join(bench.ROOT, "tests", "golden", "smpl_views.npz"))
r.init_smpl(MeshPrior(z["mesh_v"], z["mesh_f"], device=dev))
r.update_learning_rate()
ts = []
for i in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.train_clip_iteration(i); r.update_learning_rate()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms:", [round(t, 1) for t in ts])
t0 = time.perf_counter()
for i in range(14, 24):
    r.train_clip_iteration(i); r.update_learning_rate()
torch.cuda.synchronize()
print("10 steps unsynchronised: %.1f ms/step" % ((time.perf_counter() - t0) * 100))
