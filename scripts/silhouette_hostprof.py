"""Host-side profile (cProfile) of the silhouette-mode iteration -- the mode every conf of the reference uses (confs/examples/*.conf:
use_silhouettes = True), bound by the host's launch rate at 7 000 - 12 544 rays.   python scripts/silhouette_hostprof.py [max_ray_num] [H]"""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from avatarclip_amd.runner import Runner

max_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
conf = bench.make_conf(H, 64, small=False)
conf.put("train.use_silhouettes", True); conf.put("train.max_ray_num", max_rays); conf.put("train.use_bg_aug", True)
torch.manual_seed(0); np.random.seed(0)
r = Runner(None, mode="train_clip", conf=conf, device=torch.device("cuda"))
r.init_clip(); r.init_smpl(); r.update_learning_rate()
for i in range(10):
    r.train_clip_iteration(i); r.update_learning_rate()
torch.cuda.synchronize()
N = 60
t0 = time.time()
for i in range(10, 10 + N):
    r.train_clip_iteration(i); r.update_learning_rate()
torch.cuda.synchronize()
print("unprofiled: %.2f ms per iteration" % ((time.time() - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(100, 100 + N):
    r.train_clip_iteration(i); r.update_learning_rate()
torch.cuda.synchronize()
pr.disable()
for key, n in (("cumulative", 45), ("tottime", 35)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(n)
    txt = s.getvalue()
    print("\n".join(l[:170] for l in txt.splitlines()[4:]))
