"""Time the reference's DEFAULT sampling mode (confs/examples/*.conf: use_silhouettes = True, max_ray_num = 7000, background
augmentation on): a ragged ray set inside the dilated prior silhouette per iteration (main.py:360-375).  Development aid -- the
benchmark of record is bench.py's full-frame step.   python scripts/silhouette_time.py [max_ray_num] [H] [iters]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from avatarclip_amd.runner import Runner

max_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
conf = bench.make_conf(H, 64, small=False)
conf.put("train.use_silhouettes", True)
conf.put("train.max_ray_num", max_rays)
conf.put("train.use_bg_aug", True)
torch.manual_seed(0); np.random.seed(0)
r = Runner(None, mode="train_clip", conf=conf, device=torch.device("cuda"))
r.init_clip(); r.init_smpl(); r.update_learning_rate()
for i in range(8):
    r.train_clip_iteration(i); r.update_learning_rate()
torch.cuda.synchronize(); t0 = time.time(); rays = 0
for i in range(8, 8 + iters):
    r.train_clip_iteration(i); r.update_learning_rate(); rays += int(r.last_stats["rays"])
torch.cuda.synchronize(); dt = (time.time() - t0) / iters
print("silhouette mode: max_ray_num %d, dataset %dx%d, 64 spp, full nets: %.2f ms per iteration, %.0f rays per iteration on average, %.0f rays/s"
      % (max_rays, H, H, dt * 1e3, rays / iters, rays / iters / dt))
