"""torch.profiler view of Runner.make_view alone (silhouette mode).   python scripts/makeview_opprof.py [max_ray_num] [H]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from avatarclip_amd.runner import Runner
from torch.profiler import profile, ProfilerActivity
max_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
conf = bench.make_conf(H, 64, small=False)
conf.put("train.use_silhouettes", True); conf.put("train.max_ray_num", max_rays); conf.put("train.use_bg_aug", True)
torch.manual_seed(0); np.random.seed(0)
r = Runner(None, mode="train_clip", conf=conf, device=torch.device("cuda"))
r.init_clip(); r.init_smpl(); r.update_learning_rate()
for i in range(5):
    r.make_view(i)
torch.cuda.synchronize()
N = 20
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for i in range(N):
        r.make_view(i)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
print("%-90s %8s %12s" % ("op / kernel", "calls/it", "dev us/it"))
for e in rows[:45]:
    print("%-90s %8.1f %12.1f" % (e.key[:90], e.count / N, e.device_time_total / N))
