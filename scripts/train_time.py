"""Time `Runner.train` (NeuS initialisation on a rendered dataset, main.py:189-256; BASELINE config 1's loop) on the shipped stand-pose
views: batch_size random rays of one image per iteration.   python scripts/train_time.py [batch_size] [iters]"""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tests.test_gpu_dataset_train import unpack_dataset
from avatarclip_amd.runner import Runner

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tmp = tempfile.mkdtemp()
unpack_dataset("stand", os.path.join(tmp, "data"))
conf = bench.make_conf(256, 64, small=False)
conf.put("general.base_exp_dir", os.path.join(tmp, "exp")); conf.put("dataset.data_dir", os.path.join(tmp, "data"))
conf.put("train.batch_size", bs); conf.put("train.mask_weight", 0.5); conf.put("train.end_iter", 10 ** 6)
conf.put("model.rendering_network.extra_color", False); conf.put("model.neus_renderer.extra_color", False)
torch.manual_seed(0)
r = Runner(None, mode="train", conf=conf, device=torch.device("cuda"))
r.update_learning_rate()
perm = r.get_image_perm()
def it():
    data = r.dataset.gen_random_rays_at(perm[r.iter_step % len(perm)], r.batch_size)
    r.train_iteration(data); r.update_learning_rate()
for _ in range(20): it()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(iters): it()
torch.cuda.synchronize(); dt = (time.time() - t0) / iters
print("Runner.train: batch_size %d rays x 64 spp, full nets: %.2f ms per iteration = %.0f rays/s" % (bs, dt * 1e3, bs / dt))
