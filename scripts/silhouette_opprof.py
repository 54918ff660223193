"""torch.profiler view of the silhouette-mode iteration: which torch ops launch the small kernels of the glue (counts + device time per
op, grouped by the Runner stage that issues them).   python scripts/silhouette_opprof.py [max_ray_num]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from avatarclip_amd.runner import Runner
from torch.profiler import profile, ProfilerActivity, record_function

max_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
conf = bench.make_conf(512, 64, small=False)
conf.put("train.use_silhouettes", True); conf.put("train.max_ray_num", max_rays); conf.put("train.use_bg_aug", True)
torch.manual_seed(0); np.random.seed(0)
r = Runner(None, mode="train_clip", conf=conf, device=torch.device("cuda"))
r.init_clip(); r.init_smpl(); r.update_learning_rate()
for name in ("make_view", "draw_background", "shade_and_scatter", "assemble_loss"):
    fn = getattr(r, name)
    def wrap(fn=fn, name=name):
        def f(*a, **k):
            with record_function("STAGE_" + name):
                return fn(*a, **k)
        return f
    setattr(r, name, wrap())
_render = r.renderer.render
def render(*a, **k):
    with record_function("STAGE_render"):
        return _render(*a, **k)
r.renderer.render = render
for i in range(10):
    r.train_clip_iteration(i); r.update_learning_rate()
torch.cuda.synchronize()
N = 20
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for i in range(10, 10 + N):
        with record_function("STAGE_iteration"):
            r.train_clip_iteration(i); r.update_learning_rate()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
rows = sorted([e for e in ka if e.key.startswith("aten::")], key=lambda e: -e.self_device_time_total)
print("%-26s %7s %10s  %-60s %s" % ("op", "calls/it", "self us/it", "shapes", "stack"))
for e in rows[:60]:
    st = " <- ".join(x.split("/")[-1][:46] for x in (e.stack or [])[:3] if "avatarclip_amd" in x or "bench" in x or "scripts" in x)
    print("%-26s %7.1f %10.1f  %-60s %s" % (e.key[:26], e.count / N, e.self_device_time_total / N, str(e.input_shapes)[:60], st))
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.device_time_total)
print("%-60s %8s %12s %12s" % ("op", "calls/it", "dev us/it", "cpu us/it"))
for e in rows[:70]:
    print("%-60s %8.1f %12.1f %12.1f" % (e.key[:60], e.count / N, e.device_time_total / N, e.cpu_time_total / N))
