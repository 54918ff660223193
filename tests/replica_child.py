"""Child process of tests/test_gpu_parallel.py::test_two_replicas_equal_their_solo_runs: one independent train_clip run (small nets,
its own seed = its own prompt stand-in, cameras, jitter), a few iterations, the losses as JSON.
    python tests/replica_child.py <seed> <iterations> <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    seed, iters, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    assert "RANK" not in os.environ and "WORLD_SIZE" not in os.environ, "a replica is not a rank"
    import bench
    from avatarclip_amd import parallel
    from avatarclip_amd.runner import Runner, EllipsoidPrior, clip_vit_random_state_dict
    assert not parallel.is_on()
    conf = bench.make_conf(48, 32, small=True)
    conf.put("train.seed", seed)
    conf.put("train.warm_up_end", 0)
    conf.put("general.base_exp_dir", "/tmp/avc_replica_%d_%d" % (seed, os.getpid()))
    dev = torch.device("cuda", 0)        # the launcher made this replica's GPU the only visible device
    r = Runner(None, mode="train_clip", conf=conf, device=dev)
    assert r.world == 1 and r.grad_bucket is None
    r.init_clip(clip_state_dict=clip_vit_random_state_dict(0))
    r.init_smpl(EllipsoidPrior(device=dev))
    r.update_learning_rate()
    losses = []
    for i in range(iters):
        losses.append(float(r.train_clip_iteration(i)))
        r.update_learning_rate()
    torch.cuda.synchronize()
    json.dump({"seed": seed, "losses": losses, "visible": os.environ.get("HIP_VISIBLE_DEVICES"), "replica": os.environ.get("AVC_REPLICA"),
               "cores": len(os.sched_getaffinity(0))}, open(out, "w"))


if __name__ == "__main__":
    main()
