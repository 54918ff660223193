"""Benchmark of the AvatarCLIP AppearanceGen hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is ONE full CLIP-guided optimisation iteration of Runner.train_clip (reference: main.py:345-566) on a
512x512 full-frame view per GPU, 64 samples/ray (32 coarse + 32 importance in 4 up-sampling steps), full-size
networks (confs/examples/*.conf), two CLIP ViT-B/32 passes (textured + untextured shading), all losses, backward
(incl. the double backward through the SDF normals), one RCCL gradient all-reduce, Adam.  Data is synthetic:
seeded geometric-init weights, seeded CLIP weights / text embeddings, procedural silhouette prior, random cameras
drawn with the reference's sampling code.

Prints ONE JSON line (rank 0).  value = whole-job rays/s (all ranks), weak scaling (one view per GPU per step).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic GEMM FLOPs per sample point, full net (SURVEY.md §8d / BASELINE.md §2)
F_SDF, F_GRAD, F_COL = 524800, 393728, 268288
F_PT = F_SDF + F_GRAD + F_COL
F_SDF_ONLY = 2 * (39 * 256 + 2 * 256 * 256 + 256 * 217 + 256)   # SDF value only (row 0 of the last layer)
PEAK_MFMA = 2.5e15   # dense bf16/f16 MFMA, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12    # HBM3E, MI355X_MICROARCH.md (6.3e12 achievable for a pure copy)
TILE = 64            # bytes per point of one 32-feature operand tile (2 KiB per 32-point block)
# The MLP kernels of one training step.  SURVEY.md 8d prices the whole MLP path against the dense bf16/f16 MFMA peak (every one of
# these kernels is a chain of dense contractions), so `frac` is ALWAYS algorithmic FLOPs / time / 2.5 PFLOP/s; what actually
# limits a kernel of this design is reported beside it (`limited_by`, `hbm_frac_counter` = PMC bytes / time / 8 TB/s,
# `traffic_ratio` = PMC bytes / algorithmic bytes).  Algorithmic bytes per point = the operand tiles a kernel must read once and
# write once (full nets: 172 tiles = 10.75 KiB per point in total: 89 written by the forward kernel, 83 by the backward kernel).
KERNEL_ROOFLINES = {
    "avc_sdf_forward": dict(kernel="mlp_sdf_kernel", limited_by="engine (MFMA + VALU + LDS issue of the register-resident MLP)", flop_per_point=F_SDF_ONLY,
                            bytes_per_point=4 + 4 + 1,
                            note="SDF value at the sampler's points (renderer.py:337-338,187): four f16 GEMM layers + an fp32 dot product, no HBM "
                                 "traffic to speak of (the PMC bytes are the scratch traffic of 14 spilled registers) -- the SDF-MLP GEMM north_star names"),
    "avc_render_points_fwd_train": dict(kernel="mlp_render_kernel<train>", limited_by="engine, with the forward-type tile stores riding along", flop_per_point=F_PT,
                                        bytes_per_point=89 * TILE + 76,
                                        note="differentiable forward (F_pt = 1 186 816 FLOP/point) + the 89 forward-type operand tiles"),
    "avc_render_points_bwd": dict(kernel="mlp_bwd_kernel", limited_by="hbm", flop_per_point=F_PT, bytes_per_point=(83 + 62) * TILE + 80,
                                  note="colour backward + second-order + reverse sweep (F_pt FLOP/point, no forward recompute): writes 83 "
                                       "gradient-type tiles, reads h and g_a (62 tiles); its own gbar_h / ybar / second h re-reads (63 tiles) are on "
                                       "top of the algorithmic bytes"),
    "avc_weight_grad(all pairs)": dict(kernel="weight_grad_all_kernel", limited_by="hbm", flop_per_point=F_PT, bytes_per_point=172 * TILE,
                                       note="dW = sum_points A^T B from the operand panels (F_pt FLOP/point): every tile is used by ONE product "
                                            "pair, so operands streamed from HBM cap the kernel at AI = 108 FLOP/B, below the 312 FLOP/B ridge"),
}


def make_conf(res, spp, small=False):
    from avatarclip_amd.conf import ConfigFactory
    H = 128 if small else 256
    text = """
general { base_exp_dir = /tmp/avc_bench_exp
          recording = []
          allow_standins = True }
dataset { H = %d
          W = %d }
train {
    learning_rate = 5e-4
    learning_rate_alpha = 0.05
    end_iter = 100000
    batch_size = 512
    validate_resolution_level = 1
    warm_up_end = 500
    anneal_end = 0
    use_white_bkgd = False
    save_freq = 100000000
    val_freq = 100000000
    val_mesh_freq = 100000000
    report_freq = 100000000
    igr_weight = 0.1
    mask_weight = 1.0
    clip_weight = 1.0
    add_no_texture = True
    texture_cast_light = True
    use_face_prompt = True
    use_back_prompt = True
    use_silhouettes = False
    full_frame_resolution_level = 1
    seed = 0
}
clip { prompt = a 3D rendering of the Iron Man in unreal engine }
model {
    sdf_network { d_out = %d
        d_in = 3
        d_hidden = %d
        n_layers = %d
        skip_in = [%d]
        multires = 6
        bias = 0.5
        scale = 1.0
        geometric_init = True
        weight_norm = True }
    variance_network { init_val = 0.3 }
    rendering_network { d_feature = %d
        mode = no_view_dir
        d_in = 6
        d_out = 3
        d_hidden = %d
        n_layers = %d
        weight_norm = True
        multires_view = 0
        squeeze_out = True
        extra_color = True }
    neus_renderer { n_samples = %d
        n_importance = %d
        n_outside = 0
        up_sample_steps = 4
        perturb = 1.0
        extra_color = True }
}
""" % (res, res, H + 1, H, 3 if small else 4, 3 if small else 4, H, H, 1 if small else 2, spp // 2, spp // 2)
    return ConfigFactory.parse_string(text)


def cpu_baseline(spp, threads=(8, 16, 32)):
    """the reference's step on the host CPUs, in the same run, as a REPORTED number.  When the reference tree
    is present (the build container; $AVATARCLIP_REFERENCE or /root/reference) its OWN modules are timed -- models/fields.py and
    models/renderer.py imported unmodified through oracle/ref_loader.py (`kind: "reference"`); a GPU box never holds that tree, so
    there the CPU oracle is timed (oracle/neus_oracle.py, pinned to the reference's outputs by tests/golden: `kind: "port"`).  Either
    way the step is the whole iteration: NeuS render of full-size nets -> shading -> 2 CLIP ViT-B/32 passes (oracle/clip_vit_oracle.py:
    OpenAI's package is not installable offline) -> losses -> backward -> Adam.
    Bounded (~1 min): a 3-point thread sweep on a 64 x 64 view (4 096 rays: 1 warm-up + 1 timed iteration per thread count, the best
    count stated and used from there on), one more timed 64 x 64 iteration, and ONE chunk of a 224 x 224 view (6 272 of its 50 176
    rays: the autograd graph of the whole view does not fit a 64 GB host, profiles/r06_cpu_reference_vs_port.md runs all 8 chunks)."""
    from oracle import neus_oracle as O, clip_vit_oracle as C, ref_loader
    use_ref = ref_loader.reference_available()
    host = os.cpu_count() or 1
    torch.manual_seed(0)
    if use_ref:
        RF = ref_loader.load_reference()
        sdf, col = RF.SDFNetwork(**ref_loader.FULL_SDF), RF.RenderingNetwork(**ref_loader.FULL_COLOR)
        var_net = RF.SingleVarianceNetwork(0.3)
        ren = RF.NeuSRenderer(None, sdf, var_net, col, spp // 2, spp // 2, 0, 4, 1.0, extra_color=True)
        params = list(sdf.parameters()) + list(var_net.parameters()) + list(col.parameters())
    else:
        from avatarclip_amd import fields
        sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6)
        col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, extra_color=True)
        var = torch.nn.Parameter(torch.tensor(0.3))
        params = list(sdf.parameters()) + [var] + list(col.parameters())
    opt = torch.optim.Adam(params, lr=5e-4)
    clip_sd = C.random_state_dict(0)
    text = torch.nn.functional.normalize(torch.randn(1, 512, generator=torch.Generator().manual_seed(11)), dim=-1)

    def view(side):
        pose = torch.from_numpy(O.lookat(np.array([0.3, 0.2, 1.5]), np.zeros(3), np.array([0., 1, 0]))).float()
        o, v = O.gen_rays_pose(pose, side, side, 0.5 * side / np.tan(np.pi / 6))
        ro, rd = o.reshape(-1, 3).contiguous(), v.reshape(-1, 3).contiguous()
        return (ro, rd) + tuple(O.near_far_from_sphere(ro, rd))

    def step(ro, rd, near, far, side, rows):
        """one iteration on `rows` image rows of a side x side view (all rows: the whole step; fewer: one gradient-accumulation chunk,
        with the CLIP passes run on the chunk's rows pasted into a black frame)"""
        t0 = time.time()
        n = rows * side
        ro, rd, near, far = ro[:n], rd[:n], near[:n], far[:n]
        if use_ref:
            out = ren.render(ro, rd, near, far, background_rgb=torch.zeros(1, 3), cos_anneal_ratio=1.0)      # renderer.py:302-397 itself
        else:
            out = O.render(dict(sdf.named_parameters()), dict(col.named_parameters()), var, ro, rd, near, far, spp // 2, spp // 2, 4,
                           torch.rand(n, 1), torch.zeros(1, 3), 1.0)
        tex, shade = O.cast_light(out, np.array([0.3, 0.5, 0.8]), 0.1)
        loss, _, _, _ = O.neus_losses(out, torch.zeros(n, 3), torch.ones(n, 1), 0.1, 1.0)
        for img in (tex, shade):
            frame = torch.cat([img.reshape(rows, side, 3), torch.zeros(side - rows, side, 3)], 0) if rows < side else img.reshape(side, side, 3)
            enc = C.encode_image(clip_sd, O.clip_preprocess(frame))
            loss = loss + (1.0 - O.clip_cosine(enc, text))
        opt.zero_grad()
        loss.backward()
        opt.step()
        return time.time() - t0

    v64 = view(64)
    sweep = {}
    for nt in threads:
        if nt > host and sweep:
            continue
        torch.set_num_threads(min(nt, host))
        step(*v64, 64, 64)                      # warm-up at this thread count
        sweep[min(nt, host)] = step(*v64, 64, 64)
        if sweep[min(nt, host)] > 60.0:         # a slow host: keep the default bench run within minutes
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t64 = 0.5 * (sweep[best] + step(*v64, 64, 64))
    res = {"value": 4096 / t64, "unit": "rays/s", "cores": best, "host_cores": host, "kind": "reference" if use_ref else "port",
           "sample": "64x64 view (4 096 rays) x %d spp, full-size nets, whole step incl. 2 CLIP ViT-B/32 passes + Adam, 2 timed iterations at the "
                     "best of the thread counts tried: %.2f s/iter" % (spp, t64),
           "thread_sweep_s_per_iter": {str(k): round(v, 3) for k, v in sweep.items()},
           "iters_per_sec_64x64": 1.0 / t64}
    if t64 < 30.0:
        v224 = view(224)
        t224 = step(*v224, 224, 28)
        res["sample_224"] = {"value": 28 * 224 / t224, "unit": "rays/s", "rays": 28 * 224,
                             "sample": "224x224 view (BASELINE config 2), rows 0-27 of 224 = chunk 1 of the 8 its autograd graph has to be cut "
                                       "into on a 64 GB host, 1 timed iteration: %.2f s" % t224,
                             "whole_view_s_per_iter_extrapolated": 8 * t224}
    res["reference_vs_port_table"] = "profiles/r06_cpu_reference_vs_port.md"
    return res


def load_pmc():
    """HBM traffic per point measured with the PMC counters (separate rocprofv3 --pmc passes of this command, summarised by
    scripts/pmc_summary.py --json into profiles/): 2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction.  Newest round first."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        with open(path) as fh:
            return json.load(fh), os.path.relpath(path, ROOT)
    return {}, None


def build_runner(res, spp, small, dev, overrides=None, mesh_prior=True):
    from avatarclip_amd.runner import Runner
    # identical initial weights on every rank (train.seed = 0 in the conf + broadcast); the Runner then re-seeds the DATA RNGs
    # per rank (a different camera view, jitter and light per rank: view-sharded DP, Runner.seed_data_rngs)
    conf = make_conf(res, spp, small)
    for k, v in (overrides or {}).items():
        conf.put(k, v)
    runner = Runner(None, mode="train_clip", conf=conf, device=dev)
    runner.init_clip()
    # the silhouette / colour prior: the SMPL template mesh (reference data/zero_beta_smpl.obj, packed in the test fixture) through
    # the HIP rasteriser with neural_renderer's conventions -- the same per-iteration work as main.py:360
    mesh_npz = os.path.join(ROOT, "tests", "golden", "smpl_views.npz")
    if mesh_prior and os.path.exists(mesh_npz):
        from avatarclip_amd.smpl_prior import MeshPrior
        z = np.load(mesh_npz)
        runner.init_smpl(MeshPrior(z["mesh_v"], z["mesh_f"], device=dev))
    else:
        runner.init_smpl()
    runner.update_learning_rate()
    return runner


def timed_steps(runner, steps, warmup, dev, sync_debug=False):
    """`warmup` untimed + `steps` timed train_clip iterations, bracketed by barrier + synchronize, max over ranks.
    Returns (seconds, per-kernel {name: [(points, seconds), ..]} from events recorded on the launch stream)."""
    from avatarclip_amd import parallel
    from avatarclip_amd.engine import Engine

    def step(i):
        runner.train_clip_iteration(i)
        runner.update_learning_rate()

    for i in range(warmup):
        step(i)
    Engine.PROFILE = True
    Engine.prof_events = []
    parallel.barrier()
    torch.cuda.synchronize()
    if sync_debug:
        torch.cuda.set_sync_debug_mode("warn")
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    if sync_debug:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)
    Engine.PROFILE = False
    per_kernel = {}
    for name, npts, e0, e1 in Engine.prof_events:
        per_kernel.setdefault(name, []).append((npts, e0.elapsed_time(e1) * 1e-3))
    Engine.prof_events = []
    return dt, per_kernel


def kernel_rooflines(per_kernel, steps):
    """per MLP kernel: frac = algorithmic FLOPs / kernel time / dense MFMA peak, with the HBM view beside it.  One
    "launch" below = the kernel's launches of ONE step taken together (the backward pair runs once per slab of the view)."""
    pmc, pmc_path = load_pmc()
    roofs = {}
    for name, spec in KERNEL_ROOFLINES.items():
        if name not in per_kernel:
            continue
        recs = per_kernel[name]
        t_step = float(np.sum([t for _, t in recs])) / steps            # seconds of this kernel per step
        pts_step = float(np.sum([n for n, _ in recs])) / steps          # points it processes per step
        flops, nbytes = spec["flop_per_point"] * pts_step, spec["bytes_per_point"] * pts_step
        tr = pmc.get("kernels", {}).get(spec["kernel"])
        traffic = tr["bytes_per_point"] * pts_step if tr else None
        roofs[name] = {
            "kernel": "%s (%s)" % (spec["kernel"], name), "bound": "mfma", "achieved": flops / t_step / 1e12, "peak": PEAK_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": flops / t_step / PEAK_MFMA,
            "limited_by": spec["limited_by"],
            "traffic": traffic, "traffic_source": ("%s (commit %s, %s)" % (pmc_path, pmc.get("commit"), pmc.get("box"))) if tr else None,
            "hbm_frac_counter": (traffic / t_step / PEAK_HBM) if tr else None,
            "hbm_frac_algorithmic": nbytes / t_step / PEAK_HBM,
            "traffic_ratio": (traffic / nbytes) if tr else None,
            "ms_per_step": t_step * 1e3, "points_per_step": pts_step, "launches_per_step": len(recs) / steps,
            "algorithmic_flop_per_point": spec["flop_per_point"], "algorithmic_bytes_per_point": spec["bytes_per_point"],
            "note": spec["note"]}
    return roofs


def step_flops(res, spp):
    """algorithmic FLOPs of the MLP work of one step: (S + I (k-1)/k) F_sdf' + (S + I) 3 F_pt per ray"""
    return res * res * ((spp // 2 + (spp // 2) * 3 // 4) * F_SDF_ONLY + spp * 3 * F_PT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--small", action="store_true", help="128-wide nets (confs/examples_small)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs (BASELINE configs 2 and 3-per-GPU)")
    ap.add_argument("--sync-debug", action="store_true", help="development aid: warn (with a stack) on every host-device synchronisation inside the timed steps")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU, RCCL over xGMI), exactly the way
        # the driver would (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)
        have = torch.cuda.device_count()
        if have < args.gpus and not os.environ.get("AVC_SINGLE_DEVICE"):
            sys.exit("bench.py: --gpus %d requested but only %d device(s) are visible (set AVC_SINGLE_DEVICE=1 AVC_DIST_BACKEND=gloo "
                     "to exercise the multi-rank path on one GPU)" % (args.gpus, have))
        import socket
        import subprocess
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if os.environ.get("AVC_SINGLE_DEVICE"):
            # N processes on ONE device (the 1-GPU development rig): with the default 4 hardware queues per process, 8 processes
            # oversubscribe the queue slots, the scheduler starts preempting wavefronts by context save / restore, and launches
            # died with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in 2 of 5 such runs.  One process per GPU -- the
            # product's layout -- never oversubscribes.
            env.setdefault("GPU_MAX_HW_QUEUES", "2")
        sys.exit(subprocess.call(cmd, env=env))

    from avatarclip_amd import parallel
    rank, world, local_rank = parallel.init_from_env()
    if world != max(args.gpus, 1):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE = %d (launch with torch.distributed.run --nproc-per-node == --gpus)" % (args.gpus, world))
    if os.environ.get("AVC_SINGLE_DEVICE"):      # development aid: all ranks on device 0 (see parallel.init_from_env)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = "none (single process, no process group)"
    ranks_seen = {"backend": None, "ranks": 1, "devices": sorted({local_rank})}
    if parallel.is_on():
        import torch.distributed as dist
        be = dist.get_backend()
        collective = "%s%s, %d rank(s)" % (be, " (RCCL)" if be == "nccl" else "", world)
        want = os.environ.get("AVC_ASSERT_DIST")
        assert not want or want == be, "process group backend %s, expected %s" % (be, want)
        # what the process group itself reports (not what was asked for): its size, and the device ordinal every rank ended up on
        devs = [None] * world
        dist.all_gather_object(devs, int(local_rank))
        ranks_seen = {"backend": be, "ranks": dist.get_world_size(), "devices": devs}

    torch.cuda.reset_peak_memory_stats(dev)
    free0, total0 = torch.cuda.mem_get_info(dev)
    runner = build_runner(args.res, args.spp, args.small, dev)
    dt, per_kernel = timed_steps(runner, args.steps, args.warmup, dev, args.sync_debug)
    # footprint of the headline run: the operand panels are sized from the memory that was FREE when the run started (Engine.plan:
    # min(AVC_PANEL_GIB, 80 % of the free HBM); the G slab is halved before anything else gives), so the numbers below change with it
    eng0 = runner.renderer.engine
    chunk0, slab0 = eng0.plan(args.res * args.res, args.spp)
    memory = {"peak_hbm_gib": torch.cuda.max_memory_allocated(dev) / 2 ** 30, "free_at_start_gib": free0 / 2 ** 30, "device_total_gib": total0 / 2 ** 30,
              "panel_budget_gib": min(eng0.PANEL_BYTES_BUDGET, free0 * 8 // 10) / 2 ** 30, "rays_per_chunk": chunk0, "rays_per_slab": slab0,
              "gradient_slabs_per_step": -(-args.res * args.res // max(1, slab0)), "forward_reruns_in_backward": chunk0 < args.res * args.res}
    del eng0

    out = None
    if rank == 0:
        rays_per_step = args.res * args.res * world
        roofs = kernel_rooflines(per_kernel, args.steps) if not args.small else {}
        dominant = max(roofs, key=lambda k: roofs[k]["ms_per_step"]) if roofs else None
        kern_ms = {k: 1e3 * float(np.sum([t for _, t in v])) / args.steps for k, v in per_kernel.items()}
        out = {
            "metric": "rays_per_sec_512x512_64spp_clip_guided_train_iter",
            "value": rays_per_step * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "iters_per_sec": args.steps / dt,
            "peak_hbm_gib": memory["peak_hbm_gib"],
            "memory": memory,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16 (forward) / bf16 (gradient sweeps) MFMA operands, fp32 accumulate",
            "data": "synthetic",
            "config": {"workload": "AppearanceGen train_clip iteration, %dx%d full-frame view per GPU, %d spp (%d+%d, 4 up-sample steps), "
                                   "%s nets, 2 CLIP ViT-B/32 passes, Adam, %d view(s)/step"
                                   % (args.res, args.res, args.spp, args.spp // 2, args.spp // 2,
                                      "small (128-wide)" if args.small else "full-size (256-wide)", world),
                       "parallelism": "view-sharded dp%d, one flat RCCL all-reduce/step" % world,
                       "collective": collective},
            "rccl_ranks_seen": ranks_seen["ranks"] if ranks_seen["backend"] == "nccl" else 0,
            "process_group": ranks_seen,
            "kernel_ms_per_step": kern_ms,
            # whole step against the MFMA roofline: algorithmic MLP FLOPs of the step (257.3 MFLOP/ray at 64 spp) / step time
            "step_mfma_frac": (step_flops(args.res, args.spp) * args.steps / dt / PEAK_MFMA) if not args.small else None,
            "roofline": roofs.get(dominant),
            "roofline_all": roofs,
        }
        # Some boxes of the pool sustain only ~3.9 TB/s of MIXED read + write HBM traffic (6 elsewhere); the backward kernel is the one
        # kernel of the step above that ceiling and then takes ~1.45 x as long as the weight-gradient launch instead of ~1.0 x
        # (profiles/r05_slow_box.md: same bytes by PMC, the waves wait at 2.3 GHz).  Say so in the line instead of leaving a riddle.
        kb, kw = kern_ms.get("avc_render_points_bwd"), kern_ms.get("avc_weight_grad(all pairs)")
        if kb and kw and kb > 1.25 * kw:
            out["box_note"] = ("mlp_bwd_kernel %.1f ms vs weight_grad_all_kernel %.1f ms: this box caps mixed read+write HBM traffic "
                               "(profiles/r05_slow_box.md); on the other boxes the two are equal and the step is ~%.0f ms shorter"
                               % (kb, kw, kb - kw))
    # ---- extra_configs (rank 0 of a single-GPU run): BASELINE config 2 (224^2, 64 spp) and config 3's per-GPU share (512^2, 128 spp)
    if world == 1 and not args.no_extra and not args.small and (args.res, args.spp) == (512, 64):
        import gc
        del runner
        gc.collect()
        torch.cuda.empty_cache()
        extras = {}
        torch.cuda.reset_peak_memory_stats(dev)
        for label, (res, spp) in (("C2_224x224_64spp", (224, 64)), ("C3_per_gpu_512x512_128spp", (512, 128))):
            try:
                r2 = build_runner(res, spp, False, dev)
                dt2, pk2 = timed_steps(r2, 5, 3, dev)
                eng = r2.renderer.engine
                chunk, slab = eng.plan(res * res, spp)
                extras[label] = {"value": res * res * 5 / dt2, "unit": "rays/s", "ms_per_step": 1e3 * dt2 / 5, "iters_per_sec": 5 / dt2, "steps": 5,
                                 "warmup": 3, "step_mfma_frac": step_flops(res, spp) * 5 / dt2 / PEAK_MFMA,
                                 "kernel_ms_per_step": {k: 1e3 * float(np.sum([t for _, t in v])) / 5 for k, v in pk2.items()},
                                 "rays_per_chunk": chunk, "rays_per_slab": slab, "forward_reruns_in_backward": chunk < res * res,
                                 "peak_hbm_gib": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
                del r2, eng
            except Exception as e:   # an extra leg must never take the headline down (e.g. out of memory at 128 spp)
                extras[label] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            gc.collect()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(dev)
        # the reference's DEFAULT sampling mode (confs/examples/*.conf: use_silhouettes, max_ray_num = 7000, background augmentation): a ragged,
        # data-dependent ray set per iteration -- a reported extra, not the metric (profiles/r06_silhouette_mode.txt says where its time goes)
        try:
            # (the procedural ellipsoid prior, as scripts/silhouette_time.py: the T-pose template mesh of the headline run leaves the face
            # cameras of main.py:349-352 looking past it -- an empty prior, where the reference fails as well, dataset.py:253-254)
            r3 = build_runner(512, 64, False, dev, {"train.use_silhouettes": True, "train.max_ray_num": 7000, "train.use_bg_aug": True}, mesh_prior=False)
            for i in range(8):
                r3.train_clip_iteration(i); r3.update_learning_rate()
            torch.cuda.synchronize()
            t0, rays, n3 = time.perf_counter(), 0, 30
            for i in range(8, 8 + n3):
                r3.train_clip_iteration(i); r3.update_learning_rate(); rays += int(r3.last_stats["rays"])
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            extras["default_silhouette_mode_7000rays_64spp"] = {"value": rays / dt3, "unit": "rays/s", "ms_per_step": 1e3 * dt3 / n3, "iters_per_sec": n3 / dt3,
                                                                "steps": n3, "warmup": 8, "rays_per_step": rays / n3}
            del r3
        except Exception as e:
            extras["default_silhouette_mode_7000rays_64spp"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        gc.collect()
        torch.cuda.empty_cache()
        out["extra_configs"] = extras
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.spp)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    parallel.barrier()
    if parallel.is_on():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
