"""CLI with the reference's flags (AvatarGen/AppearanceGen/main.py:947-980):
   python -m avatarclip_amd.main --mode {train,train_clip,validate_mesh,render_geometry_cast_light,interpolate_i_j} --conf X [--is_continue] [--gpu N] [--case NAME]
Multi-GPU (view-sharded): torchrun --nproc-per-node N -m avatarclip_amd.main --mode train_clip --conf X"""
import argparse
import logging

import torch

from . import parallel
from .runner import Runner


def main():
    logging.basicConfig(level=logging.INFO, format="[%(filename)s:%(lineno)d] %(levelname)s %(message)s")
    parser = argparse.ArgumentParser()
    parser.add_argument("--conf", type=str, default="./confs/base.conf")
    parser.add_argument("--mode", type=str, default="train")
    parser.add_argument("--mcube_threshold", type=float, default=0.0)
    parser.add_argument("--is_continue", default=False, action="store_true")
    parser.add_argument("--gpu", type=int, default=0)
    parser.add_argument("--case", type=str, default="")
    parser.add_argument("--clip_weights", type=str, default=None, help="OpenAI ViT-B-32 state dict / .pt")
    parser.add_argument("--smpl_mesh", type=str, default=None, help="posed SMPL mesh (.obj) rendered by the HIP rasteriser as the prior")
    parser.add_argument("--smpl_prior", type=str, default=None, help="module:callable(runner) -> prior_renderer(eye, at)")
    parser.add_argument("--allow_standins", default=False, action="store_true",
                        help="benchmarks / smoke runs only: seeded CLIP weights and text embeddings, ellipsoid prior, "
                             "geometric init when train.pretrain is missing")
    args = parser.parse_args()
    rank, world, local_rank = parallel.init_from_env()
    torch.cuda.set_device(local_rank if world > 1 else args.gpu)
    if args.mode in ("validate_mesh", "render_geometry_cast_light"):
        args.is_continue = True
    runner = Runner(args.conf, args.mode, args.case, args.is_continue, allow_standins=args.allow_standins or None)
    if args.mode == "validate_mesh":   # main.py:972-974
        runner.validate_mesh(world_space=True, resolution=512, threshold=args.mcube_threshold)
        runner.render_geometry_cast_light()
    elif args.mode == "render_geometry_cast_light":
        runner.render_geometry_cast_light()
    elif args.mode == "train":
        runner.train()
    elif args.mode == "train_clip":
        if args.clip_weights is not None:
            runner.conf.put("clip.weights", args.clip_weights)
        if args.smpl_mesh is not None:
            runner.conf.put("general.smpl_mesh", args.smpl_mesh)
        if args.smpl_prior is not None:
            runner.conf.put("general.smpl_prior", args.smpl_prior)
        runner.init_clip()
        runner.init_smpl()
        runner.train_clip()
    elif args.mode.startswith("interpolate_"):   # NeuS-style "interpolate_<i>_<j>" (the reference keeps the method, main.py:921)
        _, i0, i1 = args.mode.split("_")
        runner.interpolate_view(int(i0), int(i1))
    else:
        raise ValueError("unknown mode %s" % args.mode)


if __name__ == "__main__":
    main()
