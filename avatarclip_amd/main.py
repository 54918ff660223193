"""CLI with the reference's flags (AvatarGen/AppearanceGen/main.py:947-980):
   python -m avatarclip_amd.main --mode {train,train_clip,validate_mesh} --conf X [--is_continue] [--gpu N] [--case NAME]
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
    args = parser.parse_args()
    rank, world, local_rank = parallel.init_from_env()
    torch.cuda.set_device(local_rank if world > 1 else args.gpu)
    if args.mode == "validate_mesh":
        args.is_continue = True
    runner = Runner(args.conf, args.mode, args.case, args.is_continue)
    if args.mode == "validate_mesh":
        # main.py:972-974 (the reference then also renders turn-table videos: render_geometry_cast_light, not built)
        runner.validate_mesh(world_space=True, resolution=512, threshold=args.mcube_threshold)
    elif args.mode == "train":
        runner.train()
    elif args.mode == "train_clip":
        if args.clip_weights is not None:
            runner.conf.put("clip.weights", args.clip_weights)
        runner.init_clip()
        runner.init_smpl()
        runner.train_clip()
    else:
        raise NotImplementedError("mode %s (visualisation) is outside this round's scope" % args.mode)


if __name__ == "__main__":
    main()
