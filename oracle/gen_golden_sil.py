"""Golden vectors for SMPL_Dataset.gen_rays_silhouettes (models/dataset.py:252-275), produced by RUNNING the reference's own
function (extracted with `ast`, like oracle/gen_golden.py does for the other dataset functions: the module's import block needs
cv2 / imageio).  Run in the build container only:  python oracle/gen_golden_sil.py  ->  tests/golden/rays_sil.npz"""
import json
import os
import sys

import numpy as np
import torch
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gen_golden import extract_functions, GOLD  # noqa: E402


def synthetic_mask(seed):
    """a 256x256 'body' silhouette: a few overlapping ellipses"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:256, 0:256]
    m = np.zeros((256, 256), np.float32)
    for _ in range(4):
        cx, cy = rng.uniform(90, 166, 2)
        ax, ay = rng.uniform(8, 30), rng.uniform(20, 70)
        m[((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2 < 1] = 1.0
    return m


def main():
    R = ref_loader.load_reference()
    AG = R.ref_ag
    fns = extract_functions(os.path.join(AG, "models", "dataset.py"), ["gen_rays_silhouettes", "gen_rays_pose"], "SMPL_Dataset")
    for f in fns.values():
        f.__globals__["ndimage"] = ndimage
    meta = json.load(open(os.path.join(AG, "data", "zero_beta_standpose_render", "transforms_train.json")))
    H = W = 256
    focal = 0.5 * W / np.tan(0.5 * float(meta["camera_angle_x"]))
    K = torch.from_numpy(np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]]))
    DS = type("DS", (), {"gen_rays_pose": fns["gen_rays_pose"]})
    ds = DS()
    ds.W, ds.H, ds.K = W, H, K
    pose = torch.tensor(meta["frames"][17]["transform_matrix"]).float()
    rec = {"pose": pose.numpy(), "focal": np.float64(focal)}
    for i, max_ray_num in enumerate((3000, 112 * 112)):
        mask = synthetic_mask(i)
        o, v, Wn, sel = fns["gen_rays_silhouettes"](ds, pose, max_ray_num, mask)
        rec.update({"mask%d" % i: mask, "max_ray_num%d" % i: np.int64(max_ray_num), "rays_o%d" % i: o.contiguous().numpy(),
                    "rays_v%d" % i: v.contiguous().numpy(), "W%d" % i: np.int64(Wn), "sel%d" % i: sel.numpy()})
    np.savez_compressed(os.path.join(GOLD, "rays_sil.npz"), **rec)
    print({k: np.asarray(v).shape for k, v in rec.items()})


if __name__ == "__main__":
    main()
