"""Pack the reference's own SMPL renders and mesh into a test fixture.  Run in the build container only:

    python oracle/gen_golden_views.py   ->   tests/golden/smpl_views.npz   (TEST INFRASTRUCTURE ONLY)

Contents (all produced by the reference authors with neural_renderer + smplx, shipped under AvatarGen/AppearanceGen/data/):
  stand_* / tpose_* : the 108 views of data/zero_beta_{standpose,tpose}_render (ShapeGen/render.py:33-56: 18 azimuths x 6
                      elevations, camera distance 2.2, 60 degree FOV, 256 x 256); the PNGs are grey (R = G = B: white
                      texture under neural_renderer's ambient 0.5 + directional 0.5 light), so one channel is stored;
                      poses = transforms_train.json `transform_matrix` (camera-to-world).
  mesh_v, mesh_f    : data/zero_beta_smpl.obj (6890 vertices, 13776 faces): the zero-beta SMPL template, un-posed.
Used by: tests/test_gpu_dataset_train.py (Runner.train on the reference's own dataset, BASELINE config 1),
         tests/test_smpl_prior.py (silhouette / shading of the HIP rasteriser against the neural_renderer renders)."""
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def load_set(d):
    meta = json.load(open(os.path.join(d, "transforms_train.json")))
    ims, poses = [], []
    for fr in meta["frames"]:
        im = np.asarray(Image.open(os.path.join(d, fr["file_path"] + ".png")).convert("RGB"))
        assert (im[..., 0] == im[..., 1]).all() and (im[..., 0] == im[..., 2]).all()
        ims.append(im[..., 0])
        poses.append(np.asarray(fr["transform_matrix"], np.float64))
    return np.stack(ims), np.stack(poses), float(meta["camera_angle_x"])


def read_obj(path):
    v, f = [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            v.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            f.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def main():
    data = os.path.join(ref_loader.REF_AG, "data")
    out = {}
    for tag, sub in (("stand", "zero_beta_standpose_render"), ("tpose", "zero_beta_tpose_render")):
        ims, poses, ang = load_set(os.path.join(data, sub))
        out[tag + "_images"], out[tag + "_poses"], out[tag + "_camera_angle_x"] = ims, poses, np.asarray(ang)
        print(tag, ims.shape, "coverage %.4f" % (ims > 0).mean())
    out["mesh_v"], out["mesh_f"] = read_obj(os.path.join(data, "zero_beta_smpl.obj"))
    print("mesh", out["mesh_v"].shape, out["mesh_f"].shape)
    path = os.path.join(GOLD, "smpl_views.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
