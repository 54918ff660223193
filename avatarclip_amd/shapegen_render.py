"""ShapeGen's NeuS-init dataset writer (AvatarGen/ShapeGen/render.py:31-56 `render_for_nerf`, :108-139 `render_coarse_shape`;
SURVEY.md section 8 row f-4) on the HIP rasteriser: the coarse body mesh rendered from 18 azimuths x 6 elevations (camera
distance 2.2, look_at the origin, 60 degree field of view, 256 x 256, white texture under neural_renderer's ambient 0.5 +
directional 0.5 light), written in the layout AppearanceGen's `Runner.train` reads (img/NNNN.png + transforms_train.json with
the camera-to-world matrices of render.py:17-29).  This is the link between ShapeGen and AppearanceGen in BASELINE config 5.

    python -m avatarclip_amd.shapegen_render --coarse_shape_obj X.obj --output_folder DIR [--smpl_model SMPL.npz|pkl --pose_type stand_pose|t_pose]

Without SMPL model files the .obj is rendered as it is (an already posed mesh); with them it is posed first
(render.py:108-121: `my_lbs` with the stand pose `--pose_npy` or the T pose whose root is turned by pi/2 about x)."""
import argparse
import json
import math
import os

import numpy as np
import torch

CAMERA_DISTANCE = 2.2                      # render.py:32
AZIMUTHS = tuple(range(0, 360, 20))        # render.py:48
ELEVATIONS = tuple(range(-60, 60, 20))     # render.py:49
CAMERA_ANGLE_X = 60.0 / 180.0 * np.pi      # render.py:131


def get_points_from_angles(distance, elevation, azimuth):
    """neural_renderer.get_points_from_angles (degrees): the eye of camera_mode 'look_at'"""
    e, a = math.radians(elevation), math.radians(azimuth)
    return np.array([distance * math.cos(e) * math.sin(a), distance * math.sin(e), -distance * math.cos(e) * math.cos(a)])


def lookat(eye, target, up):
    """render.py:17-29: the camera-to-world matrix written to transforms_train.json"""
    n = lambda v: v / np.linalg.norm(v)
    zaxis = n(eye - target)
    xaxis = n(np.cross(up, zaxis))
    yaxis = np.cross(zaxis, xaxis)
    view = np.array([[xaxis[0], xaxis[1], xaxis[2], -np.dot(xaxis, eye)],
                     [yaxis[0], yaxis[1], yaxis[2], -np.dot(yaxis, eye)],
                     [zaxis[0], zaxis[1], zaxis[2], -np.dot(zaxis, eye)],
                     [0.0, 0.0, 0.0, 1.0]])
    return np.linalg.inv(view)


def nerf_cameras(camera_distance=CAMERA_DISTANCE):
    """the 108 (eye, camera-to-world) pairs in the reference's loop order (azimuth outer, elevation inner: render.py:48-49).
    (The shipped data/zero_beta_standpose_render has exactly these matrices; data/zero_beta_tpose_render was rendered with an
    earlier camera_distance of 2.0 -- same rotations.)"""
    cams = []
    for angle in AZIMUTHS:
        for elevation in ELEVATIONS:
            eye = get_points_from_angles(camera_distance, elevation, angle)
            cams.append((eye, lookat(eye, np.zeros(3), np.array([0.0, 1.0, 0.0]))))
    return cams


@torch.no_grad()
def render_for_nerf(vertices, faces, device="cuda", image_size=256, camera_distance=CAMERA_DISTANCE):
    """render.py:31-56 -> (images [108, S, S] float32 in [0,1] on `device` (R = G = B: one channel), transformation list)"""
    from .smpl_prior import MeshPrior
    prior = MeshPrior(vertices, faces, device=device, image_size=image_size)     # vertices @ rot_mat, fill_back, 2x anti-aliasing
    images, transforms = [], []
    for eye, t in nerf_cameras(camera_distance):
        images.append(prior.render_grey(eye, -eye / np.linalg.norm(eye)))        # camera_mode 'look_at' with at = 0: look along -eye
        transforms.append(t)
    return torch.stack(images), transforms


def write_nerf_dataset(output_dir, vertices, faces, device="cuda", image_size=256, camera_distance=CAMERA_DISTANCE):
    """render.py:122-139: img/NNNN.png (uint8 = floor(255 x), like `.type(torch.uint8)`) + transforms_train.json"""
    from PIL import Image
    images, transforms = render_for_nerf(vertices, faces, device, image_size, camera_distance)
    u8 = (images * 255).to(torch.uint8).cpu().numpy()
    os.makedirs(os.path.join(output_dir, "img"), exist_ok=True)
    frames = []
    for i, (im, t) in enumerate(zip(u8, transforms)):
        Image.fromarray(np.repeat(im[..., None], 3, axis=2)).save(os.path.join(output_dir, "img", "%s.png" % str(i).zfill(4)))
        frames.append({"file_path": "img/%s" % str(i).zfill(4), "transform_matrix": t.tolist()})
    with open(os.path.join(output_dir, "transforms_train.json"), "w") as fh:
        json.dump({"camera_angle_x": CAMERA_ANGLE_X, "frames": frames}, fh)
    return u8, transforms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--coarse_shape_obj", type=str, required=True)
    ap.add_argument("--output_folder", type=str, default="./output/render")
    ap.add_argument("--pose_type", type=str, choices=["stand_pose", "t_pose"], default="stand_pose")
    ap.add_argument("--smpl_model", type=str, default=None, help="SMPL arrays (.npz / SMPL_NEUTRAL.pkl): pose the shape like render.py:108-121")
    ap.add_argument("--pose_npy", type=str, default="./output/stand_pose.npy")
    args = ap.parse_args()
    from .smpl_prior import read_obj
    v, f = read_obj(args.coarse_shape_obj)
    if args.smpl_model is not None:
        from . import smpl_lbs
        smpl = smpl_lbs.load_smpl_arrays(args.smpl_model, "cuda")
        if args.pose_type == "stand_pose":
            pose = np.load(args.pose_npy).astype(np.float32)
        else:
            pose = np.zeros([1, 24, 3], np.float32)
            pose[:, 0, 0] = np.pi / 2
        rot = smpl_lbs.batch_rodrigues(torch.from_numpy(pose.reshape(-1, 3)).cuda()).reshape(1, -1, 3, 3)
        verts, _ = smpl_lbs.lbs(torch.from_numpy(v).cuda().reshape(1, -1, 3), rot, smpl["posedirs"], smpl["J_regressor"], smpl["parents"],
                                smpl["lbs_weights"])
        v, f = verts[0].cpu().numpy(), smpl["faces"]
    print("Begin rendering obj: {}".format(args.coarse_shape_obj))
    write_nerf_dataset(args.output_folder, v, f)
    print("Renderings written to: {}".format(args.output_folder))


if __name__ == "__main__":
    main()
