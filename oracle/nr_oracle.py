"""CPU restatement of the part of `neural_renderer` (third-party, un-vendored: `neural_renderer_pytorch`, requirements of
AvatarCLIP; version unpinned) that the SMPL prior goes through -- TEST INFRASTRUCTURE ONLY.

Call sites in the reference: AvatarGen/AppearanceGen/models/utils.py:108-125 (`render_one_batch`: nr.Renderer(camera_mode=
'look'), white 8^3 textures, vertices @ rot_mat, output x-flipped) and AvatarGen/ShapeGen/render.py:33-56 (the dataset
renders: camera_mode='look_at').  Published algorithm restated here (neural_renderer/{renderer,look_at,look,perspective,
lighting,rasterize}.py and cuda/rasterize_cuda_kernel.cu of the PyTorch port):

  Renderer defaults: image_size 256, anti_aliasing True (rasterise at 2x, then 2x2 average pool), fill_back True (every face
  also in reversed order), perspective with viewing_angle 30 deg (x' = x / z / tan 30; vertices with z <= 0 -> (0, 0, 0): the patch of
  the reference's README.md:126-134), near 0.1, far 100, background 0,
  light = ambient 0.5 + directional 0.5 * relu(n . (0,1,0)) with the FACE normal n = normalize((v0 - v1) x (v2 - v1)) taken in
  world space before the camera transform; flat (per-face) shading of a constant white texture.
  look_at / look: z = normalize(at - eye) (or the given direction), x = normalize(up x z), y = normalize(z x x); v_cam = (v - eye) R^T.
  Rasteriser (per pixel of the is x is grid, is = 2 * 256): pixel centre xp = (2 xi + 1 - is) / is, same for y; a face is skipped when
  (y2 - y0)(x1 - x0) < (y1 - y0)(x2 - x0) (back side); inside test with the three edge functions (>= keeps the pixel); barycentric
  weights from the inverse of the pixel-space vertex matrix, clamped to [0,1] and renormalised; 1/z interpolated; nearest z in
  (near, far) wins; the image is finally flipped vertically (row 0 = top).

PARITY UNPINNED against neural_renderer itself: the package is absent offline, and the only neural_renderer outputs the
reference ships (2 x 108 renders under data/zero_beta_*_render) show a POSED body whose vertices need the licensed SMPL
files (the shipped zero_beta_smpl.obj is the un-posed template: arms horizontal, while both render sets have the arms
down), so they cannot be re-rendered here.  What tests/test_smpl_prior.py does check against those renders: orientation,
handedness, field of view and framing (the pose-independent torso / head / legs of the frontal views overlap), and the
grey-level range produced by the ambient + directional light and the 2x2 anti-aliasing (31 = one of four sub-samples at
ambient 0.5, up to 251).
"""
import numpy as np

ROT_MAT = np.array([[1., 0., 0.], [0., 0., -1.], [0., 1., 0.]])   # models/utils.py:114-118 (vertices @ rot_mat)


def _nrm(a):
    return a / np.linalg.norm(a)


def camera_rotation(direction, up=(0., 1., 0.)):
    z = _nrm(np.asarray(direction, np.float64))
    x = _nrm(np.cross(np.asarray(up, np.float64), z))
    y = _nrm(np.cross(z, x))
    return np.stack([x, y, z])             # rows


def look(vertices, eye, direction, up=(0., 1., 0.)):
    """neural_renderer/look.py (and look_at.py with direction = at - eye)."""
    r = camera_rotation(direction, up)
    return (np.asarray(vertices, np.float64) - np.asarray(eye, np.float64)) @ r.T


def perspective(v, angle_deg=30.0):
    """neural_renderer/perspective.py WITH the three lines the reference's README prescribes (README.md:126-134: `x[z<=0] = 0`,
    `y[z<=0] = 0`, `z[z<=0] = 0` after the division -- "objects behind the camera will also be rendered" otherwise): a vertex behind
    the camera becomes (0, 0, 0); a face that owns one then interpolates 1 / z = inf and fails the near test"""
    w = np.tan(np.deg2rad(angle_deg))
    out = v.copy()
    with np.errstate(divide="ignore", invalid="ignore"):
        out[:, 0] = v[:, 0] / v[:, 2] / w
        out[:, 1] = v[:, 1] / v[:, 2] / w
    out[v[:, 2] <= 0] = 0.0
    return out


def face_light(v_world, faces, ambient=0.5, directional=0.5, direction=(0., 1., 0.)):
    """neural_renderer/lighting.py on the fill_back face list: returns the intensity of the original and of the reversed copy."""
    f = v_world[faces]
    n = np.cross(f[:, 0] - f[:, 1], f[:, 2] - f[:, 1])
    n = n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-5)
    c = n @ np.asarray(direction, np.float64)
    return ambient + directional * np.maximum(c, 0), ambient + directional * np.maximum(-c, 0)


def rasterize(v_ndc, faces, light_fwd, light_rev, image_size=256, anti_aliasing=True, near=0.1, far=100.0, dtype=np.float32):
    """-> grey image [image_size, image_size] (row 0 = top).  dtype float32 follows the CUDA kernel's arithmetic (and the HIP
    kernel's) operation by operation; float64 is available for sensitivity checks."""
    is_ = image_size * 2 if anti_aliasing else image_size
    img = np.zeros((is_, is_), dtype)
    zbuf = np.full((is_, is_), np.inf, dtype)
    is_f = dtype(is_)
    fv = np.asarray(v_ndc, dtype)[faces]                      # [F,3,3]
    light_fwd, light_rev = np.asarray(light_fwd, dtype), np.asarray(light_rev, dtype)
    near, far = dtype(near), dtype(far)
    for rev in (False, True):
        tri = fv[:, ::-1] if rev else fv
        light = light_rev if rev else light_fwd
        x0, y0, z0 = tri[:, 0, 0], tri[:, 0, 1], tri[:, 0, 2]
        x1, y1, z1 = tri[:, 1, 0], tri[:, 1, 1], tri[:, 1, 2]
        x2, y2, z2 = tri[:, 2, 0], tri[:, 2, 1], tri[:, 2, 2]
        front = ~((y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0))
        for k in np.nonzero(front)[0]:
            xs, ys = tri[k, :, 0], tri[k, :, 1]
            # pixel range of the bounding box: xp = (2 xi + 1 - is) / is  <=>  xi = (xp * is + is - 1) / 2
            lo_x = int(np.floor((xs.min() * is_ + is_ - 1) / 2)); hi_x = int(np.ceil((xs.max() * is_ + is_ - 1) / 2))
            lo_y = int(np.floor((ys.min() * is_ + is_ - 1) / 2)); hi_y = int(np.ceil((ys.max() * is_ + is_ - 1) / 2))
            lo_x, lo_y, hi_x, hi_y = max(lo_x, 0), max(lo_y, 0), min(hi_x, is_ - 1), min(hi_y, is_ - 1)
            if lo_x > hi_x or lo_y > hi_y:
                continue
            xi = np.arange(lo_x, hi_x + 1, dtype=dtype)[None, :]
            yi = np.arange(lo_y, hi_y + 1, dtype=dtype)[:, None]
            xp = (dtype(2) * xi + dtype(1) - is_f) / is_f
            yp = (dtype(2) * yi + dtype(1) - is_f) / is_f
            out = ((yp - y0[k]) * (x1[k] - x0[k]) < (xp - x0[k]) * (y1[k] - y0[k])) | \
                  ((yp - y1[k]) * (x2[k] - x1[k]) < (xp - x1[k]) * (y2[k] - y1[k])) | \
                  ((yp - y2[k]) * (x0[k] - x2[k]) < (xp - x2[k]) * (y0[k] - y2[k]))
            if out.all():
                continue
            p = dtype(0.5) * (tri[k, :, :2] * is_f + is_f - dtype(1))        # pixel-space vertices
            inv = np.array([[p[1, 1] - p[2, 1], p[2, 0] - p[1, 0], p[1, 0] * p[2, 1] - p[2, 0] * p[1, 1]],
                            [p[2, 1] - p[0, 1], p[0, 0] - p[2, 0], p[2, 0] * p[0, 1] - p[0, 0] * p[2, 1]],
                            [p[0, 1] - p[1, 1], p[1, 0] - p[0, 0], p[0, 0] * p[1, 1] - p[1, 0] * p[0, 1]]])
            den = p[2, 0] * (p[0, 1] - p[1, 1]) + p[0, 0] * (p[1, 1] - p[2, 1]) + p[1, 0] * (p[2, 1] - p[0, 1])
            if den == 0:
                continue
            w = [np.clip((inv[j, 0] * xi + inv[j, 1] * yi + inv[j, 2]) / den, 0, 1) for j in range(3)]
            ws = np.maximum(w[0] + w[1] + w[2], dtype(1e-10))
            zp = dtype(1) / ((w[0] / z0[k] + w[1] / z1[k] + w[2] / z2[k]) / ws)
            ok = (~out) & (zp > near) & (zp < far)
            sub_z = zbuf[lo_y:hi_y + 1, lo_x:hi_x + 1]
            sub_i = img[lo_y:hi_y + 1, lo_x:hi_x + 1]
            win = ok & (zp < sub_z)
            sub_z[win] = zp[win]
            sub_i[win] = light[k]
    img = img[::-1]                                           # rasterize.py: vertical flip (y up -> row 0 = top)
    if anti_aliasing:
        img = img.reshape(image_size, 2, image_size, 2).mean(axis=(1, 3))
    return img


def render(v_world, faces, eye, direction, image_size=256, dtype=np.float32):
    """nr.Renderer(camera_mode='look' | 'look_at')(vertices, faces, white textures) -> grey [H,W] in [0,1]."""
    lf, lr = face_light(np.asarray(v_world, np.float64), faces)
    v = perspective(look(v_world, eye, direction))
    return rasterize(v, faces, lf, lr, image_size, dtype=dtype)


def render_one_batch(v, faces, eye, at):
    """models/utils.py:108-125: vertices @ rot_mat, camera at `eye` looking along (at - eye), image x-flipped -> [256,256,3]."""
    eye, at = np.asarray(eye, np.float64), np.asarray(at, np.float64)
    img = render(np.asarray(v, np.float64) @ ROT_MAT, faces, eye, (at - eye) / np.linalg.norm(at - eye))
    return np.repeat(img[:, ::-1, None], 3, axis=2)
