"""The SMPL silhouette / colour prior of AppearanceGen (main.py:290-335 `init_smpl`, :360 `render_one_batch`; SURVEY.md
section 8 row f-1) on the device: a posed body mesh rendered from the iteration's camera by the HIP rasteriser
(csrc/avc_raster.hip) with neural_renderer's conventions -- white texture, ambient 0.5 + directional 0.5 face lighting from
(0,1,0), 60 degree field of view, 2x super-sampling, vertices @ rot_mat, output mirrored in x (models/utils.py:108-125).
Nothing leaves the GPU: the reference renders with neural_renderer, copies the image to the host and back every iteration.

The mesh is an input: a posed .obj (`MeshPrior.from_obj`), vertices/faces tensors, or the SMPL arrays + pose
(`MeshPrior.from_smpl`, linear blend skinning in smpl_lbs.py).  No CPU fallback: the rasteriser is a HIP kernel."""
import numpy as np
import torch

from . import lib as L
from . import h2d

import os
FUSED_PRIOR = os.environ.get("AVC_FUSED_PRIOR", "1") != "0"    # projection + rasteriser + pooling as four launches (0: torch ops around avc_rasterize_faces)

ROT_MAT = ((1.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0))    # models/utils.py:114-118


def read_obj(path):
    v, f = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                v.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                f.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


class MeshPrior:
    """prior_renderer(eye, at) -> [image_size, image_size, 3] float32 on `device` (0 = background)"""

    def __init__(self, vertices, faces, device="cuda", image_size=256, viewing_angle=30.0, near=0.1, far=100.0,
                 apply_rot_mat=True, light_ambient=0.5, light_directional=0.5, light_direction=(0.0, 1.0, 0.0)):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("MeshPrior rasterises on the MI355X (no CPU fallback)")
        self.device, self.image_size, self.near, self.far = dev, int(image_size), float(near), float(far)
        self.width = float(np.tan(np.deg2rad(viewing_angle)))
        v = torch.as_tensor(np.asarray(vertices, np.float32)).to(dev).reshape(-1, 3)
        if apply_rot_mat:
            v = v @ torch.tensor(ROT_MAT, dtype=torch.float32, device=dev)
        f = torch.as_tensor(np.asarray(faces).astype(np.int64)).to(dev).reshape(-1, 3)
        self.v_world = v.contiguous()
        self.faces2 = torch.cat([f, f.flip(1)], 0)                 # fill_back=True: every face also in reversed order
        fv = v[f]                                                   # neural_renderer/lighting.py in world space (view independent)
        n = torch.cross(fv[:, 0] - fv[:, 1], fv[:, 2] - fv[:, 1], dim=1)
        n = n / n.norm(dim=1, keepdim=True).clamp(min=1e-5)
        c = n @ torch.tensor(light_direction, dtype=torch.float32, device=dev)
        self.light2 = torch.cat([light_ambient + light_directional * c.clamp(min=0),
                                 light_ambient + light_directional * (-c).clamp(min=0)]).contiguous()
        self.lib = L.load()
        self._zbufs = {}        # z-buffer scratch per HIP stream (the side-stream view preparation and main-stream validation renders never share one)
        self._ndc = None

    @classmethod
    def from_obj(cls, path, **kw):
        v, f = read_obj(path)
        return cls(v, f, **kw)

    @classmethod
    def from_smpl(cls, smpl, pose_axis_angle, v_shaped=None, **kw):
        """main.py:296-333: pose [1,24,3] axis-angle (stand_pose.npy, or the T pose with the root turned by pi/2 about x);
        v_shaped = the ShapeGen template (dataset.template_obj) or the SMPL template."""
        from . import smpl_lbs
        dev = smpl["v_template"].device
        pose = torch.as_tensor(np.asarray(pose_axis_angle, np.float32)).to(dev).reshape(-1, 3)
        rot = smpl_lbs.batch_rodrigues(pose).reshape(1, -1, 3, 3)
        vs = smpl["v_template"].reshape(1, -1, 3) if v_shaped is None else torch.as_tensor(np.asarray(v_shaped, np.float32)).to(dev).reshape(1, -1, 3)
        verts, _ = smpl_lbs.lbs(vs, rot, smpl["posedirs"], smpl["J_regressor"], smpl["parents"], smpl["lbs_weights"])
        return cls(verts[0].detach().cpu().numpy(), smpl["faces"], **kw)

    def _zbuf_for(self, need):
        """The rasteriser's persistent scratch (64-bit (depth, face) keys + the large-face list): all bits set = empty, and every
        successful call LEAVES it that way (the resolve launch restores the keys it read), which is what saves a 2-MB fill per view.
        One buffer per HIP stream -- the launches of one render are ordered by their stream, two streams must not interleave on one
        z-buffer -- dropped (and refilled on the next call) whenever a launch of the sequence reports an error, so that stale keys of
        an interrupted render cannot leak into later priors and silhouette masks."""
        key = (L.stream(), need)
        z = self._zbufs.get(key)
        if z is None:
            for k in [k for k in self._zbufs if k[0] == key[0]]:
                del self._zbufs[k]
            z = self._zbufs[key] = torch.full((need,), 255, dtype=torch.uint8, device=self.device)
        return z

    def _checked(self, status, what):
        try:
            L.check(status, what)
        except Exception:
            self._zbufs.clear()          # the z-buffer of an interrupted render is not empty any more: never reuse it
            raise

    @torch.no_grad()
    def render_grey(self, eye, direction, rgb_flipped=False):
        """nr.Renderer(camera_mode='look')(vertices, faces, ones) -> [S,S] grey image (before the x flip)"""
        dev = self.device
        # neural_renderer/look.py: the camera frame, in float32 like there -- on the host (a dozen small launches otherwise), one upload
        f = np.float32
        z = np.asarray(direction, f)
        z = z / f(np.sqrt((z * z).sum(dtype=f)))
        x = np.cross(np.array([0.0, 1.0, 0.0], f), z).astype(f)
        x = x / f(np.sqrt((x * x).sum(dtype=f)))
        y = np.cross(z, x).astype(f)
        y = y / f(np.sqrt((y * y).sum(dtype=f)))
        cam = h2d.upload(np.concatenate([np.asarray(eye, f), x, y, z]), dev)
        S = self.image_size
        if FUSED_PRIOR:      # projection + rasteriser + 2 x 2 average (+ x flip + channels) in four launches (csrc/avc_raster.hip)
            ch = 3 if rgb_flipped else 1
            out = torch.empty((S, S, 3) if rgb_flipped else (S, S), device=dev, dtype=torch.float32)
            zbuf = self._zbuf_for(self.lib.avc_rasterize_scratch_bytes(self.faces2.shape[0], 2 * S))
            if self._ndc is None:
                self._ndc = torch.empty_like(self.v_world)
                self._faces2_i32 = self.faces2.to(torch.int32).contiguous()
            self._checked(self.lib.avc_rasterize_mesh(L.ptr(self.v_world), self.v_world.shape[0], L.ptr(self._faces2_i32), self.faces2.shape[0], L.ptr(cam),
                                                      self.width, L.ptr(self.light2), S, self.near, self.far, L.ptr(self._ndc), L.ptr(out),
                                                      int(rgb_flipped), ch, L.ptr(zbuf), L.stream()), "avc_rasterize_mesh")
            return out
        v = (self.v_world - cam[:3]) @ cam[3:].reshape(3, 3).t()
        ndc = torch.stack([v[:, 0] / v[:, 2] / self.width, v[:, 1] / v[:, 2] / self.width, v[:, 2]], dim=1)   # perspective.py
        ndc = torch.where(v[:, 2:3] <= 0, torch.zeros_like(ndc), ndc)     # ... with the reference's patch (README.md:126-134): behind the camera -> 0
        fz = ndc[self.faces2].reshape(-1, 9).contiguous()
        S2 = 2 * self.image_size                                    # anti_aliasing=True
        img = torch.empty(S2, S2, device=dev, dtype=torch.float32)
        zbuf = self._zbuf_for(self.lib.avc_rasterize_scratch_bytes(fz.shape[0], S2))
        self._checked(self.lib.avc_rasterize_faces(L.ptr(fz), L.ptr(self.light2), fz.shape[0], S2, self.near, self.far, L.ptr(img),
                                                   L.ptr(zbuf), L.stream()), "avc_rasterize_faces")
        grey = torch.nn.functional.avg_pool2d(img[None, None], kernel_size=2, stride=2)[0, 0]
        return grey.flip(1)[..., None].repeat(1, 1, 3) if rgb_flipped else grey

    def __call__(self, eye, at):
        eye, at = np.asarray(eye, np.float64), np.asarray(at, np.float64)
        # models/utils.py:124 (`[:, ::-1]`), white texture: R = G = B
        return self.render_grey(eye, (at - eye) / np.linalg.norm(at - eye), rgb_flipped=True)
