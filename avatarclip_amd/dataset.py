"""SMPL_Dataset: camera / ray generation with the reference's conventions
(AvatarGen/AppearanceGen/models/dataset.py:203-347), device-explicit and generalised beyond H=W=256
(BASELINE configs render at 224^2 and 512^2).  Ray maths runs as torch elementwise ops on the target device.
"""
import json
import os

import numpy as np
import torch


class SMPL_Dataset:
    def __init__(self, conf=None, device="cuda", H=None, W=None, camera_angle_x=None, load_images=True):
        """conf: ConfigTree with `data_dir` (NeRF-synthetic layout: transforms_train.json + img/*.png, dataset.py:211-224).
        Without a data_dir the dataset is purely synthetic: H, W and camera_angle_x (default 60 deg) define the camera."""
        self.device = torch.device(device)
        self.conf = conf
        self.images = None
        self.masks = None
        self._images_dev = None
        self._pin, self._pin_i = [(None, None)] * 4, 0
        self.poses = None
        self.images_lis = []
        data_dir = None if conf is None else conf.get_string("data_dir", default=None)
        meta = None
        if data_dir is not None and os.path.exists(os.path.join(data_dir, "transforms_train.json")):
            with open(os.path.join(data_dir, "transforms_train.json")) as fp:
                meta = json.load(fp)
        if meta is not None:
            poses, imgs = [], []
            for frame in meta["frames"]:
                fname = os.path.join(data_dir, frame["file_path"] + ".png")
                self.images_lis.append(fname)
                poses.append(np.array(frame["transform_matrix"]))
                if load_images:
                    from PIL import Image
                    imgs.append(np.asarray(Image.open(fname).convert("RGB")))
            self.poses = torch.from_numpy(np.array(poses).astype(np.float32)).to(self.device)
            if load_images:
                images = (np.array(imgs) / 255.0).astype(np.float32)
                images = images[:, :, ::-1]                       # dataset.py:226 (mirrors the W axis)
                self.images = torch.from_numpy(images.copy()).cpu()
                self.masks = torch.zeros_like(self.images)
                self.masks[self.images != 0] = 1.0                # dataset.py:228-229
                H0, W0 = self.images[0].shape[:2]
            else:
                H0 = W0 = 256
            camera_angle_x = float(meta["camera_angle_x"])
        else:
            H0 = W0 = 256
        self.n_images = 0 if self.poses is None else len(self.poses)
        self.H = int(H) if H is not None else H0
        self.W = int(W) if W is not None else W0
        if camera_angle_x is None:
            camera_angle_x = np.pi / 3
        self.focal = 0.5 * self.W / np.tan(0.5 * camera_angle_x)   # dataset.py:234-235
        self.image_pixels = self.H * self.W
        self.object_bbox_min = np.array([-1.01, -1.01, -1.01])
        self.object_bbox_max = np.array([1.01, 1.01, 1.01])
        self.K = torch.tensor([[self.focal, 0, 0.5 * self.W], [0, self.focal, 0.5 * self.H], [0, 0, 1]], dtype=torch.float64)

    # ------------------------------------------------------------------ rays
    def _dirs(self, px, py, pose):
        p = torch.stack([(px - 0.5 * self.W) / self.focal, -(py - 0.5 * self.H) / self.focal, -torch.ones_like(px)], -1).float()
        v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
        pose = pose.to(v.device).float()
        v = torch.sum(v[..., None, :] * pose[:3, :3], -1)
        o = pose[:3, 3].expand(v.shape)
        return o, v

    def gen_rays_pose(self, pose, resolution_level=1):
        """dataset.py:277-293."""
        l = resolution_level
        dev = self.device
        tx = torch.linspace(0, self.W - 1, int(self.W // l), device=dev)
        ty = torch.linspace(0, self.H - 1, int(self.H // l), device=dev)
        px, py = torch.meshgrid(tx, ty, indexing="ij")
        return self._dirs(px.t(), py.t(), torch.as_tensor(pose))

    def gen_rays_at(self, img_idx, resolution_level=1):
        """dataset.py:295-312."""
        return self.gen_rays_pose(self.poses[img_idx], resolution_level)

    def gen_rays_between(self, idx_0, idx_1, ratio, resolution_level=1):
        """dataset.py:132-162 (defined on the reference's base Dataset; its SMPL_Dataset lacks the intrinsics it reads, so
        `render_novel_image` only runs there for the base class): rotation slerp + linear translation between the two
        world-to-camera matrices, rays through this dataset's pinhole camera."""
        from scipy.spatial.transform import Rotation as Rot, Slerp
        w0 = np.linalg.inv(self.poses[idx_0].detach().cpu().numpy().astype(np.float64))
        w1 = np.linalg.inv(self.poses[idx_1].detach().cpu().numpy().astype(np.float64))
        rot = Slerp([0, 1], Rot.from_matrix(np.stack([w0[:3, :3], w1[:3, :3]])))(ratio)
        pose = np.diag([1.0, 1.0, 1.0, 1.0])
        pose[:3, :3] = rot.as_matrix()
        pose[:3, 3] = ((1.0 - ratio) * w0 + ratio * w1)[:3, 3]
        pose = np.linalg.inv(pose).astype(np.float32)
        return self.gen_rays_pose(torch.from_numpy(pose), resolution_level)

    def gen_random_rays_at(self, img_idx, batch_size):
        """dataset.py:314-329 -> [B,10] = o, d, rgb, mask.  The pixel draws are the reference's (torch's CPU generator, x then y); the
        look-ups run on the device from a device copy of the images: the reference's CPU fancy-indexing + pageable uploads cost
        34 ms per 5 120-ray batch on a 256-core host (a parallel CPU gather per call, then blocking copies ordered behind the whole
        previous iteration) against 3 ms of GPU work -- profiles/r03_silhouette_mode.txt."""
        px = torch.randint(low=0, high=self.W, size=[batch_size])
        py = torch.randint(low=0, high=self.H, size=[batch_size])
        if self.device.type != "cuda":
            color = self.images[img_idx][(py, px)]
            mask = self.masks[img_idx][(py, px)]
            o, v = self._dirs(px.float(), py.float(), self.poses[img_idx])
            return torch.cat([o, v, color, mask[:, :1]], dim=-1)
        if self._images_dev is None:
            self._images_dev = self.images.to(self.device)
        pxy = self._upload_pixels(px, py)                                       # [2,B] int64 on the device, without blocking the host
        flat = pxy[1] * self.W + pxy[0]
        color = self._images_dev[int(img_idx)].reshape(-1, 3).index_select(0, flat)
        mask = (color[:, :1] != 0).to(color.dtype)                              # dataset.py:228-229: masks[images != 0] = 1, channel 0
        o, v = self._dirs(pxy[0].float(), pxy[1].float(), self.poses[int(img_idx)])
        return torch.cat([o, v, color, mask], dim=-1)

    def _upload_pixels(self, px, py):
        """pixel indices -> device through a small ring of pinned staging buffers (a slot is reused only after the event recorded
        behind its copy has completed): a `.to(device)` from pageable memory would block the host behind the previous iteration"""
        n = px.numel()
        slot = self._pin_i % 4
        self._pin_i += 1
        buf, ev = self._pin[slot]
        if buf is None or buf.shape[1] < n:
            buf = torch.empty(2, max(n, 1024), dtype=torch.int64).pin_memory()
        elif ev is not None:
            ev.synchronize()
        buf[0, :n].copy_(px)
        buf[1, :n].copy_(py)
        with torch.cuda.device(self.device):
            out = buf[:, :n].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        self._pin[slot] = (buf, ev)
        return out

    def gen_rays_silhouettes(self, pose, max_ray_num, mask):
        """dataset.py:252-275: rays only inside the 10x-dilated silhouette `mask` [256,256].
        The reference dilates on the host (scipy.ndimage.binary_dilation, full 3x3 structure, 10 iterations = every pixel
        within Chebyshev distance 10 of the mask = a 21x21 maximum filter with zero border); here the filter runs on the
        device and the only host round trip left is the scalar that fixes the ray-grid size (a dynamic shape)."""
        grid = self.silhouette_grid(max_ray_num, mask)
        if grid is None:
            return self.gen_rays_pose(pose, resolution_level=4)
        Wn, sel, self.last_sel_idx = grid
        dev = self.device
        tx = torch.linspace(0, self.W - 1, Wn, device=dev)
        ty = torch.linspace(0, self.H - 1, Wn, device=dev)
        px, py = torch.meshgrid(tx, ty, indexing="ij")
        o, v = self._dirs(px.t(), py.t(), torch.as_tensor(pose))
        return o.reshape(-1, 3).index_select(0, self.last_sel_idx), v.reshape(-1, 3).index_select(0, self.last_sel_idx), Wn, sel

    def silhouette_grid(self, max_ray_num, mask):
        """the data-dependent half of gen_rays_silhouettes (dataset.py:252-275): -> (Wn = Hn of the ray grid, the selected pixels of
        that grid as a bool image, their row-major positions), or None for an empty silhouette"""
        m0 = torch.as_tensor(mask, device=self.device)
        m0 = (m0 != 0).float()
        dilated = torch.nn.functional.max_pool2d(m0[None, None], kernel_size=21, stride=1, padding=10)[0, 0]
        n_mask, n_dilated = torch.stack([m0.sum(), dilated.sum()]).tolist()      # ONE round trip for both counts
        if n_mask == 0:
            return None
        ratio = float(n_dilated) / float(m0.shape[0] * m0.shape[1])
        Wn = Hn = min(self.H, int(np.sqrt(max_ray_num / ratio)))
        m = torch.nn.functional.interpolate(dilated.reshape(1, 1, *dilated.shape), size=(Hn, Wn)).squeeze()
        sel = m > 0
        # the row-major positions of the selected pixels: the second (and last) round trip -- it fixes the ray count.  Everything
        # downstream gathers / scatters with these indices (same order as boolean-mask indexing) without another synchronisation.
        return Wn, sel, sel.reshape(-1).nonzero().squeeze(1)

    def rays_fused(self, pose, Wn, Hn, sel_idx=None, prior=None):
        """gen_rays_pose on the Wn x Hn pixel grid (or its listed pixels) + near_far_from_sphere + the prior render resampled to the grid
        (main.py:376-380), in ONE launch (csrc/avc_glue.hip: avc_gen_rays) -> rays_o, rays_d [R,3], near, far [R,1], true_rgb [Hn*Wn,3],
        mask [Hn*Wn,1].  GPU only; the methods above are the torch statement the parity tests compare it with."""
        from . import lib as L
        dev = self.device
        R = int(sel_idx.numel()) if sel_idx is not None else Wn * Hn
        f32 = dict(device=dev, dtype=torch.float32)
        rays_o, rays_d = torch.empty(R, 3, **f32), torch.empty(R, 3, **f32)
        near, far = torch.empty(R, 1, **f32), torch.empty(R, 1, **f32)
        true_rgb = mask = None
        Hp = Wp = 0
        if prior is not None:
            prior = prior.contiguous().float()
            Hp, Wp = prior.shape[0], prior.shape[1]
            true_rgb, mask = torch.empty(Hn * Wn, 3, **f32), torch.empty(Hn * Wn, 1, **f32)
        pose = torch.as_tensor(pose).to(dev).float().contiguous()
        L.check(L.load().avc_gen_rays(L.ptr(pose), L.ptr(sel_idx), L.ptr(prior), Hp, Wp, float(self.W), float(self.H), float(self.focal), Wn, Hn, R,
                                      L.ptr(rays_o), L.ptr(rays_d), L.ptr(near), L.ptr(far), L.ptr(true_rgb), L.ptr(mask), L.stream()), "avc_gen_rays")
        return rays_o, rays_d, near, far, true_rgb, mask

    def near_far_from_sphere(self, rays_o, rays_d, is_sphere=False):
        """dataset.py:331-342 (`is_sphere` is ignored by the reference too)."""
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        near = (mid - 1).clamp(min=0)
        far = mid + 1
        return near, far

    def image_at(self, idx, resolution_level):
        from PIL import Image
        img = np.asarray(Image.open(self.images_lis[idx]).convert("RGB"))[:, ::-1, ::-1]   # BGR + mirror like cv2 path
        size = (self.W // resolution_level, self.H // resolution_level)
        return np.asarray(Image.fromarray(np.ascontiguousarray(img)).resize(size, Image.BILINEAR)).clip(0, 255)
