"""Runner: the reference's optimisation driver with the same constructor, conf keys, modes and checkpoint
format (AvatarGen/AppearanceGen/main.py:30-632), running its per-iteration hot path on the gfx950 kernels.

Differences that are deliberate and documented:
  * view-sharded data parallelism (not in the reference, which is single-GPU): when torch.distributed is
    initialised every rank draws its own camera and ONE flat-bucket all-reduce (RCCL over xGMI) averages the
    gradients before the identical Adam step;
  * the SMPL silhouette prior (smplx + neural_renderer, main.py:290-335,360) and the CLIP text tower are outside
    the hot path: `init_smpl(prior_renderer=...)` / `init_clip(perceptor=..., text_embeddings=...)` take them as
    inputs; procedural / seeded stand-ins are used (with a warning) when they are not supplied;
  * tensorboard logging is optional (absent offline) and scalar logging does not force a device sync every step.
"""
import logging
import os
import random
from shutil import copyfile

import numpy as np
import torch
import torch.nn.functional as F

from .conf import ConfigFactory
from .dataset import SMPL_Dataset
from .fields import RenderingNetwork, SDFNetwork, SingleVarianceNetwork
from .renderer import NeuSRenderer
from .utils import lookat, random_at, random_eye, random_eye_normal, sphere_coord
from . import parallel


class EllipsoidPrior:
    """Procedural stand-in for `render_one_batch(self.v, self.f, eye, at)` (models/utils.py:108-125): a Lambert-shaded
    ellipsoid 'body' rendered at 256x256 with the dataset's 60-degree camera.  Only used when no SMPL prior renderer is
    supplied (licensed SMPL files and neural_renderer are not available offline)."""

    def __init__(self, radii=(0.28, 0.85, 0.2), res=256, fov=np.pi / 3, device="cuda"):
        self.radii = torch.tensor(radii, dtype=torch.float32, device=device)
        self.res, self.focal, self.device = res, 0.5 * res / np.tan(0.5 * fov), device

    def __call__(self, eye, at):
        pose = torch.from_numpy(lookat(np.asarray(eye, np.float64), np.asarray(at, np.float64), np.array([0., 1, 0]))).float().to(self.device)
        t = torch.linspace(0, self.res - 1, self.res, device=self.device)
        px, py = torch.meshgrid(t, t, indexing="ij")
        px, py = px.t(), py.t()
        p = torch.stack([(px - 0.5 * self.res) / self.focal, -(py - 0.5 * self.res) / self.focal, -torch.ones_like(px)], -1)
        d = p / p.norm(dim=-1, keepdim=True)
        d = torch.sum(d[..., None, :] * pose[:3, :3], -1)
        o = pose[:3, 3]
        od, dd = o / self.radii, d / self.radii
        a, b, c = (dd * dd).sum(-1), 2 * (od * dd).sum(-1), (od * od).sum() - 1
        disc = b * b - 4 * a * c
        hit = disc > 0
        tt = (-b - torch.sqrt(disc.clamp(min=0))) / (2 * a)
        n = (o + d * tt[..., None]) / self.radii ** 2
        n = n / n.norm(dim=-1, keepdim=True).clamp(min=1e-6)
        shade = (0.35 + 0.65 * (-(n * d).sum(-1)).clamp(0, 1)) * hit
        return shade[..., None].repeat(1, 1, 3)


class Runner:
    def __init__(self, conf_path, mode="train", case="CASE_NAME", is_continue=False, is_colab=False, conf=None,
                 device=None, data_root=None):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("avatarclip_amd.Runner needs an MI355X (torch.cuda) -- the hot path has no CPU fallback")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.conf_path = conf_path
        if is_colab or conf is not None:
            self.conf = conf
        else:
            with open(self.conf_path) as f:
                self.conf = ConfigFactory.parse_string(f.read())
        self.rank, self.world = parallel.rank_world()
        self.base_exp_dir = self.conf["general.base_exp_dir"]
        os.makedirs(self.base_exp_dir, exist_ok=True)
        ds_conf = self.conf.get("dataset", default=None)
        if data_root is not None and ds_conf is not None and "data_dir" in ds_conf:
            ds_conf.put("data_dir", os.path.join(data_root, ds_conf["data_dir"]))
        self.dataset = SMPL_Dataset(ds_conf, device=self.device, load_images=(mode == "train"),
                                    H=self.conf.get_int("dataset.H", default=None) if ds_conf is not None else None,
                                    W=self.conf.get_int("dataset.W", default=None) if ds_conf is not None else None)
        self.iter_step = 0
        c = self.conf
        # Training parameters (main.py:50-62)
        self.end_iter = c.get_int("train.end_iter")
        self.save_freq = c.get_int("train.save_freq")
        self.report_freq = c.get_int("train.report_freq")
        self.val_freq = c.get_int("train.val_freq")
        self.val_mesh_freq = c.get_int("train.val_mesh_freq")
        self.batch_size = c.get_int("train.batch_size")
        self.validate_resolution_level = c.get_int("train.validate_resolution_level")
        self.learning_rate = c.get_float("train.learning_rate")
        self.learning_rate_alpha = c.get_float("train.learning_rate_alpha")
        self.use_white_bkgd = c.get_bool("train.use_white_bkgd")
        self.warm_up_end = c.get_float("train.warm_up_end", default=0.0)
        self.anneal_end = c.get_float("train.anneal_end", default=0.0)
        self.max_ray_num = c.get_int("train.max_ray_num", default=112 * 112)
        self.igr_weight = c.get_float("train.igr_weight")
        self.mask_weight = c.get_float("train.mask_weight")
        # optional keys with the reference's defaults (main.py:67-127)
        self.clip_weight = c.get_float("train.clip_weight", default=None)
        self.extra_color = c.get_bool("model.rendering_network.extra_color", default=False)
        self.add_no_texture = c.get_bool("train.add_no_texture", default=False)
        self.texture_cast_light = c.get_bool("train.texture_cast_light", default=False)
        self.use_face_prompt = c.get_bool("train.use_face_prompt", default=False)
        self.use_back_prompt = c.get_bool("train.use_back_prompt", default=False)
        self.use_silhouettes = c.get_bool("train.use_silhouettes", default=False)
        self.head_height = c.get_float("train.head_height", default=0.65)
        self.use_bg_aug = c.get_bool("train.use_bg_aug", default=True)
        self.full_frame_resolution_level = c.get_float("train.full_frame_resolution_level", default=2.25)  # main.py:371
        seed = c.get_int("train.seed", default=None)
        if seed is not None:
            self.seed = seed
            torch.manual_seed(seed); torch.cuda.manual_seed_all(seed); random.seed(seed); np.random.seed(seed)
        self.smpl_model_path = c.get_string("general.smpl_model_path", default="../../smpl_models")
        self.pose_type = c.get_string("general.pose_type", default="stand_pose")
        assert self.pose_type in ["stand_pose", "t_pose"]
        self.is_continue = is_continue
        self.mode = mode
        self.writer = None

        # Networks (main.py:133-151); every rank builds identical weights (same torch seed / same checkpoint)
        self.nerf_outside = None
        self.sdf_network = SDFNetwork(**self.conf["model.sdf_network"]).to(self.device)
        self.deviation_network = SingleVarianceNetwork(**self.conf["model.variance_network"]).to(self.device)
        self.color_network = RenderingNetwork(**self.conf["model.rendering_network"]).to(self.device)
        params_to_train = list(self.sdf_network.parameters()) + list(self.deviation_network.parameters()) + \
            list(self.color_network.parameters())
        self.params_to_train = params_to_train
        if self.world > 1:
            parallel.broadcast_params(params_to_train)
        self.optimizer = torch.optim.Adam(params_to_train, lr=self.learning_rate)
        self.renderer = NeuSRenderer(self.nerf_outside, self.sdf_network, self.deviation_network, self.color_network,
                                     **self.conf["model.neus_renderer"])
        pretrain_pth = c.get_string("train.pretrain", default=None)
        if pretrain_pth is not None:
            if os.path.exists(pretrain_pth):
                logging.info("Load pretrain: {}".format(pretrain_pth))
                self.load_pretrain(pretrain_pth)
            else:
                logging.warning("pretrain %s not found -- starting from the geometric initialisation", pretrain_pth)
        latest_model_name = None
        if is_continue:
            model_list = [m for m in os.listdir(os.path.join(self.base_exp_dir, "checkpoints"))
                          if m[-3:] == "pth" and int(m[5:-4]) <= self.end_iter]
            model_list.sort()
            latest_model_name = model_list[-1]
        if latest_model_name is not None:
            logging.info("Find checkpoint: {}".format(latest_model_name))
            self.load_checkpoint(latest_model_name)
        if self.mode[:5] == "train" and self.rank == 0 and conf_path is not None and os.path.exists(str(conf_path)):
            self.file_backup()
        self.perceptor = None
        self.prior_renderer = None

    # ------------------------------------------------------------------ NeuS-init stage (main.py:180-256)
    def train(self):
        self.update_learning_rate()
        res_step = self.end_iter - self.iter_step
        image_perm = self.get_image_perm()
        for _ in range(res_step):
            data = self.dataset.gen_random_rays_at(image_perm[self.iter_step % len(image_perm)], self.batch_size)
            loss = self.train_iteration(data)
            if self.iter_step % self.report_freq == 0 and self.rank == 0:
                print("iter:{:8>d} loss = {} lr={}".format(self.iter_step, loss.item(), self.optimizer.param_groups[0]["lr"]))
            if self.iter_step % self.save_freq == 0 and self.rank == 0:
                self.save_checkpoint()
            self.update_learning_rate()
            if self.iter_step % len(image_perm) == 0:
                image_perm = self.get_image_perm()

    def train_iteration(self, data):
        rays_o, rays_d, true_rgb, mask = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
        near, far = self.dataset.near_far_from_sphere(rays_o, rays_d)
        background_rgb = torch.ones([1, 3], device=self.device) if self.use_white_bkgd else None
        mask = (mask > 0.5).float() if self.mask_weight > 0.0 else torch.ones_like(mask)
        mask_sum = mask.sum() + 1e-5
        render_out = self.renderer.render(rays_o, rays_d, near, far, background_rgb=background_rgb,
                                          cos_anneal_ratio=self.get_cos_anneal_ratio())
        color_error = (render_out["color_fine"] - true_rgb) * mask
        color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
        mask_loss = F.binary_cross_entropy(render_out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
        loss = color_fine_loss + render_out["gradient_error"] * self.igr_weight + mask_loss * self.mask_weight
        self.optimizer.zero_grad()
        loss.backward()
        parallel.allreduce_grads(self.params_to_train, self.world)
        self.optimizer.step()
        self.iter_step += 1
        return loss.detach()

    # ------------------------------------------------------------------ CLIP stage set-up (main.py:258-335)
    def init_clip(self, perceptor=None, text_embeddings=None, clip_state_dict=None):
        from . import clip_vit
        if perceptor is None:
            if clip_state_dict is None:
                path = self.conf.get_string("clip.weights", default=None)
                if path is not None and os.path.exists(path):
                    obj = torch.jit.load(path, map_location="cpu") if path.endswith(".pt") else torch.load(path, map_location="cpu")
                    clip_state_dict = obj.state_dict() if hasattr(obj, "state_dict") else obj
                else:
                    logging.warning("no CLIP ViT-B/32 weights supplied (clip.weights): using seeded random weights")
                    clip_state_dict = clip_vit_random_state_dict(0)
            perceptor = clip_vit.ClipVisionB32(clip_state_dict, self.device)
        self.perceptor = perceptor
        self.clip_preprocess = clip_vit.clip_preprocess
        te = text_embeddings or {}

        tokenizer = None
        if getattr(perceptor, "_text_sd", None) is not None:
            try:
                from .tokenizer import SimpleTokenizer
                tokenizer = SimpleTokenizer(self.conf.get_string("clip.bpe_path", default=None))
            except FileNotFoundError as e:
                logging.warning("%s", e)

        def emb(key, seed):
            if key in te:
                return te[key].to(self.device).float().reshape(1, -1)
            text = self.conf.get_string("clip." + key, default=None)
            if tokenizer is not None and text is not None:
                # main.py:272-288: clip.tokenize + perceptor.encode_text, detached
                from .tokenizer import tokenize
                print("%s: %s" % (key, text))
                return self.perceptor.encode_text(tokenize([text], tokenizer)).detach()
            logging.warning("no text embedding for clip.%s (needs full CLIP weights + the BPE table, or text_embeddings=): "
                            "using a seeded random unit vector", key)
            g = torch.Generator().manual_seed(seed)
            v = torch.randn(1, 512, generator=g)
            return (v / v.norm()).to(self.device)
        self.encoded_text = emb("prompt", 11)
        if self.use_face_prompt:
            self.encoded_face_text = emb("face_prompt", 12)
        if self.use_back_prompt:
            self.encoded_back_text = emb("back_prompt", 13)

    def init_smpl(self, prior_renderer=None):
        if prior_renderer is None:
            logging.warning("no SMPL prior renderer supplied: using the procedural ellipsoid prior")
            prior_renderer = EllipsoidPrior(device=self.device)
        self.prior_renderer = prior_renderer

    # ------------------------------------------------------------------ CLIP-guided loop (main.py:337-566)
    def train_clip(self):
        self.update_learning_rate()
        res_step = self.end_iter - self.iter_step
        for iter_i in range(res_step):
            if iter_i == 30010:  # main.py:346-347
                break
            loss = self.train_clip_iteration(iter_i)
            if self.iter_step % self.report_freq == 0 and self.rank == 0:
                print(self.base_exp_dir)
                print("iter:{:8>d} loss = {} lr={}".format(self.iter_step, loss.item(), self.optimizer.param_groups[0]["lr"]))
            if self.iter_step % self.save_freq == 0 and self.rank == 0:
                self.save_checkpoint()
            self.update_learning_rate()

    def sample_camera(self, iter_i):
        """main.py:348-359 (host numpy RNG, same draw order)."""
        if self.use_face_prompt and iter_i % 4 == 0:
            eye, theta, phi, is_front = random_eye(is_front=1, distance=0.4, theta_std=np.pi / 12)
            at = np.array([0, self.head_height, 0.3]).astype(np.float32)
        else:
            eye, theta, phi, is_front = random_eye_normal()
            at = random_at().astype(np.float32)
        eye = eye.astype(np.float32) + at
        return eye, at, theta, phi, is_front

    def train_clip_iteration(self, iter_i, camera=None):
        dev = self.device
        eye, at, theta, phi, is_front = camera if camera is not None else self.sample_camera(iter_i)
        pose = torch.from_numpy(lookat(eye, at, np.array([0, 1, 0]))).float().to(dev)
        prior = self.prior_renderer(eye, at)
        true_rgb = torch.as_tensor(prior, dtype=torch.float32, device=dev)
        ori_mask = (true_rgb != 0).float()[..., 0]
        dilated_mask = None
        if self.use_silhouettes:
            rays_o, rays_d, W, dilated_mask = self.dataset.gen_rays_silhouettes(pose, self.max_ray_num, ori_mask)
            H = W
            rays_o, rays_d = rays_o.float(), rays_d.float()
        else:
            rays_o, rays_d = self.dataset.gen_rays_pose(pose, self.full_frame_resolution_level)
            H, W = rays_o.shape[0], rays_o.shape[1]
            rays_o, rays_d = rays_o.reshape(H * W, 3).float(), rays_d.reshape(H * W, 3).float()
        Hp, Wp = true_rgb.shape[0], true_rgb.shape[1]
        true_rgb = F.interpolate(true_rgb.reshape(Hp, Wp, 3).permute(2, 0, 1).unsqueeze(0), size=(H, W)) \
            .squeeze(0).permute(1, 2, 0).reshape(-1, 3)                                        # main.py:376-377 (nearest)
        mask = (true_rgb != 0).float()[..., :1]
        near, far = self.dataset.near_far_from_sphere(rays_o, rays_d)
        # background augmentation (main.py:387-405)
        background_rgb = None
        choice_i = np.random.choice(4) if self.use_bg_aug else 3
        if choice_i == 0:
            background_rgb = torch.ones([1, 3], device=dev)
        elif choice_i == 1:
            gaussian = torch.normal(torch.zeros([H, W, 1], device=dev) + 0.5, torch.zeros([H, W, 1], device=dev) + 0.2)
            background_rgb = torch.clamp(gaussian, min=0, max=1).reshape(-1, 1)
        elif choice_i == 2:
            chess_board = torch.zeros([H, W, 1], device=dev) + 0.2
            chess_length = H // np.random.choice(np.arange(10, 20))
            ii = torch.arange(H, device=dev)[:, None] // chess_length
            jj = torch.arange(W, device=dev)[None, :] // chess_length
            chess_board[((ii + jj) % 2 == 0)] = 0.8
            sigma = float(np.random.uniform(0.1, 2.0))     # torchvision GaussianBlur(kernel (5,9), sigma U(0.1,2))
            background_rgb = _gaussian_blur(chess_board.permute(2, 0, 1).unsqueeze(0), (5, 9), sigma) \
                .squeeze(0).permute(1, 2, 0).reshape(-1, 1)
        mask = (mask > 0.5).float() if self.mask_weight > 0.0 else torch.ones_like(mask)
        if self.use_silhouettes and choice_i in (1, 2):
            masked_background_rgb = background_rgb.reshape(H, W, 1)[dilated_mask].reshape(-1, 1)
        else:
            masked_background_rgb = background_rgb
        mask_sum = mask.sum() + 1e-5
        render_out = self.renderer.render(rays_o, rays_d, near, far, background_rgb=masked_background_rgb,
                                          cos_anneal_ratio=self.get_cos_anneal_ratio())
        color_fine = render_out["color_fine"]
        extra_color_fine = render_out["extra_color_fine"]
        # cast light (main.py:426-453)
        if self.add_no_texture or self.texture_cast_light:
            normals = (render_out["gradients"] * render_out["weights"][:, :, None]).sum(dim=1)
            normals = normals / (torch.norm(normals, dim=-1, keepdim=True) + 1e-7)
            light_dir = sphere_coord(theta + np.random.uniform(-np.pi / 4, np.pi / 4), phi + np.random.uniform(-np.pi / 4, np.pi / 4))
            rand_light_d = torch.zeros_like(normals) + torch.from_numpy(light_dir).float().to(dev)
            rand_light_d = rand_light_d / (torch.norm(rand_light_d, dim=-1, keepdim=True) + 1e-7)
            rand_diffuse_shading = (normals * rand_light_d).sum(-1, keepdim=True).clamp(min=0, max=1)
            rand_diffuse_shading = torch.where(torch.isnan(rand_diffuse_shading), torch.ones_like(rand_diffuse_shading), rand_diffuse_shading)
            ambience = np.random.uniform(0, 0.2)
            rand_shading = ambience + (1 - ambience) * rand_diffuse_shading
            ws = render_out["weight_sum"].reshape(-1)
            bgm = (ws < 0.5)[:, None]
            rand_shading_rgb = torch.where(bgm, extra_color_fine, rand_shading.repeat(1, 3))
            rand_shading = torch.where(bgm, torch.ones_like(rand_shading), rand_shading)
            texture_shading = (extra_color_fine * rand_shading).clamp(min=0, max=1)
        weight_sum = render_out["weight_sum"]
        if self.use_silhouettes:  # scatter the masked rays back to full images (main.py:461-487)
            background = torch.zeros([H, W, 3], device=dev)
            if choice_i == 0:
                background[:] = 1
            if choice_i in (1, 2):
                background[~dilated_mask] = background_rgb.reshape(H, W, 1).repeat(1, 1, 3)[~dilated_mask]

            def scatter(vals, base):
                full = base.clone()
                full[dilated_mask] = vals
                return full.reshape(-1, vals.shape[-1])
            if self.add_no_texture or self.texture_cast_light:
                texture_shading = scatter(texture_shading, background)
                rand_shading_rgb = scatter(rand_shading_rgb, background)
            extra_color_fine = scatter(extra_color_fine, background)
            color_fine = scatter(color_fine, torch.zeros([H, W, 3], device=dev))
            weight_sum = scatter(weight_sum, torch.zeros([H, W, 1], device=dev))
        # losses (main.py:489-534)
        color_error = (color_fine - true_rgb) * mask
        color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
        eikonal_loss = render_out["gradient_error"]
        mask_loss = F.binary_cross_entropy(weight_sum.clip(1e-3, 1.0 - 1e-3), mask)
        if self.use_face_prompt and iter_i % 4 == 0:
            text = self.encoded_face_text
        elif self.use_back_prompt and is_front == 0:
            text = self.encoded_back_text
        else:
            text = self.encoded_text
        img = texture_shading if self.texture_cast_light else extra_color_fine
        if self.add_no_texture:
            # main.py:512 and :524 encode the two images in two calls; the encoder treats batch entries independently, so
            # one B=2 pass gives the same two embeddings with every frozen ViT weight streamed once instead of twice
            enc_both = self.perceptor.encode_image(torch.cat([self.clip_preprocess(img.reshape(H, W, 3)),
                                                              self.clip_preprocess(rand_shading_rgb.reshape(H, W, 3))], dim=0))
            enc, enc2 = enc_both[0:1], enc_both[1:2]
        else:
            enc = self.perceptor.encode_image(self.clip_preprocess(img.reshape(H, W, 3)))
        cosine = torch.cosine_similarity(torch.mean(enc, dim=0), torch.mean(text, dim=0), dim=0)
        loss = color_fine_loss + eikonal_loss * self.igr_weight + mask_loss * self.mask_weight + (1.0 - cosine) * self.clip_weight
        if self.add_no_texture:
            cosine_shading = torch.cosine_similarity(torch.mean(enc2, dim=0), torch.mean(text, dim=0), dim=0)
            loss = loss + (1.0 - cosine_shading) * self.clip_weight
        self.optimizer.zero_grad()
        loss.backward()
        parallel.allreduce_grads(self.params_to_train, self.world)     # one RCCL all-reduce per step (K17)
        self.optimizer.step()
        self.iter_step += 1
        self.last_stats = dict(loss=loss.detach(), color=color_fine_loss.detach(), eikonal=eikonal_loss.detach(),
                               cosine=cosine.detach(), rays=rays_o.shape[0])
        return loss.detach()

    # ------------------------------------------------------------------ schedule / checkpoints (main.py:568-632)
    def get_image_perm(self):
        return torch.randperm(max(self.dataset.n_images, 1))

    def get_cos_anneal_ratio(self):
        if self.anneal_end == 0.0:
            return 1.0
        return np.min([1.0, self.iter_step / self.anneal_end])

    def update_learning_rate(self):
        if self.iter_step < self.warm_up_end:
            learning_factor = self.iter_step / self.warm_up_end
        else:
            alpha = self.learning_rate_alpha
            progress = (self.iter_step - self.warm_up_end) / (self.end_iter - self.warm_up_end)
            learning_factor = (np.cos(np.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha
        for g in self.optimizer.param_groups:
            g["lr"] = self.learning_rate * learning_factor

    def file_backup(self):
        dir_lis = self.conf.get("general.recording", default=[])
        os.makedirs(os.path.join(self.base_exp_dir, "recording"), exist_ok=True)
        for dir_name in dir_lis:
            if not os.path.isdir(dir_name):
                continue
            cur_dir = os.path.join(self.base_exp_dir, "recording", dir_name)
            os.makedirs(cur_dir, exist_ok=True)
            for f_name in os.listdir(dir_name):
                if f_name[-3:] == ".py":
                    copyfile(os.path.join(dir_name, f_name), os.path.join(cur_dir, f_name))
        copyfile(self.conf_path, os.path.join(self.base_exp_dir, "recording", "config.conf"))

    def load_checkpoint(self, checkpoint_name):
        checkpoint = torch.load(os.path.join(self.base_exp_dir, "checkpoints", checkpoint_name), map_location=self.device,
                                weights_only=False)
        self.sdf_network.load_state_dict(checkpoint["sdf_network_fine"])
        self.deviation_network.load_state_dict(checkpoint["variance_network_fine"])
        self.color_network.load_state_dict(checkpoint["color_network_fine"])
        self.optimizer.load_state_dict(checkpoint["optimizer"])
        self.iter_step = checkpoint["iter_step"]

    def load_pretrain(self, checkpoint_name):
        checkpoint = torch.load(checkpoint_name, map_location=self.device, weights_only=False)
        self.sdf_network.load_state_dict(checkpoint["sdf_network_fine"])
        self.deviation_network.load_state_dict(checkpoint["variance_network_fine"])
        self.color_network.load_state_dict(checkpoint["color_network_fine"], strict=False)   # main.py:617

    def save_checkpoint(self):
        checkpoint = {
            "sdf_network_fine": self.sdf_network.state_dict(),
            "variance_network_fine": self.deviation_network.state_dict(),
            "color_network_fine": self.color_network.state_dict(),
            "optimizer": self.optimizer.state_dict(),
            "iter_step": self.iter_step,
        }
        os.makedirs(os.path.join(self.base_exp_dir, "checkpoints"), exist_ok=True)
        torch.save(checkpoint, os.path.join(self.base_exp_dir, "checkpoints", "ckpt_{:0>6d}.pth".format(self.iter_step)))

    @torch.no_grad()
    def validate_image(self, idx=-1, resolution_level=-1, pose=None):
        """Renders one view in chunks of `batch_size` rays (main.py:741-820, image assembly only)."""
        if resolution_level < 0:
            resolution_level = self.validate_resolution_level
        if pose is None:
            if idx < 0:
                idx = np.random.randint(self.dataset.n_images)
            pose = self.dataset.poses[idx]
        rays_o, rays_d = self.dataset.gen_rays_pose(pose, resolution_level)
        H, W, _ = rays_o.shape
        ro, rd = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
        outs = []
        with torch.enable_grad():
            for o, d in zip(ro.split(self.batch_size * 16), rd.split(self.batch_size * 16)):
                near, far = self.dataset.near_far_from_sphere(o, d)
                bg = torch.ones([1, 3], device=self.device) if self.use_white_bkgd else None
                out = self.renderer.render(o.contiguous(), d.contiguous(), near, far, perturb_overwrite=0, background_rgb=bg,
                                           cos_anneal_ratio=self.get_cos_anneal_ratio())
                outs.append((out["extra_color_fine"] if self.extra_color else out["color_fine"]).detach())
        return torch.cat(outs, 0).reshape(H, W, 3).clamp(0, 1)


    def validate_mesh(self, world_space=False, resolution=256, threshold=0.0):
        """main.py:850-919: marching cubes of -sdf over the dataset bounding box, vertex colours picked from six axis views
        (per vertex the view whose rendered depth is closest to the true camera-vertex distance), PLY export.
        (`world_space` is accepted and unused, as in the reference.)"""
        from . import mesh
        bound_min = torch.tensor(self.dataset.object_bbox_min, dtype=torch.float32)
        bound_max = torch.tensor(self.dataset.object_bbox_max, dtype=torch.float32)
        vertices, triangles = self.renderer.extract_geometry(bound_min, bound_max, resolution=resolution, threshold=threshold)
        os.makedirs(os.path.join(self.base_exp_dir, "meshes"), exist_ok=True)
        pt = torch.from_numpy(vertices).to(self.device).reshape(-1, 3).float()
        rgb_final, diff_final = None, None
        chunk = self.batch_size * 64      # rays per render call: no gradient is kept, only the panel-free forward runs
        for eye in ([0, 0, 2], [0, 0, -2], [0, 2, 0], [0, -2, 0], [2, 0, 0], [-2, 0, 0]):
            ro_all = torch.tensor(eye, dtype=torch.float32, device=self.device).reshape(1, 3).repeat(pt.shape[0], 1)
            rd_all = pt - ro_all
            dist = torch.norm(rd_all, dim=-1)
            rd_all = rd_all / dist.reshape(-1, 1)
            rgbs, diffs = [], []
            for ro, rd, di in zip(ro_all.split(chunk), rd_all.split(chunk), dist.split(chunk)):
                near, far = self.dataset.near_far_from_sphere(ro, rd)
                bg = torch.ones([1, 3], device=self.device) if self.use_white_bkgd else None
                with torch.no_grad():
                    out = self.renderer.render(ro.contiguous(), rd.contiguous(), near, far,
                                               cos_anneal_ratio=self.get_cos_anneal_ratio(), background_rgb=bg)
                rgbs.append(out["extra_color_fine"] if self.extra_color else out["color_fine"])
                depth = (out["mid_z_vals"] * out["weights"]).sum(dim=1)
                diffs.append((depth - di).abs())
            rgb, diff = torch.cat(rgbs, 0), torch.cat(diffs, 0)
            if rgb_final is None:
                rgb_final, diff_final = rgb.clone(), diff.clone()
            else:
                ind = diff_final > diff
                rgb_final[ind] = rgb[ind]
                diff_final[ind] = diff[ind]
        colors = (255 * np.clip(rgb_final.cpu().numpy(), 0, 1)).astype(np.uint8) if rgb_final is not None else None
        path = os.path.join(self.base_exp_dir, "meshes", "{:0>8d}.ply".format(self.iter_step))
        mesh.write_ply(path, vertices, triangles, colors)
        logging.info("mesh: %d vertices, %d triangles -> %s", vertices.shape[0], triangles.shape[0], path)
        return path


def clip_vit_random_state_dict(seed):
    """Seeded ViT-B/32 weights with OpenAI's init scales and key names (used only when no real weights are given)."""
    g = torch.Generator().manual_seed(seed)
    W, L, P, T, E = 768, 12, 32, 50, 512
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    sd = {"visual.conv1.weight": rn(W, 3, P, P, std=0.02), "visual.class_embedding": rn(W, std=W ** -0.5),
          "visual.positional_embedding": rn(T, W, std=W ** -0.5), "visual.proj": rn(W, E, std=W ** -0.5)}
    for n in ("ln_pre", "ln_post"):
        sd["visual.%s.weight" % n], sd["visual.%s.bias" % n] = torch.ones(W), torch.zeros(W)
    for i in range(L):
        p = "visual.transformer.resblocks.%d." % i
        sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"] = rn(3 * W, W, std=W ** -0.5), torch.zeros(3 * W)
        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = rn(W, W, std=W ** -0.5 * (2 * L) ** -0.5), torch.zeros(W)
        for n in ("ln_1", "ln_2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = torch.ones(W), torch.zeros(W)
        sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = rn(4 * W, W, std=(2 * W) ** -0.5), torch.zeros(4 * W)
        sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = rn(W, 4 * W, std=W ** -0.5 * (2 * L) ** -0.5), torch.zeros(W)
    return sd


def _gaussian_blur(x, ksize, sigma):
    """separable gaussian blur of [1,C,H,W] with reflect padding (torchvision.transforms.GaussianBlur semantics)."""
    def k1d(k):
        r = torch.arange(k, device=x.device, dtype=x.dtype) - (k - 1) / 2
        w = torch.exp(-0.5 * (r / sigma) ** 2)
        return w / w.sum()
    kx, ky = k1d(ksize[0]).tolist(), k1d(ksize[1]).tolist()
    H, W = x.shape[-2:]
    x = F.pad(x, (ksize[0] // 2, ksize[0] // 2, ksize[1] // 2, ksize[1] // 2), mode="reflect")
    # 5 + 9 shifted multiply-adds instead of F.conv2d: MIOpen answers a [1,1,H,W] depthwise convolution with its naive
    # kernel (1.4 ms per call in the profile, 2 % of a step)
    x = sum(w * x[..., :, k:k + W] for k, w in enumerate(kx))
    x = sum(w * x[..., k:k + H, :] for k, w in enumerate(ky))
    return x
