"""Runner: the reference's optimisation driver with the same constructor, conf keys, modes and checkpoint
format (AvatarGen/AppearanceGen/main.py:30-632), running its per-iteration hot path on the gfx950 kernels.

Differences that are deliberate and documented:
  * view-sharded data parallelism (not in the reference, which is single-GPU): when torch.distributed is
    initialised every rank draws its own camera and ONE flat-bucket all-reduce (RCCL over xGMI) averages the
    gradients before the identical Adam step;
  * the SMPL silhouette prior (smplx + neural_renderer, main.py:290-335,360) is rendered by the HIP rasteriser of
    smpl_prior.py from a posed mesh (`general.smpl_mesh`, an .obj, or `init_smpl(prior_renderer=...)`); the licensed SMPL
    pickles stay an input.  Seeded CLIP weights, seeded text embeddings, the procedural ellipsoid prior and a missing
    `train.pretrain` are STAND-INS for benchmarks and tests: they must be asked for (`allow_standins=True` /
    `general.allow_standins = True`), otherwise the Runner raises instead of training on meaningless inputs;
  * tensorboard logging is optional (absent offline) and scalar logging does not force a device sync every step.
"""
import importlib
import logging
import os
import random
import types
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
from . import h2d


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


class ScalarLog:
    """main.py:102 (`SummaryWriter(log_dir=os.path.join(base_exp_dir, 'logs'))`) without tensorboard (absent offline): the same
    `add_scalar(tag, value, step)` calls, written as JSON lines to <base_exp_dir>/logs/scalars.jsonl.  The values stay device
    tensors until `flush()` (one host transfer per report interval, not one per scalar and step)."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self.pending = []

    def add_scalar(self, tag, value, step):
        self.pending.append((tag, value.detach() if torch.is_tensor(value) else value, int(step)))

    def flush(self):
        if not self.pending:
            return
        import json
        with open(self.path, "a") as fh:
            for tag, v, step in self.pending:
                fh.write(json.dumps({"tag": tag, "value": float(v), "step": step}) + "\n")
        self.pending = []


class Runner:
    def __init__(self, conf_path, mode="train", case="CASE_NAME", is_continue=False, is_colab=False, conf=None,
                 device=None, data_root=None, allow_standins=None):
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
        self.allow_standins = bool(self.conf.get_bool("general.allow_standins", default=False)
                                   if allow_standins is None else allow_standins)
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
        self.seed = seed
        if seed is not None:
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
        if parallel.is_on():
            parallel.broadcast_params(params_to_train)
            self.seed_data_rngs()
        self.grad_bucket = parallel.GradBucket(params_to_train) if parallel.is_on() else None
        # main.py:145; fused = the same update in ONE launch for all 28 tensors instead of a dozen multi-tensor launches
        self.optimizer = torch.optim.Adam(params_to_train, lr=self.learning_rate, fused=(self.device.type == "cuda"))
        self.renderer = NeuSRenderer(self.nerf_outside, self.sdf_network, self.deviation_network, self.color_network,
                                     **self.conf["model.neus_renderer"])
        pretrain_pth = c.get_string("train.pretrain", default=None)
        if pretrain_pth is not None:
            if os.path.exists(pretrain_pth):
                logging.info("Load pretrain: {}".format(pretrain_pth))
                self.load_pretrain(pretrain_pth)
            elif self.allow_standins:
                logging.warning("pretrain %s not found -- starting from the geometric initialisation (stand-in)", pretrain_pth)
            else:
                raise FileNotFoundError("train.pretrain = %s does not exist (the reference fails here too, main.py:153-155); "
                                        "set general.allow_standins = True to start from the geometric initialisation"
                                        % pretrain_pth)
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

    def seed_data_rngs(self):
        """View-sharded data parallelism: the weights are identical on every rank (broadcast), the DATA must not be.  The
        reference seeds numpy / random / torch once (main.py:104-116); with that alone every rank would draw the same
        cameras, backgrounds, lights, image permutations, pixels and z-jitter and the all-reduce would average N identical
        gradients.  Rank r > 0 re-seeds its data RNGs with seed + r (rank 0 keeps the single-GPU stream); without
        train.seed a base seed is drawn on rank 0 and broadcast."""
        base = self.seed
        if base is None:
            t = torch.tensor([int(np.random.randint(0, 2 ** 31 - 1))], dtype=torch.int64, device=self.device)
            parallel.broadcast_tensor(t)
            base = int(t.item())
        self.data_seed = base + self.rank
        if self.rank > 0 or self.seed is None:
            np.random.seed(self.data_seed); random.seed(self.data_seed)
            torch.manual_seed(self.data_seed)
            if self.device.type == "cuda":
                torch.cuda.manual_seed(self.data_seed)

    def _validation_hooks(self, clip_stage):
        """main.py:247-251 / 556-561: periodic images and meshes (rank 0)."""
        if self.rank != 0:
            return
        if self.val_freq > 0 and self.iter_step % self.val_freq == 0:
            if clip_stage:
                self.validate_image(idx=58 if self.dataset.n_images > 58 else -1, save=True)
            else:
                self.validate_image(save=True)
        if self.val_mesh_freq > 0 and self.iter_step % self.val_mesh_freq == 0:
            self.validate_mesh()

    # ------------------------------------------------------------------ NeuS-init stage (main.py:180-256)
    def train(self):
        self._open_writer()
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
            self._validation_hooks(clip_stage=False)
            self.update_learning_rate()
            if self.iter_step % len(image_perm) == 0:
                image_perm = self.get_image_perm()
        if self.writer is not None:
            self.writer.flush()

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
        self.optimizer.zero_grad(set_to_none=self.grad_bucket is None)
        loss.backward()
        if self.grad_bucket is not None:
            self.grad_bucket.allreduce_mean()
        self.optimizer.step()
        self.iter_step += 1
        if self.writer is not None:   # main.py:216,230-238
            with torch.no_grad():
                psnr = 20.0 * torch.log10(1.0 / (((render_out["color_fine"] - true_rgb) ** 2 * mask).sum() / (mask_sum * 3.0)).sqrt())
                for tag, v in (("Loss/loss", loss), ("Loss/color_loss", color_fine_loss), ("Loss/eikonal_loss", render_out["gradient_error"]),
                               ("Statistics/s_val", render_out["s_val"][:1].mean()),
                               ("Statistics/cdf", (render_out["cdf_fine"][:, :1] * mask).sum() / mask_sum),
                               ("Statistics/weight_max", (render_out["weight_max"] * mask).sum() / mask_sum), ("Statistics/psnr", psnr)):
                    self.writer.add_scalar(tag, v, self.iter_step)
            if self.iter_step % self.report_freq == 0:
                self.writer.flush()
        return loss.detach()

    # ------------------------------------------------------------------ CLIP stage set-up (main.py:258-335)
    def _standin(self, what):
        if not self.allow_standins:
            raise RuntimeError("%s is missing and stand-ins are not allowed for this run (pass allow_standins=True or set "
                               "general.allow_standins = True for benchmarks / tests)" % what)
        logging.warning("%s is missing: using the seeded / procedural stand-in", what)

    def init_clip(self, perceptor=None, text_embeddings=None, clip_state_dict=None):
        from . import clip_vit
        if perceptor is None:
            if clip_state_dict is None:
                path = self.conf.get_string("clip.weights", default=None) or os.environ.get("AVC_CLIP_WEIGHTS")
                if path is not None and os.path.exists(path):
                    clip_state_dict = clip_vit.load_state_dict(path)
                else:
                    self._standin("the CLIP ViT-B/32 state dict (clip.weights / $AVC_CLIP_WEIGHTS)")
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
            self._standin("the text embedding of clip.%s (needs the full CLIP weights + the BPE table, or text_embeddings=)" % key)
            g = torch.Generator().manual_seed(seed)
            v = torch.randn(1, 512, generator=g)
            return (v / v.norm()).to(self.device)
        self.encoded_text = emb("prompt", 11)
        if self.use_face_prompt:
            self.encoded_face_text = emb("face_prompt", 12)
        if self.use_back_prompt:
            self.encoded_back_text = emb("back_prompt", 13)

    def init_smpl(self, prior_renderer=None):
        """main.py:290-335: the posed SMPL body whose renders supervise colour and mask.  `prior_renderer(eye, at)` returns
        the [256,256,3] render of `render_one_batch` (models/utils.py:108-125).  Sources, in order: the argument; the conf
        key `general.smpl_prior = module:callable` (called with this Runner); `general.smpl_mesh = <posed mesh .obj>`
        rendered by the HIP rasteriser (smpl_prior.MeshPrior); else the procedural ellipsoid -- a stand-in."""
        if prior_renderer is None:
            spec = self.conf.get_string("general.smpl_prior", default=None)
            mesh_path = self.conf.get_string("general.smpl_mesh", default=None)
            if spec is not None:
                mod, fn = spec.split(":")
                prior_renderer = getattr(importlib.import_module(mod), fn)(self)
            elif mesh_path is not None:
                from . import smpl_prior
                prior_renderer = smpl_prior.MeshPrior.from_obj(mesh_path, device=self.device)
            else:
                self._standin("the SMPL prior (general.smpl_mesh / general.smpl_prior / init_smpl(prior_renderer=...))")
                prior_renderer = EllipsoidPrior(device=self.device)
        self.prior_renderer = prior_renderer

    # ------------------------------------------------------------------ CLIP-guided loop (main.py:337-566)
    def _open_writer(self):
        """main.py:102: the scalar log of a training run (rank 0; `general.log_scalars = False` turns it off)"""
        if self.writer is None and self.rank == 0 and self.conf.get_bool("general.log_scalars", default=True):
            self.writer = ScalarLog(os.path.join(self.base_exp_dir, "logs"))

    def train_clip(self):
        self._open_writer()
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
            self._validation_hooks(clip_stage=True)
            self.update_learning_rate()
        if self.writer is not None:
            self.writer.flush()

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

    # The iteration is split into the stages of main.py so that each stage can be driven with injected inputs by the parity
    # tests (tests/test_glue_golden.py runs the glue on fixtures produced by the reference's own lines):
    #   make_view (main.py:348-385) -> draw_background (:387-415) -> renderer.render (:417-420)
    #   -> shade_and_scatter (:422-487) -> assemble_loss (:489-534) -> backward / all-reduce / Adam (:536-538)
    def make_view(self, iter_i, camera=None):
        dev = self.device
        eye, at, theta, phi, is_front = camera if camera is not None else self.sample_camera(iter_i)
        pose = h2d.upload(lookat(eye, at, np.array([0, 1, 0])), dev)
        prior = self.prior_renderer(eye, at)
        true_rgb = torch.as_tensor(prior, dtype=torch.float32, device=dev)
        if dev.type == "cuda" and os.environ.get("AVC_FUSED_HEAD", "1") != "0":
            # rays + near / far + the prior resampled to the ray grid in one launch (dataset.rays_fused) instead of ~45
            dilated_mask = sel_idx = None
            if self.use_silhouettes:
                grid = self.dataset.silhouette_grid(self.max_ray_num, true_rgb[..., 0])
                if grid is not None:
                    W, dilated_mask, sel_idx = grid
                    H = W
                else:
                    raise RuntimeError("the prior render of this view is empty: no silhouette to sample rays in (dataset.py:262-263)")
            else:
                W, H = int(self.dataset.W // self.full_frame_resolution_level), int(self.dataset.H // self.full_frame_resolution_level)
            rays_o, rays_d, near, far, true_rgb, mask = self.dataset.rays_fused(pose, W, H, sel_idx, true_rgb.reshape(true_rgb.shape[0], true_rgb.shape[1], 3))
            ray_of_pixel = None
            if sel_idx is not None:
                ray_of_pixel = torch.full((H * W,), -1, dtype=torch.int32, device=dev)
                ray_of_pixel[sel_idx] = torch.arange(sel_idx.numel(), dtype=torch.int32, device=dev)
            return types.SimpleNamespace(eye=eye, at=at, theta=theta, phi=phi, is_front=is_front, pose=pose, H=H, W=W,
                                         rays_o=rays_o, rays_d=rays_d, near=near, far=far, true_rgb=true_rgb, mask=mask,
                                         dilated_mask=dilated_mask, sel_idx=sel_idx, ray_of_pixel=ray_of_pixel, mask_binary=True)
        ori_mask = (true_rgb != 0).float()[..., 0]
        dilated_mask = sel_idx = None
        if self.use_silhouettes:
            self.dataset.last_sel_idx = None
            rays_o, rays_d, W, dilated_mask = self.dataset.gen_rays_silhouettes(pose, self.max_ray_num, ori_mask)
            sel_idx = getattr(self.dataset, "last_sel_idx", None)
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
        ray_of_pixel = None
        if sel_idx is not None and dev.type == "cuda":     # pixel -> ray (or -1): what the fused glue scatters with (glue.ShadeLossFn)
            ray_of_pixel = torch.full((H * W,), -1, dtype=torch.int32, device=dev)
            ray_of_pixel[sel_idx] = torch.arange(sel_idx.numel(), dtype=torch.int32, device=dev)
        return types.SimpleNamespace(eye=eye, at=at, theta=theta, phi=phi, is_front=is_front, pose=pose, H=H, W=W,
                                     rays_o=rays_o, rays_d=rays_d, near=near, far=far, true_rgb=true_rgb, mask=mask,
                                     dilated_mask=dilated_mask, sel_idx=sel_idx, ray_of_pixel=ray_of_pixel, mask_binary=True)

    def _take_view(self, iter_i, camera=None):
        """the view of this iteration: the one prefetch_view prepared, or a fresh one (silhouette mode: on the side stream)"""
        fut, self._view_future = getattr(self, "_view_future", None), None
        if fut is not None:
            view = fut[1].result()              # (always collected: the helper thread must not run into the next make_view)
            if camera is None and fut[0] == iter_i:
                return self._adopt_view(view)
        return self.make_view_on_side_stream(iter_i, camera) if self.use_silhouettes else self.make_view(iter_i, camera)

    def make_view_on_side_stream(self, iter_i, camera=None):
        """make_view for the silhouette mode, enqueued on a second HIP stream.  The ray set of that mode has a data-dependent size, so
        make_view has to bring two numbers to the host (dataset.gen_rays_silhouettes); on the main stream each of those round trips
        waits for EVERYTHING enqueued before it -- the previous iteration's backward pass -- and the host, which would otherwise run
        an iteration ahead of the GPU, falls into lock-step with it.  The view depends on nothing the optimiser writes: on its own
        stream its round trips wait for its own few small kernels only.  Host order -- and with it every numpy / torch random draw --
        is unchanged; the main stream waits for the view's event before its first use, and every tensor of the view is registered
        with the main stream so the caching allocator does not hand its memory back to the side stream while main-stream kernels
        still read it.  (In full-frame mode there is nothing to gain: no round trips, and small kernels do not become resident beside
        the persistent MLP kernels -- profiles/r03_side_stream.txt.)  AVC_OVERLAP_HEAD=0: everything on one stream."""
        if self.device.type != "cuda" or os.environ.get("AVC_OVERLAP_HEAD", "1") == "0":
            return self.make_view(iter_i, camera)
        return self._make_view_side(iter_i, camera)

    # ---- the next iteration's view, prepared beside this iteration's CLIP pass
    def _prefetch_allowed(self, iter_i):
        """The camera of iteration i + 1 may be drawn inside iteration i only when NOTHING draws from the host's numpy generator between
        the two in main.py's order: not when the validation hooks fire after this iteration (validate_image picks a random view,
        main.py:741-744), not on the last iteration of the run or at the reference's iter_i == 30010 break (a left-over view would have
        consumed draws the reference never makes).  Off unless AVC_PREFETCH_VIEW=1."""
        if (self.device.type != "cuda" or os.environ.get("AVC_PREFETCH_VIEW", "0") != "1" or os.environ.get("AVC_OVERLAP_HEAD", "1") == "0"):
            return False
        nxt = self.iter_step + 1          # the step count the hooks of THIS iteration will see
        if (self.val_freq > 0 and nxt % self.val_freq == 0) or (self.val_mesh_freq > 0 and nxt % self.val_mesh_freq == 0):
            return False
        return nxt < self.end_iter and iter_i + 1 != 30010

    def prefetch_view(self, iter_i, after=None):
        """Start make_view(iter_i) for the NEXT iteration on the side stream.  The view depends on nothing the optimiser writes, and
        its ~100-140 small kernels (camera, prior rasterisation, rays; 1.0-1.3 ms of GPU time) fit beside the only other stretch of
        small kernels in the iteration -- the CLIP pass, 24-96 workgroups on a 256-CU chip -- whereas beside the persistent MLP kernels
        nothing becomes resident (profiles/r03_side_stream.txt, r04_ab_kernels.txt).  clip_loss() calls this right before it launches
        CLIP, when every numpy draw of the current iteration (background, light, ambience) has been made: the camera of iteration
        i + 1 is drawn HERE, in the caller's thread, exactly where main.py:348-358 would draw it next -- clip_loss skips the prefetch on
        the iterations where something else draws in between (_prefetch_allowed: validation hooks, end of the run) -- so the draw order is
        unchanged.
        `after` = an event on the main stream the side stream waits for (the end of the render + shading work), so that the view's
        kernels start when the GPU reaches CLIP.  In silhouette mode make_view makes two round trips to the host (pixel counts -> ray
        grid -> ray count): a helper thread does that waiting.  Injected cameras (tests) bypass it.  OPT-IN (AVC_PREFETCH_VIEW=1):
        measured, the two streams of small kernels hardly overlap -- 6.3 vs 6.6 ms per iteration at 7 000 silhouette rays, but 9.27 vs
        9.14 at 12 544, 26.98 vs 26.85 at 224^2, 121.8 vs 122.1 at 512^2 (profiles/r04_ab_kernels.txt)."""
        if (self.device.type != "cuda" or os.environ.get("AVC_PREFETCH_VIEW", "0") != "1" or os.environ.get("AVC_OVERLAP_HEAD", "1") == "0"
                or getattr(self, "_view_future", None) is not None):
            return
        camera = self.sample_camera(iter_i)
        if self.use_silhouettes:
            if getattr(self, "_view_pool", None) is None:
                from concurrent.futures import ThreadPoolExecutor
                self._view_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="avc-view")
            self._view_future = (iter_i, self._view_pool.submit(self._make_view_side, iter_i, camera, False, after))
        else:
            self._view_future = (iter_i, _Done(self._make_view_side(iter_i, camera, False, after)))

    def _make_view_side(self, iter_i, camera, adopt=True, after=None):
        main = torch.cuda.current_stream(self.device)
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream(device=self.device)
            side.wait_stream(main)       # first use: whatever initialisation is still in flight on the main stream
        if after is not None:
            side.wait_event(after)
        with torch.cuda.device(self.device), torch.cuda.stream(side):
            view = self.make_view(iter_i, camera)
            view.ready = torch.cuda.Event()
            view.ready.record(side)
        return self._adopt_view(view) if adopt else view

    def _adopt_view(self, view):
        """the consuming (main) stream waits for the view's event; every tensor of the view is registered with it so that the
        caching allocator does not hand the memory back to the side stream while main-stream kernels still read it"""
        main = torch.cuda.current_stream(self.device)
        main.wait_event(view.ready)
        for t in vars(view).values():
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
        return view

    def draw_background(self, view, choice_i=None):
        """main.py:387-415 -> (choice_i, background_rgb [1,3] | [H*W,1] | None, what render() gets)."""
        dev, H, W = self.device, view.H, view.W
        background_rgb = None
        if choice_i is None:
            choice_i = np.random.choice(4) if self.use_bg_aug else 3
        if choice_i == 0:
            background_rgb = torch.ones([1, 3], device=dev)
        elif choice_i == 1:
            # (= torch.normal(zeros + 0.5, zeros + 0.2), main.py:393-394, draw for draw -- without the host-side check of the std TENSOR
            # that form makes, a stream synchronisation; test_gpu_iteration compares the two on the device)
            gaussian = torch.randn([H, W, 1], device=dev) * 0.2 + 0.5
            background_rgb = torch.clamp(gaussian, min=0, max=1).reshape(-1, 1)
        elif choice_i == 2:
            chess_length = H // np.random.choice(np.arange(10, 20))
            sigma = torch.empty(1).uniform_(0.1, 2.0).item()   # torchvision GaussianBlur.get_params: torch CPU RNG
            background_rgb = chess_background_fused(H, W, chess_length, sigma, dev) if dev.type == "cuda" and os.environ.get("AVC_FUSED_HEAD", "1") != "0" \
                else chess_background(H, W, chess_length, sigma, dev)
        if self.use_silhouettes and choice_i in (1, 2):
            idx = getattr(view, "sel_idx", None)      # (gather by index: boolean-mask indexing would synchronise the stream for the count)
            masked_background_rgb = background_rgb.reshape(-1, 1).index_select(0, idx) if idx is not None else \
                background_rgb.reshape(H, W, 1)[view.dilated_mask].reshape(-1, 1)
        else:
            masked_background_rgb = background_rgb
        return choice_i, background_rgb, masked_background_rgb

    def shade_and_scatter(self, render_out, view, choice_i, background_rgb, light=None):
        """main.py:422-487: Lambert shading from the rendered normals (random light around the camera, random ambience),
        then -- in silhouette mode -- the masked rays are scattered back into full images over the augmentation background.
        `light` = (light_dir[3], ambience) injects the numpy draws (tests)."""
        dev, H, W = self.device, view.H, view.W
        color_fine = render_out["color_fine"]
        extra_color_fine = render_out["extra_color_fine"]
        texture_shading = rand_shading_rgb = None
        if self.add_no_texture or self.texture_cast_light:
            normals = getattr(render_out, "weighted_normals", None)      # the same sum out of the compositing kernel (renderer.RenderOut)
            if normals is None:
                normals = (render_out["gradients"] * render_out["weights"][:, :, None]).sum(dim=1)
            normals = normals / (torch.norm(normals, dim=-1, keepdim=True) + 1e-7)
            if light is None:
                light_dir = sphere_coord(view.theta + np.random.uniform(-np.pi / 4, np.pi / 4),
                                         view.phi + np.random.uniform(-np.pi / 4, np.pi / 4))
            else:
                light_dir = np.asarray(light[0])
            rand_light_d = torch.zeros_like(normals) + h2d.upload(np.asarray(light_dir), dev)
            rand_light_d = rand_light_d / (torch.norm(rand_light_d, dim=-1, keepdim=True) + 1e-7)
            rand_diffuse_shading = (normals * rand_light_d).sum(-1, keepdim=True).clamp(min=0, max=1)
            rand_diffuse_shading = torch.where(torch.isnan(rand_diffuse_shading), torch.ones_like(rand_diffuse_shading), rand_diffuse_shading)
            ambience = np.random.uniform(0, 0.2) if light is None else float(light[1])
            rand_shading = ambience + (1 - ambience) * rand_diffuse_shading
            ws = render_out["weight_sum"].reshape(-1)
            bgm = (ws < 0.5)[:, None]
            rand_shading_rgb = torch.where(bgm, extra_color_fine, rand_shading.repeat(1, 3))
            rand_shading = torch.where(bgm, torch.ones_like(rand_shading), rand_shading)
            texture_shading = (extra_color_fine * rand_shading).clamp(min=0, max=1)
        weight_sum = render_out["weight_sum"]
        if self.use_silhouettes:  # scatter the masked rays back to full images (main.py:461-487)
            dilated_mask = view.dilated_mask
            background = torch.zeros([H, W, 3], device=dev)
            if choice_i == 0:
                background[:] = 1
            if choice_i in (1, 2):   # (= background[~dilated_mask] = bg[~dilated_mask], without the count a boolean index needs)
                background = torch.where(dilated_mask[..., None], background, background_rgb.reshape(H, W, 1).expand(H, W, 3))
            idx = getattr(view, "sel_idx", None)

            def scatter(vals, base):
                if idx is not None:      # row-major positions of the mask's pixels (dataset.gen_rays_silhouettes): no synchronisation
                    return base.reshape(-1, vals.shape[-1]).index_copy(0, idx, vals)
                full = base.clone()
                full[dilated_mask] = vals
                return full.reshape(-1, vals.shape[-1])
            if self.add_no_texture or self.texture_cast_light:
                texture_shading = scatter(texture_shading, background)
                rand_shading_rgb = scatter(rand_shading_rgb, background)
            extra_color_fine = scatter(extra_color_fine, background)
            color_fine = scatter(color_fine, torch.zeros([H, W, 3], device=dev))
            weight_sum = scatter(weight_sum, torch.zeros([H, W, 1], device=dev))
        return dict(color_fine=color_fine, extra_color_fine=extra_color_fine, weight_sum=weight_sum,
                    texture_shading=texture_shading, rand_shading_rgb=rand_shading_rgb)

    def assemble_loss(self, render_out, comp, view, iter_i):
        """main.py:489-534."""
        H, W = view.H, view.W
        mask = (view.mask > 0.5).float() if self.mask_weight > 0.0 else torch.ones_like(view.mask)
        mask_sum = mask.sum() + 1e-5
        color_error = (comp["color_fine"] - view.true_rgb) * mask
        color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
        psnr = None
        if self.writer is not None:   # main.py:493 (a logged statistic only: not computed when nothing records it)
            with torch.no_grad():
                psnr = 20.0 * torch.log10(1.0 / (((comp["color_fine"] - view.true_rgb) ** 2 * mask).sum() / (mask_sum * 3.0)).sqrt())
        eikonal_loss = render_out["gradient_error"]
        mask_loss = F.binary_cross_entropy(comp["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
        if self.use_face_prompt and iter_i % 4 == 0:
            text = self.encoded_face_text
        elif self.use_back_prompt and view.is_front == 0:
            text = self.encoded_back_text
        else:
            text = self.encoded_text
        img = comp["texture_shading"] if self.texture_cast_light else comp["extra_color_fine"]
        if self.add_no_texture:
            # main.py:512 and :524 encode the two images in two calls; the encoder treats batch entries independently, so
            # one B=2 pass gives the same two embeddings with every frozen ViT weight streamed once instead of twice
            enc_both = self.perceptor.encode_image(torch.cat([self.clip_preprocess(img.reshape(H, W, 3)),
                                                              self.clip_preprocess(comp["rand_shading_rgb"].reshape(H, W, 3))], dim=0))
            enc, enc2 = enc_both[0:1], enc_both[1:2]
        else:
            enc = self.perceptor.encode_image(self.clip_preprocess(img.reshape(H, W, 3)))
        cosine = torch.cosine_similarity(torch.mean(enc, dim=0), torch.mean(text, dim=0), dim=0)
        loss = color_fine_loss + eikonal_loss * self.igr_weight + mask_loss * self.mask_weight + (1.0 - cosine) * self.clip_weight
        cosine_shading = None
        if self.add_no_texture:
            cosine_shading = torch.cosine_similarity(torch.mean(enc2, dim=0), torch.mean(text, dim=0), dim=0)
            loss = loss + (1.0 - cosine_shading) * self.clip_weight
        return loss, dict(color=color_fine_loss, eikonal=eikonal_loss, mask=mask_loss, cosine=cosine, cosine_shading=cosine_shading,
                          psnr=psnr, s_val=render_out["s_val"][:1].mean())

    def fused_shade_loss(self, render_out, view, choice_i, background_rgb, iter_i, light=None):
        """shade_and_scatter + assemble_loss (main.py:422-534) through the two fused kernels of glue.py: same host draws in the same
        order (light direction, ambience), same images into CLIP, same loss terms -- ~150 small launches fewer per iteration
        (tests/test_gpu_glue.py: values and gradients against the torch statement).  AVC_FUSED_GLUE=0: the torch statement."""
        from . import glue
        dev, H, W = self.device, view.H, view.W
        P = H * W
        shading = self.add_no_texture or self.texture_cast_light
        light4 = nsum = None
        if shading:
            if light is None:
                light_dir = sphere_coord(view.theta + np.random.uniform(-np.pi / 4, np.pi / 4),
                                         view.phi + np.random.uniform(-np.pi / 4, np.pi / 4))
                ambience = np.random.uniform(0, 0.2)
            else:
                light_dir, ambience = np.asarray(light[0]), float(light[1])
            light4 = h2d.upload(glue.unit_light(light_dir, ambience), dev)
            nsum = getattr(render_out, "weighted_normals", None)
            if nsum is None:
                nsum = (render_out["gradients"] * render_out["weights"][:, :, None]).sum(dim=1)
        bg, bg_const, rop = None, 0.0, None
        if self.use_silhouettes:
            rop = view.ray_of_pixel
            if choice_i == 0:
                bg_const = 1.0
            elif choice_i in (1, 2):
                bg = background_rgb.reshape(-1)
        if self.mask_weight > 0.0:      # main.py:489: (mask > 0.5).float() -- the identity on the 0 / 1 masks make_view produces
            mask = view.mask if getattr(view, "mask_binary", False) else (view.mask > 0.5).float()
        else:
            mask = torch.ones_like(view.mask)
        images, sums = glue.ShadeLossFn.apply(
            render_out["color_fine"], render_out["extra_color_fine"], render_out["weight_sum"].reshape(-1), nsum, view.true_rgb,
            mask.reshape(-1), rop, bg, bg_const, light4, not self.texture_cast_light)
        psnr = None
        if self.writer is not None:   # main.py:493 (a logged statistic only)
            with torch.no_grad():
                psnr = 20.0 * torch.log10(1.0 / (sums[3] / ((sums[1] + 1e-5) * 3.0)).sqrt())
        eikonal_loss = render_out["gradient_error"]
        if self.use_face_prompt and iter_i % 4 == 0:
            text = self.encoded_face_text
        elif self.use_back_prompt and view.is_front == 0:
            text = self.encoded_back_text
        else:
            text = self.encoded_text
        B = 2 if self.add_no_texture else 1
        enc_both = self.perceptor.encode_image(glue.ResizeNormFn.apply(images[:B].reshape(B, H, W, 3)))
        # main.py:491-534 from here on (colour / mask normalisation, the cosines, the weighted sum) in one launch: glue.LossTailFn
        loss, st = glue.LossTailFn.apply(enc_both, text, sums, eikonal_loss, self.igr_weight, self.mask_weight, self.clip_weight, P)
        return loss, dict(color=st[1], eikonal=eikonal_loss, mask=st[2], cosine=st[3], cosine_shading=st[4] if self.add_no_texture else None,
                          psnr=psnr, s_val=render_out["s_val"][0, 0]), images

    def clip_loss(self, iter_i, camera=None):
        """main.py:348-534: one view from camera to scalar loss (differentiable)."""
        view = self._take_view(iter_i, camera)
        choice_i, background_rgb, masked_background_rgb = self.draw_background(view)
        render_out = self.renderer.render(view.rays_o, view.rays_d, view.near, view.far, background_rgb=masked_background_rgb,
                                          cos_anneal_ratio=self.get_cos_anneal_ratio())
        fused = (self.device.type == "cuda" and os.environ.get("AVC_FUSED_GLUE", "1") != "0"
                 and not (self.use_silhouettes and getattr(view, "ray_of_pixel", None) is None))
        if fused:
            # (its host draws -- light direction, ambience -- come first, as in shade_and_scatter; the prefetch draws the next camera)
            light = None
            if self.add_no_texture or self.texture_cast_light:
                light = (sphere_coord(view.theta + np.random.uniform(-np.pi / 4, np.pi / 4), view.phi + np.random.uniform(-np.pi / 4, np.pi / 4)),
                         np.random.uniform(0, 0.2))
        else:
            comp = self.shade_and_scatter(render_out, view, choice_i, background_rgb)
        if camera is None and self._prefetch_allowed(iter_i):
            # every host draw of this iteration is made: the next view is prepared beside the CLIP pass that follows
            reached = torch.cuda.Event()
            reached.record(torch.cuda.current_stream(self.device))
            self.prefetch_view(iter_i + 1, after=reached)
        if fused:
            loss, parts, _ = self.fused_shade_loss(render_out, view, choice_i, background_rgb, iter_i, light=light)
        else:
            loss, parts = self.assemble_loss(render_out, comp, view, iter_i)
        self.last_view = view
        return loss, parts

    def train_clip_iteration(self, iter_i, camera=None):
        release = getattr(self.perceptor, "release_graphs", None)
        if release is not None:
            release()     # a graph-replayed encode_image of an earlier iteration that never reached backward (exception, probe call) does not hold its instance
        loss, parts = self.clip_loss(iter_i, camera)
        self.optimizer.zero_grad(set_to_none=self.grad_bucket is None)
        loss.backward()
        if self.grad_bucket is not None:
            self.grad_bucket.allreduce_mean()     # one RCCL all-reduce per step (K17)
        self.optimizer.step()
        self.iter_step += 1
        self.last_stats = dict(loss=loss.detach(), color=parts["color"].detach(), eikonal=parts["eikonal"].detach(),
                               cosine=parts["cosine"].detach(), rays=self.last_view.rays_o.shape[0], s_val=parts["s_val"], psnr=parts["psnr"])
        if self.writer is not None:   # main.py:542-547
            for tag, v in (("Loss/loss", loss), ("Loss/color_loss", parts["color"]), ("Loss/eikonal_loss", parts["eikonal"]),
                           ("Loss/cosine", parts["cosine"]), ("Statistics/s_val", parts["s_val"]), ("Statistics/psnr", parts["psnr"])):
                self.writer.add_scalar(tag, v, self.iter_step)
            if self.iter_step % self.report_freq == 0:
                self.writer.flush()
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
            g["lr"] = float(self.learning_rate * learning_factor)   # (a Python float: a numpy scalar here would make the checkpoint more than tensors and plain containers)

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

    def _torch_load(self, path):
        """checkpoints of this stage hold tensors, the optimizer's plain containers and an int: the tensors-only unpickler reads
        them without executing code from the file.  A file it rejects is loaded with the full unpickler -- which executes code from
        the file, as the reference's plain torch.load does -- only with the same explicit opt-in as the other loaders
        (AVC_ALLOW_UNSAFE_PICKLE=1: clip_vit.load_state_dict, smpl_lbs.load_smpl_arrays); I/O errors are not retried."""
        import pickle
        try:
            # (numpy scalars -- the learning rate the reference's update_learning_rate leaves in the optimizer's param groups -- are
            # data, not code: allow-listed for the tensors-only unpickler)
            safe = [np.dtype, type(np.dtype(np.float64)), type(np.dtype(np.float32)), type(np.dtype(np.int64))]
            try:
                from numpy._core.multiarray import scalar as np_scalar       # numpy 2
            except ImportError:
                from numpy.core.multiarray import scalar as np_scalar        # numpy 1
            # (files written under numpy 1 -- the reference's shipped checkpoint -- name it by its old module path)
            safe += [np_scalar, (np_scalar, "numpy.core.multiarray.scalar"), (np_scalar, "numpy._core.multiarray.scalar")]
            with torch.serialization.safe_globals(safe):
                return torch.load(path, map_location=self.device, weights_only=True)
        except pickle.UnpicklingError as e:
            if os.environ.get("AVC_ALLOW_UNSAFE_PICKLE") != "1":
                raise RuntimeError("%s is not a tensors-only checkpoint (%s); loading it would execute pickled code (set "
                                   "AVC_ALLOW_UNSAFE_PICKLE=1 to allow that)" % (path, str(e)[:200])) from e
            logging.warning("%s is not a tensors-only checkpoint: loading it with the full unpickler (AVC_ALLOW_UNSAFE_PICKLE=1)", path)
            return torch.load(path, map_location=self.device, weights_only=False)

    def load_checkpoint(self, checkpoint_name):
        checkpoint = self._torch_load(os.path.join(self.base_exp_dir, "checkpoints", checkpoint_name))
        self.sdf_network.load_state_dict(checkpoint["sdf_network_fine"])
        self.deviation_network.load_state_dict(checkpoint["variance_network_fine"])
        self.color_network.load_state_dict(checkpoint["color_network_fine"])
        self.optimizer.load_state_dict(checkpoint["optimizer"])
        self.iter_step = checkpoint["iter_step"]

    def load_pretrain(self, checkpoint_name):
        checkpoint = self._torch_load(checkpoint_name)
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

    def _render_chunks(self, rays_o, rays_d, keys, chunk=None, **render_kw):
        """render() over chunks of rays without keeping a graph (main.py:752-783 and its siblings); returns dict of cats."""
        chunk = chunk or self.batch_size * 16     # the kernels want >= a few thousand rays per launch; the result is chunk-invariant
        outs = {k: [] for k in keys}
        bg = torch.ones([1, 3], device=self.device) if self.use_white_bkgd else None
        for o, d in zip(rays_o.split(chunk), rays_d.split(chunk)):
            near, far = self.dataset.near_far_from_sphere(o, d)
            with torch.no_grad():
                out = self.renderer.render(o.contiguous(), d.contiguous(), near, far, background_rgb=render_kw.get("background_rgb", bg),
                                           cos_anneal_ratio=self.get_cos_anneal_ratio(),
                                           perturb_overwrite=render_kw.get("perturb_overwrite", -1))
            for k in keys:
                if k == "normals":   # main.py:773-778
                    n = out["gradients"] * out["weights"][:, :, None]
                    if render_kw.get("inside_only", True) and out.get("inside_sphere") is not None:
                        n = n * out["inside_sphere"][..., None]
                    outs[k].append(n.sum(dim=1))
                else:
                    outs[k].append(out[k].detach())
        return {k: torch.cat(v, 0) for k, v in outs.items()}

    def validate_image(self, idx=-1, resolution_level=-1, pose=None, save=False, perturb_overwrite=-1):
        """main.py:741-820.  Returns the [H,W,3] image (extra_color when the colour net has the CLIP head); with save=True
        also writes validations_fine/, validations_extra_fine/ and normals/ PNGs under base_exp_dir with the reference's
        file names and channel conventions (cv.imwrite takes BGR: the colour and normal images are written un-converted,
        i.e. channel-swapped on disk, the extra-colour image is converted first -- reproduced literally)."""
        if resolution_level < 0:
            resolution_level = self.validate_resolution_level
        if pose is None:
            if idx < 0:
                idx = np.random.randint(self.dataset.n_images)
            pose = self.dataset.poses[idx]
        print("Validate: iter: {}, camera: {}".format(self.iter_step, idx))
        rays_o, rays_d = self.dataset.gen_rays_pose(pose, resolution_level)
        H, W, _ = rays_o.shape
        keys = ["color_fine", "normals"] + (["extra_color_fine"] if self.extra_color else [])
        res = self._render_chunks(rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float(), keys, perturb_overwrite=perturb_overwrite)
        img_fine = (res["color_fine"].reshape(H, W, 3) * 255).clip(0, 255)
        extra = (res["extra_color_fine"].reshape(H, W, 3) * 255).clip(0, 255) if self.extra_color else None
        rot = torch.linalg.inv(torch.as_tensor(pose)[:3, :3].to(self.device).float())
        normal_img = (torch.matmul(rot[None], res["normals"][:, :, None]).reshape(H, W, 3) * 128 + 128).clip(0, 255)
        self.last_validation = dict(color=img_fine, extra=extra, normal=normal_img)
        if save:
            from PIL import Image
            tag = "{:0>8d}_{}_{}.png".format(self.iter_step, 0, idx)
            u8 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy().astype(np.uint8))
            for sub in ("validations_fine", "validations_extra_fine", "normals"):
                os.makedirs(os.path.join(self.base_exp_dir, sub), exist_ok=True)
            fine = u8(img_fine)
            if idx >= 0 and self.dataset.images_lis:
                fine = np.concatenate([fine, self.dataset.image_at(idx, resolution_level=resolution_level)])
            Image.fromarray(np.ascontiguousarray(fine[..., ::-1])).save(os.path.join(self.base_exp_dir, "validations_fine", tag))
            if extra is not None:
                Image.fromarray(u8(extra)).save(os.path.join(self.base_exp_dir, "validations_extra_fine", tag))
            Image.fromarray(np.ascontiguousarray(u8(normal_img)[..., ::-1])).save(os.path.join(self.base_exp_dir, "normals", tag))
        return ((extra if extra is not None else img_fine) / 255.0).clamp(0, 1)

    def render_novel_image(self, idx_0, idx_1, ratio, resolution_level):
        """main.py:822-848: the view interpolated between cameras idx_0 and idx_1 -> uint8 [H,W,3] (x256 like the reference)."""
        rays_o, rays_d = self.dataset.gen_rays_between(idx_0, idx_1, ratio, resolution_level=resolution_level)
        H, W, _ = rays_o.shape
        res = self._render_chunks(rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float(), ["color_fine"])
        return (res["color_fine"].reshape(H, W, 3).cpu().numpy() * 256).clip(0, 255).astype(np.uint8)

    def interpolate_view(self, img_idx_0, img_idx_1, n_frames=60, resolution_level=4):
        """main.py:921-944: 60 frames forth and back between two cameras.  cv2.VideoWriter is not available offline: the
        frames are written as an animated GIF (30 fps) plus numbered PNGs under base_exp_dir/render/."""
        from PIL import Image
        images = [self.render_novel_image(img_idx_0, img_idx_1, np.sin(((i / n_frames) - 0.5) * np.pi) * 0.5 + 0.5,
                                          resolution_level=resolution_level) for i in range(n_frames)]
        images = images + images[::-1]
        video_dir = os.path.join(self.base_exp_dir, "render")
        os.makedirs(video_dir, exist_ok=True)
        stem = os.path.join(video_dir, "{:0>8d}_{}_{}".format(self.iter_step, img_idx_0, img_idx_1))
        frames = [Image.fromarray(im) for im in images]
        frames[0].save(stem + ".gif", save_all=True, append_images=frames[1:], duration=33, loop=0)
        return stem + ".gif"

    def render_geometry_cast_light(self, light=None):
        """main.py:634-739: 512x512 close-up of the head, textured colour under one random Lambert light (ambience 0, black
        background) -> base_exp_dir/cast_light_texture_head_black.png."""
        phi, theta, camera_distance = 0, 0, 0.5
        eye = np.array([camera_distance * np.sin(theta) * np.cos(phi), camera_distance * np.sin(theta) * np.sin(phi),
                        camera_distance * np.cos(theta)])
        at = np.array([0, self.head_height, 0.3])
        eye = eye + at
        pose = torch.from_numpy(lookat(eye, at, np.array([0, 1, 0]))).float()
        rays_o, rays_d = self.dataset.gen_rays_pose(pose, 0.5)
        H, W = rays_o.shape[0], rays_o.shape[1]
        if light is None:
            light = sphere_coord(theta + np.random.uniform(-np.pi / 4, np.pi / 4), phi + np.random.uniform(-np.pi / 4, np.pi / 4))
        np.random.choice(np.arange(10, 20))      # main.py:676 draws the chess length although choice_i is fixed to 3
        res = self._render_chunks(rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float(),
                                  ["extra_color_fine" if self.extra_color else "color_fine", "normals", "weight_sum"],
                                  background_rgb=None, inside_only=False)
        extra = res["extra_color_fine" if self.extra_color else "color_fine"]
        normals = res["normals"] / (torch.norm(res["normals"], dim=-1, keepdim=True) + 1e-7)
        ld = torch.from_numpy(np.asarray(light)).float().to(self.device)
        ld = (torch.zeros_like(normals) + ld)
        ld = ld / (torch.norm(ld, dim=-1, keepdim=True) + 1e-7)
        shading = (normals * ld).sum(-1, keepdim=True).clamp(min=0, max=1)
        shading = torch.where(torch.isnan(shading), torch.ones_like(shading), shading)
        shading = torch.where((res["weight_sum"].reshape(-1) < 0.5)[:, None], torch.ones_like(shading), shading)
        img = (extra * shading).clamp(0, 1).reshape(H, W, 3)
        from PIL import Image
        path = os.path.join(self.base_exp_dir, "cast_light_texture_head_black.png")
        Image.fromarray((255 * np.clip(img.cpu().numpy(), 0, 1)).astype(np.uint8)).save(path)   # to8b
        return path

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


class _Done:
    """a finished future (the full-frame view needs no helper thread: no round trips to wait for)"""

    def __init__(self, value):
        self._v = value

    def result(self):
        return self._v


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


def chess_background(H, W, chess_length, sigma, device):
    """main.py:398-405: 0.2 / 0.8 chess board of `chess_length`-pixel squares, GaussianBlur(kernel (5,9), sigma) -> [H*W,1]."""
    chess_board = torch.zeros([H, W, 1], device=device) + 0.2
    ii = torch.arange(H, device=device)[:, None] // chess_length
    jj = torch.arange(W, device=device)[None, :] // chess_length
    chess_board[((ii + jj) % 2 == 0)] = 0.8
    return _gaussian_blur(chess_board.permute(2, 0, 1).unsqueeze(0), (5, 9), float(sigma)).squeeze(0).permute(1, 2, 0).reshape(-1, 1)


def chess_background_fused(H, W, chess_length, sigma, device):
    """chess_background in one launch (csrc/avc_glue.hip: avc_chess_background; the 14 blur taps are computed on the host)"""
    from . import lib as L
    def k1d(k):
        r = np.arange(k, dtype=np.float32) - np.float32((k - 1) / 2)
        w = np.exp(np.float32(-0.5) * (r / np.float32(sigma)) ** 2).astype(np.float32)
        return w / w.sum(dtype=np.float32)
    taps = h2d.upload(np.concatenate([k1d(5), k1d(9)]), device)
    out = torch.empty(H * W, 1, device=device, dtype=torch.float32)
    L.check(L.load().avc_chess_background(L.ptr(out), H, W, int(chess_length), L.ptr(taps), L.stream()), "avc_chess_background")
    return out


def _gaussian_blur(x, ksize, sigma):
    """separable gaussian blur of [1,C,H,W] with reflect padding (torchvision.transforms.GaussianBlur semantics)."""
    def k1d(k):      # (on the host: the taps are nine numbers, and a .tolist() of a device tensor would synchronise the stream)
        r = torch.arange(k, dtype=x.dtype) - (k - 1) / 2
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
