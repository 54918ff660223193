"""NeuSRenderer with the reference's constructor and `render` contract
(AvatarGen/AppearanceGen/models/renderer.py:72-404), executed by the fused gfx950 kernels:

  coarse z + jitter (renderer.py:304-319)                      torch elementwise on device
  SDF at the coarse samples (:337-338)                         avc_sdf_forward        (f16 MFMA)
  4x { up_sample + sample_pdf + cat_z_vals } (:340-352)        avc_upsample_step + avc_sdf_forward
  render_core (:195-300)                                       avc_render_points_fwd + avc_composite_fwd,
                                                               backward: avc_composite_bwd, avc_render_points_bwd,
                                                               avc_weight_grad (engine.RenderCoreFn)
"""
import torch

from .engine import Engine, RenderCoreFn, flatten_dense


class RenderOut(dict):
    """the reference's 11-key result dict (renderer.py:385-397) plus ONE attribute that is not a key: `weighted_normals` [R,3] =
    (gradients * weights[:, :, None]).sum(dim=1), the first line of the shading of main.py:428, which the compositing kernel
    has in registers anyway (differentiable like the tensors of the dict; Runner.shade_and_scatter takes it when it is there)"""
    weighted_normals = None


class NeuSRenderer:
    def __init__(self, nerf, sdf_network, deviation_network, color_network, n_samples, n_importance, n_outside,
                 up_sample_steps, perturb, extra_color=False):
        if n_outside and n_outside > 0:
            raise NotImplementedError("n_outside > 0 (background NeRF) is unused by AppearanceGen (main.py:136) "
                                      "and not implemented")
        self.nerf = nerf
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.n_samples = int(n_samples)
        self.n_importance = int(n_importance)
        self.n_outside = 0
        self.up_sample_steps = int(up_sample_steps)
        self.perturb = perturb
        self.extra_color = bool(extra_color)
        self._engine = None

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine.for_networks(self.sdf_network, self.color_network)
        return self._engine

    def flat_params(self):
        return flatten_dense(self.sdf_network, self.color_network, self.engine.spec)

    # renderer.py:304-352 (no autograd, like the reference's torch.no_grad block)
    @torch.no_grad()
    def sample_z(self, pk, rays_o, rays_d, near, far, perturb, jitter=None, return_steps=False):
        eng = self.engine
        R = rays_o.shape[0]
        dev = rays_o.device
        if perturb > 0 and jitter is None:
            jitter = torch.rand([R, 1], device=dev)
        from .engine import FUSED_PACK
        if dev.type == "cuda" and FUSED_PACK and near.numel() == R and far.numel() == R:
            from . import lib as L
            near_c, far_c = near.reshape(R).contiguous().float(), far.reshape(R).contiguous().float()
            jit = jitter.reshape(R).contiguous().float() if perturb > 0 else None
            z = torch.empty(R, self.n_samples, device=dev, dtype=torch.float32)
            L.check(L.load().avc_coarse_z(L.ptr(near_c), L.ptr(far_c), L.ptr(jit), R, self.n_samples, L.ptr(z), L.stream()), "avc_coarse_z")
        else:
            z = torch.linspace(0.0, 1.0, self.n_samples, device=dev)
            z = near + (far - near) * z[None, :]
            if perturb > 0:
                z = z + (jitter - 0.5) * 2.0 / self.n_samples
            z = z.contiguous()
        steps = []
        if self.n_importance > 0:
            sdf = eng.sdf_rays(pk, rays_o, rays_d, z)
            m = self.n_importance // self.up_sample_steps
            for i in range(self.up_sample_steps):
                z_out, sdf_out, z_new, slot = eng.upsample_step(rays_o, rays_d, z, sdf, m, 64 * 2 ** i)
                if return_steps:
                    steps.append(dict(z_in=z, sdf_in=sdf, new_z=z_new))
                if i + 1 < self.up_sample_steps:
                    eng.sdf_rays(pk, rays_o, rays_d, z_new, sdf_out=sdf_out, slot=slot, ld_out=z_out.shape[1])
                z, sdf = z_out, sdf_out
        return (z, steps) if return_steps else z

    def render_core(self, rays_o, rays_d, z_vals, sample_dist, background_rgb=None, cos_anneal_ratio=0.0, flatP=None):
        eng = self.engine
        if flatP is None:
            flatP = self.flat_params()
        inv_s, s_val = self.deviation_network.inv_s_and_s_val()
        R, S = z_vals.shape
        bg, bg_mode = None, 0
        if background_rgb is not None and self.extra_color:
            bgt = background_rgb.to(rays_o.device).float()
            # by SHAPE ([1,3] colour vs [R,1] per-ray grey), not by element count: R == 3 is a legal ray count
            if bgt.dim() == 2 and bgt.shape[0] == 1 and bgt.shape[1] == 3 or bgt.dim() == 1 and bgt.numel() == 3 and R != 3:
                bg, bg_mode = bgt.reshape(3).contiguous(), 1
            elif bgt.numel() == R and (bgt.dim() == 1 or bgt.shape[-1] == 1):
                bg, bg_mode = bgt.reshape(R).contiguous(), 2
            else:
                raise ValueError("background_rgb must be [1,3] or [R,1] (main.py:387-415)")
        color, extra, weights, gradients, gerr, cdf, mid_z, inside, sdf, wsum, wmax, nsum = RenderCoreFn.apply(
            flatP, inv_s, eng, rays_o, rays_d, z_vals, float(sample_dist), float(cos_anneal_ratio), bg, bg_mode)
        if not self.extra_color:
            extra = None
            if background_rgb is not None:  # renderer.py:280-281
                color = color + background_rgb.to(color.device) * (1.0 - wsum)
        return {"color": color, "extra_color": extra, "sdf": sdf.reshape(-1, 1), "gradients": gradients,
                "s_val": s_val, "mid_z_vals": mid_z, "weights": weights, "cdf": cdf,
                "gradient_error": gerr, "inside_sphere": inside, "weight_sum": wsum, "weight_max": wmax, "weighted_normals": nsum}

    def render(self, rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0,
               jitter=None, z_vals=None):
        """Same contract as the reference (renderer.py:302-397).  `jitter` ([R,1] uniform) and `z_vals` are
        optional injection points used by the parity tests; they default to the reference behaviour."""
        rays_o = rays_o.contiguous().float()
        rays_d = rays_d.contiguous().float()
        sample_dist = 2.0 / self.n_samples
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        flatP = self.flat_params()
        if z_vals is None:
            pk = self.engine.pack(flatP)
            z_vals = self.sample_z(pk, rays_o, rays_d, near.float(), far.float(), perturb, jitter)
        z_vals = z_vals.contiguous()
        ret = self.render_core(rays_o, rays_d, z_vals, sample_dist, background_rgb, cos_anneal_ratio, flatP)
        weights = ret["weights"]
        R, S = weights.shape
        out = RenderOut({
            "color_fine": ret["color"],
            "extra_color_fine": ret["extra_color"],
            "s_val": ret["s_val"].expand(R, 1),
            "cdf_fine": ret["cdf"],
            "weight_sum": ret["weight_sum"],          # = weights.sum(-1, keepdim=True) / torch.max(weights, -1, keepdim=True)[0]
            "weight_max": ret["weight_max"],          #   (renderer.py:391-392), reduced inside the compositing kernel
            "gradients": ret["gradients"],
            "weights": weights,
            "mid_z_vals": ret["mid_z_vals"],
            "gradient_error": ret["gradient_error"],
            "inside_sphere": ret["inside_sphere"],
        })
        out.weighted_normals = ret["weighted_normals"]
        return out

    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.0):
        """renderer.py:399-404: marching cubes of -sdf at `threshold` on a resolution^3 grid -> (vertices, triangles)"""
        from . import mesh
        dev = next(self.sdf_network.parameters()).device
        pk = self.engine.pack(self.flat_params())
        return mesh.extract_geometry(bound_min, bound_max, resolution, threshold,
                                     lambda pts: -self.engine.sdf_pts(pk, pts), dev)
