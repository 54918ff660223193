"""Neural fields of the AppearanceGen stage with the reference's constructor signatures and state-dict keys
(reference: AvatarGen/AppearanceGen/models/fields.py:9-107 SDFNetwork, :111-185 RenderingNetwork,
:270-276 SingleVarianceNetwork; models/embedder.py:6-51).

The modules own the parameters (weight-normed linears -> `linK.weight_g / weight_v / bias`, `extra_lin.*`,
`variance`, so reference checkpoints load unmodified, main.py:601-632).  The hot path never calls their
`forward`: avatarclip_amd.renderer.NeuSRenderer feeds the dense weights to the fused gfx950 kernels.
`SDFNetwork.sdf` (used by extract_geometry) runs the HIP SDF kernel.  `forward` / `gradient` are kept as plain
differentiable torch expressions for API compatibility only.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def embed(x, multires):
    """embedder.py:35-36 -- [x, sin(2^k x), cos(2^k x)]_{k<L}."""
    if multires <= 0:
        return x
    outs = [x]
    for k in range(multires):
        outs.append(torch.sin(x * (2.0 ** k)))
        outs.append(torch.cos(x * (2.0 ** k)))
    return torch.cat(outs, -1)


def _wn(lin):
    return torch.nn.utils.weight_norm(lin)


def dense_weight(lin):
    """W = g * v / ||v||_row for a weight-normed linear (fields.py:65-66), or the plain weight."""
    if hasattr(lin, "weight_g"):
        v = lin.weight_v
        return lin.weight_g * v / v.norm(dim=1, keepdim=True)
    return lin.weight


_TORCH_PATH_SEEN = set()


def _torch_module_path(module, what, x):
    """The reference's module-level entry points (fields.py:72-107, 154-185) are kept for API compatibility as plain torch expressions
    (rocBLAS through torch); the product's hot path never calls them -- NeuSRenderer.render and SDFNetwork.sdf go to the HIP engine.  A
    caller that reaches them on the GPU is NOT on the accelerated path and is told so, once per entry point: AVC_TORCH_MODULE_PATH = warn
    (default) | raise | quiet."""
    if not x.is_cuda or what in _TORCH_PATH_SEEN:
        return
    mode = os.environ.get("AVC_TORCH_MODULE_PATH", "warn")
    if mode == "raise":
        raise RuntimeError("%s on a CUDA tensor is the torch API-compatibility expression, not the HIP engine (use NeuSRenderer.render / "
                           "SDFNetwork.sdf); AVC_TORCH_MODULE_PATH=raise refuses it" % what)
    _TORCH_PATH_SEEN.add(what)
    if mode != "quiet":
        import warnings
        warnings.warn("%s on the GPU runs as plain torch ops (rocBLAS), NOT on the fused gfx950 kernels: the accelerated entry points are "
                      "NeuSRenderer.render and SDFNetwork.sdf (AVC_TORCH_MODULE_PATH=raise | quiet)" % what, RuntimeWarning, stacklevel=3)


class SDFNetwork(nn.Module):
    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1,
                 geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        dims = [d_in] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.multires = multires
        if multires > 0:
            dims[0] = d_in + d_in * 2 * multires
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        self.conf = dict(d_in=d_in, d_out=d_out, d_hidden=d_hidden, n_layers=n_layers, skip_in=list(skip_in),
                         multires=multires, scale=scale)
        for l in range(0, self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if l + 1 in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:  # fields.py:45-63
                if l == self.num_layers - 2:
                    sgn = -1.0 if inside_outside else 1.0
                    torch.nn.init.normal_(lin.weight, mean=sgn * np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    torch.nn.init.constant_(lin.bias, -sgn * bias)
                elif multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            if weight_norm:
                lin = _wn(lin)
            setattr(self, "lin" + str(l), lin)

    def dense(self):
        """[(W, b)] per linear, weight norm applied (differentiable)."""
        return [(dense_weight(getattr(self, "lin%d" % l)), getattr(self, "lin%d" % l).bias)
                for l in range(self.num_layers - 1)]

    # -- API-compat torch expressions (not the hot path)
    def forward(self, inputs):
        _torch_module_path(self, "SDFNetwork.forward / .gradient", inputs)
        inputs = embed(inputs * self.scale, self.multires)
        x = inputs
        for l, (W, b) in enumerate(self.dense()):
            if l in self.skip_in:
                x = torch.cat([x, inputs], 1) / math.sqrt(2)
            x = F.linear(x, W, b)
            if l < self.num_layers - 2:
                x = F.softplus(x, beta=100)
        return torch.cat([x[:, :1] / self.scale, x[:, 1:]], dim=-1)

    def sdf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        x.requires_grad_(True)
        y = self.forward(x)[:, :1]
        g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True, retain_graph=True, only_inputs=True)[0]
        return g.unsqueeze(1)

    # -- HIP path
    def sdf(self, x):
        """SDF values [N,1] at points x[N,3] through the fused gfx950 kernel (no autograd)."""
        from .engine import Engine, flatten_dense
        eng = Engine.for_networks(self, None)
        with torch.no_grad():
            pk = eng.pack(flatten_dense(self, None, eng.spec))
            return eng.sdf_pts(pk, x.reshape(-1, 3))


class RenderingNetwork(nn.Module):
    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=0,
                 squeeze_out=True, extra_color=False):
        super().__init__()
        self.mode = mode
        self.squeeze_out = squeeze_out
        self.extra_color = extra_color
        dims = [d_in + d_feature] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.multires_view = multires_view
        if multires_view > 0:
            dims[0] += 3 * 2 * multires_view
        self.num_layers = len(dims)
        self.conf = dict(d_feature=d_feature, mode=mode, d_in=d_in, d_out=d_out, d_hidden=d_hidden, n_layers=n_layers,
                         multires_view=multires_view, squeeze_out=squeeze_out, extra_color=extra_color)
        for l in range(0, self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                lin = _wn(lin)
            setattr(self, "lin" + str(l), lin)
        if self.extra_color:
            self.extra_lin = nn.Linear(dims[self.num_layers - 2], d_out)
            if weight_norm:
                self.extra_lin = _wn(self.extra_lin)

    def dense(self):
        """hidden layers [(W,b)] + the stacked 6xH head [lin_last ; extra_lin] (zeros when extra_color is off)."""
        out = [(dense_weight(getattr(self, "lin%d" % l)), getattr(self, "lin%d" % l).bias)
               for l in range(self.num_layers - 2)]
        last = getattr(self, "lin%d" % (self.num_layers - 2))
        Wl, bl = dense_weight(last), last.bias
        if self.extra_color:
            We, be = dense_weight(self.extra_lin), self.extra_lin.bias
        else:
            We, be = torch.zeros_like(Wl), torch.zeros_like(bl)
        out.append((torch.cat([Wl, We], 0), torch.cat([bl, be], 0)))
        return out

    def forward(self, points, normals, view_dirs, feature_vectors):
        _torch_module_path(self, "RenderingNetwork.forward", points)
        if self.mode == "idr":
            x = torch.cat([points, embed(view_dirs, self.multires_view), normals, feature_vectors], dim=-1)
        elif self.mode == "no_view_dir":
            x = torch.cat([points, normals, feature_vectors], dim=-1)
        elif self.mode == "no_normal":
            x = torch.cat([points, embed(view_dirs, self.multires_view), feature_vectors], dim=-1)
        else:
            raise ValueError(self.mode)
        dense = self.dense()
        for l, (W, b) in enumerate(dense[:-1]):
            x = F.relu(F.linear(x, W, b))
        W, b = dense[-1]
        x = F.linear(x, W, b)
        if not self.extra_color:
            x = x[:, :3]
        if self.squeeze_out:
            x = torch.sigmoid(x)
        return x


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(init_val))))

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)

    def inv_s(self):
        """exp(10 v).clip(1e-6, 1e6) as a 1-element tensor (fields.py:275-276 + renderer.py:234); on the GPU one launch each way
        (csrc/avc_rays.hip inv_s_kernel) instead of three forward and seven backward."""
        return self.inv_s_and_s_val()[0]

    def inv_s_and_s_val(self):
        """(inv_s [1], differentiable; s_val = 1 / inv_s [1,1], detached: the statistic of renderer.py:288 / main.py:546)"""
        if self.variance.is_cuda:
            both = _InvSFn.apply(self.variance)
            return both[:1], both.detach()[1:2].reshape(1, 1)
        inv_s = torch.exp(self.variance * 10.0).clip(1e-6, 1e6).reshape(1)
        return inv_s, 1.0 / inv_s.detach().reshape(1, 1)


class _InvSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, variance):
        from . import lib as L
        v = variance.detach().float().reshape(1).contiguous()
        out = torch.empty(2, device=v.device, dtype=torch.float32)       # (inv_s, 1 / inv_s)
        L.check(L.load().avc_inv_s(L.ptr(v), None, L.ptr(out), L.stream()), "avc_inv_s")
        ctx.save_for_backward(v)
        ctx.shape = variance.shape
        return out

    @staticmethod
    def backward(ctx, g):
        from . import lib as L
        (v,) = ctx.saved_tensors
        g = g.float()[:1].contiguous()        # (the second entry, 1 / inv_s, is a detached statistic)
        out = torch.empty(1, device=v.device, dtype=torch.float32)
        L.check(L.load().avc_inv_s(L.ptr(v), L.ptr(g), L.ptr(out), L.stream()), "avc_inv_s")
        return out.reshape(ctx.shape)
