"""oracle/analytic.py (explicit per-stage forward/backward formulas that the HIP kernels implement) must equal
torch.autograd applied to oracle/neus_oracle.py.  fp64, CPU."""
import pytest
import torch

from oracle import analytic as A
from oracle import neus_oracle as O
from oracle.gen_golden import scalar_loss
from tests.helpers import load_case, relerr


def dense_sd(net):
    sd_s = {}
    for l, (w, b) in enumerate(zip(net["sdf_W"], net["sdf_b"])):
        sd_s["lin%d.weight" % l] = w
        sd_s["lin%d.bias" % l] = b
    sd_c = {}
    nc = len(net["col_W"])
    for l in range(nc - 1):
        sd_c["lin%d.weight" % l] = net["col_W"][l]
        sd_c["lin%d.bias" % l] = net["col_b"][l]
    sd_c["lin%d.weight" % (nc - 1)] = net["col_W"][-1][:3]
    sd_c["lin%d.bias" % (nc - 1)] = net["col_b"][-1][:3]
    sd_c["extra_lin.weight"] = net["col_W"][-1][3:]
    sd_c["extra_lin.bias"] = net["col_b"][-1][3:]
    return sd_s, sd_c


@pytest.mark.parametrize("name,nrays", [("neus_small.npz", 48), ("neus_full.npz", 24)])
def test_full_chain_matches_autograd(name, nrays):
    rec, sd_sdf, sd_col, variance = load_case(name)
    dt = torch.float64
    net = A.dense_net(sd_sdf, sd_col, dt)
    leaves = [t.requires_grad_(True) for k in ("sdf_W", "sdf_b", "col_W", "col_b") for t in net[k]]
    ro, rd = rec["rays_o"][:nrays].to(dt), rec["rays_d"][:nrays].to(dt)
    z = rec["z_final"][:nrays].to(dt)
    bg = rec["bg"].to(dt) if rec["bg"].numel() else None
    cosr = float(rec["cos_anneal"])
    var = variance.to(dt).clone().requires_grad_(True)
    coef = {k[5:]: v[:nrays].to(dt) for k, v in rec.items() if k.startswith("coef_")}
    # ---- autograd on the oracle
    sd_s, sd_c = dense_sd(net)
    out = O.render(sd_s, sd_c, var, ro, rd, None, None, z_vals=z, background_rgb=bg, cos_anneal_ratio=cosr)
    loss = scalar_loss(out, coef)
    gref = torch.autograd.grad(loss, leaves + [var])
    # ---- explicit chain
    with torch.no_grad():
        R, S = z.shape
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 2.0 / 32)], -1)
        mid = z + dists * 0.5
        x = (ro[:, None, :] + rd[:, None, :] * mid[..., None]).reshape(-1, 3)
        f = A.mlp_forward(net, x)
        assert torch.allclose(f["sdf"], out["sdf"].detach(), atol=1e-10)
        assert torch.allclose(f["n"].reshape(R, S, 3), out["gradients"].detach(), atol=1e-9)
        inv_s = O.inv_s_from_variance(var.detach())
        cf = A.composite_forward(f["sdf"].reshape(R, S), f["n"].reshape(R, S, 3), f["rgb6"].reshape(R, S, 6), z, rd,
                                 x.norm(dim=-1).reshape(R, S), inv_s, 2.0 / 32, cosr, bg)
        assert torch.allclose(cf["color"], out["color_fine"].detach(), atol=1e-10)
        assert torch.allclose(cf["extra"], out["extra_color_fine"].detach(), atol=1e-10)
        assert torch.allclose(cf["w"], out["weights"].detach(), atol=1e-10)
        assert abs(cf["eik"].item() - out["gradient_error"].item()) < 1e-10
    # upstream grads of the scalar loss wrt the kernel outputs (this part stays in torch in the product too)
    o2 = {k: cf[k].detach().clone().requires_grad_(True) for k in ("color", "extra", "w", "eik")}
    n2 = f["n"].reshape(R, S, 3).detach().clone().requires_grad_(True)
    fake = dict(color_fine=o2["color"], extra_color_fine=o2["extra"], weights=o2["w"], gradients=n2,
                gradient_error=o2["eik"], weight_sum=o2["w"].sum(-1, keepdim=True))
    l2 = scalar_loss(fake, coef)
    assert abs(l2.item() - loss.item()) < 1e-8
    dcol, dext, dw, deik, dn_up = torch.autograd.grad(l2, [o2["color"], o2["extra"], o2["w"], o2["eik"], n2])
    with torch.no_grad():
        cb = A.composite_backward(cf, f["sdf"].reshape(R, S), f["n"].reshape(R, S, 3), f["rgb6"].reshape(R, S, 6), rd,
                                  inv_s, cosr, bg, dcol, dext, dw, dn_up, deik)
        mb = A.mlp_backward(net, f, cb["d_sdf"].reshape(-1, 1), cb["d_n"].reshape(-1, 3), cb["d_rgb6"].reshape(-1, 6))
        d_var = cb["d_inv_s"] * 10.0 * inv_s
    mine = mb["sdf_dW"] + mb["sdf_db"] + mb["col_dW"] + mb["col_db"] + [d_var]
    for i, (a, b) in enumerate(zip(mine, gref)):
        assert relerr(a, b) < 1e-7, (i, relerr(a, b))
