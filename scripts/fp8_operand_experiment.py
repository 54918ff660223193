"""CPU experiment: would block-scaled FP8 operand panels pass the gradient gate?

The weight-gradient products dW = sum_points A^T B (main.py:537) read 180 operand tiles per 32-point block: the
forward-type ones in f16, the gradient-type ones in bf16 (csrc/avc_mlp.h: PanelLayout).  gfx950's v_mfma_scale_f32_32x32x64_f8f6f4
takes MX operands: 32 values along K share one E8M0 scale.  dW's K axis IS the point axis, so one scale per (feature column, 32-point
block) is exactly one scale per column of a panel tile.  This script quantises the operands of every product that way (E4M3 payload,
power-of-two scale chosen so that the block's largest magnitude lands in [256, 448]), contracts in fp64 and prints the per-tensor
relative L2 error against the exact products, next to what the current 16-bit operands give.  Nothing else changes: the sweeps that
PRODUCE the operands stay exact, so the table isolates the operand storage format.

    python scripts/fp8_operand_experiment.py            # neus_full.npz: 256 rays (16 K points) and 512 rays (32 K points)

No kernel is built from this: if E4M3 fails the 1e-2 gate on the fixtures, bf16 / f16 stored operands are the narrowest
the parity gate admits and the byte count of the backward pair is final.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import analytic as A          # noqa: E402
from oracle import neus_oracle as O       # noqa: E402
from oracle.gen_golden import scalar_loss  # noqa: E402
from tests.helpers import load_case        # noqa: E402

BETA, SQ2 = A.BETA, A.SQ2


# ------------------------------------------------------------------ operand formats
def q_f16(x):
    return x.to(torch.float16).to(torch.float64)


def q_bf16(x):
    return x.to(torch.bfloat16).to(torch.float64)


def _round_to_grid(x, mant_bits, emin, vmax):
    """round |x| to a float grid with `mant_bits` explicit mantissa bits, minimum normal exponent emin (subnormals below), saturating"""
    ax = x.abs().clamp(max=vmax)
    e = torch.floor(torch.log2(ax.clamp(min=2.0 ** (emin - mant_bits))))
    e = e.clamp(min=emin)
    step = 2.0 ** (e - mant_bits)
    return torch.sign(x) * torch.round(ax / step) * step


def q_mx(x, fmt="e4m3", block=32):
    """MX block format along the POINT axis (dim 0): one E8M0 scale per (32 points, feature), payload E4M3 / E5M2 / E2M3 (fp6)."""
    mant, emin, vmax = {"e4m3": (3, -6, 448.0), "e5m2": (2, -14, 57344.0), "e2m3": (3, 0, 7.5)}[fmt]
    n, f = x.shape
    pad = (-n) % block
    xp = torch.cat([x, x.new_zeros(pad, f)], 0).reshape(-1, block, f)
    amax = xp.abs().amax(1, keepdim=True)
    # OCP MX: scale = 2^(floor(log2 amax) - emax_of_format)
    emax = int(torch.floor(torch.log2(torch.tensor(vmax))).item())
    sc = 2.0 ** (torch.floor(torch.log2(amax.clamp(min=1e-300))) - emax)
    q = _round_to_grid(xp / sc, mant, emin, vmax) * sc
    return q.reshape(-1, f)[:n]


# ------------------------------------------------------------------ the products, with an operand hook
def products(net, f, d_sdf, d_n, d_rgb6, qF, qG):
    """mlp_backward of oracle/analytic.py with the OPERANDS of every weight-gradient product passed through qF (forward-type:
    h, g_a, u, r) / qG (gradient-type: gbar_h, abar, ybar, delta).  The sweeps themselves are exact."""
    W, CW = net["sdf_W"], net["col_W"]
    L = len(W)
    x, h, sig = f["x"], f["h"], f["sig"]
    nh = h[-1].shape[1]
    out = {}
    delta = d_rgb6 * f["rgb6"] * (1 - f["rgb6"])
    out["col.head"] = qG(delta).t() @ qF(f["r"][-1])
    d_r = delta @ CW[-1]
    for l in range(len(CW) - 2, -1, -1):
        delta = d_r * (f["c"][l + 1] > 0).to(d_r.dtype)
        out["col.lin%d" % l] = qG(delta).t() @ qF(f["r"][l])
        d_r = delta @ CW[l]
    nbar = d_n + d_r[:, 3:6]
    ybar = torch.cat([d_sdf, d_r[:, 6:]], 1)
    dW = [torch.zeros_like(w) for w in W]
    gb_h = A.pe_jac(x, nbar)
    gb_h0 = gb_h
    abar = [None] * L
    for l in range(1, L):
        gb_a = gb_h @ W[l - 1].t()
        dW[l - 1] += qF(f["g_a"][l]).t() @ qG(gb_h)
        abar[l] = gb_a * f["g_h"][l] * (BETA * sig[l] * (1 - sig[l]))
        gb_h = gb_a * sig[l]
    gb_u = torch.cat([gb_h, gb_h0], 1) / SQ2
    dW[L - 1][0] += qG(gb_u).sum(0)
    ubar = ybar @ W[L - 1]
    dW[L - 1] += qG(ybar).t() @ qF(f["u"])
    hbar = ubar[:, :nh] / SQ2
    for l in range(L - 1, 0, -1):
        abar[l] = abar[l] + hbar * sig[l]
        dW[l - 1] += qG(abar[l]).t() @ qF(h[l - 1])
        hbar = abar[l] @ W[l - 1]
    for l in range(L):
        out["sdf.lin%d" % l] = dW[l]
    return out


def chain(nrays):
    rec, sd_sdf, sd_col, variance = load_case("neus_full.npz")
    dt = torch.float64
    net = A.dense_net(sd_sdf, sd_col, dt)
    ro, rd = rec["rays_o"][:nrays].to(dt), rec["rays_d"][:nrays].to(dt)
    z = rec["z_final"][:nrays].to(dt)
    bg = rec["bg"].to(dt) if rec["bg"].numel() else None
    cosr = float(rec["cos_anneal"])
    coef = {k[5:]: v[:nrays].to(dt) for k, v in rec.items() if k.startswith("coef_")}
    with torch.no_grad():
        R, S = z.shape
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 2.0 / 32)], -1)
        x = (ro[:, None, :] + rd[:, None, :] * (z + dists * 0.5)[..., None]).reshape(-1, 3)
        f = A.mlp_forward(net, x)
        inv_s = O.inv_s_from_variance(variance.to(dt))
        cf = A.composite_forward(f["sdf"].reshape(R, S), f["n"].reshape(R, S, 3), f["rgb6"].reshape(R, S, 6), z, rd,
                                 x.norm(dim=-1).reshape(R, S), inv_s, 2.0 / 32, cosr, bg)
    o2 = {k: cf[k].detach().clone().requires_grad_(True) for k in ("color", "extra", "w", "eik")}
    n2 = f["n"].reshape(R, S, 3).detach().clone().requires_grad_(True)
    fake = dict(color_fine=o2["color"], extra_color_fine=o2["extra"], weights=o2["w"], gradients=n2, gradient_error=o2["eik"],
                weight_sum=o2["w"].sum(-1, keepdim=True))
    dcol, dext, dw, deik, dn_up = torch.autograd.grad(scalar_loss(fake, coef), [o2["color"], o2["extra"], o2["w"], o2["eik"], n2])
    with torch.no_grad():
        cb = A.composite_backward(cf, f["sdf"].reshape(R, S), f["n"].reshape(R, S, 3), f["rgb6"].reshape(R, S, 6), rd, inv_s, cosr,
                                  bg, dcol, dext, dw, dn_up, deik)
    return net, f, cb["d_sdf"].reshape(-1, 1), cb["d_n"].reshape(-1, 3), cb["d_rgb6"].reshape(-1, 6)


def main():
    ident = lambda t: t
    variants = [
        ("f16 / bf16 (current panels)", q_f16, q_bf16),
        ("G-type MX-E4M3, F-type f16", q_f16, lambda t: q_mx(t, "e4m3")),
        ("both MX-E4M3", lambda t: q_mx(t, "e4m3"), lambda t: q_mx(t, "e4m3")),
        ("both MX-E5M2", lambda t: q_mx(t, "e5m2"), lambda t: q_mx(t, "e5m2")),
    ]
    for nrays in (256, 512):
        net, f, d_sdf, d_n, d_rgb6 = chain(nrays)
        with torch.no_grad():
            exact = products(net, f, d_sdf, d_n, d_rgb6, ident, ident)
            print("\n## neus_full.npz, %d rays = %d points; relative L2 error of every dense dW against the exact products (gate: 1e-2)" %
                  (nrays, f["x"].shape[0]))
            names = sorted(exact)
            print("| operands | " + " | ".join(names) + " | worst |")
            print("|---|" + "---|" * (len(names) + 1))
            for label, qF, qG in variants:
                got = products(net, f, d_sdf, d_n, d_rgb6, qF, qG)
                errs = [((got[k] - exact[k]).norm() / exact[k].norm()).item() for k in names]
                print("| %s | " % label + " | ".join("%.2e" % e for e in errs) + " | **%.2e** |" % max(errs))


if __name__ == "__main__":
    main()
