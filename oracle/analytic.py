"""Explicit (autograd-free) per-stage formulas of the fused kernels -- the mathematics of SURVEY.md
Appendix A.1-A.4 written out stage by stage, in torch on CPU (fp64 by default).

TEST INFRASTRUCTURE ONLY.  tests/test_analytic.py proves these formulas equal torch.autograd applied to
oracle/neus_oracle.py (which is itself pinned against the reference); the GPU parity tests then use the
per-stage intermediates returned here to localise a kernel bug to one stage.

Dense weights: `net` is a dict
   sdf_W[l], sdf_b[l]   l = 0..L-1   (weight-normed, fields.py:65-66 already applied)
   col_W[l], col_b[l]   hidden colour layers; col_W[-1] is the stacked 6xH head [lin_last ; extra_lin]
"""
import math

import torch

BETA = 100.0
SQ2 = math.sqrt(2.0)


def dense_net(sd_sdf, sd_col, dtype=torch.float64):
    from oracle.neus_oracle import wn_weight, n_linears
    nl = n_linears(sd_sdf)
    nc = n_linears(sd_col)
    net = dict(
        sdf_W=[wn_weight(sd_sdf, "lin%d" % l).to(dtype) for l in range(nl)],
        sdf_b=[sd_sdf["lin%d.bias" % l].to(dtype) for l in range(nl)],
        col_W=[wn_weight(sd_col, "lin%d" % l).to(dtype) for l in range(nc - 1)] +
              [torch.cat([wn_weight(sd_col, "lin%d" % (nc - 1)), wn_weight(sd_col, "extra_lin")], 0).to(dtype)],
        col_b=[sd_col["lin%d.bias" % l].to(dtype) for l in range(nc - 1)] +
              [torch.cat([sd_col["lin%d.bias" % (nc - 1)], sd_col["extra_lin.bias"]], 0).to(dtype)],
    )
    return net


def pe(x, L=6):
    outs = [x]
    for k in range(L):
        outs += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(outs, -1)


def pe_jac_T(x, g, L=6):
    """n_c = g[c] + sum_k f_k (cos(f_k x_c) g[sin_k,c] - sin(f_k x_c) g[cos_k,c])  (J_PE^T g)."""
    n = g[:, 0:3].clone()
    for k in range(L):
        f = 2.0 ** k
        n = n + f * (torch.cos(f * x) * g[:, 3 + 6 * k: 6 + 6 * k] - torch.sin(f * x) * g[:, 6 + 6 * k: 9 + 6 * k])
    return n


def pe_jac(x, nbar, L=6):
    """J_PE nbar: [nbar_c | f cos(f x_c) nbar_c | -f sin(f x_c) nbar_c]."""
    outs = [nbar]
    for k in range(L):
        f = 2.0 ** k
        outs += [f * torch.cos(f * x) * nbar, -f * torch.sin(f * x) * nbar]
    return torch.cat(outs, -1)


def softplus(a):
    return torch.nn.functional.softplus(a, beta=BETA)


def mlp_forward(net, x):
    """SDF forward + normal sweep + colour MLP (kernel `render_fwd`).  Returns all intermediates."""
    W, b = net["sdf_W"], net["sdf_b"]
    L = len(W)
    h0 = pe(x)
    h = [h0]
    sig = [None]
    for l in range(L - 1):
        a = h[-1] @ W[l].t() + b[l]
        h.append(softplus(a))
        sig.append(torch.sigmoid(BETA * a))
    u = torch.cat([h[-1], h0], 1) / SQ2
    y = u @ W[L - 1].t() + b[L - 1]
    sdf, feat = y[:, :1], y[:, 1:]
    # normal sweep (reverse)
    nh = h[-1].shape[1]
    g_u = W[L - 1][0:1, :].expand(x.shape[0], -1)
    g_h = [None] * L  # g_h[l] gradient wrt h_l
    g_a = [None] * L  # g_a[l] gradient wrt a_l (pre-activation of h_l)
    g_h[L - 1] = g_u[:, :nh] / SQ2
    for l in range(L - 1, 0, -1):
        g_a[l] = g_h[l] * sig[l]
        g_h[l - 1] = g_a[l] @ W[l - 1]
    g_h0 = g_h[0] + g_u[:, nh:] / SQ2
    n = pe_jac_T(x, g_h0)
    # colour
    CW, cb = net["col_W"], net["col_b"]
    r = [torch.cat([x, n, feat], 1)]
    c = [None]
    for l in range(len(CW) - 1):
        cl = r[-1] @ CW[l].t() + cb[l]
        c.append(cl)
        r.append(torch.relu(cl))
    o = r[-1] @ CW[-1].t() + cb[-1]
    rgb6 = torch.sigmoid(o)
    return dict(x=x, h=h, sig=sig, u=u, y=y, sdf=sdf, feat=feat, g_h=g_h, g_a=g_a, g_h0=g_h0, n=n, r=r, c=c,
                rgb6=rgb6)


def mlp_backward(net, f, d_sdf, d_n, d_rgb6):
    """Backward of mlp_forward wrt all dense weights (kernel `render_bwd` + `dw_gemm`).
    d_sdf [N,1], d_n [N,3], d_rgb6 [N,6].  x carries no gradient (z-values are detached, renderer.py:176,336)."""
    W, b = net["sdf_W"], net["sdf_b"]
    CW = net["col_W"]
    L = len(W)
    x, h, sig = f["x"], f["h"], f["sig"]
    nh = h[-1].shape[1]
    dCW = [None] * len(CW)
    dcb = [None] * len(CW)
    # colour backward
    delta = d_rgb6 * f["rgb6"] * (1 - f["rgb6"])
    dCW[-1] = delta.t() @ f["r"][-1]
    dcb[-1] = delta.sum(0)
    d_r = delta @ CW[-1]
    for l in range(len(CW) - 2, -1, -1):
        delta = d_r * (f["c"][l + 1] > 0).to(d_r.dtype)
        dCW[l] = delta.t() @ f["r"][l]
        dcb[l] = delta.sum(0)
        d_r = delta @ CW[l]
    nbar = d_n + d_r[:, 3:6]
    ybar = torch.cat([d_sdf, d_r[:, 6:]], 1)
    dW = [torch.zeros_like(w) for w in W]
    db = [torch.zeros_like(v) for v in b]
    # (i) second-order sweep, forward layer order
    gb_h = pe_jac(x, nbar)  # gbar_h0
    gb_h0 = gb_h
    abar = [None] * L
    for l in range(1, L):
        gb_a = gb_h @ W[l - 1].t()
        dW[l - 1] += f["g_a"][l].t() @ gb_h
        sp2 = BETA * sig[l] * (1 - sig[l])
        abar[l] = gb_a * f["g_h"][l] * sp2
        gb_h = gb_a * sig[l]
    gb_u = torch.cat([gb_h, gb_h0], 1) / SQ2
    dW[L - 1][0] += gb_u.sum(0)
    # (ii) ordinary reverse sweep
    ubar = ybar @ W[L - 1]
    dW[L - 1] += ybar.t() @ f["u"]
    db[L - 1] += ybar.sum(0)
    hbar = ubar[:, :nh] / SQ2
    for l in range(L - 1, 0, -1):
        abar[l] = abar[l] + hbar * sig[l]
        dW[l - 1] += abar[l].t() @ h[l - 1]
        db[l - 1] += abar[l].sum(0)
        hbar = abar[l] @ W[l - 1]
    return dict(sdf_dW=dW, sdf_db=db, col_dW=dCW, col_db=dcb, nbar=nbar, ybar=ybar, abar=abar)


def composite_forward(sdf, n, rgb6, z_vals, rays_d, pts_norm, inv_s, sample_dist, cos_anneal, bg=None):
    """renderer.py:207-286 restated per ray (kernel `composite_fwd`).  sdf [R,S], n [R,S,3], rgb6 [R,S,6]."""
    R, S = z_vals.shape
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], sample_dist)], -1)
    c = (rays_d[:, None, :] * n).sum(-1)
    ic = -(torch.relu(-c * 0.5 + 0.5) * (1 - cos_anneal) + torch.relu(-c) * cos_anneal)
    e_next = sdf + ic * dists * 0.5
    e_prev = sdf - ic * dists * 0.5
    P = torch.sigmoid(e_prev * inv_s)
    Q = torch.sigmoid(e_next * inv_s)
    a_raw = (P - Q + 1e-5) / (P + 1e-5)
    alpha = a_raw.clip(0, 1)
    t = 1 - alpha + 1e-7
    T = torch.cumprod(torch.cat([torch.ones_like(t[:, :1]), t], -1), -1)[:, :-1]
    w = alpha * T
    Wsum = w.sum(-1, keepdim=True)
    color = (w[..., None] * rgb6[..., :3]).sum(1)
    extra = (w[..., None] * rgb6[..., 3:]).sum(1)
    if bg is not None:
        extra = extra + bg * (1 - Wsum)
    relax = (pts_norm < 1.2).to(z_vals.dtype)
    gn = torch.linalg.norm(n, dim=-1)
    eik_num = (relax * (gn - 1) ** 2).sum()
    eik_den = relax.sum() + 1e-5
    return dict(dists=dists, c=c, ic=ic, e_next=e_next, e_prev=e_prev, P=P, Q=Q, a_raw=a_raw, alpha=alpha, T=T,
                w=w, color=color, extra=extra, relax=relax, gn=gn, eik=eik_num / eik_den, eik_den=eik_den)


def composite_backward(cf, sdf, n, rgb6, rays_d, inv_s, cos_anneal, bg, d_color, d_extra, d_w, d_n_up, d_eik):
    """Reverse of composite_forward (kernel `composite_bwd`): returns d_sdf, d_n, d_rgb6, d_inv_s."""
    w, T, alpha, P, Q = cf["w"], cf["T"], cf["alpha"], cf["P"], cf["Q"]
    R, S = w.shape
    bgv = bg if bg is not None else torch.zeros(1, 3, dtype=w.dtype)
    wbar = (d_color[:, None, :] * rgb6[..., :3]).sum(-1) + (d_extra[:, None, :] * (rgb6[..., 3:] - bgv[:, None, :])).sum(-1) + d_w
    d_rgb6 = torch.cat([w[..., None] * d_color[:, None, :], w[..., None] * d_extra[:, None, :]], -1)
    t = 1 - alpha + 1e-7
    # S_{i-1} = wbar_i alpha_i + t_i S_i ; abar_i = T_i (wbar_i - S_i)
    Ssuf = torch.zeros_like(w)
    acc = torch.zeros(R, dtype=w.dtype)
    for i in range(S - 1, -1, -1):
        Ssuf[:, i] = acc
        acc = wbar[:, i] * alpha[:, i] + t[:, i] * acc
    abar = T * (wbar - Ssuf)
    a_raw = cf["a_raw"]
    d_araw = abar * ((a_raw >= 0) & (a_raw <= 1)).to(w.dtype)
    dP = d_araw * Q / (P + 1e-5) ** 2
    dQ = -d_araw / (P + 1e-5)
    de_prev = dP * P * (1 - P) * inv_s
    de_next = dQ * Q * (1 - Q) * inv_s
    d_inv_s = (dP * P * (1 - P) * cf["e_prev"] + dQ * Q * (1 - Q) * cf["e_next"]).sum()
    d_sdf = de_prev + de_next
    d_ic = (de_next - de_prev) * cf["dists"] * 0.5
    c = cf["c"]
    d_c = d_ic * (0.5 * (1 - cos_anneal) * ((-0.5 * c + 0.5) > 0).to(w.dtype) + cos_anneal * ((-c) > 0).to(w.dtype))
    d_n = d_c[..., None] * rays_d[:, None, :] + d_n_up
    gn = cf["gn"]
    d_n = d_n + (d_eik / cf["eik_den"]) * (cf["relax"] * 2 * (gn - 1) / gn)[..., None] * n
    return dict(d_sdf=d_sdf, d_n=d_n, d_rgb6=d_rgb6, d_inv_s=d_inv_s, wbar=wbar, abar=abar)
