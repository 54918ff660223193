import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_kernels import _setup, relerr
from oracle.gen_golden import scalar_loss
for name in ("neus_small.npz", "neus_full.npz"):
    rec, sd_sdf, sd_col, variance, sdf, col, var, ren, dev = _setup(name)
    bg = rec["bg"].to(dev) if rec["bg"].numel() else None
    out = ren.render(rec["rays_o"].to(dev), rec["rays_d"].to(dev), rec["near"].to(dev), rec["far"].to(dev),
                     background_rgb=bg, cos_anneal_ratio=float(rec["cos_anneal"]), z_vals=rec["z_final"].to(dev))
    coef = {k[5:]: v.to(dev) for k, v in rec.items() if k.startswith("coef_")}
    loss = scalar_loss(out, coef); loss.backward(); torch.cuda.synchronize()
    for pfx, net in (("sdf.", sdf), ("col.", col)):
        for n_, p in net.named_parameters():
            ref = rec["grad_" + pfx + n_]
            g = p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(ref)
            print(name, "%-22s rel %.3e |g| %.3e |ref| %.3e  nan %d" % (pfx + n_, relerr(g, ref), g.norm().item(), ref.norm().item(), int(torch.isnan(g).sum())))
    eng = ren.engine
    lay = eng.dl.lay
    P = lay.panel
    R, S = rec["z_final"].shape
    nblk = R * S // 32
    pan = eng._panels[: nblk * eng.ptiles * 2048].view(torch.int16).reshape(nblk, eng.ptiles, 2, 64, 8)
    for nm in ("D1", "DO", "FEAT", "XN", "R1", "DFEAT", "AB1"):
        t = pan[:, P[nm]]
        f = t.view(torch.bfloat16 if nm in ("D1", "DO", "DFEAT", "AB1") else torch.float16).float()
        print("  panel", nm, "tile", P[nm], "abs max %.3e mean %.3e nan %d" % (f.abs().max().item(), f.abs().mean().item(), int(torch.isnan(f).sum())))
    # mask check: bit r of masks[blk][layer][tile][lane] <-> r panel (tile, e = r >> 3, lane, j = r & 7) > 0
    HT = lay.spec.HT
    mk = eng._masks[: nblk * eng.mask_u16].view(torch.int16).reshape(nblk, 2, HT, 64).int() & 0xffff
    for layer, nm in ((0, "R1"), (1, "R2")):
        if nm == "R2" and lay.spec.NCMID == 0:
            continue
        r = pan[:, P[nm]:P[nm] + HT].view(torch.float16).float()      # [nblk, HT, 2, 64, 8]
        bits = (r > 0).int()
        ref = torch.zeros(nblk, HT, 64, dtype=torch.int32, device=r.device)
        for e in range(2):
            for j in range(8):
                ref |= bits[:, :, e, :, j] << (8 * e + j)
        diff = (ref != mk[:, layer])
        print("  masks", nm, "mismatching (blk,tile,lane) entries:", int(diff.sum()), "of", diff.numel(), " popcount ref", int(bits.sum()))
    if lay.spec.NCMID == 1:
        from avatarclip_amd.packing import frag_feature
        H = lay.spec.H
        dense = [(W.detach().float(), b.detach().float()) for W, b in col.dense()]
        CH = dense[-1][0]                                       # [6,H]
        do = pan[:, P["DO"]].view(torch.bfloat16).float()       # [nblk, 2, 64, 8]
        delta = torch.zeros(nblk, 32, 6, device=do.device)
        for hh in range(2):
            for j in range(4):
                ch = 4 * hh + j
                if ch < 6:
                    delta[:, :, ch] = do[:, 0, 32 * hh:32 * hh + 32, j]
        d_r2 = delta.reshape(-1, 6) @ CH.to(do.device)                       # [N, H]
        r2 = pan[:, P["R2"]:P["R2"] + HT].view(torch.float16).float()   # [nblk, HT, 2, 64, 8]
        d2 = pan[:, P["D2"]:P["D2"] + HT].view(torch.bfloat16).float()
        # frag (tile t, e, lane (p,h), j) -> feature 32 t + 16 e + 8 (j>>2) + 4 h + (j&3)
        ref = torch.zeros_like(d2)
        for t in range(HT):
            for e in range(2):
                for hh in range(2):
                    for j in range(8):
                        f = 32 * t + 16 * e + 8 * (j >> 2) + 4 * hh + (j & 3)
                        ref[:, t, e, 32 * hh:32 * hh + 32, j] = d_r2.reshape(nblk, 32, H)[:, :, f]
        ref = ref * (r2 > 0)
        print("  D2 vs fp32 reference: rel", ((d2 - ref).norm() / ref.norm()).item(), " sum(D2) err", ((d2.sum((0, 2, 3)) - ref.sum((0, 2, 3))).norm() / ref.sum((0, 2, 3)).norm()).item())
        print("  col.lin1.bias engine vs sum(D2 panel):", col.lin1.bias.grad.norm().item(), d2.sum((0, 2, 3, 4)).shape)
