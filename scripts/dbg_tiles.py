"""development aid: the forward-type operand tiles of avc_render_points_fwd_train decoded from the panels and compared layer by
layer with oracle/analytic.py (python scripts/dbg_tiles.py [neus_small.npz|neus_full.npz])"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_kernels import _setup
from oracle import analytic as A
from avatarclip_amd.packing import frag_feature
names = sys.argv[1:] or ["neus_small.npz", "neus_full.npz"]
for name in names:
    rec, sd_sdf, sd_col, variance, sdf, col, var, ren, dev = _setup(name)
    eng = ren.engine
    pk = eng.pack(ren.flat_params())
    ro, rd, z = rec["rays_o"].to(dev), rec["rays_d"].to(dev), rec["z_final"].to(dev).contiguous()
    sd, nr, rgb = eng.points_fwd_train(pk, ro, rd, z, 2.0 / 32)
    torch.cuda.synchronize()
    R, S = z.shape
    zc = z.cpu().double()
    dists = torch.cat([zc[:, 1:] - zc[:, :-1], torch.full((R, 1), 2.0 / 32, dtype=torch.float64)], -1)
    mid = zc + dists * 0.5
    x = (rec["rays_o"].double()[:, None, :] + rec["rays_d"].double()[:, None, :] * mid[..., None]).reshape(-1, 3)
    f = A.mlp_forward(A.dense_net(sd_sdf, sd_col, torch.float64), x)
    npts = x.shape[0]; nblk = (npts + 31) // 32
    P = eng.dl.lay.panel
    panels = eng._fpanels[: nblk * eng.fwd_tiles * 2048].view(torch.int16).reshape(nblk, eng.fwd_tiles, 2, 64, 8).cpu()   # F region: P[...] of the forward-type tiles = region-local index
    def decode(p0, ntile, nfeat):
        t = panels[:, p0:p0 + ntile].view(torch.float16).double()     # [blk, tile, kstep, lane, 8]
        out = torch.zeros(nblk * 32, ntile * 32, dtype=torch.float64)
        for tt in range(ntile):
            for e in range(2):
                for lane in range(64):
                    hh, p = lane >> 5, lane & 31
                    for j in range(8):
                        feat = 32 * tt + (frag_feature(e, hh, j) % 32)
                        out[p::32, feat] = t[:, tt, e, lane, j]
        return out[:npts, :nfeat]
    H = f["h"][1].shape[1]
    print(name, "rgb err", (rgb.cpu().reshape(-1, 6).double() - f["rgb6"]).abs().max().item())
    print("  h1   err", (decode(P["H1"], H // 32, H) - f["h"][1] * 100 * np.log2(np.e)).abs().max().item(), "scale", (f["h"][1] * 144.27).abs().max().item())
    print("  feat err", (decode(P["FEAT"], H // 32, H) - f["feat"]).abs().max().item(), "scale", f["feat"].abs().max().item())
    for l in range(1, len(f["r"])):
        print("  r%d   err" % l, (decode(P["R%d" % l], H // 32, H) - f["r"][l]).abs().max().item(), "scale", f["r"][l].abs().max().item())
    xn = decode(P["XN"], 1, 6)
    xn_ref = f["r"][0][:, :6]
    print("  xn   err", (xn - xn_ref).abs().max().item(), "per column", (xn - xn_ref).abs().max(0).values.tolist())
    r1 = decode(P["R1"], H // 32, H)
    e = (r1 - f["r"][1]).abs()
    print("  r1 err: mean %.3e, fraction of entries off by > 0.05: %.4f, rows hit %d of %d, columns hit %d of %d"
          % (e.mean().item(), (e > 0.05).double().mean().item(), int((e > 0.05).any(1).sum()), e.shape[0], int((e > 0.05).any(0).sum()), e.shape[1]))
    bad_cols = torch.nonzero((e > 0.05).any(0)).reshape(-1).tolist()
    print("  bad columns", bad_cols[:40])
