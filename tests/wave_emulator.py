"""Numpy emulation of ONE wavefront of the HIP MLP kernels at MFMA-fragment level (fp32, no 16-bit rounding).

It mirrors csrc/avc_mlp.h / avc_mlp_bwd.hip statement by statement -- same fragment arrays, same packed blobs,
same table indexing -- with the documented gfx950 operand/accumulator layouts of v_mfma_f32_32x32x16.  The CPU
test-suite uses it to prove that packing.py + the kernels' index arithmetic reproduce oracle/analytic.py before
any GPU time is spent; the hardware layouts themselves are verified by tests/test_gpu_probe.py.
TEST INFRASTRUCTURE ONLY.
"""
import numpy as np

from avatarclip_amd import packing as PK

BETA = 100.0
LANES = np.arange(64)
LH = LANES >> 5
LP = LANES & 31


_ROW = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for r in range(16)] for h in range(2)])  # [h][r]
_ROWL = _ROW[LH]            # [64,16]
_COLL = LP[:, None]         # [64,1]


def mfma(a, b, c=None):
    """a,b: [64,8] operand fragments, c: [64,16] accumulator.  D = A(32x16) @ B(16x32) + C with
    A[i, 8h+j] = a[32h+i, j], B[8h+j, n] = b[32h+n, j], C/D lane (n,h) reg r <-> D[(r&3)+8(r>>2)+4h, n]."""
    A = a.reshape(2, 32, 8).transpose(1, 0, 2).reshape(32, 16)
    B = b.reshape(2, 32, 8).transpose(0, 2, 1).reshape(16, 32)
    D = A @ B
    out = D[_ROWL, _COLL]
    return out if c is None else c + out


class Blob:
    def __init__(self, lay: PK.Layout, flatP: np.ndarray):
        pz = np.concatenate([flatP.astype(np.float64), [0.0]])
        self.w = pz[lay.idx16] * lay.scale16
        self.tab = pz[lay.idx32] * lay.scale32
        self.off = {k: int(lay.offsets[v]) for k, v in PK.OFF.items() if k != "OFF_COUNT"}
        self.lay = lay

    def wfrag(self, off_name, KS, t, s):
        base = self.off[off_name] + ((t * KS + s) * 64) * 8
        return self.w[base:base + 512].reshape(64, 8)

    def load16(self, off_name, t):
        base = self.off[off_name]
        out = np.zeros((64, 16))
        for l in range(64):
            out[l] = self.tab[base + (t * 2 + (l >> 5)) * 16: base + (t * 2 + (l >> 5)) * 16 + 16]
        return out

    def load8(self, off_name, s):
        base = self.off[off_name]
        out = np.zeros((64, 8))
        for l in range(64):
            out[l] = self.tab[base + (s * 2 + (l >> 5)) * 8: base + (s * 2 + (l >> 5)) * 8 + 8]
        return out


def tile_gemm(blob, off_name, KS, t, frags):
    acc = np.zeros((64, 16))
    for s in range(KS):
        acc = mfma(blob.wfrag(off_name, KS, t, s), frags[s], acc)
    return acc


def softplus(t):
    """softplus2 of avc_common.h: base-2 units, t = S a, returns H = S h."""
    return np.maximum(t, 0) + np.log2(1.0 + np.exp2(-np.abs(t)))


def sig_from_h(H):
    return 1.0 - np.exp2(-H)


def pe_compute(x):
    """x [32,3] -> v,d [64,24] per lane (mirrors pe_compute in avc_common.h)."""
    v = np.zeros((64, 24))
    d = np.zeros((64, 24))
    for l in range(64):
        h, p = l >> 5, l & 31
        f0 = 8.0 if h else 1.0
        for q in range(24):
            if h == 0:
                if q < 3:
                    v[l, q], d[l, q] = x[p, q], 1.0
                elif q < 21:
                    k, rem = divmod(q - 3, 6)
                    fr = f0 * (1 << k)
                    c = rem % 3
                    if rem < 3:
                        v[l, q], d[l, q] = np.sin(x[p, c] * fr), fr * np.cos(x[p, c] * fr)
                    else:
                        v[l, q], d[l, q] = np.cos(x[p, c] * fr), -fr * np.sin(x[p, c] * fr)
            else:
                if q < 18:
                    k, rem = divmod(q, 6)
                    fr = f0 * (1 << k)
                    c = rem % 3
                    if rem < 3:
                        v[l, q], d[l, q] = np.sin(x[p, c] * fr), fr * np.cos(x[p, c] * fr)
                    else:
                        v[l, q], d[l, q] = np.cos(x[p, c] * fr), -fr * np.sin(x[p, c] * fr)
    return v, d


def acc_to_frags(a):
    return a[:, :8].copy(), a[:, 8:].copy()


def xhalf_sum(v):
    return v + v[(LANES + 32) % 64]


def dense_layer(blob, offw, offb, KS, NT, frags, act):
    out = []
    for t in range(NT):
        acc = tile_gemm(blob, offw, KS, t, frags) + blob.load16(offb, t)
        a = softplus(acc) if act == 1 else (np.maximum(acc, 0) if act == 2 else acc)
        f0, f1 = acc_to_frags(a)
        out += [f0, f1]
    return out


def forward_wave(spec: PK.NetSpec, blob: Blob, x):
    """mirrors sdf_trunk + sdf_feature + sdf_normal + color_forward.  x [32,3].  Returns per-point sdf[32], n[32,3],
    rgb[32,6] plus the internal fragment arrays."""
    HT, HK, ST, SK = spec.HT, spec.HK, spec.ST, spec.SK
    pev, ped = pe_compute(x)
    pef = [pev[:, 0:8].copy(), pev[:, 8:16].copy(), pev[:, 16:24].copy()]  # hi/lo split is exact in fp64: lo = 0
    st = {}
    h1 = dense_layer(blob, "OFF_W0", "OFF_B0", 3, HT, pef, 1)
    hm = [dense_layer(blob, "OFF_WM0", "OFF_BM0", HK, HT, h1, 1)]
    if spec.NMID == 2:
        hm.append(dense_layer(blob, "OFF_WM1", "OFF_BM1", HK, HT, hm[0], 1))
    hs, part = [], np.zeros(64)
    for t in range(ST):
        a = softplus(tile_gemm(blob, "OFF_WS", HK, t, hm[-1]) + blob.load16("OFF_BS", t))
        part += (blob.load16("OFF_WL0_ACC", t) * a).sum(1)
        f0, f1 = acc_to_frags(a)
        hs += [f0, f1]
    wpe = np.stack([blob.tab[blob.off["OFF_WL0_PE"] + h * 24: blob.off["OFF_WL0_PE"] + h * 24 + 24] for h in LH])
    part += (wpe * pev).sum(1)
    sdf = xhalf_sum(part) + blob.tab[blob.off["OFF_BL0"]]
    # feature
    feat = []
    for t in range(HT):
        acc = np.zeros((64, 16))
        for s in range(SK):
            acc = mfma(blob.wfrag("OFF_WL", SK + 3, t, s), hs[s], acc)
        for s in range(3):
            acc = mfma(blob.wfrag("OFF_WL", SK + 3, t, SK + s), pef[s], acc)
        f0, f1 = acc_to_frags(acc + blob.load16("OFF_BL", t))
        feat += [f0, f1]
    # normal sweep
    g_s = [blob.load8("OFF_WL0_FRAG", s) * sig_from_h(hs[s]) for s in range(SK)]
    gh_list = {"s": [blob.load8("OFF_WL0_FRAG", s) for s in range(SK)]}

    def nstep(offw, KS, NT, gin, hprev):
        gout, ghs = [], []
        for t in range(NT):
            acc = tile_gemm(blob, offw, KS, t, gin)
            ghs += [acc[:, :8].copy(), acc[:, 8:].copy()]
            gout += [acc[:, :8] * sig_from_h(hprev[2 * t]), acc[:, 8:] * sig_from_h(hprev[2 * t + 1])]
        return gout, ghs
    g, gh_m_last = nstep("OFF_WST", SK, HT, g_s, hm[-1])
    ga = {"s": g_s, "m": [None] * spec.NMID}
    gh = {"m": [None] * spec.NMID}
    ga["m"][spec.NMID - 1], gh["m"][spec.NMID - 1] = g, gh_m_last
    if spec.NMID == 2:
        g, ghh = nstep("OFF_WM1T", HK, HT, g, hm[0])
        ga["m"][0], gh["m"][0] = g, ghh
    g, ghh = nstep("OFF_WM0T", HK, HT, g, h1)
    ga["1"], gh["1"] = g, ghh
    part3 = np.zeros((64, 3))
    for t in range(2):
        acc = tile_gemm(blob, "OFF_W0T", HK, t, g)
        for r in range(16):
            q = 16 * t + r
            if q < 24:
                part3[:, q % 3] += ped[:, q] * (acc[:, r] + wpe[:, q])
    n = np.stack([xhalf_sum(part3[:, c]) for c in range(3)], 1)
    # colour
    xn = np.zeros((64, 8))
    for l in range(32):
        xn[l, 0:3] = x[l]
        xn[l, 3:6] = n[l]
    r1 = []
    for t in range(HT):
        acc = np.zeros((64, 16))
        for s in range(HK):
            acc = mfma(blob.wfrag("OFF_C0", HK + 1, t, s), feat[s], acc)
        acc = mfma(blob.wfrag("OFF_C0", HK + 1, t, HK), xn, acc)
        f0, f1 = acc_to_frags(np.maximum(acc + blob.load16("OFF_CB0", t), 0))
        r1 += [f0, f1]
    rs = [r1]
    if spec.NCMID == 1:
        rs.append(dense_layer(blob, "OFF_CM0", "OFF_CBM0", HK, HT, r1, 2))
    acco = tile_gemm(blob, "OFF_CH", HK, 0, rs[-1]) + blob.load16("OFF_CBH", 0)
    rgbl = 1 / (1 + np.exp(-acco[:, :4]))
    rgb = np.zeros((32, 6))
    rgb[:, 0:4] = rgbl[:32]
    rgb[:, 4:6] = rgbl[32:, 0:2]
    st.update(pev=pev, ped=ped, pef=pef, h1=h1, hm=hm, hs=hs, feat=feat, ga=ga, gh=gh, gh_s=gh_list["s"], xn=xn, rs=rs,
              rgbl=rgbl, wpe=wpe)
    return sdf[:32], n[:32], rgb, st


def make_sel():
    e0 = np.zeros((64, 8))
    e1 = np.zeros((64, 8))
    for l in range(64):
        n, h = l & 31, l >> 5
        for j in range(8):
            f = 8 * (j >> 2) + 4 * h + (j & 3)
            e0[l, j] = 1.0 if n == f else 0.0
            e1[l, j] = 1.0 if n == 16 + f else 0.0
    return e0, e1


E0, E1 = make_sel()


def panel_store(panels, tile, f0, f1):
    acc = mfma(f0, E0)
    acc = mfma(f1, E1, acc)
    panels[tile] = (acc[:, :8].copy(), acc[:, 8:].copy())


def backward_wave(spec: PK.NetSpec, blob: Blob, x, d_sdf, d_n, d_rgb):
    """mirrors mlp_bwd_kernel for one 32-point block; returns the panel dict {tile: (k0,k1)}."""
    P = blob.lay.panel
    HT, HK, ST, SK, NM = spec.HT, spec.HK, spec.ST, spec.SK, spec.NMID
    sdf, n, rgb, st = forward_wave(spec, blob, x)
    panels = {}
    z8 = np.zeros((64, 8))
    pef = st["pef"]
    panel_store(panels, P["H0"], pef[0], pef[1])
    panel_store(panels, P["H0"] + 1, pef[2], z8)

    def store_act(base, frags, NT):
        for t in range(NT):
            panel_store(panels, base + t, frags[2 * t], frags[2 * t + 1])
    store_act(P["H1"], st["h1"], HT)
    for m in range(NM):
        store_act(P["HM"] + m * HT, st["hm"][m], HT)
    store_act(P["HS"], st["hs"], ST)
    # phase B panels + q
    store_act(P["GAS"], st["ga"]["s"], ST)
    for m in range(NM):
        store_act(P["GAM"] + m * HT, st["ga"]["m"][m], HT)
    store_act(P["GA1"], st["ga"]["1"], HT)

    def spp(hfr):
        s = sig_from_h(hfr)
        return BETA * s * (1 - s)
    q_s = [st["gh_s"][s] * spp(st["hs"][s]) for s in range(SK)]
    q_m = [[st["gh"]["m"][m][s] * spp(st["hm"][m][s]) for s in range(HK)] for m in range(NM)]
    q_1 = [st["gh"]["1"][s] * spp(st["h1"][s]) for s in range(HK)]
    # phase C panels
    store_act(P["FEAT"], st["feat"], HT)
    panel_store(panels, P["XN"], st["xn"], z8)
    store_act(P["R1"], st["rs"][0], HT)
    if spec.NCMID == 1:
        store_act(P["R2"], st["rs"][1], HT)
    delta_o = np.zeros((64, 4))
    for l in range(64):
        h, p = l >> 5, l & 31
        for r in range(4):
            ch = 4 + r if h else r
            dr = d_rgb[p, ch] if ch < 6 else 0.0
            delta_o[l, r] = dr * st["rgbl"][l, r] * (1 - st["rgbl"][l, r])
    # phase D
    dof = np.zeros((64, 8))
    dof[:, :4] = delta_o
    panel_store(panels, P["DO"], dof, z8)
    r_last = st["rs"][-1]
    dl = []
    for t in range(HT):
        acc = tile_gemm(blob, "OFF_CHT", 1, t, [dof])
        dl += [np.where(r_last[2 * t] > 0, acc[:, :8], 0), np.where(r_last[2 * t + 1] > 0, acc[:, 8:], 0)]
    store_act(P["D2"] if spec.NCMID == 1 else P["D1"], dl, HT)
    if spec.NCMID == 1:
        d1 = []
        for t in range(HT):
            acc = tile_gemm(blob, "OFF_CM0T", HK, t, dl)
            d1 += [np.where(st["rs"][0][2 * t] > 0, acc[:, :8], 0), np.where(st["rs"][0][2 * t + 1] > 0, acc[:, 8:], 0)]
        store_act(P["D1"], d1, HT)
        dl = d1
    dfeat = []
    for t in range(HT):
        acc = tile_gemm(blob, "OFF_C0T", HK, t, dl)
        dfeat += [acc[:, :8].copy(), acc[:, 8:].copy()]
    store_act(P["DFEAT"], dfeat, HT)
    acc = tile_gemm(blob, "OFF_C0T", HK, HT, dl)
    a3, a0, a1 = acc[:, 3], acc[:, 0], acc[:, 1]
    sw = (LANES + 32) % 64
    dn0 = np.where(LH == 1, a3[sw], a3)
    dn1 = np.where(LH == 1, a0, a0[sw])
    dn2 = np.where(LH == 1, a1, a1[sw])
    nbar = np.stack([d_n[LP, 0] + dn0, d_n[LP, 1] + dn1, d_n[LP, 2] + dn2], 1)
    dsdf = d_sdf[LP]
    k0, k1 = z8.copy(), z8.copy()
    for l in range(64):
        if (l & 31) == 0:
            h = l >> 5
            for r in range(16):
                pt = (r & 3) + 8 * (r >> 2) + 4 * h
                if r < 8:
                    k0[l, r] = d_sdf[pt]
                else:
                    k1[l, r - 8] = d_sdf[pt]
    panels[P["SDF"]] = (k0, k1)
    # phase E
    ped = st["ped"]
    gb0 = [np.zeros((64, 8)) for _ in range(3)]
    for q in range(24):
        gb0[q >> 3][:, q & 7] = ped[:, q] * nbar[:, q % 3]
    panel_store(panels, P["GB0"], gb0[0], gb0[1])
    panel_store(panels, P["GB0"] + 1, gb0[2], z8)
    # column sums over the block's points (csrc/avc_bwd_body.h: col_sums): [ST tiles][half][16 acc registers] of gbar_hs, then
    # [3 fragments][half][8 slots] of gbar_h0 -- the half-wave of lanes 32 h .. 32 h + 31 holds the 32 points
    colsum = np.zeros(lay_cs_size(spec))
    for s_ in range(3):
        for h in range(2):
            colsum[ST * 32 + (s_ * 2 + h) * 8: ST * 32 + (s_ * 2 + h) * 8 + 8] = gb0[s_][32 * h:32 * h + 32].sum(0)

    def second(offw, KS, NT, gin, hfr, qfr, ptile):
        gout, ap = [], []
        for t in range(NT):
            acc = tile_gemm(blob, offw, KS, t, gin)
            gout += [acc[:, :8] * sig_from_h(hfr[2 * t]), acc[:, 8:] * sig_from_h(hfr[2 * t + 1])]
            ap += [acc[:, :8] * qfr[2 * t], acc[:, 8:] * qfr[2 * t + 1]]
        if ptile is not None:
            store_act(ptile, gout, NT)
        return gout, ap
    gb1, ap1 = second("OFF_W0G", 3, HT, gb0, st["h1"], q_1, P["GBH1"])
    gbm, apm = [], []
    g_in = gb1
    for m in range(NM):
        g_out, ap = second("OFF_WM0" if m == 0 else "OFF_WM1", HK, HT, g_in, st["hm"][m], q_m[m], P["GBHM"] + m * HT)
        gbm.append(g_out)
        apm.append(ap)
        g_in = g_out
    gbs, aps = second("OFF_WS", HK, ST, g_in, st["hs"], q_s, None)     # gbar_hs has no panel: it stays in registers ...
    for t in range(ST):                                                 # ... and its column sums go to the wavefront's slot
        for h in range(2):
            for half_regs, fr in ((0, gbs[2 * t]), (8, gbs[2 * t + 1])):
                colsum[(t * 2 + h) * 16 + half_regs: (t * 2 + h) * 16 + half_regs + 8] = fr[32 * h:32 * h + 32].sum(0)
    # phase F
    as_ = []
    for t in range(ST):
        acc = tile_gemm(blob, "OFF_WLT", HK, t, dfeat)
        wa = blob.load16("OFF_WL0_ACC", t)
        as_ += [aps[2 * t] + (acc[:, :8] + wa[:, :8] * dsdf[:, None] * PK.S_B2) * sig_from_h(st["hs"][2 * t]),
                aps[2 * t + 1] + (acc[:, 8:] + wa[:, 8:] * dsdf[:, None] * PK.S_B2) * sig_from_h(st["hs"][2 * t + 1])]
    store_act(P["ABS"], as_, ST)

    def reverse(offw, KS, NT, ain, hfr, apfr, ptile):
        aout = []
        for t in range(NT):
            acc = tile_gemm(blob, offw, KS, t, ain)
            aout += [apfr[2 * t] + acc[:, :8] * sig_from_h(hfr[2 * t]), apfr[2 * t + 1] + acc[:, 8:] * sig_from_h(hfr[2 * t + 1])]
        store_act(ptile, aout, NT)
        return aout
    am = reverse("OFF_WST", SK, HT, as_, st["hm"][-1], apm[-1], P["ABM"] + (NM - 1) * HT)
    if NM == 2:
        am = reverse("OFF_WM1T", HK, HT, am, st["hm"][0], apm[0], P["ABM"])
    reverse("OFF_WM0T", HK, HT, am, st["h1"], ap1, P["AB1"])
    panels["colsum"] = colsum
    return panels, (sdf, n, rgb)


def lay_cs_size(spec):
    return ((spec.H - 39 + 31) // 32) * 32 + 48


def weight_grad(lay: PK.Layout, panel_blocks):
    """mirrors avc_weight_grad over a list of panel dicts; returns (gout, gbias)."""
    gout = np.zeros(lay.gout_size)
    gbias = np.zeros(max(lay.gbias_size, 1))
    for (pa, ta, pb, tb, out_off, bias_off, _ta, _tb) in lay.pairs:
        for t_a in range(ta):
            for t_b in range(tb):
                acc = np.zeros((64, 16))
                for panels in panel_blocks:
                    a0, a1 = panels[pa + t_a]
                    b0, b1 = panels.get(pb + t_b, (np.zeros((64, 8)), np.zeros((64, 8))))
                    acc = mfma(a0, b0, acc)
                    acc = mfma(a1, b1, acc)
                base = out_off + (t_a * tb + t_b) * 64 * 16
                gout[base:base + 1024] += acc.reshape(-1)
            if bias_off >= 0:
                bs = np.zeros(64)
                for panels in panel_blocks:
                    a0, a1 = panels[pa + t_a]
                    bs += a0.sum(1) + a1.sum(1)
                bs = xhalf_sum(bs)
                gbias[bias_off + t_a * 32: bias_off + t_a * 32 + 32] += bs[:32]
    return gout, gbias


def unpack_grad(lay: PK.Layout, gout, gbias, colsum=None):
    grad = np.zeros(lay.nparam)
    np.add.at(grad, lay.un_tgt, gout[lay.un_src] * lay.un_scale)
    if lay.gbias_size:
        np.add.at(grad, lay.ub_tgt, gbias[lay.ub_src])
    if colsum is not None:      # second-order term of row 0 of the last SDF layer (engine.points_bwd: cs_total)
        np.add.at(grad, lay.cs_tgt, colsum[lay.cs_src] * lay.cs_scale)
    return grad
