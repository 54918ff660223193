"""Host-side parameter packing for the gfx950 MLP kernels (pure index arithmetic, numpy + torch gathers).

The kernels keep activations in MFMA-operand registers (csrc/avc_common.h), which fixes a permutation between
k-slots / accumulator rows and feature indices.  This module bakes that permutation into
  * the packed 16-bit weight blob (one layout, materialised as f16 for the forward path and bf16 for the
    gradient sweeps) and the fp32 table blob (biases, row 0 of the last SDF layer),
  * the map from the weight-gradient GEMM outputs (csrc/avc_mlp_bwd.hip: avc_weight_grad) back to dense dW / db.

Dense parameter order of the flat vector `P`:
  sdf lin0.W, lin0.b, ..., lin{L-1}.W, lin{L-1}.b, colour lin0.W, lin0.b, [lin1.W, lin1.b], heads.W (6xH =
  [lin_last ; extra_lin]), heads.b (6)         (reference modules: models/fields.py:24-68,128-150)
"""
import math
import os
import re
from dataclasses import dataclass, field
from functools import lru_cache
from typing import Dict, List, Tuple

import numpy as np

SQ2 = math.sqrt(2.0)
S_B2 = 100.0 * math.log2(math.e)   # base-2 softplus unit (AVC_S in csrc/avc_common.h)
_HERE = os.path.dirname(os.path.abspath(__file__))


def _parse_offsets_enum() -> Dict[str, int]:
    src = open(os.path.join(_HERE, "csrc", "avc_common.h")).read()
    body = re.search(r"enum AvcOff \{(.*?)\};", src, re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    names = [n.strip().split("=")[0].strip() for n in body.split(",") if n.strip()]
    return {n: i for i, n in enumerate(names)}


OFF = _parse_offsets_enum()
OFF_COUNT = OFF["OFF_COUNT"]


def frag_feature(s, h, j):
    return 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3)


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def pe_feat(h, q, lo_as_x):
    """PE slot (half h, q=8*kstep+j) -> index into the reference's 39-wide embedding (embedder.py:35-36) or -1."""
    if h == 0:
        if q < 3:
            return q
        if q < 21:
            k, rem = divmod(q - 3, 6)
            return 3 + 6 * k + rem if rem < 3 else 6 + 6 * k + (rem - 3)
        return (q - 21) if lo_as_x else -1
    if q < 18:
        k, rem = divmod(q, 6)
        k += 3
        return 3 + 6 * k + rem if rem < 3 else 6 + 6 * k + (rem - 3)
    return -1


def kmap_std(F, KS, shift=0):
    m = np.full((KS, 2, 8), -1, np.int64)
    for s in range(KS):
        for h in range(2):
            for j in range(8):
                f = frag_feature(s, h, j)
                if f < F:
                    m[s, h, j] = f + shift
    return m


def kmap_pe(lo_as_x, shift=0):
    m = np.full((3, 2, 8), -1, np.int64)
    for s in range(3):
        for h in range(2):
            for j in range(8):
                f = pe_feat(h, 8 * s + j, lo_as_x)
                if f >= 0:
                    m[s, h, j] = f + shift
    return m


def rows_std(F, NT, shift=0):
    r = np.full(32 * NT, -1, np.int64)
    n = min(F, 32 * NT)
    r[:n] = np.arange(n) + shift
    return r


def rows_pe():
    """rows of W0^T tiles: acc (tile t, half h, reg r) <-> pe slot (h, q = 16 t + r)."""
    r = np.full(64, -1, np.int64)
    for t in range(2):
        for h in range(2):
            for reg in range(16):
                q = 16 * t + reg
                if q < 24:
                    r[32 * t + acc_row(reg, h)] = pe_feat(h, q, False)
    return r


@dataclass
class NetSpec:
    H: int
    NMID: int
    NCMID: int

    @property
    def net_id(self):
        return 0 if self.H == 256 else 1

    @property
    def HT(self): return self.H // 32
    @property
    def HK(self): return self.H // 16
    @property
    def SKIP(self): return self.H - 39
    @property
    def ST(self): return (self.SKIP + 31) // 32
    @property
    def SK(self): return 2 * self.ST
    @property
    def n_sdf(self): return self.NMID + 3
    @property
    def n_col(self): return self.NCMID + 2   # lin0, [lin1], heads


FULL = NetSpec(256, 2, 1)
SMALL = NetSpec(128, 1, 0)


def spec_from_conf(sdf_conf: dict, col_conf: dict) -> NetSpec:
    H = int(sdf_conf["d_hidden"])
    nl = int(sdf_conf["n_layers"])
    spec = NetSpec(H, nl - 2, int(col_conf["n_layers"]) - 1)
    ok = (spec.H, spec.NMID, spec.NCMID) in ((256, 2, 1), (128, 1, 0))
    ok = ok and int(sdf_conf.get("multires", 0)) == 6 and list(sdf_conf.get("skip_in", [])) == [nl]
    ok = ok and int(sdf_conf["d_out"]) == H + 1 and int(col_conf["d_hidden"]) == H
    ok = ok and col_conf.get("mode") == "no_view_dir" and int(col_conf.get("multires_view", 0)) == 0
    ok = ok and float(sdf_conf.get("scale", 1.0)) == 1.0
    # everything else the kernels hard-code: 3-D points in (weight norm is resolved on the host, either way works), a 6-D [x, n] colour input next to the feature vector, 3 sigmoid colour channels (+ the 3 of the extra head)
    ok = ok and int(sdf_conf.get("d_in", 3)) == 3
    ok = ok and int(col_conf.get("d_in", 6)) == 6 and int(col_conf.get("d_out", 3)) == 3
    ok = ok and int(col_conf.get("d_feature", H)) == H and bool(col_conf.get("squeeze_out", True))
    if not ok:
        raise NotImplementedError(
            "avatarclip_amd kernels are instantiated for the two network shapes the reference ships "
            "(confs/examples: 256-wide, confs/examples_small: 128-wide); got %r / %r" % (sdf_conf, col_conf))
    return spec


def param_shapes(spec: NetSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    H, S = spec.H, spec.SKIP
    out = [("sdf.W0", (H, 39)), ("sdf.b0", (H,))]
    for m in range(spec.NMID):
        out += [("sdf.W%d" % (m + 1), (H, H)), ("sdf.b%d" % (m + 1), (H,))]
    ls, ll = spec.NMID + 1, spec.NMID + 2
    out += [("sdf.W%d" % ls, (S, H)), ("sdf.b%d" % ls, (S,))]
    out += [("sdf.W%d" % ll, (H + 1, H)), ("sdf.b%d" % ll, (H + 1,))]
    out += [("col.W0", (H, 6 + H)), ("col.b0", (H,))]
    for m in range(spec.NCMID):
        out += [("col.W%d" % (m + 1), (H, H)), ("col.b%d" % (m + 1), (H,))]
    out += [("col.Wh", (6, H)), ("col.bh", (6,))]
    return out


@dataclass
class Layout:
    spec: NetSpec
    shapes: List[Tuple[str, Tuple[int, ...]]]
    pbase: Dict[str, int]
    nparam: int
    idx16: np.ndarray
    scale16: np.ndarray
    idx32: np.ndarray
    scale32: np.ndarray
    offsets: np.ndarray
    panel: Dict[str, int]
    pairs: List[Tuple[int, int, int, int, int, int, int, int]]   # (pa, ta, pb, tb, out_off, bias_off, type_a, type_b); pa / pb = GLOBAL tile ids (see build_layout), type 0 = F region / f16 tile, 1 = G region / bf16 tile
    gout_size: int
    gbias_size: int
    un_src: np.ndarray
    un_tgt: np.ndarray
    un_scale: np.ndarray
    ub_src: np.ndarray
    ub_tgt: np.ndarray
    # the second-order term of row 0 of the last SDF layer, dW_last[0, :] += sum_points gbar_u, is a COLUMN SUM over the
    # points of gbar_hs and gbar_h0 -- not a product.  The backward kernel reduces it itself (csrc/avc_bwd_body.h: col_sums) and leaves, per
    # wavefront, cs_size floats: [ST tiles][2 halves][16 accumulator registers] of gbar_hs, then [3 fragments][2 halves][8 slots] of gbar_h0.
    cs_size: int = 0
    cs_src: np.ndarray = None     # index into that vector ...
    cs_tgt: np.ndarray = None     # ... flat parameter index (row 0 of the last SDF layer) ...
    cs_scale: np.ndarray = None   # ... and factor


def _elem(pbase, name, shape, row, col, transposed):
    """flat index of W[row, col] (or W[col,row] read transposed)."""
    ld = shape[1]
    return pbase[name] + (col * ld + row if transposed else row * ld + col)


@lru_cache(maxsize=None)
def build_layout(H: int, NMID: int, NCMID: int) -> Layout:
    spec = NetSpec(H, NMID, NCMID)
    shapes = param_shapes(spec)
    pbase, n = {}, 0
    for name, shp in shapes:
        pbase[name] = n
        n += int(np.prod(shp))
    ZERO = n  # index of the trailing zero element of P
    shp = dict(shapes)
    HT, HK, SKIP, ST, SK = spec.HT, spec.HK, spec.SKIP, spec.ST, spec.SK
    ls, ll = "sdf.W%d" % (NMID + 1), "sdf.W%d" % (NMID + 2)
    bs, bl = "sdf.b%d" % (NMID + 1), "sdf.b%d" % (NMID + 2)

    idx16, sc16, offsets = [], [], np.zeros(OFF_COUNT, np.int32)
    cur16 = [0]

    def pack(off_name, wname, rows, kmap, transposed=False, scale=1.0, kscale=None):
        NT, KS = len(rows) // 32, kmap.shape[0]
        arr = np.full((NT, KS, 64, 8), ZERO, np.int64)
        sca = np.full((NT, KS, 64, 8), scale, np.float32)
        if kscale is not None:   # per k-slot scale [KS, 2, 8]
            for lane in range(64):
                sca[:, :, lane, :] *= kscale[None, :, lane >> 5, :]
        for lane in range(64):
            hh, i = lane >> 5, lane & 31
            for t in range(NT):
                row = rows[32 * t + i]
                if row < 0:
                    continue
                cols = kmap[:, hh, :]  # [KS, 8]
                ok = cols >= 0
                ld = shp[wname][1]
                if transposed:
                    flat = pbase[wname] + cols * ld + row
                else:
                    flat = pbase[wname] + row * ld + cols
                arr[t, :, lane, :] = np.where(ok, flat, ZERO)
        offsets[OFF[off_name]] = cur16[0]
        idx16.append(arr.reshape(-1))
        sc16.append(sca.reshape(-1))
        cur16[0] += arr.size

    pack("OFF_W0", "sdf.W0", rows_std(H, HT), kmap_pe(True), scale=S_B2)          # forward: t1 = S a1
    pack("OFF_W0G", "sdf.W0", rows_std(H, HT), kmap_pe(True))                     # second-order sweep: unscaled
    pack("OFF_WM0", "sdf.W1", rows_std(H, HT), kmap_std(H, HK))
    if NMID == 2:
        pack("OFF_WM1", "sdf.W2", rows_std(H, HT), kmap_std(H, HK))
    pack("OFF_WS", ls, rows_std(SKIP, ST), kmap_std(H, HK))
    ks_last = np.concatenate([np.full((SK, 2, 8), 1.0 / S_B2, np.float32), np.ones((3, 2, 8), np.float32)], 0)
    pack("OFF_WL", ll, rows_std(H, HT, shift=1), np.concatenate([kmap_std(SKIP, SK), kmap_pe(True, shift=SKIP)], 0),
         scale=1.0 / SQ2, kscale=ks_last)                                         # skip features arrive as H = S h
    pack("OFF_W0T", "sdf.W0", rows_pe(), kmap_std(H, HK), transposed=True)
    pack("OFF_WM0T", "sdf.W1", rows_std(H, HT), kmap_std(H, HK), transposed=True)
    if NMID == 2:
        pack("OFF_WM1T", "sdf.W2", rows_std(H, HT), kmap_std(H, HK), transposed=True)
    pack("OFF_WST", ls, rows_std(H, HT), kmap_std(SKIP, SK), transposed=True)
    pack("OFF_WLT", ll, rows_std(SKIP, ST), kmap_std(H, HK, shift=1), transposed=True, scale=1.0 / SQ2)
    xn = np.full((1, 2, 8), -1, np.int64)
    xn[0, 0, :6] = np.arange(6)
    pack("OFF_C0", "col.W0", rows_std(H, HT), np.concatenate([kmap_std(H, HK, shift=6), xn], 0))
    if NCMID == 1:
        pack("OFF_CM0", "col.W1", rows_std(H, HT), kmap_std(H, HK))
    pack("OFF_CH", "col.Wh", rows_std(6, 1), kmap_std(H, HK))
    rows_c0t = np.concatenate([rows_std(H, HT, shift=6), rows_std(6, 1)])
    pack("OFF_C0T", "col.W0", rows_c0t, kmap_std(H, HK), transposed=True)
    if NCMID == 1:
        pack("OFF_CM0T", "col.W1", rows_std(H, HT), kmap_std(H, HK), transposed=True)
    pack("OFF_CHT", "col.Wh", rows_std(H, HT), kmap_std(6, 1), transposed=True)

    idx32, sc32 = [], []
    cur32 = [0]

    def table(off_name, flat_idx, scale=1.0):
        flat_idx = np.asarray(flat_idx, np.int64).reshape(-1)
        pad = (-len(flat_idx)) % 4  # keep every table 16-byte aligned for the float4 loads
        flat_idx = np.concatenate([flat_idx, np.full(pad, ZERO, np.int64)])
        offsets[OFF[off_name]] = cur32[0]
        idx32.append(flat_idx)
        sc32.append(np.full(len(flat_idx), scale, np.float32))
        cur32[0] += len(flat_idx)

    def bias_tab(bname, F, NT, shift=0):
        arr = np.full((NT, 2, 16), ZERO, np.int64)
        for t in range(NT):
            for h in range(2):
                for r in range(16):
                    f = 32 * t + acc_row(r, h)
                    if f < F:
                        arr[t, h, r] = pbase[bname] + f + shift
        return arr

    table("OFF_B0", bias_tab("sdf.b0", H, HT), S_B2)
    table("OFF_BM0", bias_tab("sdf.b1", H, HT), S_B2)
    if NMID == 2:
        table("OFF_BM1", bias_tab("sdf.b2", H, HT), S_B2)
    table("OFF_BS", bias_tab(bs, SKIP, ST), S_B2)
    table("OFF_BL", bias_tab(bl, H, HT, shift=1))
    table("OFF_BL0", [pbase[bl]])
    a = np.full((ST, 2, 16), ZERO, np.int64)
    for t in range(ST):
        for h in range(2):
            for r in range(16):
                f = 32 * t + acc_row(r, h)
                if f < SKIP:
                    a[t, h, r] = pbase[ll] + f   # row 0 of the last layer
    table("OFF_WL0_ACC", a, 1.0 / (SQ2 * S_B2))   # multiplies H = S h in the fp32 sdf dot product
    a = np.full((SK, 2, 8), ZERO, np.int64)
    for s in range(SK):
        for h in range(2):
            for j in range(8):
                f = frag_feature(s, h, j)
                if f < SKIP:
                    a[s, h, j] = pbase[ll] + f
    table("OFF_WL0_FRAG", a, 1.0 / SQ2)
    a = np.full((2, 24), ZERO, np.int64)
    for h in range(2):
        for q in range(24):
            f = pe_feat(h, q, False)
            if f >= 0:
                a[h, q] = pbase[ll] + SKIP + f
    table("OFF_WL0_PE", a, 1.0 / SQ2)
    table("OFF_CB0", bias_tab("col.b0", H, HT))
    if NCMID == 1:
        table("OFF_CBM0", bias_tab("col.b1", H, HT))
    table("OFF_CBH", bias_tab("col.bh", 6, 1))
    offsets[OFF["OFF_TAB_END"]] = cur32[0]

    # ---------------- panel layout (mirror of PanelLayout in csrc/avc_mlp.h): two regions with their own block stride.
    # P[name] is a GLOBAL tile id: F-region tiles (forward-type operands, f16, written by the forward kernel for the whole ray set)
    # are numbered 0 .. FTILES-1 in region order, G-region tiles (gradient-type operands, bf16, written by the backward kernel for
    # one slab at a time) FTILES .. FTILES+GTILES-1; the region-local index the kernels use is id (F) resp. id - FTILES (G).
    P = {}
    c = 0
    F_ORDER = [("H1", HT), ("HM", NMID * HT), ("HS", ST), ("H0", 2), ("GA1", HT), ("GAM", NMID * HT), ("GAS", ST), ("FEAT", HT),
               ("XN", 1), ("R1", HT), ("R2", NCMID * HT)]
    # (no gbar_hs panel and no constant-one panel since round 5: their only consumer was the product "1 (x) [gbar_hs | gbar_h0]" = a column
    # sum over the points, which the backward kernel now takes from its registers: 8 tiles less written, 10 tile reads less per block)
    G_ORDER = [("GBH1", HT), ("GBHM", NMID * HT), ("GB0", 2), ("AB1", HT), ("ABM", NMID * HT), ("ABS", ST),
               ("DFEAT", HT), ("SDF", 1), ("D1", HT), ("D2", NCMID * HT), ("DO", 1)]
    tile_type = []
    for region, order in ((0, F_ORDER), (1, G_ORDER)):
        for name, nt in order:
            P[name] = c
            c += nt
            tile_type += [region] * nt
        if region == 0:
            P["FTILES"] = c
    P["GTILES"] = c - P["FTILES"]
    P["TILES"] = c

    def panel_type(tile):
        return tile_type[tile]

    # ---------------- weight-gradient pairs and the map of their outputs back to the dense gradient
    pairs, un_src, un_tgt, un_scale, ub_src, ub_tgt = [], [], [], [], [], []
    gout, gbias = [0], [0]

    def feat_std(F, shift=0):
        return lambda f: (f + shift) if f < F else -1

    def feat_pe(lo_as_x, shift=0):
        inv = {}
        for h in range(2):
            for q in range(24):
                synth = frag_feature(q >> 3, h, q & 7)
                inv[synth] = pe_feat(h, q, lo_as_x)
        return lambda f: (inv[f] + shift) if f in inv and inv[f] >= 0 else -1

    def feat_xn():
        inv = {frag_feature(0, 0, j): j for j in range(6)}
        return lambda f: inv.get(f, -1)

    def add_pair(pa, ta, pb, tb, wname, rowmap, colmap, scale=1.0, bname=None, bias_rowmap=None):
        out_off = gout[0]
        bias_off = -1
        if bname is not None:
            bias_off = gbias[0]
            for t in range(ta):
                for nn in range(32):
                    row = (bias_rowmap or rowmap)(32 * t + nn)
                    if row >= 0:
                        ub_src.append(bias_off + 32 * t + nn)
                        ub_tgt.append(pbase[bname] + row)
            gbias[0] += 32 * ta
        ld = shp[wname][1]
        scale_of = scale if callable(scale) else (lambda t_b: scale)
        rows_of = np.array([[[rowmap(32 * t + acc_row(r, h)) for r in range(16)] for h in range(2)] for t in range(ta)])
        cols_of = np.array([[colmap(32 * t + nn) for nn in range(32)] for t in range(tb)])
        for t_a in range(ta):
            for t_b in range(tb):
                base = out_off + (t_a * tb + t_b) * 64 * 16
                for lane in range(64):
                    h, nn = lane >> 5, lane & 31
                    col = cols_of[t_b, nn]
                    if col < 0:
                        continue
                    rws = rows_of[t_a, h]
                    ok = rws >= 0
                    if not ok.any():
                        continue
                    src = base + lane * 16 + np.nonzero(ok)[0]
                    un_src.append(src)
                    un_tgt.append(pbase[wname] + rws[ok] * ld + col)
                    un_scale.append(np.full(len(src), scale_of(t_b), np.float32))
        assert len({panel_type(pa + t) for t in range(ta)}) == 1 and len({panel_type(pb + t) for t in range(tb)}) == 1
        pairs.append((pa, ta, pb, tb, out_off, bias_off, panel_type(pa), panel_type(pb)))
        gout[0] += ta * tb * 64 * 16

    # SDF layer 0
    add_pair(P["AB1"], HT, P["H0"], 2, "sdf.W0", feat_std(H), feat_pe(True), bname="sdf.b0")
    add_pair(P["GA1"], HT, P["GB0"], 2, "sdf.W0", feat_std(H), feat_pe(False))
    # middle layers: input of middle m is h1 (m=0) or hm[m-1]
    for m in range(NMID):
        wn, bn = "sdf.W%d" % (m + 1), "sdf.b%d" % (m + 1)
        pin = P["H1"] if m == 0 else P["HM"] + (m - 1) * HT
        gin = P["GBH1"] if m == 0 else P["GBHM"] + (m - 1) * HT
        add_pair(P["ABM"] + m * HT, HT, pin, HT, wn, feat_std(H), feat_std(H), scale=1 / S_B2, bname=bn)
        add_pair(P["GAM"] + m * HT, HT, gin, HT, wn, feat_std(H), feat_std(H))
    # skip layer
    add_pair(P["ABS"], ST, P["HM"] + (NMID - 1) * HT, HT, ls, feat_std(SKIP), feat_std(H), scale=1 / S_B2, bname=bs)
    add_pair(P["GAS"], ST, P["GBHM"] + (NMID - 1) * HT, HT, ls, feat_std(SKIP), feat_std(H))
    # last layer: rows 1..H (ybar[1:]) and row 0 (d_sdf ; second-order term through the constant-one panel)
    r1 = lambda f: (f + 1) if f < H else -1
    r0 = lambda f: 0 if f == 0 else -1
    # the d_sdf tile carries the cotangent SPLIT in two bf16 slots (feature 0 = hi, feature 1 = d_sdf - hi; csrc/avc_bwd_body.h): both
    # are contributions to row 0 -- d_sdf keeps 16 bits of mantissa, so the row-0 weight gradient and the sdf bias (a sum with heavy
    # cancellation: the eikonal term pulls both ways) come out of the same product as every other row, without a special case
    r0s = lambda f: 0 if f in (0, 1) else -1
    assert frag_feature(0, 0, 0) == 0 and frag_feature(0, 0, 1) == 1
    # (its input [hs | pe] = the adjacent panels HS, H0: one product over ST + 2 column tiles)
    assert P["H0"] == P["HS"] + ST

    def last_cols(lo_as_x):
        a, b = feat_std(SKIP), feat_pe(lo_as_x, shift=SKIP)
        return lambda f: a(f) if f < 32 * ST else b(f - 32 * ST)
    hs_scale = lambda t_b: 1 / (SQ2 * S_B2) if t_b < ST else 1 / SQ2
    # rows 1..H (DFEAT tiles) and row 0 (the SDF tile right behind them) in ONE product: [hs | pe] is read once for both
    assert P["SDF"] == P["DFEAT"] + HT
    rows_last = lambda f: r1(f) if f < 32 * HT else r0s(f - 32 * HT)
    add_pair(P["DFEAT"], HT + 1, P["HS"], ST + 2, ll, rows_last, last_cols(True), scale=hs_scale, bname=bl)
    # second-order term of row 0: sum_points [gbar_hs | gbar_h0] / sqrt2, reduced inside the backward kernel (Layout.cs_*)
    cs_src, cs_tgt = [], []
    for t in range(ST):
        for h in range(2):
            for r in range(16):
                f = 32 * t + acc_row(r, h)
                if f < SKIP:
                    cs_src.append((t * 2 + h) * 16 + r)
                    cs_tgt.append(pbase[ll] + f)
    for q in range(24):
        for h in range(2):
            f = pe_feat(h, q, False)
            if f >= 0:
                cs_src.append(ST * 32 + ((q >> 3) * 2 + h) * 8 + (q & 7))
                cs_tgt.append(pbase[ll] + SKIP + f)
    assert len(set(cs_tgt)) == len(cs_tgt) == SKIP + 39, "every column of row 0 exactly once"
    # colour
    fx = feat_xn()
    c0col = lambda f: (6 + f) if f < H else fx(f - H)
    add_pair(P["D1"], HT, P["FEAT"], HT + 1, "col.W0", feat_std(H), c0col, bname="col.b0")
    if NCMID == 1:
        add_pair(P["D2"], HT, P["R1"], HT, "col.W1", feat_std(H), feat_std(H), bname="col.b1")
        add_pair(P["DO"], 1, P["R2"], HT, "col.Wh", feat_std(6), feat_std(H), bname="col.bh")
    else:
        add_pair(P["DO"], 1, P["R1"], HT, "col.Wh", feat_std(6), feat_std(H), bname="col.bh")

    return Layout(spec=spec, shapes=shapes, pbase=pbase, nparam=n,
                  idx16=np.concatenate(idx16), scale16=np.concatenate(sc16),
                  idx32=np.concatenate(idx32), scale32=np.concatenate(sc32), offsets=offsets, panel=P, pairs=pairs,
                  gout_size=gout[0], gbias_size=gbias[0],
                  un_src=np.concatenate(un_src), un_tgt=np.concatenate(un_tgt), un_scale=np.concatenate(un_scale),
                  ub_src=np.asarray(ub_src, np.int64), ub_tgt=np.asarray(ub_tgt, np.int64),
                  cs_size=ST * 32 + 48, cs_src=np.asarray(cs_src, np.int64), cs_tgt=np.asarray(cs_tgt, np.int64),
                  cs_scale=np.full(len(cs_src), 1 / SQ2, np.float32))


def layout_for(spec: NetSpec) -> Layout:
    return build_layout(spec.H, spec.NMID, spec.NCMID)


def region_local_pairs(lay: Layout) -> np.ndarray:
    """the pair table the C ABI takes (avc_weight_grad_all): tile indices local to their region (F: id, G: id - FTILES)"""
    ft = lay.panel["FTILES"]
    out = np.asarray(lay.pairs, dtype=np.int32).reshape(-1, 8).copy()
    out[:, 0] -= np.where(out[:, 6] == 1, ft, 0)
    out[:, 2] -= np.where(out[:, 7] == 1, ft, 0)
    assert (out[:, 0] >= 0).all() and (out[:, 2] >= 0).all()
    return np.ascontiguousarray(out)
