"""CPU proof that packing.py + the kernels' fragment index arithmetic (mirrored in tests/wave_emulator.py)
reproduce oracle/analytic.py: forward values and every dense weight gradient."""
import numpy as np
import pytest
import torch

from avatarclip_amd import packing as PK
from oracle import analytic as A
from tests import wave_emulator as E
from tests.helpers import load_case


def flat_from_net(net, lay):
    vals = {}
    for l, (w, b) in enumerate(zip(net["sdf_W"], net["sdf_b"])):
        vals["sdf.W%d" % l], vals["sdf.b%d" % l] = w, b
    nc = len(net["col_W"])
    for l in range(nc - 1):
        vals["col.W%d" % l], vals["col.b%d" % l] = net["col_W"][l], net["col_b"][l]
    vals["col.Wh"], vals["col.bh"] = net["col_W"][-1], net["col_b"][-1]
    parts = []
    for name, shp in lay.shapes:
        assert tuple(vals[name].shape) == tuple(shp), (name, vals[name].shape, shp)
        parts.append(vals[name].detach().double().reshape(-1).numpy())
    return np.concatenate(parts)


def dense_grads_flat(mb, lay):
    vals = {}
    for l, (w, b) in enumerate(zip(mb["sdf_dW"], mb["sdf_db"])):
        vals["sdf.W%d" % l], vals["sdf.b%d" % l] = w, b
    nc = len(mb["col_dW"])
    for l in range(nc - 1):
        vals["col.W%d" % l], vals["col.b%d" % l] = mb["col_dW"][l], mb["col_db"][l]
    vals["col.Wh"], vals["col.bh"] = mb["col_dW"][-1], mb["col_db"][-1]
    return np.concatenate([vals[n].double().reshape(-1).numpy() for n, _ in lay.shapes])


@pytest.mark.parametrize("name,spec", [("neus_small.npz", PK.SMALL), ("neus_full.npz", PK.FULL)])
def test_emulated_wave_matches_analytic(name, spec):
    rec, sd_sdf, sd_col, variance = load_case(name)
    net = A.dense_net(sd_sdf, sd_col, torch.float64)
    lay = PK.layout_for(spec)
    flat = flat_from_net(net, lay)
    blob = E.Blob(lay, flat)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(64, 3, generator=g, dtype=torch.float64) * 2 - 1) * 0.8
    d_sdf = torch.randn(64, 1, generator=g, dtype=torch.float64)
    d_n = torch.randn(64, 3, generator=g, dtype=torch.float64)
    d_rgb = torch.randn(64, 6, generator=g, dtype=torch.float64)
    f = A.mlp_forward(net, x)
    mb = A.mlp_backward(net, f, d_sdf, d_n, d_rgb)
    ref = dense_grads_flat(mb, lay)
    blocks = []
    for b in range(2):
        sl = slice(32 * b, 32 * b + 32)
        panels, (sdf, n, rgb) = E.backward_wave(spec, blob, x[sl].numpy(), d_sdf[sl, 0].numpy(), d_n[sl].numpy(),
                                                d_rgb[sl].numpy())
        assert np.allclose(sdf, f["sdf"][sl, 0].numpy(), atol=1e-9)
        assert np.allclose(n, f["n"][sl].numpy(), atol=1e-8)
        assert np.allclose(rgb, f["rgb6"][sl].numpy(), atol=1e-9)
        blocks.append(panels)
    gout, gbias = E.weight_grad(lay, blocks)
    assert blocks[0]["colsum"].shape == (lay.cs_size,)
    grad = E.unpack_grad(lay, gout, gbias, sum(b["colsum"] for b in blocks))
    off = 0
    for pname, shp in lay.shapes:
        n_ = int(np.prod(shp))
        a, b_ = grad[off:off + n_], ref[off:off + n_]
        err = np.abs(a - b_).max() / (np.abs(b_).max() + 1e-30)
        assert err < 1e-6, (pname, err)
        off += n_


def test_unpack_tables_as_per_parameter_lists_equal_the_scatter_statement():
    """engine._DevLayout.csr_* (what avc_weight_grad_unpack gathers with) against the index_add statement of the same map
    (packing.Layout.un_* / ub_*): every dense parameter receives exactly the tile entries that scatter into it, with their scales."""
    import numpy as np
    import torch
    from avatarclip_amd import packing as PK
    from avatarclip_amd.engine import _DevLayout
    for spec in (PK.NetSpec(128, 2, 1), PK.NetSpec(256, 2, 1)):
        lay = PK.layout_for(spec)
        dl = _DevLayout(lay, "cpu")
        rng = np.random.default_rng(3)
        acc = torch.from_numpy(rng.standard_normal(lay.gout_size + lay.gbias_size))
        ref = torch.zeros(lay.nparam, dtype=torch.float64)
        ref.index_add_(0, torch.from_numpy(lay.un_tgt), acc[torch.from_numpy(lay.un_src)] * torch.from_numpy(lay.un_scale).double())
        ref.index_add_(0, torch.from_numpy(np.asarray(lay.ub_tgt, np.int64)), acc[lay.gout_size + torch.from_numpy(np.asarray(lay.ub_src, np.int64))])
        off, src, scl = dl.csr_off.long(), dl.csr_src.long(), dl.csr_scale.double()
        assert off[0] == 0 and off[-1] == len(src) == len(lay.un_src) + len(lay.ub_src)
        seg = torch.repeat_interleave(torch.arange(lay.nparam), off[1:] - off[:-1])
        got = torch.zeros(lay.nparam, dtype=torch.float64).index_add_(0, seg, acc[src] * scl)
        assert torch.allclose(got, ref, rtol=0, atol=1e-12)
        assert int((off[1:] - off[:-1]).min()) >= 1          # every parameter of the two networks has a source


def test_transposing_lane_reduction_of_the_column_sums():
    """csrc/avc_bwd_body.h: transpose_sum -- NV values per lane -> ONE per lane, the sum over the half-wave's 32 lanes of value (lane & (NV - 1)):
    every butterfly step keeps, of a pair (a, b), the one the lane's own bit selects and adds the partner's copy of the same one.  Emulated
    on 64 lanes with the exchange partners the kernel uses (xor 1, 2 = DPP quad permutes; xor 4, 8, 16 = ds_bpermute)."""
    rs = np.random.RandomState(0)
    lane = np.arange(64)

    def transpose_sum(v):                      # v [64 lanes, NV]
        nv = v.shape[1]
        vals = [v[:, k] for k in range(nv)]
        bit = 0
        while len(vals) > 1:                   # xor 1, 2, 4 (, 8)
            b = (lane >> bit) & 1
            partner = lane ^ (1 << bit)
            nxt = []
            for k in range(len(vals) // 2):
                keep = np.where(b == 1, vals[2 * k + 1], vals[2 * k])
                give = np.where(b == 1, vals[2 * k], vals[2 * k + 1])
                nxt.append(keep + give[partner])
            vals, bit = nxt, bit + 1
        z = vals[0]
        while bit < 5:                         # the remaining exchanges up to xor 16: plain sums
            z = z + z[lane ^ (1 << bit)]
            bit += 1
        return z
    for nv in (8, 16):
        v = rs.randn(64, nv)
        z = transpose_sum(v)
        for l in range(64):
            half = slice(32 * (l >> 5), 32 * (l >> 5) + 32)
            assert abs(z[l] - v[half, l & (nv - 1)].sum()) < 1e-12
