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
    grad = E.unpack_grad(lay, gout, gbias)
    off = 0
    for pname, shp in lay.shapes:
        n_ = int(np.prod(shp))
        a, b_ = grad[off:off + n_], ref[off:off + n_]
        err = np.abs(a - b_).max() / (np.abs(b_).max() + 1e-30)
        assert err < 1e-6, (pname, err)
        off += n_
