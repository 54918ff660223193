"""Marching-cubes tables and the CPU restatement of the mesh-extraction step."""
import numpy as np
import pytest
import torch

from avatarclip_amd import mc_tables as T
from oracle import mcubes_oracle as M


def test_tables_are_closed_oriented_and_symmetric():
    ntri, tab = T.tables()
    assert ntri[0] == 0 and ntri[255] == 0 and ntri.max() <= tab.shape[1]
    for m in range(256):
        loops = T.case_polygons(m)
        cut = {e for e in range(12) if ((m >> T.edge_corners(e)[0]) & 1) != ((m >> T.edge_corners(e)[1]) & 1)}
        assert {e for l in loops for e in l} == cut          # every cut edge is used, exactly once
        assert sum(len(l) for l in loops) == len(cut)
        assert all(len(l) >= 3 for l in loops)
        assert ntri[m] == sum(len(l) - 2 for l in loops)
    print("max triangles per cell", ntri.max())


def _mesh_stats(v, t, box=None):
    """area, signed volume, Euler characteristic; asserts a consistently oriented manifold that is closed -- or, with
    box = (lo, hi), open only along the faces of that box (an iso-surface cut by the grid boundary)"""
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    key = e[:, 0] * (v.shape[0] + 1) + e[:, 1]
    rkey = e[:, 1] * (v.shape[0] + 1) + e[:, 0]
    assert len(np.unique(key)) == len(key), "duplicate directed edge: inconsistent orientation"
    lonely = ~np.isin(key, rkey)
    if box is None:
        assert not lonely.any(), "boundary edge: the surface has a hole"
    else:
        lo, hi = box
        pts = v[e[lonely].reshape(-1)]
        on_face = (np.abs(pts - lo) < 1e-5) | (np.abs(pts - hi) < 1e-5)
        assert on_face.any(axis=1).all(), "an open edge away from the grid boundary: a crack"
    p0, p1, p2 = v[t[:, 0]].astype(np.float64), v[t[:, 1]].astype(np.float64), v[t[:, 2]].astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(p1 - p0, p2 - p0), axis=1).sum()
    vol = (np.einsum("ij,ij->i", p0, np.cross(p1, p2)).sum()) / 6.0
    euler = v.shape[0] - len(key) // 2 + t.shape[0]
    return area, vol, euler


def test_sphere_is_watertight_with_outward_normals():
    n = 33
    g = np.linspace(-1, 1, n, dtype=np.float32)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    r = 0.6
    u = r - np.sqrt(x * x + y * y + z * z)            # u = -sdf: positive inside (renderer.py:399-404)
    v, t = M.marching_cubes(u, 0.0)
    v = v / (n - 1.0) * 2.0 - 1.0
    area, vol, euler = _mesh_stats(v, t)
    assert euler == 2
    assert abs(vol - 4 / 3 * np.pi * r ** 3) / (4 / 3 * np.pi * r ** 3) < 0.02      # positive volume = outward normals
    assert abs(area - 4 * np.pi * r ** 2) / (4 * np.pi * r ** 2) < 0.02
    assert np.abs(np.linalg.norm(v, axis=1) - r).max() < 2e-3                        # vertices sit on the zero crossings


def test_ambiguous_configurations_do_not_crack():
    """two blobs touching diagonally and random smooth noise: closed surfaces whatever the ambiguous faces do"""
    rng = np.random.default_rng(0)
    n = 20
    f = rng.standard_normal((n, n, n)).astype(np.float32)
    for _ in range(2):                                  # mild smoothing, still full of ambiguous faces
        f = (f + np.roll(f, 1, 0) + np.roll(f, 1, 1) + np.roll(f, 1, 2)) / 4
    f = np.pad(f, 1, constant_values=-1.0)              # outside everywhere on the boundary -> closed surfaces
    v, t = M.marching_cubes(f, 0.0)
    assert t.shape[0] > 1000
    area, vol, euler = _mesh_stats(v, t)
    inside_frac = (f > 0).mean()
    assert vol > 0 and abs(vol / (f.size) - inside_frac) < 0.1     # crude: voxel fraction vs enclosed volume


def test_extract_geometry_scaling_matches_reference_formula():
    bmin, bmax = np.array([-1.0, -0.5, -0.25], np.float32), np.array([1.0, 0.5, 0.75], np.float32)
    q = lambda pts: 0.3 - pts.norm(dim=-1)
    v, t = M.extract_geometry(bmin, bmax, 24, 0.0, q)
    assert np.abs(np.linalg.norm(v, axis=1) - 0.3).max() < 5e-3
    assert (v.min(0) >= bmin).all() and (v.max(0) <= bmax).all()


def test_ply_round_trip(tmp_path):
    from avatarclip_amd import mesh
    v = np.random.default_rng(1).standard_normal((10, 3)).astype(np.float32)
    t = np.array([[0, 1, 2], [2, 3, 4], [7, 8, 9]], np.int32)
    c = np.random.default_rng(2).integers(0, 255, (10, 3)).astype(np.uint8)
    p = str(tmp_path / "m.ply")
    mesh.write_ply(p, v, t, c)
    v2, t2, c2 = mesh.read_ply(p)
    assert np.array_equal(v, v2) and np.array_equal(t, t2) and np.array_equal(c, c2[:, :3]) and (c2[:, 3] == 255).all()
    head = open(p, "rb").read(200).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 10\n")


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@gpu
def test_hip_marching_cubes_equals_oracle_bit_for_bit():
    """integer / index work: exact.  Vertices: one IEEE division and one addition per vertex, also exact."""
    from avatarclip_amd import mesh
    dev = torch.device("cuda")
    rng = np.random.default_rng(3)
    cases = []
    g = np.linspace(-1, 1, 40, dtype=np.float32)
    x, y, z = np.meshgrid(g, g[:33], g[:25], indexing="ij")           # non-cubic grid: nx, ny, nz all different
    cases.append((0.55 - np.sqrt(x * x + y * y + z * z)).astype(np.float32))
    f = rng.standard_normal((21, 19, 23)).astype(np.float32)
    for _ in range(2):
        f = (f + np.roll(f, 1, 0) + np.roll(f, 1, 1) + np.roll(f, 1, 2)) / 4
    cases.append(f)                                                     # ambiguous faces, surface cut by the boundary
    cases.append(np.full((5, 6, 7), -1.0, np.float32))                  # empty: no vertex, no triangle
    for u in cases:
        for iso in (0.0, 0.013):
            v_ref, t_ref = M.marching_cubes(u, iso)
            v, t = mesh.marching_cubes(torch.from_numpy(u).to(dev), iso)
            assert v.shape == (v_ref.shape[0], 3) and t.shape == (t_ref.shape[0], 3)
            assert np.array_equal(t.cpu().numpy().astype(np.int64), t_ref)
            assert np.array_equal(v.cpu().numpy(), v_ref)


@gpu
def test_extract_geometry_of_the_sdf_network():
    """the SDF of the seeded full-size net (geometric init: a sphere of radius ~0.5): field values vs the fp32 oracle,
    closed outward-oriented surface, vertices on the zero level set"""
    from avatarclip_amd import fields, renderer
    from oracle import neus_oracle as O
    dev = torch.device("cuda")
    torch.manual_seed(0)
    sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                            geometric_init=True, weight_norm=True).to(dev)
    col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2,
                                  weight_norm=True, multires_view=0, squeeze_out=True, extra_color=True).to(dev)
    var = fields.SingleVarianceNetwork(0.3).to(dev)
    ren = renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)
    bmin, bmax = torch.tensor([-1.01] * 3), torch.tensor([1.01] * 3)
    res = 48
    v, t = ren.extract_geometry(bmin, bmax, res, 0.0)
    sd = {k: p.detach().cpu() for k, p in sdf.named_parameters()}
    v_ref, t_ref = M.extract_geometry(bmin.numpy(), bmax.numpy(), res, 0.0, lambda pts: -O.sdf_forward(sd, pts)[:, 0])
    # (the untrained net is a noisy sphere: several components, some cut by the box.)  f16-MFMA field vs fp32 field: sign
    # flips only where |sdf| is within the f16 error of zero, so the meshes agree up to a few cells; no cracks in either
    assert abs(v.shape[0] - v_ref.shape[0]) <= 0.01 * v_ref.shape[0]
    box = (bmin.numpy(), bmax.numpy())
    area, vol, euler = _mesh_stats(v, t.astype(np.int64), box)
    a_ref, vol_ref, e_ref = _mesh_stats(v_ref, t_ref, box)
    assert abs(area - a_ref) / a_ref < 5e-3
    # vertices sit on the linearly interpolated zero crossings: the residual sdf is the interpolation error of a 0.04-wide
    # cell (same for the oracle's mesh) plus the f16 field error
    s = O.sdf_forward(sd, torch.from_numpy(v).float())[:, 0].abs()
    s_ref = O.sdf_forward(sd, torch.from_numpy(v_ref).float())[:, 0].abs()
    print("vertices", v.shape[0], "ref", v_ref.shape[0], "area", area, a_ref, "max |sdf| at vertices", s.max().item(), s_ref.max().item())
    assert s.max() < s_ref.max() + 5e-3 and abs(s.mean() - s_ref.mean()) < 5e-4


@gpu
def test_runner_validate_mesh_writes_coloured_ply(tmp_path):
    """Runner.validate_mesh (main.py:850-919) end to end on seeded small nets: geometry + six-view vertex colours + PLY"""
    import bench
    from avatarclip_amd import mesh
    from avatarclip_amd.runner import Runner
    conf = bench.make_conf(64, 64, small=True)
    conf.put("general.base_exp_dir", str(tmp_path))
    torch.manual_seed(0)
    runner = Runner(None, mode="validate_mesh", conf=conf, device=torch.device("cuda"))
    path = runner.validate_mesh(world_space=True, resolution=40, threshold=0.0)
    v, t, c = mesh.read_ply(path)
    assert v.shape[0] > 100 and t.shape[0] > 100 and c is not None and c.shape == (v.shape[0], 4)
    assert t.min() >= 0 and t.max() < v.shape[0]
    assert np.isfinite(v).all() and (np.abs(v) <= 1.01 + 1e-5).all()
    assert c[:, :3].std() > 0          # colours were actually picked from renders, not left constant
