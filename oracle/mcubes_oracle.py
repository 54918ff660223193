"""CPU restatement of the mesh-extraction step (TEST INFRASTRUCTURE: imported only by tests/).

Reference: `extract_fields` / `extract_geometry` (models/renderer.py:10-36) and `mcubes.marching_cubes(u, threshold)`
(PyMCubes: third-party C++ extension, `requirements.txt`, not vendored and not installed here -> **parity against
PyMCubes itself is unpinned**).  PyMCubes implements Lorensen & Cline marching cubes with vertices at the linear zero
crossing of every sign-changing grid edge; that vertex set is reproduced here exactly (it is a function of `u` alone), the
per-cell triangulation comes from the generated table of `avatarclip_amd/mc_tables.py` (shared with the HIP kernel on
purpose: the table is data, the algorithm is restated independently in numpy).

Ordering contract (what makes HIP == oracle comparable bit for bit):
  vertices  : sorted by (linear grid index of the edge's low corner in C order) * 3 + axis
  triangles : sorted by linear cell index (C order over (nx-1, ny-1, nz-1)), then table order
"""
import numpy as np

from avatarclip_amd import mc_tables as T


def marching_cubes(u: np.ndarray, iso: float):
    u = np.asarray(u, np.float32)
    nx, ny, nz = u.shape
    inside = u > np.float32(iso)
    # ---- vertices on sign-changing edges
    vid = np.full((nx, ny, nz, 3), -1, np.int64)
    flags = np.zeros((nx, ny, nz, 3), bool)
    flags[:-1, :, :, 0] = inside[:-1] != inside[1:]
    flags[:, :-1, :, 1] = inside[:, :-1] != inside[:, 1:]
    flags[:, :, :-1, 2] = inside[:, :, :-1] != inside[:, :, 1:]
    idx = np.flatnonzero(flags.reshape(-1))
    vid.reshape(-1)[idx] = np.arange(len(idx))
    p = idx // 3
    a = idx % 3
    i, j, k = np.unravel_index(p, (nx, ny, nz))
    i2, j2, k2 = i + (a == 0), j + (a == 1), k + (a == 2)
    u0, u1 = u[i, j, k], u[i2, j2, k2]
    t = ((np.float32(iso) - u0) / (u1 - u0)).astype(np.float32)
    verts = np.stack([i, j, k], -1).astype(np.float32)
    verts[np.arange(len(idx)), a] += t
    # ---- triangles
    ntri, tab = T.tables()
    etab = T.edge_table()
    case = np.zeros((nx - 1, ny - 1, nz - 1), np.int32)
    for c in range(8):
        dx, dy, dz = T.corner_offset(c)
        case |= inside[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz].astype(np.int32) << c
    cells = np.flatnonzero(ntri[case.reshape(-1)] > 0)
    ci, cj, ck = np.unravel_index(cells, case.shape)
    tris = []
    for n in range(len(cells)):
        m = case[ci[n], cj[n], ck[n]]
        for q in range(ntri[m]):
            tri = []
            for e in tab[m, q]:
                dx, dy, dz, ax = etab[e]
                tri.append(vid[ci[n] + dx, cj[n] + dy, ck[n] + dz, ax])
            tris.append(tri)
    tris = np.asarray(tris, np.int64).reshape(-1, 3)
    assert (tris >= 0).all()
    return verts, tris


def extract_fields(bound_min, bound_max, resolution, query_func):
    """renderer.py:10-25 (the 64^3 chunking of the reference is a memory device, not arithmetic)"""
    import torch
    X = torch.linspace(float(bound_min[0]), float(bound_max[0]), resolution)
    Y = torch.linspace(float(bound_min[1]), float(bound_max[1]), resolution)
    Z = torch.linspace(float(bound_min[2]), float(bound_max[2]), resolution)
    xx, yy, zz = torch.meshgrid(X, Y, Z, indexing="ij")
    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
    with torch.no_grad():
        return query_func(pts).reshape(resolution, resolution, resolution).numpy()


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func):
    """renderer.py:28-36"""
    u = extract_fields(bound_min, bound_max, resolution, query_func)
    v, t = marching_cubes(u, threshold)
    bmin, bmax = np.asarray(bound_min, np.float32), np.asarray(bound_max, np.float32)
    v = v / (resolution - 1.0) * (bmax - bmin)[None, :] + bmin[None, :]
    return v, t
