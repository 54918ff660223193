"""Mesh extraction of the reference (`extract_fields`, `extract_geometry`: models/renderer.py:10-36; PLY export of
`Runner.validate_mesh`: main.py:850-919) on the device: the resolution^3 field u = -sdf comes from `avc_sdf_forward`
(the same f16-MFMA kernel the hierarchical sampler uses), marching cubes from `avc_mc_classify` / `avc_mc_emit`.
"""
import struct

import numpy as np
import torch

from . import lib as L
from . import mc_tables as T

_dev_tables = {}


def _tables(device):
    key = str(device)
    if key not in _dev_tables:
        ntri, tab = T.tables()
        assert tab.shape[1] == 5
        _dev_tables[key] = (torch.from_numpy(ntri.astype(np.int32)).to(device),
                            torch.from_numpy(tab.reshape(-1).astype(np.int8)).to(device),
                            torch.from_numpy(T.edge_table().reshape(-1).astype(np.int32)).to(device))
    return _dev_tables[key]


def extract_fields(bound_min, bound_max, resolution, query_func, device, slab=64):
    """u[res,res,res] = query_func(points) on the reference's grid (renderer.py:10-25): linspace per axis, x outermost.
    The reference walks 64^3 bricks through host memory; here x-slabs of `slab` planes stay on the device."""
    bmin = [float(v) for v in bound_min]
    bmax = [float(v) for v in bound_max]
    X = torch.linspace(bmin[0], bmax[0], resolution, device=device)
    Y = torch.linspace(bmin[1], bmax[1], resolution, device=device)
    Z = torch.linspace(bmin[2], bmax[2], resolution, device=device)
    u = torch.empty(resolution, resolution, resolution, device=device, dtype=torch.float32)
    with torch.no_grad():
        for x0 in range(0, resolution, slab):
            xs = X[x0:x0 + slab]
            xx, yy, zz = torch.meshgrid(xs, Y, Z, indexing="ij")
            pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1)
            u[x0:x0 + len(xs)] = query_func(pts).reshape(len(xs), resolution, resolution)
    return u


def marching_cubes(u: torch.Tensor, iso: float):
    """device marching cubes: (vertices [NV,3] float32 in index coordinates, triangles [NT,3] int32)"""
    assert u.is_cuda and u.dtype == torch.float32 and u.dim() == 3
    u = u.contiguous()
    nx, ny, nz = u.shape
    n = nx * ny * nz
    assert 3 * n < 2 ** 31, "int32 vertex ids: at most ~894^3 grid points"
    lib = L.load()
    ntri, tab, etab = _tables(u.device)
    vflag = torch.empty(3 * n, device=u.device, dtype=torch.int32)
    ccount = torch.empty(n, device=u.device, dtype=torch.int32)
    L.check(lib.avc_mc_classify(L.ptr(u), nx, ny, nz, float(iso), L.ptr(ntri), L.ptr(vflag), L.ptr(ccount), L.stream()),
            "avc_mc_classify")
    vinc = torch.cumsum(vflag, 0, dtype=torch.int32)
    cinc = torch.cumsum(ccount, 0, dtype=torch.int32)
    nv, nt = int(vinc[-1].item()), int(cinc[-1].item())
    vid = vinc - vflag           # exclusive scans
    coff = cinc - ccount
    verts = torch.empty(nv, 3, device=u.device, dtype=torch.float32)
    tris = torch.empty(nt, 3, device=u.device, dtype=torch.int32)
    if nv and nt:
        L.check(lib.avc_mc_emit(L.ptr(u), nx, ny, nz, float(iso), L.ptr(vflag), L.ptr(vid), L.ptr(ccount), L.ptr(coff),
                                L.ptr(tab), L.ptr(etab), L.ptr(verts), L.ptr(tris), L.stream()), "avc_mc_emit")
    return verts, tris


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func, device):
    """renderer.py:28-36: numpy vertices (world units) and triangles"""
    u = extract_fields(bound_min, bound_max, resolution, query_func, device)
    v, t = marching_cubes(u, threshold)
    bmin = np.asarray([float(x) for x in bound_min], np.float32)
    bmax = np.asarray([float(x) for x in bound_max], np.float32)
    vertices = v.cpu().numpy() / (resolution - 1.0) * (bmax - bmin)[None, :] + bmin[None, :]
    return vertices, t.cpu().numpy()


def write_ply(path, vertices, triangles, vertex_colors=None):
    """binary little-endian PLY with per-vertex RGBA, the layout trimesh's exporter writes (main.py:914-915)"""
    vertices = np.asarray(vertices, np.float32)
    triangles = np.asarray(triangles, np.int32)
    header = ["ply", "format binary_little_endian 1.0", "element vertex %d" % len(vertices),
              "property float x", "property float y", "property float z"]
    if vertex_colors is not None:
        vc = np.asarray(vertex_colors)
        if vc.shape[1] == 3:
            vc = np.concatenate([vc, np.full((len(vc), 1), 255, vc.dtype)], 1)
        vc = vc.astype(np.uint8)
        header += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
    header += ["element face %d" % len(triangles), "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        if vertex_colors is not None:
            rec = np.empty(len(vertices), dtype=[("p", "<f4", 3), ("c", "u1", 4)])
            rec["p"], rec["c"] = vertices, vc
        else:
            rec = np.empty(len(vertices), dtype=[("p", "<f4", 3)])
            rec["p"] = vertices
        f.write(rec.tobytes())
        frec = np.empty(len(triangles), dtype=[("n", "u1"), ("i", "<i4", 3)])
        frec["n"], frec["i"] = 3, triangles
        f.write(frec.tobytes())


def read_ply(path):
    """inverse of write_ply (tests)"""
    with open(path, "rb") as f:
        lines = []
        while True:
            line = f.readline().decode("ascii").strip()
            lines.append(line)
            if line == "end_header":
                break
        nv = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
        nf = int([l for l in lines if l.startswith("element face")][0].split()[-1])
        has_c = any("uchar red" in l for l in lines)
        vdt = [("p", "<f4", 3), ("c", "u1", 4)] if has_c else [("p", "<f4", 3)]
        v = np.frombuffer(f.read(nv * np.dtype(vdt).itemsize), dtype=vdt)
        fr = np.frombuffer(f.read(nf * 13), dtype=[("n", "u1"), ("i", "<i4", 3)])
    return v["p"].copy(), fr["i"].copy(), (v["c"].copy() if has_c else None)
