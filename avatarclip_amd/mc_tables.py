"""Marching-cubes case tables, GENERATED (not transcribed) so that they are provably crack-free.

The reference calls `mcubes.marching_cubes(u, threshold)` (renderer.py:30; PyMCubes, a C++ extension that is not vendored
and not installed here).  Its published algorithm is Lorensen & Cline's marching cubes with vertices at the linear
zero crossings of the grid edges; the vertex SET of the isosurface is therefore determined by `u` alone, the
triangulation of each cell's polygon(s) is table-specific.  We build our own table:

  * corner c of a cell has offsets (c & 1, (c >> 1) & 1, (c >> 2) & 1) along (x, y, z); bit c of the case index is set
    when the corner is INSIDE (u > iso: the reference extracts u = -sdf at threshold 0, renderer.py:399-404);
  * edge e = 4 a + b: runs along axis a from the corner whose other two coordinates are (b & 1, b >> 1) (in axis order);
  * on every cell face the cut edges are joined pairwise; a face with four cut edges (alternating corners) is resolved by
    cutting off each INSIDE corner -- a rule that only looks at the face, so the two cells sharing it agree (no cracks);
  * the segments close into loops, oriented so that triangle normals point from inside to outside; each loop gets the
    lexicographically first triangulation none of whose diagonals lies in a cell face (such a diagonal could be produced
    by the neighbouring cell as well and make the edge non-manifold).

`tables()` returns (ntri[256] int32, tri[256, MAXT, 3] int8 of cell-edge ids, -1 padded).
"""
import functools

import numpy as np

AXES = ((1, 0, 0), (0, 1, 0), (0, 0, 1))


def corner_offset(c):
    return (c & 1, (c >> 1) & 1, (c >> 2) & 1)


def corner_id(x, y, z):
    return x | (y << 1) | (z << 2)


def edge_corners(e):
    """the two corners (low, high along the axis) of cell edge e"""
    a, b = e >> 2, e & 3
    others = [ax for ax in range(3) if ax != a]
    p = [0, 0, 0]
    p[others[0]] = b & 1
    p[others[1]] = b >> 1
    q = list(p)
    q[a] = 1
    return corner_id(*p), corner_id(*q)


def edge_id(c0, c1):
    lo, hi = min(c0, c1), max(c0, c1)
    d = lo ^ hi
    a = {1: 0, 2: 1, 4: 2}[d]
    p = corner_offset(lo)
    others = [ax for ax in range(3) if ax != a]
    return 4 * a + (p[others[0]] | (p[others[1]] << 1))


def _faces():
    """each face: 4 corners in counter-clockwise order seen from OUTSIDE the cell"""
    faces = []
    for a in range(3):
        u, v = (a + 1) % 3, (a + 2) % 3          # (a, u, v) right-handed: u x v = +a
        for side in (0, 1):
            cyc = [(0, 0), (1, 0), (1, 1), (0, 1)]   # ccw seen from +a
            if side == 0:
                cyc = cyc[::-1]                      # the outward normal of the low face is -a
            cs = []
            for (pu, pv) in cyc:
                p = [0, 0, 0]
                p[a], p[u], p[v] = side, pu, pv
                cs.append(corner_id(*p))
            faces.append(cs)
    return faces


FACES = _faces()


def case_polygons(m):
    """oriented loops of cell-edge ids for case index m (bit c set = corner c inside)"""
    inside = [(m >> c) & 1 for c in range(8)]
    nxt = {}          # directed segment: cut edge -> next cut edge of the loop
    for cs in FACES:
        cuts = []     # (position k, edge between corner k and k+1 of the face cycle)
        for k in range(4):
            c0, c1 = cs[k], cs[(k + 1) % 4]
            if inside[c0] != inside[c1]:
                cuts.append((k, edge_id(c0, c1)))
        if not cuts:
            continue
        # walking the face boundary counter-clockwise (seen from outside): a cut where we LEAVE the inside region is the
        # start of a segment that runs, with the inside on its left... we orient every segment from the cut that enters
        # the inside region to the cut that leaves it when following the boundary, i.e. the inside corner(s) between the
        # two cuts lie to the right of the directed segment seen from outside  => normals point outward after the fan.
        if len(cuts) == 2:
            (k0, e0), (k1, e1) = cuts
            # boundary step k goes corner k -> k+1; the step is "entering" when corner k+1 is inside
            enter0 = inside[cs[(k0 + 1) % 4]] == 1
            a, b = (e0, e1) if enter0 else (e1, e0)
            nxt[(a, id(cs))] = (b, cs)
        else:
            # four cuts: isolate every inside corner (pair the two cuts around it)
            for k in range(4):
                c = cs[k]
                if inside[c]:
                    e_in = edge_id(cs[(k - 1) % 4], c)     # step (k-1) enters the inside corner
                    e_out = edge_id(c, cs[(k + 1) % 4])    # step k leaves it
                    nxt[(e_in, id(cs))] = (e_out, cs)
    # every cut edge lies on exactly two faces: it ends one segment and starts another
    starts = {}
    for (a, fid), (b, cs) in nxt.items():
        starts.setdefault(a, []).append(b)
    loops, used = [], set()
    succ = {}
    for a, bs in starts.items():
        assert len(bs) == 1, (m, a, bs)
        succ[a] = bs[0]
    for a in sorted(succ):
        if a in used:
            continue
        loop, e = [], a
        while e not in used:
            used.add(e)
            loop.append(e)
            e = succ[e]
        assert e == a, "open loop"
        loops.append(loop)
    return loops


def _edge_faces(e):
    """the two cell faces (axis, side) a cell edge lies on"""
    a, b = e >> 2, e & 3
    others = [ax for ax in range(3) if ax != a]
    return {(others[0], b & 1), (others[1], b >> 1)}


def _share_face(e0, e1):
    return bool(_edge_faces(e0) & _edge_faces(e1))


def _triangulations(poly):
    """all triangulations of a polygon given as a tuple of vertex labels (orientation preserved)"""
    n = len(poly)
    if n < 3:
        return [[]]
    if n == 3:
        return [[tuple(poly)]]
    out = []
    # the edge (poly[0], poly[-1]) belongs to exactly one triangle (poly[0], poly[k], poly[-1])
    for k in range(1, n - 1):
        for left in _triangulations(poly[:k + 1]):
            for right in _triangulations(poly[k:]):
                out.append(left + [(poly[0], poly[k], poly[-1])] + right)
    return out


def _bad_diagonals(tris, loop):
    """diagonals (non-boundary triangle sides) whose two vertices lie on a common cell face: such a side lies IN that face,
    where the neighbouring cell may produce the same side -> a non-manifold edge.  A good table has none."""
    n = len(loop)
    boundary = {frozenset((loop[k], loop[(k + 1) % n])) for k in range(n)}
    bad = 0
    for t in tris:
        for k in range(3):
            side = frozenset((t[k], t[(k + 1) % 3]))
            if side not in boundary and _share_face(*side):
                bad += 1
    return bad


def _triangulate(loop):
    best = None
    for tris in _triangulations(tuple(loop)):
        key = (_bad_diagonals(tris, loop), tris)
        if best is None or key < best:
            best = key
    return best


@functools.lru_cache(maxsize=None)
def tables():
    tris = []
    for m in range(256):
        t = []
        for loop in case_polygons(m):
            bad, tl = _triangulate(loop)
            assert bad == 0, ("no face-diagonal-free triangulation", m, loop)
            t.extend(tl)
        tris.append(t)
    maxt = max(len(t) for t in tris)
    ntri = np.array([len(t) for t in tris], np.int32)
    tab = np.full((256, maxt, 3), -1, np.int8)
    for m, t in enumerate(tris):
        for k, tri in enumerate(t):
            tab[m, k] = tri
    return ntri, tab


def edge_table():
    """[12, 4] int32: (dx, dy, dz, axis) of the grid edge (owned by its low corner) for every cell edge"""
    out = np.zeros((12, 4), np.int32)
    for e in range(12):
        lo, _ = edge_corners(e)
        out[e, :3] = corner_offset(lo)
        out[e, 3] = e >> 2
    return out
