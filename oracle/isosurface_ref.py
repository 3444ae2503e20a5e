"""TEST INFRASTRUCTURE -- CPU restatement of the marching-tetrahedra surface extraction of csrc/isosurface.hip.

Parity note: the reference meshes with PyMCubes + trimesh (utils/eval_3D.py:123-153), un-vendored third-party
packages that are absent here (requirements.yaml: pymcubes unpinned, trimesh=3.12.0) -> **parity unpinned** for the
triangulation itself.  What is pinned: this restatement == the HIP kernel triangle by triangle, and analytic
properties (sphere area, vertices on the iso-surface).  Pure-Python loops: small grids only.
"""
import numpy as np

TETS = ((0, 1, 3, 7), (0, 3, 2, 7), (0, 2, 6, 7), (0, 6, 4, 7), (0, 4, 5, 7), (0, 5, 1, 7))


def _vertex(f, g, S, a, b, iso):
    pa = (g[0] + (a & 1), g[1] + ((a >> 1) & 1), g[2] + ((a >> 2) & 1))
    pb = (g[0] + (b & 1), g[1] + ((b >> 1) & 1), g[2] + ((b >> 2) & 1))
    fa, fb = np.float32(f[a]), np.float32(f[b])
    if (pb[0] * S + pb[1]) * S + pb[2] < (pa[0] * S + pa[1]) * S + pa[2]:
        pa, pb, fa, fb = pb, pa, fb, fa
    t = np.float32(np.float32(iso) - fa) / np.float32(fb - fa)
    return [np.float32(pa[k]) + t * np.float32(pb[k] - pa[k]) for k in range(3)]


def marching_tets(level, iso=0.0):
    """level [S,S,S] float32 -> triangles [T,3,3] float32 (grid-index units), in the kernel's emission order."""
    level = np.asarray(level, dtype=np.float32)
    S = level.shape[0]
    out = []
    for x in range(S - 1):
        for y in range(S - 1):
            for z in range(S - 1):
                f = [level[x + (v & 1), y + ((v >> 1) & 1), z + ((v >> 2) & 1)] for v in range(8)]
                if all(v < iso for v in f) or not any(v < iso for v in f):
                    continue
                g = (x, y, z)
                for tet in TETS:
                    ins = [v for v in tet if f[v] < iso]
                    outs = [v for v in tet if not f[v] < iso]
                    if len(ins) in (0, 4):
                        continue
                    V = lambda a, b: _vertex(f, g, S, a, b, iso)
                    if len(ins) == 2:
                        out.append([V(ins[0], outs[0]), V(ins[0], outs[1]), V(ins[1], outs[1])])
                        out.append([V(ins[0], outs[0]), V(ins[1], outs[1]), V(ins[1], outs[0])])
                    else:
                        apex, base = (ins[0], outs) if len(ins) == 1 else (outs[0], ins)
                        out.append([V(apex, base[0]), V(apex, base[1]), V(apex, base[2])])
    return np.asarray(out, dtype=np.float32).reshape(-1, 3, 3)


def triangle_areas(tris):
    e1, e2 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    return 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)


# ---------------------------------------------------------------------------------------------------------------------------
# Marching cubes (the algorithm of the reference's `mcubes.marching_cubes(level, 0)`, utils/eval_3D.py:123-153; PyMCubes itself is
# absent -> the triangulation is parity unpinned, the VERTEX SET is not: every marching-cubes implementation puts exactly one vertex
# on each grid edge whose end values lie on different sides of the iso-value, at the linear interpolation point).  Restated cube by
# cube without the kernel's table: crossing edges are joined into loops face by face (a face with four crossings cuts off its inside
# corners), loops are oriented inside -> outside and fan-triangulated from their lowest-numbered edge.
_CORNER = [((v & 1), (v >> 1) & 1, (v >> 2) & 1) for v in range(8)]
_EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if (a ^ b) in (1, 2, 4)]


def _face_cycles():
    cycles = []
    for axis, (u, w) in enumerate(((1, 2), (2, 0), (0, 1))):
        for side in (0, 1):
            cyc = []
            for du, dw in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[axis], p[u], p[w] = side, du, dw
                cyc.append(p[0] + 2 * p[1] + 4 * p[2])
            cycles.append(cyc)
    return cycles


_FACES = _face_cycles()


def mc_case(inside):
    """inside: 8 booleans (corner order of the kernel) -> triangles as triples of edge numbers (index into _EDGES)."""
    eid = {e: i for i, e in enumerate(_EDGES)}
    edge = lambda a, b: eid[(a, b) if a < b else (b, a)]
    link = {}
    for cyc in _FACES:
        fe = [edge(cyc[i], cyc[(i + 1) % 4]) for i in range(4)]
        hit = [i for i in range(4) if inside[cyc[i]] != inside[cyc[(i + 1) % 4]]]
        pairs = []
        if len(hit) == 2:
            pairs = [(fe[hit[0]], fe[hit[1]])]
        elif len(hit) == 4:
            pairs = [(fe[(k - 1) % 4], fe[k]) for k in range(4) if inside[cyc[k]]]
        for a, b in pairs:
            link.setdefault(a, []).append(b)
            link.setdefault(b, []).append(a)
    tris, done = [], set()
    for first in sorted(link):
        if first in done:
            continue
        loop, before, here = [first], first, min(link[first])
        while here != first:
            loop.append(here)
            nxt = link[here][1] if link[here][0] == before else link[here][0]
            before, here = here, nxt
        done.update(loop)
        mids = np.array([[(_CORNER[_EDGES[e][0]][k] + _CORNER[_EDGES[e][1]][k]) / 2 for k in range(3)] for e in loop])
        rel = mids - mids.mean(0)
        area_vec = sum(np.cross(rel[i], rel[(i + 1) % len(loop)]) for i in range(len(loop)))
        out_dir = np.zeros(3)
        for e in loop:
            a, b = _EDGES[e]
            src, dst = (a, b) if inside[a] else (b, a)
            out_dir += np.array(_CORNER[dst]) - np.array(_CORNER[src])
        if float(area_vec @ out_dir) < 0:
            loop = loop[:1] + loop[:0:-1]
        tris += [(loop[0], loop[i], loop[i + 1]) for i in range(1, len(loop) - 1)]
    return tris


def marching_cubes(level, iso=0.0):
    """level [S,S,S] float32 -> triangles [T,3,3] float32 (grid-index units), cube-major like the kernel."""
    level = np.asarray(level, dtype=np.float32)
    S = level.shape[0]
    out = []
    for x in range(S - 1):
        for y in range(S - 1):
            for z in range(S - 1):
                f = [level[x + (v & 1), y + ((v >> 1) & 1), z + ((v >> 2) & 1)] for v in range(8)]
                inside = [bool(v < iso) for v in f]
                if all(inside) or not any(inside):
                    continue
                for tri in mc_case(inside):
                    out.append([_vertex(f, (x, y, z), S, _EDGES[e][0], _EDGES[e][1], iso) for e in tri])
    return np.asarray(out, dtype=np.float32).reshape(-1, 3, 3)


def crossing_edge_vertices(level, iso=0.0):
    """The vertex set of marching cubes, computed without any triangulation: for every grid edge (p, p + unit axis) whose end values
    lie on different sides of iso (inside = value < iso), the linear interpolation point.  -> [V,3] float32, rows sorted."""
    level = np.asarray(level, dtype=np.float32)
    S = level.shape[0]
    pts = []
    idx = np.stack(np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing="ij"), -1).astype(np.float32)
    for axis in range(3):
        lo = [slice(None)] * 3; hi = [slice(None)] * 3
        lo[axis] = slice(0, S - 1); hi[axis] = slice(1, S)
        fa, fb = level[tuple(lo)], level[tuple(hi)]
        cross = (fa < iso) != (fb < iso)
        t = (np.float32(iso) - fa[cross]) / (fb[cross] - fa[cross])
        p = idx[tuple(lo)][cross].copy()
        p[:, axis] = p[:, axis] + t * np.float32(1.0)
        pts.append(p)
    pts = np.concatenate(pts, 0).astype(np.float32)
    return pts[np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))]
