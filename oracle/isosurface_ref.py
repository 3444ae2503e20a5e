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
