"""Iso-surface extraction on the GPU (csrc/isosurface.hip: marching cubes, the evaluation's default, and marching tetrahedra) against
the CPU restatements (oracle/isosurface_ref.py) and analytic properties.  The reference's PyMCubes/trimesh step (utils/eval_3D.py:
123-153) is third-party and absent: the triangulation is parity-unpinned; what is checked is kernel == restatement triangle by
triangle (bit-exact, same fp32 interpolation and emission order), the marching-cubes VERTEX SET (one vertex per sign-changing grid
edge -- what PyMCubes produces too) at the evaluation's vox_res = 100, the area of a sphere, and area-uniform sampling."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sphere(S, r, centre=(0.0, 0.0, 0.0)):
    ax = np.linspace(-1, 1, S)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return (np.sqrt((X - centre[0]) ** 2 + (Y - centre[1]) ** 2 + (Z - centre[2]) ** 2) - r).astype(np.float32)


def test_triangles_equal_cpu_restatement_bit_exact():
    from oracle.isosurface_ref import marching_cubes, marching_tets
    from shapeclipper_amd import ops
    rng = np.random.RandomState(0)
    grids = [rng.randn(7, 7, 7).astype(np.float32),                       # noise: every case of every tetrahedron
             _sphere(13, 0.55, (0.1, -0.05, 0.2)),
             np.full((5, 5, 5), 1.0, np.float32),                         # no surface
             _sphere(13, 0.5)]                                            # exact zeros on grid vertices (ties)
    grids[3][6, 6, 1] = 0.0
    for g in grids:
        for method, restated in (("tetrahedra", marching_tets), ("cubes", marching_cubes)):
            tris, per = ops.isosurface_triangles(torch.tensor(g[None]).cuda(), 0.0, method=method)
            ref = restated(g, 0.0)
            assert int(per[0]) == ref.shape[0] == tris.shape[0], method
            assert np.array_equal(tris.cpu().numpy(), ref), method
    with pytest.raises(ValueError):
        ops.isosurface_triangles(torch.tensor(grids[0][None]).cuda(), 0.0, method="dual contouring")


def test_batched_grid_and_offsets():
    from oracle.isosurface_ref import marching_tets
    from shapeclipper_amd import ops
    a, b, c = _sphere(9, 0.5), np.full((9, 9, 9), -1.0, np.float32), _sphere(9, 0.7, (0.1, 0.1, 0.0))
    from oracle.isosurface_ref import marching_cubes
    for method, restated in (("tetrahedra", marching_tets), ("cubes", marching_cubes)):
        tris, per = ops.isosurface_triangles(torch.tensor(np.stack([a, b, c])).cuda(), 0.0, method=method)
        ra, rc = restated(a), restated(c)
        assert per.tolist() == [ra.shape[0], 0, rc.shape[0]]
        assert np.array_equal(tris.cpu().numpy(), np.concatenate([ra, rc]))


def test_marching_cubes_vertex_set_at_evaluation_resolution():
    """vox_res = 100 (BASELINE config[4]): the vertices of the device mesh are exactly the sign-changing grid edges' interpolation
    points -- the vertex set of `mcubes.marching_cubes(level, 0)` up to its float64 interpolation -- on a sphere-with-dent field
    and on noise; every triangle has three distinct vertices; isovalue != 0 works the same."""
    from oracle.isosurface_ref import crossing_edge_vertices
    from shapeclipper_amd import ops
    S = 101
    ax = np.linspace(-0.6, 0.6, S)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    shape = (np.sqrt(X * X + Y * Y + 1.4 * Z * Z) - 0.41 + 0.05 * np.sin(9 * X) * np.cos(7 * Y)).astype(np.float32)
    noise = np.random.RandomState(5).randn(33, 33, 33).astype(np.float32)
    for level, iso in ((shape, 0.0), (shape, 0.03), (noise, 0.0)):
        tris, per = ops.isosurface_triangles(torch.tensor(level[None]).cuda(), iso, method="cubes")
        assert int(per[0]) == tris.shape[0] > 1000
        got = torch.unique(tris.reshape(-1, 3), dim=0).cpu().numpy()
        want = np.unique(crossing_edge_vertices(level, iso), axis=0)
        assert got.shape == want.shape and np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
        t = tris
        degenerate = ((t[:, 0] == t[:, 1]).all(1) | (t[:, 1] == t[:, 2]).all(1) | (t[:, 0] == t[:, 2]).all(1))
        assert int(degenerate.sum()) == 0


def test_sphere_area_and_uniform_samples():
    from shapeclipper_amd.utils import eval_3D
    S, r = 65, 0.45
    level = torch.tensor(_sphere(S, r)[None] * 0.6).cuda()               # grid spans [-0.6, 0.6]: radius 0.27
    pts, meshes = eval_3D.surface_points_device(level, -0.6, 0.6, 20000, seed=1)
    t = meshes[0]
    area = 0.5 * torch.linalg.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]).norm(dim=1).sum().item()
    scale = S / (S - 1)                                                  # undo the reference's 1/S vertex rescale
    assert abs(area * scale ** 2 / (4 * math.pi * (0.6 * r) ** 2) - 1) < 0.01
    p = (pts[0] + 0.6) * scale - 0.6
    rad = p.norm(dim=1)
    assert float((rad - 0.6 * r).abs().max()) < 2 * (1.2 / (S - 1)) ** 2 / (0.6 * r) + 1e-4
    # area-uniform: each octant of the sphere gets 1/8 of the samples (binomial 4-sigma)
    octant = ((p[:, 0] > 0).long() * 4 + (p[:, 1] > 0).long() * 2 + (p[:, 2] > 0).long()).bincount(minlength=8).float() / 20000
    assert float((octant - 0.125).abs().max()) < 4 * math.sqrt(0.125 * 0.875 / 20000)
    # z is uniform on a sphere (Archimedes): compare the lower-quartile mass
    assert abs(float((p[:, 2] < -0.5 * 0.6 * r).float().mean()) - 0.25) < 0.015


def test_empty_surface_gives_zero_points():
    from shapeclipper_amd.utils import eval_3D
    pts, meshes = eval_3D.surface_points_device(torch.ones(2, 9, 9, 9).cuda(), -0.6, 0.6, 100)
    assert meshes[0].shape[0] == 0 and float(pts.abs().max()) == 0.0


@pytest.mark.parametrize("method", ["cubes", "tetrahedra"])
def test_block_form_equals_per_cube_form(method):
    """ops.isosurface_triangles (one count per workgroup of 1,024 cubes, prefix inside the workgroup) against the per-cube entry points of
    the C ABI (a count and a 64-bit offset per cube): the same triangles in the same order, on a batch whose images differ, whose cube count
    is not a multiple of the block size, and whose first / last blocks of an image are empty."""
    import ctypes
    from shapeclipper_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.RandomState(4)
    S = 37                                                               # 36^3 = 46,656 cubes = 45.6 blocks per image
    level = np.stack([_sphere(S, 0.5, (0.1, 0.0, -0.2)), rng.randn(S, S, S).astype(np.float32), np.full((S, S, S), 2.0, np.float32),
                      _sphere(S, 0.8)]).astype(np.float32)
    lv = torch.tensor(level).cuda()
    tris, per = ops.isosurface_triangles(lv, 0.0, method=method)
    B, n_cubes = lv.shape[0], lv.shape[0] * (S - 1) ** 3
    cnt_fn, emit_fn = (lib.sc_marching_cubes_count, lib.sc_marching_cubes_emit) if method == "cubes" else (lib.sc_isosurface_count, lib.sc_isosurface_emit)
    counts = torch.empty(n_cubes, device="cuda", dtype=torch.int32)
    assert cnt_fn(_lib.ptr(lv), ctypes.c_int(B), ctypes.c_int(S), ctypes.c_float(0.0), _lib.ptr(counts), _lib.stream()) == 0
    ends = torch.cumsum(counts, 0, dtype=torch.int64)
    offsets = (ends - counts).contiguous()
    ref = torch.full((int(ends[-1]), 3, 3), float("nan"), device="cuda")
    assert emit_fn(_lib.ptr(lv), ctypes.c_int(B), ctypes.c_int(S), ctypes.c_float(0.0), _lib.ptr(counts), _lib.ptr(offsets), _lib.ptr(ref), _lib.stream()) == 0
    torch.cuda.synchronize()
    assert tris.shape == ref.shape and torch.equal(tris, ref)
    assert per.tolist() == counts.view(B, -1).sum(1).tolist() and per[2] == 0 and per[1] > per[0] > 0
    assert int(lib.sc_isosurface_blocks_per_image(ctypes.c_int(S))) == 46 and int(lib.sc_isosurface_blocks_per_image(ctypes.c_int(1))) == -1


def test_marching_cubes_block_pairs_agree_and_scan_equals_cumsum():
    """Round 6: (a) sc_isosurface_block_scan == exclusive cumsum of the block counts (+ total, + per-image sums) on a batch with empty and
    noisy images and on random counts whose length is no multiple of anything; (b) the mask-passing pair (count_masks / emit_masks: what
    ops.isosurface_triangles runs) and the recomputing pair (block_count / block_emit) of the C ABI write the same triangles in the same order."""
    import ctypes
    from shapeclipper_amd import _lib
    lib = _lib.load()
    c_int, c_f = ctypes.c_int, ctypes.c_float
    rng = np.random.RandomState(7)
    S = 41
    level = np.stack([np.full((S, S, S), 2.0, np.float32), _sphere(S, 0.5, (0.1, 0.0, -0.2)), rng.randn(S, S, S).astype(np.float32), _sphere(S, 0.8),
                      np.full((S, S, S), -1.0, np.float32)]).astype(np.float32)
    lv = torch.tensor(level).cuda()
    B, bpi, per = lv.shape[0], int(lib.sc_isosurface_blocks_per_image(c_int(S))), (S - 1) ** 3
    counts_a = torch.empty(B * bpi, device="cuda", dtype=torch.int32)
    counts_b = torch.empty_like(counts_a)
    masks = torch.full((B * per,), 255, device="cuda", dtype=torch.uint8)
    assert lib.sc_marching_cubes_block_count(_lib.ptr(lv), c_int(B), c_int(S), c_f(0.0), _lib.ptr(counts_a), _lib.stream()) == 0
    assert lib.sc_marching_cubes_block_count_masks(_lib.ptr(lv), c_int(B), c_int(S), c_f(0.0), _lib.ptr(counts_b), _lib.ptr(masks), _lib.stream()) == 0
    assert torch.equal(counts_a, counts_b) and int(counts_a.sum()) > 0
    # the case index of a cube: bit v set <=> corner v (x, y, z bits) below the iso value
    lvv = lv[:, :, :, :]
    want = torch.zeros(B, S - 1, S - 1, S - 1, dtype=torch.int32, device="cuda")
    for v in range(8):
        dx, dy, dz = v & 1, (v >> 1) & 1, (v >> 2) & 1
        want |= (lvv[:, dx:dx + S - 1, dy:dy + S - 1, dz:dz + S - 1] < 0.0).int() << v
    assert torch.equal(masks.int(), want.reshape(-1))
    offsets = torch.empty(B * bpi + 1, device="cuda", dtype=torch.int64)
    per_image = torch.empty(B, device="cuda", dtype=torch.int64)
    assert lib.sc_isosurface_block_scan(_lib.ptr(counts_a), c_int(B), c_int(S), _lib.ptr(offsets), _lib.ptr(per_image), _lib.stream()) == 0
    ends = torch.cumsum(counts_a, 0, dtype=torch.int64)
    assert torch.equal(offsets[:-1], ends - counts_a) and int(offsets[-1]) == int(ends[-1])
    assert per_image.tolist() == counts_a.view(B, bpi).sum(1).tolist() and per_image[0] == 0 and per_image[4] == 0
    total = int(ends[-1])
    t_a = torch.full((total, 3, 3), float("nan"), device="cuda")
    t_b = torch.full((total, 3, 3), float("nan"), device="cuda")
    assert lib.sc_marching_cubes_block_emit(_lib.ptr(lv), c_int(B), c_int(S), c_f(0.0), _lib.ptr(offsets), _lib.ptr(t_a), _lib.stream()) == 0
    assert lib.sc_marching_cubes_block_emit_masks(_lib.ptr(lv), c_int(B), c_int(S), c_f(0.0), _lib.ptr(offsets), _lib.ptr(masks), _lib.ptr(t_b), _lib.stream()) == 0
    torch.cuda.synchronize()
    assert not torch.isnan(t_a).any() and torch.equal(t_a, t_b)
    # the scan alone on lengths that exercise ragged segments (n = n_images * blocks_per_image(n_axis))
    for n_img, axis in ((1, 2), (3, 12), (7, 33), (32, 101)):
        nb = int(lib.sc_isosurface_blocks_per_image(c_int(axis)))
        c = torch.randint(0, 5000, (n_img * nb,), device="cuda", dtype=torch.int32)
        off = torch.empty(n_img * nb + 1, device="cuda", dtype=torch.int64)
        pim = torch.empty(n_img, device="cuda", dtype=torch.int64)
        assert lib.sc_isosurface_block_scan(_lib.ptr(c), c_int(n_img), c_int(axis), _lib.ptr(off), _lib.ptr(pim), _lib.stream()) == 0
        e = torch.cumsum(c, 0, dtype=torch.int64)
        assert torch.equal(off[:-1], e - c) and int(off[-1]) == int(e[-1]) and pim.tolist() == c.view(n_img, nb).sum(1).tolist()
