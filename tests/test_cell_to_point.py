"""Cell data -> point data of the fused grid (Reconstruction/main.cxx:151-155).
CPU: the C oracle against an independently written numpy restatement and known answers.  GPU: dmi_cell_to_point
through the C ABI, bit-exact against the oracle."""
import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene
from oracle import oracle, oracle_np
from helpers import bits_equal


def _cells(shape, seed):
    rng = np.random.default_rng(seed)
    c = rng.normal(size=shape) * 10.0 ** rng.integers(-3, 4, size=shape)
    c[rng.random(shape) < 0.2] = 0.0
    return c


def test_known_answers():
    # one cell: all eight points carry its value (count 1, w = 1)
    p = oracle.cell_to_point(np.full((1, 1, 1), 3.25))
    assert p.shape == (2, 2, 2) and np.all(p == 3.25)
    # constant grid of a short-mantissa value: every partial sum m * (v/8) is exact, every point equals the constant
    p = oracle.cell_to_point(np.full((3, 4, 5), -0.75))
    assert np.all(p == -0.75)
    # 2 x 1 x 1 cells a, b along x: the points at x = 0 / 2 carry a / b, the points at x = 1 their mean
    p = oracle.cell_to_point(np.array([[[1.0, 4.0]]]))
    assert np.all(p[:, :, 0] == 1.0) and np.all(p[:, :, 2] == 4.0) and np.all(p[:, :, 1] == 2.5)
    # interior point = plain mean of its eight cells
    c = np.arange(27, dtype=np.float64).reshape(3, 3, 3)
    p = oracle.cell_to_point(c)
    assert p[1, 1, 1] == c[0:2, 0:2, 0:2].mean() and p[2, 2, 2] == c[1:3, 1:3, 1:3].mean()
    assert p[0, 0, 0] == c[0, 0, 0] and p[3, 3, 3] == c[2, 2, 2]


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 1, 7), (2, 3, 1), (5, 4, 6), (9, 17, 33)])
def test_c_oracle_matches_numpy_restatement(shape):
    c = _cells(shape, seed=sum(shape))
    assert bits_equal(oracle.cell_to_point(c), oracle_np.cell_to_point_np(c))


def test_addition_order_bound():
    """The restated order of additions is the one thing a different VTK version could change: any order of
    the same (exact) terms lies within 3 ulp of the largest partial sum.  Stated here so that the bit-exact GPU
    comparison below is read together with it."""
    c = _cells((6, 7, 8), seed=5)
    p = oracle.cell_to_point(c)
    interior_mean = sum(c[dz:dz + 5, dy:dy + 6, dx:dx + 7] * 0.125 for dz in (1, 0) for dy in (0, 1) for dx in (1, 0))
    mag = sum(np.abs(c[dz:dz + 5, dy:dy + 6, dx:dx + 7]) * 0.125 for dz in (0, 1) for dy in (0, 1) for dx in (0, 1))
    assert np.all(np.abs(p[1:6, 1:7, 1:8] - interior_mean) <= 7 * 2.0 ** -53 * mag)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,dtype", [((1, 1, 1), "f64"), ((70, 33, 17), "f64"), ((64, 64, 64), "f32"),
                                          ((130, 5, 40), "f64"), ((63, 65, 31), "f32")])
def test_gpu_cell_to_point_is_bit_identical(cells, dtype):
    grid = scene.default_grid(cells)
    ray = scene.default_ray_potential(grid)
    nx, ny, nz = cells
    c = _cells((nz, ny, nx), seed=nx + ny + nz)
    if dtype == "f32":
        c = c.astype(np.float32).astype(np.float64)  # values an f32 grid holds exactly
    with capi.FusionContext(grid, ray, grid_dtype=dtype) as ctx:
        ctx.upload_grid(c)
        got = ctx.download_point_data()
        assert got.shape == (nz + 1, ny + 1, nx + 1)
        assert bits_equal(got, oracle.cell_to_point(c))
        assert ctx.timings().last_cell_to_point_ms > 0
        # the cached result follows the grid: reset -> zeros
        ctx.reset_grid()
        assert not ctx.download_point_data().any()


@pytest.mark.gpu
def test_gpu_cell_to_point_after_fuse():
    grid = scene.default_grid((48, 40, 36))
    ray = scene.default_ray_potential(grid)
    views = scene.make_views(5, 96, 72, seed=11, dense=True)
    with capi.FusionContext(grid, ray) as ctx:
        ctx.add_views(views)
        ctx.fuse()
        cells = ctx.download_grid()
        pts = ctx.download_point_data()
        assert np.count_nonzero(cells) > 1000
        assert bits_equal(pts, oracle.cell_to_point(cells))
        ctx.fuse()  # accumulate again: the point data must be recomputed
        assert bits_equal(ctx.download_point_data(), oracle.cell_to_point(ctx.download_grid()))


def _post_golden(name):
    import os
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post", name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def test_oracles_match_committed_cell_to_point_fixture():
    g = _post_golden("cell_to_point")
    assert bits_equal(oracle.cell_to_point(g["cells"]), g["expected_points"])
    assert bits_equal(oracle_np.cell_to_point_np(g["cells"]), g["expected_points"])
    # the corner point above the lone -0.0 cell: c = 0; c += 1 * (-0.0) gives +0.0, as VTK's accumulation does
    assert g["expected_points"][0, 0, 0] == 0 and not np.signbit(g["expected_points"][0, 0, 0])


@pytest.mark.gpu
def test_gpu_matches_committed_cell_to_point_fixture():
    g = _post_golden("cell_to_point")
    nz, ny, nx = g["cells"].shape
    grid = scene.default_grid((nx, ny, nz))
    with capi.FusionContext(grid, scene.default_ray_potential(grid)) as ctx:
        ctx.upload_grid(g["cells"])
        assert bits_equal(ctx.download_point_data(), g["expected_points"])


# ---- iso-value pre-pass (second half; Reconstruction/main.cxx:169-173) ----------------------------
def test_iso_active_cells_known_answers():
    # one cell, corners 0..7: active for an iso-value inside (0, 7], not for one at or below the minimum / above the maximum
    p = np.arange(8, dtype=np.float64).reshape(2, 2, 2)
    assert oracle.iso_active_cells(p, 3.5).tolist() == [0]
    assert oracle.iso_active_cells(p, 7.0).tolist() == [0]          # the corner equal to the iso-value is inside (>=)
    assert oracle.iso_active_cells(p, 0.0).tolist() == []           # every corner >= iso: case 255
    assert oracle.iso_active_cells(p, 7.5).tolist() == []           # no corner >= iso: case 0
    # a plane x = 1.5 through a 3 x 2 x 2 cell grid (point x coordinates 0..3): only the middle column of cells is cut
    pts = np.broadcast_to(np.arange(4, dtype=np.float64), (3, 3, 4)).copy()
    assert oracle.iso_active_cells(pts, 1.5).tolist() == [1, 4, 7, 10]
    # a NaN corner is outside: a cell of seven 1s and one NaN is active at iso 0.5, a cell of NaNs is not
    q = np.ones((2, 2, 2))
    q[1, 1, 1] = np.nan
    assert oracle.iso_active_cells(q, 0.5).tolist() == [0]
    assert oracle.iso_active_cells(np.full((2, 2, 2), np.nan), 0.5).tolist() == []


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 3, 1), (5, 4, 6), (9, 17, 33)])
def test_iso_c_oracle_matches_numpy_restatement(shape):
    pts = oracle.cell_to_point(_cells(shape, seed=sum(shape) + 1))
    for iso in (0.0, 1.0, -3.0, float(np.median(pts))):
        assert np.array_equal(oracle.iso_active_cells(pts, iso), oracle_np.iso_active_cells_np(pts, iso))


def test_oracles_match_committed_iso_fixture():
    g = _post_golden("iso_cells")
    assert np.array_equal(oracle.iso_active_cells(g["points"], float(g["iso"])), g["expected_ids"])
    assert np.array_equal(oracle_np.iso_active_cells_np(g["points"], float(g["iso"])), g["expected_ids"])


@pytest.mark.gpu
@pytest.mark.parametrize("cells,dtype", [((1, 1, 1), "f64"), ((70, 33, 17), "f64"), ((300, 5, 40), "f32"), ((257, 3, 2), "f64")])
def test_gpu_iso_active_cells(cells, dtype):
    """dmi_iso_active_cells against the oracle: count and the ascending id list, several iso-values (the CLI's default
    contour 1.0 among them), a partial list when the caller's buffer is short, and rows longer than one block of 256."""
    grid = scene.default_grid(cells)
    nx, ny, nz = cells
    c = _cells((nz, ny, nx), seed=nx * 3 + ny + nz)
    if dtype == "f32":
        c = c.astype(np.float32).astype(np.float64)
    with capi.FusionContext(grid, scene.default_ray_potential(grid), grid_dtype=dtype) as ctx:
        ctx.upload_grid(c)
        pts = ctx.download_point_data()
        for iso in (1.0, 0.0, -0.5, 1e9):
            want = oracle.iso_active_cells(pts, iso)
            n, ids = ctx.iso_active_cells(iso)
            assert n == want.size and np.array_equal(ids, want), iso
            assert ctx.iso_active_cells(iso, ids=False) == (want.size, None)
        want = oracle.iso_active_cells(pts, 0.0)
        if want.size > 3:   # a short buffer receives the first ids and the full count
            lib = capi.load()
            import ctypes
            n = ctypes.c_uint64(0)
            short = np.full(3, -1, dtype=np.int64)
            rc = lib.dmi_iso_active_cells(ctx._h, 0.0, ctypes.byref(n), short.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), 3)
            assert rc == 0 and n.value == want.size and np.array_equal(short, want[:3])
        ctx.reset_grid()    # follows the grid like the point data: all zeros -> nothing straddles 1.0, nor 0.0 (all inside)
        assert ctx.iso_active_cells(1.0)[0] == 0 and ctx.iso_active_cells(0.0)[0] == 0


@pytest.mark.gpu
def test_gpu_iso_active_cells_after_fuse_and_fixture():
    g = _post_golden("iso_cells")
    grid = scene.default_grid((48, 40, 36))
    ray = scene.default_ray_potential(grid)
    views = scene.make_views(5, 96, 72, seed=11, dense=True)
    with capi.FusionContext(grid, ray) as ctx:
        ctx.add_views(views)
        ctx.fuse()
        pts = ctx.download_point_data()
        n, ids = ctx.iso_active_cells(1.0)    # Reconstruction/main.cxx:80: the default contour value
        want = oracle.iso_active_cells(pts, 1.0)
        assert n == want.size and np.array_equal(ids, want) and 100 < n < grid.n_voxels // 4
