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
