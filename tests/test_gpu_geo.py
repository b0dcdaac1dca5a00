"""Real-world coordinate magnitudes (round 5): every other scene of this suite lives in [-3, 3]^3 with cameras at radius <= 6 and
focal lengths of 0.4 .. 1.5 image widths.  Geo-referenced SfM output does not: eastings / northings of 1e5 .. 1e7 in the grid's
origin and in every camera's translation (which then cancel seven digits in c = R w + T, Sources/Helper.h:134-165), focal lengths
of thousands of pixels, depths of tens to thousands of metres.  The family below moves random scenes into such frames
(scene.to_world_frame) and asks for the oracle's bits on every kernel path; one case carries the parameter magnitudes of the
reference's own example command line (Reconstruction/main.cxx:102) verbatim."""
import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene
from oracle import oracle
from helpers import bits_equal, oracle_params_from_scene

pytestmark = pytest.mark.gpu

W = capi.VARIANT_WINDOWS_ALWAYS
PATHS = (0, capi.VARIANT_FIXED_TILE_SHAPE, capi.VARIANT_NO_BRICK_CLASSES, capi.VARIANT_FORCE_GENERAL, W, W | capi.VARIANT_FIXED_TILE_SHAPE,
         capi.VARIANT_NO_WINDOWS, capi.VARIANT_PERSISTENT_ALWAYS)


def _geo_case(seed):
    rng = np.random.default_rng([seed, 77])
    dims = tuple(int(v) for v in rng.integers(20, 72, size=3))
    rotated = bool(rng.integers(0, 3) == 0)
    grid = scene.default_grid(dims, rotated=rotated)
    rp = scene.default_ray_potential(grid)
    n = int(rng.integers(3, 9))
    Wd, Hd = int(rng.integers(60, 200)), int(rng.integers(40, 150))
    radius = float(rng.choice([1.2, 3.0, 6.0]))
    views = scene.make_views(n, Wd, Hd, seed=int(rng.integers(1 << 30)), dense=bool(rng.integers(0, 2)), radius=radius,
                             focal_scale=float(rng.uniform(0.5, 1.4)))
    holes = rng.random(views.depth.shape) < float(rng.choice([0.0, 0.05, 0.1]))
    views.depth[holes] = -1.0
    # grid spacing 0.007 .. 0.04 m, depths 10 .. 2000 m, offsets 1e2 .. 1e6, focal 500 .. 8000 px
    spacing = float(rng.uniform(0.007, 0.04))
    scale = spacing / float(max(grid.spacing))
    if rng.integers(0, 2):
        scale = float(rng.uniform(10.0, 2000.0)) / radius  # ... or the depths decide (then the voxels are large)
    mag = 10.0 ** rng.uniform(2.0, 6.0, size=3)
    offset = mag * rng.choice([-1.0, 1.0], size=3)
    focal = float(rng.uniform(500.0, 8000.0)) if rng.integers(0, 2) else None
    return scene.to_world_frame(grid, rp, views, scale, offset, focal=focal) + (offset, scale)


@pytest.mark.parametrize("seed", range(24))
def test_geo_referenced_scenes_bit_exact(seed):
    grid, rp, views, offset, scale = _geo_case(seed)
    with np.errstate(all="ignore"):
        want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                       n_threads=oracle.max_threads())
    assert int(mh_w.sum()) > 0, "the scene must project into the maps"
    for variant in PATHS:
        for count_hits in ((True, False) if variant in (0, capi.VARIANT_FORCE_GENERAL) else (False,)):
            out, vh, mh = capi.fuse_once(grid, rp, views, count_hits=count_hits, kernel_variant=variant)
            assert bits_equal(out, want), (seed, variant, count_hits, offset, scale)
            if count_hits:
                assert np.array_equal(mh, mh_w) and np.array_equal(vh, vh_w), (seed, variant)


def test_reference_example_command_line_magnitudes():
    """--gridDims 100 100 100 --gridSpacing 0.0348 0.0391 0.0342 --gridOrigin -2.29 -2.24 -2.2 --rayThick 0.08 --rayRho 0.8 --rayEta
    0.03 --rayDelta 0.3 --threshBestCost 0.3 (Reconstruction/main.cxx:102): that grid, those ray-potential parameters and that
    threshold, around a sphere scene scaled to fill it."""
    grid = scene.GridDesc((100, 100, 100), (-2.29, -2.24, -2.2), (0.0348, 0.0391, 0.0342))
    rp = scene.RayPotential(thickness=0.08, rho=0.8, eta=0.03, delta=0.3)
    base = scene.make_views(8, 160, 120, seed=5, dense=True, with_best_cost=True)
    # the unit scene (sphere of radius 0.6 at the origin, cameras at radius 3) twice as large, centred in the grid
    centre = np.array(grid.origin) + 50 * np.array(grid.spacing)
    _, _, views = scene.to_world_frame(scene.default_grid(8), scene.default_ray_potential(scene.default_grid(8)), base, 2.0, centre)
    thr = 0.3
    d = oracle.apply_depth_threshold(views.depth, views.best_cost, thr).reshape(views.depth.shape)
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), d, views.K4, views.RT4, n_threads=oracle.max_threads())
    assert int(mh_w.sum()) > 100000
    for variant in PATHS:
        out, vh, mh = capi.fuse_once(grid, rp, views, threshold=thr, kernel_variant=variant)
        assert bits_equal(out, want), variant
        assert np.array_equal(mh, mh_w) and np.array_equal(vh, vh_w), variant


def test_view_paths_follow_the_magnitudes():
    """dmi_get_view_paths: a centred scene and the same scene at offsets of 1e6 run tier 1 with window records; at 1e9 (or with a
    focal length of 1e14) the per-view bound fails and the general kernel takes the view; a general K is counted as such."""
    grid = scene.default_grid((48, 40, 32))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(6, 160, 120, seed=3, dense=True)
    for off, expect_general in ((0.0, 0), (1e6, 0), (1e9, 6)):
        g, r, v = scene.to_world_frame(grid, rp, views, 10.0, (off, -off, 0.5 * off))
        with capi.FusionContext(g, r) as ctx:
            ctx.add_views(v)
            p = ctx.view_paths()
        assert p["general_kernel"] == expect_general, (off, p)
        if not expect_general:
            assert p["tiled_tier1"] + p["tiled_tier1_per_lane_margin"] == 6 and p["with_window_record"] == 6, (off, p)
    v2 = scene.Views(views.depth, views.K4.copy(), views.RT4)
    v2.K4[0, 0, 0] = 1e14      # an absurd focal length: the bound cannot hold
    v2.K4[1, 2, 0] = 1e-3      # a general third row
    with capi.FusionContext(grid, rp) as ctx:
        ctx.add_views(v2)
        p = ctx.view_paths()
    assert p["general_kernel"] == 1 and p["tiled_general_k"] == 1 and p["tiled_tier1"] == 4, p


def test_regional_holes_bit_exact():
    """The `blobs` scene kind (holes in discs, as best-cost filtering leaves them) at a size with more bricks than persistent
    workgroups, default launch and windows forced: the oracle's grid bit for bit."""
    grid = scene.default_grid((160, 144, 128))
    rp = scene.default_ray_potential(grid)
    views, thr = scene.make_scene_views("blobs", 12, 320, 240, seed=9, speckle=0.2)
    assert thr is None and float((views.depth == -1.0).mean()) > 0.1
    want = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())[0]
    for variant in (0, W, W | capi.VARIANT_FIXED_TILE_SHAPE):
        out, _, _ = capi.fuse_once(grid, rp, views, count_hits=False, kernel_variant=variant)
        assert bits_equal(out, want), variant


@pytest.mark.gpu
def test_the_launch_follows_the_hole_layout():
    """What the maps' holes look like decides the launch (dmi_capi.hip, fuse_run): scattered holes bring the window column from
    about 0.1 % of the pixels on; holes in regions only when a twenty-fifth of the 8-pixel strips lie on a region's border (discs
    over 40 % of the image, not over 5 %).  Whatever the rule picks, the oracle's grid bit for bit."""
    grid = scene.default_grid((96, 96, 64))
    rp = scene.default_ray_potential(grid)

    def run(views):
        want = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())[0]
        with capi.FusionContext(grid, rp) as ctx:
            ctx.add_views(views)
            ctx.fuse()
            out = ctx.download_grid()
            n_win = ctx.window_pair_count()
        assert bits_equal(out, want)
        return n_win

    discs_few, _ = scene.make_scene_views("blobs", 10, 640, 480, seed=3, speckle=0.05)
    discs_many, _ = scene.make_scene_views("blobs", 10, 640, 480, seed=3, speckle=0.4)
    assert run(discs_few) == 0
    assert run(discs_many) > 0
    dense = scene.make_views(10, 640, 480, seed=3, dense=True)
    for share, windows in ((0.0002, False), (0.004, True)):
        d = dense.depth.copy()
        d[np.random.default_rng(5).random(d.shape) < share] = -1.0
        n_win = run(scene.Views(d, dense.K4, dense.RT4))
        assert (n_win > 0) == windows, (share, n_win)
