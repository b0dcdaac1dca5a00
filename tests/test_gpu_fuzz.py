"""Randomised GPU parity: random axis-aligned and rotated grids, pinhole / skewed / general K (also mixed view by view,
with a view no fast path takes), random poses (cameras
inside and outside the grid), depth tables with sentinels, NaNs and f32-inexact values, ray-potential parameters
including the degenerate ones the reference accepts (thickness 0, delta 0, negative eta).  Every case runs the default
path (tiled kernel with brick classes when eligible), the tiled kernel without classes and the general kernel, and
all three must equal the oracle bit for bit (fp64 grid, hit counters)."""
import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene
from oracle import oracle
from helpers import bits_equal, oracle_params_from_scene

pytestmark = pytest.mark.gpu


def _random_case(seed, medium=False, many_views=False):
    rng = np.random.default_rng(seed)
    dims = tuple(int(v) for v in (rng.integers(140, 215, size=3) if medium else rng.integers(5, 45, size=3)))
    extent = rng.uniform(0.5, 3.0, size=3)
    origin = tuple(-extent / 2 + rng.uniform(-0.3, 0.3, size=3))
    spacing = tuple(extent / np.array(dims))
    gm = np.eye(4)
    kind = rng.integers(0, 4)
    if kind == 1:                                   # scaled / flipped / translated axes: still axis-aligned
        gm[:3, :3] = np.diag(rng.choice([-1.5, -1.0, 0.5, 1.0, 2.0], size=3))
        gm[:3, 3] = rng.uniform(-0.5, 0.5, size=3)
    elif kind == 2:                                 # rotated grid (about z, or any orthonormal axes): the tiled
        a = rng.uniform(-1, 1)                      # kernel's rotated path
        gm[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        if rng.integers(0, 2):
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            gm[:3, :3] = q
            gm[:3, 3] = rng.uniform(-0.2, 0.2, size=3)
    grid = scene.GridDesc(dims, origin, spacing, gm)
    n = int(rng.integers(9, 20)) if medium else int(rng.integers(1, 9))
    W, H = (int(rng.integers(200, 400)), int(rng.integers(150, 300))) if medium else (int(rng.integers(8, 90)), int(rng.integers(6, 70)))
    if many_views:  # launches of 96 views and more run persistent workgroups (fusion_tile.hip: kPersistentMinViews)
        n, W, H = int(rng.integers(96, 130)), int(rng.integers(60, 120)), int(rng.integers(40, 90))
    radius = float(rng.choice([0.3, 1.2, 3.0, 6.0]))
    views = scene.make_views(n, W, H, seed=int(rng.integers(1 << 30)), dense=bool(rng.integers(0, 2)), radius=radius,
                             focal_scale=float(rng.uniform(0.4, 1.5)))
    kk = rng.integers(0, 7)
    if kk == 1:
        views.K4[:, 0, 1] = rng.uniform(-0.5, 0.5)  # skew
    elif kk == 2:
        views.K4[:, 2, 0] = 1e-3                    # general K (h.z != c.z)
    elif kk == 3:                                   # every kind in one fusion, view by view: runs of tiled (pinhole or
        for m in range(n):                          # GENK) launches and, for the huge focal, the general kernel
            what = rng.integers(0, 5)
            if what == 1:
                views.K4[m, 2, 2] = rng.uniform(0.5, 2.0)
            elif what == 2:
                views.K4[m, 2, :] = [rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), rng.uniform(0.8, 1.2), rng.uniform(-0.3, 0.3)]
            elif what == 3:
                views.K4[m, 0, 3], views.K4[m, 1, 3], views.K4[m, 1, 0] = rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-0.2, 0.2)
            elif what == 4:
                views.K4[m, 0, 0] = 1e14
    elif kk == 4:                                   # scaled third row and a fourth column on every view
        views.K4[:, 2, 2] = rng.uniform(0.5, 2.0)
        views.K4[:, 0, 3] = rng.uniform(-3, 3)
    depth = views.depth
    m = rng.random(depth.shape)
    depth[m < float(rng.choice([0.05, 0.1, 0.2, 0.3]))] = -1.0   # "no depth" speckle (a best-cost threshold's work, RD.cxx:138-167)
    if rng.integers(0, 3) == 0:
        depth[m > 0.995] = np.nan
    if rng.integers(0, 2) == 0:                     # values that are not f32-representable -> f64 storage
        depth[:] = np.where(np.isfinite(depth) & (depth != -1.0), depth * (1 + 2.0 ** -45), depth)
    s = float(max(spacing))
    thick = float(rng.choice([0.0, 0.5 * s, 2.5 * s]))
    delta = float(rng.choice([0.0, thick, 4 * s, 10 * s]))
    rho = float(rng.choice([0.8, -0.5, 1.0]))
    eta = float(rng.choice([0.03, 0.0, -0.2, 1.0]))
    if rho == 0 and thick == 0:
        rho = 0.8
    return grid, scene.RayPotential(thick, rho, eta, delta), views


@pytest.mark.parametrize("seed", range(40))
def test_random_scenes_bit_exact(seed):
    grid, rp, views = _random_case(seed)
    init = None
    if seed % 5 == 0:
        init = np.random.default_rng(seed).normal(size=(grid.cell_dims[2], grid.cell_dims[1], grid.cell_dims[0]))
    with np.errstate(all="ignore"):
        want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                       init_grid=init, n_threads=oracle.max_threads())
    for variant in (0, capi.VARIANT_FIXED_TILE_SHAPE, capi.VARIANT_NO_BRICK_CLASSES, 96, capi.VARIANT_FORCE_GENERAL,
                    capi.VARIANT_NO_INTERIOR | capi.VARIANT_XCD_RUNS, capi.VARIANT_ZMAJOR_SLOTS,
                    capi.VARIANT_PERSISTENT_ALWAYS, capi.VARIANT_PERSISTENT_NEVER, capi.VARIANT_COST_ORDER):
        out, vh, mh = capi.fuse_once(grid, rp, views, init_grid=init, kernel_variant=variant)
        assert np.array_equal(mh, mh_w), (seed, variant)
        assert np.array_equal(vh, vh_w), (seed, variant)
        assert bits_equal(out, want), (seed, variant)
    # the shipped default (no brick classes on grids this small: every pair through the column with every test), with and
    # without hit counters
    from helpers import shipped_defaults
    with shipped_defaults():
        for count_hits in (True, False):
            out, vh, mh = capi.fuse_once(grid, rp, views, init_grid=init, count_hits=count_hits, kernel_variant=0)
            assert bits_equal(out, want), (seed, "shipped", count_hits)
            if count_hits:
                assert np.array_equal(mh, mh_w) and np.array_equal(vh, vh_w), (seed, "shipped")


@pytest.mark.parametrize("seed", range(8))
def test_random_medium_scenes_bit_exact(seed):
    """The same generator at 140..215 cells per axis (6 000 to 19 000 bricks: more than the 5120 persistent workgroups the
    chip holds, so every workgroup fuses several bricks, runs out of its XCD's share and helps the others) and 9..19 views
    of a few hundred pixels: the whole grid against the oracle, bit for bit, default path and the path without classes."""
    grid, rp, views = _random_case(1000 + seed, medium=True)
    with np.errstate(all="ignore"):
        want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                       n_threads=oracle.max_threads())
    for variant, count_hits in ((0, False), (0, True), (capi.VARIANT_NO_BRICK_CLASSES, False), (capi.VARIANT_ZMAJOR_SLOTS | capi.VARIANT_SPATIAL_ORDER, False),
                                (capi.VARIANT_PERSISTENT_ALWAYS, False), (capi.VARIANT_PERSISTENT_ALWAYS, True),
                                (capi.VARIANT_WINDOWS_ALWAYS, False), (capi.VARIANT_WINDOWS_ALWAYS | capi.VARIANT_FIXED_TILE_SHAPE, False)):
        out, vh, mh = capi.fuse_once(grid, rp, views, count_hits=count_hits, kernel_variant=variant)
        assert bits_equal(out, want), (seed, variant, count_hits)
        if count_hits:
            assert np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w), (seed, variant)


@pytest.mark.parametrize("seed", range(6))
def test_random_many_view_scenes_bit_exact(seed):
    """Medium-size grids with 96..129 small views: the persistent form of the kernel (launches of fewer views run one
    workgroup per brick), more bricks than workgroups, every kind of K and grid the generator knows: the whole grid against
    the oracle, bit for bit."""
    grid, rp, views = _random_case(2000 + seed, medium=True, many_views=True)
    with np.errstate(all="ignore"):
        want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                       n_threads=oracle.max_threads())
    for variant, count_hits in ((0, False), (0, True), (capi.VARIANT_ZMAJOR_SLOTS | capi.VARIANT_SPATIAL_ORDER, False),
                                (capi.VARIANT_PERSISTENT_NEVER, False), (capi.VARIANT_WINDOWS_ALWAYS, False),
                                (capi.VARIANT_WINDOWS_ALWAYS | capi.VARIANT_PERSISTENT_NEVER, False)):
        out, vh, mh = capi.fuse_once(grid, rp, views, count_hits=count_hits, kernel_variant=variant)
        assert bits_equal(out, want), (seed, variant, count_hits)
        if count_hits:
            assert np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w), (seed, variant)
