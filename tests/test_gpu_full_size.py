"""Parity at BASELINE.json's full sizes, where the oracle cannot fuse the whole grid in seconds:
  * the two independently written HIP kernels (register-tiled and general) must agree bit for bit on the
    whole grid;
  * a random sample of voxels (plus the grid corners) is recomputed by the oracle (oracle_fuse_voxels) and must
    match bit for bit in fp64, or to one f32 rounding for the f32 grid;
  * size-independent properties: fusing the views in two launches equals one launch (f64), and per-map hit
    totals equal the column sums of per-voxel hits (a checksum of checksums).
"""
import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene
from oracle import oracle
from helpers import oracle_params_from_scene

pytestmark = pytest.mark.gpu
G = capi.VARIANT_FORCE_GENERAL


def _sample_ids(grid, n, seed):
    rng = np.random.default_rng(seed)
    nx, ny, nz = grid.cell_dims
    ids = rng.integers(0, grid.n_voxels, size=n)
    corners = [((k * ny) + j) * nx + i for k in (0, nz - 1) for j in (0, ny - 1) for i in (0, nx - 1)]
    return np.unique(np.concatenate([ids, np.array(corners)]))


def test_cfg2_256cubed_64_maps_vga_f64_grid():
    """BASELINE configs[1]: 256^3 x 64 maps of 640x480."""
    grid = scene.default_grid(256)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(64, 640, 480, seed=1000, dense=True, layout="sphere")
    with capi.FusionContext(grid, rp, count_hits=True) as ctx:
        ctx.add_views(views)
        assert ctx.info().tiled_kernel == 1
        ctx.fuse()
        tiled = ctx.download_grid()
        vh, mh = ctx.download_hits()
        # two launches onto the same f64 grid = one launch (cu:211 accumulates in view order)
        ctx.reset_grid()
        ctx.fuse(0, 40)
        ctx.fuse(40, 24)
        twice = ctx.download_grid()
        vh2, mh2 = ctx.download_hits()
    assert np.array_equal(tiled.view(np.uint64), twice.view(np.uint64))
    assert np.array_equal(vh, vh2) and np.array_equal(mh, mh2)
    assert int(mh.sum()) == int(vh.sum(dtype=np.uint64))            # checksum of checksums
    with capi.FusionContext(grid, rp, count_hits=True, kernel_variant=G) as ctx:
        ctx.add_views(views)
        assert ctx.info().tiled_kernel == 0
        ctx.fuse()
        general = ctx.download_grid()
        vh_g, mh_g = ctx.download_hits()
    assert np.array_equal(tiled.view(np.uint64), general.view(np.uint64))
    assert np.array_equal(vh, vh_g) and np.array_equal(mh, mh_g)
    ids = _sample_ids(grid, 20000, 1)
    want, hits = oracle.fuse_voxels(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, ids,
                                    n_threads=oracle.max_threads())
    assert np.array_equal(tiled.reshape(-1)[ids].view(np.uint64), want.view(np.uint64))
    assert np.array_equal(vh.reshape(-1)[ids], hits)
    assert hits.max() > 32 and np.abs(want).max() > 1.0


def test_cfg3_512cubed_256_maps_720p_f32_grid():
    """BASELINE configs[2] (the bench workload): 512^3 x 256 maps of 1280x720, f32 grid, f32 depth storage."""
    grid = scene.default_grid(512)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(256, 1280, 720, seed=1000, dense=True, layout="sphere", dtype=np.float32)
    out = {}
    for name, variant in (("tiled", 0), ("tiled_no_classes", capi.VARIANT_NO_BRICK_CLASSES), ("general", G)):
        with capi.FusionContext(grid, rp, grid_dtype="f32", kernel_variant=variant) as ctx:
            ctx.add_views(views)
            assert ctx.info().tiled_kernel == (1 if name.startswith("tiled") else 0)
            assert ctx.info().depth_storage_in_use == capi.DMI_DEPTH_F32
            ctx.fuse()
            out[name] = ctx.download_grid(np.float32)
    assert np.array_equal(out["tiled"].view(np.uint32), out["general"].view(np.uint32))
    assert np.array_equal(out["tiled"].view(np.uint32), out["tiled_no_classes"].view(np.uint32))
    ids = _sample_ids(grid, 4096, 2)
    want, _ = oracle.fuse_voxels(oracle_params_from_scene(grid, rp, views), views.depth.astype(np.float64), views.K4,
                                 views.RT4, ids, n_threads=oracle.max_threads())
    got = out["tiled"].reshape(-1)[ids]
    assert np.array_equal(got, want.astype(np.float32))            # exactly the f32 rounding of the f64 sum
    assert np.abs(want).max() > 1.0


def test_1024cubed_grid_indexing():
    """BASELINE configs[4] grid size (1024^3 voxels = 4.3 GB f32, 2^30 cells): 64-bit indexing of the grid, the class
    table and the brick order.  Fewer and smaller views than config 5 to keep the test short."""
    grid = scene.default_grid(1024)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(24, 640, 480, seed=1234, dense=True, layout="sphere", dtype=np.float32)
    out = {}
    for name, variant in (("tiled", 0), ("general", G)):
        with capi.FusionContext(grid, rp, grid_dtype="f32", kernel_variant=variant) as ctx:
            ctx.add_views(views)
            ctx.fuse()
            out[name] = ctx.download_grid(np.float32)
            if name == "tiled":
                hist = ctx.brick_class_histogram()
                assert sum(hist.values()) == 128 * 128 * 64 * views.n and hist["free"] > 0 and hist["mixed"] > 0
    assert np.array_equal(out["tiled"].view(np.uint32), out["general"].view(np.uint32))
    ids = _sample_ids(grid, 4096, 3)
    want, _ = oracle.fuse_voxels(oracle_params_from_scene(grid, rp, views), views.depth.astype(np.float64), views.K4,
                                 views.RT4, ids, n_threads=oracle.max_threads())
    assert np.array_equal(out["tiled"].reshape(-1)[ids], want.astype(np.float32))
    assert np.abs(want).max() > 0.5
