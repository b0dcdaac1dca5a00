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


@pytest.mark.parametrize("kind", ["speckle", "noisy"])
def test_cfg3_speckled_depth_tables(kind):
    """BASELINE configs[2] on the input the reference's filter really sees: best-cost values ~ U[0, 1) and the
    threshold that turns ~10 % of the pixels, scattered, into "no depth" -- applied by dmi_add_views as
    ApplyDepthThresholdFilter does (RD.cxx:138-167, cu:348); `noisy` adds one voxel of depth noise and holes.  Nearly every
    footprint then holds sentinels next to depths: the path through MIXED_FREE_OR_NODEPTH / the FREE column and the skipped
    far-behind pairs.  tiled == tiled without classes == general over the whole grid, 4096 oracle voxels."""
    from bench import upload_scene
    grid = scene.default_grid(512)
    rp = scene.default_ray_potential(grid)
    out = {}
    views = None
    for name, variant in (("tiled", 0), ("tiled_no_classes", capi.VARIANT_NO_BRICK_CLASSES), ("general", G)):
        with capi.FusionContext(grid, rp, grid_dtype="f32", kernel_variant=variant) as ctx:
            v = upload_scene(ctx, scene, kind, 256, 1280, 720, float(max(grid.spacing)), keep_host=(views is None))
            views = views or v
            assert ctx.info().tiled_kernel == (1 if name.startswith("tiled") else 0)
            ctx.fuse()
            out[name] = ctx.download_grid(np.float32)
            if name == "tiled":
                reasons = ctx.mixed_reason_histogram()
                hist = ctx.brick_class_histogram()
    assert np.array_equal(out["tiled"].view(np.uint32), out["general"].view(np.uint32))
    assert np.array_equal(out["tiled"].view(np.uint32), out["tiled_no_classes"].view(np.uint32))
    # the classes this scene is about: free space seen through holes, and far-behind pairs skipped whatever the holes
    pairs = sum(hist.values())   # (brick, view) pairs: 512^3 / (8 x 8 x 16-voxel columns -- the height picked for maps with holes) x 256
    assert pairs == 64 * 64 * 32 * 256
    assert reasons["free_or_no_depth"] > 0.3 * pairs and hist["free"] == 0 and hist["skip"] > 0.35 * pairs
    frac = float((views.depth == -1.0).mean())
    assert 0.09 < frac < 0.13, frac
    ids = _sample_ids(grid, 4096, 6)
    want, _ = oracle.fuse_voxels(oracle_params_from_scene(grid, rp, views), views.depth.astype(np.float64), views.K4,
                                 views.RT4, ids, n_threads=oracle.max_threads())
    assert np.array_equal(out["tiled"].reshape(-1)[ids], want.astype(np.float32))
    assert np.abs(want).max() > 1.0


def test_cfg3_room_scene():
    """A second geometry at BASELINE configs[2]'s size: the cameras stand INSIDE the grid and look outward at the walls of a room
    (scene.make_room_views) -- voxels behind every camera (cu:177), walls at grazing angles, depths over an order of magnitude,
    footprints of a brick from a few pixels to hundreds -- with the same 10 % speckle.  tiled == tiled without classes ==
    general over the whole grid, 4096 oracle voxels; the classes that this geometry is about really occur."""
    from bench import upload_scene
    grid = scene.default_grid(512)
    rp = scene.default_ray_potential(grid)
    out = {}
    views = None
    for name, variant in (("tiled", 0), ("tiled_no_classes", capi.VARIANT_NO_BRICK_CLASSES), ("general", G)):
        with capi.FusionContext(grid, rp, grid_dtype="f32", kernel_variant=variant) as ctx:
            v = upload_scene(ctx, scene, "room", 256, 1280, 720, float(max(grid.spacing)), keep_host=(views is None))
            views = views or v
            assert ctx.info().tiled_kernel == (1 if name.startswith("tiled") else 0)
            ctx.fuse()
            out[name] = ctx.download_grid(np.float32)
            if name == "tiled":
                reasons = ctx.mixed_reason_histogram()
                hist = ctx.brick_class_histogram()
    assert np.array_equal(out["tiled"].view(np.uint32), out["general"].view(np.uint32))
    assert np.array_equal(out["tiled"].view(np.uint32), out["tiled_no_classes"].view(np.uint32))
    pairs = sum(hist.values())
    assert pairs in (64 * 64 * 32 * 256, 64 * 64 * 64 * 256)
    # every camera has bricks behind it and bricks its image plane cuts; most of the room is free space seen through holes
    assert hist["skip"] > 0.3 * pairs and reasons["camera_plane"] > 0 and reasons["free_or_no_depth"] > 0, (hist, reasons)
    frac = float((views.depth == -1.0).mean())
    assert 0.09 < frac < 0.13, frac
    ids = _sample_ids(grid, 4096, 8)
    want, _ = oracle.fuse_voxels(oracle_params_from_scene(grid, rp, views), views.depth.astype(np.float64), views.K4,
                                 views.RT4, ids, n_threads=oracle.max_threads())
    assert np.array_equal(out["tiled"].reshape(-1)[ids], want.astype(np.float32))
    assert np.abs(want).max() > 1.0


def test_wide_depth_maps_leave_tier_one():
    """Depth maps so large that W * py + px is no longer exact in fp32 ((H + 2) * W >= 2^24): the tiled kernel's packed-fp32
    pixel selection (tier 1) declines the view on the host and every pixel comes from the fp64 tier; also a view whose
    camera sits inside the grid (no positive lower bound of c.z: tier 1 declines as well).  Bit for bit against the
    general kernel and oracle samples."""
    grid = scene.default_grid((96, 80, 72))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(3, 4608, 3648, seed=31, dense=True, layout="sphere", dtype=np.float32)
    views.RT4[2] = scene.look_at_rt(np.array([0.2, -0.1, 0.3]), target=(1.0, 0.5, -0.2))   # inside the grid
    rng = np.random.default_rng(5)
    views.depth[rng.random(views.depth.shape) < 0.1] = -1.0
    want = _tiled_general_and_oracle_sample(grid, rp, views, 4096, 7)
    assert np.abs(want).max() > 0.1


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


def _tiled_general_and_oracle_sample(grid, rp, views, n_sample, seed):
    """tiled == general over the whole f32 grid, and n_sample oracle spot voxels (exactly the f32 rounding of the f64 sum)."""
    out = {}
    for name, variant in (("tiled", 0), ("general", G)):
        with capi.FusionContext(grid, rp, grid_dtype="f32", kernel_variant=variant) as ctx:
            ctx.add_views(views)
            assert ctx.info().tiled_kernel == (1 if name == "tiled" else 0)
            ctx.fuse()
            out[name] = ctx.download_grid(np.float32)
    assert np.array_equal(out["tiled"].view(np.uint32), out["general"].view(np.uint32))
    ids = _sample_ids(grid, n_sample, seed)
    want, _ = oracle.fuse_voxels(oracle_params_from_scene(grid, rp, views), views.depth.astype(np.float64), views.K4,
                                 views.RT4, ids, n_threads=oracle.max_threads())
    assert np.array_equal(out["tiled"].reshape(-1)[ids], want.astype(np.float32))
    return want


def test_cfg4_share_512cubed_128_maps_vga():
    """BASELINE configs[3], one GPU's share of the 8-way split: 512^3 x 128 of the 1024 views of 640x480 (the views
    rank 0 takes, dmi_multi_view_shard) -- the very views bench.py's strong-scaling run of cfg4 gives that rank."""
    grid = scene.default_grid(512)
    rp = scene.default_ray_potential(grid)
    lo, hi = capi.multi_view_shard(1024, 0, 8)
    assert (lo, hi) == (0, 128)
    views = scene.make_views(1024, 640, 480, seed=1004, dense=True, layout="sphere", dtype=np.float32, view_range=(lo, hi))
    want = _tiled_general_and_oracle_sample(grid, rp, views, 4096, 4)
    assert np.abs(want).max() > 1.0


def test_cfg5_share_1024cubed_64_maps_1080p():
    """BASELINE configs[4], one GPU's share: 1024^3 voxels x 64 of the 512 views of 1920x1080."""
    grid = scene.default_grid(1024)
    rp = scene.default_ray_potential(grid)
    lo, hi = capi.multi_view_shard(512, 3, 8)
    views = scene.make_views(512, 1920, 1080, seed=1005, dense=True, layout="sphere", dtype=np.float32, view_range=(lo, hi))
    want = _tiled_general_and_oracle_sample(grid, rp, views, 4096, 5)
    assert np.abs(want).max() > 0.5


def test_grid_transfers_convert_on_the_device():
    """dmi_upload_grid into an f32 grid and dmi_download_grid_* of the other type convert on the device, chunk by chunk
    (32 Mi elements per chunk: 320^3 needs two chunks); values are exactly the host casts they replace."""
    grid = scene.default_grid((320, 320, 330))
    rp = scene.default_ray_potential(grid)
    rng = np.random.default_rng(8)
    init = rng.standard_normal(grid.n_voxels) * 3.0
    with capi.FusionContext(grid, rp, grid_dtype="f32") as ctx:
        ctx.upload_grid(init)
        as32 = ctx.download_grid(np.float32).reshape(-1)
        as64 = ctx.download_grid(np.float64).reshape(-1)
    assert np.array_equal(as32, init.astype(np.float32)) and np.array_equal(as64, init.astype(np.float32).astype(np.float64))
    with capi.FusionContext(grid, rp, grid_dtype="f64") as ctx:
        ctx.upload_grid(init)
        assert np.array_equal(ctx.download_grid(np.float32).reshape(-1), init.astype(np.float32))
        assert np.array_equal(ctx.download_grid(np.float64).reshape(-1), init)
