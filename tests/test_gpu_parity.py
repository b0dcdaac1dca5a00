"""GPU parity tests: the HIP path, called through the C ABI, against the committed golden fixtures
and the CPU oracle.  Bar: TSDF bit-exact in fp64 on one GPU (f64 grid), hit counts bit-exact;
with an f32 grid |delta| <= 2^-24 |v| (one final rounding)."""
import contextlib

import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene
from oracle import oracle
from helpers import bits_equal, oracle_params_from_golden, oracle_params_from_scene

pytestmark = pytest.mark.gpu

G = capi.VARIANT_FORCE_GENERAL
# 0 = default: the register-tiled kernel when its preconditions hold, else the general kernel;
# 32/64/96 = other tile shapes; G | x = the general kernel with its own switches
NC = capi.VARIANT_NO_BRICK_CLASSES
# 0 = default: the register-tiled kernel with brick classes when its preconditions hold, else the general
# kernel; NC = tiled kernel, every pair on the per-voxel path; 32..224 = other tile shapes; G | x = the general
# kernel with its own switches
FX = capi.VARIANT_FIXED_TILE_SHAPE   # shape 0 (16-voxel columns) also on the small grids of these tests
CO = capi.VARIANT_COST_ORDER         # bricks ordered by their number of mixed views, dealt to the XCDs in short runs
VARIANTS = [0, FX, FX | NC, NC, capi.VARIANT_SPATIAL_ORDER, CO, CO | FX | capi.VARIANT_PERSISTENT_ALWAYS, 32, 64, 96 | NC, 128, 160 | NC, 192, 224, G, G | capi.VARIANT_EXACT_DIVISION, G | capi.VARIANT_GENERAL_K,
            G | capi.VARIANT_EXACT_DIVISION | capi.VARIANT_GENERAL_K, G | 4, G | 8, G | 12]
SO = capi.VARIANT_SPATIAL_ORDER
TILE_SHAPES = [0, FX, FX | NC, NC, SO, FX | SO, CO, CO | capi.VARIANT_WINDOWS_ALWAYS, 32, 64 | NC, 96, 96 | SO, 128, 160, 192 | NC, 224]


def _golden_inputs(g):
    grid = scene.GridDesc(tuple(int(c) for c in g["cell_dims"]), tuple(g["origin"]), tuple(g["spacing"]), g["grid_matrix"])
    t, rho, eta, delta = (float(x) for x in g["ray"])
    views = scene.Views(g["depth"], g["K4"], g["RT4"], g.get("best_cost"))
    thr = float(g["threshold"]) if "threshold" in g else None
    return grid, scene.RayPotential(t, rho, eta, delta), views, thr


@pytest.mark.parametrize("variant", VARIANTS)
def test_golden_bit_exact_f64(golden, variant):
    grid, rp, views, thr = _golden_inputs(golden)
    out, vh, mh = capi.fuse_once(grid, rp, views, threshold=thr, init_grid=golden.get("init_grid"),
                                 kernel_variant=variant)
    assert np.array_equal(mh, golden["expected_map_hits"])
    assert np.array_equal(vh, golden["expected_voxel_hits"])
    assert bits_equal(out, golden["expected_grid"])


@pytest.mark.parametrize("count_hits", [True, False])
def test_golden_bit_exact_under_shipped_defaults(golden, count_hits):
    """What a caller gets with kernel_variant 0: on grids of at most 1024 bricks no brick classes at all (the other tests of this
    file switch them on, conftest.py).  Rotated grids, general K, initial grids, border footprints: all the golden scenes."""
    from helpers import shipped_defaults
    grid, rp, views, thr = _golden_inputs(golden)
    with shipped_defaults():
        out, vh, mh = capi.fuse_once(grid, rp, views, threshold=thr, init_grid=golden.get("init_grid"), count_hits=count_hits)
    assert bits_equal(out, golden["expected_grid"])
    if count_hits:
        assert np.array_equal(mh, golden["expected_map_hits"]) and np.array_equal(vh, golden["expected_voxel_hits"])


def test_slab_fuses_of_a_larger_grid_keep_their_classes_under_shipped_defaults():
    """64 x 64 x 256 cells fused in slabs of 32 layers: each launch has 512 bricks, the grid 4096 -- the no-classes rule looks at
    the grid (dmi_capi.hip), so the slabs are classified like a whole-grid launch; bit for bit the oracle's grid."""
    from helpers import shipped_defaults
    grid = scene.default_grid((64, 64, 256))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(5, 160, 120, seed=3, dense=True)
    views.depth[np.random.default_rng(6).random(views.depth.shape) < 0.1] = -1.0
    want = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())[0]
    with shipped_defaults():
        with capi.FusionContext(grid, rp) as ctx:
            ctx.add_views(views)
            for z in range(0, 256, 32):
                ctx.fuse_slab(z, 32)
            out = ctx.download_grid()
            hist = ctx.brick_class_histogram()
    assert bits_equal(out, want)
    assert sum(hist.values()) > 0, hist


@pytest.mark.parametrize("dims,grid_dtype,n_slabs", [((64, 64, 200), "f64", 3), ((64, 64, 200), "f64", 100), ((64, 64, 200), "f32", 4),
                                                     ((48, 40, 24), "f64", 8), ((64, 64, 96), "f64", 1), ((64, 64, 96), "f64", 2)])
def test_last_chunk_fused_under_the_copy_back(dims, grid_dtype, n_slabs):
    """dmi_fuse_range_download, the last step of FusionDriver::ProcessDepthMap: the views arrive in chunks, the last chunk is
    fused slab by slab and every slab copied to the host under the next one's fusion.  Bit for bit the chunks fused whole and
    downloaded afterwards, and (f64) the oracle; a grid shorter than one slab unit, a last slab that is not a whole unit, more
    slabs asked for than the grid has units, the converting fallback (another type than the grid's) and count = 0 alike."""
    grid = scene.default_grid(dims)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(7, 160, 120, seed=21, dense=True)
    views.depth[np.random.default_rng(8).random(views.depth.shape) < 0.1] = -1.0
    np_grid = np.float64 if grid_dtype == "f64" else np.float32
    want = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())[0]
    chunks = [(0, 3), (3, 2), (5, 2)]
    with capi.FusionContext(grid, rp, grid_dtype=grid_dtype) as ref:
        for v0, n in chunks:
            ref.add_views(scene.Views(views.depth[v0:v0 + n], views.K4[v0:v0 + n], views.RT4[v0:v0 + n]))
            ref.fuse(v0, n)
        plain = ref.download_grid(np_grid).copy()
    pinned = capi.pinned_empty((grid.n_voxels,), np_grid)
    with capi.FusionContext(grid, rp, grid_dtype=grid_dtype) as ctx:
        for v0, n in chunks[:-1]:
            ctx.add_views(scene.Views(views.depth[v0:v0 + n], views.K4[v0:v0 + n], views.RT4[v0:v0 + n]))
            ctx.fuse(v0, n)
        v0, n = chunks[-1]
        ctx.add_views(scene.Views(views.depth[v0:v0 + n], views.K4[v0:v0 + n], views.RT4[v0:v0 + n]))
        got = ctx.fuse_download(v0, n, np_grid, out=pinned, n_slabs=n_slabs).copy()
        again = ctx.fuse_download(7, 0, np_grid, n_slabs=n_slabs)        # nothing left to fuse: the copy alone, pageable memory
        other = ctx.fuse_download(7, 0, np.float32 if grid_dtype == "f64" else np.float64, n_slabs=n_slabs)  # the converting path
        with pytest.raises(capi.DmiError):
            ctx.fuse_download(5, 3, np_grid)                             # range outside the resident views
    assert bits_equal(got, plain) and bits_equal(again, plain)
    assert bits_equal(other, plain.astype(other.dtype))
    if grid_dtype == "f64":
        assert bits_equal(got, want)


W = capi.VARIANT_WINDOWS_ALWAYS   # the FREE column's bit windows whatever the depth maps look like (default: maps with scattered holes)


@pytest.mark.parametrize("variant", [0, FX, FX | capi.VARIANT_KEEP_BEHIND_ADDS, 96, NC, W, FX | W, capi.VARIANT_NO_WINDOWS])
def test_golden_bit_exact_without_hit_counters(golden, variant):
    """Without hit counters and from a zero grid the kernel skips the +0.0 adds of bricks proven to lie behind every
    surface (a sum that starts at +0.0 is never -0.0, so x + 0.0 == x): same bits, including the sign of zeros.  With
    an uploaded grid (which may hold -0.0) the adds are kept."""
    grid, rp, views, thr = _golden_inputs(golden)
    out, _, _ = capi.fuse_once(grid, rp, views, threshold=thr, init_grid=golden.get("init_grid"), count_hits=False,
                               kernel_variant=variant)
    assert bits_equal(out, golden["expected_grid"])


def test_behind_bricks_keep_minus_zero_semantics():
    """A grid uploaded as -0.0 everywhere: voxels that only ever receive `+ 0` (far behind every surface, cu:115) must
    come out as +0.0, untouched voxels stay -0.0 -- whether or not hits are counted."""
    grid = scene.default_grid((64, 64, 64))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(3, 160, 120, seed=5, dense=False)   # sparse: many voxels see no depth in any view
    init = np.full((64, 64, 64), -0.0)
    p = oracle_params_from_scene(grid, rp, views)
    want, _, _ = oracle.fuse(p, views.depth, views.K4, views.RT4, init_grid=init, n_threads=oracle.max_threads())
    assert np.signbit(want[want == 0]).any() and (~np.signbit(want[want == 0])).any()
    for count_hits in (False, True):
        for variant in (0, capi.VARIANT_KEEP_BEHIND_ADDS):
            out, _, _ = capi.fuse_once(grid, rp, views, init_grid=init, count_hits=count_hits, kernel_variant=variant)
            assert bits_equal(out, want)
    # zero grid, no counters: the skip is active; a second fuse accumulates onto the first (grid no longer known zero)
    with capi.FusionContext(grid, rp) as ctx:
        ctx.add_views(views)
        ctx.fuse()
        ctx.fuse()
        twice = ctx.download_grid()
    want2, _, _ = oracle.fuse(p, views.depth, views.K4, views.RT4,
                              init_grid=oracle.fuse(p, views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())[0],
                              n_threads=oracle.max_threads())
    assert bits_equal(twice, want2)


@pytest.mark.parametrize("storage", ["auto", "f64"])
def test_golden_depth_storage_modes(golden, storage):
    grid, rp, views, thr = _golden_inputs(golden)
    out, vh, mh = capi.fuse_once(grid, rp, views, threshold=thr, init_grid=golden.get("init_grid"),
                                 depth_storage=storage)
    assert np.array_equal(vh, golden["expected_voxel_hits"]) and bits_equal(out, golden["expected_grid"])


def test_auto_storage_picks_f32_only_when_lossless():
    from conftest import load_golden
    for name, want in [("generic_sphere_32", capi.DMI_DEPTH_F32), ("general_k_f64_depth", capi.DMI_DEPTH_F64)]:
        grid, rp, views, thr = _golden_inputs(load_golden(name))
        with capi.FusionContext(grid, rp) as ctx:
            ctx.add_views(views, thr)
            assert ctx.info().depth_storage_in_use == want


def test_promotion_after_lossless_batches_keeps_every_bit():
    """First batch is f32-exact, second is not: the store is promoted to f64 and results stay exact."""
    from conftest import load_golden
    g = load_golden("general_k_f64_depth")
    grid, rp, views, thr = _golden_inputs(g)
    a = views.subset(0, 2)
    a32 = scene.Views(a.depth.astype(np.float32).astype(np.float64), a.K4, a.RT4)
    b = views.subset(2, views.n)
    with capi.FusionContext(grid, rp, count_hits=True) as ctx:
        ctx.add_views(a32)
        assert ctx.info().depth_storage_in_use == capi.DMI_DEPTH_F32
        ctx.add_views(b)
        assert ctx.info().depth_storage_in_use == capi.DMI_DEPTH_F64
        ctx.fuse()
        out = ctx.download_grid()
    p = oracle_params_from_golden(g)
    want, _, _ = oracle.fuse(p, np.concatenate([a32.depth, b.depth]), views.K4, views.RT4)
    assert bits_equal(out, want)


def test_f32_grid_is_one_rounding_away(golden):
    grid, rp, views, thr = _golden_inputs(golden)
    out, vh, _ = capi.fuse_once(grid, rp, views, threshold=thr, init_grid=None, grid_dtype="f32")
    if "init_grid" in golden:
        pytest.skip("f32 grid upload rounds the initial grid; covered by tolerance test below")
    want = golden["expected_grid"]
    assert np.array_equal(vh, golden["expected_voxel_hits"])
    assert np.array_equal(out.astype(np.float32), want.astype(np.float32))   # exactly the f32 rounding of the f64 sum
    assert np.all(np.abs(out - want) <= 2.0 ** -24 * np.abs(want) + 1e-300)


def test_batched_fuse_equals_single_fuse_f64(golden):
    """Accumulating views in two launches onto the f64 grid gives the same bits as one launch (cu:211 order)."""
    grid, rp, views, thr = _golden_inputs(golden)
    if views.n < 2:
        pytest.skip("needs two views")
    with capi.FusionContext(grid, rp, count_hits=True) as ctx:
        if "init_grid" in golden:
            ctx.upload_grid(golden["init_grid"])
        ctx.add_views(views, thr)
        half = views.n // 2
        ctx.fuse(0, half)
        ctx.fuse(half, views.n - half)
        out = ctx.download_grid()
        vh, mh = ctx.download_hits()
    assert bits_equal(out, golden["expected_grid"])
    assert np.array_equal(vh, golden["expected_voxel_hits"]) and np.array_equal(mh, golden["expected_map_hits"])


def test_reset_and_refuse_is_idempotent():
    from conftest import load_golden
    g = load_golden("generic_sphere_32")
    grid, rp, views, thr = _golden_inputs(g)
    with capi.FusionContext(grid, rp) as ctx:
        ctx.add_views(views)
        ctx.fuse()
        first = ctx.download_grid()
        ctx.reset_grid()
        assert not ctx.download_grid().any()
        ctx.reset_grid()
        ctx.fuse()
        second = ctx.download_grid()
    assert bits_equal(first, second) and bits_equal(first, g["expected_grid"])


def test_error_paths_on_device():
    g = scene.default_grid(8)
    rp = scene.default_ray_potential(g)
    with capi.FusionContext(g, rp) as ctx:
        with pytest.raises(capi.DmiError) as e:
            ctx.fuse()
        assert e.value.code == 4
        v = scene.make_views(2, 16, 12, seed=0)
        ctx.add_views(v)
        with pytest.raises(capi.DmiError):
            ctx.add_views(scene.make_views(1, 20, 12, seed=0))       # mismatching dims (filt.cxx:167-168)
        with pytest.raises(capi.DmiError):
            ctx.fuse(1, 5)
        with pytest.raises(capi.DmiError):
            ctx.download_hits()                                        # created without count_hits


@pytest.mark.parametrize("dims,n_maps,wh,dense,rotated", [
    ((64, 64, 64), 4, (320, 240), False, False),      # BASELINE configs[0]
    ((96, 80, 72), 12, (160, 120), True, True),
    ((200, 37, 19), 9, (128, 96), True, False),
])
def test_medium_scenes_against_oracle(dims, n_maps, wh, dense, rotated):
    grid = scene.default_grid(dims, rotated=rotated)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(n_maps, wh[0], wh[1], seed=7, dense=dense)
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                   n_threads=oracle.max_threads())
    for variant in (0, NC, 64, G, G | capi.VARIANT_EXACT_DIVISION | capi.VARIANT_GENERAL_K):
        out, vh, mh = capi.fuse_once(grid, rp, views, kernel_variant=variant)
        assert np.array_equal(mh, mh_w) and np.array_equal(vh, vh_w) and bits_equal(out, want)


def test_near_half_pixel_stress_fast_path_equals_exact():
    """Projections engineered to sit within a few ulps of half-integers force the fast path's
    undecided branch; it must agree with exact division everywhere."""
    grid = scene.GridDesc((128, 16, 4), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), np.eye(4))
    rp = scene.RayPotential(0.5, 0.8, 0.03, 1.5)
    W, H = 160, 24
    n = 6
    K4 = np.tile(np.eye(4), (n, 1, 1))
    RT4 = np.tile(np.eye(4), (n, 1, 1))
    eps = [0.0, 2.0 ** -52, -2.0 ** -52, 2.0 ** -30, -2.0 ** -30, 2.0 ** -19]
    for m in range(n):
        K4[m, 0, 0] = 0.5 * (1 + eps[m])       # u = (i+.5)(1+eps) at the k = 0 layer (z_cam = .5)
        K4[m, 1, 1] = 0.5 * (1 - eps[m])
    rng = np.random.default_rng(5)
    depth = rng.uniform(0.25, 3.0, size=(n, H, W))
    views = scene.Views(depth, K4, RT4)
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), depth, K4, RT4)
    for variant in TILE_SHAPES + [G, G | capi.VARIANT_EXACT_DIVISION]:
        out, vh, mh = capi.fuse_once(grid, rp, views, kernel_variant=variant)
        assert np.array_equal(mh, mh_w) and np.array_equal(vh, vh_w) and bits_equal(out, want)


def test_tiled_kernel_is_selected_only_when_its_preconditions_hold():
    """Pinhole or general K, axis-aligned or rotated grid -> every view through the tiled kernel (the GENK instantiation
    for a K whose third row is not 0 0 1 0); VARIANT_FORCE_GENERAL -> the general kernel."""
    from conftest import load_golden
    expect = {"generic_sphere_32": 1, "noncubic_70x33x17": 1, "anisotropic_rotated": 1, "general_k_f64_depth": 1}
    for name, want in expect.items():
        grid, rp, views, thr = _golden_inputs(load_golden(name))
        with capi.FusionContext(grid, rp) as ctx:
            ctx.add_views(views, thr)
            assert ctx.info().tiled_kernel == want, name
        with capi.FusionContext(grid, rp, kernel_variant=G) as ctx:
            ctx.add_views(views, thr)
            assert ctx.info().tiled_kernel == 0


@pytest.mark.parametrize("rotated", [False, True])
@pytest.mark.parametrize("count_hits", [True, False])
def test_views_of_every_kind_in_one_fusion(rotated, count_hits):
    """The reference takes any 4x4 K at one speed (cu:176).  Here: pinhole, skewed, scaled third row (K[2][2] != 1), a
    third row that depends on x and y, a fourth column, and one view no fast path can take (a focal length of 1e14: its
    error bound exceeds the tiled kernel's limit) -- fused in view order by runs (tiled, general kernel, tiled GENK ...),
    bit-identical to the oracle with exact hit counters, from a zero grid and onto an uploaded one."""
    grid = scene.default_grid((40, 36, 44), rotated=rotated)
    rp = scene.default_ray_potential(grid)
    v = scene.make_views(10, 96, 72, seed=29, dense=True)
    v.K4[1, 0, 1] = 0.7                      # skew: still a pinhole for the tiled kernel
    v.K4[2, 2, 2] = 1.25                     # non-unit homogeneous scale
    v.K4[3, 0, 3] = 3.0                      # 4th column (cu:90-92)
    v.K4[3, 1, 3] = -2.0
    v.K4[4, 2, 0] = 0.01                     # h.z depends on x and y
    v.K4[4, 2, 1] = -0.02
    v.K4[5, 2, 3] = 0.5                      # h.z offset
    v.K4[6, 0, 0] = 1e14                     # no fast path: the general kernel takes this view alone
    v.K4[8, 1, 0] = 0.3                      # h.y depends on c.x
    init = np.random.default_rng(3).standard_normal(grid.n_voxels)
    p = oracle_params_from_scene(grid, rp, v)
    for start in (None, init):
        want, vh_w, mh_w = oracle.fuse(p, v.depth, v.K4, v.RT4, init_grid=start, n_threads=oracle.max_threads())
        with capi.FusionContext(grid, rp, count_hits=count_hits) as ctx:
            if start is not None:
                ctx.upload_grid(start)
            ctx.add_views(v)
            assert ctx.info().tiled_kernel == 0 and ctx.info().k_mode == 0   # one view needs the general kernel
            ctx.fuse()
            out = ctx.download_grid()
            assert bits_equal(out, want)
            if count_hits:
                vh, mh = ctx.download_hits()
                assert np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w)
        assert mh_w[2] > 0 and mh_w[4] > 0 and np.abs(want).max() > 0.5
    # the same views without the one that needs the general kernel: one tiled (GENK) launch; f32 grid = one rounding
    keep = [m for m in range(10) if m != 6]
    sub = scene.Views(v.depth[keep], v.K4[keep], v.RT4[keep])
    want, _, _ = oracle.fuse(oracle_params_from_scene(grid, rp, sub), sub.depth, sub.K4, sub.RT4, n_threads=oracle.max_threads())
    with capi.FusionContext(grid, rp, grid_dtype="f32") as ctx:
        ctx.add_views(sub)
        assert ctx.info().tiled_kernel == 1
        ctx.fuse()
        assert np.array_equal(ctx.download_grid(np.float32), want.astype(np.float32))


@pytest.mark.parametrize("shape", TILE_SHAPES)
@pytest.mark.parametrize("case", ["scaled_translated_grid", "cameras_inside", "skewed_k", "thin_grid", "f64_depth",
                                  "negative_axes", "rotated_grid", "rotated_cameras_inside"])
def test_tiled_kernel_cases_against_oracle(shape, case):
    """Inputs that exercise the tiled kernel's preconditions and its fallbacks, checked bit for bit."""
    rng = np.random.default_rng(11)
    dims, wh, n, dense = (40, 27, 21), (96, 72), 7, True
    gm = np.eye(4)
    radius = 3.0
    if case == "scaled_translated_grid":
        gm = np.diag([1.25, 0.75, 2.0, 1.0])
        gm[:3, 3] = [0.125, -0.25, 0.0625]
    elif case == "negative_axes":
        gm = np.diag([-1.0, 1.0, -1.0, 1.0])
    elif case in ("rotated_grid", "rotated_cameras_inside"):
        # orthonormal axes that are not the coordinate axes (main.cxx:345-359 takes any orthogonal gridVecX/Y/Z):
        # the tiled kernel's rotated path, w and c.z formed per voxel
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        gm[:3, :3] = q
        gm[:3, 3] = [0.05, -0.1, 0.02]
        if case == "rotated_cameras_inside":
            radius = 0.45
    elif case == "cameras_inside":
        radius = 0.45      # cameras inside the grid: voxels behind them take the c.z < 0 exit (cu:177)
    elif case == "thin_grid":
        dims = (130, 9, 3)
    grid = scene.GridDesc(dims, (-1.0, -1.0, -1.0), tuple(2.0 / d for d in dims), gm)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(n, wh[0], wh[1], seed=3, dense=dense, radius=radius)
    if case == "skewed_k":
        views.K4[:, 0, 1] = 0.37
    if case == "f64_depth":
        views.depth[:] = np.where(views.depth == -1.0, -1.0, views.depth * (1 + 2.0 ** -40))
    p = oracle_params_from_scene(grid, rp, views)
    want, vh_w, mh_w = oracle.fuse(p, views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())
    with capi.FusionContext(grid, rp, count_hits=True, kernel_variant=shape) as ctx:
        ctx.add_views(views)
        assert ctx.info().tiled_kernel == 1
        if case == "f64_depth":
            assert ctx.info().depth_storage_in_use == capi.DMI_DEPTH_F64
        ctx.fuse()
        out = ctx.download_grid()
        vh, mh = ctx.download_hits()
    assert mh_w.sum() > 0
    assert np.array_equal(mh, mh_w) and np.array_equal(vh, vh_w) and bits_equal(out, want)


def test_tiled_kernel_keeps_negative_zero_semantics():
    """-0.0 in the initial grid survives only until the first accumulate: x + 0.0 turns it into +0.0
    exactly where the reference adds a zero (far behind the surface, cu:115)."""
    grid = scene.default_grid(24)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(3, 64, 48, seed=9, dense=True)
    init = np.full((24, 24, 24), -0.0)
    want, _, _ = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, init_grid=init)
    for variant in (0, G):
        out, _, _ = capi.fuse_once(grid, rp, views, init_grid=init, kernel_variant=variant)
        assert bits_equal(out, want)
    assert np.signbit(want).any() and (~np.signbit(want)).any()


@pytest.mark.parametrize("variant", [0, G])
def test_z_slab_contexts_are_bit_identical_to_one_fusion(variant):
    """The zero-collective multi-GPU partition (dmi_options.z_first): each slab is fused by its own context
    (on its own GPU in production) and the concatenation equals the single fusion bit for bit."""
    from cudadepthmapintegration_amd import sharding
    grid = scene.default_grid((40, 24, 37))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(6, 80, 60, seed=13, dense=True)
    want, vh_w, _ = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                n_threads=oracle.max_threads())
    world = 3
    parts, hits = [], []
    for r in range(world):
        z0, z1 = sharding.z_slab(37, r, world)
        if z1 == z0:
            continue
        slab = scene.GridDesc((40, 24, z1 - z0), grid.origin, grid.spacing, grid.grid_matrix)
        with capi.FusionContext(slab, rp, count_hits=True, kernel_variant=variant, z_first=z0) as ctx:
            ctx.add_views(views)
            ctx.fuse()
            parts.append(ctx.download_grid())
            hits.append(ctx.download_hits()[0])
    assert bits_equal(np.concatenate(parts, axis=0), want)
    assert np.array_equal(np.concatenate(hits, axis=0), vh_w)


def test_view_shards_summed_in_f32_are_within_the_stated_tolerance():
    """The north-star partition on one GPU: each shard of views fused into its own f32 grid, grids summed as
    the all-reduce does; hit counters sum exactly."""
    from cudadepthmapintegration_amd import sharding
    grid = scene.default_grid((48, 40, 32))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(10, 96, 72, seed=17, dense=True)
    want, vh_w, _ = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                n_threads=oracle.max_threads())
    world = 4
    total = np.zeros(want.shape, dtype=np.float32)
    absum = np.zeros(want.shape)
    hits = np.zeros(want.shape, dtype=np.int64)
    for r in range(world):
        lo, hi = sharding.view_shard(views.n, r, world)
        with capi.FusionContext(grid, rp, grid_dtype="f32", count_hits=True) as ctx:
            ctx.add_views(views.subset(lo, hi))
            ctx.fuse()
            part = ctx.download_grid(np.float32)
            total = total + part            # f32 adds, as ncclFloat sum
            absum += np.abs(part)
            hits += ctx.download_hits()[0]
    assert np.array_equal(hits, vh_w.astype(np.int64))
    assert np.all(np.abs(total.astype(np.float64) - want) <= sharding.sharded_tolerance(world, absum))


def test_brick_classes_cover_every_case_and_change_nothing():
    """A scene with free space, occluded space, no-depth regions, NaN depths and cameras whose frustum
    misses part of the grid: all four brick classes occur, and the grid and hit counters are bit-identical to
    the oracle and to the per-voxel path."""
    grid = scene.default_grid((64, 56, 48))
    rp = scene.RayPotential(thickness=0.02, rho=0.8, eta=0.03, delta=0.05)
    views = scene.make_views(8, 160, 120, seed=31, dense=True)
    views.depth[0, 40:60, 50:90] = -1.0            # a hole without depth inside a valid region
    views.depth[1, 10:14, 10:14] = np.nan          # NaN depths: never classified, handled per voxel
    views.depth[2] = -1.0                          # a view without any depth
    sparse = scene.make_views(3, 160, 120, seed=32, dense=False)
    views = scene.Views(np.concatenate([views.depth, sparse.depth]), np.concatenate([views.K4, sparse.K4]),
                        np.concatenate([views.RT4, sparse.RT4]))
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                   n_threads=oracle.max_threads())
    for variant in (FX, 0, 96, NC, G):   # FX: 16-voxel columns; 0 picks 8-voxel columns on a grid this small
        with capi.FusionContext(grid, rp, count_hits=True, kernel_variant=variant) as ctx:
            ctx.add_views(views)
            ctx.fuse()
            out = ctx.download_grid()
            vh, mh = ctx.download_hits()
            hist = ctx.brick_class_histogram()
        assert bits_equal(out, want), variant
        assert np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w), variant
        if variant in (FX, 0, 96):
            assert all(hist[k] > 0 for k in ("mixed", "free", "behind", "skip")), hist
            assert sum(hist.values()) == 8 * 7 * (48 // (16 if variant == FX else 8)) * views.n
        else:
            assert sum(hist.values()) == 0


@pytest.mark.parametrize("dims,n_views,classes", [((64, 64, 64), 6, False), ((64, 64, 128), 6, False), ((72, 64, 128), 6, True),
                                                  ((64, 64, 64), 47, False), ((64, 64, 64), 48, True)])
def test_tiny_launches_fuse_without_classes(dims, n_views, classes):
    """The shipped rule (dmi_capi.hip): a launch of at most 1024 bricks -- one per SIMD -- and fewer than 48 views skips the
    classification and ordering launches and takes every (brick, view) pair per voxel; from 1025 bricks or 48 views on the
    classes are built.  Bit for bit the oracle's grid either way, and VARIANT_BRICK_CLASSES_ALWAYS (what the other tests run
    with) brings the classes back."""
    from helpers import shipped_defaults
    grid = scene.default_grid(dims)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(n_views, 160, 120, seed=77, dense=True)
    views.depth[np.random.default_rng(5).random(views.depth.shape) < 0.1] = -1.0
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                   n_threads=oracle.max_threads())
    for shipped in (True, False):
        with (shipped_defaults() if shipped else contextlib.nullcontext()):
            with capi.FusionContext(grid, rp, count_hits=True) as ctx:
                ctx.add_views(views)
                ctx.fuse()
                out = ctx.download_grid()
                vh, mh = ctx.download_hits()
                hist = ctx.brick_class_histogram()
        assert bits_equal(out, want) and np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w), (dims, shipped)
        assert (sum(hist.values()) > 0) == (classes or not shipped), (dims, shipped, hist)


@pytest.mark.parametrize("focal_scale,wh", [(2.5, (96, 72)), (6.0, (48, 36)), (1.4, (160, 120))])
@pytest.mark.parametrize("speckle", [False, True])
def test_footprints_that_stick_out_of_the_image(focal_scale, wh, speckle):
    """Long focal lengths and small images: most bricks project across the image border, many by more than the 32-pixel
    margin of the validity maps.  Pairs whose part inside the image proves them unobservable are skipped,
    those in free space within the margin take the FREE column (where "outside" reads as "no depth"), the rest the full
    column: the grid is the oracle's bit for bit either way, with and without hit counters (which switch both off), and
    the classes really occur."""
    grid = scene.default_grid((96, 80, 72))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(9, wh[0], wh[1], seed=41, dense=True, focal_scale=focal_scale)
    views.K4[:, 0, 2] += 3.25   # principal point off the centre
    views.K4[:, 1, 2] -= 2.5
    if speckle:
        views.depth[np.random.default_rng(3).random(views.depth.shape) < 0.12] = -1.0
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                   n_threads=oracle.max_threads())
    for count_hits in (False, True):
        with capi.FusionContext(grid, rp, count_hits=count_hits) as ctx:
            ctx.add_views(views)
            ctx.fuse()
            out = ctx.download_grid()
            why = ctx.mixed_reason_histogram()
            if count_hits:
                vh, mh = ctx.download_hits()
                assert np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w)
        assert bits_equal(out, want), (focal_scale, wh, speckle, count_hits)
        if count_hits:   # hit counters: border pairs stay IMAGE_BORDER; without holes nothing is "free or no depth"
            assert why["image_border"] > 0 and (speckle or why["free_or_no_depth"] == 0), why
        else:
            assert why["free_or_no_depth"] > 0, why   # border pairs in free space (and, with speckle, the holes' pairs)


def test_brick_classes_with_fuse_range_and_initial_grid():
    """Classes are indexed by absolute view id: fusing sub-ranges onto a non-zero grid stays bit-exact."""
    grid = scene.default_grid((40, 40, 40))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(21, 96, 72, seed=8, dense=True)
    init = np.random.default_rng(1).normal(size=(40, 40, 40))
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                   init_grid=init, n_threads=oracle.max_threads())
    with capi.FusionContext(grid, rp, count_hits=True) as ctx:
        ctx.upload_grid(init)
        ctx.add_views(views)
        ctx.fuse(0, 5)
        ctx.fuse(5, 11)
        ctx.fuse(16, 5)
        out = ctx.download_grid()
        vh, mh = ctx.download_hits()
    assert bits_equal(out, want) and np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w)


@pytest.mark.parametrize("variant", [0, NC, G])
def test_slab_fuses_equal_one_fuse(variant):
    """dmi_fuse_slab: the grid fused slab by slab (what bench.py overlaps with the all-reduce) is bit-identical."""
    from cudadepthmapintegration_amd import sharding
    grid = scene.default_grid((40, 24, 100))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(5, 80, 60, seed=19, dense=True)
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                   n_threads=oracle.max_threads())
    with capi.FusionContext(grid, rp, count_hits=True, kernel_variant=variant) as ctx:
        ctx.add_views(views)
        for z0, zc in sharding.slab_ranges(100, 3):
            ctx.fuse_slab(z0, zc)
        out = ctx.download_grid()
        vh, mh = ctx.download_hits()
        assert bits_equal(out, want) and np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w)
        with pytest.raises(capi.DmiError):
            ctx.fuse_slab(8, 32)        # not aligned
        with pytest.raises(capi.DmiError):
            ctx.fuse_slab(96, 32)       # beyond the grid


@pytest.mark.parametrize("mode", ["one_process", "rank"])
@pytest.mark.parametrize("exchange", ["all_reduce", "reduce_scatter"])
def test_multi_context_with_one_rank_equals_plain_fuse(mode, exchange):
    """dmi_multi_* (the multi-GPU step behind the C ABI) on this one GPU, world = 1, both ways of joining (all ranks
    in this process: ncclCommInitAll; one rank per process: unique id + ncclCommInitRank): RCCL is loaded, the
    communicator reports one rank, the fusion runs slab by slab with the all-reduce of each slab on the second stream,
    and the grid is bit-identical to a plain dmi_fuse.  N > 1 arithmetic: tests/test_sharding.py."""
    grid = scene.default_grid((48, 40, 96))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(5, 80, 60, seed=23, dense=True, dtype=np.float32)
    with capi.FusionContext(grid, rp, grid_dtype="f32") as ctx:
        ctx.add_views(views)
        ctx.fuse()
        plain = ctx.download_grid(np.float32).copy()
    kw = dict(devices=[0]) if mode == "one_process" else dict(rank=0, world=1, unique_id=capi.multi_unique_id(), device=0)
    with capi.MultiContext(grid, rp, grid_dtype="f32", exchange=exchange, n_slabs=3, **kw) as m:
        if mode == "one_process":
            with pytest.raises(capi.DmiError):
                m.fuse()                               # no views anywhere: an error
        else:
            m.fuse()                                   # a rank of several may hold no view: it contributes zeros
            zeros, _ = m.download_grid(np.float32)
            assert not zeros.any()
        m.add_views(views.subset(0, 3))
        m.add_views(views.subset(3, 5))
        for _ in range(2):                             # a second step starts from zeros again (filt.cxx:133)
            m.fuse()
        got, (first, count) = m.download_grid(np.float32)
        info = m.info()
        t = m.timings()
        assert (info.world, info.n_local, info.rccl_ranks, info.n_views_total, info.n_views_local) == (1, 1, 1, 5, 5)
        assert info.rccl_version > 0 and info.n_slabs == (3 if exchange == "all_reduce" else 3)
        assert (first, count) == (0, grid.n_voxels)
        assert t.steps >= 2 and t.last_step_ms > 0 and 0 < t.last_fuse_kernel_ms <= t.last_step_ms * 1.05
        got64, _ = m.download_grid(np.float64)
    assert np.array_equal(got.view(np.uint32), plain.view(np.uint32)) and np.abs(plain).max() > 0
    assert np.array_equal(got64, plain.astype(np.float64))


@pytest.mark.parametrize("world,n_slabs,grid_dtype", [(1, 3, "f32"), (2, 1, "f32"), (2, 3, "f32"), (4, 4, "f32"), (3, 2, "f64")])
def test_peer_copy_exchange_with_ranks_sharing_this_gpu(world, n_slabs, grid_dtype):
    """DMI_EXCHANGE_PEER_COPY (reduce-scatter + all-gather by hipMemcpyPeerAsync, the sum a kernel behind the fusion, no
    RCCL): its ranks may share a device, so this one GPU runs the whole choreography with 2, 3 and 4 ranks -- every slab,
    every chunk, every copy, event and sum.  The additions are in rank order, one rounding each in the grid's type, so the
    expectation is exact: sum over the ranks, in order, of each rank's own fusion of its view shard.  Every rank must end
    with those bits; a second step starts from zeros again."""
    grid = scene.default_grid((40, 33, 100))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(7, 80, 60, seed=29, dense=True, dtype=np.float32)
    np_t = np.float32 if grid_dtype == "f32" else np.float64
    expect = None
    for r in range(world):
        lo, hi = capi.multi_view_shard(views.n, r, world)
        with capi.FusionContext(grid, rp, grid_dtype=grid_dtype) as ctx:
            if hi > lo:
                ctx.add_views(views.subset(lo, hi))
                ctx.fuse()
            part = ctx.download_grid(np_t).copy()
        expect = part if expect is None else (expect + part).astype(np_t)
    with capi.MultiContext(grid, rp, devices=[0] * world, grid_dtype=grid_dtype, exchange="peer_copy", n_slabs=n_slabs) as m:
        m.add_views(views)          # one batch: rank r takes multi_view_shard(7, r, world), as the expectation above does
        for _ in range(4):          # back to back, never synchronised: a step's reset must not overtake the previous step's copies
            m.fuse()
        got, (first, count) = m.download_grid(np_t)
        info = m.info()
        assert (info.world, info.n_local, info.rccl_ranks, info.rccl_version) == (world, world, 0, 0)   # RCCL never loaded
        assert (first, count) == (0, grid.n_voxels)
        assert bits_equal(got.astype(np.float64), expect.astype(np.float64))
        for i in range(1, world):   # every rank holds the same bits, not only rank 0
            ctx_i = m.local_context_grid(i, np_t)
            assert bits_equal(ctx_i.astype(np.float64), expect.astype(np.float64)), i
        t = m.timings()
        assert t.steps >= 4 and t.last_step_ms > 0
    assert np.abs(expect).max() > 0.5


def test_peer_copy_exchange_needs_one_process():
    grid = scene.default_grid((16, 16, 32))
    rp = scene.default_ray_potential(grid)
    with pytest.raises(capi.DmiError) as e:
        capi.MultiContext(grid, rp, rank=0, world=2, unique_id=None, device=0, exchange="peer_copy")
    assert e.value.code == 1 and "one process" in str(e.value)
    with pytest.raises(capi.DmiError) as e:      # the RCCL exchanges keep refusing a device listed twice
        capi.MultiContext(grid, rp, devices=[0, 0], exchange="all_reduce")
    assert "twice" in str(e.value)


def test_z_slab_rank_without_a_cell_layer_keeps_its_step_clock():
    """A z-slab rank of a short grid owns nothing (nz < 16 * world) and has no context: its steps are empty, but fuse /
    synchronize / timings / download must all work on it (one rank per process, as bench.py launches them; no GPU work
    crosses ranks under this partition, so each rank of the would-be world can be played here in turn)."""
    grid = scene.default_grid((24, 20, 40))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(3, 64, 48, seed=5, dense=True)
    want, _, _ = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4)
    out = np.zeros(grid.n_voxels)
    owned = 0
    for rank in range(8):   # 40 layers = 3 units of 16: ranks 3..7 own none
        with capi.MultiContext(grid, rp, rank=rank, world=8, device=0, grid_dtype="f64", partition="z_slabs") as m:
            m.add_views(views)
            for _ in range(2):
                m.fuse()
            m.synchronize()
            t = m.timings()
            assert t.steps == 2
            _, (first, count) = m.download_grid(np.float64, out=out)
            owned += count
            assert (count == 0) == (rank >= 3)
    assert owned == grid.n_voxels and bits_equal(out.reshape(want.shape), want)


def test_multi_context_z_slab_partition_is_bit_identical():
    """DMI_PARTITION_Z_SLABS with one rank: no communicator at all, f64 grid bit-identical to the oracle."""
    grid = scene.default_grid((40, 24, 37))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(6, 80, 60, seed=13, dense=True)
    want, _, _ = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                             n_threads=oracle.max_threads())
    with capi.MultiContext(grid, rp, devices=[0], grid_dtype="f64", partition="z_slabs") as m:
        m.add_views(views)
        m.fuse()
        got, (first, count) = m.download_grid(np.float64)
        assert m.info().rccl_ranks == 0 and (first, count) == (0, grid.n_voxels)
    assert bits_equal(got, want)


def test_multi_context_rejects_bad_arguments():
    grid = scene.default_grid((16, 16, 16))
    rp = scene.default_ray_potential(grid)
    with pytest.raises(capi.DmiError):
        capi.MultiContext(grid, rp, devices=[0, 0])                      # a device twice
    with pytest.raises(capi.DmiError):
        capi.MultiContext(grid, rp, devices=[capi.device_count()])       # no such device
    with pytest.raises(capi.DmiError):
        capi.MultiContext(grid, rp, rank=1, world=1, unique_id=bytes(128))  # rank outside the world
    with pytest.raises(capi.DmiError):
        capi.MultiContext(scene.default_grid((3, 3, 3)), rp, rank=0, world=2, unique_id=bytes(128),
                          exchange="reduce_scatter")                     # 27 voxels do not split over 2 ranks


def test_diagnostics_and_layer_tracking():
    """dmi_get_mixed_reason_histogram accounts for every mixed pair; a slab fused after a reset starts from zero (no grid
    read, +0.0 adds skipped) while a slab fused after an upload accumulates onto the uploaded values -- same bits as the
    oracle either way; download into a caller's (pinned) buffer."""
    grid = scene.default_grid((64, 64, 64))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(7, 160, 120, seed=9, dense=False)
    views.depth[3, 30:34, 40:44] = np.nan
    p = oracle_params_from_scene(grid, rp, views)
    with np.errstate(all="ignore"):
        want, _, _ = oracle.fuse(p, views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())
    out = capi.pinned_empty((grid.n_voxels,), np.float64)
    with capi.FusionContext(grid, rp) as ctx:
        ctx.add_views(views)
        ctx.fuse_slab(32, 32)          # upper half first, then the lower one: both from a fresh grid
        ctx.fuse_slab(0, 32)
        got = ctx.download_grid(np.float64, out=out)
        assert got.base is not None and bits_equal(got, want)
        hist, why = ctx.brick_class_histogram(), ctx.mixed_reason_histogram()
        # both slabs have been classified: the table holds all 8 x 8 x 8 bricks of 8 voxels
        assert sum(hist.values()) == 8 * 8 * 8 * views.n and sum(why.values()) == hist["mixed"]
        # (no "image_border" here: this scene's maps hold no depth near their borders, and a footprint that sticks out of the
        # image with no depth in its part inside is skipped, 4b.9)
        assert why["sentinel_and_depth"] > 0 and why["unspecified"] == 0
        ctx.fuse()
        assert ctx.mixed_reason_histogram()["nan_depth"] > 0     # view 3's NaN patch lands in some footprint
        init = np.random.default_rng(2).normal(size=(64, 64, 64))
        init[:8] = -0.0
        ctx.upload_grid(init)
        ctx.fuse_slab(0, 32)
        ctx.fuse_slab(32, 32)
        with np.errstate(all="ignore"):
            want2, _, _ = oracle.fuse(p, views.depth, views.K4, views.RT4, init_grid=init, n_threads=oracle.max_threads())
        assert bits_equal(ctx.download_grid(), want2)
    with pytest.raises(ValueError):
        with capi.FusionContext(grid, rp) as ctx:
            ctx.download_grid(np.float64, out=np.zeros(5))


@pytest.mark.parametrize("rotated", [False, True])
@pytest.mark.parametrize("focal_scale,wh,fits", [(0.9, (320, 240), True), (1.8, (320, 240), None), (2.6, (640, 480), None)])
def test_free_column_through_bit_windows(rotated, focal_scale, wh, fits):
    """Depth maps with holes scattered all over them (what a best-cost threshold leaves): the pairs in free space take the FREE
    column, and where the brick's footprint fits 32 x 64 pixels that column asks a window of validity bits fetched once per
    (brick, view) instead of gathering per voxel.  Short and long focal lengths (footprints of a few pixels, of
    tens, and -- at 9 x the image width -- wider than any window: those pairs keep the gathering column), either kind of grid,
    8- and 16-voxel columns, an uploaded grid, windows forced on maps without scattered holes and switched off: the oracle's
    grid bit for bit every time."""
    grid = scene.default_grid((96, 80, 64), rotated=rotated)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(10, wh[0], wh[1], seed=57, dense=True, focal_scale=focal_scale)
    views.K4[:, 0, 2] -= 1.75   # principal point off the centre
    holes = views.depth.copy()
    holes[np.random.default_rng(11).random(holes.shape) < 0.1] = -1.0
    speckled = scene.Views(holes, views.K4, views.RT4)
    want = oracle.fuse(oracle_params_from_scene(grid, rp, speckled), speckled.depth, speckled.K4, speckled.RT4,
                       n_threads=oracle.max_threads())[0]
    counts = {}
    for variant in (0, FX, capi.VARIANT_NO_WINDOWS, FX | capi.VARIANT_SPATIAL_ORDER):
        with capi.FusionContext(grid, rp, kernel_variant=variant) as ctx:
            ctx.add_views(speckled)
            ctx.fuse()
            out = ctx.download_grid()
            counts[variant] = (ctx.window_pair_count(), ctx.mixed_reason_histogram()["free_or_no_depth"])
        assert bits_equal(out, want), (rotated, focal_scale, variant)
    assert counts[capi.VARIANT_NO_WINDOWS][0] == 0 and (counts[capi.VARIANT_NO_WINDOWS][1] > 0 or fits is not True), counts
    for variant in (0, FX):
        n_win, n_free = counts[variant]
        assert n_win <= n_free, counts
        # (a brick of 16-voxel columns spans 38 pixels and more in these views: many of its pairs keep the gathering column)
        if fits is True:
            assert n_win > (0.5 if variant == 0 else 0.25) * n_free, (focal_scale, counts)

    # onto an uploaded grid (the sums do not start at zero: no uniform prefix, +0 adds kept; since round 5 such a launch gets no
    # windows -- they come with the kernel's specialisation for grids free of -0.0 --: its FREE column gathers)
    init = np.random.default_rng(4).normal(size=(64, 80, 96))
    want_init = oracle.fuse(oracle_params_from_scene(grid, rp, speckled), speckled.depth, speckled.K4, speckled.RT4, init_grid=init,
                            n_threads=oracle.max_threads())[0]
    with capi.FusionContext(grid, rp) as ctx:
        ctx.upload_grid(init)
        ctx.add_views(speckled)
        ctx.fuse()
        assert bits_equal(ctx.download_grid(), want_init), (rotated, focal_scale)
        assert ctx.window_pair_count() == 0
    # maps with ONE large hole each: not "scattered", so no windows by default; forced, the border and hole pairs get them
    views.depth[:, 60:120, 80:200] = -1.0
    want_one = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())[0]
    for variant, expect in ((0, False), (W, True), (FX | W, True)):
        with capi.FusionContext(grid, rp, kernel_variant=variant) as ctx:
            ctx.add_views(views)
            ctx.fuse()
            assert bits_equal(ctx.download_grid(), want_one), (rotated, focal_scale, variant)
            if not expect:
                assert ctx.window_pair_count() == 0, (variant, ctx.window_pair_count())
            elif fits is True:
                assert ctx.window_pair_count() > 0, (variant, ctx.mixed_reason_histogram())


def test_footprints_wider_than_a_window_keep_the_gathering_column():
    """A coarse grid under large images: a brick of 8 voxels spans some 80 pixels, no footprint fits a window of 32 x 64: every
    "free space or no depth" pair keeps the gathering column, and the grid is the oracle's bit for bit."""
    grid = scene.default_grid((48, 40, 32))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(6, 640, 480, seed=19, dense=True)
    views.depth[np.random.default_rng(13).random(views.depth.shape) < 0.1] = -1.0
    want = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, n_threads=oracle.max_threads())[0]
    for variant in (0, FX):
        with capi.FusionContext(grid, rp, kernel_variant=variant) as ctx:
            ctx.add_views(views)
            ctx.fuse()
            assert bits_equal(ctx.download_grid(), want), variant
            assert ctx.window_pair_count() == 0 and ctx.mixed_reason_histogram()["free_or_no_depth"] > 0, variant


@pytest.mark.parametrize("eta", [0.03, -0.2])
def test_free_column_leaves_negative_zero_alone(eta):
    """A grid uploaded as -0.0 everywhere, depth maps with holes, and a free-space constant -eta*rho of either sign: a voxel whose
    pixel holds no depth keeps its -0.0 (cu:202 returns before the add).  The FREE column's multiply-add would turn it into +0.0
    when the constant is positive, so that class is not given then (class_from_bounds); with a negative constant the product
    on such a lane is -0.0 and the sum is untouched.  Bit for bit the oracle's grid, signs of zeros included."""
    grid = scene.default_grid((64, 64, 48))
    rp = scene.RayPotential(thickness=0.05, rho=0.8, eta=eta, delta=0.2)
    views = scene.make_views(7, 160, 120, seed=23, dense=True)
    views.depth[np.random.default_rng(8).random(views.depth.shape) < 0.15] = -1.0
    init = np.full((48, 64, 64), -0.0)
    want, _, _ = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, init_grid=init,
                             n_threads=oracle.max_threads())
    assert np.signbit(want[want == 0]).any()          # some voxel is never added to and keeps its -0.0
    with capi.FusionContext(grid, rp) as ctx:
        ctx.upload_grid(init)
        ctx.add_views(views)
        ctx.fuse()
        out = ctx.download_grid()
        why = ctx.mixed_reason_histogram()
    assert np.array_equal(np.ascontiguousarray(out).view(np.uint64), np.ascontiguousarray(want).view(np.uint64))
    assert (why["free_or_no_depth"] > 0) == (eta > 0), why   # -eta*rho < 0: the class is given; > 0 on this grid: it is not


def test_info_counts_the_pixels_without_depth():
    """dmi_info::pixels_without_depth: the -1 pixels of the resident tables as the kernel sees them (after the best-cost
    threshold; a NaN is not one), counted on the device while the validity maps are built -- over several batches, odd image
    sizes (the last wave of that kernel is a partial one) and f64 storage."""
    grid = scene.default_grid((32, 32, 32))
    rp = scene.default_ray_potential(grid)
    for wh, storage in (((161, 119), "auto"), ((64, 48), "f64")):
        views = scene.make_views(5, wh[0], wh[1], seed=12, dense=True, with_best_cost=True)
        views.depth[1, 7, 9] = np.nan
        views.depth[2, :5, :] = -1.0
        want = int((oracle.apply_depth_threshold(views.depth, views.best_cost, 0.8).reshape(views.depth.shape) == -1.0).sum())
        with capi.FusionContext(grid, rp, depth_storage=storage) as ctx:
            ctx.add_views(scene.Views(views.depth[:2], views.K4[:2], views.RT4[:2], views.best_cost[:2]), threshold=0.8)
            ctx.add_views(scene.Views(views.depth[2:], views.K4[2:], views.RT4[2:], views.best_cost[2:]), threshold=0.8)
            assert int(ctx.info().pixels_without_depth) == want > 0


def test_pcie_probe_reports_plausible_rates():
    """dmi_pcie_probe: the pinned copy rates bench.py uses as the floor of its PCIe-inclusive figures."""
    h2d, d2h = capi.pcie_probe(0, 64 << 20)
    assert 1.0 < h2d < 500.0 and 1.0 < d2h < 500.0
    with pytest.raises(capi.DmiError):
        capi.pcie_probe(0, 1024)


def test_fp64_probe_reports_a_plausible_rate():
    """dmi_fp64_probe: the fp64 vector rate of the box, quoted by bench.py next to its figures (peak 78.6 TFLOP/s)."""
    rate = capi.fp64_probe(0, 5.0)
    assert 5.0 < rate < 100.0
    with pytest.raises(capi.DmiError):
        capi.fp64_probe(0, -1.0)


def test_more_views_than_the_free_sums_table_holds():
    """4200 views in one fusion: beyond kFreeSumsMax (4096) the kernel has no table of n-fold free-space sums and adds view by
    view from the start; class rows are 8192 bytes wide.  Same bits as the oracle, with and without hit counters."""
    grid = scene.default_grid((16, 16, 16))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(4200, 16, 12, seed=3, dense=True)
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                   n_threads=oracle.max_threads())
    for count_hits in (False, True):
        out, vh, mh = capi.fuse_once(grid, rp, views, count_hits=count_hits)
        assert bits_equal(out, want), count_hits
        if count_hits:
            assert np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w)
    # and the same views fused in two calls: the second run starts from the first one's grid (no table either)
    with capi.FusionContext(grid, rp) as ctx:
        ctx.add_views(views)
        ctx.fuse(0, 2000)
        ctx.fuse(2000, 2200)
        assert bits_equal(ctx.download_grid(), want)


def test_f32_grid_never_stores_negative_zero():
    """An f32 grid rounds a voxel's fp64 sum once per launch.  A tiny negative sum (here -eta*rho = -3e-162) would round to
    -0.0f; a later fusion onto that grid drops the +0.0 adds of the pairs far behind every surface, which is only sound while no
    sum is -0.0 -- so the store writes +0.0f for both zeros (fusion_device.h: stored_sum).  The grid of the first
    chunk holds no -0.0, and two chunks fused one after the other give the values of one fusion (zeros compared as equal: an f32
    grid is compared within a tolerance anyway)."""
    grid = scene.default_grid((48, 40, 32))
    s = float(max(grid.spacing))
    rp = scene.RayPotential(thickness=2.5 * s, rho=1e-160, eta=0.03, delta=10.0 * s)
    views = scene.make_views(8, 160, 120, seed=21, dense=True)
    with capi.FusionContext(grid, rp, grid_dtype="f32") as ctx:
        ctx.add_views(views.subset(0, 4))
        ctx.fuse(0, 4)
        first = ctx.download_grid(np.float32).copy()
        assert (first == 0.0).all() and not np.signbit(first).any(), "a -0.0f in the stored grid"
        ctx.add_views(views.subset(4, 8))
        ctx.fuse(4, 4)
        chunked = ctx.download_grid(np.float32).copy()
    with capi.FusionContext(grid, rp, grid_dtype="f32") as ctx:
        ctx.add_views(views)
        ctx.fuse()
        whole = ctx.download_grid(np.float32)
    assert not np.signbit(chunked).any() and np.array_equal(chunked, whole)
