"""MeshColoration pass: oracle cross-checks on the CPU, GPU parity through the C ABI.
All three outputs are integers: the bar is bit-exact."""
import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene
from oracle import oracle, oracle_np


def _views(n, W, H, seed, radius=3.0):
    v = scene.make_views(n, W, H, seed=seed, radius=radius)
    return v.K4, v.RT4, scene.make_colors(n, W, H, seed=seed + 1)


def test_known_answers_single_pixel():
    """One vertex on the optical axis of identity-pose cameras: pixel = (cx, cy) of K; mean / median / count follow
    MC.cxx:174-186 (integer mean, median of an even count = mean of the middle two, truncated)."""
    W, H, n = 8, 6, 4
    K4 = np.tile(np.eye(4), (n, 1, 1))
    K4[:, 0, 0] = K4[:, 1, 1] = 10.0
    K4[:, 0, 2], K4[:, 1, 2] = 3.0, 2.0
    RT4 = np.tile(np.eye(4), (n, 1, 1))
    colors = np.zeros((n, H, W, 3), dtype=np.uint8)
    vals = [10, 13, 200, 20]
    for m in range(n):
        colors[m, H - 1 - 2, 3] = (vals[m], 255 - vals[m], m)      # image pixel (3, 2) lives in vtk row H-1-2
    mean, median, count = oracle.color_mesh(np.array([[0.0, 0.0, 1.0]]), colors, K4, RT4)
    assert count[0] == 4
    assert list(mean[0]) == [sum(vals) // 4, (4 * 255 - sum(vals)) // 4, 1]          # 243/4 = 60.75 -> 60
    assert list(median[0]) == [(13 + 20) // 2, ((255 - 20) + (255 - 13)) // 2, 1]     # (1 + 2) / 2 = 1.5 -> 1
    # a vertex behind the cameras still projects (no z test in RD.cxx:169-182): (0,0,-1) -> the same pixel
    _, _, count = oracle.color_mesh(np.array([[0.0, 0.0, -1.0]]), colors, K4, RT4)
    assert count[0] == 4
    # z = 0: division by zero -> not a pixel
    _, _, count = oracle.color_mesh(np.array([[0.5, 0.5, 0.0]]), colors, K4, RT4)
    assert count[0] == 0


def test_c_oracle_matches_numpy_restatement():
    K4, RT4, colors = _views(7, 48, 36, seed=3)
    pts = scene.make_mesh_points(300, seed=4)
    a = oracle.color_mesh(pts, colors, K4, RT4)
    b = oracle_np.color_mesh_np(pts, colors, K4, RT4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[2].max() >= 3 and a[2].min() < a[2].max()


def test_color_mesh_without_gpu_fails_loudly():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    K4, RT4, colors = _views(2, 16, 12, seed=1)
    with pytest.raises(capi.DmiError) as e:
        capi.color_mesh(np.zeros((4, 3)), colors, K4, RT4)
    assert e.value.code == 2


@pytest.mark.gpu
@pytest.mark.parametrize("n_views,wh,nv,radius", [(5, (64, 48), 1000, 3.0), (33, (160, 120), 20000, 3.0),
                                                   (8, (96, 72), 5000, 0.8)])
def test_gpu_color_mesh_bit_exact(n_views, wh, nv, radius):
    K4, RT4, colors = _views(n_views, wh[0], wh[1], seed=11, radius=radius)
    pts = scene.make_mesh_points(nv, seed=12)
    pts[:3] = [[0, 0, 0], [1e9, -1e9, 1e9], [np.nan, 0, 0]]       # degenerate vertices
    want = oracle.color_mesh(pts, colors, K4, RT4)
    got = capi.color_mesh(pts, colors, K4, RT4)
    for name, g, w in zip(("mean", "median", "count"), got, want):
        assert np.array_equal(g, w), name
    assert want[2].max() >= min(n_views, 4)


@pytest.mark.gpu
def test_gpu_color_mesh_argument_errors():
    K4, RT4, colors = _views(2, 16, 12, seed=1)
    with pytest.raises(capi.DmiError):
        capi.color_mesh(np.zeros((4, 3)), colors, K4, RT4, device=99)
    m, d, c = capi.color_mesh(np.zeros((0, 3)), colors, K4, RT4)
    assert m.shape == (0, 3) and c.shape == (0,)


@pytest.mark.gpu
def test_gpu_mesh_coloration_from_list_files(tmp_path):
    """The reference's own input form: MeshColoration(mesh, vtiList, krtdList) with "Color" arrays in the .vti files."""
    views = scene.make_views(4, 40, 30, seed=21)
    colors = scene.make_colors(4, 40, 30, seed=22)
    lv, lk = scene.write_view_files(str(tmp_path), views, colors)
    pts = scene.make_mesh_points(2000, seed=23)
    want = oracle.color_mesh(pts, colors, views.K4, views.RT4)
    got = capi.mesh_coloration_from_lists(pts, lv, lk)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_mesh_coloration_from_lists_reports_missing_files(tmp_path):
    with pytest.raises(RuntimeError):
        capi.mesh_coloration_from_lists(np.zeros((3, 3)), str(tmp_path / "a.txt"), str(tmp_path / "b.txt"))


@pytest.mark.gpu
def test_gpu_color_context_resident_views_chunks_and_batches():
    """dmi_color_context: views added in two batches stay resident, several vertex sets are coloured against them,
    and a small scratch budget forces the chunked path -- all bit-identical to the oracle."""
    K4, RT4, colors = _views(9, 80, 60, seed=31)
    pts = scene.make_mesh_points(7000, seed=32)
    want = oracle.color_mesh(pts, colors, K4, RT4)
    with capi.ColorContext() as c:
        with pytest.raises(capi.DmiError):
            c.process(pts[:10])                      # no views yet (MC.cxx:102-106)
        c.add_views(colors[:4], K4[:4], RT4[:4])
        c.add_views(colors[4:], K4[4:], RT4[4:])
        got = c.process(pts)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        assert c.kernel_ms() > 0
        c.set_scratch_budget(9 * 4 * 1024)           # 1024 vertices per chunk -> 7 chunks
        got = c.process(pts)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        c.set_vertex_reorder(True)                   # Z-order processing per chunk: same bits, with and without chunks
        weird = pts.copy()
        weird[5] = np.nan                            # a NaN vertex takes no part in the bounding box and colours to 0
        for budget in (9 * 4 * 1024, 1 << 30):
            c.set_scratch_budget(budget)
            got = c.process(pts)
            for g, w in zip(got, want):
                assert np.array_equal(g, w)
        gw = c.process(weird)
        assert gw[2][5] == 0 and all(np.array_equal(np.delete(g, 5, axis=0), np.delete(w, 5, axis=0)) for g, w in zip(gw, want))
        c.set_vertex_reorder(False)
        # vertices in a mesh's order (neighbours next to each other) take the pipelined view loop of the projection kernel, an odd
        # and an even number of views, one view, chunked or not: the same bits as in any other order
        order = scene.morton_order(pts)
        for budget in (9 * 4 * 1024, 1 << 30):
            c.set_scratch_budget(budget)
            got = c.process(pts[order])
            for g, w in zip(got, want):
                assert np.array_equal(g, w[order])
        sub = c.process(pts[100:1100])               # another vertex set, same resident views
        for g, w in zip(sub, want):
            assert np.array_equal(g, w[100:1100])
        for n_v in (2, 1):
            c.clear_views()
            c.add_views(colors[:n_v], K4[:n_v], RT4[:n_v])
            for sel in (np.arange(500), order[:3000]):
                got2 = c.process(pts[sel])
                want2 = oracle.color_mesh(pts[sel], colors[:n_v], K4[:n_v], RT4[:n_v])
                for g, w in zip(got2, want2):
                    assert np.array_equal(g, w)
        with pytest.raises(capi.DmiError):
            c.add_views(colors[:1, :10], K4[:1], RT4[:1])   # a view of another size (MC.cxx:111 reads view 0's)


def _post_golden(name):
    import os
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post", name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def test_oracles_match_committed_coloration_fixture():
    g = _post_golden("coloration")
    for fn in (oracle.color_mesh, oracle_np.color_mesh_np):
        mean, median, count = fn(g["points"], g["colors"], g["K4"], g["RT4"])
        assert np.array_equal(mean, g["expected_mean"]) and np.array_equal(median, g["expected_median"])
        assert np.array_equal(count, g["expected_count"])


@pytest.mark.gpu
def test_gpu_matches_committed_coloration_fixture():
    g = _post_golden("coloration")
    mean, median, count = capi.color_mesh(g["points"], g["colors"], g["K4"], g["RT4"])
    assert np.array_equal(mean, g["expected_mean"]) and np.array_equal(median, g["expected_median"])
    assert np.array_equal(count, g["expected_count"])


@pytest.mark.gpu
def test_gpu_coloration_at_config5_scale():
    """BASELINE configs[4]'s coloration pass at one GPU's share: 64 views of 1920x1080 resident, 1 M mesh vertices (several
    chunks of the 1 GiB scratch budget would be needed at 512 views; here one).  All three outputs of a 12 000-vertex
    sample -- spread over the whole vertex range, so every part of every chunk is probed -- match the oracle exactly, and
    the whole result obeys what the arithmetic implies (count <= views; mean and median 0 where count is 0)."""
    n, W, H, nv = 64, 1920, 1080, 1_000_000
    views = scene.make_views(n, 8, 8, seed=41)                 # cameras only
    K4 = views.K4.copy()
    K4[:, 0, 0] = K4[:, 1, 1] = 0.9 * W
    K4[:, 0, 2], K4[:, 1, 2] = W / 2.0, H / 2.0
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([(xx * 3 + yy * 7) % 256, (xx * 5 + yy * 11 + 80) % 256, (xx ^ yy) % 256], axis=-1).astype(np.uint8)
    colors = np.empty((n, H, W, 3), dtype=np.uint8)
    for m in range(n):                                         # a different image per view, cheap to make
        colors[m] = np.roll(base, shift=(13 * m, 29 * m), axis=(0, 1)) + np.uint8(3 * m)
    pts = scene.make_mesh_points(nv, seed=42)
    with capi.ColorContext() as c:
        c.add_views(colors, K4, views.RT4)
        mean, median, count = c.process(pts)
        assert c.kernel_ms() > 0
    assert count.max() <= n and count.min() >= 0 and count.mean() > n / 4
    none = count == 0
    assert not mean[none].any() and not median[none].any()
    ids = np.unique(np.concatenate([np.arange(0, nv, 83), np.array([0, nv - 1])]))
    assert len(ids) >= 12000
    want = oracle.color_mesh(pts[ids], colors, K4, views.RT4)
    for got, w in zip((mean[ids], median[ids], count[ids]), want):
        assert np.array_equal(got, w)
