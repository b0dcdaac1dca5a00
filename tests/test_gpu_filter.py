"""GPU parity at the operator level: the mirror of vtkCudaReconstructionFilter (setters -> Update() ->
"reconstruction_scalar" cell array) against the oracle, with in-memory views and with the reference's
list-file inputs (.vti + .krtd).  Bar: bit-exact fp64 (the filter keeps the reference's f64 grid)."""
import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene
from oracle import oracle
from helpers import bits_equal, oracle_params_from_scene

pytestmark = pytest.mark.gpu


def _configure(f, grid, rp, thr):
    f.SetRayPotentialThickness(rp.thickness)
    f.SetRayPotentialRho(rp.rho)
    f.SetRayPotentialEta(rp.eta)
    f.SetRayPotentialDelta(rp.delta)
    f.SetThresholdBestCost(thr)
    f.SetGridMatrix(grid.grid_matrix)
    f.SetInputData([c + 1 for c in grid.cell_dims], grid.origin, grid.spacing)


def _oracle(grid, rp, views, thr):
    d = oracle.apply_depth_threshold(views.depth, views.best_cost, thr).reshape(views.depth.shape)
    want, _, _ = oracle.fuse(oracle_params_from_scene(grid, rp, views), d, views.K4, views.RT4,
                             n_threads=oracle.max_threads())
    return want


@pytest.mark.parametrize("rotated", [False, True])
def test_filter_in_memory_views_bit_exact(rotated):
    grid = scene.default_grid((40, 33, 29), rotated=rotated)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(5, 96, 72, seed=21, dense=True, with_best_cost=True)
    thr = 0.8
    with capi.ReconstructionFilter() as f:
        _configure(f, grid, rp, thr)
        f.SetFilePathKRTD("unused: views are in memory")
        f.SetFilePathVTI("unused: views are in memory")
        for m in range(views.n):
            f.AddView(views.depth[m], views.K4[m][:3, :3], views.RT4[m], views.best_cost[m])
        assert f.Update() == 1, f.LastError()
        out = f.GetOutputScalars()
        assert f.GetExecutionTime() >= 0 and f.GetFuseKernelMs() > 0
    assert out.shape == (29, 33, 40)
    assert bits_equal(out, _oracle(grid, rp, views, thr))


def test_filter_from_list_files_bit_exact(tmp_path):
    """The reference's own input form: vtiList.txt + krtdList.txt next to the files they name."""
    grid = scene.default_grid((24, 20, 16))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(3, 48, 36, seed=4, dense=True, with_best_cost=True)
    lv, lk = scene.write_view_files(str(tmp_path), views)
    thr = 0.7
    with capi.ReconstructionFilter() as f:
        _configure(f, grid, rp, thr)
        f.SetFilePathVTI(lv)
        f.SetFilePathKRTD(lk)
        assert f.Update() == 1, f.LastError()
        out = f.GetOutputScalars()
    assert bits_equal(out, _oracle(grid, rp, views, thr))


def test_filter_from_compressed_appended_vti_files_bit_exact(tmp_path):
    """Depth maps as vtkXMLImageDataWriter writes them by default (appended, base64, zlib): same bits as the ascii form."""
    from vti_writer import write_vti
    grid = scene.default_grid((24, 20, 16))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(3, 48, 36, seed=4, dense=True, with_best_cost=True)
    lv, lk = scene.write_view_files(str(tmp_path), views)
    for m, name in enumerate(capi.extract_all_file_path(lv)):
        write_vti(name, {"Depths": views.depth[m], "Best Cost Values": views.best_cost[m]}, views.width, views.height,
                  mode="appended-base64", compress=True, block=4096)
    with capi.ReconstructionFilter() as f:
        _configure(f, grid, rp, 0.7)
        f.SetFilePathVTI(lv)
        f.SetFilePathKRTD(lk)
        assert f.Update() == 1, f.LastError()
        out = f.GetOutputScalars()
    assert bits_equal(out, _oracle(grid, rp, views, 0.7))


def test_filter_rejects_mismatching_view_sizes():
    grid = scene.default_grid(8)
    rp = scene.default_ray_potential(grid)
    a = scene.make_views(1, 16, 12, seed=0)
    b = scene.make_views(1, 20, 12, seed=0)
    with capi.ReconstructionFilter() as f:
        _configure(f, grid, rp, 1.0)
        f.SetFilePathKRTD("m")
        f.SetFilePathVTI("m")
        f.AddView(a.depth[0], a.K4[0][:3, :3], a.RT4[0])
        f.AddView(b.depth[0], b.K4[0][:3, :3], b.RT4[0])
        assert f.Update() == 0 and "size" in f.LastError()
