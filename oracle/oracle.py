"""ctypes loader for the C oracle (oracle/tsdf_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see the header of tsdf_oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package never does.

The oracle takes the same quantities the reference's driver hands to its
kernel (Reconstruction/CudaReconstruction.cu:269-298 and :343-365): point
dimensions of the vtkImageData grid, origin, spacing, 4x4 grid matrix, the four
ray-potential parameters, depth-map dimensions, and per depth map a W*H f64
depth table (vtk row order, -1 = no depth) plus row-major 4x4 K and RT.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class _Params(ctypes.Structure):
    _fields_ = [
        ("grid_matrix", ctypes.c_double * 16),
        ("grid_orig", ctypes.c_double * 3),
        ("point_dims", ctypes.c_int32 * 3),
        ("grid_spacing", ctypes.c_double * 3),
        ("depth_dims", ctypes.c_int32 * 2),
        ("thick", ctypes.c_double),
        ("rho", ctypes.c_double),
        ("eta", ctypes.c_double),
        ("delta", ctypes.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    srcs = [os.path.join(_HERE, "tsdf_oracle.c"), os.path.join(_HERE, "coloration_oracle.c")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(_LIB_PATH) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        L.oracle_fuse.restype = None
        L.oracle_fuse.argtypes = [ctypes.POINTER(_Params), dp, dp, dp, ctypes.c_int, dp,
                                  ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64),
                                  ctypes.c_int]
        L.oracle_fuse_voxels.restype = None
        L.oracle_fuse_voxels.argtypes = [ctypes.POINTER(_Params), dp, dp, dp, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, dp, dp,
                                         ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        L.oracle_apply_depth_threshold.restype = None
        L.oracle_apply_depth_threshold.argtypes = [dp, dp, ctypes.c_int64, ctypes.c_double]
        L.oracle_k3_to_k4.restype = None
        L.oracle_k3_to_k4.argtypes = [dp, dp]
        L.oracle_ray_potential.restype = ctypes.c_double
        L.oracle_ray_potential.argtypes = [ctypes.POINTER(_Params), ctypes.c_double, ctypes.c_double]
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_color_mesh.restype = None
        L.oracle_color_mesh.argtypes = [dp, ctypes.c_int64, ctypes.POINTER(ctypes.c_uint8), dp, dp, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint8),
                                        ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)]
        L.oracle_cell_to_point.restype = None
        L.oracle_cell_to_point.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
        L.oracle_iso_active_cells.restype = ctypes.c_int64
        L.oracle_iso_active_cells.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                              ctypes.POINTER(ctypes.c_int64), ctypes.c_int64]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def make_params(cell_dims, origin, spacing, grid_matrix, thick, rho, eta, delta, depth_w, depth_h) -> _Params:
    """cell_dims are voxel (cell) counts; the reference's c_gridDims are cell_dims + 1."""
    p = _Params()
    gm = _c64(grid_matrix, (16,))
    for i in range(16):
        p.grid_matrix[i] = gm[i]
    for i in range(3):
        p.grid_orig[i] = float(origin[i])
        p.point_dims[i] = int(cell_dims[i]) + 1
        p.grid_spacing[i] = float(spacing[i])
    p.depth_dims[0] = int(depth_w)
    p.depth_dims[1] = int(depth_h)
    p.thick, p.rho, p.eta, p.delta = float(thick), float(rho), float(eta), float(delta)
    return p


def max_threads() -> int:
    """Threads worth using: OpenMP's maximum, capped by this process's CPU affinity and by the cgroup's CPU quota (a GPU
    box hands a job a share of the host's cores; oversubscribing that share only adds context switches)."""
    n = int(lib().oracle_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                if text[0] != "max":
                    n = min(n, max(1, -(-int(text[0]) // int(text[1]))))
            else:
                quota = int(text[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, -(-quota // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def fuse(params: _Params, depths, K4, RT4, init_grid=None, count_hits=True, n_threads=1):
    """Fuse all maps into a grid.  Returns (grid f64 [nz,ny,nx], voxel_hits u32, map_hits u64)."""
    depths = _c64(depths)
    n = depths.shape[0]
    W, H = params.depth_dims[0], params.depth_dims[1]
    assert depths.size == n * W * H, "depth table does not match depth_dims"
    K4 = _c64(K4, (n, 16))
    RT4 = _c64(RT4, (n, 16))
    nx, ny, nz = (params.point_dims[i] - 1 for i in range(3))
    nvox = nx * ny * nz
    grid = np.zeros(nvox, dtype=np.float64) if init_grid is None else _c64(init_grid, (nvox,)).copy()
    vh = np.zeros(nvox, dtype=np.uint32) if count_hits else None
    mh = np.zeros(n, dtype=np.uint64) if count_hits else None
    lib().oracle_fuse(ctypes.byref(params), _dp(depths), _dp(K4), _dp(RT4), n, _dp(grid),
                      vh.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)) if count_hits else None,
                      mh.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)) if count_hits else None,
                      int(n_threads))
    grid = grid.reshape(nz, ny, nx)
    if count_hits:
        vh = vh.reshape(nz, ny, nx)
    return grid, vh, mh


def fuse_voxels(params: _Params, depths, K4, RT4, voxel_ids, init=None, n_threads=1):
    """Fuse only the listed voxels (x-fastest linear ids).  Returns (values f64, hits u32)."""
    depths = _c64(depths)
    n = depths.shape[0]
    K4 = _c64(K4, (n, 16))
    RT4 = _c64(RT4, (n, 16))
    ids = np.ascontiguousarray(voxel_ids, dtype=np.int64)
    out = np.zeros(ids.size, dtype=np.float64)
    hits = np.zeros(ids.size, dtype=np.uint32)
    init_c = None if init is None else _c64(init, (ids.size,))
    lib().oracle_fuse_voxels(ctypes.byref(params), _dp(depths), _dp(K4), _dp(RT4), n,
                             ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ids.size,
                             _dp(init_c) if init_c is not None else None, _dp(out),
                             hits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(n_threads))
    return out, hits


def apply_depth_threshold(depths, best_cost, threshold) -> np.ndarray:
    """RD.cxx:138-167 on a copy of `depths`."""
    d = _c64(depths).copy()
    b = _c64(best_cost)
    assert d.size == b.size
    lib().oracle_apply_depth_threshold(_dp(d), _dp(b), d.size, float(threshold))
    return d


def k3_to_k4(K3) -> np.ndarray:
    K3 = _c64(K3, (9,))
    out = np.zeros(16, dtype=np.float64)
    lib().oracle_k3_to_k4(_dp(K3), _dp(out))
    return out.reshape(4, 4)


def ray_potential(params: _Params, real_distance: float, depth_map_distance: float) -> float:
    return float(lib().oracle_ray_potential(ctypes.byref(params), real_distance, depth_map_distance))


def color_mesh(points, colors, K4, RT4):
    """MeshColoration::ProcessColoration (Coloration/MeshColoration.cxx:98-199).
    points [nv,3] f64; colors [n,H,W,3] u8 in vtk row order.  Returns (mean u8[nv,3], median u8[nv,3], count i32[nv])."""
    pts = _c64(points).reshape(-1, 3)
    col = np.ascontiguousarray(colors, dtype=np.uint8)
    n, H, W, _ = col.shape
    K4 = _c64(K4, (n, 16))
    RT4 = _c64(RT4, (n, 16))
    nv = pts.shape[0]
    mean = np.zeros((nv, 3), dtype=np.uint8)
    median = np.zeros((nv, 3), dtype=np.uint8)
    count = np.zeros(nv, dtype=np.int32)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    lib().oracle_color_mesh(_dp(pts), nv, col.ctypes.data_as(u8), _dp(K4), _dp(RT4), n, W, H, mean.ctypes.data_as(u8),
                            median.ctypes.data_as(u8), count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return mean, median, count


def iso_active_cells(points, iso: float) -> np.ndarray:
    """Linear ids (ascending) of the cells whose eight corner point values straddle `iso` -- the cells marching cubes can
    get triangles from (Reconstruction/main.cxx:169-173).  points [nz+1,ny+1,nx+1] f64."""
    p = _c64(points)
    nz, ny, nx = (d - 1 for d in p.shape)
    n = int(lib().oracle_iso_active_cells(_dp(p), nx, ny, nz, float(iso), None, 0))
    ids = np.zeros(n, dtype=np.int64)
    if n:
        lib().oracle_iso_active_cells(_dp(p), nx, ny, nz, float(iso), ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n)
    return ids


def cell_to_point(cells) -> np.ndarray:
    """vtkCellDataToPointData on an image grid (Reconstruction/main.cxx:151-155).  cells [nz,ny,nx] f64 ->
    points [nz+1,ny+1,nx+1] f64."""
    c = _c64(cells)
    nz, ny, nx = c.shape
    out = np.zeros((nz + 1, ny + 1, nx + 1), dtype=np.float64)
    lib().oracle_cell_to_point(_dp(c), nx, ny, nz, _dp(out))
    return out
