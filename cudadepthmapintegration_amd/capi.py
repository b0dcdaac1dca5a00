"""ctypes binding of the C ABI (include/dmi.h) -- used by tests, smoke and bench.

There is NO CPU fallback: if libdmi_hip.so is missing or no HIP device is present every
entry point raises.  The library is built in-tree by cudadepthmapintegration_amd.build.
"""
from __future__ import annotations

import ctypes
import os
import sys

import numpy as np

from . import build as _build
from .scene import GridDesc, RayPotential, Views

DMI_OK = 0
DMI_F32, DMI_F64 = 0, 1
DMI_DEPTH_AUTO, DMI_DEPTH_F32, DMI_DEPTH_F64 = 0, 1, 2

# kernel_variant bits (tuning knobs)
VARIANT_EXACT_DIVISION = 1  # disable the checked-reciprocal fast path
VARIANT_GENERAL_K = 2  # ignore K structure, evaluate the full 4x4 rows
VARIANT_FORCE_GENERAL = 16  # never use the register-tiled kernel
VARIANT_NO_BRICK_CLASSES = 256  # tiled kernel without the proven per-brick shortcuts
VARIANT_KEEP_BEHIND_ADDS = 1024  # tiled kernel: perform +0.0 adds even when they cannot change a sum
VARIANT_NO_INTERIOR = 2048  # tiled kernel: keep the in-front / in-image tests for every mixed pair
VARIANT_XCD_RUNS = 8192  # tiled kernel: ordered bricks dealt to the XCDs in runs (round 1) instead of an eighth of a level each
VARIANT_ZMAJOR_SLOTS = 16384  # tiled kernel: super-bricks enumerated x, y, z (until r03h) instead of in Z-order
VARIANT_PERSISTENT_ALWAYS = 32768  # tiled kernel: persistent one-wave workgroups whatever the number of views (default: from 96 on)
VARIANT_PERSISTENT_NEVER = 65536  # tiled kernel: one workgroup per brick whatever the number of views
VARIANT_NO_WINDOWS = 262144  # tiled kernel: the FREE column always gathers from the validity maps (no bit windows)
VARIANT_WINDOWS_ALWAYS = 524288  # tiled kernel: bit windows whatever the depth maps look like (default: maps with scattered holes)
VARIANT_COST_ORDER = 1048576  # tiled kernel: bricks ordered by their number of mixed views whatever the grid's size (default: up to 2^16 bricks)
VARIANT_NO_COST_ORDER = 2097152  # tiled kernel: never (four work levels, an eighth of each per XCD)
VARIANT_BRICK_CLASSES_ALWAYS = 131072  # tiled kernel: brick classes for tiny grids too (default: none up to 1024 bricks per launch)
VARIANT_FIXED_TILE_SHAPE = 4096  # tile-shape bits 0 mean shape 0 (tk16_w5) whatever the grid size; without it grids
                                 # below 512^3 pick tk8_w7 on their own
VARIANT_SPATIAL_ORDER = 512  # tiled kernel: workgroups in spatial order instead of heaviest bricks first
VARIANT_TILE_SHAPE = {"tk16_w5_g8_1x1": 0, "tk8_w6_g8_2x2": 32, "tk16_w5_g8_2x2": 64, "tk8_w6": 96, "tk8_w7_g8": 128, "tk16_w5_g4": 160,
                      "tk16_w6_g2": 192, "tk8_w6_g8_1x1": 224}  # tiled kernel: column height / compiler register budget / load group


class GridDescC(ctypes.Structure):
    _fields_ = [("cell_dims", ctypes.c_int32 * 3), ("origin", ctypes.c_double * 3),
                ("spacing", ctypes.c_double * 3), ("grid_matrix", ctypes.c_double * 16)]


class RayPotentialC(ctypes.Structure):
    _fields_ = [("thickness", ctypes.c_double), ("rho", ctypes.c_double), ("eta", ctypes.c_double),
                ("delta", ctypes.c_double)]


class OptionsC(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("grid_dtype", ctypes.c_int32), ("depth_storage", ctypes.c_int32),
                ("count_hits", ctypes.c_int32), ("kernel_variant", ctypes.c_int32), ("z_first", ctypes.c_int32),
                ("stream", ctypes.c_void_p), ("external_grid", ctypes.c_void_p)]


class TimingsC(ctypes.Structure):
    _fields_ = [("last_fuse_kernel_ms", ctypes.c_double), ("total_fuse_kernel_ms", ctypes.c_double),
                ("fuse_launches", ctypes.c_uint64), ("last_upload_ms", ctypes.c_double),
                ("last_download_ms", ctypes.c_double), ("last_cell_to_point_ms", ctypes.c_double),
                ("last_fuse_main_kernel_ms", ctypes.c_double), ("total_fuse_main_kernel_ms", ctypes.c_double)]


class InfoC(ctypes.Structure):
    _fields_ = [("n_voxels", ctypes.c_int64), ("n_views", ctypes.c_int32), ("depth_width", ctypes.c_int32),
                ("depth_height", ctypes.c_int32), ("depth_storage_in_use", ctypes.c_int32),
                ("grid_dtype", ctypes.c_int32), ("k_mode", ctypes.c_int32), ("kernel_variant", ctypes.c_int32),
                ("tiled_kernel", ctypes.c_int32), ("device_bytes", ctypes.c_uint64), ("pixels_without_depth", ctypes.c_uint64)]


class MultiOptionsC(ctypes.Structure):
    _fields_ = [("grid_dtype", ctypes.c_int32), ("depth_storage", ctypes.c_int32), ("kernel_variant", ctypes.c_int32),
                ("partition", ctypes.c_int32), ("exchange", ctypes.c_int32), ("n_slabs", ctypes.c_int32)]


class MultiInfoC(ctypes.Structure):
    _fields_ = [("world", ctypes.c_int32), ("n_local", ctypes.c_int32), ("first_rank", ctypes.c_int32),
                ("rccl_ranks", ctypes.c_int32), ("rccl_version", ctypes.c_int32), ("partition", ctypes.c_int32),
                ("exchange", ctypes.c_int32), ("n_slabs", ctypes.c_int32), ("n_voxels", ctypes.c_int64),
                ("n_views_total", ctypes.c_int64), ("n_views_local", ctypes.c_int64)]


class MultiTimingsC(ctypes.Structure):
    _fields_ = [("last_step_ms", ctypes.c_double), ("last_fuse_kernel_ms", ctypes.c_double),
                ("total_step_ms", ctypes.c_double), ("steps", ctypes.c_uint64)]


DMI_PARTITION_VIEWS, DMI_PARTITION_Z_SLABS = 0, 1
DMI_EXCHANGE_ALL_REDUCE, DMI_EXCHANGE_REDUCE_SCATTER, DMI_EXCHANGE_PEER_COPY = 0, 1, 2
DMI_UNIQUE_ID_BYTES = 128

# every symbol include/dmi.h declares (tests/test_abi.py checks the library exports them all)
ABI_SYMBOLS = [
    "dmi_default_options", "dmi_create", "dmi_destroy", "dmi_last_error", "dmi_add_views", "dmi_add_views_f32",
    "dmi_clear_views", "dmi_reset_grid", "dmi_upload_grid", "dmi_fuse", "dmi_fuse_range", "dmi_fuse_slab", "dmi_fuse_range_download", "dmi_synchronize",
    "dmi_download_grid_f64", "dmi_download_grid_f32", "dmi_download_hits", "dmi_grid_device_pointer",
    "dmi_get_brick_class_histogram", "dmi_get_timings", "dmi_get_info", "dmi_alloc_pinned", "dmi_free_pinned", "dmi_pcie_probe", "dmi_fp64_probe", "dmi_abi_version", "dmi_device_count",
    "dmi_color_mesh", "dmi_color_last_error", "dmi_cell_to_point", "dmi_download_point_data_f64",
    "dmi_point_data_device_pointer", "dmi_color_create", "dmi_color_destroy", "dmi_color_add_views",
    "dmi_color_clear_views", "dmi_color_process", "dmi_color_get_kernel_ms", "dmi_get_mixed_reason_histogram", "dmi_get_window_pair_count", "dmi_get_view_paths", "dmi_get_upload_kernel_ms", "dmi_sizeof_info", "dmi_sizeof_timings",
    "dmi_color_set_scratch_budget", "dmi_color_set_vertex_reorder", "dmi_iso_active_cells",
    "dmi_multi_default_options", "dmi_multi_view_shard", "dmi_multi_z_slab", "dmi_multi_slab_ranges", "dmi_multi_peer_chunk", "dmi_multi_create",
    "dmi_multi_get_unique_id", "dmi_multi_create_rank", "dmi_multi_destroy", "dmi_multi_last_error", "dmi_multi_add_views",
    "dmi_multi_add_views_f32", "dmi_multi_add_local_views", "dmi_multi_add_local_views_f32", "dmi_multi_clear_views", "dmi_multi_fuse", "dmi_multi_synchronize",
    "dmi_multi_download_grid_f32", "dmi_multi_download_grid_f64", "dmi_multi_get_info", "dmi_multi_get_timings",
    "dmi_multi_local_context",
]

_lib = None


class DmiError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"dmi error {code}: {message}")
        self.code = code


def library_path() -> str:
    return _build.LIB_PATH


def load() -> ctypes.CDLL:
    """Load libdmi_hip.so (building it with hipcc if the sources are newer).  Raises if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    if "torch" not in sys.modules:
        # torch ships its own libamdhip64 / librccl (same SONAMEs as ROCm's).  Whichever is loaded first serves the whole
        # process; if ROCm's came first, a later `import torch` finds no GPU.  So when torch exists it goes first and this
        # library, bench.py and torch.distributed all share one HIP runtime.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = _build.build()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: the HIP extension is required, there is no CPU fallback")
    L = ctypes.CDLL(path)
    vp, i32, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    L.dmi_default_options.argtypes = [ctypes.POINTER(OptionsC)]
    L.dmi_default_options.restype = None
    L.dmi_create.argtypes = [ctypes.POINTER(GridDescC), ctypes.POINTER(RayPotentialC), ctypes.POINTER(OptionsC),
                             ctypes.POINTER(vp)]
    L.dmi_destroy.argtypes = [vp]
    L.dmi_destroy.restype = None
    L.dmi_last_error.argtypes = [vp]
    L.dmi_last_error.restype = ctypes.c_char_p
    L.dmi_add_views.argtypes = [vp, dp, dp, dbl, dp, dp, i32, i32, i32]
    L.dmi_add_views_f32.argtypes = [vp, ctypes.POINTER(ctypes.c_float), dp, dp, i32, i32, i32]
    L.dmi_clear_views.argtypes = [vp]
    L.dmi_reset_grid.argtypes = [vp]
    L.dmi_upload_grid.argtypes = [vp, dp]
    L.dmi_fuse.argtypes = [vp]
    L.dmi_fuse_range.argtypes = [vp, i32, i32]
    L.dmi_fuse_slab.argtypes = [vp, i32, i32]
    if hasattr(L, "dmi_fuse_range_download"):  # (absent from an older prebuilt library loaded for an A/B timing, tools/gpu_exp.py)
        L.dmi_fuse_range_download.argtypes = [vp, i32, i32, vp, i32, i32]
    L.dmi_synchronize.argtypes = [vp]
    L.dmi_download_grid_f64.argtypes = [vp, dp]
    L.dmi_download_grid_f32.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.dmi_download_hits.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)]
    L.dmi_grid_device_pointer.argtypes = [vp, ctypes.POINTER(vp)]
    L.dmi_cell_to_point.argtypes = [vp]
    L.dmi_download_point_data_f64.argtypes = [vp, dp]
    L.dmi_point_data_device_pointer.argtypes = [vp, ctypes.POINTER(vp)]
    if hasattr(L, "dmi_iso_active_cells"):  # (an older build loaded for an A/B, tools/gpu_exp.py, lacks the newest entry points)
        L.dmi_iso_active_cells.argtypes = [vp, ctypes.c_double, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int64),
                                           ctypes.c_uint64]
    L.dmi_get_brick_class_histogram.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.dmi_get_mixed_reason_histogram.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    if hasattr(L, "dmi_get_window_pair_count"):  # (absent from an older prebuilt library loaded for an A/B timing, tools/gpu_exp.py)
        L.dmi_get_window_pair_count.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    if hasattr(L, "dmi_get_view_paths"):
        L.dmi_get_view_paths.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    if hasattr(L, "dmi_get_upload_kernel_ms"):
        L.dmi_get_upload_kernel_ms.argtypes = [vp, dp, dp]
    for name in ("dmi_sizeof_info", "dmi_sizeof_timings"):
        if hasattr(L, name):
            getattr(L, name).restype = ctypes.c_size_t
    L.dmi_get_timings.argtypes = [vp, ctypes.POINTER(TimingsC)]
    L.dmi_get_info.argtypes = [vp, ctypes.POINTER(InfoC)]
    L.dmi_alloc_pinned.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    L.dmi_free_pinned.argtypes = [vp]
    L.dmi_pcie_probe.argtypes = [i32, ctypes.c_size_t, dp, dp]
    L.dmi_fp64_probe.argtypes = [i32, ctypes.c_double, dp]
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L.dmi_color_mesh.argtypes = [dp, ctypes.c_int64, u8p, dp, dp, i32, i32, i32, i32, u8p, u8p, ctypes.POINTER(ctypes.c_int32)]
    L.dmi_color_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.dmi_color_destroy.argtypes = [vp]
    L.dmi_color_destroy.restype = None
    L.dmi_color_add_views.argtypes = [vp, u8p, dp, dp, i32, i32, i32]
    L.dmi_color_clear_views.argtypes = [vp]
    L.dmi_color_process.argtypes = [vp, dp, ctypes.c_int64, u8p, u8p, ctypes.POINTER(ctypes.c_int32)]
    L.dmi_color_get_kernel_ms.argtypes = [vp, dp]
    L.dmi_color_set_scratch_budget.argtypes = [vp, ctypes.c_uint64]
    L.dmi_color_set_vertex_reorder.argtypes = [vp, i32]
    L.dmi_color_last_error.argtypes = []
    L.dmi_color_last_error.restype = ctypes.c_char_p
    i64, i64p, i32p = ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)
    L.dmi_multi_default_options.argtypes = [ctypes.POINTER(MultiOptionsC)]
    L.dmi_multi_default_options.restype = None
    L.dmi_multi_view_shard.argtypes = [i64, i32, i32, i64p, i64p]
    L.dmi_multi_z_slab.argtypes = [i32, i32, i32, i32p, i32p]
    L.dmi_multi_slab_ranges.argtypes = [i32, i32, i32p, i32p, i32]
    if hasattr(L, "dmi_multi_peer_chunk"):
        L.dmi_multi_peer_chunk.argtypes = [i64, i32, i32, i64p, i64p]
    L.dmi_multi_create.argtypes = [ctypes.POINTER(GridDescC), ctypes.POINTER(RayPotentialC), ctypes.POINTER(MultiOptionsC),
                                   i32p, i32, ctypes.POINTER(vp)]
    L.dmi_multi_get_unique_id.argtypes = [u8p]
    L.dmi_multi_create_rank.argtypes = [ctypes.POINTER(GridDescC), ctypes.POINTER(RayPotentialC), ctypes.POINTER(MultiOptionsC),
                                        i32, i32, i32, u8p, ctypes.POINTER(vp)]
    L.dmi_multi_destroy.argtypes = [vp]
    L.dmi_multi_destroy.restype = None
    L.dmi_multi_last_error.argtypes = [vp]
    L.dmi_multi_last_error.restype = ctypes.c_char_p
    L.dmi_multi_add_views.argtypes = [vp, dp, dp, dbl, dp, dp, i32, i32, i32]
    L.dmi_multi_add_views_f32.argtypes = [vp, ctypes.POINTER(ctypes.c_float), dp, dp, i32, i32, i32]
    L.dmi_multi_add_local_views.argtypes = [vp, i32, dp, dp, dbl, dp, dp, i32, i32, i32]
    L.dmi_multi_add_local_views_f32.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float), dp, dp, i32, i32, i32]
    L.dmi_multi_clear_views.argtypes = [vp]
    L.dmi_multi_fuse.argtypes = [vp]
    L.dmi_multi_synchronize.argtypes = [vp]
    L.dmi_multi_download_grid_f32.argtypes = [vp, ctypes.POINTER(ctypes.c_float), i64p, i64p]
    L.dmi_multi_download_grid_f64.argtypes = [vp, dp, i64p, i64p]
    L.dmi_multi_get_info.argtypes = [vp, ctypes.POINTER(MultiInfoC)]
    L.dmi_multi_get_timings.argtypes = [vp, ctypes.POINTER(MultiTimingsC)]
    L.dmi_multi_local_context.argtypes = [vp, i32, ctypes.POINTER(vp)]
    from . import build as _b
    for name in ABI_SYMBOLS:
        if _b.LIB_OVERRIDE and not hasattr(L, name):
            continue  # an older build loaded side by side for an A/B (tools/gpu_exp.py); the shipped library has them all
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int:
            fn.restype = ctypes.c_int
    _lib = L
    return L


def pinned_empty(shape, dtype) -> np.ndarray:
    """A numpy array over pinned host memory (dmi_alloc_pinned): hipMemcpyAsync from it is a DMA transfer that overlaps
    with kernels.  Never freed by this helper's user explicitly: the block lives until the process ends."""
    L = load()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = ctypes.c_void_p()
    rc = L.dmi_alloc_pinned(n, ctypes.byref(p))
    if rc != DMI_OK:
        raise DmiError(rc, "dmi_alloc_pinned failed")
    buf = (ctypes.c_char * n).from_address(p.value)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def pcie_probe(device: int = 0, n_bytes: int = 256 << 20) -> tuple[float, float]:
    """(host-to-device, device-to-host) GB/s of one pinned hipMemcpyAsync each way (dmi_pcie_probe)."""
    L = load()
    a, b = ctypes.c_double(), ctypes.c_double()
    rc = L.dmi_pcie_probe(int(device), int(n_bytes), ctypes.byref(a), ctypes.byref(b))
    if rc != DMI_OK:
        raise DmiError(rc, L.dmi_last_error(None).decode())
    return float(a.value), float(b.value)


def fp64_probe(device: int = 0, milliseconds: float = 20.0) -> float:
    """fp64 vector TFLOP/s the device sustains right now on independent v_fma_f64 chains (dmi_fp64_probe)."""
    L = load()
    a = ctypes.c_double()
    rc = L.dmi_fp64_probe(int(device), float(milliseconds), ctypes.byref(a))
    if rc != DMI_OK:
        raise DmiError(rc, L.dmi_last_error(None).decode())
    return float(a.value)


def device_count() -> int:
    return int(load().dmi_device_count())


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class FusionContext:
    """One fusion = one context (dmi_create ... dmi_destroy)."""

    def __init__(self, grid: GridDesc, ray: RayPotential, *, device: int = 0, grid_dtype: str = "f64",
                 depth_storage: str = "auto", count_hits: bool = False, kernel_variant: int = 0,
                 stream: int | None = None, external_grid: int | None = None, z_first: int = 0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        g = _grid_c(grid)
        r = RayPotentialC(float(ray.thickness), float(ray.rho), float(ray.eta), float(ray.delta))
        o = OptionsC()
        self._lib.dmi_default_options(ctypes.byref(o))
        o.device = device
        o.grid_dtype = {"f32": DMI_F32, "f64": DMI_F64}[grid_dtype]
        o.depth_storage = {"auto": DMI_DEPTH_AUTO, "f32": DMI_DEPTH_F32, "f64": DMI_DEPTH_F64}[depth_storage]
        o.count_hits = 1 if count_hits else 0
        o.kernel_variant = int(kernel_variant)
        o.z_first = int(z_first)
        o.stream = stream
        o.external_grid = external_grid
        self.grid = grid
        self.grid_dtype = grid_dtype
        self.n_voxels = grid.n_voxels
        rc = self._lib.dmi_create(ctypes.byref(g), ctypes.byref(r), ctypes.byref(o), ctypes.byref(self._h))
        if rc != DMI_OK:
            self._h = ctypes.c_void_p()
            raise DmiError(rc, self._lib.dmi_last_error(None).decode())

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != DMI_OK:
            raise DmiError(rc, self._lib.dmi_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.dmi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- views ---------------------------------------------------------------------------------
    def add_views(self, views: Views, threshold: float | None = None):
        n, H, W = views.depth.shape
        K4 = np.ascontiguousarray(views.K4, dtype=np.float64).reshape(n, 16)
        RT4 = np.ascontiguousarray(views.RT4, dtype=np.float64).reshape(n, 16)
        if views.depth.dtype == np.float32:
            if views.best_cost is not None and threshold is not None:
                raise ValueError("f32 depth upload takes already-thresholded depths")
            d = np.ascontiguousarray(views.depth)
            self._check(self._lib.dmi_add_views_f32(self._h, d.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                    _dp(K4), _dp(RT4), n, W, H))
            return
        d = np.ascontiguousarray(views.depth, dtype=np.float64)
        bc = None
        if views.best_cost is not None and threshold is not None:
            bc = np.ascontiguousarray(views.best_cost, dtype=np.float64)
        self._check(self._lib.dmi_add_views(self._h, _dp(d), _dp(bc) if bc is not None else None,
                                            float(threshold) if threshold is not None else 0.0,
                                            _dp(K4), _dp(RT4), n, W, H))

    def clear_views(self):
        self._check(self._lib.dmi_clear_views(self._h))

    # -- grid ----------------------------------------------------------------------------------
    def reset_grid(self):
        self._check(self._lib.dmi_reset_grid(self._h))

    def upload_grid(self, grid: np.ndarray):
        g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1)
        if g.size != self.n_voxels:
            raise ValueError("grid size does not match the context")
        self._check(self._lib.dmi_upload_grid(self._h, _dp(g)))

    def fuse(self, first: int | None = None, count: int | None = None):
        if first is None:
            self._check(self._lib.dmi_fuse(self._h))
        else:
            self._check(self._lib.dmi_fuse_range(self._h, int(first), int(count)))

    def fuse_slab(self, z_first: int, z_count: int):
        """Fuse every resident view into cell layers [z_first, z_first + z_count) (multiples of 32 cells)."""
        self._check(self._lib.dmi_fuse_slab(self._h, int(z_first), int(z_count)))

    def synchronize(self):
        self._check(self._lib.dmi_synchronize(self._h))

    def fuse_download(self, first: int, count: int, dtype=np.float64, out: np.ndarray | None = None, n_slabs: int = 8) -> np.ndarray:
        """Fuse views [first, first + count) and return the grid as [nz, ny, nx]: with `dtype` the grid's own type the grid is
        fused in `n_slabs` z-slabs and every slab is copied to the host under the fusion of the next ones
        (dmi_fuse_range_download); bit for bit what fuse(first, count) + download_grid(dtype) return."""
        nx, ny, nz = (int(c) for c in self.grid.cell_dims)
        if out is None:
            out = np.empty(self.n_voxels, dtype=dtype)
        if out.dtype != np.dtype(dtype) or out.size != self.n_voxels or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous array of n_voxels elements of the requested dtype")
        if not hasattr(self._lib, "dmi_fuse_range_download"):  # an older prebuilt library: the same bits, nothing overlapped
            if count > 0:
                self.fuse(int(first), int(count))
            return self.download_grid(dtype, out)
        self._check(self._lib.dmi_fuse_range_download(self._h, int(first), int(count), ctypes.c_void_p(out.ctypes.data),
                                                      DMI_F64 if np.dtype(dtype) == np.float64 else DMI_F32, int(n_slabs)))
        return out.reshape(nz, ny, nx)

    def download_grid(self, dtype=np.float64, out: np.ndarray | None = None) -> np.ndarray:
        """The grid as [nz, ny, nx]; `out` (flat, contiguous, e.g. from pinned_empty) receives it when given."""
        nx, ny, nz = (int(c) for c in self.grid.cell_dims)
        if out is None:
            out = np.empty(self.n_voxels, dtype=dtype)
        if out.dtype != np.dtype(dtype) or out.size != self.n_voxels or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous array of n_voxels elements of the requested dtype")
        if np.dtype(dtype) == np.float64:
            self._check(self._lib.dmi_download_grid_f64(self._h, _dp(out)))
        else:
            self._check(self._lib.dmi_download_grid_f32(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out.reshape(nz, ny, nx)

    def download_hits(self):
        nx, ny, nz = (int(c) for c in self.grid.cell_dims)
        vh = np.empty(self.n_voxels, dtype=np.uint32)
        mh = np.empty(self.info().n_views, dtype=np.uint64)
        self._check(self._lib.dmi_download_hits(self._h, vh.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                                mh.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return vh.reshape(nz, ny, nx), mh

    def grid_device_pointer(self) -> int:
        p = ctypes.c_void_p()
        self._check(self._lib.dmi_grid_device_pointer(self._h, ctypes.byref(p)))
        return int(p.value)

    def cell_to_point(self):
        """vtkCellDataToPointData of the grid on the device (asynchronous; Reconstruction/main.cxx:151-155)."""
        self._check(self._lib.dmi_cell_to_point(self._h))

    def download_point_data(self) -> np.ndarray:
        """The point-data form of the grid, [nz+1, ny+1, nx+1] f64."""
        nx, ny, nz = (int(c) for c in self.grid.cell_dims)
        out = np.empty((nz + 1) * (ny + 1) * (nx + 1), dtype=np.float64)
        self._check(self._lib.dmi_download_point_data_f64(self._h, _dp(out)))
        return out.reshape(nz + 1, ny + 1, nx + 1)

    def iso_active_cells(self, iso: float, ids: bool = True):
        """(count, ids): the cells whose corner point values straddle `iso` (dmi_iso_active_cells); ids ascending int64, or
        None when ids=False."""
        n = ctypes.c_uint64(0)
        self._check(self._lib.dmi_iso_active_cells(self._h, float(iso), ctypes.byref(n), None, 0))
        if not ids:
            return int(n.value), None
        out = np.empty(int(n.value), dtype=np.int64)
        if n.value:
            self._check(self._lib.dmi_iso_active_cells(self._h, float(iso), ctypes.byref(n),
                                                       out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), out.size))
        return int(n.value), out

    def brick_class_histogram(self) -> dict:
        """(brick, view) pairs of the last fuse by proven class (diagnostic)."""
        h = (ctypes.c_uint64 * 4)()
        self._check(self._lib.dmi_get_brick_class_histogram(self._h, h))
        return {"mixed": int(h[0]), "free": int(h[1]), "behind": int(h[2]), "skip": int(h[3])}

    def mixed_reason_histogram(self) -> dict:
        """Why the mixed (brick, view) pairs of the last fuse could not be proven uniform (diagnostic)."""
        h = (ctypes.c_uint64 * 8)()
        self._check(self._lib.dmi_get_mixed_reason_histogram(self._h, h))
        names = ["unspecified", "degenerate", "camera_plane", "image_border", "nan_depth", "sentinel_and_depth", "near_surface",
                 "free_or_no_depth"]
        return {n: int(h[i]) for i, n in enumerate(names)}

    def window_pair_count(self) -> int:
        """How many "free_or_no_depth" pairs of the last fuse were served from a window of validity bits (diagnostic)."""
        n = ctypes.c_uint64(0)
        if not hasattr(self._lib, "dmi_get_window_pair_count"):
            return 0
        self._check(self._lib.dmi_get_window_pair_count(self._h, ctypes.byref(n)))
        return int(n.value)

    def upload_kernel_ms(self) -> tuple:
        """(last, total) hipEvent milliseconds of the upload pass's kernels (dmi_get_upload_kernel_ms)."""
        if not hasattr(self._lib, "dmi_get_upload_kernel_ms"):
            return (0.0, 0.0)
        a, b = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.dmi_get_upload_kernel_ms(self._h, ctypes.byref(a), ctypes.byref(b)))
        return (float(a.value), float(b.value))

    def timings(self) -> TimingsC:
        t = TimingsC()
        self._check(self._lib.dmi_get_timings(self._h, ctypes.byref(t)))
        return t

    def info(self) -> InfoC:
        i = InfoC()
        self._check(self._lib.dmi_get_info(self._h, ctypes.byref(i)))
        return i

    def view_paths(self) -> dict:
        """How many resident views take which path (dmi_get_view_paths)."""
        names = ("general_kernel", "tiled_general_k", "tiled_fp64_selection", "tiled_tier1", "tiled_tier1_per_lane_margin", "with_window_record")
        if not hasattr(self._lib, "dmi_get_view_paths"):
            return {}
        a = (ctypes.c_uint64 * 6)()
        self._check(self._lib.dmi_get_view_paths(self._h, a))
        return {k: int(v) for k, v in zip(names, a)}


def _grid_c(grid: GridDesc) -> GridDescC:
    g = GridDescC()
    for i in range(3):
        g.cell_dims[i] = int(grid.cell_dims[i])
        g.origin[i] = float(grid.origin[i])
        g.spacing[i] = float(grid.spacing[i])
    gm = np.ascontiguousarray(grid.grid_matrix, dtype=np.float64).reshape(16)
    for i in range(16):
        g.grid_matrix[i] = gm[i]
    return g


def multi_view_shard(n: int, rank: int, world: int) -> tuple[int, int]:
    """[lo, hi) of n items for `rank` of `world` (dmi_multi_view_shard; needs no GPU)."""
    a, b = ctypes.c_int64(), ctypes.c_int64()
    if load().dmi_multi_view_shard(int(n), int(rank), int(world), ctypes.byref(a), ctypes.byref(b)) != DMI_OK:
        raise ValueError("rank/world out of range")
    return int(a.value), int(a.value + b.value)


def multi_z_slab(nz: int, rank: int, world: int) -> tuple[int, int]:
    """Cell layers [z0, z1) of `rank` under DMI_PARTITION_Z_SLABS (dmi_multi_z_slab; needs no GPU)."""
    a, b = ctypes.c_int32(), ctypes.c_int32()
    if load().dmi_multi_z_slab(int(nz), int(rank), int(world), ctypes.byref(a), ctypes.byref(b)) != DMI_OK:
        raise ValueError("rank/world out of range")
    return int(a.value), int(a.value + b.value)


def multi_peer_chunk(n: int, world: int, c: int) -> tuple[int, int]:
    """[first, first + count) of an n-element slab that rank c of a peer-copy exchange sums (dmi_multi_peer_chunk)."""
    a, b = ctypes.c_int64(), ctypes.c_int64()
    rc = load().dmi_multi_peer_chunk(int(n), int(world), int(c), ctypes.byref(a), ctypes.byref(b))
    if rc != DMI_OK:
        raise DmiError(rc, "dmi_multi_peer_chunk: invalid argument")
    return int(a.value), int(b.value)


def multi_slab_ranges(nz: int, n_slabs: int) -> list[tuple[int, int]]:
    """(z_first, z_count) of the z-slabs of the overlapped exchange (dmi_multi_slab_ranges; needs no GPU)."""
    a, b = (ctypes.c_int32 * 64)(), (ctypes.c_int32 * 64)()
    n = load().dmi_multi_slab_ranges(int(nz), int(n_slabs), a, b, 64)
    return [(int(a[i]), int(b[i])) for i in range(n)]


def multi_unique_id() -> bytes:
    """The 128-byte id rank 0 creates and the launcher hands to every rank (dmi_multi_get_unique_id)."""
    L = load()
    buf = (ctypes.c_uint8 * DMI_UNIQUE_ID_BYTES)()
    rc = L.dmi_multi_get_unique_id(buf)
    if rc != DMI_OK:
        raise DmiError(rc, L.dmi_multi_last_error(None).decode())
    return bytes(buf)


class MultiContext:
    """One fusion over several GPUs (dmi_multi_create / dmi_multi_create_rank ... dmi_multi_destroy).

    devices=[...]                      all ranks in this process
    rank=, world=, unique_id=, device= one rank per process (unique_id from multi_unique_id() on rank 0)"""

    def __init__(self, grid: GridDesc, ray: RayPotential, *, devices=None, rank: int | None = None, world: int | None = None,
                 unique_id: bytes | None = None, device: int = 0, grid_dtype: str = "f32", depth_storage: str = "auto",
                 kernel_variant: int = 0, partition: str = "views", exchange: str = "all_reduce", n_slabs: int = 0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        o = MultiOptionsC()
        self._lib.dmi_multi_default_options(ctypes.byref(o))
        o.grid_dtype = {"f32": DMI_F32, "f64": DMI_F64}[grid_dtype]
        o.depth_storage = {"auto": DMI_DEPTH_AUTO, "f32": DMI_DEPTH_F32, "f64": DMI_DEPTH_F64}[depth_storage]
        o.kernel_variant = int(kernel_variant)
        o.partition = {"views": DMI_PARTITION_VIEWS, "z_slabs": DMI_PARTITION_Z_SLABS}[partition]
        o.exchange = {"all_reduce": DMI_EXCHANGE_ALL_REDUCE, "reduce_scatter": DMI_EXCHANGE_REDUCE_SCATTER,
                      "peer_copy": DMI_EXCHANGE_PEER_COPY}[exchange]
        o.n_slabs = int(n_slabs)
        g = _grid_c(grid)
        r = RayPotentialC(float(ray.thickness), float(ray.rho), float(ray.eta), float(ray.delta))
        self.grid = grid
        self.grid_dtype = grid_dtype
        self.n_voxels = grid.n_voxels
        if devices is not None:
            d = (ctypes.c_int32 * len(devices))(*[int(x) for x in devices])
            rc = self._lib.dmi_multi_create(ctypes.byref(g), ctypes.byref(r), ctypes.byref(o), d, len(devices), ctypes.byref(self._h))
        else:
            if rank is None or world is None:
                raise ValueError("MultiContext needs devices=[...] or rank= and world=")
            idbuf = None
            if unique_id is not None:
                if len(unique_id) != DMI_UNIQUE_ID_BYTES:
                    raise ValueError("unique_id must be 128 bytes")
                idbuf = (ctypes.c_uint8 * DMI_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
            rc = self._lib.dmi_multi_create_rank(ctypes.byref(g), ctypes.byref(r), ctypes.byref(o), int(device), int(rank),
                                                 int(world), idbuf, ctypes.byref(self._h))
        if rc != DMI_OK:
            self._h = ctypes.c_void_p()
            raise DmiError(rc, self._lib.dmi_multi_last_error(None).decode())

    def _check(self, rc: int):
        if rc != DMI_OK:
            raise DmiError(rc, self._lib.dmi_multi_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.dmi_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_views(self, views: Views, threshold: float | None = None, local_index: int | None = None):
        """local_index None: the SAME batch on every rank; each rank takes its share (views partition) or all of it
        (z-slabs).  local_index i: views of local rank i alone (the caller has partitioned, dmi_multi_add_local_views)."""
        n, H, W = views.depth.shape
        K4 = np.ascontiguousarray(views.K4, dtype=np.float64).reshape(n, 16)
        RT4 = np.ascontiguousarray(views.RT4, dtype=np.float64).reshape(n, 16)
        L = self._lib
        if views.depth.dtype == np.float32:
            d = np.ascontiguousarray(views.depth)
            dptr = d.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            if local_index is None:
                self._check(L.dmi_multi_add_views_f32(self._h, dptr, _dp(K4), _dp(RT4), n, W, H))
            else:
                self._check(L.dmi_multi_add_local_views_f32(self._h, int(local_index), dptr, _dp(K4), _dp(RT4), n, W, H))
            return
        d = np.ascontiguousarray(views.depth, dtype=np.float64)
        bc = None
        if views.best_cost is not None and threshold is not None:
            bc = np.ascontiguousarray(views.best_cost, dtype=np.float64)
        thr = float(threshold) if threshold is not None else 0.0
        if local_index is None:
            self._check(L.dmi_multi_add_views(self._h, _dp(d), _dp(bc) if bc is not None else None, thr, _dp(K4), _dp(RT4), n, W, H))
        else:
            self._check(L.dmi_multi_add_local_views(self._h, int(local_index), _dp(d), _dp(bc) if bc is not None else None, thr,
                                                    _dp(K4), _dp(RT4), n, W, H))

    def clear_views(self):
        self._check(self._lib.dmi_multi_clear_views(self._h))

    def fuse(self):
        self._check(self._lib.dmi_multi_fuse(self._h))

    def synchronize(self):
        self._check(self._lib.dmi_multi_synchronize(self._h))

    def download_grid(self, dtype=np.float32, out: np.ndarray | None = None):
        """(grid as [nz, ny, nx], (first, count)): the element range of the flat grid this process's ranks own."""
        nx, ny, nz = (int(c) for c in self.grid.cell_dims)
        if out is None:
            out = np.zeros(self.n_voxels, dtype=dtype)
        a, b = ctypes.c_int64(), ctypes.c_int64()
        if np.dtype(dtype) == np.float64:
            self._check(self._lib.dmi_multi_download_grid_f64(self._h, _dp(out), ctypes.byref(a), ctypes.byref(b)))
        else:
            self._check(self._lib.dmi_multi_download_grid_f32(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                              ctypes.byref(a), ctypes.byref(b)))
        return out.reshape(nz, ny, nx), (int(a.value), int(b.value))

    def info(self) -> MultiInfoC:
        i = MultiInfoC()
        self._check(self._lib.dmi_multi_get_info(self._h, ctypes.byref(i)))
        return i

    def timings(self) -> MultiTimingsC:
        t = MultiTimingsC()
        self._check(self._lib.dmi_multi_get_timings(self._h, ctypes.byref(t)))
        return t

    def local_timings(self, local_index: int = 0) -> TimingsC:
        """dmi_timings of one local rank's single-GPU context."""
        h = ctypes.c_void_p()
        self._check(self._lib.dmi_multi_local_context(self._h, int(local_index), ctypes.byref(h)))
        t = TimingsC()
        if h:
            rc = self._lib.dmi_get_timings(h, ctypes.byref(t))
            if rc != DMI_OK:
                raise DmiError(rc, self._lib.dmi_last_error(h).decode())
        return t

    def local_context_grid(self, local_index: int, dtype=np.float32) -> np.ndarray:
        """The grid as local rank `local_index` holds it (after an all-reduce exchange every rank holds the sums)."""
        h = ctypes.c_void_p()
        self._check(self._lib.dmi_multi_local_context(self._h, int(local_index), ctypes.byref(h)))
        nx, ny, nz = (int(c) for c in self.grid.cell_dims)
        out = np.zeros(self.n_voxels, dtype=dtype)
        if np.dtype(dtype) == np.float64:
            rc = self._lib.dmi_download_grid_f64(h, _dp(out))
        else:
            rc = self._lib.dmi_download_grid_f32(h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        if rc != DMI_OK:
            raise DmiError(rc, self._lib.dmi_last_error(h).decode())
        return out.reshape(nz, ny, nx)

    def local_info(self, local_index: int = 0) -> InfoC:
        h = ctypes.c_void_p()
        self._check(self._lib.dmi_multi_local_context(self._h, int(local_index), ctypes.byref(h)))
        i = InfoC()
        if h:
            self._lib.dmi_get_info(h, ctypes.byref(i))
        return i


def color_mesh(points, colors, K4, RT4, device: int = 0):
    """MeshColoration::ProcessColoration on the GPU (include/dmi.h: dmi_color_mesh).
    points [nv,3] f64; colors [n,H,W,3] u8 in vtk row order; returns (mean u8[nv,3], median u8[nv,3], count i32[nv])."""
    L = load()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    col = np.ascontiguousarray(colors, dtype=np.uint8)
    n, H, W, _ = col.shape
    k = np.ascontiguousarray(K4, dtype=np.float64).reshape(n, 16)
    rt = np.ascontiguousarray(RT4, dtype=np.float64).reshape(n, 16)
    nv = pts.shape[0]
    mean = np.zeros((nv, 3), dtype=np.uint8)
    median = np.zeros((nv, 3), dtype=np.uint8)
    count = np.zeros(nv, dtype=np.int32)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    rc = L.dmi_color_mesh(_dp(pts), nv, col.ctypes.data_as(u8), _dp(k), _dp(rt), n, W, H, device, mean.ctypes.data_as(u8),
                          median.ctypes.data_as(u8), count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if rc != DMI_OK:
        raise DmiError(rc, L.dmi_color_last_error().decode())
    return mean, median, count


class ColorContext:
    """MeshColoration with resident views (dmi_color_create ... dmi_color_destroy)."""

    def __init__(self, device: int = 0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        rc = self._lib.dmi_color_create(device, ctypes.byref(self._h))
        if rc != DMI_OK:
            self._h = ctypes.c_void_p()
            raise DmiError(rc, self._lib.dmi_color_last_error().decode())

    def _check(self, rc):
        if rc != DMI_OK:
            raise DmiError(rc, self._lib.dmi_color_last_error().decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.dmi_color_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_views(self, colors, K4, RT4):
        col = np.ascontiguousarray(colors, dtype=np.uint8)
        n, H, W, _ = col.shape
        k = np.ascontiguousarray(K4, dtype=np.float64).reshape(n, 16)
        rt = np.ascontiguousarray(RT4, dtype=np.float64).reshape(n, 16)
        self._check(self._lib.dmi_color_add_views(self._h, col.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _dp(k), _dp(rt),
                                                  n, W, H))

    def clear_views(self):
        self._check(self._lib.dmi_color_clear_views(self._h))

    def process(self, points, out=None):
        """(mean [n, 3] u8, median [n, 3] u8, count [n] i32) of the vertices `points` [n, 3] f64.  out: the three arrays to fill
        (e.g. over pinned memory, pinned_empty -- with `points` pinned too the copies are DMA transfers that overlap the kernels of
        the neighbouring chunks); else fresh ones."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        nv = pts.shape[0]
        if out is None:
            mean = np.zeros((nv, 3), dtype=np.uint8)
            median = np.zeros((nv, 3), dtype=np.uint8)
            count = np.zeros(nv, dtype=np.int32)
        else:
            mean, median, count = out
            if (mean.shape, median.shape, count.shape) != ((nv, 3), (nv, 3), (nv,)) or mean.dtype != np.uint8 or median.dtype != np.uint8 \
                    or count.dtype != np.int32 or not (mean.flags.c_contiguous and median.flags.c_contiguous and count.flags.c_contiguous):
                raise ValueError("out = (mean [n, 3] u8, median [n, 3] u8, count [n] i32), contiguous")
        u8 = ctypes.POINTER(ctypes.c_uint8)
        self._check(self._lib.dmi_color_process(self._h, _dp(pts), nv, mean.ctypes.data_as(u8), median.ctypes.data_as(u8),
                                                count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return mean, median, count

    def set_scratch_budget(self, n_bytes: int):
        """Bound the device scratch of one vertex chunk (more, smaller chunks; same result)."""
        self._check(self._lib.dmi_color_set_scratch_budget(self._h, int(n_bytes)))

    def set_vertex_reorder(self, enable: bool):
        """Work through each chunk's vertices along a Z-order curve (same results, better gathers for unordered vertices)."""
        self._check(self._lib.dmi_color_set_vertex_reorder(self._h, 1 if enable else 0))

    def kernel_ms(self) -> float:
        v = ctypes.c_double(0)
        self._check(self._lib.dmi_color_get_kernel_ms(self._h, ctypes.byref(v)))
        return float(v.value)


def fuse_once(grid: GridDesc, ray: RayPotential, views: Views, *, threshold: float | None = None,
              init_grid: np.ndarray | None = None, count_hits: bool = True, **ctx_kwargs):
    """create -> (upload grid) -> add views -> fuse -> download.  Returns (grid, voxel_hits, map_hits)."""
    with FusionContext(grid, ray, count_hits=count_hits, **ctx_kwargs) as ctx:
        if init_grid is not None:
            ctx.upload_grid(init_grid)
        ctx.add_views(views, threshold)
        ctx.fuse()
        out = ctx.download_grid(np.float64)
        vh, mh = ctx.download_hits() if count_hits else (None, None)
    return out, vh, mh


# ---- host-side mirror of the reference's operator interface (include/dmi_host.h) -----------------------
HOST_ABI_SYMBOLS = [
    "dmi_filter_new", "dmi_filter_delete", "dmi_filter_set_ray_potential_thickness", "dmi_filter_set_ray_potential_rho",
    "dmi_filter_set_ray_potential_eta", "dmi_filter_set_ray_potential_delta", "dmi_filter_set_threshold_best_cost",
    "dmi_filter_set_file_path_krtd", "dmi_filter_set_file_path_vti", "dmi_filter_set_grid_matrix",
    "dmi_filter_set_input_data", "dmi_filter_add_view", "dmi_filter_clear_views", "dmi_filter_set_device",
    "dmi_filter_set_kernel_variant", "dmi_filter_set_devices", "dmi_filter_set_partition", "dmi_filter_set_host_chunk_bytes", "dmi_filter_set_fill_on_calling_thread", "dmi_filter_update", "dmi_filter_get_execution_time",
    "dmi_filter_get_fuse_kernel_ms", "dmi_filter_get_number_of_cells", "dmi_filter_get_output",
    "dmi_filter_last_error", "dmi_read_krtd_file", "dmi_extract_all_file_path", "dmi_k3_to_k4",
    "dmi_apply_depth_threshold", "dmi_read_depth_map", "dmi_read_depth_map_color", "dmi_mesh_coloration_from_lists",
    "dmi_cli_read_arguments", "dmi_cli_main",
]

_host_bound = False


def load_host() -> ctypes.CDLL:
    """Bind the dmi_host.h entry points of the same shared library."""
    global _host_bound
    L = load()
    if _host_bound:
        return L
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int32)
    L.dmi_filter_new.restype = vp
    L.dmi_filter_new.argtypes = []
    L.dmi_filter_delete.restype = None
    L.dmi_filter_delete.argtypes = [vp]
    for name in ("thickness", "rho", "eta", "delta"):
        fn = getattr(L, f"dmi_filter_set_ray_potential_{name}")
        fn.restype, fn.argtypes = None, [vp, dbl]
    L.dmi_filter_set_threshold_best_cost.restype, L.dmi_filter_set_threshold_best_cost.argtypes = None, [vp, dbl]
    L.dmi_filter_set_file_path_krtd.restype, L.dmi_filter_set_file_path_krtd.argtypes = None, [vp, ctypes.c_char_p]
    L.dmi_filter_set_file_path_vti.restype, L.dmi_filter_set_file_path_vti.argtypes = None, [vp, ctypes.c_char_p]
    L.dmi_filter_set_grid_matrix.restype, L.dmi_filter_set_grid_matrix.argtypes = None, [vp, dp]
    L.dmi_filter_set_input_data.restype, L.dmi_filter_set_input_data.argtypes = None, [vp, ip, dp, dp]
    L.dmi_filter_add_view.restype, L.dmi_filter_add_view.argtypes = ctypes.c_int, [vp, dp, dp, i32, i32, dp, dp]
    L.dmi_filter_clear_views.restype, L.dmi_filter_clear_views.argtypes = None, [vp]
    L.dmi_filter_set_device.restype, L.dmi_filter_set_device.argtypes = None, [vp, i32]
    L.dmi_filter_set_kernel_variant.restype, L.dmi_filter_set_kernel_variant.argtypes = None, [vp, i32]
    L.dmi_filter_set_devices.restype, L.dmi_filter_set_devices.argtypes = None, [vp, ip, i32]
    L.dmi_filter_set_partition.restype, L.dmi_filter_set_partition.argtypes = None, [vp, i32]
    L.dmi_filter_set_host_chunk_bytes.restype, L.dmi_filter_set_host_chunk_bytes.argtypes = None, [vp, ctypes.c_uint64]
    L.dmi_filter_set_fill_on_calling_thread.restype, L.dmi_filter_set_fill_on_calling_thread.argtypes = None, [vp, i32]
    L.dmi_filter_update.restype, L.dmi_filter_update.argtypes = ctypes.c_int, [vp]
    L.dmi_filter_get_execution_time.restype, L.dmi_filter_get_execution_time.argtypes = dbl, [vp]
    L.dmi_filter_get_fuse_kernel_ms.restype, L.dmi_filter_get_fuse_kernel_ms.argtypes = dbl, [vp]
    L.dmi_filter_get_number_of_cells.restype, L.dmi_filter_get_number_of_cells.argtypes = i64, [vp]
    L.dmi_filter_get_output.restype, L.dmi_filter_get_output.argtypes = i64, [vp, dp]
    L.dmi_filter_last_error.restype, L.dmi_filter_last_error.argtypes = ctypes.c_char_p, [vp]
    L.dmi_read_krtd_file.restype, L.dmi_read_krtd_file.argtypes = ctypes.c_int, [ctypes.c_char_p, dp, dp]
    L.dmi_extract_all_file_path.restype = ctypes.c_int
    L.dmi_extract_all_file_path.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.dmi_k3_to_k4.restype, L.dmi_k3_to_k4.argtypes = None, [dp, dp]
    L.dmi_apply_depth_threshold.restype, L.dmi_apply_depth_threshold.argtypes = i64, [dp, dp, i64, dbl]
    L.dmi_read_depth_map.restype, L.dmi_read_depth_map.argtypes = ctypes.c_int, [ctypes.c_char_p, ip, dp, dp, ip]
    L.dmi_read_depth_map_color.restype = ctypes.c_int
    L.dmi_read_depth_map_color.argtypes = [ctypes.c_char_p, ip, ctypes.POINTER(ctypes.c_uint8), ip]
    L.dmi_mesh_coloration_from_lists.restype = ctypes.c_int
    L.dmi_mesh_coloration_from_lists.argtypes = [dp, i64, ctypes.c_char_p, ctypes.c_char_p, i32, ctypes.POINTER(ctypes.c_uint8),
                                                 ctypes.POINTER(ctypes.c_uint8), ip, ctypes.c_char_p, ctypes.c_size_t]
    _host_bound = True
    return L


class ReconstructionFilter:
    """The reference's vtkCudaReconstructionFilter (filt.h:48-120) through the host mirror: same setter
    names, Update() returns RequestData's 1 / 0, the output is the "reconstruction_scalar" cell array."""

    def __init__(self):
        self._lib = load_host()
        self._h = self._lib.dmi_filter_new()
        if not self._h:
            raise MemoryError("dmi_filter_new failed")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dmi_filter_delete(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def SetRayPotentialThickness(self, v): self._lib.dmi_filter_set_ray_potential_thickness(self._h, float(v))
    def SetRayPotentialRho(self, v): self._lib.dmi_filter_set_ray_potential_rho(self._h, float(v))
    def SetRayPotentialEta(self, v): self._lib.dmi_filter_set_ray_potential_eta(self._h, float(v))
    def SetRayPotentialDelta(self, v): self._lib.dmi_filter_set_ray_potential_delta(self._h, float(v))
    def SetThresholdBestCost(self, v): self._lib.dmi_filter_set_threshold_best_cost(self._h, float(v))

    def SetFilePathKRTD(self, path):
        self._lib.dmi_filter_set_file_path_krtd(self._h, None if path is None else os.fsencode(path))

    def SetFilePathVTI(self, path):
        self._lib.dmi_filter_set_file_path_vti(self._h, None if path is None else os.fsencode(path))

    def SetGridMatrix(self, m):
        if m is None:
            self._lib.dmi_filter_set_grid_matrix(self._h, None)
            return
        a = np.ascontiguousarray(m, dtype=np.float64).reshape(16)
        self._lib.dmi_filter_set_grid_matrix(self._h, _dp(a))

    def SetInputData(self, point_dims, origin, spacing):
        d = (ctypes.c_int32 * 3)(*[int(x) for x in point_dims])
        o = np.ascontiguousarray(origin, dtype=np.float64)
        s = np.ascontiguousarray(spacing, dtype=np.float64)
        self._point_dims = tuple(int(x) for x in point_dims)
        self._lib.dmi_filter_set_input_data(self._h, d, _dp(o), _dp(s))

    def AddView(self, depths, K3, RT4, best_cost=None):
        d = np.ascontiguousarray(depths, dtype=np.float64)
        H, W = d.shape
        bc = None if best_cost is None else np.ascontiguousarray(best_cost, dtype=np.float64)
        k = np.ascontiguousarray(K3, dtype=np.float64).reshape(9)
        rt = np.ascontiguousarray(RT4, dtype=np.float64).reshape(16)
        if not self._lib.dmi_filter_add_view(self._h, _dp(d), _dp(bc) if bc is not None else None, W, H, _dp(k), _dp(rt)):
            raise ValueError("dmi_filter_add_view rejected the view")

    def ClearViews(self): self._lib.dmi_filter_clear_views(self._h)
    def SetDevice(self, d): self._lib.dmi_filter_set_device(self._h, int(d))
    def SetKernelVariant(self, v): self._lib.dmi_filter_set_kernel_variant(self._h, int(v))

    def SetDevices(self, devices):
        d = (ctypes.c_int32 * len(devices))(*[int(x) for x in devices])
        self._lib.dmi_filter_set_devices(self._h, d, len(devices))

    def SetPartition(self, partition: str):
        self._lib.dmi_filter_set_partition(self._h, {"views": DMI_PARTITION_VIEWS, "z_slabs": DMI_PARTITION_Z_SLABS}[partition])

    def SetHostChunkBytes(self, n): self._lib.dmi_filter_set_host_chunk_bytes(self._h, int(n))
    def SetFillOnCallingThread(self, yes): self._lib.dmi_filter_set_fill_on_calling_thread(self._h, 1 if yes else 0)
    def Update(self) -> int: return int(self._lib.dmi_filter_update(self._h))
    def GetExecutionTime(self) -> float: return float(self._lib.dmi_filter_get_execution_time(self._h))
    def GetFuseKernelMs(self) -> float: return float(self._lib.dmi_filter_get_fuse_kernel_ms(self._h))
    def GetNumberOfCells(self) -> int: return int(self._lib.dmi_filter_get_number_of_cells(self._h))
    def LastError(self) -> str: return self._lib.dmi_filter_last_error(self._h).decode()

    def GetOutputScalars(self) -> np.ndarray:
        """The "reconstruction_scalar" cell array as [nz, ny, nx] (x fastest, filt.cxx:129-135)."""
        n = self.GetNumberOfCells()
        out = np.zeros(n, dtype=np.float64)
        got = self._lib.dmi_filter_get_output(self._h, _dp(out))
        if got != n:
            return out[:got]
        px, py, pz = self._point_dims
        return out.reshape(pz - 1, py - 1, px - 1)


class CliOptionsC(ctypes.Structure):
    _fields_ = [("grid_dims", ctypes.c_int32 * 3), ("grid_spacing", ctypes.c_double * 3), ("grid_origin", ctypes.c_double * 3),
                ("grid_end", ctypes.c_double * 3), ("grid_matrix", ctypes.c_double * 16), ("ray_thick", ctypes.c_double),
                ("ray_rho", ctypes.c_double), ("ray_eta", ctypes.c_double), ("ray_delta", ctypes.c_double),
                ("thresh_best_cost", ctypes.c_double), ("contour", ctypes.c_double), ("verbose", ctypes.c_int32),
                ("summary", ctypes.c_int32), ("force_cubic_voxel", ctypes.c_int32)]


def cli_read_arguments(args):
    """ReadArguments of the `Reconstruction` tool (Reconstruction/main.cxx:216-343) on ["prog", "--flag", ...]:
    (options or None, the text the tool would print)."""
    L = load_host()
    L.dmi_cli_read_arguments.restype = ctypes.c_int
    L.dmi_cli_read_arguments.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(CliOptionsC),
                                         ctypes.c_char_p, ctypes.c_size_t]
    argv = (ctypes.c_char_p * len(args))(*[os.fsencode(a) for a in args])
    out = CliOptionsC()
    err = ctypes.create_string_buffer(1 << 15)
    ok = L.dmi_cli_read_arguments(len(args), argv, ctypes.byref(out), err, len(err))
    return (out if ok else None), err.value.decode()


def cli_binary() -> str:
    """Path of the dmi_reconstruction executable next to the library (linked now if the build has not done so: it needs
    hipcc, which a box that only runs a prebuilt library may lack -- loading the library never depends on it)."""
    from . import build as _build
    # always through build_cli(): it returns at once when the executable's digest names the library's sources, and relinks one
    # left over from older sources (an older ABI's main against this library) -- only without hipcc is an existing file taken as is
    try:
        _build.build_cli()
    except Exception:
        if not os.path.exists(_build.CLI_PATH):
            raise
    return _build.CLI_PATH


def read_krtd_file(path):
    L = load_host()
    K = np.zeros(9)
    RT = np.zeros(16)
    ok = L.dmi_read_krtd_file(os.fsencode(path), _dp(K), _dp(RT))
    return bool(ok), K.reshape(3, 3), RT.reshape(4, 4)


def extract_all_file_path(list_path):
    L = load_host()
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.dmi_extract_all_file_path(os.fsencode(list_path), buf, len(buf))
    paths = buf.value.decode().split("\n") if n else []
    return paths


def k3_to_k4(K3):
    L = load_host()
    k = np.ascontiguousarray(K3, dtype=np.float64).reshape(9)
    out = np.zeros(16)
    L.dmi_k3_to_k4(_dp(k), _dp(out))
    return out.reshape(4, 4)


def apply_depth_threshold(depths, best_cost, threshold):
    L = load_host()
    d = np.ascontiguousarray(depths, dtype=np.float64).copy()
    b = np.ascontiguousarray(best_cost, dtype=np.float64)
    changed = L.dmi_apply_depth_threshold(_dp(d.reshape(-1)), _dp(b.reshape(-1)), d.size, float(threshold))
    return d, int(changed)


def read_depth_map(path):
    L = load_host()
    dims = (ctypes.c_int32 * 3)()
    has = ctypes.c_int32(0)
    if not L.dmi_read_depth_map(os.fsencode(path), dims, None, None, ctypes.byref(has)):
        return None
    n = dims[0] * dims[1] * dims[2]
    d = np.zeros(n)
    bc = np.zeros(n)
    L.dmi_read_depth_map(os.fsencode(path), dims, _dp(d), _dp(bc), ctypes.byref(has))
    shape = (dims[1], dims[0])
    return d.reshape(shape), (bc.reshape(shape) if has.value else None)


def read_depth_map_color(path):
    """The "Color" array of a depth-map .vti as [H, W, 3] u8 (vtk row order), or None when the file has none."""
    L = load_host()
    dims = (ctypes.c_int32 * 3)()
    has = ctypes.c_int32(0)
    if not L.dmi_read_depth_map_color(os.fsencode(path), dims, None, ctypes.byref(has)) or not has.value:
        return None
    c = np.zeros(dims[0] * dims[1] * dims[2] * 3, dtype=np.uint8)
    L.dmi_read_depth_map_color(os.fsencode(path), dims, c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(has))
    return c.reshape(dims[1], dims[0], 3)


def mesh_coloration_from_lists(points, vti_list, krtd_list, device: int = 0):
    """MeshColoration(mesh, vtiList, krtdList).ProcessColoration() through the host mirror (reads the .vti/.krtd files)."""
    L = load_host()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    nv = pts.shape[0]
    mean = np.zeros((nv, 3), dtype=np.uint8)
    median = np.zeros((nv, 3), dtype=np.uint8)
    count = np.zeros(nv, dtype=np.int32)
    err = ctypes.create_string_buffer(512)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    ok = L.dmi_mesh_coloration_from_lists(_dp(pts), nv, os.fsencode(vti_list), os.fsencode(krtd_list), device,
                                          mean.ctypes.data_as(u8), median.ctypes.data_as(u8),
                                          count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), err, len(err))
    if not ok:
        raise RuntimeError(err.value.decode())
    return mean, median, count
