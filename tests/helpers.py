"""Shared test helpers: adapt golden fixtures / scenes to the oracle's argument lists."""
import numpy as np

from oracle import oracle


def oracle_params_from_golden(g):
    n, H, W = g["depth"].shape
    t, rho, eta, delta = (float(x) for x in g["ray"])
    return oracle.make_params(g["cell_dims"], g["origin"], g["spacing"], g["grid_matrix"], t, rho, eta, delta, W, H)


def thresholded_depth(g):
    """Depth table as the kernel sees it: best-cost filter applied first (cu:348, RD.cxx:138-167)."""
    if "best_cost" in g:
        return oracle.apply_depth_threshold(g["depth"], g["best_cost"], float(g["threshold"])).reshape(g["depth"].shape)
    return g["depth"]


def oracle_params_from_scene(grid, rp, views):
    return oracle.make_params(grid.cell_dims, grid.origin, grid.spacing, grid.grid_matrix,
                              rp.thickness, rp.rho, rp.eta, rp.delta, views.width, views.height)


def bits_equal(a, b):
    """Bit-exact fp64 comparison.  A NaN must be matched by a NaN, but its sign and payload bits are not compared:
    they are a property of the platform that produced it (x86 gives inf*0 the sign bit, gfx950 does not), and
    the reference itself yields NaN in these places (NaN depths; thickness 0 with diff == 0, cu:119)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(both_nan | (a.view(np.uint64) == b.view(np.uint64))))


# ---- brick classes on the tiny grids the tests use -------------------------------------------------------------------
# The library fuses launches of up to 1024 bricks without brick classes (dmi_capi.hip: classifying them costs more than it
# saves).  Most GPU tests use such grids BECAUSE the oracle is quick on them, and many exist to exercise the classes: a fixture
# in conftest.py therefore adds VARIANT_BRICK_CLASSES_ALWAYS to every context a test creates, unless the test asks for the
# shipped behaviour with `with shipped_defaults():` (tests/test_gpu_parity.py covers the rule itself).
import contextlib

_classes_always = {"on": True}


@contextlib.contextmanager
def shipped_defaults():
    _classes_always["on"] = False
    try:
        yield
    finally:
        _classes_always["on"] = True


def variant_for_tests(kernel_variant: int) -> int:
    from cudadepthmapintegration_amd import capi
    if _classes_always["on"] and not (kernel_variant & capi.VARIANT_NO_BRICK_CLASSES):
        return kernel_variant | capi.VARIANT_BRICK_CLASSES_ALWAYS
    return kernel_variant
