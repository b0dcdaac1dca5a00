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
    """Bit-exact fp64 comparison that treats equal NaN payloads as equal."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))
