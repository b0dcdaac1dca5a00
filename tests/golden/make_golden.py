#!/usr/bin/env python3
"""Mint the committed golden fixtures under tests/golden/*.npz.

PARITY UNPINNED: the reference has no golden vectors and cannot be built or run
in this image (it needs nvcc + CUDA runtime + VTK), so these vectors are
produced by this project's own restatement of the reference arithmetic
(oracle/tsdf_oracle.c), and every case is required to agree bit for bit with the
independently written numpy restatement (oracle/oracle_np.py) before it is
written.  They freeze the oracle's behaviour so that the HIP path, the C oracle
and the numpy oracle are all held to the same committed numbers.

Each fixture stores inputs AND expected outputs, so the tests do not depend on
the scene generator staying unchanged.

    python tests/golden/make_golden.py          # rewrites every fixture
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cudadepthmapintegration_amd import scene  # noqa: E402
from oracle import oracle, oracle_np  # noqa: E402


def _case(name, grid, rp, views, threshold=None, init_grid=None, note=""):
    depth = views.depth
    if views.best_cost is not None and threshold is not None:
        # Sources/ReconstructionData.cxx:138-167, applied before the kernel (cu:348)
        depth_used = oracle.apply_depth_threshold(depth, views.best_cost, threshold).reshape(depth.shape)
    else:
        depth_used = depth
    n, H, W = depth.shape
    p = oracle.make_params(grid.cell_dims, grid.origin, grid.spacing, grid.grid_matrix,
                           rp.thickness, rp.rho, rp.eta, rp.delta, W, H)
    g, vh, mh = oracle.fuse(p, depth_used, views.K4, views.RT4, init_grid=init_grid)
    g2, vh2, mh2 = oracle_np.fuse(grid.cell_dims, grid.origin, grid.spacing, grid.grid_matrix,
                                  rp.thickness, rp.rho, rp.eta, rp.delta, depth_used, views.K4, views.RT4,
                                  init_grid=init_grid)
    same = (g.tobytes() == g2.tobytes()) or np.array_equal(g, g2, equal_nan=True)
    assert same and np.array_equal(vh, vh2) and np.array_equal(mh, mh2), f"{name}: C and numpy oracles disagree"
    out = dict(
        cell_dims=np.asarray(grid.cell_dims, dtype=np.int32),
        origin=np.asarray(grid.origin, dtype=np.float64),
        spacing=np.asarray(grid.spacing, dtype=np.float64),
        grid_matrix=np.asarray(grid.grid_matrix, dtype=np.float64),
        ray=np.asarray([rp.thickness, rp.rho, rp.eta, rp.delta], dtype=np.float64),
        depth=depth, K4=views.K4, RT4=views.RT4,
        expected_grid=g, expected_voxel_hits=vh, expected_map_hits=mh,
        note=np.asarray(note),
    )
    if views.best_cost is not None and threshold is not None:
        out["best_cost"] = views.best_cost
        out["threshold"] = np.asarray(threshold, dtype=np.float64)
    if init_grid is not None:
        out["init_grid"] = np.asarray(init_grid, dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    rate = mh.sum() / (grid.n_voxels * n)
    print(f"{name:28s} {grid.cell_dims} x {n} maps {W}x{H}  hit-rate {rate:.3f}  "
          f"range [{np.nanmin(g):.4f}, {np.nanmax(g):.4f}]  {os.path.getsize(path) / 1024:.0f} KiB")


def branch_census(grid, rp, views):
    """How many voxel x map pairs land in each ray-potential branch (fixture 1 must cover all)."""
    p = oracle.make_params(grid.cell_dims, grid.origin, grid.spacing, grid.grid_matrix,
                           rp.thickness, rp.rho, rp.eta, rp.delta, views.width, views.height)
    vals = set()
    for m in range(views.n):
        g, _, _ = oracle.fuse(p, views.depth[m:m + 1], views.K4[m:m + 1], views.RT4[m:m + 1])
        vals.update(np.unique(np.round(g, 12)).tolist())
    return vals


def engineered_edges():
    """Pixel-centre and sentinel edge cases with exactly representable numbers.

    Grid origin 0, spacing 1, identity grid matrix: voxel centres are (i+.5, j+.5, k+.5).
    """
    grid = scene.GridDesc((8, 6, 3), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), np.eye(4))
    rp = scene.RayPotential(thickness=0.5, rho=0.8, eta=0.03, delta=1.5)
    W, H = 8, 6
    maps = []

    def add(fx, fy, cx, cy, tx, ty, tz, depth_value):
        K = np.eye(4)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
        RT = np.eye(4)
        RT[:3, 3] = (tx, ty, tz)
        d = np.full((H, W), depth_value, dtype=np.float64)
        maps.append((d, K, RT))

    # z_cam = k + 0.5 (tz = 0).  For the k = 0 layer u = 2 f x + cx.
    add(0.5, 0.5, 0.0, 0.0, 0, 0, 0, 0.5)      # k=0: u = i+0.5, v = j+0.5: exact halves -> i+1, j+1; diff == 0 at k=0
    add(0.5, 0.5, 0.0, 0.0, -2, -1, 0, 1.5)    # k=0: u = i-1.5 -> -2 (out), -0.5 -> -1 (out), 0.5 -> 1 ...
    add(0.3, 0.3, 0.0, 0.0, -1, -1, 0, 2.5)    # u in (-0.5, 0) for i = 0: rounds to -0 -> pixel 0 (in)
    add(0.5, 0.5, 7.0, 0.0, -0.5, 0, 0, 0.5)   # k=0: u = i + 7 -> reaches W-0.5+... beyond W-1 (out on the right)
    add(0.5, 0.5, 0.0, 0.0, 0, 0, -0.5, 1.0)   # z_cam = k: k=0 layer has h.z == 0 (x/0 = inf, out by the project rule)
    add(0.5, 0.5, 0.0, 0.0, 0, 0, -1.5, 1.0)   # z_cam = k-1: k=0 behind the camera (h.z < 0, cu:177), k=1 is z == 0
    add(1.0, 1.0, 0.0, 0.0, -0.5, -0.5, 0.5, -1.0)  # all-sentinel map: no voxel may hit (cu:202)
    # u = W - 0.5 exactly: k=0, f=0.5 -> u = i + 0.5 + cx*... choose tx so that i=7 -> 7.5 = W-0.5 -> rounds to 8 = W (out)
    add(0.5, 0.5, 0.0, 0.0, 0, 0, 0, 3.0)
    depth = np.stack([m[0] for m in maps])
    # sprinkle sentinels and distinct depths so pixel selection is observable
    for m in range(depth.shape[0] - 2):
        depth[m, 0, :] = -1.0                       # bottom vtk row = image row H-1
        depth[m, :, 3] = depth[m, :, 3] + 0.25 * (1 + np.arange(H))
    views = scene.Views(depth, np.stack([m[1] for m in maps]), np.stack([m[2] for m in maps]))
    return grid, rp, views


def main():
    # 1. generic sphere scene
    g = scene.default_grid(32)
    rp = scene.default_ray_potential(g)
    v = scene.make_views(6, 64, 48, seed=1)
    census = branch_census(g, rp, v)
    need = {0.0, round(-rp.eta * rp.rho, 12), rp.rho, -rp.rho}
    assert need <= census, f"generic fixture misses a ray-potential branch: {need - census}"
    _case("generic_sphere_32", g, rp, v, note="sphere scene, identity grid matrix; all ray-potential branches")

    # 2. anisotropic spacing + rotated orthonormal grid matrix
    g = scene.default_grid((40, 24, 20), rotated=True)
    rp = scene.default_ray_potential(g)
    v = scene.make_views(5, 80, 60, seed=2, layout="ring")
    _case("anisotropic_rotated", g, rp, v, note="anisotropic spacing, rotated grid matrix")

    # 3. cameras inside the grid: exercises the h.z < 0 exit (cu:177)
    g = scene.default_grid(24)
    rp = scene.default_ray_potential(g)
    v = scene.make_views(6, 48, 36, seed=3, radius=0.8)
    _case("cameras_inside_grid", g, rp, v, note="cameras at radius 0.8 inside the cube; voxels behind cameras")

    # 4. large -1 regions + best-cost threshold (RD.cxx:138-167)
    g = scene.default_grid(28)
    rp = scene.default_ray_potential(g)
    v = scene.make_views(6, 64, 48, seed=4, dense=True, with_best_cost=True)
    v.depth[:, :, :20] = -1.0
    _case("threshold_and_sentinels", g, rp, v, threshold=0.9, note="dense background, 10 % pixels killed by best cost")

    # 5/6/7. engineered pixel-centre, diff == 0, z == 0, z < 0 and all-sentinel cases
    g, rp, v = engineered_edges()
    _case("engineered_edges", g, rp, v, note="exact half-integer projections, diff==0, h.z==0, h.z<0, sentinel map")

    # 8. non-cubic grid, nx not a multiple of 64, dense scene
    g = scene.default_grid((70, 33, 17))
    rp = scene.default_ray_potential(g)
    v = scene.make_views(4, 96, 64, seed=8, dense=True)
    _case("noncubic_70x33x17", g, rp, v, note="ragged dims; dense background (hit rate ~ in-frustum rate)")

    # 9. accumulate onto a non-zero initial grid (cu:323-327)
    g = scene.default_grid(20)
    rp = scene.default_ray_potential(g)
    v = scene.make_views(5, 40, 30, seed=9)
    init = np.random.default_rng(9).standard_normal(g.n_voxels)
    _case("accumulate_onto_initial", g, rp, v, init_grid=init, note="non-zero initial grid")

    # 10. general K (skew, K[2][2] != 1, non-zero 4th column) and depths that are NOT float32-representable
    g = scene.default_grid(24)
    rp = scene.default_ray_potential(g)
    v = scene.make_views(5, 56, 40, seed=10, dense=True)
    rng = np.random.default_rng(10)
    v.depth = np.where(v.depth == -1, -1.0, v.depth * (1 + 1e-9 * rng.standard_normal(v.depth.shape)))
    v.K4[:, 0, 1] = 0.7                      # skew
    v.K4[1:, 2, 2] = 1.25                    # non-unit homogeneous scale
    v.K4[2:, 0, 3] = 3.0                     # 4th column used by cu:90
    v.K4[3:, 2, 0] = 0.01                    # h.z depends on x
    _case("general_k_f64_depth", g, rp, v, note="general 4x4 K rows, f64-only depths")

    # 11. single voxel row / degenerate dims
    g = scene.GridDesc((130, 1, 1), (-1.0, -0.05, -0.05), (2.0 / 130, 0.1, 0.1), np.eye(4))
    rp = scene.default_ray_potential(g)
    v = scene.make_views(3, 32, 24, seed=11, dense=True)
    _case("single_row_130x1x1", g, rp, v, note="one x-row of voxels")


if __name__ == "__main__":
    main()
