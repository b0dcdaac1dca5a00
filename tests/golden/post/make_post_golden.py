#!/usr/bin/env python3
"""Mint the fixtures of the two passes next to the fusion path (tests/golden/post/*.npz): the MeshColoration pass and
the cell -> point averaging.  Same status as the fusion fixtures (PARITY UNPINNED: produced by this project's C
restatement, required to agree with the independent numpy restatement before they are written); they freeze the
oracle so that the C oracle, the numpy oracle and the HIP path are held to the same committed numbers.

    python tests/golden/post/make_post_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)

from cudadepthmapintegration_amd import scene  # noqa: E402
from oracle import oracle, oracle_np  # noqa: E402


def main():
    # MeshColoration: 7 views of 64 x 48, 400 vertices incl. degenerate ones (behind the cameras, on a camera plane, NaN)
    views = scene.make_views(7, 64, 48, seed=41)
    colors = scene.make_colors(7, 64, 48, seed=42)
    pts = scene.make_mesh_points(400, seed=43)
    pts[:4] = [[0, 0, 0], [1e9, -1e9, 1e9], [np.nan, 0, 0], [0.0, 0.0, 3.0]]
    a = oracle.color_mesh(pts, colors, views.K4, views.RT4)
    b = oracle_np.color_mesh_np(pts, colors, views.K4, views.RT4)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)), "coloration: C and numpy oracles disagree"
    assert a[2].max() >= 4 and (a[2] % 2 == 0).any() and (a[2] % 2 == 1).any()   # even and odd counts: both median rules
    np.savez_compressed(os.path.join(HERE, "coloration.npz"), points=pts, colors=colors, K4=views.K4, RT4=views.RT4,
                        expected_mean=a[0], expected_median=a[1], expected_count=a[2])
    # cell -> point: a 9 x 6 x 5 cell grid with mixed magnitudes, zeros and a -0.0
    rng = np.random.default_rng(44)
    cells = rng.normal(size=(5, 6, 9)) * 10.0 ** rng.integers(-3, 4, size=(5, 6, 9))
    cells[rng.random(cells.shape) < 0.2] = 0.0
    cells[0, 0, 0] = -0.0
    p = oracle.cell_to_point(cells)
    q = oracle_np.cell_to_point_np(cells)
    assert p.tobytes() == q.tobytes(), "cell_to_point: C and numpy oracles disagree"
    np.savez_compressed(os.path.join(HERE, "cell_to_point.npz"), cells=cells, expected_points=p)
    # iso-value pre-pass (Reconstruction/main.cxx:169-173): the cells of that point lattice whose corners straddle the
    # iso-value; values equal to the iso-value (>= counts as inside), a NaN (outside) and both infinities among the corners
    pts = p.copy()
    iso = 0.25
    pts[1, 2, 3] = iso
    pts[2, 2, 2] = np.nan
    pts[3, 1, 4] = np.inf
    pts[0, 5, 8] = -np.inf
    ids = oracle.iso_active_cells(pts, iso)
    assert np.array_equal(ids, oracle_np.iso_active_cells_np(pts, iso)), "iso pre-pass: C and numpy oracles disagree"
    assert 0 < ids.size < cells.size
    np.savez_compressed(os.path.join(HERE, "iso_cells.npz"), points=pts, iso=iso, expected_ids=ids)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
