#!/usr/bin/env python3
"""The coloration pass of bench.py's probe for ONE vertex order, so that a rocprofv3 --pmc pass over it is unambiguous
(tools/gpu_coloration_pmc.sh): 2 M vertices x 64 views of 1280 x 720, views resident.  Prints one JSON line with the kernel time."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cudadepthmapintegration_amd import capi, scene  # noqa: E402

order = sys.argv[1] if len(sys.argv) > 1 else "random"
n_vertices, n_views, W, H = 2_000_000, 64, 1280, 720
views = scene.make_views(n_views, 8, 8, seed=77)
colors = np.empty((n_views, H, W, 3), dtype=np.uint8)
colors[:] = (np.arange(H * W * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(1, H, W, 3)
K4 = views.K4.copy()
K4[:, 0, 0] = K4[:, 1, 1] = 0.9 * W
K4[:, 0, 2], K4[:, 1, 2] = W / 2.0, H / 2.0
pts = scene.make_mesh_points(n_vertices, seed=78)
if order == "mesh":
    pts = pts[scene.morton_order(pts)]
with capi.ColorContext() as c:
    c.add_views(colors, K4, views.RT4)
    c.set_vertex_reorder(order == "reordered")   # random input, Z-order processing inside the library (dmi_color_set_vertex_reorder)
    mean, median, count = c.process(pts)
    print(json.dumps({"order": order, "vertices": n_vertices, "views": n_views, "kernel_ms": c.kernel_ms(),
                      "pairs_in_image": int(count.sum(dtype=np.int64))}))
