#!/usr/bin/env python3
"""The MeshColoration pass at BASELINE config 5's scale on one GPU: all 512 views of 1920x1080 resident (4.2 GB as
RGBA), one rank's share of the mesh (1 M of 8 M vertices).  Checks a sample of the result against the oracle."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cudadepthmapintegration_amd import capi, scene  # noqa: E402
from oracle import oracle  # noqa: E402

n_views, W, H, n_vert = 512, 1920, 1080, 1_000_000
views = scene.make_views(n_views, 8, 8, seed=77)
K4 = views.K4.copy()
K4[:, 0, 0] = K4[:, 1, 1] = 0.9 * W
K4[:, 0, 2], K4[:, 1, 2] = W / 2.0, H / 2.0
pts = scene.make_mesh_points(n_vert, seed=78)
rng = np.random.default_rng(5)
base = rng.integers(0, 256, size=(8, H, W, 3), dtype=np.uint8)   # 8 distinct planes, reused: content does not matter
with capi.ColorContext() as c:
    t0 = time.perf_counter()
    for v0 in range(0, n_views, 8):
        c.add_views(base, K4[v0:v0 + 8], views.RT4[v0:v0 + 8])
    t_up = time.perf_counter() - t0
    c.process(pts[:1000])
    t0 = time.perf_counter()
    mean, median, count = c.process(pts)
    dt = time.perf_counter() - t0
    kms = c.kernel_ms()
    ordered = pts[scene.morton_order(pts)]          # the same vertices in mesh order (neighbours in neighbouring lanes)
    c.process(ordered)
    kms_ordered = c.kernel_ms()
    c.set_vertex_reorder(True)                      # Z-order processing inside the library, for the random vertices
    m2, md2, c2 = c.process(pts)
    kms_reordered = c.kernel_ms()
    same = bool(np.array_equal(m2, mean) and np.array_equal(md2, median) and np.array_equal(c2, count))
colors = np.concatenate([base] * (n_views // 8))
sample = rng.choice(n_vert, size=300, replace=False)
want = oracle.color_mesh(pts[sample], colors, K4, views.RT4)
ok = all(np.array_equal(g[sample], w) for g, w in zip((mean, median, count), want))
rec = {"views": n_views, "image": f"{W}x{H}", "vertices": n_vert, "upload_s": t_up, "process_s": dt, "kernel_ms": kms,
       "gvertex_projections_per_s_kernels": n_vert * n_views / kms / 1e6,
       "kernel_ms_mesh_ordered_vertices": kms_ordered, "kernel_ms_random_vertices_reordered_on_device": kms_reordered,
       "reordered_result_identical": same, "mean_views_per_vertex": float(count.mean()),
       "sample_of_300_matches_oracle": bool(ok)}
print(json.dumps(rec))
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "coloration_cfg5.json"), "w"), indent=1)
sys.exit(0 if ok else 1)
