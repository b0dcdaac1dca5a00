#!/usr/bin/env python3
"""A rotated grid (orthonormal axes that are not the coordinate axes) at 512^3: the tiled kernel's rotated path against
the general kernel and against the same scene on an axis-aligned grid."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cudadepthmapintegration_amd import capi, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
res = []
for rotated in (False, True):
    grid = scene.default_grid(512, rotated=rotated)
    ray = scene.default_ray_potential(grid)
    views = scene.make_views(n, 1280, 720, seed=1000, dense=True, dtype=np.float32)
    for variant in (0, capi.VARIANT_FIXED_TILE_SHAPE, capi.VARIANT_FORCE_GENERAL):
        with capi.FusionContext(grid, ray, grid_dtype="f32", kernel_variant=variant) as c:
            c.add_views(views)
            ts = []
            for r in range(3):
                c.reset_grid()
                c.fuse()
                c.synchronize()
                ts.append(c.timings().last_fuse_kernel_ms)
            rec = {"rotated": rotated, "views": n, "variant": variant, "tiled": int(c.info().tiled_kernel), "ms": float(np.median(ts[1:])),
                   "gproj_per_s": grid.n_voxels * n / np.median(ts[1:]) / 1e6, "hist": c.brick_class_histogram()}
            res.append(rec)
            print(json.dumps(rec), flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "rotated_grid.json"), "w"), indent=1)
