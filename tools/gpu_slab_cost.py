#!/usr/bin/env python3
"""What fusing in z-slabs (the multi-GPU overlap of bench.py --slabs) costs on one GPU: hipEvent time of the
launches of one fusion as a whole and as 2 / 4 / 8 slabs (cfg3)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cudadepthmapintegration_amd import capi, scene, sharding  # noqa: E402

grid = scene.default_grid(512)
ray = scene.default_ray_potential(grid)
views = scene.make_views(256, 1280, 720, seed=1000, dense=True, dtype=np.float32)
with capi.FusionContext(grid, ray, grid_dtype="f32") as ctx:
    ctx.add_views(views)
    for n in (1, 2, 4, 8, 1):
        ts = []
        for r in range(4):
            ctx.reset_grid()
            t0 = ctx.timings().total_fuse_kernel_ms
            for z0, zc in sharding.slab_ranges(512, n):
                ctx.fuse_slab(z0, zc)
            ctx.synchronize()
            ts.append(ctx.timings().total_fuse_kernel_ms - t0)
        print(json.dumps({"slabs": n, "kernel_ms_sum": float(np.median(ts[1:]))}), flush=True)
