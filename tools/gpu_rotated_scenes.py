#!/usr/bin/env python3
"""cfg 3 on the dense and the speckle scene with (a) the axis-aligned grid, (b) a rotated grid, (c) the axis-aligned grid and
views whose K has a general third row (the kernel instantiations without tier 1 and without validity maps): ms per fusion."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bench import upload_scene  # noqa: E402
from cudadepthmapintegration_amd import capi, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
res = []
for kind in ("dense", "speckle"):
    for rotated in (False, True):
        grid = scene.default_grid(512, rotated=rotated)
        ray = scene.default_ray_potential(grid)
        with capi.FusionContext(grid, ray, grid_dtype="f32") as c:
            upload_scene(c, scene, kind, n, 1280, 720, float(max(grid.spacing)))
            ts = []
            for r in range(4):
                c.reset_grid()
                c.fuse()
                c.synchronize()
                ts.append(c.timings().last_fuse_kernel_ms)
            rec = {"scene": kind, "rotated": rotated, "views": n, "tiled": int(c.info().tiled_kernel), "ms": float(np.median(ts[1:])),
                   "hist": c.brick_class_histogram()}
            res.append(rec)
            print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "rotated_scenes.json"), "w"), indent=1)
