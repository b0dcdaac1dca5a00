#!/usr/bin/env python3
"""What each of the eight 32-view fusions of a chunked reconstruction costs (cfg3, speckle): the first starts from a zero grid,
the others accumulate onto the sums so far.  Prints fuse / main-kernel ms per chunk, for f32 and f64 grids."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from cudadepthmapintegration_amd import capi, scene
from bench import upload_scene

grid = scene.default_grid(512); ray = scene.default_ray_potential(grid)
res = {}
for gd in ("f32", "f64"):
    for chunk in (32, 64):
        with capi.FusionContext(grid, ray, grid_dtype=gd) as c:
            upload_scene(c, scene, "speckle", 256, 1280, 720, float(max(grid.spacing)))
            rows = []
            for rep in range(3):
                c.reset_grid(); c.synchronize()
                rows = []
                for v0 in range(0, 256, chunk):
                    c.fuse(v0, chunk); c.synchronize()
                    t = c.timings()
                    rows.append((round(t.last_fuse_kernel_ms, 3), round(t.last_fuse_main_kernel_ms, 3)))
            res[f"{gd}_chunk{chunk}"] = rows
            print(gd, chunk, rows, "sum", round(sum(r[0] for r in rows), 2), round(sum(r[1] for r in rows), 2), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "chunk_fusions.json"), "w"), indent=1)
