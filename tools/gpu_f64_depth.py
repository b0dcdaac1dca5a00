#!/usr/bin/env python3
"""cfg3 with depth tables that are NOT exactly f32 (what a real stereo pipeline hands over as vtkDoubleArray): the
context keeps them as f64 (DMI_DEPTH_AUTO promotes) and the fusion gathers 8-byte values."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bench import upload_scene  # noqa: E402
from cudadepthmapintegration_amd import capi, scene  # noqa: E402

grid = scene.default_grid(512)
ray = scene.default_ray_potential(grid)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
res = []
for sc in ("dense", "speckle"):
    for storage in ("f32", "f64"):
        with capi.FusionContext(grid, ray, grid_dtype="f32", depth_storage=storage) as c:
            upload_scene(c, scene, sc, n, 1280, 720, float(max(grid.spacing)))
            ts = []
            for r in range(4):
                c.reset_grid()
                c.fuse()
                c.synchronize()
                ts.append(c.timings().last_fuse_kernel_ms)
            rec = {"scene": sc, "views": n, "depth_storage": storage, "ms": float(np.median(ts[1:])),
                   "gproj_per_s": grid.n_voxels * n / np.median(ts[1:]) / 1e6, "hist": c.brick_class_histogram()}
            res.append(rec)
            print(json.dumps(rec), flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "f64_depth.json"), "w"), indent=1)
