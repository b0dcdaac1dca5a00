#!/usr/bin/env python3
"""Where the PCIe-inclusive time of one fusion goes (cfg3, pinned host memory): uploads alone, uploads + pipelined
fuses, download alone, for several chunk sizes."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cudadepthmapintegration_amd import capi, scene  # noqa: E402

grid = scene.default_grid(512)
ray = scene.default_ray_potential(grid)
views = scene.make_views(256, 1280, 720, seed=1000, dense=True, dtype=np.float32)
pinned = capi.pinned_empty(views.depth.shape, np.float32)
pinned[:] = views.depth
out = capi.pinned_empty((grid.n_voxels,), np.float32)
res = []
with capi.FusionContext(grid, ray, grid_dtype="f32") as c:
    for chunk in (8, 16, 32, 64, 256):
        for fuse in (False, True):
            ts = []
            for rep in range(3):
                c.clear_views()
                c.reset_grid()
                c.synchronize()
                t0 = time.perf_counter()
                for v0 in range(0, 256, chunk):
                    c.add_views(scene.Views(pinned[v0:v0 + chunk], views.K4[v0:v0 + chunk], views.RT4[v0:v0 + chunk]))
                    if fuse:
                        c.fuse(v0, chunk)
                t1 = time.perf_counter()
                c.synchronize()
                t2 = time.perf_counter()
                c.download_grid(np.float32, out=out)
                t3 = time.perf_counter()
                ts.append((t1 - t0, t2 - t1, t3 - t2))
            a = np.median(np.array(ts[1:]), axis=0) * 1e3
            rec = {"chunk_views": chunk, "fuse": fuse, "upload_loop_ms": float(a[0]), "drain_ms": float(a[1]), "download_ms": float(a[2]),
                   "h2d_GBps": pinned.nbytes / a[0] / 1e6, "d2h_GBps": out.nbytes / a[2] / 1e6}
            res.append(rec)
            print(json.dumps(rec), flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "e2e_probe.json"), "w"), indent=1)
