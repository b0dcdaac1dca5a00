#!/usr/bin/env python3
"""Per-step fusion time from a cold start: does the first second of a fresh process run slower (clock ramp, first-touch
of device memory)?  Prints hipEvent ms of each of N consecutive cfg3 steps, synchronised one by one, then back to back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cudadepthmapintegration_amd import capi, scene
grid = scene.default_grid(512); ray = scene.default_ray_potential(grid)
views = scene.make_views(256, 1280, 720, seed=1000, dense=True, layout="sphere", dtype=np.float32)
c = capi.FusionContext(grid, ray, grid_dtype="f32"); c.add_views(views)
ms = []
t0 = time.perf_counter()
for i in range(40):
    c.reset_grid(); c.fuse(); c.synchronize()
    ms.append(round(c.timings().last_fuse_kernel_ms, 2))
print("sync each:", ms, "wall", round(time.perf_counter() - t0, 3))
for rep in range(3):
    k0 = c.timings().total_fuse_kernel_ms; t0 = time.perf_counter()
    for i in range(20):
        c.reset_grid(); c.fuse()
    c.synchronize()
    print("20 back to back: kernel ms/step", round((c.timings().total_fuse_kernel_ms - k0) / 20, 3), "wall ms/step", round((time.perf_counter() - t0) * 50, 3))
    time.sleep(1.0)
