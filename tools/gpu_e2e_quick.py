#!/usr/bin/env python3
"""Where the PCIe-inclusive time of one fusion goes on the bench's scene (cfg3, speckle): bench.py's end_to_end figures, then the
f32 / f32 case taken apart (upload loop, drain of the queued fusions, download) for several chunk sizes."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from cudadepthmapintegration_amd import capi, scene
from bench import end_to_end_probe, upload_scene

grid = scene.default_grid(512); ray = scene.default_ray_potential(grid)
ctx = capi.FusionContext(grid, ray, grid_dtype="f32", depth_storage="auto")
views = upload_scene(ctx, scene, "speckle", 256, 1280, 720, float(max(grid.spacing)), keep_host=True)
ctx.close()
pcie = capi.pcie_probe(0)
res = {"pcie_GBps": pcie, "end_to_end": [], "breakdown": []}
for hd, gd in (("f32", "f32"), ("f64", "f64")):
    r = end_to_end_probe(scene, capi, grid, ray, views, hd, gd, pcie)
    res["end_to_end"].append(r)
    print(hd, gd, round(r["seconds"] * 1e3, 1), "ms floor", round(r["pcie_floor_s"] * 1e3, 1), "x", round(r["seconds_over_floor"], 2), flush=True)
pinned = capi.pinned_empty(views.depth.shape, np.float32)
pinned[:] = views.depth
out = capi.pinned_empty((grid.n_voxels,), np.float32)
with capi.FusionContext(grid, ray, grid_dtype="f32") as c:
    for chunk in (16, 32, 64, 128, 256):
        for fuse in (False, True):
            ts = []
            for rep in range(3):
                c.clear_views(); c.reset_grid(); c.synchronize()
                k0 = c.timings()
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
                k1 = c.timings()
                ts.append((t1 - t0, t2 - t1, t3 - t2, (k1.total_fuse_kernel_ms - k0.total_fuse_kernel_ms) * 1e-3, (k1.total_fuse_main_kernel_ms - k0.total_fuse_main_kernel_ms) * 1e-3))
            a = np.median(np.array(ts[1:]), axis=0) * 1e3
            rec = {"chunk_views": chunk, "fuse": fuse, "upload_loop_ms": float(a[0]), "drain_ms": float(a[1]), "download_ms": float(a[2]),
                   "fuse_kernels_ms": float(a[3]), "main_kernels_ms": float(a[4]), "total_ms": float(a[0] + a[1] + a[2])}
            res["breakdown"].append(rec)
            print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "e2e_quick.json"), "w"), indent=1)
