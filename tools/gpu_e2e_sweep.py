#!/usr/bin/env python3
"""bench.py's PCIe-inclusive probe (cfg3, speckle, f32 in / f32 out and f64 / f64) for several chunk sizes and numbers of download
slabs (dmi_fuse_range_download): which pair the probe and FusionDriver::ProcessDepthMap should use."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from cudadepthmapintegration_amd import capi, scene
from bench import end_to_end_probe, upload_scene

grid = scene.default_grid(512); ray = scene.default_ray_potential(grid)
ctx = capi.FusionContext(grid, ray, grid_dtype="f32", depth_storage="auto")
views = upload_scene(ctx, scene, "speckle", 256, 1280, 720, float(max(grid.spacing)), keep_host=True)
ctx.close()
pcie = capi.pcie_probe(0)
res = []
for hd, gd in (("f32", "f32"), ("f64", "f64")):
    for chunk in (16, 32, 48, 64):
        for slabs in (1, 4, 8, 16):
            r = end_to_end_probe(scene, capi, grid, ray, views, hd, gd, pcie, chunk_views=chunk, download_slabs=slabs)
            res.append(r)
            print(hd, gd, "chunk", chunk, "slabs", slabs, round(r["seconds"] * 1e3, 2), "ms x", round(r["seconds_over_floor"], 3), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"pcie_GBps": pcie, "runs": res}, open(os.path.join(ROOT, "gpurun_out", "e2e_sweep.json"), "w"), indent=1)
