#!/usr/bin/env python3
"""The upload pass's kernel time call by call (dmi_get_upload_kernel_ms after every dmi_add_views of 32 views, cfg3 speckle: f64
tables + best-cost values), twice: does the first call of a process carry one-time costs?"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cudadepthmapintegration_amd import capi, scene
from bench import SCENE_SEED

grid = scene.default_grid(512); ray = scene.default_ray_potential(grid)
spacing = float(max(grid.spacing))
chunks = [scene.make_scene_views("speckle", 256, 1280, 720, seed=SCENE_SEED, view_range=(c0, c0 + 32), noise_sigma=spacing) for c0 in range(0, 256, 32)]
res = {}
with capi.FusionContext(grid, ray, grid_dtype="f32", depth_storage="auto") as c:
    for rep in range(2):
        c.clear_views()
        per = []
        for v, thr in chunks:
            c.add_views(v, threshold=thr)
            per.append(round(c.upload_kernel_ms()[0], 3))
        res[f"f64_cost_rep{rep}"] = per
        print("f64+cost rep", rep, per, "sum", round(sum(per), 3), flush=True)
    for rep in range(2):
        c.clear_views()
        per = []
        for v, thr in chunks:
            d = v.depth.astype(np.float32)
            c.add_views(scene.Views(d, v.K4, v.RT4))
            per.append(round(c.upload_kernel_ms()[0], 3))
        res[f"f32_rep{rep}"] = per
        print("f32 rep", rep, per, "sum", round(sum(per), 3), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "upload_calls.json"), "w"), indent=1)
