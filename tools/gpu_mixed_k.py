#!/usr/bin/env python3
"""What a view that is not a plain pinhole costs: cfg3 dense with (a) 256 pinhole views, (b) one of
them with a scaled third row of K (K[2][2] = 1.25: the whole launch takes the GENK instantiation), (c) one of them with
a K no fast path takes (focal 1e14: that view alone goes through the general kernel, three launches), (d) every view
general.  Rounds interleaved in one process; writes gpurun_out/mixed_k_<tag>.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cudadepthmapintegration_amd import capi, scene

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
grid = scene.default_grid(512); ray = scene.default_ray_potential(grid)
base = scene.make_views(256, 1280, 720, seed=1000, dense=True, layout="sphere", dtype=np.float32)
def variant(kind):
    K = base.K4.copy()
    if kind == "one_general":
        K[100, 2, 2] = 1.25
    elif kind == "one_untileable":
        K[100, 0, 0] = 1e14
    elif kind == "all_general":
        K[:, 2, 2] = 1.25
    return scene.Views(base.depth, K, base.RT4)
ctxs = {}
for kind in ("all_pinhole", "one_general", "one_untileable", "all_general"):
    c = capi.FusionContext(grid, ray, grid_dtype="f32"); c.add_views(variant(kind)); ctxs[kind] = c
times = {k: [] for k in ctxs}
for r in range(8):
    for k, c in ctxs.items():
        t0 = c.timings().total_fuse_kernel_ms
        for _ in range(4):
            c.reset_grid(); c.fuse()
        c.synchronize()
        if r: times[k].append((c.timings().total_fuse_kernel_ms - t0) / 4)
out = {k: {"fuse_ms": float(np.median(v)), "tiled_kernel": int(ctxs[k].info().tiled_kernel), "k_mode": int(ctxs[k].info().k_mode)} for k, v in times.items()}
for k in out: out[k]["vs_all_pinhole"] = out[k]["fuse_ms"] / out["all_pinhole"]["fuse_ms"]
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"mixed_k_{tag}.json"), "w"), indent=1)
