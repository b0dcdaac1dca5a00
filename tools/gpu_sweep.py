#!/usr/bin/env python3
"""Kernel-variant sweep on one GPU: interleaved rounds in ONE process (methodology rule 24),
median and min hipEvent kernel time per variant.  Writes gpurun_out/sweep_<tag>.json.

    python tools/gpu_sweep.py --workload cfg2 --variants 0,1,2,3,4,8 --rounds 5
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from bench import parse_workload  # noqa: E402
from cudadepthmapintegration_amd import capi, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--variants", default="0,1,2,3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--scenes", default="dense,sparse")
    ap.add_argument("--grid-dtype", default="f32")
    ap.add_argument("--tag", default="r01")
    args = ap.parse_args()
    cells, n_maps, W, H = parse_workload(args.workload)
    grid = scene.default_grid(cells)
    ray = scene.default_ray_potential(grid)
    variants = [int(v) for v in args.variants.split(",")]
    results = []
    for sc in args.scenes.split(","):
        t0 = time.time()
        views = scene.make_views(n_maps, W, H, seed=1000, dense=(sc == "dense"), dtype=np.float32)
        gen_s = time.time() - t0
        ctxs = {}
        for v in variants:
            c = capi.FusionContext(grid, ray, grid_dtype=args.grid_dtype, kernel_variant=v)
            c.add_views(views)
            ctxs[v] = c
        times = {v: [] for v in variants}
        main_times = {v: [] for v in variants}
        for r in range(args.rounds + 1):
            for v in variants:
                c = ctxs[v]
                c.reset_grid()
                c.fuse()
                c.synchronize()
                if r > 0:
                    times[v].append(c.timings().last_fuse_kernel_ms)
                    main_times[v].append(c.timings().last_fuse_main_kernel_ms)
        proj = grid.n_voxels * n_maps
        for v in variants:
            t = np.array(times[v])
            rec = {"workload": args.workload, "scene": sc, "variant": v, "k_mode": int(ctxs[v].info().k_mode),
                   "median_ms": float(np.median(t)), "min_ms": float(t.min()),
                   "main_median_ms": float(np.median(main_times[v])),
                   "gproj_per_s_median": proj / np.median(t) / 1e6, "gen_s": gen_s,
                   "tiled": int(ctxs[v].info().tiled_kernel), "brick_classes": ctxs[v].brick_class_histogram(), "mixed_reasons": ctxs[v].mixed_reason_histogram()}
            results.append(rec)
            print(json.dumps(rec), flush=True)
        for c in ctxs.values():
            c.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"sweep_{args.tag}_{args.workload}.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
