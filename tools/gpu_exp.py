#!/usr/bin/env python3
"""A/B timing of several builds of the library in ONE process, rounds interleaved (clock and thermal drift of the box hit
every build alike): each line of the list is NAME:-Dflags (built by tools/exp_build.sh as libdmi_hip_exp_NAME.so) or
NAME:@path/to/prebuilt.so.

    python tools/gpu_exp.py tools/exp_list.txt --workload cfg3 --rounds 9 --variants 0
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from bench import parse_workload, upload_scene  # noqa: E402
from cudadepthmapintegration_amd import build as _build, capi, scene  # noqa: E402


def load_lib(path):
    _build.LIB_PATH = path
    _build.LIB_OVERRIDE = path
    capi._lib = None
    return capi.load()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("list")
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--variants", default="0")
    ap.add_argument("--scenes", default="dense,sparse")
    ap.add_argument("--tag", default="exp")
    ap.add_argument("--steps", type=int, default=8, help="fusions queued back to back per measurement, as bench.py's timed loop does")
    args = ap.parse_args()
    libs = []
    for line in open(args.list):
        line = line.strip()
        if not line:
            continue
        name, _, defs = line.partition(":")
        path = defs[1:] if defs.startswith("@") else os.path.join(_build.CSRC, f"libdmi_hip_exp_{name}.so")
        libs.append((name, os.path.abspath(path)))
    cells, n_maps, W, H = parse_workload(args.workload)
    grid = scene.default_grid(cells)
    ray = scene.default_ray_potential(grid)
    variants = [int(v) for v in args.variants.split(",")]
    out = []
    for sc in args.scenes.split(","):
        ctxs = {}
        for name, path in libs:
            load_lib(path)
            for v in variants:
                c = capi.FusionContext(grid, ray, grid_dtype="f32", kernel_variant=v)
                upload_scene(c, scene, sc, n_maps, W, H, float(max(grid.spacing)))
                ctxs[(name, v)] = c
        fuse = {k: [] for k in ctxs}
        main_ms = {k: [] for k in ctxs}
        for r in range(args.rounds + 1):
            for k, c in ctxs.items():
                t0 = c.timings()
                for _ in range(args.steps):
                    c.reset_grid()
                    c.fuse()
                c.synchronize()
                if r > 0:
                    t = c.timings()
                    fuse[k].append((t.total_fuse_kernel_ms - t0.total_fuse_kernel_ms) / args.steps)
                    main_ms[k].append((t.total_fuse_main_kernel_ms - t0.total_fuse_main_kernel_ms) / args.steps)
        ref = None
        ref_grid = None
        same = {}
        for k, c in ctxs.items():                     # every build must leave the same bits in the grid as the first one
            g = np.ascontiguousarray(c.download_grid(np.float32)).view(np.uint8)
            if ref_grid is None:
                ref_grid = g
            same[k] = bool(np.array_equal(g, ref_grid))
        for k in ctxs:
            rec = {"scene": sc, "lib": k[0], "variant": k[1], "brick_classes": ctxs[k].brick_class_histogram(),
                   "mixed_reasons": ctxs[k].mixed_reason_histogram(), "window_pairs": ctxs[k].window_pair_count(), "fuse_ms": float(np.median(fuse[k])), "main_ms": float(np.median(main_ms[k])),
                   "main_min_ms": float(np.min(main_ms[k])), "grid_bits_equal_first": same[k]}
            if ref is None:
                ref = rec["main_ms"]
                ref_rounds = np.array(main_ms[k])
            rec["main_vs_first"] = rec["main_ms"] / ref
            # round by round against the first build (measured back to back: the box's drift mostly cancels)
            ratios = np.array(main_ms[k]) / ref_rounds
            rec["paired_ratio_median"] = float(np.median(ratios))
            rec["paired_ratio_quartiles"] = [float(np.percentile(ratios, 25)), float(np.percentile(ratios, 75))]
            rec["main_rounds_ms"] = [round(float(x), 3) for x in main_ms[k]]
            # the rounds are a base level plus spikes of a millisecond or two (the box, not the build): the lower quartile
            # is the steadier figure
            rec["main_p25_ms"] = float(np.percentile(main_ms[k], 25))
            out.append(rec)
            print(sc, k[0], k[1], "fuse", round(rec["fuse_ms"], 3), "main", round(rec["main_ms"], 3), "min", round(rec["main_min_ms"], 3),
                  "x%.3f" % rec["main_vs_first"], "p25", round(rec["main_p25_ms"], 3), "paired x%.3f [%.3f, %.3f]" % (rec["paired_ratio_median"], *rec["paired_ratio_quartiles"]),
                  "" if same[k] else "GRID DIFFERS", flush=True)
        for c in ctxs.values():
            c.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"exp_{args.tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
