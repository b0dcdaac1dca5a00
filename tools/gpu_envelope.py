#!/usr/bin/env python3
"""The magnitude envelope (round 5): at which world offsets and focal lengths does a view leave tier 1 / the tiled kernel, and what
does the fusion cost there?  256^3 voxels x 32 views of 640 x 480, the speckle scene scaled to a 20 m cube (3.9 cm ... 7.8 cm
voxels) and moved by `offset` along every axis; focal length in pixels (the default lens is 0.9 x 640 = 576 px; longer lenses see
less of the scene).  One JSON line per point -> gpurun_out/<tag>_envelope.jsonl (INTEGRATION.md "Magnitudes" quotes it)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from cudadepthmapintegration_amd import capi, scene  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "env"
    n, W, H, cells = 32, 640, 480, (256, 256, 256)
    grid0 = scene.default_grid(cells)
    ray0 = scene.default_ray_potential(grid0)
    base, thr = scene.make_scene_views("speckle", n, W, H, seed=1000)
    out = []
    for focal in (None, 2000.0, 8000.0):
        for off in (0.0, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8):
            grid, ray, views = scene.to_world_frame(grid0, ray0, base, 10.0, (off, -off * 0.83, off * 0.01), focal=focal)
            with capi.FusionContext(grid, ray, grid_dtype="f32") as ctx:
                ctx.add_views(views, threshold=thr)
                paths = ctx.view_paths()
                ms = []
                for _ in range(4):
                    ctx.reset_grid()
                    ctx.fuse()
                    ctx.synchronize()
                    ms.append(ctx.timings().last_fuse_kernel_ms)
                rec = {"offset": off, "focal_px": focal or 0.9 * W, "fuse_ms": round(float(np.median(ms[1:])), 3), "view_paths": paths,
                       "window_pairs": ctx.window_pair_count(), "mixed": ctx.brick_class_histogram()["mixed"]}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_envelope.jsonl"), "w") as fh:
        for r in out:
            fh.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
