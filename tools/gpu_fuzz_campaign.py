#!/usr/bin/env python3
"""Randomised parity campaign on one GPU: the scene generator of tests/test_gpu_fuzz.py over many seeds, every kernel
path (default tiled kernel with brick classes, with and without hit counters; tiled without classes; another tile shape;
general kernel) against the CPU oracle, bit for bit (fp64 grid, hit counters).  Also the cell -> point pass of each
result.  Writes gpurun_out/fuzz_campaign.json; stops at the first mismatch and records it.

    python tools/gpu_fuzz_campaign.py --first 1000 --count 1500
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from cudadepthmapintegration_amd import capi  # noqa: E402
from helpers import bits_equal, oracle_params_from_scene  # noqa: E402
from oracle import oracle  # noqa: E402
from test_gpu_fuzz import _random_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--count", type=int, default=1000)
    ap.add_argument("--seconds", type=float, default=420.0, help="stop starting new cases after this long")
    ap.add_argument("--medium-every", type=int, default=0,
                    help="every n-th seed is a medium-size scene (more bricks than persistent workgroups); 0 = none")
    args = ap.parse_args()
    fx = capi.VARIANT_FIXED_TILE_SHAPE
    SHIPPED = -1  # kernel_variant 0 as a caller passes it
    variants = [(0, True), (0, False), (fx, True), (fx, False), (fx | capi.VARIANT_NO_BRICK_CLASSES, True), (96, True),
                (capi.VARIANT_FORCE_GENERAL, True), (capi.VARIANT_NO_INTERIOR, False), (capi.VARIANT_XCD_RUNS, False),
                (capi.VARIANT_ZMAJOR_SLOTS, False), (capi.VARIANT_WINDOWS_ALWAYS, False), (fx | capi.VARIANT_WINDOWS_ALWAYS, False),
                (capi.VARIANT_COST_ORDER | capi.VARIANT_WINDOWS_ALWAYS, False), (SHIPPED, False)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    t0 = time.time()
    done = 0
    voxel_projections = 0
    paths = {"tiled": 0, "general": 0, "f64_depth": 0}
    failure = None
    for seed in range(args.first, args.first + args.count):
        if time.time() - t0 > args.seconds:
            break
        medium = args.medium_every > 0 and seed % args.medium_every == 0
        # every other medium scene has 96+ views: the persistent form of the kernel (fewer views: one workgroup per brick)
        grid, rp, views = _random_case(seed, medium=medium, many_views=medium and (seed // max(1, args.medium_every)) % 2 == 0)
        init = None
        if seed % 5 == 0:
            init = np.random.default_rng(seed).normal(size=(grid.cell_dims[2], grid.cell_dims[1], grid.cell_dims[0]))
            if seed % 10 == 0:
                init[np.random.default_rng(seed + 1).random(init.shape) < 0.3] = -0.0
        with np.errstate(all="ignore"):
            want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4,
                                           init_grid=init, n_threads=oracle.max_threads())
            want_pts = oracle.cell_to_point(want)
        for variant, count in variants:
            # (tiny grids fuse without classes by default: the campaign is about the classes, so they stay on; the last path
            # of the list is the shipped default)
            kv = variant if variant == SHIPPED or (variant & capi.VARIANT_NO_BRICK_CLASSES) else variant | capi.VARIANT_BRICK_CLASSES_ALWAYS
            with capi.FusionContext(grid, rp, count_hits=count, kernel_variant=0 if variant == SHIPPED else kv) as ctx:
                if init is not None:
                    ctx.upload_grid(init)
                ctx.add_views(views)
                ctx.fuse()
                out = ctx.download_grid()
                ok = bits_equal(out, want)
                if ok and count:
                    vh, mh = ctx.download_hits()
                    ok = np.array_equal(vh, vh_w) and np.array_equal(mh, mh_w)
                if ok and variant == 0:
                    ok = bits_equal(ctx.download_point_data(), want_pts)
                info = ctx.info()
            if variant == 0 and count:
                paths["tiled" if info.tiled_kernel else "general"] += 1
                paths["f64_depth"] += 1 if info.depth_storage_in_use == capi.DMI_DEPTH_F64 else 0
            if not ok:
                failure = {"seed": seed, "variant": variant, "count_hits": count}
                break
        if failure:
            break
        done += 1
        voxel_projections += grid.n_voxels * views.n
        if done % 50 == 0:
            print(f"{done} cases bit-exact, {time.time() - t0:.0f} s", flush=True)
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_campaign_progress.txt"), "a") as f:  # survives a cut-off call
                f.write(f"{done} cases bit-exact from seed {args.first}, {time.time() - t0:.0f} s\n")
    res = {"first_seed": args.first, "cases": done, "kernel_paths_per_case": len(variants), "voxel_projections_per_path": voxel_projections,
           "default_path_ran": paths, "failure": failure, "seconds": time.time() - t0}
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fuzz_campaign.json"), "w"), indent=1)
    return 1 if failure else 0


if __name__ == "__main__":
    sys.exit(main())
