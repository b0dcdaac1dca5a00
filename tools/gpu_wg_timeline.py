#!/usr/bin/env python3
"""When do the workgroups of the fusion kernel run, and where?  A tuning build (DMI_TUNING=1 python -m
cudadepthmapintegration_amd.build) records s_memrealtime at the start and end of every workgroup and the XCC it ran on
(TileArgs::wg_times).  One record per brick (the workgroups are persistent and take bricks from their XCD's counter).  Prints and writes: per XCD the time its last brick ends, the share of the kernel's duration in
which fewer than 50 / 90 % of the peak number of workgroups are resident (the tail), and the work (sum of workgroup
durations) per XCD.

    DMI_DEBUG_WG_TIMES=1 DMI_LIB_OVERRIDE=cudadepthmapintegration_amd/csrc/libdmi_hip_tuning.so python tools/gpu_wg_timeline.py
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bench import parse_workload  # noqa: E402
from cudadepthmapintegration_amd import capi, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scene", default="dense")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--tag", default="wg_timeline")
    args = ap.parse_args()
    assert os.environ.get("DMI_DEBUG_WG_TIMES"), "set DMI_DEBUG_WG_TIMES=1 (and load the tuning build)"
    lib = capi.load()
    lib.dmi_debug_wg_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    cells, n_maps, W, H = parse_workload(args.workload)
    grid = scene.default_grid(cells)
    ray = scene.default_ray_potential(grid)
    views = scene.make_views(n_maps, W, H, seed=1000, dense=(args.scene == "dense"), dtype=np.float32)
    with capi.FusionContext(grid, ray, grid_dtype="f32", kernel_variant=args.variant) as ctx:
        ctx.add_views(views)
        for _ in range(3):
            ctx.reset_grid()
            ctx.fuse()
        ctx.synchronize()
        n = ctypes.c_int64()
        lib.dmi_debug_wg_times(ctx._h, None, 0, ctypes.byref(n))
        blocks = int(n.value)
        buf = np.zeros(3 * blocks, dtype=np.uint64)
        rc = lib.dmi_debug_wg_times(ctx._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), buf.size, ctypes.byref(n))
        assert rc == 0
        kernel_ms = ctx.timings().last_fuse_main_kernel_ms
    t = buf[:2 * blocks].reshape(blocks, 2).astype(np.int64)
    xcc = (buf[2 * blocks:] & np.uint64(15)).astype(np.int64)
    block = (buf[2 * blocks:] >> np.uint64(8)).astype(np.int64)   # the (persistent) workgroup that fused the brick
    ran = t[:, 1] > 0
    t0 = t[ran, 0].min()
    start = (t[ran, 0] - t0) * 1e-5   # ms (100 MHz)
    end = (t[ran, 1] - t0) * 1e-5
    xcc = xcc[ran]
    dur = end - start
    total = end.max()
    # resident workgroups over time
    ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    level = np.cumsum(ev[:, 1])
    peak = level.max()
    dt = np.diff(np.append(ev[:, 0], total))
    below90 = float(dt[level < 0.9 * peak].sum())
    below50 = float(dt[level < 0.5 * peak].sum())
    # resident workgroups (share of the peak) sampled every 0.25 ms, whole chip and per XCD
    grid_t = np.arange(0.0, total, 0.25)
    def resident(sel):
        return [int(((start[sel] <= x) & (end[sel] > x)).sum()) for x in grid_t]
    curve = {"t_ms": [round(float(x), 2) for x in grid_t], "all": resident(np.ones_like(start, dtype=bool))}
    for x in sorted(set(xcc.tolist())):
        curve[f"xcc{x}"] = resident(xcc == x)
    # mean duration of the workgroups that START in each quarter millisecond (what kind of brick is being started when)
    curve["mean_wg_ms_started"] = [float(dur[(start >= x) & (start < x + 0.25)].mean()) if ((start >= x) & (start < x + 0.25)).any() else 0.0 for x in grid_t]
    per_xcd = []
    for x in sorted(set(xcc.tolist())):
        sel = xcc == x
        per_xcd.append({"xcc": int(x), "workgroups": int(sel.sum()), "last_end_ms": float(end[sel].max()),
                        "work_ms": float(dur[sel].sum()), "blockidx_mod8": sorted(set((block[ran][sel] % 8).tolist()))})
    rec = {"workload": args.workload, "scene": args.scene, "variant": args.variant, "kernel_ms_by_events": kernel_ms,
           "span_ms_by_memrealtime": float(total), "bricks_fused": int(ran.sum()), "workgroups_that_fused_a_brick": int(len(set(block[ran].tolist()))), "peak_resident_workgroups": int(peak),
           "ms_below_90pct_of_peak": below90, "ms_below_50pct_of_peak": below50,
           "workgroup_ms": {"median": float(np.median(dur)), "p99": float(np.percentile(dur, 99)), "max": float(dur.max())},
           "per_xcd": per_xcd, "curve": curve,
           "first_xcd_done_ms": min(p["last_end_ms"] for p in per_xcd), "last_xcd_done_ms": max(p["last_end_ms"] for p in per_xcd)}
    print(json.dumps({k: v for k, v in rec.items() if k != "curve"}, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", f"{args.tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
