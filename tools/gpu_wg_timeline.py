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

from bench import parse_workload, upload_scene  # noqa: E402
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
    with capi.FusionContext(grid, ray, grid_dtype="f32", kernel_variant=args.variant) as ctx:
        upload_scene(ctx, scene, args.scene, n_maps, W, H, float(max(grid.spacing)))
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
    block = ((buf[2 * blocks:] >> np.uint64(8)) & np.uint64(0xffffff)).astype(np.int64)   # the (persistent) workgroup that fused the brick
    n_cols = ((buf[2 * blocks:] >> np.uint64(32)) & np.uint64(0x3ff)).astype(np.int64)    # views with a column of their own
    n_win = ((buf[2 * blocks:] >> np.uint64(42)) & np.uint64(0x3ff)).astype(np.int64)     # ... through a bit window
    n_win_early = ((buf[2 * blocks:] >> np.uint64(52)) & np.uint64(0x3ff)).astype(np.int64)  # ... before the brick's first other column
    n_redo = (buf[2 * blocks:] >> np.uint64(62)).astype(np.int64)  # voxels redone after their column (saturates at 3)
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
    step = 0.25 if total > 2.5 else total / 16
    grid_t = np.arange(0.0, total, step)
    def resident(sel):
        return [int(((start[sel] <= x) & (end[sel] > x)).sum()) for x in grid_t]
    curve = {"t_ms": [round(float(x), 2) for x in grid_t], "all": resident(np.ones_like(start, dtype=bool))}
    for x in sorted(set(xcc.tolist())):
        curve[f"xcc{x}"] = resident(xcc == x)
    # mean duration of the workgroups that START in each quarter millisecond (what kind of brick is being started when)
    curve["mean_wg_ms_started"] = [float(dur[(start >= x) & (start < x + step)].mean()) if ((start >= x) & (start < x + step)).any() else 0.0 for x in grid_t]
    per_xcd = []
    for x in sorted(set(xcc.tolist())):
        sel = xcc == x
        per_xcd.append({"xcc": int(x), "workgroups": int(sel.sum()), "last_end_ms": float(end[sel].max()),
                        "work_ms": float(dur[sel].sum()), "blockidx_mod8": sorted(set((block[ran][sel] % 8).tolist()))})
    # persistent workgroups: what lies between a brick's end stamp (taken before its sums are stored) and the same workgroup's
    # next start stamp (taken when it has found its next brick): the stores, the counter's round trip, the order entry
    gaps = []
    blk = block[ran]
    order_ = np.lexsort((start, blk))
    sb, ss, se = blk[order_], start[order_], end[order_]
    same = sb[1:] == sb[:-1]
    gaps = (ss[1:] - se[:-1])[same]
    between = {"n": int(gaps.size)}
    if gaps.size:
        between.update({"median_us": float(np.median(gaps) * 1e3), "mean_us": float(gaps.mean() * 1e3), "p90_us": float(np.percentile(gaps, 90) * 1e3),
                        "sum_over_workgroup_time": float(gaps.sum() / (dur.sum() + gaps.sum()))})
    # the bricks that end last: position in the order (heaviest level first), start and duration
    uid = np.nonzero(ran)[0]
    last = np.argsort(end)[-24:]
    cols_r, redo_r = n_cols[ran], n_redo[ran]
    late = [{"order_pos": int(uid[x]), "of": int(ran.size), "start_us": round(float(start[x]) * 1e3, 1), "dur_us": round(float(dur[x]) * 1e3, 1),
             "columns": int(cols_r[x]), "redone_voxels": int(redo_r[x])} for x in last]
    # duration against the number of views with a column of their own, and the redone voxels (all bricks)
    by_cols = []
    for lo_c in range(0, int(cols_r.max()) + 1, 8):
        sel = (cols_r >= lo_c) & (cols_r < lo_c + 8)
        if sel.any():
            by_cols.append({"columns": f"{lo_c}-{lo_c + 7}", "bricks": int(sel.sum()), "median_us": round(float(np.median(dur[sel])) * 1e3, 1),
                            "p99_us": round(float(np.percentile(dur[sel], 99)) * 1e3, 1), "mean_redone": round(float(redo_r[sel].mean()), 2),
                            "us_per_column": round(float(dur[sel].sum() / max(1, cols_r[sel].sum())) * 1e3, 2)})
    # duration by position in the order, twenty equal parts
    parts = np.array_split(np.arange(uid.size), 20)
    by_pos = [{"median_us": round(float(np.median(dur[q])) * 1e3, 1), "max_us": round(float(dur[q].max()) * 1e3, 1),
               "median_start_us": round(float(np.median(start[q])) * 1e3, 1)} for q in parts]
    brick_us = {f"p{q}": float(np.percentile(dur, q) * 1e3) for q in (1, 10, 25, 50, 75, 90, 99)}
    win_r, early_r = n_win[ran], n_win_early[ran]
    rec = {"late_bricks": late, "by_columns": by_cols, "redone_voxels_total_saturating": int(redo_r.sum()),
           "columns_total": int(cols_r.sum()), "window_pairs": int(win_r.sum()), "window_pairs_before_first_other_column": int(early_r.sum()),
           "bricks_with_only_window_columns": int(((win_r == cols_r) & (cols_r > 0)).sum()), "by_order_position": by_pos, "between_bricks": between, "brick_us": brick_us, "workload": args.workload, "scene": args.scene, "variant": args.variant, "kernel_ms_by_events": kernel_ms,
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
