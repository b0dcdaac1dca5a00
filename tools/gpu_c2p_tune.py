#!/usr/bin/env python3
"""Times dmi_cell_to_point at 512^3 for the column heights x block heights a tuning build holds (DMI_C2P_KZ, DMI_C2P_BY),
f32 and f64 grids.  Needs the tuning library: DMI_TUNING=1 at build time and in the environment of this script."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cudadepthmapintegration_amd import capi, scene  # noqa: E402

cells = (512, 512, 512)
grid = scene.default_grid(cells)
ray = scene.default_ray_potential(grid)
res = []
for dtype in ("f32", "f64"):
    # a caller-owned grid: the context recomputes the point data on every call (nothing invalidates by memset here)
    g = torch.randn(512 ** 3, dtype=torch.float32 if dtype == "f32" else torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx = capi.FusionContext(grid, ray, grid_dtype=dtype, external_grid=g.data_ptr())
    nbytes = (4 if dtype == "f32" else 8) * 512 ** 3 + 8 * 513 ** 3
    for spec in sys.argv[1:] or ["1x4", "2x4", "4x4", "8x4", "16x4", "4x2", "8x2", "16x2", "4x8", "8x8", "8x1", "16x1"]:
        kz, by = spec.split("x")
        os.environ["DMI_C2P_KZ"], os.environ["DMI_C2P_BY"] = kz, by
        ts = []
        for _ in range(21):
            ctx.cell_to_point()
            ctx.synchronize()
            ts.append(ctx.timings().last_cell_to_point_ms)
        ms = float(np.median(ts[1:]))
        rec = {"grid": dtype, "kz": int(kz), "by": int(by), "ms": ms, "min_ms": float(min(ts[1:])), "GBps": nbytes / ms / 1e6}
        res.append(rec)
        print(json.dumps(rec), flush=True)
    ctx.close()
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "c2p_tune.json"), "w"), indent=1)
