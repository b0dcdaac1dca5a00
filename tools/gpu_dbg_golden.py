#!/usr/bin/env python3
"""Debug aid: one golden case through the default path without hit counters; prints where the grid differs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from cudadepthmapintegration_amd import capi, scene  # noqa: E402
from conftest import load_golden  # noqa: E402
from test_gpu_parity import _golden_inputs  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "noncubic_70x33x17"
g = load_golden(name)
grid, rp, views, thr = _golden_inputs(g)
print("ray", rp, "dims", grid.cell_dims, "views", views.n, views.depth.shape)
for variant in (0, capi.VARIANT_NO_INTERIOR):
    with capi.FusionContext(grid, rp, count_hits=False, kernel_variant=variant) as ctx:
        ctx.add_views(views, threshold=thr)
        if g.get("init_grid") is not None:
            ctx.upload_grid(g["init_grid"])
        ctx.fuse()
        out = ctx.download_grid()
        hist = ctx.mixed_reason_histogram() if hasattr(ctx, "mixed_reason_histogram") else None
    want = g["expected_grid"]
    bad = np.argwhere(out.view(np.uint64) != want.view(np.uint64))
    print("variant", variant, "differing voxels", len(bad), "of", out.size, "reasons", hist)
    for z, y, x in bad[:20]:
        print("  k,j,i", z, y, x, "got", out[z, y, x], "want", want[z, y, x], "delta", out[z, y, x] - want[z, y, x])
    if len(bad):
        print("  k range", bad[:, 0].min(), bad[:, 0].max(), "j", bad[:, 1].min(), bad[:, 1].max(), "i", bad[:, 2].min(), bad[:, 2].max())
