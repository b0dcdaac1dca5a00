import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_fuzz import _random_case
from cudadepthmapintegration_amd import capi
from oracle import oracle
from helpers import oracle_params_from_scene
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
grid, rp, views = _random_case(seed)
print("dims", grid.cell_dims, "gm", np.diag(grid.grid_matrix), grid.grid_matrix[:3,3], "rp", rp, "n", views.n, views.width, views.height)
print("K", views.K4[0][:3,:3].tolist())
init = np.random.default_rng(seed).normal(size=(grid.cell_dims[2], grid.cell_dims[1], grid.cell_dims[0])) if seed % 5 == 0 else None
with np.errstate(all="ignore"):
    want, vh_w, mh_w = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4, init_grid=init)
for variant in (0, 256, 16):
    with capi.FusionContext(grid, rp, count_hits=True, kernel_variant=variant) as ctx:
        if init is not None: ctx.upload_grid(init)
        ctx.add_views(views); ctx.fuse(); out = ctx.download_grid(); info = ctx.info(); hist = ctx.brick_class_histogram()
    bad = np.argwhere(out.view(np.uint64) != want.view(np.uint64))
    print("variant", variant, "tiled", info.tiled_kernel, "storage", info.depth_storage_in_use, "hist", hist, "mismatches", len(bad))
    for b in bad[:6]:
        k, j, i = b
        print("   voxel", (i, j, k), "got", out[k, j, i], "want", want[k, j, i], "diff", out[k,j,i]-want[k,j,i], "hits", vh_w[k,j,i])
    # per-map: isolate which map
    if len(bad) and variant == 0:
        k, j, i = bad[0]
        for m in range(views.n):
            sub = views.subset(m, m+1)
            w1, _, _ = oracle.fuse(oracle_params_from_scene(grid, rp, sub), sub.depth, sub.K4, sub.RT4)
            with capi.FusionContext(grid, rp, kernel_variant=0) as ctx:
                ctx.add_views(sub); ctx.fuse(); o1 = ctx.download_grid(); h1 = ctx.brick_class_histogram()
            nb = (o1.view(np.uint64) != w1.view(np.uint64)).sum()
            print("    map", m, "mismatches", nb, "hist", h1, "got", o1[k,j,i], "want", w1[k,j,i])
