#!/usr/bin/env python3
"""Headline benchmark: Gvoxel-projections/s of the TSDF depth-map fusion path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg1|NxM@WxH]

A step = one pass of the hot path over one batch of synthetic input: zero the grid, fuse every
HBM-resident depth map of this rank into it (ONE kernel launch), and for N > 1 all-reduce the
f32 grid over RCCL (the path's only exchange step).  Depth maps are resident before the timed
region.  N = 1 workload = BASELINE.json configs[2]: 512^3 voxels x 256 depth maps of 1280x720.
For N > 1 every rank fuses its own shard of 256 maps (weak scaling: 256*N maps in total).

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      HBM view of the fusion launch: algorithmic bytes / hipEvent time vs 8 TB/s
  roofline_valu the binding roof of the per-voxel path: fp64 VALU issue (DESIGN.md "Roofline"), measured on
                the same workload with brick classes switched off (every projection computed), plus how many
                (brick, view) pairs the default path proved uniform
  ablation      the same fusion without brick classes / with workgroups in spatial order
  cpu_baseline  the CPU oracle (restated reference arithmetic) timed on this host's cores on a
                bounded sample of the same workload (N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # vendor peak, FMA counted as 2 flop
FLOP_PER_PROJECTION = 48.0  # algorithmic fp64 flop per voxel-projection
MAPREC_BYTES = 208  # per-map camera record read by the kernel (fusion_kernels.h)

WORKLOADS = {
    # name: (grid cells, maps per GPU, W, H)   -- BASELINE.json configs
    "cfg1": ((64, 64, 64), 4, 320, 240),
    "cfg2": ((256, 256, 256), 64, 640, 480),
    "cfg3": ((512, 512, 512), 256, 1280, 720),
    "cfg3vga": ((512, 512, 512), 256, 640, 480),
}


def parse_workload(s: str):
    if s in WORKLOADS:
        return WORKLOADS[s]
    grid_s, rest = s.split("x", 1)
    maps_s, wh = rest.split("@")
    w, h = wh.split("x")
    n = int(grid_s)
    return (n, n, n), int(maps_s), int(w), int(h)


def algorithmic_bytes(n_vox: int, n_maps: int, w: int, h: int, grid_bytes: int, depth_bytes: int) -> float:
    """B_alg of one fusion launch: grid written once, every depth value and camera
    record read once."""
    return float(grid_bytes * n_vox + depth_bytes * n_maps * w * h + MAPREC_BYTES * n_maps)


def cpu_baseline(grid, ray, views, target_seconds: float = 15.0):
    """Time the CPU oracle on all host cores on the first m maps over the full grid.  A one-map pass
    measures this host's rate first, so that m is sized for about `target_seconds` of CPU work."""
    from oracle import oracle

    cores = oracle.max_threads()
    n_vox = grid.n_voxels
    p = oracle.make_params(grid.cell_dims, grid.origin, grid.spacing, grid.grid_matrix, ray.thickness, ray.rho,
                           ray.eta, ray.delta, views.width, views.height)

    def run(m):
        depth = np.ascontiguousarray(views.depth[:m], dtype=np.float64)
        t0 = time.perf_counter()
        oracle.fuse(p, depth, views.K4[:m], views.RT4[:m], count_hits=False, n_threads=cores)
        return time.perf_counter() - t0

    probe = run(1)
    m = int(max(1, min(views.n, round(target_seconds / max(probe, 1e-3)))))
    dt = run(m)
    single = None
    if n_vox <= 2e8:  # one thread, one map over the full grid: a few seconds
        depth1 = np.ascontiguousarray(views.depth[:1], dtype=np.float64)
        t0 = time.perf_counter()
        oracle.fuse(p, depth1, views.K4[:1], views.RT4[:1], count_hits=False, n_threads=1)
        single = {"value": n_vox / (time.perf_counter() - t0) / 1e9, "cores": 1, "sample": "the first depth map over the full grid"}
    return {
        "single_thread": single,
        "value": n_vox * m / dt / 1e9,
        "unit": "Gvoxel-projections/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{grid.cell_dims[0]}x{grid.cell_dims[1]}x{grid.cell_dims[2]} voxels x first {m} of {views.n} depth "
                  f"maps, oracle/tsdf_oracle.c (gcc -O2 -ffp-contract=off), OpenMP over z on {cores} threads, {dt:.1f} s",
    }


def coloration_probe(scene, capi, n_vertices: int, W: int, H: int, n_views: int = 64):
    """Secondary measurement: the MeshColoration pass on the GPU with the colour planes
    resident (dmi_color_context), on synthetic vertices x views of the bench's image size.  `value` counts the
    kernels only (hipEvents); `seconds` is the whole dmi_color_process call, vertex upload and result download included."""
    views = scene.make_views(n_views, 8, 8, seed=77)          # cameras only; the depth tables are not used
    # per-pixel content is irrelevant to the timing: one byte pattern, tiled over all views (fast to generate)
    colors = np.empty((n_views, H, W, 3), dtype=np.uint8)
    colors[:] = (np.arange(H * W * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(1, H, W, 3)
    K4 = views.K4.copy()
    K4[:, 0, 0] = K4[:, 1, 1] = 0.9 * W
    K4[:, 0, 2], K4[:, 1, 2] = W / 2.0, H / 2.0
    pts = scene.make_mesh_points(n_vertices, seed=78)
    with capi.ColorContext() as c:
        c.add_views(colors, K4, views.RT4)
        c.process(pts[:1000])   # warm-up
        t0 = time.perf_counter()
        mean, median, count = c.process(pts)
        dt = time.perf_counter() - t0
        kms = c.kernel_ms()
    return {"value": n_vertices * n_views / (kms * 1e-3) / 1e9, "unit": "Gvertex-projections/s (kernels, views resident)",
            "vertices": n_vertices, "views": n_views, "image": f"{W}x{H}", "kernel_ms": kms, "seconds": dt,
            "value_call": n_vertices * n_views / dt / 1e9, "mean_views_per_vertex": float(count.mean())}


def end_to_end_probe(scene, capi, grid, ray, views, host_dtype, grid_dtype, chunk_views: int = 32):
    """PCIe-inclusive rate (never the headline value): depth tables start in pinned host memory, go up chunk by chunk on
    the context's upload stream while the previous chunk is being fused (dmi_add_views + dmi_fuse_range), and the grid
    comes back into pinned host memory.  host f64 / grid f64 is the reference's contract (vtkDoubleArray in and out)."""
    n = views.n
    np_host = np.float64 if host_dtype == "f64" else np.float32
    np_grid = np.float64 if grid_dtype == "f64" else np.float32
    pinned = capi.pinned_empty(views.depth.shape, np_host)
    pinned[:] = views.depth
    out = capi.pinned_empty((grid.n_voxels,), np_grid)
    times = []
    with capi.FusionContext(grid, ray, grid_dtype=grid_dtype, depth_storage="auto") as c:
        for rep in range(3):
            c.clear_views()
            c.reset_grid()
            c.synchronize()
            t0 = time.perf_counter()
            for v0 in range(0, n, chunk_views):
                v1 = min(n, v0 + chunk_views)
                c.add_views(scene.Views(pinned[v0:v1], views.K4[v0:v1], views.RT4[v0:v1]))
                c.fuse(v0, v1 - v0)
            c.download_grid(np_grid, out=out)
            times.append(time.perf_counter() - t0)
    dt = float(np.median(times[1:]))
    moved = pinned.nbytes + out.nbytes
    return {"host_depth": host_dtype, "grid": grid_dtype, "seconds": dt, "value": grid.n_voxels * n / dt / 1e9,
            "unit": "Gvoxel-projections/s including H2D of every depth table and D2H of the grid",
            "pcie_bytes": moved, "pcie_GBps_if_alone": moved / dt / 1e9, "chunk_views": chunk_views}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scene", default="dense", choices=["dense", "sparse"],
                    help="dense: background behind the sphere, ~every in-frustum voxel accumulates")
    ap.add_argument("--grid-dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ablation", action="store_true")
    ap.add_argument("--slabs", type=int, default=4,
                    help="N > 1: z-slabs per fusion; the all-reduce of a slab overlaps the fusion of the next (1 = no overlap)")
    ap.add_argument("--exchange", default="all_reduce", choices=["all_reduce", "reduce_scatter"],
                    help="N > 1: all_reduce = the contract (every rank gets the whole grid, overlapped slab by slab); "
                         "reduce_scatter = every rank gets the sum of its own 1/N of the grid (half the traffic, no overlap)")
    ap.add_argument("--no-coloration", action="store_true")
    ap.add_argument("--coloration-vertices", type=int, default=2_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--secondary", action="store_true", help="also time the other scene variant (N = 1)")
    ap.add_argument("--no-end-to-end", action="store_true")
    args = ap.parse_args()

    import torch

    from cudadepthmapintegration_amd import capi, scene, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the fusion path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl")

    cells, maps_per_gpu, W, H = parse_workload(args.workload)
    grid = scene.default_grid(cells)
    ray = scene.default_ray_potential(grid)
    n_vox = grid.n_voxels

    def make(scene_kind: str):
        return scene.make_views(maps_per_gpu, W, H, seed=1000 + rank, dense=(scene_kind == "dense"),
                                layout="sphere", dtype=np.float32)

    views = make(args.scene)

    torch_dtype = torch.float32 if args.grid_dtype == "f32" else torch.float64
    grid_t = torch.zeros(n_vox, dtype=torch_dtype, device="cuda")
    # an explicit (non-default) torch stream: the fusion kernel, the grid memset and the RCCL
    # all-reduce are all ordered on it, and its handle is non-NULL for the C ABI
    tstream = torch.cuda.Stream()
    torch.cuda.synchronize()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    ctx = capi.FusionContext(grid, ray, device=local_rank, grid_dtype=args.grid_dtype, depth_storage="auto",
                             kernel_variant=args.variant, stream=stream, external_grid=grid_t.data_ptr())
    t_up = time.perf_counter()
    ctx.add_views(views)
    upload_s = time.perf_counter() - t_up
    info = ctx.info()
    depth_bytes = 8 if info.depth_storage_in_use == capi.DMI_DEPTH_F64 else 4
    grid_bytes = 4 if args.grid_dtype == "f32" else 8

    comm_stream = torch.cuda.Stream() if dist is not None else None

    def step():
        ctx.reset_grid()
        if dist is None:
            ctx.fuse()
        elif args.exchange == "reduce_scatter":
            ctx.fuse()
            sharding.reduce_scatter_grid(grid_t, rank, world)
        elif args.slabs <= 1:
            ctx.fuse()
            sharding.all_reduce_grid(grid_t)  # the single RCCL all-reduce of the TSDF grid over xGMI
        else:
            # the same exchange, slab by slab: the all-reduce of slab i overlaps the fusion of slab i + 1
            sharding.fuse_and_all_reduce(ctx, grid_t, grid.cell_dims, args.slabs, tstream, comm_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps: int, warmup: int):
        for _ in range(warmup):
            step()
        barrier()
        k0 = ctx.timings()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        k1 = ctx.timings()
        kern_ms = (k1.total_fuse_kernel_ms - k0.total_fuse_kernel_ms) / max(1, steps)  # per step (a step may fuse in slabs)
        timed.main_ms = (k1.total_fuse_main_kernel_ms - k0.total_fuse_main_kernel_ms) / max(1, steps)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kern_ms

    dt, kern_ms = timed(args.steps, args.warmup)
    main_ms = timed.main_ms  # the fusion kernel proper (what rocprofv3 lists as fuse_tile_kernel / fuse_kernel)
    ms_per_step = dt / args.steps * 1e3
    total_maps = maps_per_gpu * world
    value = n_vox * total_maps * args.steps / dt / 1e9

    # ablations on the same resident views (N = 1): brick classes off = every voxel-projection computed
    ablation = None
    hist = ctx.brick_class_histogram()
    if world == 1 and not args.no_ablation:
        ablation = {}
        for name, var in (("no_brick_classes", capi.VARIANT_NO_BRICK_CLASSES), ("spatial_order", capi.VARIANT_SPATIAL_ORDER)):
            g2 = torch.zeros(n_vox, dtype=torch_dtype, device="cuda")
            c2 = capi.FusionContext(grid, ray, device=local_rank, grid_dtype=args.grid_dtype, depth_storage="auto",
                                    kernel_variant=args.variant | var, stream=stream, external_grid=g2.data_ptr())
            c2.add_views(views)
            for i in range(3):
                c2.reset_grid()
                c2.fuse()
                c2.synchronize()
                if i == 0:
                    k0 = c2.timings().total_fuse_kernel_ms
            ms = (c2.timings().total_fuse_kernel_ms - k0) / 2
            ablation[name] = {"kernel_ms": ms, "value": n_vox * maps_per_gpu / ms / 1e6}
            if name == "no_brick_classes":
                same = bool(torch.equal(g2, grid_t))
                ablation[name]["grid_bit_identical_to_default"] = same
            c2.close()
            del g2

    # the step right after the path: cell data -> point data of the fused grid, HBM-bound
    cell_to_point = None
    if world == 1:
        ts = []
        for _ in range(4):
            ctx.cell_to_point()
            ctx.synchronize()
            ts.append(ctx.timings().last_cell_to_point_ms)
        c2p_ms = float(np.median(ts[1:]))
        n_pts = (cells[0] + 1) * (cells[1] + 1) * (cells[2] + 1)
        c2p_bytes = float((4 if args.grid_dtype == "f32" else 8) * n_vox + 8 * n_pts)
        cell_to_point = {"kernel": "dmi::cell_to_point_kernel", "kernel_ms": c2p_ms, "bound": "hbm",
                         "algorithmic_bytes": c2p_bytes, "achieved": c2p_bytes / (c2p_ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": c2p_bytes / (c2p_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}

    secondary = None
    if args.secondary and world == 1:
        other = "sparse" if args.scene == "dense" else "dense"
        v2 = make(other)
        ctx.clear_views()
        ctx.add_views(v2)
        dt2, kern2 = timed(max(2, args.steps // 2), 1)
        secondary = {"scene": other, "value": n_vox * maps_per_gpu * max(2, args.steps // 2) / dt2 / 1e9,
                     "kernel_ms": kern2}
        del v2

    b_alg = algorithmic_bytes(n_vox, maps_per_gpu, W, H, grid_bytes, depth_bytes)
    achieved_gbps = b_alg / (main_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            rec = json.load(open(pmc_path)).get(f"{args.workload}:{args.scene}:{args.grid_dtype}")
            if rec:
                traffic = rec["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    proj_per_launch = float(n_vox) * maps_per_gpu
    # fp64 VALU view: meaningful for the path that computes every projection (brick classes off)
    valu_ms = ablation["no_brick_classes"]["kernel_ms"] if ablation else kern_ms
    valu_tflops = FLOP_PER_PROJECTION * proj_per_launch / (valu_ms * 1e-3) / 1e12

    out = {
        "metric": "Gvoxel-projections/s",
        "value": value,
        "unit": "Gvoxel-projections/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{cells[0]}x{cells[1]}x{cells[2]} voxels x {maps_per_gpu} depth maps {W}x{H} per GPU "
                        f"({args.workload}, {args.scene} sphere scene)",
            "grid_dtype": args.grid_dtype,
            "depth_storage": "f64" if depth_bytes == 8 else "f32",
            "k_mode": int(info.k_mode),
            "tiled_kernel": int(info.tiled_kernel),
            "kernel_variant": args.variant,
            "maps_total": total_maps,
            "parallelism": ((f"depth-map shards x{world}, RCCL reduce-scatter of the grid (each rank keeps 1/{world})"
                             if args.exchange == "reduce_scatter" else
                             f"depth-map shards x{world}, RCCL all-reduce of the grid in {args.slabs} z-slabs overlapped with "
                             f"the fusion") if world > 1 else "single GPU"),
            "host_upload_s": round(upload_s, 3),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved_gbps / HBM_PEAK_GBPS,
            "traffic": traffic,
            "kernel": "dmi::fuse_tile_kernel" if info.tiled_kernel else "dmi::fuse_kernel",
            "kernel_ms": main_ms,
            "fuse_ms": kern_ms,
            "algorithmic_bytes_per_launch": b_alg,
            "note": "kernel_ms = hipEvent time of the fusion kernel alone (the launch rocprofv3 lists under this name), "
                    "fuse_ms = all launches of one dmi_fuse (+ cz table, two classification passes, ordering); the path "
                    "is bound by fp64 VALU issue, not HBM: see roofline_valu and DESIGN.md",
        },
        "roofline_valu": {
            "bound": "valu_fp64",
            "achieved": valu_tflops,
            "peak": FP64_VECTOR_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": valu_tflops / FP64_VECTOR_PEAK_TFLOPS,
            "flop_per_projection": FLOP_PER_PROJECTION,
            "kernel_ms": valu_ms,
            "note": "per-voxel path only (brick classes off): 48 algorithmic fp64 flop x every voxel-projection / time; "
                    "the default path proves most (brick, view) pairs uniform and skips their projections",
        },
        "brick_classes": hist,
    }
    if ablation:
        out["ablation"] = ablation
    if secondary:
        out["secondary"] = secondary
    if cell_to_point:
        out["cell_to_point"] = cell_to_point
    if rank == 0 and world == 1 and not args.no_end_to_end:
        out["end_to_end"] = [end_to_end_probe(scene, capi, grid, ray, views, "f32", "f32"),
                             end_to_end_probe(scene, capi, grid, ray, views, "f64", "f64")]
    if rank == 0 and world == 1 and not args.no_coloration:
        out["coloration"] = coloration_probe(scene, capi, args.coloration_vertices, W, H)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(grid, ray, views, args.cpu_seconds)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
