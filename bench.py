#!/usr/bin/env python3
"""Headline benchmark: Gvoxel-projections/s of the TSDF depth-map fusion path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg1|NxM@WxH] [--scaling weak|strong]

A step = one pass of the hot path over one batch of synthetic input: zero the grid, fuse every HBM-resident depth map
of this rank into it and, for N > 1, sum the f32 grids of the ranks over RCCL (the path's only exchange step; issued
by the library itself, dmi_multi_fuse, slab by slab behind the fusion).  Depth maps are resident before the timed
region.  N = 1 workload = BASELINE.json configs[2]: 512^3 voxels x 256 depth maps of 1280x720, SURVEY.md 8d's scene: a sphere in
front of a background, best-cost values ~ U[0,1) with the threshold that invalidates ~10 % of the pixels (--scene speckle).

N > 1: one process per GPU.  Launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) the script joins as a rank; launched plainly (`python bench.py --gpus 2`) it starts the N rank
processes itself BEFORE anything touches a GPU and relays rank 0's line.  torch.distributed (gloo, CPU) carries only
the rendezvous: the RCCL unique id, the barriers and the max over ranks; every GPU operation is the C ABI's.
  weak scaling (top level by default): every rank fuses `maps` views of its own -> N x maps views in total
  strong scaling ("strong" objects):   the fixed problems of BASELINE.json -- cfg3's 256 views and cfg4's 1024 VGA
                                       views -- split over the N ranks; the same views whatever N is, and rank 0
                                       checks the N-rank grid of cfg3 against its own single-GPU fusion of all views

Rank 0 prints ONE JSON line (contract in the task description) with extra objects:
  roofline      HBM view of the fusion launch: algorithmic bytes / hipEvent time vs 8 TB/s; flop_frac_* say how much of
                SURVEY 8d's arithmetic the timed kernel executes (the brick classes prove most of it away)
  scenes        the other scene kinds on the same context: dense (every pixel valid), speckle (the default: 10 % of the
                pixels invalidated by the best-cost threshold), noisy (+ depth noise and holes), room (a second
                geometry: cameras inside the grid looking outward at walls)
  roofline_valu the binding roof of the per-voxel path: fp64 VALU issue (DESIGN.md "Roofline"), measured on
                the same workload with brick classes switched off (every projection computed)
  roofline_issue what bounds the default path: vector / scalar instruction issue -- vector-pipe busy cycles and SALU
                counts per launch from the committed PMC passes of the same workload (profiles/pmc_traffic.json)
                against this run's kernel time
  box_state     the fp64 vector rate the box sustains on bare FMAs right after the timed steps (dmi_fp64_probe)
  ablation      the same fusion without brick classes / with workgroups in spatial order
  end_to_end, cell_to_point, coloration   the PCIe-inclusive contract figures against their floor; the passes either side
  cpu_baseline  the CPU oracle (restated reference arithmetic) timed on this host's cores on a
                bounded sample of the same workload (N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # vendor peak, FMA counted as 2 flop
L2_GATHER_PEAK_GBPS = 17800.0  # MI355X_MICROARCH.md "Indexed rows": rows served by the XCDs' L2, 16.8-18.8 TB/s chip-wide
FLOP_PER_PROJECTION = 48.0  # algorithmic fp64 flop per voxel-projection
MAPREC_BYTES = 208  # per-map camera record read by the kernel (fusion_kernels.h)

SCENE_KINDS = ("dense", "sparse", "speckle", "noisy", "room", "blobs", "geo")  # scene.SCENE_KINDS (scene.py is imported after the argument parser)
SCENE_SEED = 1000


HOLE_FRACTION = None  # --hole-fraction: the share of pixels without a depth in the speckle / noisy / room / blobs scenes (default 0.1)


def upload_scene(ctx, scene, kind: str, n: int, W: int, H: int, spacing: float, keep_host: bool = False, chunk: int = 32,
                 only_first_chunk: bool = False):
    """The n views of scene `kind` onto ctx, `chunk` views at a time (bounded host memory).  Scenes with best-cost values go
    through dmi_add_views with the threshold, as the reference's driver applies it per view (cu:348); the others through the
    f32 entry point.  keep_host: also return the views as the device now holds them (f32, thresholded) for the CPU baseline
    and the secondary contexts."""
    kept = []
    for c0 in range(0, n, chunk):
        extra = {} if HOLE_FRACTION is None else {"speckle": float(HOLE_FRACTION)}
        v, thr = scene.make_scene_views(kind, n, W, H, seed=SCENE_SEED, view_range=(c0, min(n, c0 + chunk)), noise_sigma=spacing, **extra)
        if thr is None:
            v = scene.Views(v.depth.astype(np.float32), v.K4, v.RT4)
            ctx.add_views(v)
        else:
            ctx.add_views(v, threshold=thr)
        if only_first_chunk:
            return None
        if keep_host:
            d = v.depth.astype(np.float32)
            if thr is not None:
                d[v.best_cost > thr] = -1.0   # RD.cxx:159-166
            kept.append(scene.Views(d, v.K4, v.RT4))
    if not keep_host:
        return None
    return scene.Views(np.concatenate([k.depth for k in kept]), np.concatenate([k.K4 for k in kept]),
                       np.concatenate([k.RT4 for k in kept]))


WORKLOADS = {
    # name: (grid cells, maps per GPU, W, H)   -- BASELINE.json configs
    "cfg1": ((64, 64, 64), 4, 320, 240),
    "cfg2": ((256, 256, 256), 64, 640, 480),
    "cfg3": ((512, 512, 512), 256, 1280, 720),
    "cfg3vga": ((512, 512, 512), 256, 640, 480),
    "cfg4": ((512, 512, 512), 1024, 640, 480),
    "cfg5share": ((1024, 1024, 1024), 64, 1920, 1080),
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

    def run(m, threads):
        depth = np.ascontiguousarray(views.depth[:m], dtype=np.float64)
        t0 = time.perf_counter()
        oracle.fuse(p, depth, views.K4[:m], views.RT4[:m], count_hits=False, n_threads=threads)
        return time.perf_counter() - t0

    probe = run(2, cores)
    m = int(max(2, min(views.n, round(2 * target_seconds / max(probe, 1e-3)))))
    dt = run(m, cores)
    single = None
    efficiency = None
    if n_vox <= 2e8:  # one thread, one map over the full grid: a few seconds
        dt1 = run(1, 1)
        single = {"value": n_vox / dt1 / 1e9, "cores": 1, "sample": "the first depth map over the full grid"}
        efficiency = (n_vox * m / dt / 1e9) / (cores * single["value"])
    return {
        "single_thread": single,
        "value": n_vox * m / dt / 1e9,
        "unit": "Gvoxel-projections/s",
        "cores": cores,
        "kind": "port",
        "scaling_efficiency": efficiency,
        "sample": f"{grid.cell_dims[0]}x{grid.cell_dims[1]}x{grid.cell_dims[2]} voxels x first {m} of {views.n} depth "
                  f"maps, oracle/tsdf_oracle.c (gcc -O2 -ffp-contract=off), one OpenMP region, z-layers dealt dynamically "
                  f"to {cores} threads, every layer fused through all maps by the thread that first touches it, {dt:.1f} s; "
                  f"scaling_efficiency = this rate / ({cores} x the single-thread rate)",
    }


def coloration_probe(scene, capi, n_vertices: int, W: int, H: int, n_views: int = 64, pcie=None):
    """Secondary measurement: the MeshColoration pass on the GPU with the colour planes
    resident (dmi_color_context), on synthetic vertices x views of the bench's image size.  `value` counts the
    kernels only (hipEvents); `seconds` is the whole dmi_color_process call, vertex upload and result download included.
    Two vertex orders: random points (every lane gathers from a different place in every plane) and the same points
    sorted along a space-filling curve, as the vertices of a real mesh are (neighbours in neighbouring lanes)."""
    views = scene.make_views(n_views, 8, 8, seed=77)          # cameras only; the depth tables are not used
    # per-pixel content is irrelevant to the timing: one byte pattern, tiled over all views (fast to generate)
    colors = np.empty((n_views, H, W, 3), dtype=np.uint8)
    colors[:] = (np.arange(H * W * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(1, H, W, 3)
    K4 = views.K4.copy()
    K4[:, 0, 0] = K4[:, 1, 1] = 0.9 * W
    K4[:, 0, 2], K4[:, 1, 2] = W / 2.0, H / 2.0
    # (vertices and results in pinned host memory, as the depth tables of end_to_end: the call's copies are DMA transfers, and a
    # chunk's copies run beside its neighbours' kernels)
    pts = capi.pinned_empty((n_vertices, 3), np.float64)
    pts[:] = scene.make_mesh_points(n_vertices, seed=78)
    ordered = capi.pinned_empty((n_vertices, 3), np.float64)
    ordered[:] = pts[scene.morton_order(pts)]
    results = (capi.pinned_empty((n_vertices, 3), np.uint8), capi.pinned_empty((n_vertices, 3), np.uint8), capi.pinned_empty((n_vertices,), np.int32))
    out = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_coloration.json")))
    except Exception:
        pmc = {}
    with capi.ColorContext() as c:
        c.add_views(colors, K4, views.RT4)
        c.process(pts[:1000])   # warm-up
        for name, p in (("random_vertices", pts), ("mesh_ordered_vertices", ordered), ("random_vertices_reordered_on_device", pts)):
            c.set_vertex_reorder(name.endswith("on_device"))   # dmi_color_set_vertex_reorder: Z-order processing inside the library
            c.process(p, out=results)   # (the first call in a mode sizes the work buffers)
            calls = []
            for _ in range(5):          # the median of five calls (the process's first large copies on a stream take milliseconds)
                t0 = time.perf_counter()
                mean, median, count = c.process(p, out=results)
                calls.append((time.perf_counter() - t0, c.kernel_ms()))
            calls.sort()
            dt, kms = calls[len(calls) // 2]
            seconds_all = [round(x[0], 6) for x in calls]
            pcie_floor = (n_vertices * 24 / (pcie[0] * 1e9) + n_vertices * 10 / (pcie[1] * 1e9)) if pcie else None
            hits = int(count.sum(dtype=np.int64))
            # algorithmic traffic: every vertex read once (24 B), one RGBA texel gathered per (vertex, view) hit (4 B),
            # the three outputs written once (3 + 3 + 4 B)
            b_alg = float(n_vertices * 24 + hits * 4 + n_vertices * 10)
            l2 = None
            key = {"random_vertices": "random", "mesh_ordered_vertices": "mesh", "random_vertices_reordered_on_device": "reordered"}.get(name)
            if key and pmc.get(key, {}).get("counters_project_and_median_kernels", {}).get("TCP_TCC_READ_REQ_sum") \
                    and (pmc[key].get("vertices"), pmc[key].get("views")) == (n_vertices, n_views):
                req_bytes = pmc[key]["counters_project_and_median_kernels"]["TCP_TCC_READ_REQ_sum"] * 64.0
                l2 = {"bound": "l2_requests", "request_bytes": req_bytes, "achieved": req_bytes / (kms * 1e-3) / 1e9,
                      "peak": L2_GATHER_PEAK_GBPS, "unit": "GB/s", "frac": req_bytes / (kms * 1e-3) / 1e9 / L2_GATHER_PEAK_GBPS,
                      "source": "TCP_TCC_READ_REQ_sum x 64 B of the projection + median kernels from profiles/pmc_coloration.json "
                                "(rocprofv3 --pmc over tools/gpu_coloration_pmc.py at an earlier commit, NOT measured in this run) "
                                "over this run's kernel time; peak = the L2-resident gather rate MI355X_MICROARCH.md measures "
                                "(16.8-18.8 TB/s chip-wide)"}
            out[name] = {"value": n_vertices * n_views / (kms * 1e-3) / 1e9, "kernel_ms": kms, "seconds": dt, "roofline_l2": l2,
                         "pcie_floor_s": pcie_floor, "seconds_over_floor_plus_kernels": (dt / (pcie_floor + kms * 1e-3)) if pcie_floor else None,
                         "seconds_of_five_calls": seconds_all,
                         "value_call": n_vertices * n_views / dt / 1e9, "mean_views_per_vertex": float(count.mean()),
                         "roofline": {"bound": "hbm", "algorithmic_bytes": b_alg, "achieved": b_alg / (kms * 1e-3) / 1e9,
                                      "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": b_alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
    out.update({"unit": "Gvertex-projections/s (kernels, views resident)", "vertices": n_vertices, "views": n_views,
                "image": f"{W}x{H}",
                "note": "algorithmic_bytes = 24 B per vertex + 4 B per (vertex, view) pair inside an image + 10 B of "
                        "outputs per vertex; the gathers are scattered 4-byte reads, so the HBM fraction is low by nature: "
                        "roofline_l2 prices the same kernels by their L2 requests instead"})
    # keep the round-1 top-level keys (random vertices) for continuity
    out["value"] = out["random_vertices"]["value"]
    out["kernel_ms"] = out["random_vertices"]["kernel_ms"]
    return out


def end_to_end_probe(scene, capi, grid, ray, views, host_dtype, grid_dtype, pcie, chunk_views: int = 32, download_slabs: int = 8):
    """PCIe-inclusive rate (never the headline value): depth tables start in pinned host memory, go up chunk by chunk on
    the context's upload stream while the previous chunk is being fused (dmi_add_views + dmi_fuse_range), and the grid
    comes back into pinned host memory, slab by slab under the last chunk's fusion (dmi_fuse_range_download: what
    FusionDriver::ProcessDepthMap calls).  host f64 / grid f64 is the reference's contract (vtkDoubleArray in and out).
    pcie_floor_s = the same bytes at the copy rates dmi_pcie_probe measured on this box, nothing else counted."""
    n = views.n
    np_host = np.float64 if host_dtype == "f64" else np.float32
    np_grid = np.float64 if grid_dtype == "f64" else np.float32
    pinned = capi.pinned_empty(views.depth.shape, np_host)
    pinned[:] = views.depth
    out = capi.pinned_empty((grid.n_voxels,), np_grid)
    times = []
    with capi.FusionContext(grid, ray, grid_dtype=grid_dtype, depth_storage="auto") as c:
        for rep in range(5):  # (the first is the warm-up; one of the others in three runs is 3 ms late: the median of four)
            c.clear_views()
            c.reset_grid()
            c.synchronize()
            t0 = time.perf_counter()
            for v0 in range(0, n, chunk_views):
                v1 = min(n, v0 + chunk_views)
                c.add_views(scene.Views(pinned[v0:v1], views.K4[v0:v1], views.RT4[v0:v1]))
                if v1 < n:
                    c.fuse(v0, v1 - v0)
                else:  # the last chunk: fused slab by slab, every slab on its way back under the next one's fusion
                    c.fuse_download(v0, v1 - v0, np_grid, out=out, n_slabs=download_slabs)
            times.append(time.perf_counter() - t0)
    dt = float(np.median(times[1:]))
    moved = pinned.nbytes + out.nbytes
    floor = pinned.nbytes / (pcie[0] * 1e9) + out.nbytes / (pcie[1] * 1e9)
    return {"host_depth": host_dtype, "grid": grid_dtype, "seconds": dt, "value": grid.n_voxels * n / dt / 1e9,
            "unit": "Gvoxel-projections/s including H2D of every depth table and D2H of the grid",
            "pcie_bytes": moved, "pcie_GBps_if_alone": moved / dt / 1e9, "chunk_views": chunk_views, "download_slabs": download_slabs,
            "pcie_floor_s": floor, "seconds_over_floor": dt / floor, "seconds_of_calls": [round(t, 6) for t in times[1:]]}


# ---- launching -------------------------------------------------------------------------------------------------
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters, before this
    process has made any GPU call -- it never makes one) and relay rank 0's output."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DMI_BENCH_LAUNCHED_BY="bench.py")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies takes the others with it (they would wait in the rendezvous for ever): poll, do not just wait
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                rc = max(rc, abs(code))
                for q in alive:
                    q.terminate()       # the exact processes this launcher started
        time.sleep(0.2)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scene", default="speckle", choices=list(SCENE_KINDS),
                    help="speckle (default): SURVEY.md 8d's input -- the dense sphere scene with best-cost values ~ U[0,1) and the "
                         "threshold that turns ~10 %% of the pixels into 'no depth', applied by dmi_add_views as the reference's "
                         "filter applies it (RD.cxx:138-167); dense: every pixel holds a depth; sparse: sphere only; noisy: speckle "
                         "+ one voxel of depth noise + holes; room: a second geometry -- cameras inside the grid looking outward at the "
                         "walls of a room, depths over an order of magnitude, grazing walls, 10 %% speckle")
    ap.add_argument("--hole-fraction", type=float, default=None,
                    help="share of the pixels without a depth in the speckle / noisy / room scenes (scattered at random) and the blobs "
                         "scene (in discs); default 0.1 -- the hole-density sweep of profiles/")
    ap.add_argument("--no-scenes", action="store_true", help="N = 1: skip the `scenes` object (the other scene kinds, timed beside the headline)")
    ap.add_argument("--grid-dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: which measurement is the top-level line (the other one is reported beside it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ablation", action="store_true")
    ap.add_argument("--slabs", type=int, default=4,
                    help="N > 1: z-slabs per fusion; the all-reduce of a slab overlaps the fusion of the next (1 = no overlap)")
    ap.add_argument("--exchange", default="all_reduce", choices=["all_reduce", "reduce_scatter", "peer_copy"],
                    help="N > 1: all_reduce = the contract (every rank gets the whole grid, overlapped slab by slab); "
                         "reduce_scatter = every rank gets the sum of its own 1/N of the grid (half the traffic, no overlap); "
                         "peer_copy = the all-reduce by peer-to-peer copies and a sum kernel behind the fusion, no RCCL "
                         "(ranks of one process: implies --one-process)")
    ap.add_argument("--partition", default="views", choices=["views", "z_slabs"],
                    help="N > 1: views = the north star's depth-map shards + exchange; z_slabs = every rank fuses all views into "
                         "its own cell layers, no collective")
    ap.add_argument("--one-process", action="store_true",
                    help="N > 1: one process drives all N devices (dmi_multi_create / ncclCommInitAll) instead of one process per GPU")
    ap.add_argument("--share-device", action="store_true",
                    help="--exchange peer_copy or --partition z_slabs (no communicator): every rank on device 0 -- a rehearsal of the "
                         "N-rank run, launcher and all, on a one-GPU box; the figures say nothing about N GPUs")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling objects")
    ap.add_argument("--force-multi", action="store_true",
                    help="rehearsal on a one-GPU box: run the N > 1 code path (dmi_multi_*, RCCL with one rank) with --gpus 1")
    ap.add_argument("--no-coloration", action="store_true")
    ap.add_argument("--coloration-vertices", type=int, default=2_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-end-to-end", action="store_true")
    args = ap.parse_args()
    global HOLE_FRACTION
    HOLE_FRACTION = args.hole_fraction

    env_world = os.environ.get("WORLD_SIZE")
    if args.exchange == "peer_copy":
        args.one_process = True   # DMI_EXCHANGE_PEER_COPY: every rank in one process (no IPC handles between processes)
        if env_world is not None and int(env_world) > 1:
            raise SystemExit("bench.py --exchange peer_copy drives every GPU from ONE process: start it plainly, not under a launcher")
    if args.gpus > 1 and env_world is None and not args.one_process:
        sys.exit(self_launch(args.gpus))
    # ONE JSON line on stdout: native libraries that write to file descriptor 1 (RCCL prints a version banner there) are
    # sent to stderr for the whole run; the line itself goes to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(obj):
        real_stdout.write(json.dumps(obj) + "\n")
        real_stdout.flush()

    world = 1 if args.one_process else int(env_world or "1")
    rank = 0 if args.one_process else int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_process else int(os.environ.get("LOCAL_RANK", "0"))
    if not args.one_process and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")

    import torch  # before the HIP library: one HIP runtime per process (capi.load explains)

    from cudadepthmapintegration_amd import capi, scene

    n_dev = capi.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU: the fusion path has no CPU fallback")
    n_ranks = args.gpus  # ranks of the fusion = GPUs, however they are spread over processes
    # rehearsals on a one-GPU box: the exchanges that need no communicator may put every rank on device 0
    # (with an RCCL exchange the communicator refuses two ranks on one device: that rehearses the fallback of multi_gpu)
    share = args.share_device
    if share:
        local_rank = 0
    if (n_dev < args.gpus and not share) or local_rank >= n_dev:   # one node: every rank sees every GPU, so every rank decides alike
        raise SystemExit(f"bench.py: {args.gpus} GPUs asked for, {n_dev} visible")
    have_torch_gpu = torch.cuda.is_available()
    if have_torch_gpu:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="gloo")  # rendezvous only: unique id, barriers, max over ranks (all on the CPU)

    def device_sync():
        if have_torch_gpu:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    cells, maps_per_gpu, W, H = parse_workload(args.workload)
    grid = scene.default_grid(cells)
    ray = scene.default_ray_potential(grid)
    if args.scene == "geo":
        args.no_scenes = True  # (the other scene kinds live in the unit cube: they are not timed beside this one)
    if args.scene == "geo":  # real-world magnitudes: the speckle scene, grid and all, in a geo-referenced frame (scene.to_world_frame)
        grid, ray = scene.geo_grid(grid, ray)
    n_vox = grid.n_voxels
    grid_bytes = 4 if args.grid_dtype == "f32" else 8
    np_grid = np.float32 if args.grid_dtype == "f32" else np.float64

    if n_ranks > 1 or args.force_multi:
        out = multi_gpu(args, capi, scene, dist, barrier, device_sync, max_over_ranks, world, rank, local_rank, n_ranks)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit(out)
        return

    # ---------------------------------------------------------------- N = 1 -----------------------------------
    spacing = float(max(grid.spacing))
    ctx = capi.FusionContext(grid, ray, device=local_rank, grid_dtype=args.grid_dtype, depth_storage="auto",
                             kernel_variant=args.variant)
    # one view up and away again first: a process's first launch of the upload kernel carries ~1 ms of one-time cost (its code
    # object is loaded at that launch: profiles/r17i_upload_kernel_ms_by_call.json), which is not the upload pass's
    upload_scene(ctx, scene, args.scene, maps_per_gpu, W, H, spacing, chunk=1, only_first_chunk=True)
    ctx.clear_views()
    upload_first_call_ms = ctx.upload_kernel_ms()[1]
    t_up = time.perf_counter()
    views = upload_scene(ctx, scene, args.scene, maps_per_gpu, W, H, spacing, keep_host=True)
    upload_s = time.perf_counter() - t_up
    upload_kernels_ms = ctx.upload_kernel_ms()[1] - upload_first_call_ms
    info = ctx.info()
    depth_bytes = 8 if info.depth_storage_in_use == capi.DMI_DEPTH_F64 else 4

    def step():
        ctx.reset_grid()
        ctx.fuse()

    def timed(steps: int, warmup: int):
        for _ in range(warmup):
            step()
        ctx.synchronize()
        device_sync()
        barrier()
        k0 = ctx.timings()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.synchronize()
        device_sync()
        barrier()
        dt = time.perf_counter() - t0
        k1 = ctx.timings()
        kern_ms = (k1.total_fuse_kernel_ms - k0.total_fuse_kernel_ms) / max(1, steps)
        timed.main_ms = (k1.total_fuse_main_kernel_ms - k0.total_fuse_main_kernel_ms) / max(1, steps)
        return max_over_ranks(dt), kern_ms

    dt, kern_ms = timed(args.steps, args.warmup)
    main_ms = timed.main_ms  # the fusion kernel proper (what rocprofv3 lists as fuse_tile_kernel / fuse_kernel)
    # what this box's vector units sustain right now on bare fp64 FMAs: boxes of the same model differ (the same library
    # has measured 5.0 and 7.1 ms per fusion kernel within the hour), and the fusion kernel is bound by fp64 vector issue
    fp64_now = capi.fp64_probe(local_rank, 20.0)
    ms_per_step = dt / args.steps * 1e3
    value = n_vox * maps_per_gpu * args.steps / dt / 1e9

    # ablations on the same resident views: brick classes off = every voxel-projection computed
    ablation = None
    hist = ctx.brick_class_histogram()
    if not args.no_ablation:
        ablation = {}
        default_grid = ctx.download_grid(np_grid).copy()
        for name, var in (("no_brick_classes", capi.VARIANT_NO_BRICK_CLASSES), ("spatial_order", capi.VARIANT_SPATIAL_ORDER)):
            c2 = capi.FusionContext(grid, ray, device=local_rank, grid_dtype=args.grid_dtype, depth_storage="auto",
                                    kernel_variant=args.variant | var)
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
                other = c2.download_grid(np_grid)
                ablation[name]["grid_bit_identical_to_default"] = bool(
                    np.array_equal(other.view(np.uint32 if grid_bytes == 4 else np.uint64),
                                   default_grid.view(np.uint32 if grid_bytes == 4 else np.uint64)))
                del other
            c2.close()
        del default_grid

    # the step right after the path: cell data -> point data of the fused grid, HBM-bound
    ts = []
    for _ in range(4):
        step()                      # a context-owned grid keeps its point data until the grid changes: change it
        ctx.cell_to_point()
        ctx.synchronize()
        ts.append(ctx.timings().last_cell_to_point_ms)
    c2p_ms = float(np.median(ts[1:]))
    n_pts = (cells[0] + 1) * (cells[1] + 1) * (cells[2] + 1)
    c2p_bytes = float(grid_bytes * n_vox + 8 * n_pts)
    cell_to_point = {"kernel": "dmi::cell_to_point_kernel", "kernel_ms": c2p_ms, "bound": "hbm",
                     "algorithmic_bytes": c2p_bytes, "achieved": c2p_bytes / (c2p_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": c2p_bytes / (c2p_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}

    # the other scene kinds on the same context, timed like the headline (fewer steps): what the path does when the depth
    # tables change character -- every pixel valid (dense), 10 % invalid speckle (speckle), speckle + noise + holes (noisy)
    scenes = None
    if not args.no_scenes:
        scenes = {}
        resident = args.scene
        for kind in ("dense", "speckle", "noisy", "room"):
            if kind != resident:
                ctx.clear_views()
                upload_scene(ctx, scene, kind, maps_per_gpu, W, H, spacing)
                resident = kind
            n_steps = max(2, args.steps // 2)
            dt2, kern2 = timed(n_steps, 1)
            bc = ctx.brick_class_histogram()
            pairs = sum(bc.values())
            scenes[kind] = {"value": n_vox * maps_per_gpu * n_steps / dt2 / 1e9, "ms_per_step": dt2 / n_steps * 1e3,
                            "fuse_ms": kern2, "kernel_ms": timed.main_ms, "brick_classes": bc,
                            "mixed_reasons": ctx.mixed_reason_histogram(), "window_pairs": ctx.window_pair_count(),
                            # the launch shape the library picked for these depth maps (dmi_capi.hip): voxels per column, from the
                            # number of (brick, view) pairs it classified
                            "column_height": (int(round(n_vox * maps_per_gpu / 64 / pairs)) if pairs else None)}
        if resident != args.scene:   # leave the context as the sections below expect it: the headline scene resident
            ctx.clear_views()
            upload_scene(ctx, scene, args.scene, maps_per_gpu, W, H, spacing)

    b_alg = algorithmic_bytes(n_vox, maps_per_gpu, W, H, grid_bytes, depth_bytes)
    achieved_gbps = b_alg / (main_ms * 1e-3) / 1e9
    traffic = None
    issue_counts = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            rec = json.load(open(pmc_path)).get(f"{args.workload}:{args.scene}:{args.grid_dtype}")
            if rec:
                traffic = rec["hbm_bytes_per_launch"]
                if rec.get("valu_insts"):
                    issue_counts = rec
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
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
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
            "hole_fraction": 0.1 if args.hole_fraction is None else args.hole_fraction,
            "maps_total": maps_per_gpu,
            "parallelism": "single GPU",
            "rccl_ranks": 0,
            "host_upload_s": round(upload_s, 3),
            # hipEvent time of the upload pass's kernels over all the views (one kernel per chunk: threshold, row flip, narrowing,
            # pyramid, validity bytes and bits), outside the timed region by the metric's definition
            "upload_kernels_ms": round(upload_kernels_ms, 3), "upload_kernels_first_call_of_the_process_ms": round(upload_first_call_ms, 3),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved_gbps / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": (f"profiles/pmc_traffic.json [{args.workload}:{args.scene}:{args.grid_dtype}]: FETCH_SIZE / WRITE_SIZE "
                               "passes of rocprofv3 over this workload at an earlier commit, NOT measured in this run") if traffic else None,
            # the same launch against the fp64 vector peak, by SURVEY.md 8d's count of 48 flop per voxel-projection: above 1
            # means the timed kernel does not execute the algorithmic work -- the brick classes prove most (brick, view)
            # pairs uniform and never project them (brick_classes); the path that does project every pair is beside it
            "flop_frac_default_path": FLOP_PER_PROJECTION * proj_per_launch / (main_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            "flop_frac_per_voxel_path": valu_tflops / FP64_VECTOR_PEAK_TFLOPS if ablation else None,
            "kernel": "dmi::fuse_tile_kernel" if info.tiled_kernel else "dmi::fuse_kernel",
            "kernel_ms": main_ms,
            "fuse_ms": kern_ms,
            "algorithmic_bytes_per_launch": b_alg,
            "note": "kernel_ms = hipEvent time of the fusion kernel alone (the launch rocprofv3 lists under this name), "
                    "fuse_ms = all launches of one dmi_fuse (+ two classification passes -- the first fills the launch's tables --, window origins, two ordering launches); "
                    "the path is bound by instruction issue (vector, then scalar) -- not by HBM: see roofline_issue, "
                    "roofline_valu and DESIGN.md 9",
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
        "mixed_reasons": ctx.mixed_reason_histogram(),
        "window_pairs": ctx.window_pair_count(),
        "view_paths": ctx.view_paths(),
        # what actually bounds the default path: instruction issue (issue_roofline below)
        "roofline_issue": issue_roofline(issue_counts, main_ms),
        "box_state": {"fp64_vector_tflops_now": fp64_now, "peak": FP64_VECTOR_PEAK_TFLOPS,
                      "note": "dmi_fp64_probe right after the timed steps: independent v_fma_f64 chains on every SIMD for 20 ms"},
    }
    if ablation:
        out["ablation"] = ablation
    if scenes:
        out["scenes"] = scenes
    out["cell_to_point"] = cell_to_point
    if not args.no_end_to_end:
        pcie = capi.pcie_probe(local_rank)
        out["pcie_GBps"] = {"h2d": pcie[0], "d2h": pcie[1]}
        out["end_to_end"] = [end_to_end_probe(scene, capi, grid, ray, views, "f32", "f32", pcie),
                             end_to_end_probe(scene, capi, grid, ray, views, "f64", "f64", pcie),
                             end_to_end_probe(scene, capi, grid, ray, views, "f64", "f32", pcie)]
    if not args.no_coloration:
        out["coloration"] = coloration_probe(scene, capi, args.coloration_vertices, W, H, pcie=capi.pcie_probe(local_rank))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(grid, ray, views, args.cpu_seconds)
    ctx.close()
    emit(out)


def issue_roofline(counts, kernel_ms):
    """Instruction-issue floors of the fusion launch, from the committed PMC passes of the same workload.  Vector: the
    quad-cycles the vector pipes spent executing (SQ_ACTIVE_INST_VALU: a quarter-rate v_rcp_f64 counts four times; without
    that counter, one quad-cycle per vector instruction) spread over 1024 SIMDs at 2.4 GHz.  Scalar: one SALU instruction
    per cycle per CU (four SIMDs share the scalar unit); branches and scalar loads have issue ports of their own and are
    reported, not added.  `frac` = the larger floor / the kernel's time of THIS run."""
    if not counts:
        return None
    vector = counts["valu_insts"]
    busy = counts.get("valu_active_quad_cycles") or vector
    salu = counts.get("salu_insts") or 0
    ta = counts.get("ta_busy_cycles")
    vector_ms = busy * 4 / (1024 * 2.4e9) * 1e3
    scalar_ms = salu / (256 * 2.4e9) * 1e3
    ta_ms = ta / (256 * 2.4e9) * 1e3 if ta else None   # one texture addresser per CU: its busy cycles, summed over the chip
    floors = {"vector_issue": vector_ms, "scalar_issue": scalar_ms}
    if ta_ms is not None:
        floors["texture_addresser"] = ta_ms
    bound = max(floors, key=floors.get)
    return {"bound": bound,
            "vector_wave_instructions": vector, "vector_busy_quad_cycles": busy, "salu_wave_instructions": salu,
            "branch_wave_instructions": counts.get("branch_insts"), "smem_wave_instructions": counts.get("smem_insts"),
            "gather_wave_instructions": counts.get("vmem_rd_insts"), "cross_lane_lookup_wave_instructions": counts.get("lds_insts"),
            "texture_addresser_busy_cycles": ta,
            "vector_floor_ms": vector_ms, "scalar_floor_ms": scalar_ms, "ta_floor_ms": ta_ms, "kernel_ms": kernel_ms,
            "frac": floors[bound] / kernel_ms,
            "source": counts.get("tag"),
            "note": "counters of the rocprofv3 passes recorded in profiles/pmc_traffic.json under the same key and tag as "
                    "roofline.traffic (an earlier run of this workload, NOT this run); kernel_ms is this run's"}


# ---- N > 1 -------------------------------------------------------------------------------------------------------
def multi_gpu(args, capi, scene, dist, barrier, device_sync, max_over_ranks, world, rank, local_rank, n_ranks):
    """Every measurement of the N-GPU run; returns rank 0's JSON object.  `world` processes drive `n_ranks` GPUs: one
    each (the usual launch), or one process all of them (--one-process)."""
    cells, maps_per_gpu, W, H = parse_workload(args.workload)
    grid = scene.default_grid(cells)
    ray = scene.default_ray_potential(grid)
    n_vox = grid.n_voxels
    np_grid = np.float32 if args.grid_dtype == "f32" else np.float64
    def create(g):
        kw = dict(grid_dtype=args.grid_dtype, depth_storage="auto", kernel_variant=args.variant, partition=args.partition,
                  exchange=args.exchange, n_slabs=args.slabs)
        if args.one_process:
            devs = [0] * n_ranks if (args.share_device and args.exchange == "peer_copy") else list(range(n_ranks))
            return capi.MultiContext(g, ray, devices=devs, **kw)
        unique_id = None
        if args.partition == "views":  # every communicator needs an id of its own: rank 0 makes it, the launcher's store carries it
            box = [capi.multi_unique_id() if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(box, src=0)
            unique_id = box[0]
        return capi.MultiContext(g, ray, rank=rank, world=world, unique_id=unique_id, device=local_rank, **kw)

    notes = []
    requested_partition = args.partition

    def create_agreed(g):
        """create(g) on every rank, or -- when the communicator of the views partition cannot be set up on some rank (RCCL
        missing or refusing the topology) -- the z-slab partition, which needs none, on all of them.  The switch is reported
        (`config.fallback`, and `config.parallelism` names what ran), never silent; a failure of the z-slab partition itself, or
        with --one-process, is raised."""
        m, err = None, None
        try:
            m = create(g)
        except Exception as e:  # noqa: BLE001 - reported below
            err = repr(e)
        if max_over_ranks(1.0 if err else 0.0) == 0.0:
            return m
        if m is not None:
            m.close()
        if args.partition != "views" or args.one_process:
            raise RuntimeError(err or "another rank could not create its context")
        notes.append(f"views partition unavailable ({err or 'on another rank'}): z-slab partition instead")
        args.partition = "z_slabs"
        return create(g)

    my_ranks = list(range(n_ranks)) if args.one_process else [rank]

    g_spacing = float(max(grid.spacing))

    def scene_chunk(n_total, w, h, seed, c0, c1, sigma):
        """(views, threshold) of views c0 .. c1-1 of the n_total-camera scene of kind args.scene, as dmi_multi_add_views takes
        them: f32 depths, or f64 depths with best-cost values and the threshold (RD.cxx:138-167)."""
        v, thr = scene.make_scene_views(args.scene, n_total, w, h, seed=seed, view_range=(c0, c1), noise_sigma=sigma)
        if thr is None:
            v = scene.Views(v.depth.astype(np.float32), v.K4, v.RT4)
        return v, thr

    def upload(m, n_total, w, h, seed, shard_views: bool):
        """Views of an n_total-camera scene onto this process's ranks: each rank its share (views partition of a fixed
        problem), or every rank everything (z-slabs)."""
        for li, r in enumerate(my_ranks):
            lo, hi = capi.multi_view_shard(n_total, r, n_ranks) if shard_views else (0, n_total)
            for c0 in range(lo, hi, 32):   # bounded host memory: 32 views at a time
                c1 = min(hi, c0 + 32)
                m.add_views(*scene_chunk(n_total, w, h, seed, c0, c1, g_spacing), local_index=li)

    def timed(m, steps, warmup):
        for _ in range(warmup):
            m.fuse()
        m.synchronize()
        device_sync()
        barrier()
        k0 = m.local_timings(0)
        t0 = time.perf_counter()
        for _ in range(steps):
            m.fuse()
        m.synchronize()
        device_sync()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        t = m.timings()
        k1 = m.local_timings(0)
        timed.main_ms = (k1.total_fuse_main_kernel_ms - k0.total_fuse_main_kernel_ms) / max(1, steps)
        return dt, t.last_step_ms, t.last_fuse_kernel_ms

    def measure(name, cells_, n_total, w, h, seed, steps, warmup, check: bool):
        g = scene.default_grid(cells_)
        m = create_agreed(g)
        upload(m, n_total, w, h, seed, shard_views=(args.partition == "views"))
        dt, step_ms, fuse_ms = timed(m, steps, warmup)
        info = m.info()
        li = m.local_info(0)
        views_rank0 = int(li.n_views)
        depth_b = 8 if li.depth_storage_in_use == capi.DMI_DEPTH_F64 else 4
        b_alg = algorithmic_bytes(int(li.n_voxels), views_rank0, w, h, 4 if args.grid_dtype == "f32" else 8, depth_b)
        main_ms = timed.main_ms
        rec = {"workload": name, "maps_total": n_total, "ms_per_step": dt / steps * 1e3,
               "value": g.n_voxels * n_total * steps / dt / 1e9, "rank0_step_ms": step_ms, "rank0_fuse_kernel_ms": fuse_ms,
               "rank0_exchange_exposed_ms": max(0.0, step_ms - fuse_ms), "rccl_ranks": int(info.rccl_ranks),
               "views_on_this_process": int(info.n_views_local),
               "roofline": {"bound": "hbm", "achieved": b_alg / (main_ms * 1e-3) / 1e9 if main_ms > 0 else None,
                            "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": b_alg / (main_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if main_ms > 0 else None, "traffic": None,
                            "kernel": "dmi::fuse_tile_kernel" if li.tiled_kernel else "dmi::fuse_kernel", "kernel_ms": main_ms,
                            "algorithmic_bytes_per_launch": b_alg,
                            "note": "rank 0's fusion kernel (all slabs of one step) over the algorithmic bytes of its own "
                                    "views and grid; the path is bound by instruction issue and the texture addresser, not HBM (DESIGN.md 9)"}}
        if check and rank == 0 and args.partition == "views" and args.exchange in ("all_reduce", "peer_copy"):
            # the N-rank grid against this GPU's own fusion of ALL views (same views whatever N is)
            got, _ = m.download_grid(np_grid)
            got = got.copy()
            with capi.FusionContext(g, ray, device=local_rank, grid_dtype=args.grid_dtype) as one:
                for c0 in range(0, n_total, 32):
                    c1 = min(n_total, c0 + 32)
                    one.add_views(*scene_chunk(n_total, w, h, seed, c0, c1, float(max(g.spacing))))
                one.fuse()
                want = one.download_grid(np_grid)
            diff = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))))
            # |delta| <= 2 G 2^-24 sum|partials| per voxel (sharding.sharded_tolerance); sum|partials| <= views x rho
            tol = 2 * n_ranks * 2.0 ** -24 * n_total * abs(ray.rho)
            rec["check_vs_single_gpu"] = {"max_abs_diff": diff, "tolerance": tol, "within_tolerance": bool(diff <= tol),
                                          "max_abs_value": float(np.max(np.abs(want)))}
        m.close()
        return rec, info

    # weak: every rank `maps_per_gpu` views of its own = the shares of one (N x maps)-camera scene
    weak, info = measure(f"{args.workload} x {n_ranks} (weak: {maps_per_gpu} views per GPU)", cells, maps_per_gpu * n_ranks, W, H,
                         1000, args.steps, args.warmup, check=False)
    strong = []
    if not args.no_strong:
        # the fixed problems: the same `maps_per_gpu` views as the 1-GPU run, and BASELINE.json's cfg4
        rec, _ = measure(f"{args.workload} (strong: {maps_per_gpu} views in all)", cells, maps_per_gpu, W, H, 1000,
                         args.steps, args.warmup, check=True)
        strong.append(rec)
        if args.workload == "cfg3":
            c4, n4, w4, h4 = WORKLOADS["cfg4"]
            rec, _ = measure("cfg4 (strong: 512^3 x 1024 views of 640x480 in all)", c4, n4, w4, h4, 1004, args.steps,
                             args.warmup, check=False)
            strong.append(rec)
    # The first fixed problem once more with the z-slab partition -- every rank fuses ALL views into its own cell layers: no
    # exchange, the grid bit-identical to one GPU's; what the filter and the command-line tool do by default -- next to the
    # depth-map shards the metric's configuration names.  Reported, never the headline.
    strong_z = None
    if not args.no_strong and requested_partition == "views" and not args.one_process:
        saved = args.partition
        args.partition = "z_slabs"
        try:
            strong_z, _ = measure(f"{args.workload} (strong, z-slab partition: {maps_per_gpu} views in all, no exchange)", cells,
                                  maps_per_gpu, W, H, 1000, args.steps, args.warmup, check=False)
        finally:
            args.partition = saved
    top = weak if (args.scaling == "weak" or not strong) else strong[0]
    exchange_txt = {"all_reduce": f"one RCCL all-reduce of the {args.grid_dtype} grid in {info.n_slabs} z-slabs overlapped with the fusion",
                    "reduce_scatter": f"RCCL reduce-scatter of the grid (each rank keeps 1/{n_ranks})",
                    "peer_copy": f"all-reduce of the {args.grid_dtype} grid by peer-to-peer copies (SDMA) in {info.n_slabs} z-slabs, "
                                 "summed in rank order by a kernel behind the fusion, no RCCL"}[args.exchange]
    out = {
        "metric": "Gvoxel-projections/s",
        "value": top["value"],
        "unit": "Gvoxel-projections/s",
        "n_gpus": n_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": top["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak" if top is weak else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{cells[0]}x{cells[1]}x{cells[2]} voxels, depth maps {W}x{H}, {top['maps_total']} in all "
                        f"({args.workload}, {args.scene} sphere scene)",
            "grid_dtype": args.grid_dtype,
            "kernel_variant": args.variant,
            "maps_total": top["maps_total"],
            "parallelism": (f"depth-map shards x{n_ranks}, {exchange_txt}" if args.partition == "views"
                            else f"z-slabs x{n_ranks}: every rank fuses all views into its own cell layers, no collective"),
            "processes": world,
            "rccl_ranks": int(info.rccl_ranks),
            "rccl_version": int(info.rccl_version),
            "launched_by": os.environ.get("DMI_BENCH_LAUNCHED_BY", "torch.distributed.run" if world > 1 else "bench.py --one-process"),
            "ranks_share_device_0": bool(args.share_device),
        },
        "roofline": top["roofline"],
        "weak": weak,
        "strong": strong,
    }
    if strong_z is not None:
        out["strong_z_slabs"] = strong_z
    if notes:
        out["config"]["fallback"] = notes[0]
    return out


if __name__ == "__main__":
    main()
