"""Multi-GPU partitioning of one fusion (one process per GPU, torch.distributed; backend "nccl" is RCCL).

The fusion is a sum over depth maps of independent per-voxel terms (CudaReconstruction.cu:211), so it
shards two ways:

  * depth-map shards (the north-star contract): rank r fuses views [lo, hi) into its own full grid, then
    ONE all-reduce(sum) of the f32 TSDF grid over xGMI.  The summation order changes: |delta| <=
    (G-1) * 2^-24 * sum|partials| per voxel; hit counters all-reduce as integers and stay exact.
  * z-slabs (no collective): rank r owns cell layers [z0, z1) of the grid (dmi_options.z_first), fuses
    ALL views into them and hands its slab to the host: bit-identical to a single-GPU fusion.

Only the partition arithmetic and the collective live here; the fusion itself is the C ABI (capi.py).
"""
from __future__ import annotations


def view_shard(n_views: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced range [lo, hi) of views for `rank` (first n_views % world ranks get one more)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank/world out of range")
    base, extra = divmod(int(n_views), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def z_slab(nz: int, rank: int, world: int, multiple: int = 1) -> tuple[int, int]:
    """Cell layers [z0, z1) owned by `rank`; slab boundaries fall on multiples of `multiple` (the tiled
    kernel's column height) except at the top of the grid."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank/world out of range")
    units = -(-int(nz) // multiple)
    lo, hi = view_shard(units, rank, world)
    return min(lo * multiple, nz), min(hi * multiple, nz)


def vertex_shard(n_vertices: int, rank: int, world: int) -> tuple[int, int]:
    """MeshColoration on several GPUs (BASELINE config 5): every rank holds all views (dmi_color_context) and colours
    the vertices [lo, hi); the three output arrays are concatenated in rank order.  No exchange step: a vertex's mean,
    median and count depend on that vertex alone (Coloration/MeshColoration.cxx:140-192)."""
    return view_shard(n_vertices, rank, world)


def all_reduce_grid(grid_tensor, group=None):
    """The path's single exchange step: sum the per-rank TSDF grids in place (RCCL ring/direct over xGMI
    on GPUs, gloo in the CPU tests).  Also used for the integer hit counters."""
    import torch.distributed as dist

    dist.all_reduce(grid_tensor, op=dist.ReduceOp.SUM, group=group)
    return grid_tensor


def reduce_scatter_grid(grid_tensor, rank: int, world: int, group=None):
    """The cheaper exchange when only the host consumes the grid: every rank ends up with the sum of
    its own 1/world slice (contiguous in z) and downloads just that -- half the traffic of the all-reduce.  Returns
    (slice tensor, first element, element count).  The element count must divide evenly (pad the grid otherwise).
    RCCL only (gloo has no reduce-scatter); bench.py --exchange reduce_scatter."""
    import torch
    import torch.distributed as dist

    n = grid_tensor.numel()
    if n % world != 0:
        raise ValueError("reduce_scatter_grid: the grid size must be a multiple of the world size")
    per = n // world
    out = torch.empty(per, dtype=grid_tensor.dtype, device=grid_tensor.device)
    dist.reduce_scatter_tensor(out, grid_tensor, op=dist.ReduceOp.SUM, group=group)
    return out, rank * per, per


def sharded_tolerance(world: int, abs_partial_sum):
    """Bound on |all-reduced f32 grid - single-GPU f64 grid| per voxel: each rank rounds its partial to f32
    (2^-24 relative), the reduction adds world-1 f32 roundings of partial sums, fp64 reordering is
    below that."""
    return (2 * world) * 2.0 ** -24 * abs_partial_sum + 1e-30


def slab_ranges(nz: int, n_slabs: int, align: int = 32, taper: bool = True) -> list[tuple[int, int]]:
    """nz cell layers split into at most n_slabs contiguous (z_first, z_count) ranges whose inner boundaries are
    multiples of `align` (dmi.h: DMI_SLAB_ALIGNMENT).  With `taper` the last slab is about half as thick as the
    others: its all-reduce is the one piece of the exchange that no fusion hides (fuse_and_all_reduce), so it should
    be the smallest message; the earlier slabs take up the difference."""
    units = -(-int(nz) // align)
    n = max(1, min(int(n_slabs), units))
    sizes = []
    if taper and n >= 2 and units >= 2 * n:
        last = max(1, units // (2 * n))
        rest, extra = divmod(units - last, n - 1)
        sizes = [rest + (1 if s < extra else 0) for s in range(n - 1)] + [last]
    else:
        for s in range(n):
            lo, hi = view_shard(units, s, n)
            sizes.append(hi - lo)
    out, lo = [], 0
    for sz in sizes:
        z0, z1 = min(lo * align, nz), min((lo + sz) * align, nz)
        if z1 > z0:
            out.append((z0, z1 - z0))
        lo += sz
    return out


def fuse_and_all_reduce(ctx, grid_t, cell_dims, n_slabs, fuse_stream, comm_stream, group=None):
    """One fusion step of the depth-map-sharded multi-GPU path with the exchange hidden behind the compute:
    the grid is fused slab by slab (dmi_fuse_slab, on `fuse_stream`, the stream the context was created with) and
    the all-reduce of slab i runs on `comm_stream` while slab i+1 is being fused.  Same result as ctx.fuse()
    followed by one all-reduce of the whole grid.  grid_t is the context's external grid ([nz*ny*nx] tensor)."""
    import torch

    nx, ny, nz = (int(c) for c in cell_dims)
    plane = nx * ny
    for z0, zc in slab_ranges(nz, n_slabs):
        ctx.fuse_slab(z0, zc)
        done = torch.cuda.Event()
        done.record(fuse_stream)
        comm_stream.wait_event(done)
        with torch.cuda.stream(comm_stream):
            all_reduce_grid(grid_t[z0 * plane:(z0 + zc) * plane], group=group)
    fuse_stream.wait_stream(comm_stream)  # whoever touches the grid next on fuse_stream sees the reduced values
