"""Multi-GPU partitioning of one fusion: a thin Python face of the dmi_multi_* C ABI (include/dmi.h, csrc/dmi_multi.hip).

The fusion is a sum over depth maps of independent per-voxel terms (CudaReconstruction.cu:211), so it shards two
ways:

  * depth-map shards (the north-star contract): rank r fuses its contiguous share of the views into its own full grid,
    then ONE all-reduce(sum) of the f32 TSDF grid over xGMI -- inside the library: RCCL, overlapped slab by slab with
    the fusion (capi.MultiContext(..., partition="views")).  The summation order changes: |delta| <=
    2G * 2^-24 * sum|partials| per voxel (sharded_tolerance).
  * z-slabs (no collective): rank r owns cell layers [z0, z1) of the grid, fuses ALL views into them and hands its slab
    to the host: bit-identical to a single-GPU fusion (partition="z_slabs").

The partition arithmetic below IS the library's (dmi_multi_view_shard / dmi_multi_z_slab / dmi_multi_slab_ranges; no GPU
needed), so tests that rehearse N ranks on the CPU partition exactly as the GPUs will.
"""
from __future__ import annotations

from . import capi


def view_shard(n_views: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced range [lo, hi) of views for `rank` (first n_views % world ranks get one more)."""
    return capi.multi_view_shard(n_views, rank, world)


def z_slab(nz: int, rank: int, world: int) -> tuple[int, int]:
    """Cell layers [z0, z1) owned by `rank`; boundaries are multiples of 16 cells except at the top of the grid."""
    return capi.multi_z_slab(nz, rank, world)


def vertex_shard(n_vertices: int, rank: int, world: int) -> tuple[int, int]:
    """MeshColoration on several GPUs (BASELINE config 5): every rank holds all views (dmi_color_context) and colours
    the vertices [lo, hi); the three output arrays are concatenated in rank order.  No exchange step: a vertex's mean,
    median and count depend on that vertex alone (Coloration/MeshColoration.cxx:140-192)."""
    return view_shard(n_vertices, rank, world)


def slab_ranges(nz: int, n_slabs: int) -> list[tuple[int, int]]:
    """The (z_first, z_count) slabs of the overlapped all-reduce: inner boundaries on multiples of 32 cells, the last
    slab about half as thick as the others (its exchange is the one piece no fusion hides)."""
    return capi.multi_slab_ranges(nz, n_slabs)


def peer_chunk(n: int, world: int, c: int) -> tuple[int, int]:
    """DMI_EXCHANGE_PEER_COPY: the piece (first, count) of an n-element slab that rank c sums, in rank order, and hands back."""
    return capi.multi_peer_chunk(n, world, c)


def sharded_tolerance(world: int, abs_partial_sum):
    """Bound on |all-reduced f32 grid - single-GPU f64 grid| per voxel: each rank rounds its partial to f32
    (2^-24 relative), the reduction adds world-1 f32 roundings of partial sums, fp64 reordering is
    below that."""
    return (2 * world) * 2.0 ** -24 * abs_partial_sum + 1e-30
