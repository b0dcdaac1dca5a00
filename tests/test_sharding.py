"""Multi-process CPU tests (gloo, world_size 2) of the N > 1 path: view shards + one all-reduce of the f32
grid, and z-slab ownership.  The partition arithmetic is the library's own (dmi_multi_view_shard / dmi_multi_z_slab /
dmi_multi_slab_ranges through cudadepthmapintegration_amd.sharding: host code, no GPU needed); the per-rank fusion
is done by the oracle here (there is no GPU in this container; the oracle is the checker's arithmetic) and gloo
stands in for the RCCL all-reduce that dmi_multi_fuse issues, so what is under test is the partition, the shape of
the exchange and the stated tolerance; tests/test_gpu_parity.py drives dmi_multi_* itself on one GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from cudadepthmapintegration_amd import scene, sharding


def test_view_shard_is_a_balanced_partition():
    for n in (0, 1, 7, 256, 1024, 1025):
        for world in (1, 2, 3, 8):
            r = [sharding.view_shard(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_z_slab_respects_column_height():
    for nz in (1, 17, 512, 1000, 1024):
        for world in (1, 2, 8):
            r = [sharding.z_slab(nz, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == nz
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert all(lo % 16 == 0 for lo, hi in r if lo < nz)
    assert [sharding.z_slab(512, k, 8) for k in range(8)] == [(64 * k, 64 * k + 64) for k in range(8)]


def test_slab_ranges_cover_the_grid_on_aligned_boundaries():
    for nz in (1, 31, 32, 100, 512, 1000):
        for n in (1, 2, 4, 7, 64):
            r = sharding.slab_ranges(nz, n)
            assert r[0][0] == 0 and r[-1][0] + r[-1][1] == nz and len(r) <= n
            assert all(a[0] + a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(z0 % 32 == 0 for z0, _ in r)
    # the last slab is the thin one: its all-reduce is what no fusion hides
    assert sharding.slab_ranges(512, 4) == [(0, 160), (160, 160), (320, 128), (448, 64)]


def test_partition_functions_reject_bad_ranks():
    for fn in (lambda: sharding.view_shard(10, 2, 2), lambda: sharding.view_shard(10, -1, 2), lambda: sharding.z_slab(10, 0, 0)):
        with pytest.raises(ValueError):
            fn()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle
    from helpers import oracle_params_from_scene

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grid = scene.default_grid((28, 24, 72))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(9, 64, 48, seed=5, dense=True)   # 9 views over 2 ranks: 5 + 4
    p = oracle_params_from_scene(grid, rp, views)
    lo, hi = sharding.view_shard(views.n, rank, world)
    part, vh, mh = oracle.fuse(p, views.depth[lo:hi], views.K4[lo:hi], views.RT4[lo:hi])
    g32 = torch.from_numpy(part.astype(np.float32))            # each rank's grid is f32 on the device
    absum = torch.from_numpy(np.abs(part))
    hits = torch.from_numpy(vh.astype(np.int64))
    # the path's single exchange step: a sum all-reduce of the f32 grid (RCCL inside dmi_multi_fuse on GPUs, gloo here),
    # slab by slab exactly as the library cuts it
    plane = grid.cell_dims[0] * grid.cell_dims[1]
    for z0, zc in sharding.slab_ranges(grid.cell_dims[2], 2):
        dist.all_reduce(g32.view(-1)[z0 * plane:(z0 + zc) * plane], op=dist.ReduceOp.SUM)
    dist.all_reduce(absum, op=dist.ReduceOp.SUM)
    dist.all_reduce(hits, op=dist.ReduceOp.SUM)
    # the same exchange the third way (DMI_EXCHANGE_PEER_COPY): per slab, rank c collects every rank's chunk c, adds them in
    # rank order -- one f32 rounding each -- and hands the sum back to everybody; point-to-point messages only
    mine = torch.from_numpy(part.astype(np.float32)).view(-1)
    for z0, zc in sharding.slab_ranges(grid.cell_dims[2], 2):
        e0, n = z0 * plane, zc * plane
        first, count = sharding.peer_chunk(n, world, rank)
        own = mine[e0 + first:e0 + first + count]
        received = {rank: own.clone()}
        for other in range(world):                     # send the others their chunk of my grid, receive mine from them
            if other == rank:
                continue
            f2, c2 = sharding.peer_chunk(n, world, other)
            send = dist.isend(mine[e0 + f2:e0 + f2 + c2].clone(), dst=other)
            buf = torch.empty(count, dtype=torch.float32)
            dist.recv(buf, src=other)
            send.wait()
            received[other] = buf
        total = received[0].clone()
        for j in range(1, world):
            total += received[j]                       # rank order
        own.copy_(total)
        for owner in range(world):                     # the all-gather
            f2, c2 = sharding.peer_chunk(n, world, owner)
            dist.broadcast(mine[e0 + f2:e0 + f2 + c2], src=owner)
    np.save(os.path.join(out_dir, f"peer_{rank}.npy"), mine.numpy())
    # z-slab ownership: every rank fuses all views into its own layers
    z0, z1 = sharding.z_slab(grid.cell_dims[2], rank, world)
    if rank == 0:
        np.savez(os.path.join(out_dir, "reduced.npz"), grid=g32.numpy(), absum=absum.numpy(), hits=hits.numpy())
    np.save(os.path.join(out_dir, f"slab_{rank}.npy"), np.array([z0, z1]))
    dist.barrier()
    dist.destroy_process_group()


def test_view_shards_plus_all_reduce_match_single_fusion(tmp_path):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle
    from helpers import oracle_params_from_scene

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    grid = scene.default_grid((28, 24, 72))
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(9, 64, 48, seed=5, dense=True)
    want, vh, _ = oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth, views.K4, views.RT4)
    got = np.load(tmp_path / "reduced.npz")
    assert np.array_equal(got["hits"], vh.astype(np.int64))                  # integers: bit-exact
    tol = sharding.sharded_tolerance(world, got["absum"])
    assert np.all(np.abs(got["grid"].astype(np.float64) - want) <= tol)    # stated float tolerance
    assert np.abs(want).max() > 0.1
    # the peer-copy exchange: both ranks hold the same bits, exactly the rank-ordered f32 sum of the f32 partials
    peers = [np.load(tmp_path / f"peer_{r}.npy") for r in range(world)]
    parts = []
    for r in range(world):
        lo, hi = sharding.view_shard(views.n, r, world)
        parts.append(oracle.fuse(oracle_params_from_scene(grid, rp, views), views.depth[lo:hi], views.K4[lo:hi], views.RT4[lo:hi])[0]
                     .astype(np.float32).reshape(-1))
    expect = parts[0]
    for r in range(1, world):
        expect = (expect + parts[r]).astype(np.float32)
    assert all(np.array_equal(p.view(np.uint32), expect.view(np.uint32)) for p in peers)
    assert np.all(np.abs(expect.astype(np.float64) - want.reshape(-1)) <= tol.reshape(-1))
    slabs = [tuple(np.load(tmp_path / f"slab_{r}.npy")) for r in range(world)]
    assert slabs[0][0] == 0 and slabs[-1][1] == 72 and slabs[0][1] == slabs[1][0] and slabs[0][1] % 16 == 0


def _color_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    views = scene.make_views(6, 48, 36, seed=3)
    colors = scene.make_colors(6, 48, 36, seed=4)
    pts = scene.make_mesh_points(501, seed=5)
    lo, hi = sharding.vertex_shard(len(pts), rank, world)          # every rank: all views, its own vertices
    mean, median, count = oracle.color_mesh(pts[lo:hi], colors, views.K4, views.RT4)
    # the only "exchange" is the concatenation of the rank results (a gather to whoever writes the mesh)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, mean, median, count))
    if rank == 0:
        np.savez(os.path.join(out_dir, "colored.npz"), mean=np.concatenate([g[2] for g in gathered]),
                 median=np.concatenate([g[3] for g in gathered]), count=np.concatenate([g[4] for g in gathered]),
                 bounds=np.array([[g[0], g[1]] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_vertex_shards_of_the_coloration_pass_concatenate_to_the_whole(tmp_path):
    """BASELINE config 5's coloration pass on several GPUs: vertex shards, no collective on the data path."""
    import torch.multiprocessing as mp

    from oracle import oracle

    world = 2
    mp.spawn(_color_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    views = scene.make_views(6, 48, 36, seed=3)
    colors = scene.make_colors(6, 48, 36, seed=4)
    pts = scene.make_mesh_points(501, seed=5)
    want = oracle.color_mesh(pts, colors, views.K4, views.RT4)
    got = np.load(tmp_path / "colored.npz")
    assert got["bounds"].tolist() == [[0, 251], [251, 501]]
    for name, w in zip(("mean", "median", "count"), want):
        assert np.array_equal(got[name], w), name


def test_peer_chunks_tile_a_slab_and_short_grids_leave_ranks_empty():
    for n, world in ((1, 1), (255, 2), (256, 2), (257, 3), (40 * 33 * 36, 8), (1 << 20, 7), (100, 16)):
        pieces = [sharding.peer_chunk(n, world, c) for c in range(world)]
        assert pieces[0][0] == 0 and sum(c for _, c in pieces) == n
        for (f0, c0), (f1, _) in zip(pieces, pieces[1:]):
            assert f1 == f0 + c0 and (c0 % 256 == 0 or f0 + c0 == n)
    # z-slab partition of a short grid: ranks beyond nz / 16 own nothing (dmi_multi_fuse must cope: test_gpu_parity.py
    # test_z_slab_rank_without_a_cell_layer_keeps_its_step_clock)
    owned = [sharding.z_slab(40, r, 8) for r in range(8)]
    assert owned[:3] == [(0, 16), (16, 32), (32, 40)] and all(z0 == z1 for z0, z1 in owned[3:])
