"""The N > 1 forms of bench.py, rehearsed on ONE GPU so that the path the driver runs on a multi-GPU node does not rot while no
such node is at hand: two ranks sharing device 0 with the peer-copy exchange (one process), and the launcher form (two processes
under torch.distributed.run) where RCCL refuses two ranks on one device and every rank falls back to the z-slab partition.
The JSON line is what is checked -- n_gpus, the strong-scaling check against a single-GPU fusion, the exposed exchange time --
never a scaling figure: two ranks on one GPU say nothing about two GPUs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOAD = "128x12@320x240"   # 128^3 cells, 12 views per rank: seconds, not minutes


def _line(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_peer_copy():
    d = _line([sys.executable, "bench.py", "--gpus", "2", "--exchange", "peer_copy", "--share-device", "--workload", WORKLOAD,
               "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["metric"] == "Gvoxel-projections/s" and d["value"] > 0
    assert d["config"]["ranks_share_device_0"] and d["config"]["rccl_ranks"] == 0 and d["config"]["processes"] == 1
    assert "fallback" not in d["config"]
    assert d["weak"]["maps_total"] == 24 and d["weak"]["rank0_exchange_exposed_ms"] >= 0
    strong = d["strong"][0]
    assert strong["maps_total"] == 12 and strong["check_vs_single_gpu"]["within_tolerance"], strong
    assert strong["check_vs_single_gpu"]["max_abs_value"] > 0.1
    assert strong["rank0_exchange_exposed_ms"] >= 0 and strong["rank0_fuse_kernel_ms"] > 0


@pytest.mark.gpu
def test_bench_two_processes_fall_back_to_z_slabs_on_one_device():
    """torch.distributed.run with two ranks on a one-GPU box: --share-device puts both on device 0, the RCCL communicator of the
    views partition cannot be set up there, and every rank switches to the z-slab partition -- the line says so."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2", "--share-device", "--workload", WORKLOAD, "--steps", "2",
               "--warmup", "1"], env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["config"]["processes"] == 2 and d["value"] > 0
    assert "z-slab" in d["config"].get("fallback", ""), d["config"]
    assert d["config"]["parallelism"].startswith("z-slabs x2")
    assert d["weak"]["rccl_ranks"] == 0 and d["weak"]["rank0_exchange_exposed_ms"] >= 0
