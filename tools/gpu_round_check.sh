#!/bin/bash
# One GPU call: the whole -m gpu suite, then the bench lines that matter (default run, the N > 1 code path rehearsed with
# one rank, the launcher form).  Logs under gpurun_out/<tag>_*.  Usage: tools/gpu_round_check.sh <tag>
set -u
TAG=${1:-check}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/${TAG}_bench.json; echo
timeout -k 10 600 python bench.py --force-multi --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_multi1.json 2> gpurun_out/${TAG}_bench_multi1.err; echo "bench multi rc=$?"
tail -c 1500 gpurun_out/${TAG}_bench_multi1.json; echo
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end > gpurun_out/${TAG}_bench_torchrun1.json 2> gpurun_out/${TAG}_bench_torchrun1.err; echo "bench torchrun rc=$?"
tail -c 300 gpurun_out/${TAG}_bench_torchrun1.json; echo
echo "cpu share:"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(len(os.sched_getaffinity(0)))"; python -c "from oracle import oracle; print('oracle threads', oracle.max_threads())"
