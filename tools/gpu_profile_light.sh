#!/bin/bash
# kernel trace + two SQ counter passes over the headline bench (a quick look between kernel edits; tools/gpu_profile.sh is the
# full set).  Usage: tools/gpu_profile_light.sh <tag> [bench args]
set -u
TAG=${1:-prof}; shift || true
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/$TAG
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes $* > $O.kt.log 2>&1; echo "kt rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- $P > $O.sq1.log 2>&1; echo "pmc1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq3 -- $P > $O.sq3.log 2>&1; echo "pmc6 rc=$?"
