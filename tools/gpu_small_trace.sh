#!/bin/bash
# Kernel timeline of one fusion at the small configs (rocprofv3 --kernel-trace): where a step's time goes between launches.
# Usage: tools/gpu_small_trace.sh <tag>
set -u
TAG=${1:-small}; mkdir -p gpurun_out; export TMPDIR=/tmp
for W in cfg1:dense cfg2:dense cfg2:speckle; do
  wl=${W%%:*}; sc=${W##*:}
  O=gpurun_out/${TAG}_${wl}_${sc}
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --workload $wl --scene $sc --steps 6 --warmup 2 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes > $O.log 2>&1; echo "$W rc=$?"
  python3 tools/step_timeline.py $O > $O.timeline.txt 2>&1
  tail -40 $O.timeline.txt
done
