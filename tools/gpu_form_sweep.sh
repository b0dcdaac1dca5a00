#!/bin/bash
# The two forms of the tiled kernel's one-wave launches (persistent workgroups / one workgroup per brick) at several sizes and
# scenes, with the four work levels and with the cost order: variants of ONE library, rounds interleaved in one process per workload.
# Usage: tools/gpu_form_sweep.sh <tag> [scenes]
set -u
TAG=${1:-forms}; SCENES=${2:-dense,speckle}
mkdir -p gpurun_out
NEVER=65536; ALWAYS=32768; COST=1048576; NOCOST=2097152
for W in 128x64@640x480 256x32@640x480 256x64@640x480 256x128@640x480 384x64@640x480 384x128@640x480 512x32@1280x720 512x64@1280x720 512x96@1280x720 512x128@640x480 512x256@1280x720 1024x64@1920x1080; do
  echo "== $W"
  r=5; case $W in 512x256*|1024*) r=3;; esac
  timeout -k 10 400 python tools/gpu_exp.py tools/exp_list_head.txt --workload $W --rounds $r --variants $((NEVER+NOCOST)),$((ALWAYS+NOCOST)),$((NEVER+COST)),$((ALWAYS+COST)) --scenes $SCENES --tag ${TAG}_${W%%@*} > gpurun_out/${TAG}_${W%%@*}.log 2>&1 || { tail -3 gpurun_out/${TAG}_${W%%@*}.log; exit 1; }
  grep " fuse " gpurun_out/${TAG}_${W%%@*}.log
done
