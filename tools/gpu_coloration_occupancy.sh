#!/bin/bash
# What keeping a vertex's colour values in LDS would cost the projection pass: the same kernels with extra dynamic LDS per
# workgroup of 256 vertices (tuning library, DMI_DEBUG_COLOR_EXTRA_LDS), i.e. with fewer waves resident per CU -- 24 KB of histograms
# today (six workgroups = 24 waves per CU); + 64 KB = 64 views x 256 vertices x 4 B (one workgroup per CU = 4 waves) ...
# usage: tools/gpu_coloration_occupancy.sh <tag>
set -u
TAG=${1:-r21z}
mkdir -p gpurun_out
export TMPDIR=/tmp DMI_TUNING=1
OUT=gpurun_out/${TAG}_coloration_occupancy.txt; : > $OUT
for extra in 0 8192 16384 29000 40000 56000 65536; do
  export DMI_DEBUG_COLOR_EXTRA_LDS=$extra
  echo "== extra LDS per workgroup $extra B (24576 + $extra: $(( 163840 / (24576 + extra) )) workgroups = $(( 4 * (163840 / (24576 + extra)) )) waves per CU)" >> $OUT
  timeout -k 10 200 python tools/gpu_coloration_call.py 2>&1 | grep -v "^pcie" | cut -d' ' -f1,7- >> $OUT || exit 1
done
cat $OUT
