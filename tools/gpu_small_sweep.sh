#!/bin/bash
# Small fusions with and without brick classes (variants 131072 = classes always / 256 = never, of ONE library, rounds interleaved): where does the chain of
# preparation launches cost more than the classes save?  Usage: tools/gpu_small_sweep.sh <tag>
set -u
TAG=${1:-small}
mkdir -p gpurun_out
printf 'head:@%s/cudadepthmapintegration_amd/csrc/libdmi_hip.so\n' "$PWD" > gpurun_out/small_list.txt
for W in 64x4@320x240 64x16@320x240 64x64@320x240 96x8@320x240 128x4@640x480 128x8@640x480 128x16@640x480 128x32@640x480 192x4@640x480 192x8@640x480 256x4@640x480 256x8@640x480; do
  for S in dense speckle; do
  echo "== $W $S"
  timeout -k 10 200 python tools/gpu_exp.py gpurun_out/small_list.txt --workload $W --rounds 5 --variants 131072,256 --scenes $S --steps 16 --tag ${TAG}_${W}_$S 2>/dev/null | grep "^$S"
  done
done
