#!/bin/bash
# The two forms of the tiled kernel's one-wave launches (persistent workgroups / one workgroup per brick) at several sizes:
# variants 32768 (always persistent) and 65536 (never) of ONE library, rounds interleaved in one process per workload.
# Usage: tools/gpu_form_sweep.sh <tag>
set -u
TAG=${1:-forms}
mkdir -p gpurun_out
printf 'head:@%s/cudadepthmapintegration_amd/csrc/libdmi_hip.so\n' "$PWD" > gpurun_out/form_list.txt
for W in 256x64@640x480 256x128@640x480 384x64@640x480 384x128@640x480 512x32@1280x720 512x64@1280x720 512x96@1280x720 512x128@640x480 768x64@1280x720 1024x32@1920x1080 1024x64@1920x1080; do
  echo "== $W"
  timeout -k 10 300 python tools/gpu_exp.py gpurun_out/form_list.txt --workload $W --rounds 5 --variants 65536,32768 --scenes dense --tag ${TAG}_$W 2>/dev/null | grep "^dense"
done
