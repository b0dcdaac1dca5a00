#!/bin/bash
# The small configs after a change of the preparation launches: parity first, then the kernel timelines of one fusion
# (tools/gpu_small_trace.sh) and the workgroup timelines of the fusion kernel at cfg 2 (tuning build).  Usage: tools/gpu_small_round.sh <tag>
set -u
TAG=${1:-small}; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/${TAG}_parity.log 2>&1
rc=$?; tail -3 gpurun_out/${TAG}_parity.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_small_trace.sh $TAG || exit 1
T=cudadepthmapintegration_amd/csrc/libdmi_hip_tuning.so
if [ -f $T ]; then
  for sc in dense speckle; do
    DMI_DEBUG_WG_TIMES=1 DMI_LIB_OVERRIDE=$T timeout -k 10 300 python tools/gpu_wg_timeline.py --workload cfg2 --scene $sc --tag ${TAG}_wg_cfg2_$sc > gpurun_out/${TAG}_wg_cfg2_$sc.log 2>&1 || { tail -5 gpurun_out/${TAG}_wg_cfg2_$sc.log; exit 1; }
  done
fi
echo done
