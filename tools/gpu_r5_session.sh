#!/bin/bash
# One GPU call of round 5: parity (+ the geo family), A/B against round 4, the pair-cost counters (tuning library), column heights.
set -u
TAG=${1:-r19c}
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_r5_ab.sh $TAG speckle,dense tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_geo.py || exit 1
bash tools/gpu_pair_cost.sh ${TAG}_pairs 2>&1 | tail -8
timeout -k 10 400 python tools/gpu_exp.py tools/exp_list_head.txt --workload cfg2 --rounds 7 --variants 0,224,4096 --scenes speckle,dense --tag ${TAG}_cfg2 > gpurun_out/${TAG}_cfg2.log 2>&1; grep -v "^\[" gpurun_out/${TAG}_cfg2.log | tail -7
timeout -k 10 400 python tools/gpu_exp.py tools/exp_list_head.txt --workload cfg3 --rounds 5 --variants 0,224 --scenes speckle --tag ${TAG}_cfg3tk > gpurun_out/${TAG}_cfg3tk.log 2>&1; grep -v "^\[" gpurun_out/${TAG}_cfg3tk.log | tail -3
