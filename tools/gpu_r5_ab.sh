#!/bin/bash
# Round 5: parity of the changed kernels, then an in-process A/B against round 4's library (build/ref/libdmi_hip_r4final.so).
# usage: tools/gpu_r5_ab.sh <tag> [scenes] [pytest files...]
set -u
TAG=${1:-r19a}; SCENES=${2:-speckle,dense}; shift; shift
TESTS=${@:-tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest $TESTS -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/gpu_exp.py tools/exp_list_r5.txt --workload cfg3 --rounds 7 --scenes $SCENES --tag $TAG > gpurun_out/${TAG}_exp.log 2>&1; echo "exp rc=$?"
grep -v "^\[" gpurun_out/${TAG}_exp.log | tail -8
