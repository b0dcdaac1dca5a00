#!/bin/bash
# Parity of the changed kernels, then an in-process A/B of the library in the tree against the one saved as build/ref/libdmi_hip_<prev>.so
# (tools/exp_list_prev_vs_head.txt).  usage: tools/gpu_prev_ab.sh <tag> [scenes] [workload] [pytest files...]
set -u
TAG=${1:-r21b}; SCENES=${2:-speckle,dense}; WL=${3:-cfg3}; shift; shift; shift
TESTS=${@:-tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_geo.py}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest $TESTS -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/gpu_exp.py tools/exp_list_prev_vs_head.txt --workload $WL --rounds 7 --scenes $SCENES --tag $TAG > gpurun_out/${TAG}_exp.log 2>&1; echo "exp rc=$?"
grep -v "^\[" gpurun_out/${TAG}_exp.log | tail -8
