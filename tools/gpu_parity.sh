#!/bin/bash
# parity subset after a kernel edit: tests/test_gpu_parity.py, the fuzz test and the full-size configs
set -u
TAG=${1:-par}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/${TAG}_pytest.log
tail -6 gpurun_out/${TAG}_pytest.log
exit $rc
