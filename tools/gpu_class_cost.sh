#!/bin/bash
# Where the fusion kernel's time goes (tuning build, results wrong on purpose): class bytes rewritten after the
# classification.  table byte c = what class c becomes (0 mixed, 1 free, 2 behind, 3 skip).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp DMI_TUNING=1
TAG=${1:-cc}
run() { echo "== $1"; DMI_DEBUG_CLASS_REMAP=$2 timeout -k 10 300 python tools/gpu_sweep.py --workload cfg3 --variants ${3:-0} --rounds 5 --scenes dense --tag ${TAG}_$1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['variant'], 'fuse', round(d['median_ms'], 3), 'main', round(d['main_median_ms'], 3), d['brick_classes'])
"; }
run identity 0x03020100
run free_to_skip 0x03020300
run mixed_to_skip 0x03020103
run mixed_to_free 0x03020101
run only_mixed 0x03030300
run all_skip 0x03030303
