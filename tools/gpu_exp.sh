#!/bin/bash
# Times every experiment library of the list on cfg3 (dense and sparse), one process each, in one GPU call.
set -u
LIST=${1:-tools/exp_list.txt}; TAG=${2:-exp}
mkdir -p gpurun_out
export TMPDIR=/tmp
while IFS= read -r line; do
  [ -z "$line" ] && continue
  name=${line%%:*}; defs=${line#*:}
  unset DMI_LIB_OVERRIDE
  case "$defs" in @*) export DMI_LIB_OVERRIDE=${defs#@}; line="";; esac
  DMI_EXP="$line" timeout -k 10 300 python tools/gpu_sweep.py --workload cfg3 --variants ${VARIANTS:-0} --rounds 5 --tag ${TAG}_$name 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$name', d['scene'], d['variant'], 'fuse', round(d['median_ms'], 3), 'main', round(d['main_median_ms'], 3))
"
done < "$LIST"
