#!/bin/bash
# Development A/B in one GPU call: timing of the listed libraries on cfg3 (speckle, dense) + instruction counters of the last one.
# usage: tools/gpu_fast_ab.sh <tag> <list> <lib-for-pmc>
set -u
TAG=$1; LIST=$2; LIB=$3
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 420 python tools/gpu_exp.py $LIST --workload cfg3 --rounds 4 --scenes ${SCENES:-speckle,dense} --tag $TAG > gpurun_out/${TAG}_exp.log 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_exp.log | tail -8
for SCENE in ${PMC_SCENES:-speckle dense}; do
  export DMI_LIB_OVERRIDE=$LIB
  O=gpurun_out/${TAG}_pmc_$SCENE
  P="python3 bench.py --scene $SCENE --steps 2 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes"
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -- $P > $O.log 2>&1
  python3 - $O $SCENE <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fuse_tile_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(sys.argv[2], {k: "%.3e" % (v / max(1, n[k])) for k, v in sorted(tot.items())})
PY
done
