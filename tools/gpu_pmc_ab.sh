#!/bin/bash
# Instruction counters of the fusion kernel for several builds of the library on one scene, one rocprofv3 pass each
# (PMC passes on their own, never with a trace domain).  usage: tools/gpu_pmc_ab.sh <tag> <scene> <name=lib.so> ...
set -u
TAG=$1; SCENE=$2; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  export DMI_LIB_OVERRIDE=$lib
  O=gpurun_out/${TAG}_${name}
  P="python3 bench.py --scene $SCENE --steps 2 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes ${BENCH_ARGS:-}"
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq1 -- $P > $O.sq1.log 2>&1; echo "$name pmc1 rc=$?"
  timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM TA_TA_BUSY_sum --output-format csv -d $O/sq3 -- $P > $O.sq3.log 2>&1; echo "$name pmc3 rc=$?"
  timeout 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ SQC_DCACHE_MISSES SQC_DCACHE_REQ --output-format csv -d $O/sq4 -- $P > $O.sq4.log 2>&1; echo "$name pmc4 rc=$?"
  timeout 400 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/sq2 -- $P > $O.sq2.log 2>&1; echo "$name pmc2 rc=$?"
done
python3 - "$TAG" "$@" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
for spec in sys.argv[2:]:
    name = spec.split("=")[0]
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/{tag}_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fuse_tile_kernel" not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(name, {k: round(v / max(1, n[k] // 1) , 1) for k, v in sorted(tot.items())}, "dispatches", {k: n[k] for k in list(n)[:1]})
PY
