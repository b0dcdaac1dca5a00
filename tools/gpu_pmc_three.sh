#!/bin/bash
export TMPDIR=/tmp
for name in fast noredo skipw; do
  export DMI_LIB_OVERRIDE=cudadepthmapintegration_amd/csrc/libdmi_hip_exp_$name.so
  O=gpurun_out/r14c_pmc_$name
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -- python3 bench.py --scene speckle --steps 2 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes > $O.log 2>&1
  python3 - $O $name <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fuse_tile_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(sys.argv[2], {k: "%.3e" % (v / max(1, n[k])) for k, v in sorted(tot.items())})
PY
done
