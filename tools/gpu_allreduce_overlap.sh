#!/bin/bash
# Rehearsal (one rank): dmi_multi_fuse with the ALL_REDUCE exchange and n_slabs > 1 under rocprofv3 --kernel-trace -- do RCCL's
# kernels run beside the persistent fusion kernel of the next slab, or only at the slab boundaries?  usage: tools/gpu_allreduce_overlap.sh <tag>
set -u
TAG=${1:-r19u}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --gpus 1 --force-multi --slabs 4 --exchange all_reduce --steps 3 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes --no-strong > $O.log 2>&1; echo "rc=$?"
tail -c 600 $O.log
python3 - $O <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
fuse = [(s, e) for n, s, e in ks if "fuse_tile_kernel" in n]
rccl = [(n, s, e) for n, s, e in ks if "nccl" in n.lower() or "rccl" in n.lower()]
print("fusion launches", len(fuse), "RCCL kernels", len(rccl), sorted(set(n[:60] for n, _, _ in rccl))[:4])
out = []
for n, s, e in rccl:
    ov = sum(max(0, min(e, fe) - max(s, fs)) for fs, fe in fuse)
    out.append({"kernel": n[:50], "ms": (e - s) / 1e6, "overlap_with_fusion_ms": ov / 1e6})
tot = sum(o["ms"] for o in out); tov = sum(o["overlap_with_fusion_ms"] for o in out)
print("RCCL kernel time %.3f ms, of which beside a fusion kernel %.3f ms (%.0f %%)" % (tot, tov, 100 * tov / max(tot, 1e-9)))
for o in out[-8:]:
    print(o)
json.dump({"rccl_kernels": out, "rccl_ms": tot, "overlap_ms": tov, "fusion_launches": len(fuse)}, open(sys.argv[1] + "_overlap.json", "w"), indent=1)
PY
