#!/bin/bash
# The round's bench lines in one GPU call: the default run (speckle headline + scenes object), the N > 1 code path rehearsed
# on one GPU in its three exchange modes, and the 4-rank peer-copy rehearsal with every rank on device 0.
set -u
TAG=${1:-bench}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_default.json 2> gpurun_out/${TAG}_default.err; echo "default rc=$?"
python - gpurun_out/${TAG}_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], round(d["value"], 1), "Gproj/s", round(d["ms_per_step"], 3), "ms/step", "kernel", round(d["roofline"]["kernel_ms"], 3))
for k, v in (d.get("scenes") or {}).items():
    print("  scene", k, round(v["value"], 1), "Gproj/s", round(v["ms_per_step"], 3), "ms", v["brick_classes"])
print("  ablation", {k: round(v["kernel_ms"], 2) for k, v in d.get("ablation", {}).items()}, "cpu", d.get("cpu_baseline", {}).get("value"))
PY
for ex in all_reduce reduce_scatter; do
  timeout -k 10 600 python bench.py --force-multi --exchange $ex --steps 5 --warmup 2 --no-strong > gpurun_out/${TAG}_multi1_$ex.json 2> gpurun_out/${TAG}_multi1_$ex.err; echo "multi1 $ex rc=$?"
done
timeout -k 10 600 python bench.py --force-multi --exchange peer_copy --steps 5 --warmup 2 --no-strong > gpurun_out/${TAG}_multi1_peer_copy.json 2> gpurun_out/${TAG}_multi1_peer_copy.err; echo "multi1 peer rc=$?"
timeout -k 10 600 python bench.py --gpus 4 --exchange peer_copy --share-device --workload 256x16@640x480 --steps 3 --warmup 1 > gpurun_out/${TAG}_peer4_shared.json 2> gpurun_out/${TAG}_peer4_shared.err; echo "peer4 rc=$?"
python - $TAG <<'PY'
import json, sys, glob
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}_multi1_*.json") + glob.glob(f"gpurun_out/{sys.argv[1]}_peer4*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    w = d["weak"]
    print(f.split("/")[-1], "ms/step", round(d["ms_per_step"], 3), "exposed", round(w["rank0_exchange_exposed_ms"], 3), "strong", [(s["workload"][:12], round(s["ms_per_step"], 3), s.get("check_vs_single_gpu", {}).get("within_tolerance")) for s in d.get("strong", [])])
PY
