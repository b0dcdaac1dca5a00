#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace directory: the launches of the last fusion step (from its first preparation launch on), each with
its start relative to the step's first launch, its duration and the gap to the launch before it.  usage: step_timeline.py DIR"""
import csv, glob, re, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step's first launch: the coarse classification pass (it fills the per-launch tables too), or launch_tables_kernel in a fusion
# without brick classes (cz_table_kernel: traces of rounds 1-3)
firsts = ("classify_coarse_kernel", "launch_tables_kernel", "cz_table_kernel", "wk_table_kernel")
starts = [i for i, r in enumerate(rows) if any(f in r[2] for f in firsts)]
if not starts:
    sys.exit("no fusion step in the trace")
lo = starts[-1]
# a step may be preceded by the grid reset (fill); include what lies between the previous fusion kernel and this step
hi = lo
while hi < len(rows) and "fuse_tile_kernel" not in rows[hi][2] and "fuse_general" not in rows[hi][2]:
    hi += 1
t0 = rows[lo][0]
prev_end = None
busy = 0
for s, e, n in rows[lo:hi + 1]:
    short = re.sub(r"<.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "")).split("(")[0].split("::")[-1][:40]
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {short}")
    prev_end = e
    busy += e - s
print(f"step span {(rows[hi][1] - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
# spans of the last 4 steps, start of one step's first launch to the next one's
for a, b in zip(starts[-5:-1], starts[-4:]):
    print(f"step to step {(rows[b][0] - rows[a][0]) / 1e3:.1f} us")
