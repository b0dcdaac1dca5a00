#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/, scratch) into small tracked files under profiles/.

    python tools/summarize_profiles.py --tag r01_baseline --kernel fuse \
        --kt gpurun_out/prof_kt --pmc gpurun_out/pmc_sq1 gpurun_out/pmc_fetch ... [--key cfg3:dense:f32]

Writes
    profiles/<tag>_kernel_stats.csv   the `rocprofv3 --kernel-trace --stats` summary, verbatim
    profiles/<tag>_pmc.json           per kernel: mean counter value per dispatch, every --pmc pass merged
    profiles/<tag>_summary.md         the derived figures (clock, VALU issue share, HBM-side traffic ...)
and, with --key, records the HBM-side bytes per launch of the fusion kernel in profiles/pmc_traffic.json
(bench.py reports it as roofline.traffic for the matching workload).

Counter units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM, rocprofv3 PMC
slots): FETCH_SIZE / WRITE_SIZE are KiB summed over the 8 XCDs; FETCH_SIZE tallies every fabric read
request at 64 B although a 128-B line fetch is one request, so it can read as little as 1/2 of the bytes
moved (exactly 1/2 for wide coalesced streams); Infinity-Cache hits are included.  The summary therefore
gives the read side as a [FETCH_SIZE, 2 x FETCH_SIZE] bracket unless a calibration factor measured on
this access pattern (tools/fetch_calibration.hip) is passed with --fetch-scale.
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def first(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def read_pmc(dirs):
    """kernel name -> counter -> list of per-dispatch values."""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for d in dirs:
        f = first(os.path.join(d, "**", "*counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                       "scratch": int(r["Scratch_Size"]), "workgroup": int(r["Workgroup_Size"]),
                       "grid": int(r["Grid_Size"])}
    return out, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--kernel", default="fuse", help="substring selecting the fusion kernel")
    ap.add_argument("--kt", default=None, help="directory of the --kernel-trace --stats run")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--key", default=None, help="workload:scene:grid_dtype key for profiles/pmc_traffic.json")
    ap.add_argument("--fetch-scale", type=float, default=None,
                    help="bytes really moved per FETCH_SIZE byte for this access pattern (calibrated)")
    ap.add_argument("--projections", type=float, default=None, help="voxel-projections per launch")
    args = ap.parse_args()

    pdir = os.path.join(ROOT, "profiles")
    os.makedirs(pdir, exist_ok=True)
    lines = [f"# rocprofv3 summary `{args.tag}`", ""]
    kern_ms = None
    if args.kt:
        stats = first(os.path.join(args.kt, "**", "*kernel_stats.csv"))
        if stats:
            shutil.copyfile(stats, os.path.join(pdir, f"{args.tag}_kernel_stats.csv"))
            lines += ["## kernel trace (`rocprofv3 --kernel-trace --stats`)", "", "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
            for r in csv.DictReader(open(stats)):
                lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.4f} | {r['Percentage']} |")
                if args.kernel in r["Name"] and kern_ms is None:
                    kern_ms = float(r["AverageNs"]) / 1e6
            lines.append("")
    pmc, meta = read_pmc(args.pmc)
    merged = {}
    for k, counters in pmc.items():
        merged[k] = {c: sum(v) / len(v) for c, v in counters.items()}
        merged[k]["_dispatches"] = max(len(v) for v in counters.values())
        merged[k]["_meta"] = meta.get(k, {})
    with open(os.path.join(pdir, f"{args.tag}_pmc.json"), "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)

    sel = [k for k in merged if args.kernel in k]
    for k in sel:
        c = merged[k]
        lines += [f"## PMC, mean per dispatch: `{k[:100]}`", "", f"resources as rocprofv3 prints them: {c['_meta']}",
                  "(on gfx950 its VGPR_Count is half of what the kernel descriptor allocates: the build's register audit, "
                  "build/obj/acc_audit.json, and llvm-readelf give the number of 32-bit registers)", ""]
        if "SQ_INSTS_VALU" in c:
            v = c["SQ_INSTS_VALU"]
            lines.append(f"* SQ_INSTS_VALU = {v:.4g} wave-instructions" +
                         (f" = {v * 64 / args.projections:.1f} per voxel-projection-wave (64 projections)" if args.projections else ""))
            if kern_ms:
                lines.append(f"* VALU issue floor at 4 cycles per fp64-rate wave-instruction on 1024 SIMDs, 2.4 GHz: "
                             f"{v * 4 / 1024 / 2.4e9 * 1e3:.2f} ms of {kern_ms:.2f} ms")
        if "TA_TA_BUSY_sum" in c and kern_ms:
            lines.append(f"* texture-addresser floor: TA_TA_BUSY_sum over 256 CUs (one addresser each) at 2.4 GHz: "
                         f"{c['TA_TA_BUSY_sum'] / 256 / 2.4e9 * 1e3:.2f} ms of {kern_ms:.2f} ms")
        if "SQ_INSTS_SALU" in c and kern_ms:
            sa = c["SQ_INSTS_SALU"] + c.get("SQ_INSTS_BRANCH", 0.0) + c.get("SQ_INSTS_SMEM", 0.0)
            lines.append(f"* scalar-issue floor: SALU + branch + SMEM wave-instructions at one per cycle per CU (the four SIMDs of "
                         f"a CU share one scalar unit), 256 CUs, 2.4 GHz: {sa / 256 / 2.4e9 * 1e3:.2f} ms of {kern_ms:.2f} ms")
        for name in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
                     "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                     "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "TA_BUSY_avr", "TA_TA_BUSY_sum",
                     "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
            if name in c:
                lines.append(f"* {name} = {c[name]:.5g}")
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            lines.append(f"* L2 hit rate = {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
        fetch = c.get("FETCH_SIZE")
        write = c.get("WRITE_SIZE")
        traffic = None
        if fetch is not None:
            fb = fetch * 1024
            if args.fetch_scale:
                lines.append(f"* FETCH_SIZE = {fetch:.5g} KiB -> {fb * args.fetch_scale / 1e9:.2f} GB read at the fabric side "
                             f"(calibrated scale {args.fetch_scale:.2f}; Infinity-Cache hits included)")
                fb *= args.fetch_scale
            else:
                lines.append(f"* FETCH_SIZE = {fetch:.5g} KiB -> between {fb / 1e9:.2f} and {2 * fb / 1e9:.2f} GB read at the fabric side "
                             f"(64-B tally of possibly 128-B requests; Infinity-Cache hits included)")
                fb *= 2.0  # conservative: the guide's gfx950 correction
            traffic = fb
        if write is not None:
            lines.append(f"* WRITE_SIZE = {write:.5g} KiB -> {write * 1024 / 1e9:.3f} GB written")
            traffic = (traffic or 0.0) + write * 1024
        if traffic is not None:
            lines.append(f"* HBM-side traffic per launch (read bracket's upper end + writes) = {traffic / 1e9:.2f} GB")
            if kern_ms:
                lines.append(f"* = {traffic / (kern_ms * 1e-3) / 1e9:.0f} GB/s over the {kern_ms:.2f} ms launch")
            if args.key:
                path = os.path.join(pdir, "pmc_traffic.json")
                db = json.load(open(path)) if os.path.exists(path) else {}
                db[args.key] = {"hbm_bytes_per_launch": traffic, "fetch_kib": fetch, "write_kib": write,
                                "fetch_scale": args.fetch_scale or 2.0, "tag": args.tag, "kernel": k[:120],
                                # wave-instructions per launch (bench.py's roofline_issue object)
                                "valu_insts": c.get("SQ_INSTS_VALU"), "salu_insts": c.get("SQ_INSTS_SALU"),
                                "branch_insts": c.get("SQ_INSTS_BRANCH"), "smem_insts": c.get("SQ_INSTS_SMEM"),
                                # quad-cycles the vector pipes spent executing (a quarter-rate instruction counts four times)
                                "valu_active_quad_cycles": c.get("SQ_ACTIVE_INST_VALU"),
                                # busy cycles of the texture addressers, summed over the 256 CUs' (one per CU); gathers and
                                # cross-lane look-ups (ds_bpermute_b32) issued
                                "ta_busy_cycles": c.get("TA_TA_BUSY_sum"), "vmem_rd_insts": c.get("SQ_INSTS_VMEM_RD"),
                                "lds_insts": c.get("SQ_INSTS_LDS")}
                json.dump(db, open(path, "w"), indent=1, sort_keys=True)
        lines.append("")
    with open(os.path.join(pdir, f"{args.tag}_summary.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
