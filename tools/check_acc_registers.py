#!/usr/bin/env python3
"""Audit of the tiled kernel's accumulator file (fusion_tile_acc.inc): compiles fusion_tile.hip to assembly
and checks, for every fuse_tile_kernel instantiation, that no compiler-generated instruction (anything outside
the ;;#ASMSTART / ;;#ASMEND brackets) names a VGPR at or above the instantiation's accumulator base.
Run after every change of the kernel or the compiler flags:  python tools/check_acc_registers.py
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudadepthmapintegration_amd import build  # noqa: E402

BASES = {8: 64, 7: 72, 6: 80, 5: 96}


def main():
    src = os.path.join(build.CSRC, "fusion_tile.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "tile.s")
        cmd = [build.hipcc_path()] + build.COMMON_FLAGS + build.HIP_FLAGS + ["--cuda-device-only", "-S", src, "-o", out]
        subprocess.check_call(cmd)
        text = open(out).read()
    bad = 0
    kernels = re.split(r"\n(?=_ZN3dmi\S*fuse_tile_kernel\S*:)", text)
    checked = 0
    for body in kernels:
        m = re.match(r"(_ZN3dmi\S*fuse_tile_kernelI\w+):", body)
        if not m:
            continue
        name = m.group(1)
        body = body[: body.find("s_endpgm")]
        tpl = re.search(r"fuse_tile_kernelI\w\wLi(\d+)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb([01])", name)
        tk, minw = int(tpl.group(1)), int(tpl.group(4))
        base = BASES.get(minw, 128)
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            if in_asm or not re.match(r"\s+(v_|global_|buffer_|ds_|scratch_|flat_)", line):
                continue
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", line)]
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", line):
                regs += [int(a), int(b)]
            if any(r >= base for r in regs):
                print(f"{name}: compiler instruction touches the accumulator file (base v{base}): {line.strip()}")
                bad += 1
        checked += 1
    print(f"checked {checked} fuse_tile_kernel instantiations, {bad} violations")
    return 1 if bad or checked == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
