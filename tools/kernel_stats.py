#!/usr/bin/env python3
"""Per-instantiation statistics of fuse_tile_kernel from the gfx950 assembly (no GPU needed): registers, scratch, code
bytes, instruction mix.  usage: tools/kernel_stats.py [asm-out.s] [--filter SUBSTR]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudadepthmapintegration_amd import build  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")]
    out = args[0] if args else os.path.join(tempfile.gettempdir(), "fusion_tile.s")
    if not args or not os.path.exists(out):
        cmd = [build.hipcc_path()] + build.COMMON_FLAGS + build.HIP_FLAGS + ["--cuda-device-only", "-S", os.path.join(build.CSRC, "fusion_tile.hip"), "-o", out]
        subprocess.check_call(cmd)
    text = open(out).read()
    for body in re.split(r"\n(?=_ZN3dmi\S*fuse_tile_kernel\S*:)", text):
        m = re.match(r"(_ZN3dmi\S*fuse_tile_kernelI\w+):", body)
        if not m:
            continue
        name = m.group(1)
        if flt and not any(f in name for f in flt):
            continue
        code = body[: body.find("s_endpgm")]
        ins = [l.split()[0] for l in code.splitlines() if re.match(r"\s+[a-z]", l) and not l.strip().startswith((".", ";"))]
        d = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\b(.*?)\.end_amdhsa_kernel", text, re.S).group(1)
        g = lambda k: re.search(k + r" (\d+)", d).group(1)
        cnt = lambda p: sum(1 for i in ins if re.match(p, i))
        sz = re.search(r"; codeLenInByte = (\d+)", body)
        print(name[-60:], "vgpr", g(r"\.amdhsa_next_free_vgpr"), "sgpr", g(r"\.amdhsa_next_free_sgpr"), "scratch", g(r"\.amdhsa_private_segment_fixed_size"),
              "bytes", sz.group(1) if sz else "?", "insts", len(ins), "valu", cnt(r"v_"), "salu", cnt(r"s_(?!load|buffer|waitcnt|cbranch|branch|nop)"),
              "branch", cnt(r"s_c?branch"), "smem", cnt(r"s_(load|buffer)"), "vmem", cnt(r"(buffer|global)_"), "scratch_ops", cnt(r"scratch_"))


if __name__ == "__main__":
    main()
