#!/bin/bash
# Development: compiles fusion_tile.hip with DMI_FAST_BUILD (the two default instantiations only) to assembly and prints the
# spill / size statistics of the default kernels.  usage: tools/fast_asm.sh [out.s]
OUT=${1:-/tmp/fast_fusion_tile.s}
cd "$(dirname "$0")/../cudadepthmapintegration_amd/csrc" || exit 1
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-inline-asm -Wno-pass-failed -Wno-unused-function -DDMI_FAST_BUILD \
  --offload-arch=gfx950 -mllvm -disable-promote-alloca-to-vector -mllvm -amdgpu-use-amdgpu-trackers=1 --cuda-device-only -S fusion_tile.hip -o "$OUT" 2>&1 | grep -v "hip-link" | head -20
python3 - "$OUT" <<'PY'
import re, sys, collections
text = open(sys.argv[1]).read()
for body in re.split(r"\n(?=_ZN3dmi\S*fuse_tile_kernel\S*:)", text):
    m = re.match(r"(_ZN3dmi\S*fuse_tile_kernelI\w+):", body)
    if not m: continue
    name = m.group(1)
    code = body[: body.find("s_endpgm")]
    c = collections.Counter()
    for l in code.splitlines():
        mm = re.match(r"\s+([a-z_0-9]+)", l)
        if mm: c[mm.group(1)] += 1
    sz = re.search(r"; codeLenInByte = (\d+)", body)
    d = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\b(.*?)\.end_amdhsa_kernel", text, re.S).group(1)
    scr = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", d).group(1)
    print(name[name.index("kernelI")+7:name.index("EEvNS")], "bytes", sz.group(1) if sz else "?", "readlane", c["v_readlane_b32"], "writelane", c["v_writelane_b32"],
          "scratch", scr, "scratch_ops", sum(v for k, v in c.items() if k.startswith("scratch_")), "s_nop", c["s_nop"])
PY
