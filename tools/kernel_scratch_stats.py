#!/usr/bin/env python3
"""Scratch bytes per lane of the tiled kernel's instantiations in a FAST build (-DDMI_FAST_BUILD: the two default shapes, 16
kernels, 20 s instead of two minutes), and the gfx950 assembly itself in /tmp/dis/tile_fast3.s for reading.

    python tools/kernel_scratch_stats.py [-DSOMETHING ...]        # TILE_SRC=other.hip compiles another file of csrc/

What a change does to register pressure shows here before a full build's audit (cudadepthmapintegration_amd/build.py) fails on
it: the 16-voxel production kernels may use no scratch at all, the 8-voxel ones at most 48 bytes."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudadepthmapintegration_amd import build as b  # noqa: E402

os.makedirs("/tmp/dis", exist_ok=True)
out = "/tmp/dis/tile_fast3.s"
cmd = [b.hipcc_path()] + b.COMMON_FLAGS + b.HIP_FLAGS + ["-DDMI_FAST_BUILD"] + sys.argv[1:] + \
      ["--cuda-device-only", "-S", os.path.join(b.CSRC, os.environ.get("TILE_SRC", "fusion_tile.hip")), "-o", out]
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
text = open(out).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
    name, body = m.group(1), m.group(2)
    scratch = re.search(r"private_segment_fixed_size (\d+)", body).group(1)
    k = re.search(r"fuse_tile_kernelI(\w\wLi\d+E.*?)EvNS", name)
    if k:  # depth, grid, column height, then MINW, GROUP and the boolean parameters COUNT ROT GENK STAY WIN ZF
        print(k.group(1).replace("ELi1ELi1E", "").replace("ELb", ""), "scratch", scratch)
