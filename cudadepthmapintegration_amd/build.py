"""In-tree build of the HIP library (libdmi_hip.so) for gfx950.

hipcc cross-compiles without a GPU.  The shared object lands next to the sources
(cudadepthmapintegration_amd/csrc/libdmi_hip.so) so it travels with the repo snapshot
to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libdmi_hip.so")

SOURCES = ["fusion_kernels.hip", "dmi_capi.hip"]
HEADERS = ["fusion_kernels.h", os.path.join("..", "..", "include", "dmi.h")]

# -ffp-contract=off: no FMA contraction anywhere on the result path (parity contract).
HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (no CPU fallback exists)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc_path()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH
