"""In-tree build of the HIP library (libdmi_hip.so) for gfx950.

hipcc cross-compiles without a GPU.  The shared object lands next to the sources
(cudadepthmapintegration_amd/csrc/libdmi_hip.so) so it travels with the repo snapshot
to the GPU box; it is git-ignored.  Objects are compiled in parallel (one hipcc per source)
into build/obj/ and relinked only when a source or header is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
LIB_PATH = os.path.join(CSRC, "libdmi_hip.so")

SOURCES = ["fusion_kernels.hip", "fusion_tile.hip", "fusion_classify.hip", "coloration_kernels.hip", "grid_post.hip", "dmi_capi.hip", "host/recon_host.cpp", "host/vti_reader.cpp", "host/dmi_host_capi.cpp"]
HEADERS = ["fusion_kernels.h", "fusion_device.h", "fusion_tile_acc.inc", os.path.join("host", "recon_host.h"),
           os.path.join("host", "vti_reader.h"),
           os.path.join("..", "..", "include", "dmi.h"), os.path.join("..", "..", "include", "dmi_host.h")]

# -ffp-contract=off: no FMA contraction anywhere on the result path (parity contract).
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
                # the tiled kernel's accumulator file names VGPRs above the compiler's budget on purpose (fusion_tile_acc.inc)
                "-Wno-inline-asm", "-Wno-pass-failed"]
# -disable-promote-alloca-to-vector: without it hipcc turns the tiled kernel's register accumulators into one
# 32-register tuple that it spills and reloads whole (3000+ spill instructions at 4 waves per SIMD).
HIP_FLAGS = ["--offload-arch=gfx950", "-mllvm", "-disable-promote-alloca-to-vector"]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (no CPU fallback exists)")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    return [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, src.replace("/", "_") + ".o")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers()
    return _stale(LIB_PATH, deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    hipcc = hipcc_path()
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = _headers()

    def compile_one(src: str):
        path = os.path.join(CSRC, src)
        obj = _obj(src)
        if not force and not _stale(obj, [path] + headers):
            return
        # .cpp files are host-only C++ above the C ABI: no device code, no HIP headers
        cmd = [hipcc] + COMMON_FLAGS + (HIP_FLAGS if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in _sources()] + ["-lz", "-o", LIB_PATH]  # zlib: compressed .vti arrays (host/vti_reader.cpp)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH
