"""In-tree build of the HIP library (libdmi_hip.so) for gfx950.

hipcc cross-compiles without a GPU.  The shared object lands next to the sources
(cudadepthmapintegration_amd/csrc/libdmi_hip.so) so it travels with the repo snapshot
to the GPU box; it is git-ignored.  Objects are compiled in parallel (one hipcc per source)
into build/obj/ and relinked only when a source or header is newer.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
LIB_PATH = os.path.join(CSRC, "libdmi_hip.so")

SOURCES = ["fusion_kernels.hip", "fusion_tile.hip", "fusion_classify.hip", "coloration_kernels.hip", "grid_post.hip", "dmi_capi.hip", "dmi_multi.hip", "host/recon_host.cpp", "host/vti_reader.cpp", "host/recon_cli.cpp", "host/dmi_host_capi.cpp"]
HEADERS = ["fusion_kernels.h", "fusion_device.h", "fusion_tile_acc.inc", os.path.join("host", "recon_host.h"),
           os.path.join("host", "vti_reader.h"), os.path.join("host", "recon_cli.h"),
           os.path.join("..", "..", "include", "dmi.h"), os.path.join("..", "..", "include", "dmi_host.h")]

# -ffp-contract=off: no FMA contraction anywhere on the result path (parity contract).
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
                # the tiled kernel's accumulator file names VGPRs above the compiler's budget on purpose (fusion_tile_acc.inc)
                "-Wno-inline-asm", "-Wno-pass-failed"]
# -disable-promote-alloca-to-vector: without it hipcc turns the tiled kernel's register accumulators into one
# 32-register tuple that it spills and reloads whole (3000+ spill instructions at 4 waves per SIMD).
# -amdgpu-use-amdgpu-trackers: the scheduler measures register pressure with the AMDGPU-specific trackers; the tiled kernel's view
# body, whose every change shows up as spill traffic, runs 3-4 % faster for it (cfg 3 speckle 13.72 -> 13.29 ms, dense 4.49 -> 4.33;
# max-ilp scheduling and reversed local assignment cost 3-5 %: profiles/r18c_exp_llvm_flags.json)
# (--discard-all at the device link: two thirds of a code object's .symtab are LOCAL symbols -- rocPRIM's mangled names run to
# kilobytes each --: 0.65 MB of 1.9 in coloration_kernels.hip.o alone.  --strip-all would save 0.15 MB more, but a code object
# without any .symtab crashes the process at its first kernel launch on ROCm 7.2.)
HIP_FLAGS = ["--offload-arch=gfx950", "-mllvm", "-disable-promote-alloca-to-vector", "-Xoffload-linker", "--discard-all"]
# hidden LLVM options: kept only where this hipcc knows them (probed once with an empty translation unit; a toolchain without
# the option would abort every compile with "Unknown command line argument").  build_record.json lists the flags used.
OPTIONAL_LLVM_FLAGS = ["-amdgpu-use-amdgpu-trackers=1"]


def _probe_llvm_flag(flag: str) -> bool:
    import tempfile

    try:
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "probe.hip")
            with open(src, "w") as fh:
                fh.write("__global__ void probe() {}\n")
            r = subprocess.run([hipcc_path(), "--offload-arch=gfx950", "--cuda-device-only", "-mllvm", flag, "-c", src, "-o",
                                os.path.join(tmp, "probe.o")], capture_output=True)
            return r.returncode == 0
    except (OSError, RuntimeError):
        return False


_FLAG_CACHE = os.path.join(ROOT, "build", "llvm_flag_probe.json")


def _optional_flags() -> list:
    import json

    # The toolchain's identity WITHOUT running it: this module is imported by every process that loads the library, and a
    # process whose GPU is already initialised (any program under `rocprofv3 --pmc`) must not start another one -- the GPU boxes
    # of this pool refuse that.  Path, size and modification time of the driver and of the clang it ships with.
    try:
        parts = []
        hip = os.path.realpath(hipcc_path())
        for f in (hip, os.path.realpath(os.path.join(os.path.dirname(hip), "..", "lib", "llvm", "bin", "clang"))):
            if os.path.exists(f):
                st = os.stat(f)
                parts.append("%s:%d:%d" % (f, st.st_size, int(st.st_mtime)))
        version = ";".join(parts)
    except (OSError, RuntimeError):
        return []
    try:
        with open(_FLAG_CACHE) as fh:
            rec = json.load(fh)
        if rec.get("hipcc") == version and set(rec.get("flags", {})) == set(OPTIONAL_LLVM_FLAGS):
            return [x for f in OPTIONAL_LLVM_FLAGS if rec["flags"][f] for x in ("-mllvm", f)]
    except (OSError, ValueError):
        pass
    flags = {f: _probe_llvm_flag(f) for f in OPTIONAL_LLVM_FLAGS}
    try:
        os.makedirs(os.path.dirname(_FLAG_CACHE), exist_ok=True)
        with open(_FLAG_CACHE, "w") as fh:
            json.dump({"hipcc": version, "flags": flags}, fh)
    except OSError:
        pass
    return [x for f in OPTIONAL_LLVM_FLAGS if flags[f] for x in ("-mllvm", f)]
# DMI_TUNING=1 in the environment of the BUILD compiles the experiment switches of tools/ in (getenv-driven launch
# geometry, dropped depth loads ...).  The default library contains none of them.
if os.environ.get("DMI_EXP"):  # tools/gpu_exp.sh: "NAME:-DDMI_EXP_X=0 ..." -> build/obj_exp_NAME, libdmi_hip_exp_NAME.so
    _name, _, _defs = os.environ["DMI_EXP"].partition(":")
    COMMON_FLAGS = COMMON_FLAGS + _defs.split()
    OBJ_DIR = os.path.join(ROOT, "build", "obj_exp_" + _name)
    LIB_PATH = os.path.join(CSRC, "libdmi_hip_exp_" + _name + ".so")
if os.environ.get("DMI_TUNING"):  # a separate library and object directory: never mistaken for the shipped one
    COMMON_FLAGS = COMMON_FLAGS + ["-DDMI_TUNING"]
    OBJ_DIR = os.path.join(ROOT, "build", "obj_tuning")
    LIB_PATH = os.path.join(CSRC, "libdmi_hip_tuning.so")


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (no CPU fallback exists)")


HIP_FLAGS = HIP_FLAGS + _optional_flags()


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


# A prebuilt library to load instead (A/B timing against an older build in the same GPU call): never rebuilt
LIB_OVERRIDE = os.environ.get("DMI_LIB_OVERRIDE")
if LIB_OVERRIDE:
    LIB_PATH = os.path.abspath(LIB_OVERRIDE)


def source_digest(files=None) -> str:
    """sha256 over the compiler flags and the contents of the sources and headers: what the library was built FROM.
    Written next to the library (libdmi_hip.so.digest) by build(); a library whose digest matches is up to date whatever
    the files' timestamps say (a checkout, a copy to another machine), one whose digest differs is not."""
    import hashlib

    h = hashlib.sha256()
    h.update(" ".join(COMMON_FLAGS + HIP_FLAGS).encode())
    if files is None:
        files = [os.path.join(CSRC, s) for s in _sources()] + _headers() + [os.path.join(CSRC, "host", "recon_cli_main.cpp")]
    for f in files:
        h.update(os.path.relpath(f, CSRC).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _digest_path() -> str:
    return LIB_PATH + ".digest"


def needs_build() -> bool:
    if LIB_OVERRIDE:
        return False
    if not os.path.exists(LIB_PATH):
        return True
    if os.path.exists(_digest_path()):
        with open(_digest_path()) as fh:
            return fh.read().strip() != source_digest()
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers()
    return _stale(LIB_PATH, deps)


ACC_BASES = {8: 64, 7: 72, 6: 80, 5: 96}  # __launch_bounds__ MINW -> first accumulator VGPR (fusion_tile.hip acc_base)


def audit_accumulator_registers(asm_text: str) -> tuple[int, list[str]]:
    """The tiled kernel keeps its running sums in v[BASE .. BASE+2*TK), registers the compiler is told nothing about
    beyond clobber lists (fusion_tile_acc.inc).  That is only sound if no COMPILER-generated instruction of any
    fuse_tile_kernel instantiation names a VGPR at or above BASE, and if the kernel descriptor allocates all of them.
    Returns (instantiations checked, violations)."""
    import re

    bad: list[str] = []
    checked = 0
    for body in re.split(r"\n(?=_ZN3dmi\S*fuse_tile_kernel\S*:)", asm_text):
        m = re.match(r"(_ZN3dmi\S*fuse_tile_kernelI\w+):", body)
        if not m:
            continue
        name = m.group(1)
        tpl = re.search(r"fuse_tile_kernelI\w\wLi(\d+)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb([01])", name)
        tk, minw = int(tpl.group(1)), int(tpl.group(4))
        base = ACC_BASES.get(minw, 128)
        code = body[: body.find("s_endpgm")]
        in_asm = False
        scc_from_asm = False  # SCC currently holds what an asm statement's s_and_saveexec_b64 left there
        for line in code.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            # The EXEC-masked statements write SCC (s_and_saveexec_b64) and say so in their clobber lists; a compiler
            # instruction that READS SCC right after one would mean a clobber list lost it (results silently wrong).
            op = line.split()[0] if line.split() else ""
            if in_asm:
                if op.startswith("s_and_saveexec"):
                    scc_from_asm = True
            elif op.startswith("s_"):
                if scc_from_asm and re.match(r"s_(cbranch_scc|cselect|cmov|addc|subb)", op):
                    bad.append(f"{name}: {op} reads the SCC an asm statement's s_and_saveexec_b64 wrote: {line.strip()}")
                if not re.match(r"s_(mov|movk|load|buffer_load|waitcnt|nop|branch|cbranch|cselect|cmov|getpc|setpc|swappc|sleep|barrier|endpgm)", op):
                    scc_from_asm = False  # scalar ALU and compare instructions define SCC anew
            elif line.rstrip().endswith(":") and not line.startswith((" ", "\t")):
                scc_from_asm = False  # a label: control flow joins here, the compiler's own bookkeeping applies
            if in_asm or not re.match(r"\s+(v_|global_|buffer_|ds_|scratch_|flat_)", line):
                continue
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", line)]
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", line):
                regs += [int(a), int(b)]
            if any(r >= base for r in regs):
                bad.append(f"{name}: compiler instruction touches the accumulator file (base v{base}): {line.strip()}")
        # the descriptor must allocate the accumulators: .amdhsa_next_free_vgpr of this kernel >= BASE + 2*TK
        d = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\b.*?\.amdhsa_next_free_vgpr (\d+)", asm_text, re.S)
        if not d:
            bad.append(f"{name}: no kernel descriptor found")
        elif int(d.group(1)) < base + 2 * tk:
            bad.append(f"{name}: descriptor allocates {d.group(1)} VGPRs, the accumulators need {base + 2 * tk}")
        checked += 1
    if checked == 0:
        bad.append("no fuse_tile_kernel instantiation found in the assembly")
    return checked, bad


def scratch_sizes(asm_text: str) -> dict:
    """.amdhsa_private_segment_fixed_size of every fuse_tile_kernel instantiation (bytes of scratch memory per lane)."""
    import re

    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (_ZN3dmi\S*fuse_tile_kernel\S*)\b(.*?)\.end_amdhsa_kernel", asm_text, re.S):
        size = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2))
        out[m.group(1)] = int(size.group(1)) if size else 0
    return out


def wait_state_counts(asm_text: str) -> dict:
    """s_nop instructions per fuse_tile_kernel instantiation: a canary, not a rule.  hipcc puts a wait state between an asm
    statement and an immediate reader of its result; the tier-1 chains are written so that few are needed (fusion_tile.hip, struct
    ordered: 11.95 -> 11.34 ms at cfg 3).  A toolchain that schedules them differently shows here first (the headline kernel of
    round 5's build: ~300 in 12 000 lines)."""
    import re

    out = {}
    for m in re.finditer(r"^(_ZN3dmi\S*fuse_tile_kernel\S*):.*?\.end_amdhsa_kernel", asm_text, re.S | re.M):
        out[m.group(1)] = len(re.findall(r"^\s*s_nop\b", m.group(0), re.M))
    return out


def _audit_digest() -> str:
    return source_digest([os.path.join(CSRC, "fusion_tile.hip")] + _headers())


def audit_is_current() -> bool:
    """The audit record of this object directory is about the tiled kernel's present source and flags."""
    import json

    marker = os.path.join(OBJ_DIR, "acc_audit.json")
    try:
        with open(marker) as fh:
            return json.load(fh).get("digest") == _audit_digest()
    except (OSError, ValueError):
        return False


def run_accumulator_audit(verbose: bool = False) -> int:
    """Compile fusion_tile.hip to gfx950 assembly with the build's flags and audit it; raises on any violation.
    Runs inside build() whenever fusion_tile.hip is recompiled, and as a non-GPU test."""
    import json
    import tempfile

    hipcc = hipcc_path()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "fusion_tile.s")
        cmd = [hipcc] + COMMON_FLAGS + HIP_FLAGS + ["--cuda-device-only", "-S", os.path.join(CSRC, "fusion_tile.hip"), "-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(out) as fh:
            text = fh.read()
    checked, bad = audit_accumulator_registers(text)
    scratch = scratch_sizes(text)
    # Scratch memory, every instantiation (round 5).  The production kernels -- f32 depth tables, pinhole, no hit counters -- with
    # 16-voxel columns (what 512^3 and the maps with holes run) use none at all; with 8-voxel columns (80 VGPRs for the compiler,
    # five waves per SIMD) a few spill slots are tolerated, bounded here; the counted kernels (an array of hit counters per
    # column), f64 depth tables and general K are the rare paths (launch_shape): listed, not bounded.  A toolchain or source
    # change that makes a production kernel spill is caught here, not in a profile months later.
    nops = wait_state_counts(text)
    table = {}
    for name, size in scratch.items():
        m = re.search(r"fuse_tile_kernelI(\w)(\w)Li(\d+)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb([01])ELb([01])ELb([01])ELb([01])ELb([01])(?:ELb([01]))?E", name)
        if not m:
            continue
        depth, grid, tk, wx, wy, minw, group, count, rot, genk, stay, win, zf = m.groups()
        key = f"depth={depth} grid={grid} tk={tk} waves={wx}x{wy} count={count} rot={rot} genk={genk} stay={stay} win={win} zf={zf or 0}"
        production = depth == "f" and count == "0" and genk == "0" and wx == "1" and wy == "1"
        # (a tuning build carries debug counters through the kernel: a few slots are its own)
        limit = None if not production else ((16 if "-DDMI_TUNING" in COMMON_FLAGS else 0) if tk == "16" else 48)
        table[key] = {"scratch_bytes_per_lane": size, "production": production, "limit": limit, "s_nop": nops.get(name)}
        if limit is not None and size > limit:
            bad.append(f"{name}: {size} bytes of scratch memory per lane in a production instantiation (limit {limit})")
    if bad:
        raise RuntimeError("accumulator-register audit of fusion_tile.hip FAILED (a toolchain change broke the hidden "
                           "register file; results would be silently wrong):\n  " + "\n  ".join(bad[:20]))
    version = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, "acc_audit.json"), "w") as fh:
        json.dump({"instantiations": checked, "violations": 0, "hipcc": version[:2], "digest": _audit_digest(),
                   "scratch_bytes_per_lane": {"max": max(scratch.values()) if scratch else None,
                                              "instantiations_with_scratch": sum(1 for v in scratch.values() if v),
                                              "production_max": max([v["scratch_bytes_per_lane"] for v in table.values() if v["production"]] or [0])},
                   "instantiation_table": table}, fh, indent=1)
    if verbose:
        print(f"accumulator audit: {checked} fuse_tile_kernel instantiations clean ({version[0] if version else 'hipcc'})", flush=True)
    return checked


def _record(mode: str) -> None:
    """build/obj/build_record.json: what the last build() call in this tree did (a reader can tell a recompilation from
    the acceptance of a prebuilt library by its digest)."""
    import json
    import time

    try:
        os.makedirs(OBJ_DIR, exist_ok=True)
        with open(os.path.join(OBJ_DIR, "build_record.json"), "w") as fh:
            json.dump({"mode": mode, "library": os.path.relpath(LIB_PATH, ROOT), "digest": source_digest(), "time": time.time(),
                       "flags": COMMON_FLAGS + HIP_FLAGS}, fh)
    except OSError:
        pass


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        _record("up_to_date: the library's digest file names the present sources and flags; nothing compiled")
        return LIB_PATH  # (the command-line tool is built with the library below, or on demand: capi.cli_binary)
    hipcc = hipcc_path()
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = _headers()
    digest = source_digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(_digest_path()):
        # the contents changed but perhaps not the timestamps' order (files restored with old dates): trust no object
        srcs = [os.path.join(CSRC, s) for s in _sources()]
        if not any(_stale(_obj(s), [p] + headers) for s, p in zip(_sources(), srcs)):
            force = True

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

    def audit_if_needed(_):
        # same staleness rule as the object: a recompiled tiled kernel is a re-audited one
        if force or not audit_is_current():
            run_accumulator_audit(verbose)

    with ThreadPoolExecutor(max_workers=4) as ex:
        jobs = [ex.submit(compile_one, s) for s in _sources()] + [ex.submit(audit_if_needed, None)]
        for j in jobs:
            j.result()
    # linked inside the object directory (hipcc leaves its offload-bundler temporaries next to the output), then moved
    # into place in one step: a concurrent loader sees the old library or the new one, never half of one
    staged = os.path.join(OBJ_DIR, os.path.basename(LIB_PATH) + f".{os.getpid()}.tmp")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in _sources()] + ["-lz", "-ldl", "-pthread", "-o", staged]  # zlib: compressed .vti arrays (host/vti_reader.cpp)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(staged, LIB_PATH)
    with open(_digest_path(), "w") as fh:
        fh.write(digest + "\n")
    build_cli(verbose)
    _record("rebuilt: hipcc compiled the stale objects and linked the library")
    return LIB_PATH


CLI_PATH = os.path.join(CSRC, "dmi_reconstruction")


def build_cli(verbose: bool = False) -> str:
    """The reference's `Reconstruction` command line over the library (csrc/host/recon_cli_main.cpp): a few lines of
    main() linked against the .so next to it (rpath $ORIGIN).  Only for the default library.  Built by build() right after
    the library and on demand by capi.cli_binary(); up to date = its digest file names the library's source digest
    (timestamps say nothing after a checkout or a copy).  Linked to a temporary and moved into place."""
    if os.path.basename(LIB_PATH) != "libdmi_hip.so" or LIB_OVERRIDE:
        return ""
    src = os.path.join(CSRC, "host", "recon_cli_main.cpp")
    digest = source_digest()
    marker = CLI_PATH + ".digest"
    if os.path.exists(CLI_PATH) and os.path.exists(marker):
        with open(marker) as fh:
            if fh.read().strip() == digest:
                return CLI_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    staged = os.path.join(OBJ_DIR, f"dmi_reconstruction.{os.getpid()}.tmp")
    cmd = [hipcc_path()] + COMMON_FLAGS + [src, "-L" + CSRC, "-ldmi_hip", "-Wl,-rpath,$ORIGIN", "-o", staged]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(staged, CLI_PATH)
    with open(marker, "w") as fh:
        fh.write(digest + "\n")
    return CLI_PATH
