"""CPU-side checks of the C-ABI boundary: the library builds, loads and exports exactly what
include/dmi.h declares; with no GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import sys

import pytest

from cudadepthmapintegration_amd import capi, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_match_binding_list():
    assert _declared_symbols() == sorted(capi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.load()._name)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/dmi.h but not exported"


def test_abi_version_and_defaults():
    lib = capi.load()
    assert lib.dmi_abi_version() == 5
    o = capi.OptionsC()
    lib.dmi_default_options(ctypes.byref(o))
    assert (o.device, o.grid_dtype, o.depth_storage, o.count_hits, o.kernel_variant) == (0, capi.DMI_F64, 0, 0, 0)
    assert lib.dmi_last_error(None) is not None


def test_struct_sizes_match_the_bindings():
    lib = capi.load()
    assert lib.dmi_sizeof_info() == ctypes.sizeof(capi.InfoC)
    assert lib.dmi_sizeof_timings() == ctypes.sizeof(capi.TimingsC)


def test_issue_roofline_reads_the_counters_of_the_traffic_record():
    """bench.py's roofline.traffic and roofline_issue come from ONE record of profiles/pmc_traffic.json (same key, same tag): a
    record that has instruction counters also names the profile they came from, and the texture addresser's busy cycles."""
    import json
    db = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    rec = db["cfg3:speckle:f32"]
    assert rec["tag"] and rec["hbm_bytes_per_launch"] > 0 and rec["valu_insts"] > 0
    assert os.path.exists(os.path.join(ROOT, "profiles", rec["tag"] + "_pmc.json"))
    sys.path.insert(0, ROOT)
    import bench
    r = bench.issue_roofline(rec, 10.0)
    assert r["source"] == rec["tag"] and r["bound"] in ("vector_issue", "scalar_issue", "texture_addresser")
    assert r["ta_floor_ms"] is not None and r["frac"] == max(r["vector_floor_ms"], r["scalar_floor_ms"], r["ta_floor_ms"]) / 10.0


def test_invalid_arguments_are_reported_not_fatal():
    g = scene.default_grid(4)
    with pytest.raises(capi.DmiError) as e:
        capi.FusionContext(scene.GridDesc((0, 4, 4), g.origin, g.spacing), scene.default_ray_potential(g))
    assert e.value.code == 1 and "cell_dims" in str(e.value)
    with pytest.raises(capi.DmiError) as e:      # filt.cxx:138-142
        capi.FusionContext(g, scene.RayPotential(0.0, 0.0, 0.1, 0.1))
    assert e.value.code == 1 and "rho" in str(e.value)


def test_no_gpu_means_loud_failure():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    g = scene.default_grid(4)
    with pytest.raises(capi.DmiError) as e:
        capi.FusionContext(g, scene.default_ray_potential(g))
    assert e.value.code == 2


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "cudadepthmapintegration_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in src and "tsdf_oracle.c\"" not in src


def test_accumulator_register_audit_covers_this_build():
    """fusion_tile.hip keeps its running sums in VGPRs the compiler only knows as clobbers (fusion_tile_acc.inc): a
    toolchain that allocates one of them would corrupt sums silently.  build() audits the gfx950 assembly whenever the
    kernel is recompiled; here: the audit record exists, is about the kernel's present source and flags (content digest), and saw every shape."""
    import json

    from cudadepthmapintegration_amd import build

    capi.load()
    marker = os.path.join(build.OBJ_DIR, "acc_audit.json")
    if not build.audit_is_current():
        build.run_accumulator_audit()           # e.g. a prebuilt .so copied without build/: audit now (hipcc -S, minutes)
    rec = json.load(open(marker))
    assert rec["violations"] == 0 and rec["instantiations"] >= 30 and rec["digest"]


def test_library_freshness_is_decided_by_content_not_timestamps():
    """A checkout or a copy to another machine (the GPU box) changes timestamps, not contents: the library built from
    these sources must not be rebuilt there (build.source_digest, libdmi_hip.so.digest)."""
    from cudadepthmapintegration_amd import build

    capi.load()
    assert os.path.exists(build.LIB_PATH + ".digest") and not build.needs_build()
    src = os.path.join(build.CSRC, "grid_post.hip")
    st = os.stat(src)
    try:
        os.utime(src, None)  # "newer than the library"
        assert not build.needs_build()
    finally:
        os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns))


def test_accumulator_register_audit_flags_a_violation():
    from cudadepthmapintegration_amd import build

    name = "_ZN3dmi12_GLOBAL__N_116fuse_tile_kernelIffLi8ELi1ELi1ELi6ELi8ELb0ELb0EEEvNS_8TileArgsE"
    ok = (f"\n{name}:\n\tv_add_f64 v[2:3], v[4:5], v[6:7]\n\t;;#ASMSTART\n\tv_add_f64 v[80:81], v[80:81], s[2:3]\n\t;;#ASMEND\n"
          f"\ts_endpgm\n.amdhsa_kernel {name}\n\t\t.amdhsa_next_free_vgpr 96\n.end_amdhsa_kernel\n")
    assert build.audit_accumulator_registers(ok) == (1, [])
    clobbered = ok.replace("v_add_f64 v[2:3], v[4:5], v[6:7]", "v_mov_b32_e32 v81, v3")
    n, bad = build.audit_accumulator_registers(clobbered)
    assert n == 1 and len(bad) == 1 and "v81" in bad[0]
    short = ok.replace("next_free_vgpr 96", "next_free_vgpr 88")
    assert len(build.audit_accumulator_registers(short)[1]) == 1
    # a compare whose SCC is read only after an EXEC-masked asm statement (s_and_saveexec_b64 writes SCC): flagged;
    # with a scalar instruction that defines SCC anew in between: clean
    masked = "\t;;#ASMSTART\n\ts_and_saveexec_b64 s[8:9], s[6:7]\n\tv_add_f64 v[80:81], v[80:81], s[2:3]\n\ts_mov_b64 exec, s[8:9]\n\t;;#ASMEND\n"
    carried = ok.replace("\ts_endpgm", "\ts_cmp_eq_u64 s[0:1], -1\n" + masked + "\ts_cbranch_scc1 .LBB0_1\n.LBB0_1:\n\ts_endpgm")
    n, bad = build.audit_accumulator_registers(carried)
    assert n == 1 and len(bad) == 1 and "SCC" in bad[0]
    fine = ok.replace("\ts_endpgm", masked + "\ts_cmp_eq_u64 s[0:1], -1\n\ts_cbranch_scc1 .LBB0_1\n.LBB0_1:\n\ts_endpgm")
    assert build.audit_accumulator_registers(fine) == (1, [])


def test_issue_roofline_takes_the_larger_floor():
    """bench.py's `roofline_issue`: vector floor from the pipes' busy quad-cycles (falling back to the instruction count),
    scalar floor from SALU alone, `frac` against the run's kernel time."""
    import bench

    assert bench.issue_roofline(None, 5.0) is None
    r = bench.issue_roofline({"valu_insts": 1024 * 2.4e6, "salu_insts": 256 * 2.4e6, "tag": "t"}, 8.0)
    assert r["bound"] == "vector_issue" and abs(r["vector_floor_ms"] - 4.0) < 1e-9 and abs(r["scalar_floor_ms"] - 1.0) < 1e-9
    assert abs(r["frac"] - 0.5) < 1e-9 and r["source"] == "t"
    r = bench.issue_roofline({"valu_insts": 1024 * 2.4e6, "valu_active_quad_cycles": 1024 * 3.0e6, "salu_insts": 256 * 2.4e6 * 6}, 8.0)
    assert abs(r["vector_floor_ms"] - 5.0) < 1e-9 and r["bound"] == "scalar_issue" and abs(r["frac"] - 0.75) < 1e-9


def test_build_accepts_the_library_by_digest_and_says_what_it_did():
    """build() on an up-to-date tree compiles nothing (the library's digest file names the present sources and flags,
    whatever the timestamps say) and leaves a record of that; the command-line tool is keyed on the same digest and is
    linked on demand, never as a side effect of loading the library."""
    import json
    from cudadepthmapintegration_amd import build as b
    path = b.build()
    assert os.path.exists(path) and not b.needs_build()
    rec = json.load(open(os.path.join(b.OBJ_DIR, "build_record.json")))
    assert rec["digest"] == b.source_digest() and rec["mode"].split(":")[0] in ("up_to_date", "rebuilt")
    cli = capi.cli_binary()
    assert os.path.exists(cli) and open(cli + ".digest").read().strip() == b.source_digest()
    # no offload-bundler temporaries next to the sources (they used to be committed by accident)
    assert not [f for f in os.listdir(b.CSRC) if ".so." in f and not f.endswith(".digest")]


def test_fp64_only_build_of_the_tiled_kernel_compiles():
    """-DDMI_TIER1=0 (pixels selected in fp64 only: the A/B build fusion_kernels.h advertises) still compiles: the window
    instantiations, which are tier-1 columns, are left out of such a build (fusion_tile.hip: launch_win)."""
    import shutil
    import subprocess
    import tempfile
    from cudadepthmapintegration_amd import build as b
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([hipcc] + b.COMMON_FLAGS + ["--offload-arch=gfx950", "-mllvm", "-disable-promote-alloca-to-vector", "-DDMI_FAST_BUILD",
                            "-DDMI_TIER1=0", "--cuda-device-only", "-c", os.path.join(b.CSRC, "fusion_tile.hip"), "-o", os.path.join(tmp, "t.o")],
                           capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_loading_the_package_starts_no_other_program():
    """A process whose GPU is already initialised (anything under `rocprofv3 --pmc`) must not start another program: the GPU
    boxes refuse it.  With the flag probe's record in place (build/llvm_flag_probe.json, written by the first import), importing
    the build module -- which every load of the library does -- runs nothing."""
    import subprocess as sp
    import sys

    code = (
        "import sys\n"
        "import cudadepthmapintegration_amd.build\n"  # (writes the record if it is missing)
        "seen = []\n"
        "sys.addaudithook(lambda ev, a: seen.append(ev) if ev in ('subprocess.Popen', 'os.exec', 'os.posix_spawn', 'os.system') else None)\n"
        "import importlib\n"
        "importlib.reload(cudadepthmapintegration_amd.build)\n"
        "print('SPAWNED' if seen else 'QUIET')\n"
    )
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = sp.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().endswith("QUIET"), out.stdout + out.stderr
