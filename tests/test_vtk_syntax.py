"""A syntax gate for the VTK binding (vtk/vtkCudaReconstructionFilter.{h,cxx}), the drop-in replacement of the reference's
Reconstruction/vtkCudaReconstructionFilter.{h,cxx} (filt.h:48-120, filt.cxx:96-179).  The image has no VTK, so the two files
are parsed and type-checked by `g++ -fsyntax-only` against tests/cpp/vtk_decl/: hand-written minimal DECLARATIONS of the VTK
classes and macros they name (labelled NOT VTK in every header).  Green means: well-formed C++ whose every call matches a
declared member with the documented arity and argument types, and whose use of dmi::host::FusionDriver matches
csrc/host/recon_host.h.  It does NOT mean the binding builds or behaves against a real VTK."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = os.path.join(ROOT, "tests", "cpp", "vtk_decl")
FLAGS = ["-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", DECL,
         "-I", os.path.join(ROOT, "cudadepthmapintegration_amd", "csrc", "host"), "-I", os.path.join(ROOT, "include"),
         "-I", os.path.join(ROOT, "vtk")]


def _gxx():
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++ in this image")
    return gxx


def test_vtk_binding_parses_and_type_checks():
    r = subprocess.run([_gxx()] + FLAGS + [os.path.join(ROOT, "vtk", "vtkCudaReconstructionFilter.cxx")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_the_gate_rejects_a_call_the_declarations_do_not_have(tmp_path):
    """Negative control: the declarations are not permissive catch-alls -- a member VTK does not have fails the check."""
    src = tmp_path / "bad.cxx"
    src.write_text('#include "vtkCudaReconstructionFilter.h"\n#include "vtkImageData.h"\n'
                   "void f(vtkCudaReconstructionFilter* x, vtkImageData* g) { x->SetRayPotentialRho(1.0); g->NoSuchMember(); }\n")
    r = subprocess.run([_gxx()] + FLAGS + [str(src)], capture_output=True, text=True)
    assert r.returncode != 0 and "NoSuchMember" in r.stderr


def test_declaration_headers_say_what_they_are():
    for name in sorted(os.listdir(DECL)):
        with open(os.path.join(DECL, name)) as fh:
            head = fh.read(400)
        assert "NOT" in head, name
