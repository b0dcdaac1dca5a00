"""CPU tests of the host-side mirror of the reference's operator interface (include/dmi_host.h,
cudadepthmapintegration_amd/csrc/host/): file formats, ReconstructionData semantics and the filter's
error behaviour.  No GPU: Update() must fail loudly instead of falling back to any CPU path."""
import os
import re

import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmi_[a-z0-9_]+)\s*\(", text)))


def test_host_header_symbols_are_exported_and_bound():
    import ctypes
    lib = ctypes.CDLL(capi.load()._name)
    names = _declared("dmi_host.h")
    assert names == sorted(capi.HOST_ABI_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dmi_host.h but not exported"


KRTD = """1152 0.5 640
0 1150 360
0 0 1

0.36 0.48 -0.8
-0.8 0.6 0
0.48 0.64 0.6

1.5 -2.25 3.125
0
"""


def test_read_krtd_file_layout(tmp_path):
    """Helper.h:105-168: 3 lines K, blank, 3 lines R, blank, T; RT = [R|T; 0 0 0 1]."""
    p = tmp_path / "cam.krtd"
    p.write_text(KRTD)
    ok, K, RT = capi.read_krtd_file(str(p))
    assert ok
    assert np.array_equal(K, [[1152, 0.5, 640], [0, 1150, 360], [0, 0, 1]])
    assert np.array_equal(RT, [[0.36, 0.48, -0.8, 1.5], [-0.8, 0.6, 0, -2.25], [0.48, 0.64, 0.6, 3.125], [0, 0, 0, 1]])
    ok, _, _ = capi.read_krtd_file(str(tmp_path / "missing.krtd"))
    assert not ok


def test_extract_all_file_path(tmp_path):
    """Helper.h:60-100: last space-separated token of each non-empty line, relative to the list's directory."""
    lst = tmp_path / "vtiList.txt"
    lst.write_text("0 frame_000.vti\n\n1 sub/frame_001.vti\nframe_002.vti\n")
    got = capi.extract_all_file_path(str(lst))
    assert got == [f"{tmp_path}/frame_000.vti", f"{tmp_path}/sub/frame_001.vti", f"{tmp_path}/frame_002.vti"]
    assert capi.extract_all_file_path(str(tmp_path / "nope.txt")) == []


def test_extract_all_file_path_line_endings_and_blanks(tmp_path):
    """Lists written on Windows (CRLF), entries followed by blanks, blank-only lines, no final newline: the entry is
    still the last blank-separated token.  (The reference keeps the CR in the name and turns a blank-only line into the
    directory itself -- both would only make it fail to open a file; here they read as intended.)"""
    lst = tmp_path / "list.txt"
    lst.write_bytes(b"0 a.vti\r\n1 b.vti   \r\n   \r\n\r\n2  c d.vti \n3 last.vti")
    got = capi.extract_all_file_path(str(lst))
    assert got == [f"{tmp_path}/a.vti", f"{tmp_path}/b.vti", f"{tmp_path}/d.vti", f"{tmp_path}/last.vti"]
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    assert capi.extract_all_file_path(str(empty)) == []


def test_read_krtd_file_tolerates_crlf_and_short_files(tmp_path):
    """CRLF files read the same; missing numbers and missing lines read as zeros (a failed operator>> in the
    reference, Helper.h:124-128), the last row of RT is always 0 0 0 1."""
    p = tmp_path / "crlf.krtd"
    p.write_bytes(KRTD.replace("\n", "\r\n").encode())
    ok, K, RT = capi.read_krtd_file(str(p))
    assert ok and K[0, 1] == 0.5 and RT[2, 3] == 3.125 and np.array_equal(RT[3], [0, 0, 0, 1])
    q = tmp_path / "short.krtd"
    q.write_text("1 2\n3\n")
    ok, K, RT = capi.read_krtd_file(str(q))
    assert ok and np.array_equal(K, [[1, 2, 0], [3, 0, 0], [0, 0, 0]])
    assert np.array_equal(RT, [[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1]])


def test_k3_to_k4_matches_oracle():
    """RD.cxx:192-212."""
    K3 = np.array([[1152.0, 0.25, 640.5], [0, 1150.0, 360.25], [0, 0, 1]])
    assert np.array_equal(capi.k3_to_k4(K3), oracle.k3_to_k4(K3))


def test_apply_depth_threshold_matches_oracle():
    """RD.cxx:138-167: strictly greater than the threshold => -1."""
    rng = np.random.default_rng(0)
    d = rng.uniform(0.5, 4.0, size=(12, 16))
    bc = rng.random((12, 16))
    bc[0, 0] = 0.5          # equal to the threshold: kept
    got, changed = capi.apply_depth_threshold(d, bc, 0.5)
    want = oracle.apply_depth_threshold(d, bc, 0.5).reshape(d.shape)
    assert np.array_equal(got, want) and changed == int((bc > 0.5).sum()) and got[0, 0] == d[0, 0]


VTI = """<?xml version="1.0"?>
<VTKFile type="ImageData" version="0.1" byte_order="LittleEndian">
  <ImageData WholeExtent="0 3 0 2 0 0" Origin="0 0 0" Spacing="1 1 1">
    <Piece Extent="0 3 0 2 0 0">
      <PointData Scalars="Depths">
        <DataArray type="Float64" Name="Depths" format="ascii">
          1 2 3 4 5 6 7 8 -1 10 11 12.5
        </DataArray>
        <DataArray type="Float64" Name="Best Cost Values" format="ascii">
          0.1 0.2 0.3 0.4 0.5 0.6 0.7 0.8 0.9 0.05 0.15 0.25
        </DataArray>
        <DataArray type="UInt8" Name="Color" NumberOfComponents="3" format="ascii">
          0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0
        </DataArray>
      </PointData>
    </Piece>
  </ImageData>
</VTKFile>
"""


def test_read_depth_map_ascii_vti(tmp_path):
    p = tmp_path / "d.vti"
    p.write_text(VTI)
    d, bc = capi.read_depth_map(str(p))
    assert d.shape == (3, 4) and d[2, 0] == -1 and d[2, 3] == 12.5 and bc[2, 1] == 0.05
    assert capi.read_depth_map(str(tmp_path / "none.vti")) is None


def _configured_filter(f, grid, rp):
    f.SetRayPotentialThickness(rp.thickness)
    f.SetRayPotentialRho(rp.rho)
    f.SetRayPotentialEta(rp.eta)
    f.SetRayPotentialDelta(rp.delta)
    f.SetGridMatrix(grid.grid_matrix)
    f.SetInputData([c + 1 for c in grid.cell_dims], grid.origin, grid.spacing)


def test_filter_error_behaviour_matches_request_data():
    """filt.cxx:114-118 and :138-142: RequestData returns 0 when a path is unset or rho == thickness == 0."""
    g = scene.default_grid(4)
    rp = scene.default_ray_potential(g)
    v = scene.make_views(1, 8, 6, seed=0)
    with capi.ReconstructionFilter() as f:
        _configured_filter(f, g, rp)
        f.AddView(v.depth[0], v.K4[0][:3, :3], v.RT4[0])
        assert f.GetNumberOfCells() == 64
        assert f.Update() == 0 and "inputs have not been set" in f.LastError()      # no paths yet
        f.SetFilePathKRTD("in-memory")
        assert f.Update() == 0
        f.SetFilePathVTI("in-memory")
        f.SetRayPotentialRho(0.0)
        f.SetRayPotentialThickness(0.0)
        assert f.Update() == 0 and "Rho or Thickness" in f.LastError()
        assert f.GetExecutionTime() == -1                                             # filt.cxx:101, never reached :148


def test_filter_without_gpu_fails_loudly():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    g = scene.default_grid(4)
    rp = scene.default_ray_potential(g)
    v = scene.make_views(1, 8, 6, seed=0)
    with capi.ReconstructionFilter() as f:
        _configured_filter(f, g, rp)
        f.SetFilePathKRTD("in-memory")
        f.SetFilePathVTI("in-memory")
        f.AddView(v.depth[0], v.K4[0][:3, :3], v.RT4[0])
        assert f.Update() == 0
        assert "HIP device" in f.LastError() or "hip" in f.LastError().lower()


def test_filter_list_files_missing_is_an_error(tmp_path):
    """filt.cxx:161-165."""
    g = scene.default_grid(4)
    rp = scene.default_ray_potential(g)
    with capi.ReconstructionFilter() as f:
        _configured_filter(f, g, rp)
        f.SetFilePathKRTD(str(tmp_path / "krtdList.txt"))
        f.SetFilePathVTI(str(tmp_path / "vtiList.txt"))
        assert f.Update() == 0 and "no enough vti files" in f.LastError()


MODES = [("ascii", False, "UInt32", False), ("binary", False, "UInt32", False), ("binary", True, "UInt32", False),
         ("binary", True, "UInt64", False), ("appended-raw", False, "UInt32", False), ("appended-raw", True, "UInt64", False),
         ("appended-base64", False, "UInt64", False), ("appended-base64", True, "UInt32", False),   # the writer's default
         ("appended-raw", True, "UInt32", True), ("binary", False, "UInt64", True)]


@pytest.mark.parametrize("mode,compress,header,big_endian", MODES)
def test_read_depth_map_every_vti_data_mode(tmp_path, mode, compress, header, big_endian):
    """RD.cxx:223-229 reads whatever vtkXMLImageDataWriter wrote: every data mode must give back the same arrays."""
    from vti_writer import write_vti
    rng = np.random.default_rng(7)
    W, H = 37, 23                     # 851 points: several compression blocks of 1000 bytes, a partial last one
    depths = rng.uniform(0.5, 9.0, size=(H, W))
    depths[rng.random((H, W)) < 0.2] = -1.0
    cost = rng.random((H, W))
    color = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    p = tmp_path / "d.vti"
    write_vti(str(p), {"Depths": depths, "Other": depths.astype(np.float32), "Best Cost Values": cost, "Color": color}, W, H,
              mode=mode, compress=compress, header=header, big_endian=big_endian, block=1000)
    d, bc = capi.read_depth_map(str(p))
    assert d.shape == (H, W) and np.array_equal(d, depths) and np.array_equal(bc, cost)
    assert np.array_equal(capi.read_depth_map_color(str(p)), color)


def test_read_depth_map_rejects_what_the_reference_cannot_use(tmp_path):
    from vti_writer import write_vti
    W, H = 5, 4
    # "Depths" as Float32: the reference's SafeDownCast to vtkDoubleArray gives NULL (cu:249-250); here an error
    write_vti(str(tmp_path / "f32.vti"), {"Depths": np.ones((H, W), dtype=np.float32)}, W, H, mode="binary")
    assert capi.read_depth_map(str(tmp_path / "f32.vti")) is None
    # truncated appended data
    write_vti(str(tmp_path / "ok.vti"), {"Depths": np.ones((H, W))}, W, H, mode="appended-raw", compress=True)
    raw = (tmp_path / "ok.vti").read_bytes()
    cut = raw.index(b"_") + 20
    (tmp_path / "cut.vti").write_bytes(raw[:cut])
    assert capi.read_depth_map(str(tmp_path / "ok.vti")) is not None
    assert capi.read_depth_map(str(tmp_path / "cut.vti")) is None
    # no depth array at all
    write_vti(str(tmp_path / "none.vti"), {"Best Cost Values": np.ones((H, W))}, W, H)
    assert capi.read_depth_map(str(tmp_path / "none.vti")) is None
