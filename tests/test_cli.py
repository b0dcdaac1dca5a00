"""The `Reconstruction` command line (Reconstruction/main.cxx:216-343 -> csrc/host/recon_cli.cpp): flags, defaults,
validation and the derived grid, through the C binding (no GPU); the tool end to end on a GPU box."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from cudadepthmapintegration_amd import capi, scene

BASE = ["Reconstruction", "--gridOrigin", "-2.29", "-2.24", "-2.2", "--gridEnd", "1.19", "1.67", "1.22", "--dataFolder", "data",
        "--outputGridFilename", "out.vts", "--outputMeshFilename", "mesh.vtp"]


def test_example_command_line_of_the_reference():
    """rmain:93-94's first example, with --gridEnd in place of --gridSpacing (both with --gridDims is an error, rmain:257)."""
    o, text = capi.cli_read_arguments(BASE + ["--rayThick", "0.08", "--rayRho", "0.8", "--rayEta", "0.03", "--rayDelta", "0.3",
                                              "--threshBestCost", "0.3", "--gridDims", "100", "100", "100",
                                              "--gridVecX", "1", "0", "0", "--gridVecY", "0", "1", "0", "--gridVecZ", "0", "0", "1"])
    assert o is not None, text
    assert list(o.grid_dims) == [100, 100, 100]
    assert np.allclose(list(o.grid_spacing), [(1.19 + 2.29) / 100, (1.67 + 2.24) / 100, (1.22 + 2.2) / 100], rtol=0, atol=1e-15)
    assert list(o.grid_origin) == [-2.29, -2.24, -2.2]
    assert (o.ray_thick, o.ray_rho, o.ray_eta, o.ray_delta, o.thresh_best_cost, o.contour) == (0.08, 0.8, 0.03, 0.3, 0.3, 1.0)
    assert np.array_equal(np.array(o.grid_matrix).reshape(4, 4), np.eye(4))
    assert (o.verbose, o.summary, o.force_cubic_voxel) == (0, 0, 0)


def test_defaults_and_derived_grid():
    # the defaults themselves violate rayDelta >= rayThick (0.3 < 2, rmain:272-278): the reference rejects its own defaults
    o, text = capi.cli_read_arguments(BASE + ["--gridDims", "50"])
    assert o is None and "Error arguments." in text
    args = BASE + ["--rayThick", "0.1"]
    o, text = capi.cli_read_arguments(args + ["--gridDims", "50"])                        # one value stands for three (rmain:265)
    assert o is not None and list(o.grid_dims) == [50, 50, 50], text
    assert (o.ray_thick, o.ray_rho, o.ray_eta, o.ray_delta, o.thresh_best_cost, o.contour) == (0.1, 0.8, 0.03, 0.3, 0.14, 1.0)
    o, text = capi.cli_read_arguments(args + ["--gridSpacing", "0.5", "0.25", "1.0"])     # dimensions from the spacing, truncated
    assert o is not None, text
    assert list(o.grid_dims) == [int(3.48 / 0.5), int(3.91 / 0.25), int(3.42 / 1.0)]
    o, _ = capi.cli_read_arguments(args + ["--gridSpacing", "0.5", "0.25", "1.0", "--forceCubicVoxel", "--verbose", "--summary"])
    assert list(o.grid_spacing) == [0.25, 0.25, 0.25] and list(o.grid_dims) == [6, 15, 3]   # dims keep their value (rmain:337-344)
    assert (o.verbose, o.summary, o.force_cubic_voxel) == (1, 1, 1)
    o, _ = capi.cli_read_arguments(args + ["--gridDims", "10", "--gridVecX", "0", "1", "0", "--gridVecY", "-1", "0", "0"])
    m = np.array(o.grid_matrix).reshape(4, 4)
    assert np.array_equal(m[0, :3], [0, 1, 0]) and np.array_equal(m[1, :3], [-1, 0, 0]) and np.array_equal(m[2, :3], [0, 0, 1])
    assert np.array_equal(m[3], [0, 0, 0, 1]) and np.array_equal(m[:3, 3], [0, 0, 0])


@pytest.mark.parametrize("extra, needle", [
    (["--gridDims", "10", "--gridSpacing", "0.1", "0.1", "0.1"], "Spacing and dimensions can't be both set"),
    (["--gridDims", "10", "--rayDelta", "0.05"], "Error arguments."),                         # delta below thick
    (["--gridDims", "10", "--rayEta", "1.5"], "Error arguments."),
    (["--gridDims", "10", "--outputGridFilename", "out.vti"], "Bad output extension"),
    (["--gridDims", "10", "--outputMeshFilename", "mesh.obj"], "Bad output extension"),
    (["--gridDims", "10", "--gridVecX", "1", "1", "0"], "not orthogonals"),
    (["--nonsense", "--gridDims", "10"], "Unknown argument"),
    (["--gridDims", "ten"], "Bad value"),
    (["--gridDims", "10", "20"], "three values"),
    ([], "one of --gridDims"),
    (["--help"], "dmi_reconstruction"),
])
def test_rejected_command_lines(extra, needle):
    o, text = capi.cli_read_arguments(BASE + ["--rayThick", "0.1"] + extra)
    assert o is None and needle in text, text


def test_missing_required_names():
    args = [a for a in BASE if a not in ("--outputMeshFilename", "mesh.vtp")] + ["--rayThick", "0.1", "--gridDims", "10"]
    o, text = capi.cli_read_arguments(args)
    assert o is None and "Error arguments." in text
    o, text = capi.cli_read_arguments([a for a in BASE if a not in ("--gridEnd", "1.19", "1.67", "1.22")] +
                                      ["--rayThick", "0.1", "--gridDims", "10"])
    assert o is None and "--gridEnd" in text       # undefined behaviour in the reference (rmain:311), an error here


def test_cli_binary_is_built_and_prints_help():
    exe = capi.cli_binary()
    capi.load()
    assert os.path.exists(exe), exe
    r = subprocess.run([exe, "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "--gridDims" in r.stderr and "--outputGridFilename" in r.stderr


def _read_vts(path):
    raw = open(path, "rb").read()
    head, _, tail = raw.partition(b'<AppendedData encoding="raw">\n   _')
    text = head.decode()
    ext = [int(v) for v in text.split('WholeExtent="')[1].split('"')[0].split()]
    n_cells = ext[1] * ext[3] * ext[5]
    n_points = (ext[1] + 1) * (ext[3] + 1) * (ext[5] + 1)
    (nb,) = struct.unpack_from("<Q", tail, 0)
    assert nb == 8 * n_cells
    cells = np.frombuffer(tail, dtype=np.float64, count=n_cells, offset=8)
    (nb2,) = struct.unpack_from("<Q", tail, 8 + nb)
    assert nb2 == 24 * n_points and f'offset="{8 + nb}"' in text
    points = np.frombuffer(tail, dtype=np.float64, count=3 * n_points, offset=16 + nb).reshape(-1, 3)
    return ext, cells, points


def _read_mha(path):
    raw = open(path, "rb").read()
    head, _, data = raw.partition(b"ElementDataFile = LOCAL\n")
    fields = dict(line.split(" = ", 1) for line in head.decode().strip().splitlines())
    assert fields["CompressedData"] == "True" and int(fields["CompressedDataSize"]) == len(data) and fields["ElementType"] == "MET_DOUBLE"
    dims = [int(v) for v in fields["DimSize"].split()]
    return dims, fields, np.frombuffer(zlib.decompress(data), dtype=np.float64)


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path):
    """The tool on a data folder of .vti / .krtd files: the .vts volume holds the oracle's fusion bit for bit (cell data)
    and the image's points under the grid matrix, meta_image_volume.mha the cell -> point pass of it, summary.txt exists;
    no mesh is written."""
    from oracle import oracle
    from helpers import bits_equal, oracle_params_from_scene
    grid = scene.default_grid((24, 20, 16), rotated=True)
    rp = scene.default_ray_potential(grid)
    views = scene.make_views(5, 48, 36, seed=4, dense=True, with_best_cost=True)
    data = tmp_path / "data"
    data.mkdir()
    lv, lk = scene.write_view_files(str(data), views)
    gm = np.asarray(grid.grid_matrix).reshape(4, 4)
    end = [grid.origin[a] + (grid.cell_dims[a] + 1) * grid.spacing[a] for a in range(3)]
    args = [capi.cli_binary(), "--dataFolder", str(data), "--depthMapFile", os.path.basename(lv), "--KRTFile", os.path.basename(lk),
            "--gridDims"] + [str(c + 1) for c in grid.cell_dims] + ["--gridOrigin"] + [repr(float(v)) for v in grid.origin] + \
           ["--gridEnd"] + [repr(float(v)) for v in end] + ["--gridVecX"] + [repr(float(v)) for v in gm[0, :3]] + \
           ["--gridVecY"] + [repr(float(v)) for v in gm[1, :3]] + ["--gridVecZ"] + [repr(float(v)) for v in gm[2, :3]] + \
           ["--rayThick", repr(rp.thickness), "--rayRho", repr(rp.rho), "--rayEta", repr(rp.eta), "--rayDelta", repr(rp.delta),
            "--threshBestCost", "0.7", "--outputGridFilename", str(tmp_path / "volume.vts"), "--outputMeshFilename",
            str(tmp_path / "mesh.vtp"), "--summary", "--verbose"]
    r = subprocess.run(args, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "---START---" in r.stdout and "---END---" in r.stdout
    o, _ = capi.cli_read_arguments(args)
    g2 = scene.GridDesc(tuple(int(d) - 1 for d in o.grid_dims), tuple(o.grid_origin), tuple(o.grid_spacing), np.array(o.grid_matrix).reshape(4, 4))
    d = oracle.apply_depth_threshold(views.depth, views.best_cost, 0.7).reshape(views.depth.shape)
    want, _, _ = oracle.fuse(oracle_params_from_scene(g2, rp, views), d, views.K4, views.RT4, n_threads=oracle.max_threads())
    ext, cells, points = _read_vts(str(tmp_path / "volume.vts"))
    assert ext == [0, o.grid_dims[0] - 1, 0, o.grid_dims[1] - 1, 0, o.grid_dims[2] - 1]
    assert bits_equal(cells.reshape(want.shape), want)
    i, j, k = np.meshgrid(np.arange(o.grid_dims[0]), np.arange(o.grid_dims[1]), np.arange(o.grid_dims[2]), indexing="ij")
    p = np.stack([o.grid_origin[0] + i * o.grid_spacing[0], o.grid_origin[1] + j * o.grid_spacing[1], o.grid_origin[2] + k * o.grid_spacing[2]], -1)
    expect = (p @ np.array(o.grid_matrix).reshape(4, 4)[:3, :3].T).transpose(2, 1, 0, 3).reshape(-1, 3)
    assert np.allclose(points, expect, rtol=0, atol=1e-12)
    dims, fields, pts = _read_mha(str(tmp_path / "meta_image_volume.mha"))
    assert dims == list(o.grid_dims)
    assert bits_equal(pts.reshape(dims[2], dims[1], dims[0]), oracle.cell_to_point(want))
    summary = open(data / "summary.txt").read()
    assert "reconstruction" in summary
    # no mesh: said so whatever --verbose is, with the iso-value pre-pass's count of the cells a contour filter would visit
    assert not os.path.exists(tmp_path / "mesh.vtp")
    assert "is NOT written" in r.stdout + r.stderr
    n_active = oracle.iso_active_cells(oracle.cell_to_point(want), float(o.contour)).size
    assert f"cells straddling the value  {n_active} " in summary and f" {n_active} of " in r.stdout + r.stderr
