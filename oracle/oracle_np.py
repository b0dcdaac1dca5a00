"""numpy restatement of the reference's fusion arithmetic (small cases only).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED.  Written independently of
tsdf_oracle.c from the same reference lines, vectorised over voxels; used by
tests/test_oracle.py to cross-check the C oracle bit for bit.  numpy's
elementwise float64 multiply/add are separately rounded IEEE operations, i.e.
the same canonical arithmetic as the C build with -ffp-contract=off.

Citations: cu = Reconstruction/CudaReconstruction.cu of the reference.
"""
from __future__ import annotations

import numpy as np


def _round_half_away(u: np.ndarray) -> np.ndarray:
    """C/CUDA round(): nearest integer, halves away from zero (cu:187-188).

    u - trunc(u) is exact in fp64, so the comparison with 0.5 is exact too
    (floor(u + 0.5) would be wrong for u = 0.49999999999999994).
    """
    t = np.trunc(u)
    frac = np.abs(u - t)
    return t + np.where(frac >= 0.5, np.copysign(1.0, u), 0.0)


def _rows(M, x, y, z):
    """cu:88-93: ((m0*x + m1*y) + m2*z) + m3 for rows 0..2 of a 4x4."""
    out = []
    for r in range(3):
        out.append(((M[r, 0] * x + M[r, 1] * y) + M[r, 2] * z) + M[r, 3])
    return out


def fuse(cell_dims, origin, spacing, grid_matrix, thick, rho, eta, delta, depths, K4, RT4, init_grid=None):
    """Returns (grid f64 [nz,ny,nx], voxel_hits u32 [nz,ny,nx], map_hits u64 [n])."""
    nx, ny, nz = (int(c) for c in cell_dims)
    G = np.asarray(grid_matrix, dtype=np.float64).reshape(4, 4)
    depths = np.asarray(depths, dtype=np.float64)
    n, H, W = depths.shape
    K4 = np.asarray(K4, dtype=np.float64).reshape(n, 4, 4)
    RT4 = np.asarray(RT4, dtype=np.float64).reshape(n, 4, 4)

    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    # cu:78-83
    gx = origin[0] + (i + 0.5) * spacing[0]
    gy = origin[1] + (j + 0.5) * spacing[1]
    gz = origin[2] + (k + 0.5) * spacing[2]
    wx, wy, wz = _rows(G, gx, gy, gz)  # cu:168

    grid = np.zeros((nz, ny, nx)) if init_grid is None else np.array(init_grid, dtype=np.float64).reshape(nz, ny, nx)
    vhits = np.zeros((nz, ny, nx), dtype=np.uint32)
    mhits = np.zeros(n, dtype=np.uint64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for m in range(n):
            cx, cy, cz = _rows(RT4[m], wx, wy, wz)  # cu:172
            hx, hy, hz = _rows(K4[m], cx, cy, cz)  # cu:176
            ok = ~(hz < 0)  # cu:177
            u = hx / hz  # cu:183
            v = hy / hz  # cu:184
            ru = _round_half_away(u)
            rv = _round_half_away(v)
            lim = 2147483648.0
            ok &= (ru > -lim) & (ru < lim) & (rv > -lim) & (rv < lim)  # project rule, see tsdf_oracle.c
            ok &= (ru >= 0) & (rv >= 0) & (ru < W) & (rv < H)  # cu:192-197
            px = np.where(ok, ru, 0).astype(np.int64)
            py = np.where(ok, rv, 0).astype(np.int64)
            d = depths[m][H - 1 - py, px]  # cu:141-149
            ok &= ~(d == -1)  # cu:202
            diff = cz - d  # cu:108
            a = np.abs(diff)
            sign = np.where(diff != 0, np.trunc(diff / np.where(a == 0, 1.0, a)), 0.0)  # cu:112
            far = np.where(diff > 0, 0.0, -eta * rho)  # cu:115
            plateau = rho * sign  # cu:117
            ramp = (rho / thick) * diff if thick != 0 else np.full_like(diff, np.nan)  # cu:119
            val = np.where(a > delta, far, np.where(a > thick, plateau, ramp))
            grid = np.where(ok, grid + val, grid)  # cu:211
            vhits += ok.astype(np.uint32)
            mhits[m] = np.count_nonzero(ok)
    return grid, vhits, mhits


def color_mesh_np(points, colors, K4, RT4):
    """Independent numpy restatement of MeshColoration::ProcessColoration (Coloration/MeshColoration.cxx:98-199,
    Sources/ReconstructionData.cxx:92-116,169-182, Sources/Helper.h:174-187) for cross-checking the C oracle."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    col = np.asarray(colors, dtype=np.uint8)
    n, H, W, _ = col.shape
    nv = pts.shape[0]
    mean = np.zeros((nv, 3), dtype=np.uint8)
    median = np.zeros((nv, 3), dtype=np.uint8)
    count = np.zeros(nv, dtype=np.int32)
    for i in range(nv):
        x, y, z = (np.float64(v) for v in pts[i])
        lists = [[], [], []]
        for m in range(n):
            RT = np.asarray(RT4[m], dtype=np.float64).reshape(4, 4)
            K = np.asarray(K4[m], dtype=np.float64).reshape(4, 4)
            c = [((RT[r, 0] * x + RT[r, 1] * y) + RT[r, 2] * z) + RT[r, 3] for r in range(3)]   # TransformPoint
            d = [(K[r, 0] * c[0] + K[r, 1] * c[1]) + K[r, 2] * c[2] for r in range(3)]           # TransformVector
            with np.errstate(all="ignore"):
                u, v = np.float64(d[0]) / np.float64(d[2]), np.float64(d[1]) / np.float64(d[2])
            if not (np.isfinite(u) and np.isfinite(v)):
                continue
            ru = np.sign(u) * np.floor(np.abs(u) + 0.5)     # std::round: half away from zero
            rv = np.sign(v) * np.floor(np.abs(v) + 0.5)
            if not (abs(ru) < 2.0 ** 31 and abs(rv) < 2.0 ** 31):
                continue
            px, py = int(ru), int(rv)
            if px < 0 or py < 0 or px >= W or py >= H:
                continue
            rgb = col[m, H - 1 - py, px]
            for ch in range(3):
                lists[ch].append(int(rgb[ch]))
        if lists[0]:
            k = len(lists[0])
            count[i] = k
            for ch in range(3):
                mean[i, ch] = int(float(sum(lists[ch])) / float(k))
                s = sorted(lists[ch])
                median[i, ch] = int((s[k // 2] + s[k // 2 - 1]) / 2) if k % 2 == 0 else s[k // 2]
    return mean, median, count


def cell_to_point_np(cells):
    """vtkCellDataToPointData on an image grid (Reconstruction/main.cxx:151-155), written independently of
    oracle_cell_to_point: padded arrays and masks instead of per-point id lists; same offset order
    (vtkStructuredData::GetPointCells) and the same c = 0; c += w*v accumulation."""
    c = np.asarray(cells, dtype=np.float64)
    nz, ny, nx = c.shape
    pad = np.zeros((nz + 2, ny + 2, nx + 2))
    pad[1:-1, 1:-1, 1:-1] = c
    valid = np.zeros((nz + 2, ny + 2, nx + 2), dtype=bool)
    valid[1:-1, 1:-1, 1:-1] = True
    offsets = [(-1, 0, 0), (-1, -1, 0), (-1, -1, -1), (-1, 0, -1), (0, 0, 0), (0, -1, 0), (0, -1, -1), (0, 0, -1)]

    def shifted(a, dx, dy, dz):  # value of cell (i+dx, j+dy, k+dz) for every point (i, j, k)
        return a[1 + dz:nz + 2 + dz, 1 + dy:ny + 2 + dy, 1 + dx:nx + 2 + dx]

    count = np.zeros((nz + 1, ny + 1, nx + 1))
    for dx, dy, dz in offsets:
        count += shifted(valid, dx, dy, dz)
    w = 1.0 / count
    out = np.zeros((nz + 1, ny + 1, nx + 1))
    for dx, dy, dz in offsets:
        m = shifted(valid, dx, dy, dz)
        out = np.where(m, out + w * shifted(pad, dx, dy, dz), out)
    return out


def iso_active_cells_np(points, iso):
    """Independent restatement of oracle_iso_active_cells: eight shifted views of the point lattice, corner >= iso."""
    p = np.asarray(points, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        ge = p >= iso
    inside = sum(ge[dz:ge.shape[0] - 1 + dz, dy:ge.shape[1] - 1 + dy, dx:ge.shape[2] - 1 + dx].astype(np.int32)
                 for dz in (0, 1) for dy in (0, 1) for dx in (0, 1))
    return np.flatnonzero(((inside > 0) & (inside < 8)).reshape(-1)).astype(np.int64)
