"""Deterministic synthetic depth-map scenes (inputs for tests, smoke and bench).

The reference ships no sample data, so every input is generated:
a voxel cube [-1,1]^3, a sphere of radius 0.6 at the origin, pinhole cameras on
a ring / Fibonacci sphere of radius 3 looking at the origin, K = [[f,0,W/2],
[0,f,H/2],[0,0,1]] with f = 0.9 W, depth = camera-space z of the first ray /
sphere intersection (the quantity the reference compares against,
Reconstruction/CudaReconstruction.cu:207), -1 where the ray misses (the
reference's "no depth" sentinel, cu:202 and Sources/ReconstructionData.cxx:164).

Conventions reproduced from the reference:
  * depth tables are stored in vtkImageData order: row 0 is the BOTTOM image row,
    so image pixel (px, py) lives at index W*(H-1-py)+px (cu:141-149);
  * x_cam = R x_world + T, RT = [R|T; 0 0 0 1] (Sources/Helper.h:134-165);
  * K4 = 3x3 K in the top-left of a 4x4 identity (ReconstructionData.cxx:192-212).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class GridDesc:
    """Voxel grid = the cells of the reference's input vtkImageData (filt.cxx:121-126)."""
    cell_dims: tuple  # (nx, ny, nz) voxels; vtk point dims are these + 1
    origin: tuple
    spacing: tuple
    grid_matrix: np.ndarray = field(default_factory=lambda: np.eye(4))  # rows = gridVecX/Y/Z (main.cxx:345-359)

    @property
    def n_voxels(self) -> int:
        return int(self.cell_dims[0]) * int(self.cell_dims[1]) * int(self.cell_dims[2])


@dataclass
class RayPotential:
    """The four TSDF parameters (cu:60-63; README 'TSDF')."""
    thickness: float
    rho: float
    eta: float
    delta: float


@dataclass
class Views:
    """A batch of depth maps with their cameras, as ReconstructionData exposes them."""
    depth: np.ndarray  # [n, H, W] f64, vtk row order, -1 = no depth
    K4: np.ndarray  # [n, 4, 4]
    RT4: np.ndarray  # [n, 4, 4]
    best_cost: np.ndarray | None = None  # [n, H, W] f64 or None

    @property
    def n(self) -> int:
        return int(self.depth.shape[0])

    @property
    def width(self) -> int:
        return int(self.depth.shape[2])

    @property
    def height(self) -> int:
        return int(self.depth.shape[1])

    def subset(self, lo: int, hi: int) -> "Views":
        return Views(self.depth[lo:hi], self.K4[lo:hi], self.RT4[lo:hi],
                     None if self.best_cost is None else self.best_cost[lo:hi])


def default_grid(n: int | tuple, rotated: bool = False) -> GridDesc:
    """Cube [-1,1]^3 with n^3 (or nx,ny,nz) voxels."""
    dims = (n, n, n) if isinstance(n, int) else tuple(int(v) for v in n)
    spacing = tuple(2.0 / d for d in dims)
    G = np.eye(4)
    if rotated:
        # orthonormal rows (the CLI requires orthogonal gridVec*, main.cxx:363-382)
        a, b = 0.3, -0.2
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        G[:3, :3] = Rz @ Rx
    return GridDesc(dims, (-1.0, -1.0, -1.0), spacing, G)


def default_ray_potential(grid: GridDesc) -> RayPotential:
    """Same ratios as the reference's example command lines (main.cxx:102-103)."""
    s = float(max(grid.spacing))
    return RayPotential(thickness=2.5 * s, rho=0.8, eta=0.03, delta=10.0 * s)


def look_at_rt(cam_pos: np.ndarray, target=(0.0, 0.0, 0.0)) -> np.ndarray:
    """4x4 [R|T] with camera z forward, x right, y down (image rows grow downwards)."""
    c = np.asarray(cam_pos, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - c
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    if abs(fwd @ up) > 0.99:
        up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd])
    RT = np.eye(4)
    RT[:3, :3] = R
    RT[:3, 3] = -R @ c
    return RT


def camera_positions(n: int, radius: float = 3.0, layout: str = "sphere") -> np.ndarray:
    if layout == "ring":
        t = 2 * np.pi * (np.arange(n) + 0.25) / n
        return np.stack([radius * np.cos(t), radius * np.sin(t), 0.35 * radius * np.sin(2 * t)], axis=1)
    # Fibonacci sphere
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    theta = np.pi * (1 + 5 ** 0.5) * i
    return radius * np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def render_sphere_depth(K: np.ndarray, RT: np.ndarray, W: int, H: int, center=(0.0, 0.0, 0.0),
                        radius: float = 0.6, background: float | None = None) -> np.ndarray:
    """Camera-z depth of a sphere, float32-representable f64, vtk row order."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    px = np.arange(W, dtype=np.float64)
    py = np.arange(H, dtype=np.float64)
    dx = ((px - cx) / fx)[None, :]
    dy = ((py - cy) / fy)[:, None]
    # ray = t * (dx, dy, 1) in the camera frame; t is the camera-space z
    c = RT[:3, :3] @ np.asarray(center, dtype=np.float64) + RT[:3, 3]
    a = dx * dx + dy * dy + 1.0
    b = dx * c[0] + dy * c[1] + c[2]
    disc = b * b - a * (c @ c - radius * radius)
    hit = disc >= 0
    t = (b - np.sqrt(np.where(hit, disc, 0.0))) / a
    hit &= t > 0
    miss = -1.0 if background is None else float(background)
    img = np.where(hit, t, miss)
    img = img.astype(np.float32).astype(np.float64)  # lossless f32 storage on the device
    return img[::-1].copy()  # image row py -> vtk row H-1-py


def make_views(n: int, W: int, H: int, seed: int = 0, layout: str = "sphere", dense: bool = False,
               with_best_cost: bool = False, radius: float = 3.0, focal_scale: float = 0.9,
               dtype=np.float64, view_range: tuple | None = None) -> Views:
    """n cameras around the sphere scene.  dense=True adds a background at camera z = radius + 0.5
    so nearly every in-frustum voxel reaches the accumulate (hit rate ~100 % instead of ~30 %).
    view_range=(lo, hi) renders only views lo .. hi-1 of that same n-camera scene (a rank of a multi-GPU run renders
    its own share; view g is the same whoever renders it)."""
    K = np.eye(4)
    K[0, 0] = K[1, 1] = focal_scale * W
    K[0, 2] = W / 2.0
    K[1, 2] = H / 2.0
    pos = camera_positions(n, radius=radius, layout=layout)
    rng = np.random.default_rng(seed)
    # small deterministic jitter so no two cameras share exact symmetries (drawn for all n: the stream position of
    # view g's jitter does not depend on who asks)
    jitter = 0.02 * rng.standard_normal((n, 3))
    lo, hi = (0, n) if view_range is None else (int(view_range[0]), int(view_range[1]))
    cnt = hi - lo
    depth = np.empty((cnt, H, W), dtype=dtype)  # values are f32-representable either way
    K4 = np.empty((cnt, 4, 4))
    RT4 = np.empty((cnt, 4, 4))
    for m in range(lo, hi):
        RT4[m - lo] = look_at_rt(pos[m], target=jitter[m])
        K4[m - lo] = K
        depth[m - lo] = render_sphere_depth(K[:3, :3], RT4[m - lo], W, H,
                                            background=(radius + 0.5) if dense else None)
    best = rng.random((n, H, W))[lo:hi] if with_best_cost else None
    return Views(depth, K4, RT4, best)


# ---- a second geometry: cameras INSIDE the volume, looking outward at the walls of a room -----------------------------
ROOM_HALF = (0.92, 0.86, 0.78)  # half extents of the room (an axis-aligned box inside the grid cube [-1, 1]^3)


def room_camera_positions(n: int, seed: int = 0) -> np.ndarray:
    """n camera centres inside the room, up to 0.8 of the way from its centre to the walls (deterministic, view g the same
    whoever asks): some stand a hand's breadth from a wall, whose depth then is a tenth of the far corner's."""
    rng = np.random.default_rng([int(seed), 7])
    p = rng.uniform(-1.0, 1.0, size=(n, 3))
    return 0.8 * p * np.asarray(ROOM_HALF)


def render_room_depth(K: np.ndarray, RT: np.ndarray, W: int, H: int) -> np.ndarray:
    """Camera-z depth of the room's walls seen from inside (every ray leaves the box through one of six planes): the depth
    range within one image reaches an order of magnitude, walls are seen at every angle down to grazing.  f32-representable
    f64, vtk row order."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    px = np.arange(W, dtype=np.float64)
    py = np.arange(H, dtype=np.float64)
    dx = ((px - cx) / fx)[None, :]
    dy = ((py - cy) / fy)[:, None]
    R = RT[:3, :3]
    c = -R.T @ RT[:3, 3]                       # camera centre in the world
    # ray direction in the world per unit of camera z: R^T (dx, dy, 1)
    d = (R.T[:, 0][:, None, None] * dx[None] + R.T[:, 1][:, None, None] * dy[None] + R.T[:, 2][:, None, None])
    half = np.asarray(ROOM_HALF)[:, None, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        t_hi = (half - c[:, None, None]) / d
        t_lo = (-half - c[:, None, None]) / d
    t_exit = np.where(d > 0, t_hi, np.where(d < 0, t_lo, np.inf))
    t = t_exit.min(axis=0)                     # camera-space z of the wall along this pixel's ray
    img = t.astype(np.float32).astype(np.float64)
    return img[::-1].copy()


def make_room_views(n: int, W: int, H: int, seed: int = 0, view_range: tuple | None = None, focal_scale: float = 0.6) -> "Views":
    """n cameras inside the room looking outward in directions spread over the sphere (a wider lens than the sphere scene's:
    f = 0.6 W)."""
    K = np.eye(4)
    K[0, 0] = K[1, 1] = focal_scale * W
    K[0, 2] = W / 2.0
    K[1, 2] = H / 2.0
    pos = room_camera_positions(n, seed)
    dirs = camera_positions(n, radius=1.0)     # Fibonacci directions
    lo, hi = (0, n) if view_range is None else (int(view_range[0]), int(view_range[1]))
    depth = np.empty((hi - lo, H, W))
    K4 = np.empty((hi - lo, 4, 4))
    RT4 = np.empty((hi - lo, 4, 4))
    for m in range(lo, hi):
        RT4[m - lo] = look_at_rt(pos[m], target=pos[m] + dirs[m])
        K4[m - lo] = K
        depth[m - lo] = render_room_depth(K[:3, :3], RT4[m - lo], W, H)
    return Views(depth, K4, RT4)


# ---- real-world magnitudes: the same scene in a geo-referenced frame --------------------------------------------------
def to_world_frame(grid: GridDesc, ray: RayPotential, views: "Views", scale: float, offset, focal: float | None = None):
    """The scene as an SfM pipeline in a projected coordinate system hands it over: world' = scale * world + offset (metres,
    with eastings / northings of 1e5 .. 1e7 as offsets), camera coordinates scaled alike.  Grid origin and spacing, the ray
    potential's lengths and the depths scale; R stays, T' = scale * T - R offset (so |T'| ~ |offset| and c = R w + T' cancels
    seven digits, as in real data: Sources/ReconstructionData.cxx:192-221, Sources/Helper.h:134-165 read such matrices).
    focal: optionally replace fx = fy by this many pixels (depths are unchanged: it only zooms the image)."""
    off = np.asarray(offset, dtype=np.float64)
    g = GridDesc(grid.cell_dims, tuple(float(scale) * np.asarray(grid.origin, dtype=np.float64) + off),
                 tuple(float(scale) * np.asarray(grid.spacing, dtype=np.float64)), grid.grid_matrix.copy())
    if not np.allclose(grid.grid_matrix[:3, :3], np.eye(3)) or np.any(grid.grid_matrix[:3, 3] != 0):
        # a grid matrix G maps p = origin + (idx + 0.5) spacing to the world: w = G3 p + g3; keep w' = scale w + offset by scaling
        # p (origin, spacing above, without the offset) and moving the offset into the matrix's translation
        g = GridDesc(grid.cell_dims, tuple(float(scale) * np.asarray(grid.origin, dtype=np.float64)),
                     tuple(float(scale) * np.asarray(grid.spacing, dtype=np.float64)), grid.grid_matrix.copy())
        g.grid_matrix[:3, 3] = float(scale) * grid.grid_matrix[:3, 3] + off
    r = RayPotential(ray.thickness * scale, ray.rho, ray.eta, ray.delta * scale)
    RT = views.RT4.copy()
    for m in range(views.n):
        RT[m, :3, 3] = scale * views.RT4[m, :3, 3] - views.RT4[m, :3, :3] @ off
    K = views.K4.copy()
    if focal is not None:
        K[:, 0, 0] = K[:, 1, 1] = float(focal)
    d = views.depth
    valid = np.isfinite(d) & (d != -1.0)
    depth = np.where(valid, (d * scale).astype(np.float32).astype(np.float64), d)
    return g, r, Views(depth, K, RT, views.best_cost)


GEO_OFFSET = (448262.5, 5411932.25, 312.0)  # an easting / northing / height a UTM-projected survey would carry
GEO_SCALE = 10.0                            # the cube [-1, 1]^3 becomes 20 m wide: 512^3 voxels of 3.9 cm


def geo_grid(grid: GridDesc, ray: RayPotential):
    """Grid and ray potential of the `geo` scene kind: the caller's, moved into the same frame as its views."""
    empty = Views(np.zeros((0, 1, 1)), np.zeros((0, 4, 4)), np.zeros((0, 4, 4)))
    g, r, _ = to_world_frame(grid, ray, empty, GEO_SCALE, GEO_OFFSET)
    return g, r


# ---- scenes as a stereo pipeline hands them over: invalid speckle, noise, holes --------------------------------------
SCENE_KINDS = ("dense", "sparse", "speckle", "noisy", "room", "blobs", "geo")
SPECKLE_THRESHOLD = 0.9  # best cost ~ U[0, 1): "threshold chosen to kill ~10 % of pixels" 


def view_rng(seed: int, m: int, stream: int) -> np.random.Generator:
    """Generator of view m's own random content: the same whoever renders the view and in whatever chunks."""
    return np.random.default_rng([int(seed), int(m), int(stream)])


def make_scene_views(kind: str, n: int, W: int, H: int, seed: int = 0, view_range: tuple | None = None,
                     noise_sigma: float = 0.0, speckle: float = 1.0 - SPECKLE_THRESHOLD, holes: int = 6,
                     layout: str = "sphere") -> tuple["Views", float | None]:
    """The bench / test scenes by name; returns (views, best-cost threshold or None).
      dense    sphere + background plane, every pixel holds a depth (hit rate ~100 %)
      sparse   sphere only, -1 where a ray misses it
      speckle  dense + "Best Cost Values" ~ U[0, 1) per pixel; with the returned threshold (0.9) the reference's
               ApplyDepthThresholdFilter (RD.cxx:138-167, called at cu:348) turns ~10 % of the pixels, scattered at
               random, into the -1 sentinel -- SURVEY.md 8d's scene, what the filter sees on real stereo output
      noisy    speckle + depth noise (every depth += N(0, noise_sigma), in scene units: the bench passes one voxel's
               spacing) + `holes` discs of 8-40 pixels radius without depth per view
      room     a second geometry: cameras INSIDE the grid looking outward at the walls of a room (make_room_views) + the speckle
      geo      speckle, geo-referenced: world' = 10 world + (448262.5, 5411932.25, 312) -- a 20 m cube of 3.9 cm voxels somewhere in a
               UTM zone, cameras 30 m away, every translation ~5e6 (use geo_grid for the grid and the ray potential)
      blobs    dense + REGIONAL holes: discs of 8-40 pixels radius without depth, as many as cover about `speckle` of the image
               (what best-cost filtering leaves of real stereo output: whole patches, not salt and pepper); no best-cost values
    Depths stay f32-representable (the device keeps them as f32 without changing a bit); best cost is f64 as the
    reference's array is.  Views lo .. hi-1 of the n-camera scene when view_range is given."""
    if kind not in SCENE_KINDS:
        raise ValueError(f"scene kind {kind!r}: one of {SCENE_KINDS}")
    if kind == "geo":
        # the speckle scene in a geo-referenced frame (to_world_frame with GEO_SCALE / GEO_OFFSET): the caller's grid and ray
        # potential must be moved alike (geo_grid)
        v, thr = make_scene_views("speckle", n, W, H, seed=seed, view_range=view_range, noise_sigma=noise_sigma, speckle=speckle,
                                  holes=holes, layout=layout)
        unit = default_grid(8)
        _, _, v = to_world_frame(unit, default_ray_potential(unit), v, GEO_SCALE, GEO_OFFSET)
        return v, thr
    if kind == "room":
        # room: cameras inside the grid looking outward at the walls of a box (depths over an order of magnitude, grazing
        # walls, voxels behind every camera) + the same 10 % speckle as `speckle`
        base = make_room_views(n, W, H, seed=seed, view_range=view_range)
        lo, hi = (0, n) if view_range is None else (int(view_range[0]), int(view_range[1]))
        best = np.empty((hi - lo, H, W), dtype=np.float64)
        for m in range(lo, hi):
            best[m - lo] = view_rng(seed, m, 1).random((H, W), dtype=np.float32)
        return Views(base.depth, base.K4, base.RT4, best), 1.0 - float(speckle)
    base = make_views(n, W, H, seed=seed, layout=layout, dense=(kind != "sparse"), view_range=view_range)
    if kind in ("dense", "sparse"):
        return base, None
    if kind == "blobs":
        lo, hi = (0, n) if view_range is None else (int(view_range[0]), int(view_range[1]))
        yy, xx = np.ogrid[0:H, 0:W]
        n_discs = max(1, int(round(float(speckle) * W * H / 2000.0)))  # a disc of radius U[8, 41) covers ~2000 pixels on average
        for m in range(lo, hi):
            rng = view_rng(seed, m, 3)
            d = base.depth[m - lo]
            for _ in range(n_discs):
                cx, cy, r = rng.integers(0, W), rng.integers(0, H), rng.integers(8, 41)
                y0, y1, x0, x1 = max(0, cy - r), min(H, cy + r + 1), max(0, cx - r), min(W, cx + r + 1)
                sub = d[y0:y1, x0:x1]
                sub[(xx[:, x0:x1] - cx) ** 2 + (yy[y0:y1] - cy) ** 2 <= r * r] = -1.0
        return base, None
    lo, hi = (0, n) if view_range is None else (int(view_range[0]), int(view_range[1]))
    best = np.empty((hi - lo, H, W), dtype=np.float64)
    for m in range(lo, hi):
        best[m - lo] = view_rng(seed, m, 1).random((H, W), dtype=np.float32)
        if kind == "noisy":
            d = base.depth[m - lo]
            rng = view_rng(seed, m, 2)
            valid = d != -1.0
            d += np.where(valid, noise_sigma * rng.standard_normal((H, W), dtype=np.float32), 0.0)
            yy, xx = np.ogrid[0:H, 0:W]
            for _ in range(holes):
                cx, cy, r = rng.integers(0, W), rng.integers(0, H), rng.integers(8, 41)
                d[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = -1.0
            base.depth[m - lo] = d.astype(np.float32).astype(np.float64)
    # the speckle fraction is what the threshold leaves above it
    return Views(base.depth, base.K4, base.RT4, best), 1.0 - float(speckle)


# ---- file forms of a view (what the reference's filter reads: Sources/Helper.h:105-168, RD.cxx:223-229) ----
def write_krtd(path: str, K3: np.ndarray, RT4: np.ndarray) -> None:
    """.krtd text: 3 lines K, blank, 3 lines R, blank, 1 line T (%.17g round-trips every double)."""
    f = lambda row: " ".join("%.17g" % float(v) for v in row)
    lines = [f(K3[i]) for i in range(3)] + [""] + [f(RT4[i, :3]) for i in range(3)] + ["", f(RT4[:3, 3]), "0"]
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def write_vti_ascii(path: str, depth: np.ndarray, best_cost: np.ndarray | None = None,
                    color: np.ndarray | None = None) -> None:
    """Minimal ascii VTK XML ImageData with the point arrays the path reads ("Depths", "Best Cost Values", "Color")."""
    H, W = depth.shape
    def arr(name, a):
        vals = " ".join("%.17g" % float(v) for v in np.asarray(a, dtype=np.float64).reshape(-1))
        return f'        <DataArray type="Float64" Name="{name}" format="ascii">\n          {vals}\n        </DataArray>\n'
    body = arr("Depths", depth) + (arr("Best Cost Values", best_cost) if best_cost is not None else "")
    if color is not None:
        vals = " ".join(str(int(v)) for v in np.asarray(color, dtype=np.uint8).reshape(-1))
        body += ('        <DataArray type="UInt8" Name="Color" NumberOfComponents="3" format="ascii">\n'
                 f'          {vals}\n        </DataArray>\n')
    with open(path, "w") as fh:
        fh.write('<?xml version="1.0"?>\n<VTKFile type="ImageData" version="0.1" byte_order="LittleEndian">\n'
                 f'  <ImageData WholeExtent="0 {W - 1} 0 {H - 1} 0 0" Origin="0 0 0" Spacing="1 1 1">\n'
                 f'    <Piece Extent="0 {W - 1} 0 {H - 1} 0 0">\n      <PointData Scalars="Depths">\n{body}'
                 '      </PointData>\n    </Piece>\n  </ImageData>\n</VTKFile>\n')


def write_view_files(directory: str, views: "Views", colors: np.ndarray | None = None):
    """Writes frame_XXXX.vti / .krtd plus vtiList.txt / krtdList.txt; returns the two list paths."""
    import os
    vti, krtd = [], []
    for m in range(views.n):
        v, k = f"frame_{m:04d}.vti", f"frame_{m:04d}.krtd"
        write_vti_ascii(os.path.join(directory, v), views.depth[m], None if views.best_cost is None else views.best_cost[m],
                        None if colors is None else colors[m])
        write_krtd(os.path.join(directory, k), views.K4[m][:3, :3], views.RT4[m])
        vti.append(v)
        krtd.append(k)
    lv, lk = os.path.join(directory, "vtiList.txt"), os.path.join(directory, "krtdList.txt")
    with open(lv, "w") as fh:
        fh.write("".join(f"{i} {p}\n" for i, p in enumerate(vti)))
    with open(lk, "w") as fh:
        fh.write("".join(f"{i} {p}\n" for i, p in enumerate(krtd)))
    return lv, lk


def make_colors(n: int, W: int, H: int, seed: int = 0) -> np.ndarray:
    """Synthetic "Color" arrays [n, H, W, 3] u8 (vtk row order): smooth gradients plus noise, different per view."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.empty((n, H, W, 3), dtype=np.uint8)
    for m in range(n):
        base = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy + 37 * m) % 256)], axis=-1)
        noise = rng.integers(-20, 21, size=(H, W, 3))
        out[m] = np.clip(base + noise, 0, 255).astype(np.uint8)
    return out


def make_mesh_points(n: int, seed: int = 0, radius: float = 0.6) -> np.ndarray:
    """Vertices of the synthetic scene's surface (the sphere) plus a share of points elsewhere in and around the
    grid (some outside every frustum), as a mesh extracted from the TSDF would have."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = radius * d
    k = n // 4
    pts[:k] = rng.uniform(-1.6, 1.6, size=(k, 3))
    return pts


def morton_order(points: np.ndarray, bits: int = 10) -> np.ndarray:
    """Indices that sort points along a Z-order curve of their bounding box: neighbours in space become neighbours in
    the array, as the vertices a marching-cubes sweep emits are (the coloration bench's "mesh-ordered" vertex set)."""
    p = np.asarray(points, dtype=np.float64)
    lo, hi = p.min(axis=0), p.max(axis=0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-300) * ((1 << bits) - 1)).astype(np.uint64), 0, (1 << bits) - 1)
    key = np.zeros(len(p), dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            key |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(key, kind="stable")
