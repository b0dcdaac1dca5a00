/*
 * tsdf_oracle.c -- CPU restatement of the reference's TSDF depth-map fusion path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cudadepthmapintegration_amd/ may
 * import, link or execute this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed
 * CPU baseline -- never as the product path.
 *
 * PARITY UNPINNED.  The reference (bastienjacquet/CudaDepthMapIntegration)
 * ships no tests, golden vectors or fixtures for this path, and its only
 * implementation (Reconstruction/CudaReconstruction.cu) needs nvcc, the CUDA
 * runtime and VTK, none of which exist in the build image, so it cannot be
 * built or run here.  This file is a by-hand restatement of the reference's
 * arithmetic, statement by statement, in plain C (fp64, no FMA contraction:
 * build with -ffp-contract=off).  Every function cites the reference lines
 * it follows.  An independently written numpy restatement (oracle_np.py)
 * must agree with it bit for bit (tests/test_oracle.py); that is a
 * cross-check, not a pin.
 *
 * All file:line citations are relative to the reference repository root:
 *   cu       = Reconstruction/CudaReconstruction.cu
 *   filt.cxx = Reconstruction/vtkCudaReconstructionFilter.cxx
 *   RD.cxx   = Sources/ReconstructionData.cxx
 *
 * Behaviour defined by this project (the reference leaves it to the
 * platform's double->int conversion, cu:187-188):
 *   a projected coordinate whose round() is NaN, infinite or outside
 *   (-2^31, 2^31) is OUT of the depth map.  CUDA saturates such conversions
 *   (NaN -> 0), x86 returns INT_MIN; both agree with this rule for every
 *   finite in-range value, and differ from each other only for NaN (which
 *   needs h.x == 0 and h.z == 0 exactly, or a non-finite input).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* The reference's __constant__ block (cu:55-63) plus the launch geometry
 * source ch_gridDims (cu:64).  point_dims are vtkImageData POINT dimensions
 * (filt.cxx:123-124); the voxel (cell) grid is point_dims - 1 per axis
 * (cu:128-129, cu:330-331). */
typedef struct oracle_params {
  double grid_matrix[16]; /* c_gridMatrix, row-major 4x4 (cu:220-230) */
  double grid_orig[3];    /* c_gridOrig */
  int32_t point_dims[3];  /* c_gridDims */
  double grid_spacing[3]; /* c_gridSpacing */
  int32_t depth_dims[2];  /* c_depthMapDims: {W, H} */
  double thick;           /* c_rayPotentialThick */
  double rho;             /* c_rayPotentialRho */
  double eta;             /* c_rayPotentialEta */
  double delta;           /* c_rayPotentialDelta */
} oracle_params;

/* cu:78-83 computeVoxelCenter */
static void voxel_center(const oracle_params *p, const int idx[3], double out[3])
{
  out[0] = p->grid_orig[0] + (idx[0] + 0.5) * p->grid_spacing[0];
  out[1] = p->grid_orig[1] + (idx[1] + 0.5) * p->grid_spacing[1];
  out[2] = p->grid_orig[2] + (idx[2] + 0.5) * p->grid_spacing[2];
}

/* cu:88-93 transformFrom4Matrix: rows 0..2 of a row-major 4x4 times [pt,1],
 * evaluated left to right exactly as the C expression in the reference. */
static void transform4(const double M[16], const double pt[3], double out[3])
{
  out[0] = M[0 * 4 + 0] * pt[0] + M[0 * 4 + 1] * pt[1] + M[0 * 4 + 2] * pt[2] + M[0 * 4 + 3];
  out[1] = M[1 * 4 + 0] * pt[0] + M[1 * 4 + 1] * pt[1] + M[1 * 4 + 2] * pt[2] + M[1 * 4 + 3];
  out[2] = M[2 * 4 + 0] * pt[0] + M[2 * 4 + 1] * pt[1] + M[2 * 4 + 2] * pt[2] + M[2 * 4 + 3];
}

/* cu:105-120 rayPotential<double>.  `sign` is an int in the reference
 * (cu:112): diff/|diff| is exactly +-1.0 for finite non-zero diff. */
static double ray_potential(const oracle_params *p, double real_distance, double depth_map_distance)
{
  double diff = real_distance - depth_map_distance;
  double absolute_diff = fabs(diff);
  int sign = diff != 0 ? (int)(diff / absolute_diff) : 0;
  double res;
  if (absolute_diff > p->delta)
    res = diff > 0 ? 0 : -p->eta * p->rho;
  else if (absolute_diff > p->thick)
    res = p->rho * sign;
  else
    res = (p->rho / p->thick) * diff;
  return res;
}

/* double -> pixel index, cu:187-188 `pixel = round(u)`, with the out-of-range
 * rule stated in the header.  Returns 0 when the value cannot be a pixel. */
static int to_pixel(double u, int *px)
{
  double r = round(u); /* half away from zero, as CUDA's and C's round() */
  if (!(r > -2147483648.0 && r < 2147483648.0))
    return 0;
  *px = (int)r;
  return 1;
}

/* One thread of depthMapKernel (cu:158-212) for voxel (i,j,k) and one depth
 * map.  Returns 1 and writes the increment when the thread reaches cu:211. */
static int project_one(const oracle_params *p, const double *depths, const double K[16],
                       const double RT[16], int i, int j, int k, double *increment)
{
  int voxel_index[3] = {i, j, k}; /* cu:163 */
  double center_grid[3], center_world[3], center_cam[3], homogen[3];
  voxel_center(p, voxel_index, center_grid);             /* cu:166 */
  transform4(p->grid_matrix, center_grid, center_world); /* cu:168 */
  transform4(RT, center_world, center_cam);              /* cu:172 */
  transform4(K, center_cam, homogen);                    /* cu:176 */
  if (homogen[2] < 0)                                    /* cu:177 */
    return 0;
  double u = homogen[0] / homogen[2]; /* cu:183 */
  double v = homogen[1] / homogen[2]; /* cu:184 */
  int px, py;
  if (!to_pixel(u, &px) || !to_pixel(v, &py)) /* cu:187-188 */
    return 0;
  if (px < 0 || py < 0 || px >= p->depth_dims[0] || py >= p->depth_dims[1]) /* cu:192-197 */
    return 0;
  /* cu:141-149 computeVoxelIDDepth: vtkImageData rows start at the bottom */
  int64_t depth_id = (int64_t)p->depth_dims[0] * (p->depth_dims[1] - 1 - py) + px;
  double depth = depths[depth_id]; /* cu:201 */
  if (depth == -1)                 /* cu:202 */
    return 0;
  *increment = ray_potential(p, center_cam[2], depth); /* cu:207-209 */
  return 1;
}

/* The driver loop of ProcessDepthMap<double> (cu:343-365) with the launch
 * geometry of cu:330-331 replayed on the CPU: for every depth map, for every
 * voxel (x fastest, cu:126-134), one project_one().  `grid` is accumulated in
 * place (cu:211, cu:323-327: it starts from whatever the caller supplies).
 *
 *   depths   [n_maps][H*W]  f64, vtk storage order (row 0 = bottom), -1 = no depth
 *   K16, RT16 [n_maps][16]  row-major 4x4 (cu:352-353)
 *   voxel_hits (nullable) [n_voxels] += 1 for every map reaching cu:211
 *   map_hits   (nullable) [n_maps]   += number of voxels reaching cu:211
 *
 * The hit counters are not a reference output; they expose every
 * in-frustum / sentinel decision so parity can be checked on integers.
 * n_threads <= 1 replays the reference's order exactly (map by map, cu:343; within a map z, y, x).  n_threads > 1
 * is the timed CPU baseline: ONE parallel region, z-layers dealt dynamically, and the thread that takes a layer runs
 * it through every map (k outer, m inner).  Each voxel still receives its increments in depth-map order, so the grid
 * and both counters are bit-identical to the serial replay (tests/test_oracle.py); what changes is that there is no
 * barrier per map, layers of uneven work balance, and a layer's grid pages are first touched -- hence NUMA-placed --
 * by the thread that keeps working on them. */
static void fuse_layer(const oracle_params *p, const double *dm, const double *K, const double *RT, int nx, int ny,
                       int k, double *grid, uint32_t *voxel_hits, uint64_t *hits_this_map)
{
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      double inc;
      if (project_one(p, dm, K, RT, i, j, k, &inc)) {
        int64_t grid_id = ((int64_t)k * ny + j) * nx + i; /* cu:126-134 */
        grid[grid_id] += inc;                             /* cu:211 */
        if (voxel_hits)
          voxel_hits[grid_id] += 1;
        *hits_this_map += 1;
      }
    }
}

void oracle_fuse(const oracle_params *p, const double *depths, const double *K16,
                 const double *RT16, int n_maps, double *grid, uint32_t *voxel_hits,
                 uint64_t *map_hits, int n_threads)
{
  const int nx = p->point_dims[0] - 1, ny = p->point_dims[1] - 1, nz = p->point_dims[2] - 1;
  const int64_t n_pix = (int64_t)p->depth_dims[0] * p->depth_dims[1];
  if (n_threads <= 1) {
    for (int m = 0; m < n_maps; ++m) { /* cu:343 */
      uint64_t hits_this_map = 0;
      for (int k = 0; k < nz; ++k)
        fuse_layer(p, depths + (int64_t)m * n_pix, K16 + 16 * m, RT16 + 16 * m, nx, ny, k, grid, voxel_hits, &hits_this_map);
      if (map_hits)
        map_hits[m] += hits_this_map;
    }
    return;
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
  for (int k = 0; k < nz; ++k)
    for (int m = 0; m < n_maps; ++m) {
      uint64_t hits_this_map = 0;
      fuse_layer(p, depths + (int64_t)m * n_pix, K16 + 16 * m, RT16 + 16 * m, nx, ny, k, grid, voxel_hits, &hits_this_map);
      if (map_hits && hits_this_map) {
#ifdef _OPENMP
#pragma omp atomic
#endif
        map_hits[m] += hits_this_map;
      }
    }
}

/* Same arithmetic for a caller-chosen list of voxel ids (x-fastest linear
 * ids), used to spot-check grids too large to fuse on the CPU in seconds.
 * out[v] starts from init[v] (nullable -> 0) and accumulates in map order. */
void oracle_fuse_voxels(const oracle_params *p, const double *depths, const double *K16,
                        const double *RT16, int n_maps, const int64_t *voxel_ids, int64_t n_ids,
                        const double *init, double *out, uint32_t *hits, int n_threads)
{
  const int nx = p->point_dims[0] - 1, ny = p->point_dims[1] - 1;
  const int64_t n_pix = (int64_t)p->depth_dims[0] * p->depth_dims[1];
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 1 ? n_threads : 1)
#endif
  for (int64_t v = 0; v < n_ids; ++v) {
    int64_t id = voxel_ids[v];
    int i = (int)(id % nx), j = (int)((id / nx) % ny), k = (int)(id / ((int64_t)nx * ny));
    double acc = init ? init[v] : 0.0;
    uint32_t h = 0;
    for (int m = 0; m < n_maps; ++m) {
      double inc;
      if (project_one(p, depths + (int64_t)m * n_pix, K16 + 16 * m, RT16 + 16 * m, i, j, k, &inc)) {
        acc += inc;
        h += 1;
      }
    }
    out[v] = acc;
    if (hits)
      hits[v] = h;
  }
}

/* RD.cxx:138-167 ReconstructionData::ApplyDepthThresholdFilter:
 * bestCost > threshold  =>  depth = -1, in place. */
void oracle_apply_depth_threshold(double *depths, const double *best_cost, int64_t n, double threshold)
{
  for (int64_t i = 0; i < n; ++i)
    if (best_cost[i] > threshold)
      depths[i] = -1;
}

/* RD.cxx:192-212 ReconstructionData::SetMatrixK: 3x3 K in the top-left of a
 * 4x4 identity. */
void oracle_k3_to_k4(const double K3[9], double K4[16])
{
  memset(K4, 0, 16 * sizeof(double));
  K4[0] = K4[5] = K4[10] = K4[15] = 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      K4[i * 4 + j] = K3[i * 3 + j];
}

/* Reconstruction/main.cxx:151-155: vtkCellDataToPointData on the filter's output, the step right after the path.
 * VTK is a third-party dependency (unpinned by the reference's CMakeLists.txt:8-17, absent from this image);
 * restated from its published algorithm, vtkCellDataToPointData::InterpolatePointData:
 *   for every point: cellIds = vtkStructuredData::GetPointCells(point); w = 1.0 / numCells;
 *   value = sum over cellIds IN THAT ORDER of w * cell value (vtkDataArray::InterpolateTuple: c = 0; c += w*v).
 * GetPointCells walks the eight offsets below and skips cells outside the grid.  numCells is 1, 2, 4 or 8 on an
 * image grid, so w*v is exact; only the order of additions can matter (a different order moves the result by at
 * most 3 ulp of the largest partial sum; tests/test_cell_to_point.py states that bound next to the bit-exact
 * comparison with this loop).
 * cells [nz][ny][nx] x fastest, points [(nz+1)][(ny+1)][(nx+1)] x fastest. */
void oracle_cell_to_point(const double *cells, int nx, int ny, int nz, double *points)
{
  static const int offset[8][3] = {{-1, 0, 0}, {-1, -1, 0}, {-1, -1, -1}, {-1, 0, -1},
                                   {0, 0, 0},  {0, -1, 0},  {0, -1, -1},  {0, 0, -1}};
#pragma omp parallel for schedule(static)
  for (int k = 0; k <= nz; ++k)
    for (int j = 0; j <= ny; ++j)
      for (int i = 0; i <= nx; ++i) {
        int64_t ids[8];
        int n = 0;
        for (int o = 0; o < 8; ++o) {
          const int ci = i + offset[o][0], cj = j + offset[o][1], ck = k + offset[o][2];
          if (ci < 0 || ci >= nx || cj < 0 || cj >= ny || ck < 0 || ck >= nz)
            continue;
          ids[n++] = ((int64_t)ck * ny + cj) * nx + ci;
        }
        const double w = 1.0 / (double)n; /* n >= 1: every lattice point touches a cell */
        double c = 0;
        for (int q = 0; q < n; ++q)
          c += w * cells[ids[q]];
        points[((int64_t)k * (ny + 1) + j) * (nx + 1) + i] = c;
      }
}

/* Reconstruction/main.cxx:169-173: vtkContourFilter (SetValue(0, contourValue)) over the point data -- marching cubes.
 * VTK is absent from this image; restated from its published algorithm (vtkMarchingCubes / the image-data contour path):
 * for every cell the case index collects one bit per corner, set when `s[corner] >= value`; index 0 and index 255 produce
 * no triangle.  So the cells that can contribute are those with 0 < (corners >= iso) < 8; a NaN corner compares false
 * (outside).  points [(nz+1)][(ny+1)][(nx+1)] x fastest; writes the first `capacity` active cells' linear ids
 * (k*ny + j)*nx + i in ascending order to ids (nullable) and returns how many cells are active. */
int64_t oracle_iso_active_cells(const double *points, int nx, int ny, int nz, double iso, int64_t *ids, int64_t capacity)
{
  int64_t n = 0;
  const int64_t prow = nx + 1, pplane = (int64_t)(nx + 1) * (ny + 1);
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        int inside = 0;
        for (int c = 0; c < 8; ++c)
          inside += points[(k + (c >> 2)) * pplane + (j + ((c >> 1) & 1)) * prow + i + (c & 1)] >= iso ? 1 : 0;
        if (inside == 0 || inside == 8)
          continue;
        if (ids && n < capacity)
          ids[n] = ((int64_t)k * ny + j) * nx + i;
        ++n;
      }
  return n;
}

/* Exposed for known-answer tests of the ray-potential function alone. */
double oracle_ray_potential(const oracle_params *p, double real_distance, double depth_map_distance)
{
  return ray_potential(p, real_distance, depth_map_distance);
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
