// grid_post.hip -- what the reference's CLI does to the fused grid right after the filter: cell data -> point
// data (Reconstruction/main.cxx:151-155, vtkCellDataToPointData on the filter's output).  Streaming, HBM-bound.
//
// Semantics restated (VTK is a third-party dependency, absent here; vtkCellDataToPointData::InterpolatePointData
// over vtkStructuredData::GetPointCells): the value of lattice point (i, j, k) of the (nx+1)(ny+1)(nz+1) point
// grid is  c = 0; for every EXISTING adjacent cell, in the order of the offset table below, c += w * v(cell)
// with w = 1.0 / (number of adjacent cells).  On an image grid that number is 1, 2, 4 or 8, so every product
// is exact and only the order of the additions matters; oracle/tsdf_oracle.c (oracle_cell_to_point) states the
// same loop on the CPU and the GPU result is bit-identical to it.
#include <stdlib.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "fusion_kernels.h"

namespace dmi {
namespace {

// One layer of the 2 x 2 cells around a point column: (i-1, j-1), (i, j-1), (i-1, j), (i, j), in the grid's storage
// type.  Loads are UNCONDITIONAL from clamped (always valid) addresses; which values exist is decided when they are
// summed.  A predicated load followed by a widening conversion makes hipcc wait for every load inside its own
// branch, which serialised them (measured: the f32 grid ran slower than the f64 one).
template <typename GridT>
struct Quad {
  GridT mm, pm, mp, pp;
};

template <typename GridT>
__device__ __forceinline__ Quad<GridT> load_quad(const GridT *__restrict__ layer, int64_t row_m, int64_t row_p, int im,
                                                 int ip) {
  Quad<GridT> q;
  q.mm = layer[row_m + im];
  q.pm = layer[row_m + ip];
  q.mp = layer[row_p + im];
  q.pp = layer[row_p + ip];
  return q;
}

// Block = 64 x 4 points in (i, j); every thread produces KZ point layers along k from KZ + 1 cell layers, all
// loaded before the first use (KZ + 1 independent loads of 4 values in flight per thread), so each cell value is
// requested 4 (1 + 1/KZ) times -- served by L1/L2 -- instead of 8.
template <typename GridT, int KZ, int BY>
__global__ __launch_bounds__(64 * BY) void cell_to_point_kernel(const GridT *__restrict__ cells, double *__restrict__ points,
                                                            int nx, int ny, int nz) {
  const int i = blockIdx.x * 64 + threadIdx.x;  // point indices: 0..nx, 0..ny, 0..nz
  const int j = blockIdx.y * BY + threadIdx.y;
  const int k0 = blockIdx.z * KZ;
  if (i > nx || j > ny) return;
  const bool xm = i >= 1, xp = i < nx, ym = j >= 1, yp = j < ny;
  const int nxy = ((xm ? 1 : 0) + (xp ? 1 : 0)) * ((ym ? 1 : 0) + (yp ? 1 : 0));
  const int64_t plane = (int64_t)nx * ny;
  const int64_t prow = (int64_t)(nx + 1);
  // clamped cell indices: always inside the grid (cell_dims >= 1)
  const int im = xm ? i - 1 : 0, ip = xp ? i : nx - 1;
  const int64_t row_m = (int64_t)(ym ? j - 1 : 0) * nx, row_p = (int64_t)(yp ? j : ny - 1) * nx;
  Quad<GridT> q[KZ + 1];  // q[t] = cell layer k0 - 1 + t (clamped)
#pragma unroll
  for (int t = 0; t <= KZ; ++t) {
    int kc = k0 - 1 + t;  // wave-uniform
    kc = kc < 0 ? 0 : (kc >= nz ? nz - 1 : kc);
    q[t] = load_quad(cells + kc * plane, row_m, row_p, im, ip);
  }
#pragma unroll
  for (int kk = 0; kk < KZ; ++kk) {
    const int k = k0 + kk;
    if (k <= nz) {
      const bool zm = k >= 1, zp = k < nz;  // wave-uniform
      const Quad<GridT> below = q[kk], here = q[kk + 1];
      const int n = nxy * ((zm ? 1 : 0) + (zp ? 1 : 0));
      const double w = 1.0 / (double)n;
      // vtkStructuredData::GetPointCells offset order: (-1,0,0) (-1,-1,0) (-1,-1,-1) (-1,0,-1) (0,0,0) (0,-1,0)
      // (0,-1,-1) (0,0,-1); a cell outside the grid is skipped
      double c = 0.0;
      if (xm && yp && zp) c += w * (double)here.mp;
      if (xm && ym && zp) c += w * (double)here.mm;
      if (xm && ym && zm) c += w * (double)below.mm;
      if (xm && yp && zm) c += w * (double)below.mp;
      if (xp && yp && zp) c += w * (double)here.pp;
      if (xp && ym && zp) c += w * (double)here.pm;
      if (xp && ym && zm) c += w * (double)below.pm;
      if (xp && yp && zm) c += w * (double)below.pp;
      // written once and not read again by this kernel: a non-temporal store (5-6 % off the pass, same box, A/B:
      // profiles/r05i_cell_to_point_stores.json)
      __builtin_nontemporal_store(c, &points[((int64_t)k * (ny + 1) + j) * prow + i]);
    }
  }
}

template <typename GridT, int KZ, int BY>
void launch_kz(const GridT *cells, double *points, int nx, int ny, int nz, hipStream_t stream) {
  const dim3 block(64, BY);
  const dim3 grid((unsigned)((nx + 1 + 63) / 64), (unsigned)((ny + 1 + BY - 1) / BY), (unsigned)((nz + 1 + KZ - 1) / KZ));
  hipLaunchKernelGGL((cell_to_point_kernel<GridT, KZ, BY>), grid, block, 0, stream, cells, points, nx, ny, nz);
}

template <typename GridT>
void launch_typed(const GridT *cells, double *points, int nx, int ny, int nz, hipStream_t stream) {
#ifdef DMI_TUNING
  // tuning experiments (tools/gpu_c2p_tune.py): column height and block height from the environment
  const char *e = getenv("DMI_C2P_KZ"), *f = getenv("DMI_C2P_BY");
  const int kz = e ? atoi(e) : 8, by = f ? atoi(f) : 4;
#define DMI_C2P_CASE(KZ_, BY_) \
  if (kz == KZ_ && by == BY_) return launch_kz<GridT, KZ_, BY_>(cells, points, nx, ny, nz, stream);
  DMI_C2P_CASE(1, 4) DMI_C2P_CASE(2, 4) DMI_C2P_CASE(4, 4) DMI_C2P_CASE(16, 4)
  DMI_C2P_CASE(4, 2) DMI_C2P_CASE(8, 2) DMI_C2P_CASE(16, 2) DMI_C2P_CASE(4, 8) DMI_C2P_CASE(8, 8) DMI_C2P_CASE(8, 1) DMI_C2P_CASE(16, 1)
#undef DMI_C2P_CASE
#endif
  return launch_kz<GridT, 8, 4>(cells, points, nx, ny, nz, stream);  // profiles/r01w_cell_to_point_tuning.json, r05i
}

// ---- iso-value pre-pass ------------------------------------------------------------------------------------------
// Reconstruction/main.cxx:169-173 contours the point data at `contour` (vtkContourFilter -> marching cubes): every cell is
// visited and its eight corner values compared with the iso-value; only the cells whose corners are not all on one side
// produce triangles.  This pass marks those cells on the GPU -- corner c is "inside" when value >= iso (the marching-cubes
// case index's bit, a NaN is outside), a cell is ACTIVE when 0 < inside corners < 8 -- so that a host marching cubes can
// visit them alone.  Cells are taken in rows of x: block b handles 256 consecutive cells of one (j, k) row, so block
// order is the cells' linear order and the compacted list comes out ascending.
constexpr int kIsoBlock = 256;

__device__ __forceinline__ bool iso_cell_active(const double *__restrict__ points, int nx, int ny, int i, int j, int k, double iso) {
  const int64_t prow = nx + 1, pplane = (int64_t)(nx + 1) * (ny + 1);
  const double *p = points + (int64_t)k * pplane + (int64_t)j * prow + i;
  int inside = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const double v = p[(c & 1) + ((c >> 1) & 1) * prow + (c >> 2) * pplane];
    inside += v >= iso ? 1 : 0;
  }
  return inside != 0 && inside != 8;
}

template <bool WRITE>
__global__ __launch_bounds__(kIsoBlock) void iso_cells_kernel(const double *__restrict__ points, int nx, int ny, int nz,
                                                              int blocks_per_row, double iso, uint32_t *__restrict__ counts,
                                                              const uint64_t *__restrict__ bases, int64_t *__restrict__ ids,
                                                              uint64_t capacity) {
  __shared__ uint32_t wave_counts[kIsoBlock / 64];
  const int64_t row = blockIdx.x / blocks_per_row;  // = k * ny + j
  const int i = (int)(blockIdx.x - row * blocks_per_row) * kIsoBlock + threadIdx.x;
  const int k = (int)(row / ny), j = (int)(row - (int64_t)k * ny);
  const bool active = i < nx && iso_cell_active(points, nx, ny, i, j, k, iso);
  const unsigned long long m = __builtin_amdgcn_ballot_w64(active);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_counts[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if constexpr (!WRITE) {
    if (threadIdx.x == 0) {
      uint32_t n = 0;
      for (int w = 0; w < kIsoBlock / 64; ++w) n += wave_counts[w];
      counts[blockIdx.x] = n;
    }
  } else {
    if (!active) return;
    uint64_t pos = bases[blockIdx.x];
    for (int w = 0; w < wave; ++w) pos += wave_counts[w];
    pos += (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
    if (pos < capacity) ids[pos] = row * nx + i;  // the cell's linear id, x fastest (cu:126-134)
  }
}

}  // namespace

size_t iso_block_count(int nx, int ny, int nz) { return (size_t)((nx + kIsoBlock - 1) / kIsoBlock) * (size_t)ny * (size_t)nz; }

// counts[b] = active cells of block b, bases[b] = their exclusive prefix sum, *(bases + n_blocks) = the total
hipError_t launch_iso_count(const double *points, int nx, int ny, int nz, double iso, uint32_t *counts, uint64_t *bases,
                            void *scan_temp, size_t *scan_temp_bytes, hipStream_t stream) {
  const size_t n_blocks = iso_block_count(nx, ny, nz);
  if (!scan_temp) {  // size query
    return rocprim::exclusive_scan(nullptr, *scan_temp_bytes, counts, bases, (uint64_t)0, n_blocks + 1, rocprim::plus<uint64_t>(), stream);
  }
  hipLaunchKernelGGL((iso_cells_kernel<false>), dim3((unsigned)n_blocks), dim3(kIsoBlock), 0, stream, points, nx, ny, nz,
                     (nx + kIsoBlock - 1) / kIsoBlock, iso, counts, (const uint64_t *)nullptr, (int64_t *)nullptr, (uint64_t)0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // n_blocks + 1 inputs (the last one a zero the caller keeps there): bases[n_blocks] is the total
  return rocprim::exclusive_scan(scan_temp, *scan_temp_bytes, counts, bases, (uint64_t)0, n_blocks + 1, rocprim::plus<uint64_t>(), stream);
}

hipError_t launch_iso_write(const double *points, int nx, int ny, int nz, double iso, const uint64_t *bases, int64_t *ids,
                            uint64_t capacity, hipStream_t stream) {
  const size_t n_blocks = iso_block_count(nx, ny, nz);
  hipLaunchKernelGGL((iso_cells_kernel<true>), dim3((unsigned)n_blocks), dim3(kIsoBlock), 0, stream, points, nx, ny, nz,
                     (nx + kIsoBlock - 1) / kIsoBlock, iso, (uint32_t *)nullptr, bases, ids, capacity);
  return hipGetLastError();
}

hipError_t launch_cell_to_point(const void *cells, int cells_are_f64, double *points, int nx, int ny, int nz,
                                hipStream_t stream) {
  if (cells_are_f64)
    launch_typed(static_cast<const double *>(cells), points, nx, ny, nz, stream);
  else
    launch_typed(static_cast<const float *>(cells), points, nx, ny, nz, stream);
  return hipGetLastError();
}

}  // namespace dmi
