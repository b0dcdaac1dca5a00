// fusion_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the TSDF depth-map fusion path.
//
// What the reference does (Reconstruction/CudaReconstruction.cu:158-212, launched once per depth
// map at cu:363): one thread per voxel projects the voxel centre into ONE depth map and does a
// read-modify-write of the fp64 grid, so the whole grid crosses HBM twice per depth map.
//
// What this kernel does instead (voxel-stationary): one lane owns one voxel for the whole fusion,
// loops over every resident depth map, keeps the running sum in an fp64 register and writes the
// grid once.  A wavefront is 64 consecutive voxels along x, so the final store is one contiguous
// 256 B (f32) / 512 B (f64) segment per wave.  The per-map camera record (MapRec) is indexed by
// the wave-uniform loop counter, so it arrives through the scalar cache into SGPRs and feeds the
// fp64 VALU directly as scalar operands.  The only per-lane memory access in the loop is the
// depth gather, served by L1/L2/Infinity Cache (a depth map is re-read by many voxels).
//
// Arithmetic contract: IEEE fp64, the reference's expression order, no FMA contraction in any value
// that reaches the result (this file is compiled with -ffp-contract=off; the explicit fma() calls
// below only feed a self-checked reciprocal that selects WHICH exact code path runs).  The grid is
// therefore bit-identical to oracle/tsdf_oracle.c.
#include "fusion_kernels.h"
#include "fusion_device.h"

namespace dmi {

namespace {

constexpr int kWave = 64;

template <typename DepthT, typename GridT, int KMODE, bool FAST, bool COUNT>
__global__ __launch_bounds__(256) void fuse_kernel(const FuseArgs a) {
  // lane -> voxel: x is the wave dimension (cu:163 uses threadIdx.x for x as well)
  const int i = blockIdx.x * kWave + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z * blockDim.z + threadIdx.z + a.k_first;  // slab [k_first, k_first + k_count)
  const bool inside = i < a.nx && j < a.ny && k < a.k_first + a.k_count;

  // cu:78-83 computeVoxelCenter, cu:168 grid matrix -- once per voxel instead of once per map
  const double gx = a.ox + (i + 0.5) * a.sx;
  const double gy = a.oy + (j + 0.5) * a.sy;
  const double gz = a.oz + ((k + a.kz0) + 0.5) * a.sz;  // global cell index (z-slab contexts)
  const double wx = row4(a.g + 0, gx, gy, gz);
  const double wy = row4(a.g + 4, gx, gy, gz);
  const double wz = row4(a.g + 8, gx, gy, gz);

  const int64_t gid = ((int64_t)k * a.ny + j) * a.nx + i;  // cu:126-134
  GridT *__restrict__ grid = static_cast<GridT *>(a.grid);
  double acc = 0.0;
  if (a.init_from_grid && inside) acc = (double)grid[gid];  // cu:211 accumulates onto what is there
  uint32_t nhit = 0;

  const int m_end = a.first_map + a.n_maps;
  for (int m = a.first_map; m < m_end; ++m) {
    const MapRec *__restrict__ rec = a.maps + m;  // wave-uniform -> scalar loads
    // cu:172 world -> camera
    const double cx = row4(rec->rt + 0, wx, wy, wz);
    const double cy = row4(rec->rt + 4, wx, wy, wz);
    const double cz = row4(rec->rt + 8, wx, wy, wz);
    // cu:176 camera -> homogeneous pixel
    double hx, hy, hz;
    if (KMODE == K_GENERAL) {
      hx = row4(rec->k + 0, cx, cy, cz);
      hy = row4(rec->k + 4, cx, cy, cz);
      hz = row4(rec->k + 8, cx, cy, cz);
    } else {
      // K = [fx s cx0 0; 0 fy cy0 0; 0 0 1 0]: the dropped terms are products with an exact 0
      // (value +-0 for finite operands) and additions of +-0, which change no non-zero value;
      // a zero result can only change sign, which no later step observes.
      if (KMODE == K_PINHOLE_SKEW)
        hx = (rec->k[0] * cx + rec->k[1] * cy) + rec->k[2] * cz;
      else
        hx = rec->k[0] * cx + rec->k[2] * cz;
      hy = rec->k[5] * cy + rec->k[6] * cz;
      hz = cz;
    }

    int px = 0, py = 0;
    bool in;
    if (FAST) {
      int s = inside ? pixel_fast(hx, hy, hz, a.W, a.H, px, py) : 0;
      if (s < 0) s = pixel_exact(hx, hy, hz, a.W, a.H, px, py) ? 1 : 0;  // rare lanes only
      in = s > 0;
    } else {
      in = inside && pixel_exact(hx, hy, hz, a.W, a.H, px, py);
    }

    bool hit = false;
    if (in) {
      const DepthT *__restrict__ dm = static_cast<const DepthT *>(rec->depth);
      // cu:141-149, cu:201: the reference indexes W*(H-1-py)+px into the bottom-up vtk table; the
      // table is stored top-down here (flipped once at upload), so the same value sits at W*py+px
      const double depth = (double)dm[a.W * py + px];
      if (depth != -1.0) {                                         // cu:202
        acc += ray_potential(a, cz, depth);                        // cu:207-211
        hit = true;
      }
    }
    if (COUNT) {
      nhit += hit ? 1u : 0u;
      // wavefront ballot: one popcount per wave and map instead of 64 atomics
      const unsigned long long b = __ballot(hit);
      if (b != 0 && (threadIdx.x & (kWave - 1)) == 0) atomicAdd(&a.map_hits[m], (unsigned long long)__popcll(b));
    }
  }

  if (inside) {
    grid[gid] = stored_sum<GridT>(acc);
    if (COUNT) a.voxel_hits[gid] += nhit;
  }
}

template <typename DepthT, typename GridT, int KMODE, bool FAST>
hipError_t launch_count(const FuseArgs &a, const FuseConfig &cfg, dim3 grid, dim3 block, hipStream_t s) {
  if (cfg.count_hits)
    hipLaunchKernelGGL((fuse_kernel<DepthT, GridT, KMODE, FAST, true>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((fuse_kernel<DepthT, GridT, KMODE, FAST, false>), grid, block, 0, s, a);
  return hipGetLastError();
}

template <typename DepthT, typename GridT, int KMODE>
hipError_t launch_fast(const FuseArgs &a, const FuseConfig &cfg, dim3 grid, dim3 block, hipStream_t s) {
  if (cfg.variant & VAR_EXACT_DIVISION) return launch_count<DepthT, GridT, KMODE, false>(a, cfg, grid, block, s);
  return launch_count<DepthT, GridT, KMODE, true>(a, cfg, grid, block, s);
}

template <typename DepthT, typename GridT>
hipError_t launch_kmode(const FuseArgs &a, const FuseConfig &cfg, dim3 grid, dim3 block, hipStream_t s) {
  const int km = (cfg.variant & VAR_GENERAL_K) ? (int)K_GENERAL : cfg.k_mode;
  switch (km) {
    case K_PINHOLE: return launch_fast<DepthT, GridT, K_PINHOLE>(a, cfg, grid, block, s);
    case K_PINHOLE_SKEW: return launch_fast<DepthT, GridT, K_PINHOLE_SKEW>(a, cfg, grid, block, s);
    default: return launch_fast<DepthT, GridT, K_GENERAL>(a, cfg, grid, block, s);
  }
}

// ---- depth upload: threshold + row flip + optional narrowing ---------------------------------------
__global__ __launch_bounds__(256) void widen_depth_kernel(const float *__restrict__ in, double *__restrict__ out,
                                                          int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (double)in[i];
}

// grid element type conversion on the device (dmi_upload_grid / dmi_download_grid_* when the caller's type is not the
// grid's): the same IEEE conversions as the host casts they replace, at HBM speed instead of one core's
template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void convert_grid_kernel(const InT *__restrict__ in, OutT *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (OutT)in[i];
}

inline int blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

hipError_t launch_fuse(const FuseArgs &a, const FuseConfig &cfg, hipStream_t stream) {
  // block = 64 lanes along x times 4 rows; variant bits 2..3 pick how the 4 rows are laid out
  dim3 block(kWave, 4, 1);
  switch ((cfg.variant >> 2) & 3) {
    case 1: block = dim3(kWave, 2, 2); break;
    case 2: block = dim3(kWave, 1, 4); break;
    case 3: block = dim3(kWave, 1, 1); break;
    default: break;
  }
  dim3 grid((a.nx + kWave - 1) / kWave, (a.ny + block.y - 1) / block.y, (a.k_count + block.z - 1) / block.z);
  if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidConfiguration;
  if (cfg.depth_is_f64) {
    if (cfg.grid_is_f64) return launch_kmode<double, double>(a, cfg, grid, block, stream);
    return launch_kmode<double, float>(a, cfg, grid, block, stream);
  }
  if (cfg.grid_is_f64) return launch_kmode<float, double>(a, cfg, grid, block, stream);
  return launch_kmode<float, float>(a, cfg, grid, block, stream);
}

// eight independent chains per lane: enough to cover the FMA latency at any occupancy
__global__ __launch_bounds__(256) void fp64_probe_kernel(double *__restrict__ out, int iters) {
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-12 * blockIdx.x;
  double v0 = 0.1, v1 = 0.2, v2 = 0.3, v3 = 0.4, v4 = 0.5, v5 = 0.6, v6 = 0.7, v7 = 0.8;
  for (int n = 0; n < iters; ++n) {
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
    v4 = __builtin_fma(v4, a, b); v5 = __builtin_fma(v5, a, b); v6 = __builtin_fma(v6, a, b); v7 = __builtin_fma(v7, a, b);
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
}

hipError_t launch_fp64_probe(double *out, int blocks, int iters, hipStream_t stream) {
  hipLaunchKernelGGL(fp64_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, iters);
  return hipGetLastError();
}

hipError_t launch_convert_grid(const void *in, int in_is_f64, void *out, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (in_is_f64)
    hipLaunchKernelGGL((convert_grid_kernel<double, float>), dim3(blocks_for(n)), dim3(256), 0, stream,
                       static_cast<const double *>(in), static_cast<float *>(out), n);
  else
    hipLaunchKernelGGL((convert_grid_kernel<float, double>), dim3(blocks_for(n)), dim3(256), 0, stream,
                       static_cast<const float *>(in), static_cast<double *>(out), n);
  return hipGetLastError();
}

hipError_t launch_widen_depth(const float *in, double *out, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(widen_depth_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, in, out, n);
  return hipGetLastError();
}

}  // namespace dmi
