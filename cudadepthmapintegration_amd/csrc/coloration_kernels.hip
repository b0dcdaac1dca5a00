// coloration_kernels.hip -- the MeshColoration pass on the GPU (BASELINE config 5).
//
// Reference (Coloration/MeshColoration.cxx:98-199, CPU only): for every mesh vertex and every view, project the
// vertex (ReconstructionData::TransformWorldToDepthMapPosition, RD.cxx:169-182: RT as a point transform, K as a
// vector transform, divide, std::round; NO test of the sign of z, NO depth test), keep the views whose pixel is
// inside the image, fetch that pixel's RGB (GetColorValue, RD.cxx:92-116: row flip) and store per vertex the mean
// (integer accumulation, MC.cxx:176-180), the median (Helper.h:174-187) and the number of views.
//
// Here: one lane per vertex.  Kernel 1 loops over the views (camera records through scalar loads), projects with
// the reference's expression in fp64 (correctly rounded divisions: the pixel decides which colour is read),
// accumulates count and integer sums and writes the fetched colour of every (view, vertex) pair to a scratch table
// [view][vertex] (uchar4, alpha = valid).  Kernel 2 finds the median per channel by an 8-step radix selection over
// that table (coalesced: consecutive lanes read consecutive entries).  Everything after the projection is integer
// arithmetic, so the three outputs are bit-identical to the reference's.
#include "../../include/dmi.h"
#include "fusion_kernels.h"

#include <string>
#include <vector>

namespace {

struct ColorView {
  double rt[12];           // rows 0..2 of [R|T]
  double k[9];             // rows 0..2, columns 0..2 of the 4x4 K (TransformVector ignores column 3)
  const uint8_t *color;    // [H][W][3], vtk row order (row 0 = bottom)
};

__device__ __forceinline__ bool to_pixel(double u, int &p) {  // round half away from zero; NaN/inf/|x| >= 2^31 outside
  const double r = round(u);
  if (!(r > -2147483648.0 && r < 2147483648.0)) return false;
  p = (int)r;
  return true;
}

__global__ __launch_bounds__(256) void project_color_kernel(const double *__restrict__ points, int64_t nv,
                                                            const ColorView *__restrict__ views, int n, int W, int H,
                                                            uchar4 *__restrict__ scratch, uint8_t *__restrict__ mean,
                                                            int32_t *__restrict__ count) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nv) return;
  const double x = points[3 * id], y = points[3 * id + 1], z = points[3 * id + 2];
  int cnt = 0, s0 = 0, s1 = 0, s2 = 0;
  for (int m = 0; m < n; ++m) {
    const ColorView *__restrict__ v = views + m;  // wave-uniform
    // vtkTransform::TransformPoint with MatrixTR (RD.cxx:173): M[i][0]*x + M[i][1]*y + M[i][2]*z + M[i][3], left to right
    const double cx = ((v->rt[0] * x + v->rt[1] * y) + v->rt[2] * z) + v->rt[3];
    const double cy = ((v->rt[4] * x + v->rt[5] * y) + v->rt[6] * z) + v->rt[7];
    const double cz = ((v->rt[8] * x + v->rt[9] * y) + v->rt[10] * z) + v->rt[11];
    // vtkTransform::TransformVector with Matrix4K (RD.cxx:175): no translation
    const double dx = (v->k[0] * cx + v->k[1] * cy) + v->k[2] * cz;
    const double dy = (v->k[3] * cx + v->k[4] * cy) + v->k[5] * cz;
    const double dz = (v->k[6] * cx + v->k[7] * cy) + v->k[8] * cz;
    uchar4 out = make_uchar4(0, 0, 0, 0);
    int px, py;
    if (to_pixel(dx / dz, px) && to_pixel(dy / dz, py) &&           // RD.cxx:177-181
        px >= 0 && py >= 0 && px < W && py < H) {                   // MC.cxx:158-163
      const uint8_t *c = v->color + ((int64_t)(H - 1 - py) * W + px) * 3;  // RD.cxx:106-108
      out = make_uchar4(c[0], c[1], c[2], 1);
      cnt += 1;
      s0 += c[0];  // std::accumulate(..., 0): integer running sums (MC.cxx:176-178)
      s1 += c[1];
      s2 += c[2];
    }
    scratch[(int64_t)m * nv + id] = out;
  }
  count[id] = cnt;  // MC.cxx:186 (0 when no view sees the vertex, MC.cxx:130)
  // sum / nbVal in double, then static_cast<unsigned char> (MC.cxx:179-180): exactly the integer quotient
  mean[3 * id + 0] = cnt ? (uint8_t)(s0 / cnt) : 0;
  mean[3 * id + 1] = cnt ? (uint8_t)(s1 / cnt) : 0;
  mean[3 * id + 2] = cnt ? (uint8_t)(s2 / cnt) : 0;
}

// k-th smallest (0-based) of the valid entries of one channel, by radix selection from the top bit down
__device__ __forceinline__ int select_kth(const uchar4 *__restrict__ scratch, int64_t nv, int64_t id, int n, int channel,
                                          int k) {
  int prefix = 0;
  for (int bit = 7; bit >= 0; --bit) {
    int zeros = 0;  // valid entries that match the prefix above `bit` and have a 0 at `bit`
    for (int m = 0; m < n; ++m) {
      const uchar4 e = scratch[(int64_t)m * nv + id];
      const int val = channel == 0 ? e.x : (channel == 1 ? e.y : e.z);
      zeros += (e.w != 0 && (val >> (bit + 1)) == prefix && ((val >> bit) & 1) == 0) ? 1 : 0;
    }
    if (k < zeros) {
      prefix = prefix << 1;
    } else {
      k -= zeros;
      prefix = (prefix << 1) | 1;
    }
  }
  return prefix;
}

__global__ __launch_bounds__(256) void median_kernel(const uchar4 *__restrict__ scratch, int64_t nv, int n,
                                                     const int32_t *__restrict__ count, uint8_t *__restrict__ median) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nv) return;
  const int cnt = count[id];
  for (int c = 0; c < 3; ++c) {
    int med = 0;
    if (cnt > 0) {
      // Helper.h:174-187: sorted[n/2], or the mean of sorted[n/2] and sorted[n/2 - 1] for an even count;
      // (a + b) / 2 in double, then static_cast<unsigned char> (MC.cxx:185): the integer (a + b) >> 1
      const int hi = select_kth(scratch, nv, id, n, c, cnt / 2);
      med = hi;
      if ((cnt & 1) == 0) med = (hi + select_kth(scratch, nv, id, n, c, cnt / 2 - 1)) >> 1;
    }
    median[3 * id + c] = (uint8_t)med;
  }
}

thread_local std::string g_color_error;

}  // namespace

extern "C" {

const char *dmi_color_last_error(void) { return g_color_error.c_str(); }

int dmi_color_mesh(const double *points, int64_t n_points, const uint8_t *colors, const double *K4, const double *RT4,
                   int32_t n_views, int32_t width, int32_t height, int32_t device, uint8_t *mean, uint8_t *median,
                   int32_t *count) {
  if (!points || !colors || !K4 || !RT4 || !mean || !median || !count) {
    g_color_error = "dmi_color_mesh: null argument";
    return DMI_ERR_INVALID_ARGUMENT;
  }
  if (n_points < 0 || n_views < 1 || width < 1 || height < 1) {
    g_color_error = "dmi_color_mesh: n_points >= 0, n_views >= 1, width >= 1, height >= 1 required";  // MC.cxx:102-106
    return DMI_ERR_INVALID_ARGUMENT;
  }
  if (n_points == 0) return DMI_OK;
  void *d_points = nullptr, *d_colors = nullptr, *d_views = nullptr, *d_scratch = nullptr, *d_mean = nullptr,
       *d_median = nullptr, *d_count = nullptr;
  hipStream_t stream = nullptr;
  auto cleanup = [&]() {
    for (void *p : {d_points, d_colors, d_views, d_scratch, d_mean, d_median, d_count})
      if (p) (void)hipFree(p);
    if (stream) (void)hipStreamDestroy(stream);
  };
  auto check = [&](hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    g_color_error = std::string("dmi_color_mesh: ") + what + ": " + hipGetErrorString(e);
    cleanup();
    return false;
  };
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    g_color_error = "dmi_color_mesh: no HIP device available";
    return DMI_ERR_DEVICE;
  }
  if (device < 0 || device >= ndev) {
    g_color_error = "dmi_color_mesh: device ordinal out of range";
    return DMI_ERR_INVALID_ARGUMENT;
  }
  if (!check(hipSetDevice(device), "hipSetDevice")) return DMI_ERR_DEVICE;
  if (!check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate")) return DMI_ERR_DEVICE;
  const size_t npix = (size_t)width * height;
  const size_t color_bytes = npix * 3 * (size_t)n_views;
  if (!check(hipMalloc(&d_points, (size_t)n_points * 24), "hipMalloc(points)")) return DMI_ERR_OUT_OF_MEMORY;
  if (!check(hipMalloc(&d_colors, color_bytes), "hipMalloc(colors)")) return DMI_ERR_OUT_OF_MEMORY;
  if (!check(hipMalloc(&d_views, sizeof(ColorView) * (size_t)n_views), "hipMalloc(views)")) return DMI_ERR_OUT_OF_MEMORY;
  if (!check(hipMalloc(&d_scratch, (size_t)n_points * (size_t)n_views * 4), "hipMalloc(scratch)")) return DMI_ERR_OUT_OF_MEMORY;
  if (!check(hipMalloc(&d_mean, (size_t)n_points * 3), "hipMalloc(mean)")) return DMI_ERR_OUT_OF_MEMORY;
  if (!check(hipMalloc(&d_median, (size_t)n_points * 3), "hipMalloc(median)")) return DMI_ERR_OUT_OF_MEMORY;
  if (!check(hipMalloc(&d_count, (size_t)n_points * 4), "hipMalloc(count)")) return DMI_ERR_OUT_OF_MEMORY;
  std::vector<ColorView> views((size_t)n_views);
  for (int m = 0; m < n_views; ++m) {
    for (int i = 0; i < 12; ++i) views[m].rt[i] = RT4[16 * (size_t)m + i];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) views[m].k[3 * r + c] = K4[16 * (size_t)m + 4 * r + c];
    views[m].color = static_cast<const uint8_t *>(d_colors) + (size_t)m * npix * 3;
  }
  if (!check(hipMemcpyAsync(d_points, points, (size_t)n_points * 24, hipMemcpyHostToDevice, stream), "copy points")) return DMI_ERR_DEVICE;
  if (!check(hipMemcpyAsync(d_colors, colors, color_bytes, hipMemcpyHostToDevice, stream), "copy colors")) return DMI_ERR_DEVICE;
  if (!check(hipMemcpyAsync(d_views, views.data(), sizeof(ColorView) * (size_t)n_views, hipMemcpyHostToDevice, stream), "copy views")) return DMI_ERR_DEVICE;
  const unsigned blocks = (unsigned)((n_points + 255) / 256);
  hipLaunchKernelGGL(project_color_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const double *>(d_points),
                     n_points, static_cast<const ColorView *>(d_views), n_views, width, height,
                     static_cast<uchar4 *>(d_scratch), static_cast<uint8_t *>(d_mean), static_cast<int32_t *>(d_count));
  if (!check(hipGetLastError(), "project_color_kernel")) return DMI_ERR_DEVICE;
  hipLaunchKernelGGL(median_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const uchar4 *>(d_scratch), n_points,
                     n_views, static_cast<const int32_t *>(d_count), static_cast<uint8_t *>(d_median));
  if (!check(hipGetLastError(), "median_kernel")) return DMI_ERR_DEVICE;
  if (!check(hipMemcpyAsync(mean, d_mean, (size_t)n_points * 3, hipMemcpyDeviceToHost, stream), "copy mean")) return DMI_ERR_DEVICE;
  if (!check(hipMemcpyAsync(median, d_median, (size_t)n_points * 3, hipMemcpyDeviceToHost, stream), "copy median")) return DMI_ERR_DEVICE;
  if (!check(hipMemcpyAsync(count, d_count, (size_t)n_points * 4, hipMemcpyDeviceToHost, stream), "copy count")) return DMI_ERR_DEVICE;
  if (!check(hipStreamSynchronize(stream), "synchronize")) return DMI_ERR_DEVICE;
  cleanup();
  return DMI_OK;
}

}  // extern "C"
