// microbench_window.hip -- what the FREE column's per-voxel look-up costs on gfx950, two ways:
//   A  byte from the validity map through the texture addresser (3 FMAs + rounding + cvt + add, buffer_load_ubyte, v_bfm, v_fma_f64)
//   B  bit from a 32 x 64 window held one row per lane (v_lshlrev, ds_bpermute_b32, v_bfe_i32, v_and, v_fma_f64)
// and whether ds_bpermute_b32 / v_bfe take the low bits of a "magic number" float (0x4B400000 + n) as the kernel plans to use them.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_window.hip -o build/microbench_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int ITER = 256, UNROLL = 8;

__global__ void check_kernel(uint32_t *out) {
  const int lane = threadIdx.x & 63;
  const uint32_t word = 0x9e3779b9u * (lane + 1);          // lane l's "window row"
  const int row = (lane * 7 + 3) & 63, col = (lane * 5 + 1) & 31;
  const float fy = 12582912.0f + (float)row, fx = 12582912.0f + (float)col;  // 1.5 * 2^23 + n
  const uint32_t by = __float_as_uint(fy), bx = __float_as_uint(fx);
  const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(by << 2), (int)word);
  int bit;
  asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(bit) : "v"(got), "v"(bx));
  out[lane] = got;
  out[64 + lane] = (uint32_t)bit;
  out[128 + lane] = 0x9e3779b9u * (row + 1);
  out[192 + lane] = (uint32_t)(-(int)(((0x9e3779b9u * (row + 1)) >> col) & 1u));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ map, double *out, int Wp8, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lx = lane & 7, ly = lane >> 3;
  float px = (float)((lx * 3) / 2), py = (float)((ly * 3) / 2);   // a 12 x 12 px patch
  double acc[UNROLL];
  for (int q = 0; q < UNROLL; ++q) acc[q] = 0.0;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(map), (short)0, 1 << 24, 0x00020000);
  uint32_t window = 0x9e3779b9u * (lane + 1 + wave);
  const double fs = -0.024;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    const float ox = (float)((it * 37 + wave * 11) & 255), oy = (float)((it * 13 + wave * 7) & 127);
#pragma unroll
    for (int q = 0; q < UNROLL; ++q) {
      const float x = px + ox + (float)q, y = py + oy + (float)(q & 3);
      if constexpr (MODE == 0) {
        const float yt = __builtin_rintf(__builtin_fmaf(y, 0.125f, -0.4375f));
        const int pix = (int)__builtin_fmaf(yt, (float)Wp8, __builtin_fmaf(x, 8.0f, y)) + 64;
        const unsigned short b = __builtin_amdgcn_raw_buffer_load_b8(rsrc, pix, 0, 0);
        unsigned hi;
        asm("v_bfm_b32 %0, %1, 20" : "=v"(hi) : "v"(__builtin_bit_cast(_Float16, b)));
        acc[q] = __builtin_fma(__hiloint2double((int)hi, 0), fs, acc[q]);
      } else {
        const uint32_t bx = __float_as_uint(x + 12582912.0f), by = __float_as_uint(y + 12582912.0f);
        const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(by << 2), (int)window);
        int bit;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(bit) : "v"(w), "v"(bx));
        acc[q] = __builtin_fma(__hiloint2double(bit & 0x3ff00000, 0), fs, acc[q]);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < UNROLL; ++q) s += acc[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
  uint32_t *chk;
  hipMalloc(&chk, 256 * 4);
  hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, chk);
  std::vector<uint32_t> h(256);
  hipMemcpy(h.data(), chk, 256 * 4, hipMemcpyDeviceToHost);
  int bad_perm = 0, bad_bfe = 0;
  for (int l = 0; l < 64; ++l) {
    bad_perm += h[l] != h[128 + l];
    bad_bfe += h[64 + l] != h[192 + l];
  }
  printf("ds_bpermute_b32 with a magic-float address (high bits set): %d of 64 lanes wrong\n", bad_perm);
  printf("v_bfe_i32 with a magic-float offset (high bits set):        %d of 64 lanes wrong\n", bad_bfe);
  uint8_t *map;
  double *out;
  unsigned long long *cyc;
  hipMalloc(&map, 1 << 24);
  hipMemset(map, 10, 1 << 24);
  for (int wps : {4, 5}) {
    const int waves = 256 * 4 * wps;
    hipMalloc(&out, (size_t)waves * 64 * 8);
    hipMalloc(&cyc, waves * 8);
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      for (int rep = 0; rep < 2; ++rep) {
        if (rep) hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, map, out, 8 * 1344 - 8, cyc);
        else hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, map, out, 8 * 1344 - 8, cyc);
        if (rep) hipEventRecord(b);
        hipDeviceSynchronize();
      }
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      const double lookups_per_simd = (double)wps * ITER * UNROLL;
      printf("%s  waves/SIMD %d: %7.1f cycles per wave-look-up per SIMD (kernel %.3f ms)\n",
             mode == 0 ? "A byte map through the TA  " : "B bit window, ds_bpermute ", wps, ms * 1e-3 * 2.4e9 / lookups_per_simd, ms);
    }
    hipFree(out); hipFree(cyc);
  }
  return 0;
}
