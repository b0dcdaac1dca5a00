// microbench_gather.hip -- what one wave-wide dword gather costs the texture addresser / L1 on gfx950, by address pattern.
// Every wave issues ITER x 8 independent buffer-style loads from a small (L1 / L2 resident) table; the pattern decides how
// many distinct cache lines the 64 lanes touch.  Reported: cycles per gather per CU (4 SIMDs share one TA).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o build/microbench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int ITER = 512;
__global__ void k(const float *__restrict__ tab, float *out, int pattern, int pitch, int span, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lx = lane & 7, ly = lane >> 3;
  int base;
  switch (pattern) {
    case 0: base = 0; break;                                  // every lane the same dword
    case 1: base = lane; break;                               // 64 consecutive dwords: 2 lines
    case 2: base = ly * pitch + lx; break;                    // 8 rows x 8 consecutive px: 8 lines
    case 3: base = ly * pitch + (lx * 3) / 2; break;          // 8 rows x 12 px
    case 4: base = (ly * 3 / 2) * pitch + (lx * 3) / 2; break;  // 12 rows x 12 px, row-major table
    case 5: base = ((ly * 5 + lx * 3) / 4) * pitch + (lx * 5 - ly * 3 + 24) / 4; break;  // a rotated 8x8 patch, ~14 x 14 px
    case 6: base = lane * pitch; break;                       // 64 rows: 64 lines
    case 7: { const int x = (lx * 3) / 2, y = (ly * 3) / 2; base = ((y >> 3) * pitch + x) * 8 + (y & 7); break; }  // 12x12 px, 8-row column tiles
    default: base = lane * 32; break;                         // 64 lines, 128 B apart
  }
  float acc = 0.f;
  const int wrap = span - 1;  // span: power of two, dwords
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    const int o = (it * 977 + wave * 131) & wrap;
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += tab[(base + o + q * 4099) & wrap];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (lane == 0) cyc[wave] = t1 - t0;
}
int main() {
  const int span = 1 << 22;  // 16 MB table of floats: L2 / infinity-cache resident after the warm-up
  float *tab, *out;
  unsigned long long *cyc;
  hipMalloc(&tab, (size_t)span * 4);
  hipMemset(tab, 0, (size_t)span * 4);
  const int waves_per_simd = 5, waves = 256 * 4 * waves_per_simd;
  hipMalloc(&out, (size_t)waves * 64 * 4);
  hipMalloc(&cyc, waves * 8);
  const char *names[] = {"same dword", "64 consecutive", "8 rows x 8 px", "8 rows x 12 px", "12 x 12 px patch", "rotated patch", "64 rows", "12x12 in 8-row column tiles", "64 lines 128B apart"};
  for (int small = 0; small < 2; ++small)
    for (int p = 0; p < 9; ++p) {
      const int sp = small ? (1 << 12) : span;   // 16 KB: L1 resident
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, tab, out, p, 1280, sp, cyc);
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, tab, out, p, 1280, sp, cyc);
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      const double gathers_per_cu = (double)waves / 256 * ITER * 8;
      printf("%-30s table %5d KB: %7.1f cycles per gather per CU (kernel %.3f ms)\n", names[p], sp / 256, ms * 1e-3 * 2.4e9 / gathers_per_cu, ms);
    }
  return 0;
}
