// fetch_calibration.hip -- what does rocprofv3's FETCH_SIZE report on gfx950 for dword-per-lane reads?
// The guide (MI355X_MICROARCH.md, HBM) calibrates only 16-B-per-lane streams (FETCH_SIZE = 1/2 of the bytes).
// The fusion kernel gathers one dword per lane, so this tool reads known byte counts with that width:
//   stream_dword   every dword of a 2 GiB buffer once, coalesced (256 B per wave-instruction)
//   sparse_dword   one dword out of every 128 B of the same buffer (each wave-instruction touches 64 lines)
//   sparse64_dword one dword out of every 64 B
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calibration.hip -o build/fetch_calibration
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_cal -- ./build/fetch_calibration
// Read:  FETCH_SIZE (KiB) * 1024 / bytes_touched for each kernel; the scale for the fusion kernel is
//        bytes_really_fetched / (FETCH_SIZE*1024) under the assumption stated in profiles/*_summary.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void stream_dword(const float *__restrict__ p, float *out, size_t n) {
  float acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 123.456f) out[0] = acc;
}
template <int STRIDE_DWORDS>
__global__ __launch_bounds__(256) void sparse_dword(const float *__restrict__ p, float *out, size_t n) {
  float acc = 0;
  const size_t lines = n / STRIDE_DWORDS;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < lines; i += (size_t)gridDim.x * blockDim.x)
    acc += p[i * STRIDE_DWORDS];
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  const size_t bytes = size_t(2) << 30, n = bytes / 4;
  float *p, *out;
  CHECK(hipMalloc(&p, bytes));
  CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(p, 0, bytes));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(stream_dword, dim3(8192), dim3(256), 0, 0, p, out, n);
    hipLaunchKernelGGL(sparse_dword<32>, dim3(8192), dim3(256), 0, 0, p, out, n);
    hipLaunchKernelGGL(sparse_dword<16>, dim3(8192), dim3(256), 0, 0, p, out, n);
    CHECK(hipDeviceSynchronize());
  }
  printf("bytes=%zu stream_dword touches all; sparse<32> touches %zu lines of 128 B; sparse<16> %zu lines of 64 B\n", bytes,
         n / 32, n / 16);
  return 0;
}
