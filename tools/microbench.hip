// microbench.hip -- gfx950 VALU issue costs of the instructions the fusion kernel is made of, and the
// accuracy of v_rcp_f64 / v_rcp_f32 seeds.  Standalone: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb
//
// Method: every wave runs ITER iterations of 16 independent instances of one instruction (inline asm so
// the compiler can neither fuse nor drop them); grid = 256 CUs x 4 SIMDs x WPS waves.  Reported:
// cycles per wave-instruction per SIMD = s_memtime delta / (ITER*16) / WPS (the SIMD interleaves WPS waves).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                           \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

constexpr int ITER = 2048;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// f64 two-operand: d[i] = op(d[i], s)
#define KERNEL_F64_2(NAME, ASM)                                                                     \
  __global__ void NAME(double *out, unsigned long long *cyc, double s) {                            \
    double d[16];                                                                                   \
    for (int i = 0; i < 16; ++i) d[i] = 1.0 + 0.001 * (threadIdx.x + i);                            \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
    for (int it = 0; it < ITER; ++it) {                                                             \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(d[i]) : "v"(s));       \
    }                                                                                               \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
    double acc = 0;                                                                                 \
    for (int i = 0; i < 16; ++i) acc += d[i];                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                               \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;       \
  }

KERNEL_F64_2(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_F64_2(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_F64_2(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
KERNEL_F64_2(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_F64_2(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL_F64_2(k_floor_f64, "v_floor_f64 %0, %0")
KERNEL_F64_2(k_min_f64, "v_min_f64 %0, %0, %1")
KERNEL_F64_2(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL_F64_2(k_mov_b64, "v_mov_b64 %0, %1")

// f32 / int ops on one dword
#define KERNEL_B32(NAME, ASM)                                                                       \
  __global__ void NAME(double *out, unsigned long long *cyc, double sd) {                           \
    float d[16];                                                                                    \
    float s = (float)sd;                                                                            \
    for (int i = 0; i < 16; ++i) d[i] = 1.0f + 0.001f * (threadIdx.x + i);                          \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
    for (int it = 0; it < ITER; ++it) {                                                             \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(d[i]) : "v"(s));       \
    }                                                                                               \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
    float acc = 0;                                                                                  \
    for (int i = 0; i < 16; ++i) acc += d[i];                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                               \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;       \
  }

KERNEL_B32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL_B32(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
KERNEL_B32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_B32(k_rndne_f32, "v_rndne_f32 %0, %0")
KERNEL_B32(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_B32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %1")
KERNEL_B32(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL_B32(k_bfi_b32, "v_bfi_b32 %0, %1, %0, %1")

KERNEL_B32(k_max_abs_f32, "v_max_f32 %0, |%0|, |%1|")
KERNEL_B32(k_cmp_f32, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL_B32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL_B32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL_B32(k_add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 2")
KERNEL_B32(k_lshlrev_b32, "v_lshlrev_b32 %0, 2, %0")
KERNEL_B32(k_med3_f32, "v_med3_f32 %0, %0, %1, %1")
KERNEL_B32(k_fma_f32_sgpr, "v_fma_f32 %0, s8, %1, %0")
// packed f32: two floats per lane in a 64-bit register pair
KERNEL_F64_2(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
KERNEL_F64_2(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL_F64_2(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL_F64_2(k_pk_fma_f32_bcast, "v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[1,0,1]")

// conversions: f64 <-> f32 / i32 (dst and src differ in width)
#define KERNEL_CVT(NAME, ASM_A, ASM_B)                                                              \
  __global__ void NAME(double *out, unsigned long long *cyc, double s) {                            \
    double d[16];                                                                                   \
    float f[16];                                                                                    \
    for (int i = 0; i < 16; ++i) {                                                                  \
      d[i] = s + 0.001 * (threadIdx.x + i);                                                         \
      f[i] = (float)d[i];                                                                           \
    }                                                                                               \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
    for (int it = 0; it < ITER / 2; ++it) {                                                         \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM_A : "=v"(f[i]) : "v"(d[i]));  \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM_B : "=v"(d[i]) : "v"(f[i]));  \
    }                                                                                               \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
    double acc = 0;                                                                                 \
    for (int i = 0; i < 16; ++i) acc += d[i] + f[i];                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                               \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;       \
  }

KERNEL_CVT(k_cvt_f32_f64_pair, "v_cvt_f32_f64 %0, %1", "v_cvt_f64_f32 %0, %1")
KERNEL_CVT(k_cvt_i32_f64_pair, "v_cvt_i32_f64 %0, %1", "v_cvt_f64_i32 %0, %1")

// ---- seed accuracy -----------------------------------------------------------------------------
__global__ void k_rcp_accuracy(const double *x, double *err, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double h = x[i];
  double r0 = __builtin_amdgcn_rcp(h);
  double e = __builtin_fma(-h, r0, 1.0);
  double r1 = __builtin_fma(r0, e, r0);
  double e1 = __builtin_fma(-h, r1, 1.0);
  double r2 = __builtin_fma(r1, e1, r1);
  double e2 = __builtin_fma(-h, r2, 1.0);
  float hf = (float)h;
  double s0 = (double)__builtin_amdgcn_rcpf(hf);
  double f0 = __builtin_fma(-h, s0, 1.0);
  double s1 = __builtin_fma(s0, f0, s0);
  double f1 = __builtin_fma(-h, s1, 1.0);
  err[i * 5 + 0] = fabs(e);   // rcp_f64 seed
  err[i * 5 + 1] = fabs(e1);  // + 1 Newton
  err[i * 5 + 2] = fabs(e2);  // + 2 Newton
  err[i * 5 + 3] = fabs(f0);  // rcp_f32 seed (incl. f64->f32 rounding of h)
  err[i * 5 + 4] = fabs(f1);  // + 1 Newton in f64
}

template <typename K>
void run(const char *name, K kernel, int wps, double *d_out, unsigned long long *d_cyc, double clock_ghz_hint) {
  const int cus = 256;
  const int waves = cus * 4 * wps;
  const int threads = 256;
  const int blocks = waves * 64 / threads;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 1.0000001);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 1.0000001);
  CHECK(hipEventRecord(b));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> cyc(waves);
  CHECK(hipMemcpy(cyc.data(), d_cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : cyc) mean += (double)c;
  mean /= waves;
  const double per_inst_wave = mean / (ITER * 16.0);        // cycles one wave sees per instruction
  const double per_inst_simd = per_inst_wave / wps;         // SIMD issue cost per wave-instruction
  const double wall_cyc = ms * 1e-3 * clock_ghz_hint * 1e9 / (ITER * 16.0) / wps;
  printf("%-22s wps=%d  memtime cyc/inst/SIMD = %6.2f   (wall @%.1fGHz: %6.2f)  kernel %.3f ms\n", name, wps,
         per_inst_simd, clock_ghz_hint, wall_cyc, ms);
}

int main() {
  double *d_out;
  unsigned long long *d_cyc;
  CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 4 * 8 * 64));
  CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 256 * 4 * 8));
  const double ghz = 2.4;
  for (int wps : {1, 5, 8}) {
    run("v_add_f64", k_add_f64, wps, d_out, d_cyc, ghz);
    run("v_mul_f64", k_mul_f64, wps, d_out, d_cyc, ghz);
    run("v_fma_f64", k_fma_f64, wps, d_out, d_cyc, ghz);
    run("v_rcp_f64", k_rcp_f64, wps, d_out, d_cyc, ghz);
    run("v_rndne_f64", k_rndne_f64, wps, d_out, d_cyc, ghz);
    run("v_floor_f64", k_floor_f64, wps, d_out, d_cyc, ghz);
    run("v_min_f64", k_min_f64, wps, d_out, d_cyc, ghz);
    run("v_cmp_lt_f64", k_cmp_f64, wps, d_out, d_cyc, ghz);
    run("v_mov_b64", k_mov_b64, wps, d_out, d_cyc, ghz);
    run("v_add_f32", k_add_f32, wps, d_out, d_cyc, ghz);
    run("v_fma_f32", k_fma_f32, wps, d_out, d_cyc, ghz);
    run("v_rcp_f32", k_rcp_f32, wps, d_out, d_cyc, ghz);
    run("v_rndne_f32", k_rndne_f32, wps, d_out, d_cyc, ghz);
    run("v_cndmask_b32", k_cndmask_b32, wps, d_out, d_cyc, ghz);
    run("v_mad_u32_u24", k_mad_u32_u24, wps, d_out, d_cyc, ghz);
    run("v_cmp_lt_u32", k_cmp_u32, wps, d_out, d_cyc, ghz);
    run("v_bfi_b32", k_bfi_b32, wps, d_out, d_cyc, ghz);
    run("v_max_f32 |a|,|b|", k_max_abs_f32, wps, d_out, d_cyc, ghz);
    run("v_cmp_lt_f32", k_cmp_f32, wps, d_out, d_cyc, ghz);
    run("v_cvt_i32_f32", k_cvt_i32_f32, wps, d_out, d_cyc, ghz);
    run("v_lshl_add_u32", k_lshl_add_u32, wps, d_out, d_cyc, ghz);
    run("v_add_lshl_u32", k_add_lshl_u32, wps, d_out, d_cyc, ghz);
    run("v_lshlrev_b32", k_lshlrev_b32, wps, d_out, d_cyc, ghz);
    run("v_med3_f32", k_med3_f32, wps, d_out, d_cyc, ghz);
    run("v_fma_f32 (sgpr)", k_fma_f32_sgpr, wps, d_out, d_cyc, ghz);
    run("v_pk_fma_f32", k_pk_fma_f32, wps, d_out, d_cyc, ghz);
    run("v_pk_mul_f32", k_pk_mul_f32, wps, d_out, d_cyc, ghz);
    run("v_pk_add_f32", k_pk_add_f32, wps, d_out, d_cyc, ghz);
    run("v_pk_fma_f32 bcast", k_pk_fma_f32_bcast, wps, d_out, d_cyc, ghz);
    run("cvt f32<->f64 (avg)", k_cvt_f32_f64_pair, wps, d_out, d_cyc, ghz);
    run("cvt i32<->f64 (avg)", k_cvt_i32_f64_pair, wps, d_out, d_cyc, ghz);
    printf("\n");
  }

  // reciprocal seed accuracy over a log-uniform range of positive doubles
  const int n = 1 << 20;
  std::vector<double> x(n);
  srand(7);
  for (int i = 0; i < n; ++i) {
    double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
    x[i] = exp2(-20.0 + 40.0 * u) * (1.0 + v);
  }
  double *d_x, *d_err;
  CHECK(hipMalloc(&d_x, n * sizeof(double)));
  CHECK(hipMalloc(&d_err, n * 5 * sizeof(double)));
  CHECK(hipMemcpy(d_x, x.data(), n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_rcp_accuracy, dim3(n / 256), dim3(256), 0, 0, d_x, d_err, n);
  std::vector<double> err(n * 5);
  CHECK(hipMemcpy(err.data(), d_err, n * 5 * sizeof(double), hipMemcpyDeviceToHost));
  const char *names[5] = {"v_rcp_f64 seed", "v_rcp_f64 + 1 Newton", "v_rcp_f64 + 2 Newton", "v_rcp_f32 seed",
                          "v_rcp_f32 + 1 Newton(f64)"};
  for (int k = 0; k < 5; ++k) {
    double mx = 0;
    for (int i = 0; i < n; ++i) mx = fmax(mx, err[i * 5 + k]);
    printf("max |1 - h*r|  %-28s = %.3e  (2^%.1f)\n", names[k], mx, log2(mx));
  }
  return 0;
}
