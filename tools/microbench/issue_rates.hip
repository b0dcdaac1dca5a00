// CAUTION: run it with instruction names on the command line (profiles/r21d_issue_rates_selected.txt shows a call).  Without
// names it runs every entry, and on the box of 2026-10-05 one of the entries added last (between `v_lshlrev_b32 v,v` and
// `v_cvt_u32_f32`; `v_and_b32 literal`, `v_and_b32 sgpr`, `v_fma_f32 4.0,v,v`, `v_mul_f32 4.0`, `v_sub_f32`, `v_min_f32` are cleared) did
// not come back in the sixteen-wave mode until `timeout` killed the process.
// Issue cost of the vector instructions the fusion kernel is made of, one wave on one SIMD, independent instructions:
// cycles (s_memtime) per instruction, relative to v_add_f32.  Build: hipcc --offload-arch=gfx950 -O2 -o issue_rates issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define BODY(name, INSTR)                                                                                         \
  __global__ void name(uint64_t *out, int iters) {                                                               \
    uint64_t t0 = __builtin_readcyclecounter();                                                                   \
    for (int i = 0; i < iters; ++i) {                                                                             \
      asm volatile(REP8(INSTR) REP8(INSTR) REP8(INSTR) REP8(INSTR)                                                \
                   :                                                                                              \
                   :                                                                                              \
                   : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "vcc", "s20", "s21");  \
    }                                                                                                             \
    uint64_t t1 = __builtin_readcyclecounter();                                                                   \
    if ((threadIdx.x & 63) == 0) {                                                                                \
      out[2 * (threadIdx.x >> 6)] = t0;                                                                           \
      out[2 * (threadIdx.x >> 6) + 1] = t1;                                                                       \
    }                                                                                                             \
  }

BODY(k_add_f32, "v_add_f32 v10, v12, v13\n\t")
BODY(k_fma_f32, "v_fma_f32 v10, v12, v13, v14\n\t")
BODY(k_pk_fma_f32, "v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[16:17]\n\t")
BODY(k_pk_add_f32, "v_pk_add_f32 v[10:11], v[12:13], v[14:15]\n\t")
BODY(k_rcp_f32, "v_rcp_f32 v10, v12\n\t")
BODY(k_max_f32, "v_max_f32 v10, |v12|, |v13|\n\t")
BODY(k_cmp_f32, "v_cmp_lt_f32 vcc, v12, v13\n\t")
BODY(k_bfe_u32, "v_bfe_u32 v10, v12, v13, 1\n\t")
BODY(k_lshl_b32, "v_lshlrev_b32 v10, 2, v12\n\t")
BODY(k_and_b32, "v_and_b32 v10, v12, v13\n\t")
BODY(k_cvt_f64_u32, "v_cvt_f64_u32 v[10:11], v12\n\t")
BODY(k_cvt_f64_f32, "v_cvt_f64_f32 v[10:11], v12\n\t")
BODY(k_add_f64, "v_add_f64 v[10:11], v[12:13], v[14:15]\n\t")
BODY(k_fma_f64, "v_fma_f64 v[10:11], v[12:13], v[14:15], v[16:17]\n\t")
BODY(k_mul_f64, "v_mul_f64 v[10:11], v[12:13], v[14:15]\n\t")
BODY(k_cmp_f64, "v_cmp_lt_f64 vcc, v[12:13], v[14:15]\n\t")
BODY(k_rcp_f64, "v_rcp_f64 v[10:11], v[12:13]\n\t")
BODY(k_rndne_f64, "v_rndne_f64 v[10:11], v[12:13]\n\t")
BODY(k_cvt_i32_f64, "v_cvt_i32_f64 v10, v[12:13]\n\t")
BODY(k_mad_u32_u24, "v_mad_u32_u24 v10, v12, v13, v14\n\t")
BODY(k_readlane, "v_readlane_b32 s20, v12, 3\n\t")
BODY(k_mov_b64, "v_mov_b64 v[10:11], v[12:13]\n\t")
BODY(k_bpermute, "ds_bpermute_b32 v10, v12, v13\n\t")
BODY(k_snop, "s_nop 0\n\t")
BODY(k_cndmask, "v_cndmask_b32 v10, v12, v13, vcc\n\t")

BODY(kx_0, "v_mul_f32 v10, v12, v13\n\t")
BODY(kx_1, "v_fmac_f32 v10, v12, v13\n\t")
BODY(kx_2, "v_fma_f32 v10, s20, v13, v14\n\t")
BODY(kx_3, "v_max_f32 v10, v12, v13\n\t")
BODY(kx_4, "v_add_u32 v10, v12, v13\n\t")
BODY(kx_5, "v_sub_u32 v10, v12, v13\n\t")
BODY(kx_6, "v_or_b32 v10, v12, v13\n\t")
BODY(kx_7, "v_mov_b32 v10, v12\n\t")
BODY(kx_8, "v_lshrrev_b32 v10, 2, v12\n\t")
BODY(kx_9, "v_lshl_add_u32 v10, v12, 2, v13\n\t")
BODY(kx_10, "v_add3_u32 v10, v12, v13, v14\n\t")
BODY(kx_11, "v_and_or_b32 v10, v12, v13, v14\n\t")
BODY(kx_12, "v_bfe_i32 v10, v12, v13, 1\n\t")
BODY(kx_13, "v_bfm_b32 v10, v12, 20\n\t")
BODY(kx_14, "v_alignbit_b32 v10, v12, v13, v14\n\t")
BODY(kx_15, "v_mul_u32_u24 v10, v12, v13\n\t")
BODY(kx_16, "v_mul_lo_u32 v10, v12, v13\n\t")
BODY(kx_17, "v_cvt_f32_i32 v10, v12\n\t")
BODY(kx_18, "v_cvt_i32_f32 v10, v12\n\t")
BODY(kx_19, "v_rndne_f32 v10, v12\n\t")
BODY(kx_20, "v_pk_mul_f32 v[10:11], v[12:13], v[14:15]\n\t")
BODY(kx_21, "v_pk_fma_f32 v[10:11], v[12:13], v[14:15], s[20:21]\n\t")
BODY(kx_22, "v_cmp_lt_u32 vcc, v12, v13\n\t")
BODY(kx_23, "v_cmp_lt_f32 s[20:21], v12, v13\n\t")
BODY(kx_24, "v_cmpx_lt_f64 vcc, v[12:13], v[14:15]\n\t")
BODY(kx_25, "v_cndmask_b32 v10, v12, v13, s[20:21]\n\t")
BODY(kx_26, "v_add_f32_dpp v10, v12, v13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
BODY(kx_27, "v_mov_b32_dpp v10, v12 row_shr:1 row_mask:0xf bank_mask:0xf\n\t")
BODY(kx_28, "v_perm_b32 v10, v12, v13, v14\n\t")
BODY(kx_29, "v_cvt_f64_i32 v[10:11], v12\n\t")
BODY(kx_30, "v_ldexp_f64 v[10:11], v[12:13], v14\n\t")
BODY(kx_31, "v_max_f64 v[10:11], |v[12:13]|, |v[14:15]|\n\t")
BODY(kx_32, "v_cvt_f32_f64 v10, v[12:13]\n\t")
BODY(kx_33, "v_rcp_iflag_f32 v10, v12\n\t")
BODY(kx_34, "v_med3_f32 v10, v12, v13, v14\n\t")
BODY(kx_35, "v_max3_f32 v10, v12, v13, v14\n\t")
BODY(kx_36, "v_xad_u32 v10, v12, v13, v14\n\t")
BODY(kx_37, "v_mad_u32_u24 v10, v12, v13, v14\n\t")
BODY(kx_38, "v_mad_i32_i24 v10, v12, v13, v14\n\t")
BODY(kx_39, "s_mov_b64 s[20:21], s[22:23]\n\t")
BODY(kx_40, "s_and_b64 s[20:21], s[22:23], s[24:25]\n\t")
BODY(kx_41, "ds_swizzle_b32 v10, v12 offset:swizzle(SWAP,1)\n\t")
BODY(ky_0, "v_and_b32 v10, 0x3ff00000, v13\n\t")
BODY(ky_1, "v_and_b32 v10, s20, v13\n\t")
BODY(ky_2, "v_fma_f32 v10, 4.0, v13, v14\n\t")
BODY(ky_3, "v_lshlrev_b32 v10, v13, v12\n\t")
BODY(ky_4, "v_mul_f32 v10, 4.0, v12\n\t")
BODY(ky_5, "v_add_f32 v10, s20, v12\n\t")
BODY(ky_6, "v_ashrrev_i32 v10, 31, v12\n\t")
BODY(ky_7, "v_fmac_f32 v10, s20, v13\n\t")
BODY(ky_8, "v_fmamk_f32 v10, v12, 0x40800000, v13\n\t")
BODY(ky_9, "v_fmaak_f32 v10, v12, v13, 0x4b400000\n\t")
BODY(ky_10, "v_sub_f32 v10, v12, v13\n\t")
BODY(ky_11, "v_min_f32 v10, v12, v13\n\t")
BODY(ky_12, "v_xor_b32 v10, v12, v13\n\t")
BODY(ky_13, "v_cvt_u32_f32 v10, v12\n\t")
typedef void (*kern_t)(uint64_t *, int);
struct Entry { const char *name; kern_t k; };

#include <cstring>
int main(int argc, char **argv) {
  Entry list[] = {{"v_add_f32", k_add_f32}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_add_f32", k_pk_add_f32},
                  {"v_rcp_f32", k_rcp_f32}, {"v_max_f32", k_max_f32}, {"v_cmp_lt_f32", k_cmp_f32}, {"v_bfe_u32", k_bfe_u32},
                  {"v_lshlrev_b32", k_lshl_b32}, {"v_and_b32", k_and_b32}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_f64_f32", k_cvt_f64_f32},
                  {"v_add_f64", k_add_f64}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_cmp_lt_f64", k_cmp_f64},
                  {"v_rcp_f64", k_rcp_f64}, {"v_rndne_f64", k_rndne_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_mad_u32_u24", k_mad_u32_u24},
                  {"v_readlane_b32", k_readlane}, {"v_mov_b64", k_mov_b64}, {"ds_bpermute_b32", k_bpermute}, {"s_nop 0", k_snop},
                  {"v_cndmask_b32", k_cndmask}, {"v_mul_f32", kx_0}, {"v_fmac_f32", kx_1}, {"v_fma_f32 s,v,v", kx_2}, {"v_max_f32 (vop2)", kx_3}, {"v_add_u32", kx_4}, {"v_sub_u32", kx_5}, {"v_or_b32", kx_6}, {"v_mov_b32", kx_7}, {"v_lshrrev_b32", kx_8}, {"v_lshl_add_u32", kx_9}, {"v_add3_u32", kx_10}, {"v_and_or_b32", kx_11}, {"v_bfe_i32", kx_12}, {"v_bfm_b32", kx_13}, {"v_alignbit_b32", kx_14}, {"v_mul_u32_u24", kx_15}, {"v_mul_lo_u32", kx_16}, {"v_cvt_f32_i32", kx_17}, {"v_cvt_i32_f32", kx_18}, {"v_rndne_f32", kx_19}, {"v_pk_mul_f32", kx_20}, {"v_pk_fma_f32 s", kx_21}, {"v_cmp_lt_u32", kx_22}, {"v_cmp_lt_f32 e64", kx_23}, {"v_cmpx_lt_f64", kx_24}, {"v_cndmask e64", kx_25}, {"v_add_f32 dpp", kx_26}, {"v_mov_b32 dpp", kx_27}, {"v_perm_b32", kx_28}, {"v_cvt_f64_i32", kx_29}, {"v_ldexp_f64", kx_30}, {"v_max_f64", kx_31}, {"v_cvt_f32_f64", kx_32}, {"v_rcp_iflag_f32", kx_33}, {"v_med3_f32", kx_34}, {"v_max3_f32", kx_35}, {"v_xad_u32", kx_36}, {"v_mad_u32_u24 vvv", kx_37}, {"v_mad_i32_i24", kx_38}, {"s_mov_b64", kx_39}, {"s_and_b64", kx_40}, {"ds_swizzle", kx_41}, {"v_and_b32 literal", ky_0}, {"v_and_b32 sgpr", ky_1}, {"v_fma_f32 4.0,v,v", ky_2}, {"v_lshlrev_b32 v,v", ky_3}, {"v_mul_f32 4.0", ky_4}, {"v_add_f32 sgpr", ky_5}, {"v_ashrrev_i32", ky_6}, {"v_fmac_f32 sgpr", ky_7}, {"v_fmamk_f32", ky_8}, {"v_fmaak_f32", ky_9}, {"v_sub_f32", ky_10}, {"v_min_f32", ky_11}, {"v_xor_b32", ky_12}, {"v_cvt_u32_f32", ky_13}};
  uint64_t *d;
  hipMalloc(&d, 4096 * sizeof(uint64_t));
  const int iters = 2000;
  // (a) one wave alone on its SIMD; (b) four waves per SIMD on every SIMD of a CU (256 threads x 4 workgroups... one WG of 1024)
  for (int mode = 0; mode < 2; ++mode) {
    printf("%s\n", mode == 0 ? "one wave on the SIMD: cycles per instruction" : "sixteen waves per CU (four per SIMD): first start to last end / instructions issued per SIMD");
    double base = 0;
    for (auto &e : list) {
      if (argc > 1) {  // only the instructions named on the command line (and the unit, v_add_f32)
        bool wanted = std::strcmp(e.name, "v_add_f32") == 0;
        for (int a = 1; a < argc; ++a) wanted = wanted || std::strcmp(e.name, argv[a]) == 0;
        if (!wanted) continue;
      }
      const int threads = mode == 0 ? 64 : 1024;
      hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, 10);
      hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, iters);
      hipDeviceSynchronize();
      uint64_t h[32];
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      const int waves = threads / 64;
      uint64_t lo = ~0ull, hi = 0;
      for (int w = 0; w < waves; ++w) lo = h[2 * w] < lo ? h[2 * w] : lo, hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi;
      // first start to last end over the instructions ONE SIMD issued (a quarter of the waves)
      const double per = (double)(hi - lo) / ((double)iters * 32.0 * (waves >= 4 ? waves / 4 : 1));
      if (base == 0) base = per;
      printf("  %-18s %8.3f counter ticks  (%.2f x v_add_f32)\n", e.name, per, per / base);
      fflush(stdout);
    }
  }
  hipFree(d);
  return 0;
}
