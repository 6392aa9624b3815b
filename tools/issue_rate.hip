// issue_rate.hip — microbenchmark (development aid): what does one wave64 vector instruction
// cost a SIMD of this chip, by instruction type and by resident waves per SIMD?
//
// One workgroup per CU of 256 x w threads (w = 1, 2, 4 waves on each of the CU's four SIMDs; 8 = two such workgroups of
// 1024 threads per CU), every wave runs the same long loop of 32 independent instructions of ONE type; the kernel is
// long enough (tens of ms) for the wall clock to be the measure.  v_fma_f32 is the yardstick: the CDNA4 guide gives it
// 2 cycles per wave64 instruction at >= 2 waves per SIMD (SIMD-32), so cycles(X) = 2 x time(X) / time(v_fma_f32 at w >= 2).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 60000;   // x 32 instructions per iteration

#define REP8(S) S S S S S S S S
#define BODY(INSTR0, INSTR1, INSTR2, INSTR3)                                                                     \
  for (int i = 0; i < ITERS; ++i) {                                                                              \
    asm volatile(REP8(INSTR0 "\n" INSTR1 "\n" INSTR2 "\n" INSTR3 "\n")                                           \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i), "v"(c0), "s"(m0) : "vcc");                           \
  }

template <int KIND>
__global__ void __launch_bounds__(1024) k_issue(int* sink) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  const int c0 = threadIdx.x * 3 + 1;
  const unsigned long long m0 = 0x5555555555555555ull;
  if (KIND == 0) { BODY("v_fma_f32 %0, %0, %5, %4", "v_fma_f32 %1, %1, %5, %4", "v_fma_f32 %2, %2, %5, %4", "v_fma_f32 %3, %3, %5, %4") }
  if (KIND == 1) { BODY("v_max_i32 %0, %0, %4", "v_max_i32 %1, %1, %4", "v_max_i32 %2, %2, %4", "v_max_i32 %3, %3, %4") }
  if (KIND == 2) { BODY("v_alignbit_b32 %0, %0, %5, 3", "v_alignbit_b32 %1, %1, %5, 5", "v_alignbit_b32 %2, %2, %5, 7", "v_alignbit_b32 %3, %3, %5, 9") }
  if (KIND == 3) { BODY("v_add_u32 %0, %0, %4", "v_add_u32 %1, %1, %4", "v_add_u32 %2, %2, %4", "v_add_u32 %3, %3, %4") }
  if (KIND == 4) { BODY("v_pk_max_i16 %0, %0, %5", "v_pk_max_i16 %1, %1, %5", "v_pk_add_i16 %2, %2, %5", "v_pk_add_i16 %3, %3, %5") }
  if (KIND == 5) { BODY("v_cndmask_b32 %0, %0, %5, %6", "v_cndmask_b32 %1, %1, %5, %6", "v_cndmask_b32 %2, %2, %5, %6", "v_cndmask_b32 %3, %3, %5, %6") }
  if (KIND == 6) { BODY("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf",
                        "v_mov_b32_dpp %2, %3 row_shl:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %0 row_shl:1 row_mask:0xf bank_mask:0xf") }
  if (KIND == 7) { BODY("v_ffbl_b32 %0, %0", "v_ffbl_b32 %1, %1", "v_xor_b32 %2, %2, %5", "v_min_u32 %3, %3, %5") }
  if (KIND == 8) { BODY("v_and_b32 %0, %0, %4", "v_or_b32 %1, %1, %4", "v_xor_b32 %2, %2, %4", "v_and_b32 %3, %3, %5") }
  if (KIND == 9) { BODY("v_lshlrev_b32 %0, 1, %0", "v_lshrrev_b32 %1, 1, %1", "v_ashrrev_i32 %2, 1, %2", "v_lshlrev_b32 %3, 3, %3") }
  if (KIND == 10) { BODY("v_sub_u32 %0, %0, %4", "v_sub_u32 %1, %1, %4", "v_add3_u32 %2, %2, %4, %5", "v_lshl_add_u32 %3, %3, 2, %4") }
  if (KIND == 11) { BODY("v_cmp_gt_i32 vcc, %0, %4", "v_cndmask_b32 %1, %1, %5, vcc", "v_cmp_lt_i32 vcc, %2, %4", "v_cndmask_b32 %3, %3, %5, vcc") }
  if (KIND == 12) { BODY("v_mov_b32 %0, %1", "v_mov_b32 %1, %2", "v_mov_b32 %2, %3", "v_mov_b32 %3, %0") }
  if (KIND == 13) { BODY("v_bfe_u32 %0, %0, 1, 31", "v_bfi_b32 %1, %4, %1, %5", "v_and_or_b32 %2, %2, %4, %5", "v_perm_b32 %3, %3, %5, %4") }
  if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345) *sink = 1;
}

template <int KIND>
static double run_kind(int cus, int w) {
  const int threads = (w == 8) ? 1024 : 256 * w;
  const int grid = (w == 8) ? 2 * cus : cus;
  int* sink; CHECK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(threads), 0, 0, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(threads), 0, 0, sink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipFree(sink)); CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return ms;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs; %d x 32 wave-instructions per wave; per SIMD = w waves\n", p.gcnArchName, cus, ITERS);
  const char* names[14] = {"v_fma_f32", "v_max_i32", "v_alignbit_b32", "v_add_u32", "v_pk_max/add_i16", "v_cndmask_b32(sgpr mask)", "v_mov_b32_dpp", "ffbl/xor/min mix", "v_and/or/xor_b32", "v_lsh*_b32", "v_sub/add3/lshl_add", "v_cmp + v_cndmask(vcc)", "v_mov_b32", "v_bfe/bfi/and_or/perm"};
  double ms[14][4];
  const int ws[4] = {1, 2, 4, 8};
  for (int wi = 0; wi < 4; ++wi) {
    const int w = ws[wi];
    ms[0][wi] = run_kind<0>(cus, w); ms[1][wi] = run_kind<1>(cus, w); ms[2][wi] = run_kind<2>(cus, w); ms[3][wi] = run_kind<3>(cus, w);
    ms[4][wi] = run_kind<4>(cus, w); ms[5][wi] = run_kind<5>(cus, w); ms[6][wi] = run_kind<6>(cus, w); ms[7][wi] = run_kind<7>(cus, w);
    ms[8][wi] = run_kind<8>(cus, w); ms[9][wi] = run_kind<9>(cus, w); ms[10][wi] = run_kind<10>(cus, w); ms[11][wi] = run_kind<11>(cus, w);
    ms[12][wi] = run_kind<12>(cus, w); ms[13][wi] = run_kind<13>(cus, w);
  }
  // yardstick: v_fma_f32 at 4 waves per SIMD = 2 cycles per wave-instruction per SIMD
  const double insts4 = (double)ITERS * 32 * 4;
  const double clock_ghz = insts4 * 2.0 / (ms[0][2] * 1e6);
  printf("implied shader clock from v_fma_f32 @ 4 waves/SIMD = 2 cycles: %.3f GHz\n", clock_ghz);
  for (int k = 0; k < 14; ++k) {
    printf("%-26s", names[k]);
    for (int wi = 0; wi < 4; ++wi) {
      const double insts = (double)ITERS * 32 * ws[wi];
      printf("  w=%d: %7.3f ms = %5.3f ns = %5.2f cyc /instr/SIMD", ws[wi], ms[k][wi], ms[k][wi] * 1e6 / insts, ms[k][wi] * 1e6 * clock_ghz / insts);
    }
    printf("\n");
  }
  return 0;
}
