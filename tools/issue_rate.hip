// issue_rate.hip — microbenchmark (development aid): how many cycles does a SIMD / the CU's
// scalar unit need per wave-instruction on this chip, at 1, 2, 4 and 8 resident waves per SIMD?
//   valu:  independent v_max_i32 / v_alignbit_b32 streams
//   salu:  independent s_add_u32 / s_and_b64 / s_bcnt1 streams
//   mixed: the two interleaved 1:1 (do the vector and the scalar unit overlap across waves?)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
// Every wave stamps s_memtime around its loop; the figure printed is the mean wave duration divided by the
// wave-instructions the SIMD (valu) or the CU (salu: one scalar unit per CU, 4 SIMDs x w waves) had to issue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 4096;

__global__ void __launch_bounds__(64) k_valu(unsigned long long* out, int* sink) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_max_i32 %0, %0, %8\n v_alignbit_b32 %1, %1, %1, 3\n v_max_i32 %2, %2, %8\n v_alignbit_b32 %3, %3, %3, 5\n"
        "v_max_i32 %4, %4, %8\n v_alignbit_b32 %5, %5, %5, 7\n v_max_i32 %6, %6, %8\n v_alignbit_b32 %7, %7, %7, 9\n"
        "v_max_i32 %0, %0, %8\n v_alignbit_b32 %1, %1, %1, 3\n v_max_i32 %2, %2, %8\n v_alignbit_b32 %3, %3, %3, 5\n"
        "v_max_i32 %4, %4, %8\n v_alignbit_b32 %5, %5, %5, 7\n v_max_i32 %6, %6, %8\n v_alignbit_b32 %7, %7, %7, 9\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) *sink = 1;
}

__global__ void __launch_bounds__(64) k_salu(unsigned long long* out, int* sink) {
  unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
  unsigned long long m0 = blockIdx.x * 77ull, m1 = m0 + 5, m2 = m0 + 9, m3 = m0 + 11;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "s_add_u32 %0, %0, 3\n s_and_b64 %4, %4, %5\n s_add_u32 %1, %1, 5\n s_or_b64 %5, %5, %6\n"
        "s_add_u32 %2, %2, 7\n s_lshl_b64 %6, %6, 1\n s_add_u32 %3, %3, 9\n s_xor_b64 %7, %7, %4\n"
        "s_add_u32 %0, %0, 3\n s_and_b64 %4, %4, %5\n s_add_u32 %1, %1, 5\n s_or_b64 %5, %5, %6\n"
        "s_add_u32 %2, %2, 7\n s_lshl_b64 %6, %6, 1\n s_add_u32 %3, %3, 9\n s_xor_b64 %7, %7, %4\n"
        : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if ((s0 ^ s1 ^ s2 ^ s3 ^ (unsigned)m0 ^ (unsigned)m1 ^ (unsigned)m2 ^ (unsigned)m3) == 0x12345u) *sink = 1;
}

__global__ void __launch_bounds__(64) k_mixed(unsigned long long* out, int* sink) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  unsigned s0 = blockIdx.x, s1 = s0 + 1;
  unsigned long long m0 = blockIdx.x * 77ull, m1 = m0 + 5;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_max_i32 %0, %0, %8\n s_add_u32 %4, %4, 3\n v_alignbit_b32 %1, %1, %1, 3\n s_and_b64 %6, %6, %7\n"
        "v_max_i32 %2, %2, %8\n s_add_u32 %5, %5, 5\n v_alignbit_b32 %3, %3, %3, 5\n s_or_b64 %7, %7, %6\n"
        "v_max_i32 %0, %0, %8\n s_add_u32 %4, %4, 3\n v_alignbit_b32 %1, %1, %1, 3\n s_and_b64 %6, %6, %7\n"
        "v_max_i32 %2, %2, %8\n s_add_u32 %5, %5, 5\n v_alignbit_b32 %3, %3, %3, 5\n s_or_b64 %7, %7, %6\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(m0), "+s"(m1) : "v"(i) : "scc");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if ((a0 ^ a1 ^ a2 ^ a3 ^ (int)s0 ^ (int)s1 ^ (int)m0 ^ (int)m1) == 0x12345) *sink = 1;
}

// LDS reads, three dwords per instruction-triple like the extension round of wfa_seg.hpp
__global__ void __launch_bounds__(64) k_lds(unsigned long long* out, int* sink) {
  __shared__ uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc = 0; int idx = threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc ^= lds[(idx + j * 37 + i) & 1023];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345u) *sink = 1;
}

template <typename K>
static void run(const char* name, K kern, int cus, double per_unit_div) {
  for (int w : {1, 2, 4, 8}) {
    const int grid = cus * 4 * w;
    unsigned long long* d; int* sink;
    CHECK(hipMalloc(&d, grid * sizeof(unsigned long long))); CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, sink);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), d, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v; mean /= grid;
    const double insts_wave = (double)ITERS * 16;
    // per SIMD: w waves share it; per CU: 4 w waves share the scalar unit
    printf("%-6s waves/SIMD=%d  wave duration %.0f ticks (%.3f ms wall)  ticks per wave-instruction: per wave %.2f, per SIMD %.2f, per CU %.3f\n",
           name, w, mean, ms, mean / insts_wave, mean / (insts_wave * w), mean / (insts_wave * w * 4));
    (void)per_unit_div;
    CHECK(hipFree(d)); CHECK(hipFree(sink)); CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  }
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s, %d CUs, clock %d kHz; readcyclecounter = s_memtime (100 MHz constant clock on some parts: see wall ms)\n",
         p.gcnArchName, p.multiProcessorCount, p.clockRate);
  run("valu", k_valu, p.multiProcessorCount, 1);
  run("salu", k_salu, p.multiProcessorCount, 4);
  run("mixed", k_mixed, p.multiProcessorCount, 1);
  run("lds", k_lds, p.multiProcessorCount, 1);
  return 0;
}
