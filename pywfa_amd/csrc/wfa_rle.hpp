// wfa_rle.hpp — device-side result surface: run-length encode every pair's op string
// into pywfa's cigartuples (align.pyx:759-786, codes M=0 I=1 D=2 X=8) and derive the `locations`
// coordinates (align.pyx:788-833) — what the reference does per pair in interpreted Python.
// One wave per pair (grid-stride); run boundaries are found with wave ballots.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wfa {

__device__ __forceinline__ int op_code(uint32_t c) {
  return c == 'M' ? 0 : c == 'I' ? 1 : c == 'D' ? 2 : c == 'X' ? 8 : 3;
}

// pass 1 (runs == nullptr): counts[pair] = number of runs, locs[pair] = {pattern_start, pattern_end,
// text_start, text_end}.  pass 2: for run j of the pair, run_code[run_off[pair]+j] and
// run_start[run_off[pair]+j] (position of the run inside the op string; lengths are differences).
__global__ void __launch_bounds__(256)
wfa_rle_kernel(const uint8_t* __restrict__ ops, const int64_t* __restrict__ cigar_begin,
               const int32_t* __restrict__ cigar_len, const int32_t* __restrict__ plen_arr,
               const int32_t* __restrict__ tlen_arr, int64_t n, int32_t* __restrict__ counts,
               int32_t* __restrict__ locs, const int64_t* __restrict__ run_off,
               uint8_t* __restrict__ run_code, int32_t* __restrict__ run_start) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t pair = wave; pair < n; pair += nwaves) {
    const uint8_t* p = ops + cigar_begin[pair];
    const int len = cigar_len[pair];
    int nruns = 0;
    const int64_t out0 = run_off ? run_off[pair] : 0;
    for (int base = 0; base < len; base += 64) {
      const int i = base + lane;
      const bool in = i < len;
      const uint32_t c = in ? p[i] : 0u;
      const uint32_t prev = (in && i > 0) ? p[i - 1] : 0xFFu;
      const bool start = in && (c != prev);
      const unsigned long long bm = __ballot(start);
      if (run_off && start) {
        const int j = nruns + __builtin_popcountll(bm & ((1ull << lane) - 1ull));
        run_code[out0 + j] = (uint8_t)op_code(c);
        run_start[out0 + j] = i;
      }
      nruns += __builtin_popcountll(bm);
    }
    if (!run_off) {
      // locations (align.pyx:797-831 with a threshold of 1): ops before the first / after the last M
      int ps = 0, ts = 0, pe = plen_arr[pair], te = tlen_arr[pair];
      bool found = false;
      for (int base = 0; base < len && !found; base += 64) {
        const int i = base + lane;
        const uint32_t c = (i < len) ? p[i] : 0u;
        const unsigned long long mm = __ballot(c == 'M');
        const unsigned long long below = mm ? ((1ull << __builtin_ctzll(mm)) - 1ull) : ~0ull;
        ps += __builtin_popcountll(__ballot(c == 'D' || c == 'X') & below);
        ts += __builtin_popcountll(__ballot(c == 'I' || c == 'X') & below);
        found = mm != 0;
      }
      found = false;
      for (int top = len; top > 0 && !found; top -= 64) {
        const int i = top - 64 + lane;
        const uint32_t c = (i >= 0) ? p[i] : 0u;
        const unsigned long long mm = __ballot(c == 'M');
        const unsigned long long above = mm ? ~((2ull << (63 - __builtin_clzll(mm))) - 1ull) : ~0ull;
        pe -= __builtin_popcountll(__ballot(c == 'D' || c == 'X') & above);
        te -= __builtin_popcountll(__ballot(c == 'I' || c == 'X') & above);
        found = mm != 0;
      }
      if (lane == 0) {
        counts[pair] = nruns;
        const bool zero = (len == 0) || plen_arr[pair] == 0 || tlen_arr[pair] == 0;
        locs[4 * pair + 0] = zero ? 0 : ps; locs[4 * pair + 1] = zero ? 0 : pe;
        locs[4 * pair + 2] = zero ? 0 : ts; locs[4 * pair + 3] = zero ? 0 : te;
      }
    }
  }
}

}  // namespace wfa
