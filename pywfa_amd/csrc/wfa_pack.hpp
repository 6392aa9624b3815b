// wfa_pack.hpp — device-side 2-bit packing of the ASCII batch (the layout the extend step reads).
//
// The reference compares raw bytes (wavefront_sequences.c:250) after pywfa upper-cases the strings
// (align.pyx:432,435).  A pair whose two sequences are pure ACGT can be compared on 2-bit codes
// (any injective code preserves equality); every other pair is flagged and aligned on its bytes.
#pragma once
#include <hip/hip_runtime.h>
#include "wfa_common.hpp"

namespace wfa {

// code = (c >> 1) & 3 : 'A'(0x41)->0  'C'(0x43)->1  'T'(0x54)->2  'G'(0x47)->3
__device__ __forceinline__ bool is_acgt(uint32_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// One wave per pair (grid-stride); lane j packs word j (16 bases) of the pattern, then of the text.
__global__ void __launch_bounds__(256)
wfa_pack_kernel(const uint8_t* __restrict__ bytes, const int64_t* __restrict__ p_boff,
                const int64_t* __restrict__ t_boff, const WfaPairMeta* __restrict__ meta, int64_t n,
                uint32_t* __restrict__ words, uint8_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t pair = wave; pair < n; pair += nwaves) {
    const WfaPairMeta pm = meta[pair];
    bool bad = false;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const uint8_t* src = bytes + (which ? t_boff[pair] : p_boff[pair]);
      const int len = which ? pm.tlen : pm.plen;
      uint32_t* dst = words + (which ? pm.t_woff : pm.p_woff);
      const int nw = (len + 15) >> 4;
      for (int w = lane; w < nw; w += 64) {
        uint32_t packed = 0;
        const int b0 = w << 4;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (b0 + j < len) {
            const uint32_t c = src[b0 + j];
            bad |= !is_acgt(c);
            packed |= ((c >> 1) & 3u) << (2 * j);
          }
        }
        dst[w] = packed;
      }
    }
    if (__any(bad) && lane == 0) flags[pair] = 1;
  }
}

}  // namespace wfa
