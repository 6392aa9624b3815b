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

// A wave packs 64 >> log2slots pairs per round: lane -> (pair = lane >> log2slots, word = lane & (slots - 1)),
// the words of a pair counted through pattern then text (they are contiguous in `words`).  A lane reads its 16
// bases as five aligned dwords + v_alignbyte, packs four bases per dword with shifts / ors, and checks them
// against the alphabet with one v_perm_b32 (the 2-bit codes select the expected letter from 'G''T''C''A').
__global__ void __launch_bounds__(256)
wfa_pack_kernel(const uint8_t* __restrict__ bytes, const int64_t* __restrict__ p_boff,
                const int64_t* __restrict__ t_boff, const WfaPairMeta* __restrict__ meta, int64_t n,
                uint32_t* __restrict__ words, uint8_t* __restrict__ flags, int log2slots,
                uint32_t* __restrict__ any_flag) {
  const int lane = threadIdx.x & 63;
  const int slots = 1 << log2slots, group = 64 >> log2slots;
  const int sub = lane >> log2slots, wl = lane & (slots - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p0 = wave * group; p0 < n; p0 += nwaves * group) {
    const int64_t pair = p0 + sub;
    bool bad = false;
    if (pair < n) {
      const WfaPairMeta pm = meta[pair];
      const int nwp = (pm.plen + 15) >> 4, ntot = nwp + ((pm.tlen + 15) >> 4);
      const int64_t pb = p_boff[pair], tb = t_boff[pair];
      for (int w = wl; w < ntot; w += slots) {
        const bool is_text = w >= nwp;
        const int ws = is_text ? w - nwp : w;
        const int nvalid = min(16, (is_text ? pm.tlen : pm.plen) - ws * 16);
        const uintptr_t addr = (uintptr_t)(bytes + (is_text ? tb : pb) + (int64_t)ws * 16);
        const uint32_t* a0 = (const uint32_t*)(addr & ~(uintptr_t)3);  // (the blob allocation is padded by 64 bytes)
        const uint32_t sh = (uint32_t)(addr & 3);
        uint32_t d[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) d[j] = a0[j];
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cnt = min(max(nvalid - 4 * j, 0), 4);
          const uint32_t m = (cnt >= 4) ? 0xffffffffu : ((1u << (8 * cnt)) - 1u);
          const uint32_t x = __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh) & m;
          const uint32_t codes = (x >> 1) & 0x03030303u;
          const uint32_t t = codes | (codes >> 6);
          packed |= ((t | (t >> 12)) & 0xffu) << (8 * j);
          bad |= (__builtin_amdgcn_perm(0u, 0x47544341u, codes) & m) != x;
        }
        words[pm.p_woff + w] = packed;
      }
    }
    const unsigned long long b = __ballot(bad);
    const unsigned long long field = (slots == 64) ? b : ((b >> (sub * slots)) & ((1ull << slots) - 1ull));
    if (field != 0ull && wl == 0 && pair < n) flags[pair] = 1;
    if (b != 0ull && lane == 0) atomicOr(any_flag, 1u);   // the host fetches the per-pair flags only if some pair was flagged
  }
}

// 2-bit input (wfa_hip_*_packed2bits): the caller's reads are already 2 bits per base in the reference's packed form
// (R/wavefront_sequences.c:102-139: four bases per byte, base j of a byte in bits 2j..2j+1, A 0 / C 1 / G 2 / T 3), a
// sequence at any byte offset.  The kernels read whole words that start a sequence and the code (c >> 1) & 3 of the ASCII
// letter (A 0 / C 1 / T 2 / G 3), so every word is re-based and re-coded: code' = code ^ (code >> 1) swaps 2 and 3.
// 16 lanes per pair, four pairs per wave.
__global__ void __launch_bounds__(256)
wfa_repack2_kernel(const uint8_t* __restrict__ bytes, const int64_t* __restrict__ p_boff, const int64_t* __restrict__ t_boff,
                   const WfaPairMeta* __restrict__ meta, int64_t n, uint32_t* __restrict__ words) {
  const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> 4;
  const int wl = threadIdx.x & 15;
  for (int64_t pair = group; pair < n; pair += ngroups) {
    const WfaPairMeta pm = meta[pair];
    const int nwp = (pm.plen + 15) >> 4, ntot = nwp + ((pm.tlen + 15) >> 4);
    const int64_t pb = p_boff[pair], tb = t_boff[pair];
    for (int w = wl; w < ntot; w += 16) {
      const bool is_text = w >= nwp;
      const int ws = is_text ? w - nwp : w;
      const int len = is_text ? pm.tlen : pm.plen;
      const int nbases = min(16, len - ws * 16);            // bases of this word
      const int nbytes = (nbases + 3) >> 2;                 // bytes of the caller's sequence that hold them
      const uint8_t* src = bytes + (is_text ? tb : pb) + (int64_t)ws * 4;
      uint32_t v = 0;
      for (int j = 0; j < nbytes; ++j) v |= (uint32_t)src[j] << (8 * j);
      v ^= (v >> 1) & 0x55555555u;
      if (nbases < 16) v &= (1u << (2 * nbases)) - 1u;      // zero beyond the end, like the other packers
      words[pm.p_woff + w] = v;
    }
  }
}

// Host-packed upload with 16-bit lengths (round 6): the pinned ring carries 4 B per pair {plen, tlen} instead of the 16 B WfaPairMeta;
// the word offsets are rebuilt here, one workgroup per piece of the upload (a piece knows its first pair and its first word: the
// host's pass 1 summed the words of every 64-pair block): thread t takes a contiguous run of the piece's pairs, the runs' word
// counts are scanned through LDS, then every thread writes its pairs' metadata.
struct WfaPieceDesc { long long lo, hi; unsigned long long wlo; };
__global__ void __launch_bounds__(256)
wfa_meta_from_len16_kernel(const WfaPieceDesc* __restrict__ pieces, const uint32_t* __restrict__ len16, WfaPairMeta* __restrict__ meta) {
  __shared__ uint32_t part[256];
  const WfaPieceDesc pc = pieces[blockIdx.x];
  const long long npairs = pc.hi - pc.lo;
  const long long per = (npairs + 255) / 256;
  const long long a = pc.lo + (long long)threadIdx.x * per, b = (a + per < pc.hi) ? a + per : pc.hi;
  uint32_t sum = 0;
  for (long long i = a; i < b; ++i) { const uint32_t v = len16[i]; sum += (((v & 0xffffu) + 15u) >> 4) + (((v >> 16) + 15u) >> 4); }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {   // inclusive scan
    const uint32_t add = (threadIdx.x >= (unsigned)d) ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t w = (uint32_t)pc.wlo + part[threadIdx.x] - sum;
  for (long long i = a; i < b; ++i) {
    const uint32_t v = len16[i];
    const uint32_t pl = v & 0xffffu, tl = v >> 16;
    WfaPairMeta m;
    m.p_woff = w; w += (pl + 15u) >> 4;
    m.t_woff = w; w += (tl + 15u) >> 4;
    m.plen = (int)pl; m.tlen = (int)tl;
    meta[i] = m;
  }
}

// Host-packed upload: byte offsets and flags of the pairs that hold a letter outside ACGT (their bytes sit in a compact blob).
__global__ void __launch_bounds__(256)
wfa_flag_scatter_kernel(const uint32_t* __restrict__ ids, const int64_t* __restrict__ pb, const int64_t* __restrict__ tb, uint32_t nb,
                        int64_t* __restrict__ p_boff, int64_t* __restrict__ t_boff, uint8_t* __restrict__ flags) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  const uint32_t id = ids[j];
  p_boff[id] = pb[j]; t_boff[id] = tb[j]; flags[id] = 1;
}

}  // namespace wfa
