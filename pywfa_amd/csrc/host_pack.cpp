// host_pack.cpp — host-side 2-bit packing of ASCII sequences (the layout wfa_pack.hpp produces on the device), so that a
// large batch crosses PCIe as 76 B per 150 bp pair instead of 300 B.  Host code only (g++), one
// translation unit with three forms chosen once at run time: AVX-512BW (64 bases per round), AVX2 (32) and plain C.
//
// Layout (wfa_pack.hpp): word w of a sequence holds bases 16 w .. 16 w + 15, base j in bits 2 j .. 2 j + 1, code
// (c >> 1) & 3 ('A' 0, 'C' 1, 'T' 2, 'G' 3), zero beyond the end.  A sequence with any byte outside ACGT is reported
// (the pair is aligned on its bytes: the reference compares raw bytes, wavefront_sequences.c:250).
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

namespace wfa {

static inline bool is_acgt(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

static bool pack_scalar(const uint8_t* s, int len, uint32_t* out) {
  bool bad = false;
  const int nw = (len + 15) >> 4;
  for (int w = 0; w < nw; ++w) {
    uint32_t v = 0;
    const int cnt = (len - 16 * w < 16) ? len - 16 * w : 16;
    for (int j = 0; j < cnt; ++j) {
      const uint8_t c = s[16 * w + j];
      bad |= !is_acgt(c);
      v |= (uint32_t)((c >> 1) & 3) << (2 * j);
    }
    out[w] = v;
  }
  return bad;
}

__attribute__((target("avx2"))) static inline __m256i pack32_avx2(__m256i x, uint32_t* bad_mask) {
  const __m256i codes = _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(3));
  const __m256i lut = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  *bad_mask = ~(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(lut, codes), x));
  const __m256i t = _mm256_maddubs_epi16(codes, _mm256_set1_epi16(0x0401));     // c0 + 4 c1 per 16-bit lane
  return _mm256_madd_epi16(t, _mm256_set1_epi32(0x00100001));                   // + 16 (c2 + 4 c3): one byte per 32-bit lane
}

__attribute__((target("avx2"))) static bool pack_avx2(const uint8_t* s, int len, uint32_t* out) {
  uint32_t bad = 0;
  int i = 0, w = 0;
  for (; i + 32 <= len; i += 32, w += 2) {
    uint32_t bm;
    const __m256i q = pack32_avx2(_mm256_loadu_si256((const __m256i*)(s + i)), &bm);
    bad |= bm;
    const __m256i p16 = _mm256_packus_epi32(q, q);
    const __m256i p8 = _mm256_packus_epi16(p16, p16);
    out[w] = (uint32_t)_mm256_extract_epi32(p8, 0);
    out[w + 1] = (uint32_t)_mm256_extract_epi32(p8, 4);
  }
  const int rem = len - i;
  if (rem > 0) {
    alignas(32) uint8_t tmp[32];
    memset(tmp, 0, 32);
    memcpy(tmp, s + i, (size_t)rem);
    uint32_t bm;
    const __m256i q = pack32_avx2(_mm256_load_si256((const __m256i*)tmp), &bm);
    bad |= bm & ((rem >= 32) ? 0xffffffffu : ((1u << rem) - 1u));
    const __m256i p16 = _mm256_packus_epi32(q, q);
    const __m256i p8 = _mm256_packus_epi16(p16, p16);
    // (a zero byte packs to code 0: the words are zero beyond the end)
    out[w] = (uint32_t)_mm256_extract_epi32(p8, 0);
    if (rem > 16) out[w + 1] = (uint32_t)_mm256_extract_epi32(p8, 4);
  }
  return bad != 0;
}

__attribute__((target("avx512f,avx512bw"))) static inline __m128i pack64_avx512(__m512i x, uint64_t* bad_mask) {
  const __m512i codes = _mm512_and_si512(_mm512_srli_epi16(x, 1), _mm512_set1_epi8(3));
  const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0));
  *bad_mask = _mm512_cmpneq_epi8_mask(_mm512_shuffle_epi8(lut, codes), x);
  const __m512i t = _mm512_maddubs_epi16(codes, _mm512_set1_epi16(0x0401));
  return _mm512_cvtepi32_epi8(_mm512_madd_epi16(t, _mm512_set1_epi32(0x00100001)));
}

__attribute__((target("avx512f,avx512bw"))) static bool pack_avx512(const uint8_t* s, int len, uint32_t* out) {
  uint64_t bad = 0;
  int i = 0, w = 0;
  for (; i + 64 <= len; i += 64, w += 4) {
    uint64_t bm;
    const __m128i q = pack64_avx512(_mm512_loadu_si512((const void*)(s + i)), &bm);
    bad |= bm;
    _mm_storeu_si128((__m128i*)(out + w), q);
  }
  const int rem = len - i;
  if (rem > 0) {
    // masked load: bytes beyond the end are not touched and read as zero
    const __mmask64 k = (rem >= 64) ? ~0ull : ((1ull << rem) - 1ull);
    uint64_t bm;
    const __m128i q = pack64_avx512(_mm512_maskz_loadu_epi8(k, (const void*)(s + i)), &bm);
    bad |= bm & k;
    alignas(16) uint32_t tmp[4];
    _mm_store_si128((__m128i*)tmp, q);
    const int nw = (rem + 15) >> 4;
    for (int j = 0; j < nw; ++j) out[w + j] = tmp[j];
  }
  return bad != 0;
}

typedef bool (*pack_fn)(const uint8_t*, int, uint32_t*);

static pack_fn pick() {
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f")) return pack_avx512;
  if (__builtin_cpu_supports("avx2")) return pack_avx2;
  return pack_scalar;
}

// form: -1 the best this CPU has; 0 plain C, 1 AVX2, 2 AVX-512BW (tests; a form the CPU lacks falls back to plain C)
bool host_pack_seq(const uint8_t* s, int len, uint32_t* out, int form) {
  static const pack_fn best = pick();
  if (len <= 0) return false;
  if (form < 0) return best(s, len, out);
  __builtin_cpu_init();
  if (form == 2 && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f")) return pack_avx512(s, len, out);
  if (form == 1 && __builtin_cpu_supports("avx2")) return pack_avx2(s, len, out);
  return pack_scalar(s, len, out);
}

// 2-bit input (the reference's packed form, R/wavefront_sequences.c:102-139: four bases per byte, base j of a byte in bits 2 j .. 2 j + 1,
// A 0 / C 1 / G 2 / T 3) -> the device layout above: the same bit positions, G and T swapped (code ^ (code >> 1)), zero beyond the end.
// What wfa_repack2_kernel does on the device, done by the upload workers on their way into the pinned ring (round 6: the 2-bit
// entry then sends the same words + 4 B of lengths per pair as the ASCII entry instead of the caller's bytes + 48 B per pair).
void host_repack2_seq(const uint8_t* s, int len, uint32_t* out) {
  if (len <= 0) return;
  const int nw = (len + 15) >> 4, nbytes = (len + 3) >> 2;
  out[nw - 1] = 0u;                                  // (the last word is only partly covered by the caller's bytes)
  memcpy(out, s, (size_t)nbytes);
  const int tail = len & 15;
  for (int w = 0; w < nw; ++w) { const uint32_t v = out[w]; out[w] = v ^ ((v >> 1) & 0x55555555u); }
  if (tail) out[nw - 1] &= (1u << (2 * tail)) - 1u;
}

}  // namespace wfa
