/*
 * synth.c — deterministic synthetic read-pair generator for bench.py and the parity tests
 * (host-side utility; not part of the alignment path).  Same stream as the NumPy
 * implementation in pywfa_amd/datagen.py (tests/test_datagen.py checks they agree).
 *
 * Error model of SURVEY.md §8(d): pattern i.i.d. uniform ACGT of exact length L; walking the
 * pattern, per base u~U[0,1): u<e/3 substitute by a different base; u<2e/3 insert one uniform
 * base before it; u<e delete it; else copy.  Randomness is counter-based (one splitmix64
 * hash per (pair, position)) so that any slice of the stream can be produced in parallel.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint64_t synth_hash(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + (idx + 1) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* thresholds on the 24-bit uniform */
static inline void synth_thresholds(double error, uint32_t* t_sub, uint32_t* t_ins, uint32_t* t_del) {
  *t_sub = (uint32_t)(error / 3.0 * 16777216.0 + 0.5);
  *t_ins = (uint32_t)(2.0 * error / 3.0 * 16777216.0 + 0.5);
  *t_del = (uint32_t)(error * 16777216.0 + 0.5);
}

/* Pass 1: text length of pairs [first, first+n). */
void wfa_synth_text_lengths(uint64_t seed, int64_t first, int64_t n, int32_t length, double error,
                            int32_t* t_len) {
  uint32_t t_sub, t_ins, t_del;
  synth_thresholds(error, &t_sub, &t_ins, &t_del);
  int64_t i;
#pragma omp parallel for schedule(static)
  for (i = 0; i < n; ++i) {
    const uint64_t base_idx = (uint64_t)(first + i) * (uint64_t)length;
    int32_t tl = 0, j;
    for (j = 0; j < length; ++j) {
      const uint32_t u = (uint32_t)(synth_hash(seed, base_idx + j) >> 8) & 0xFFFFFFu;
      tl += 1 + ((u >= t_sub) & (u < t_ins)) - ((u >= t_ins) & (u < t_del));
    }
    t_len[i] = tl;
  }
}

/* Pass 2: write ASCII patterns (n*length bytes at pat) and texts (at text + t_off[i]). */
void wfa_synth_fill(uint64_t seed, int64_t first, int64_t n, int32_t length, double error,
                    uint8_t* pat, uint8_t* text, const int64_t* t_off) {
  static const char acgt[4] = {'A', 'C', 'G', 'T'};
  uint32_t t_sub, t_ins, t_del;
  synth_thresholds(error, &t_sub, &t_ins, &t_del);
  int64_t i;
#pragma omp parallel for schedule(static)
  for (i = 0; i < n; ++i) {
    const uint64_t base_idx = (uint64_t)(first + i) * (uint64_t)length;
    uint8_t* p = pat + i * (int64_t)length;
    uint8_t* t = text + t_off[i];
    int32_t j;
    for (j = 0; j < length; ++j) {
      const uint64_t z = synth_hash(seed, base_idx + j);
      const uint32_t b = (uint32_t)z & 3u;
      const uint32_t u = (uint32_t)(z >> 8) & 0xFFFFFFu;
      p[j] = (uint8_t)acgt[b];
      if (u < t_sub) {
        const uint32_t d = 1u + (uint32_t)((z >> 32) & 0xFFFFu) % 3u;
        *t++ = (uint8_t)acgt[(b + d) & 3u];
      } else if (u < t_ins) {
        *t++ = (uint8_t)acgt[(uint32_t)(z >> 48) & 3u];
        *t++ = (uint8_t)acgt[b];
      } else if (u < t_del) {
        /* deleted */
      } else {
        *t++ = (uint8_t)acgt[b];
      }
    }
  }
}

/* The batch as a caller of wavefront_align_packed2bits would hold it (wavefront_sequences.c:102-139): four bases per byte,
 * base j of a byte in bits 2j..2j+1, A 0 / C 1 / G 2 / T 3; sequence i = (len[i] + 3) / 4 bytes at out + out_off[i].
 * Letters outside ACGT are not representable: returns the number of such letters (their codes are written as 0). */
int64_t wfa_synth_pack2bits(const uint8_t* seqs, const int64_t* off, const int32_t* len, int64_t n, uint8_t* out, const int64_t* out_off) {
  int64_t i, bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (i = 0; i < n; ++i) {
    const uint8_t* s = seqs + off[i];
    uint8_t* o = out + out_off[i];
    const int32_t L = len[i];
    int32_t j;
    for (j = 0; j < L; j += 4) {
      uint8_t b = 0;
      int32_t q;
      for (q = 0; q < 4 && j + q < L; ++q) {
        const uint8_t c = s[j + q];
        uint8_t code = 0;
        if (c == 'C') code = 1; else if (c == 'G') code = 2; else if (c == 'T') code = 3; else if (c != 'A') ++bad;
        b |= (uint8_t)(code << (2 * q));
      }
      o[j >> 2] = b;
    }
  }
  return bad;
}
