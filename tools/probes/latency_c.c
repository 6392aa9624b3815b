/* Per-call latency of wfa_hip_align_batch for one 150 bp pair, from C (no Python in the loop).
 *   gcc -O2 -I include tools/probes/latency_c.c -o /tmp/latency_c -L pywfa_amd -lwfa_hip -Wl,-rpath,$PWD/pywfa_amd && /tmp/latency_c */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "wfa_hip.h"
static double now_us(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
int main(int argc, char** argv) {
  const int L = 150, calls = argc > 1 ? atoi(argv[1]) : 3000;
  char p[256], t[256];
  srand(1);
  for (int i = 0; i < L; ++i) p[i] = "ACGT"[rand() & 3];
  memcpy(t, p, L);
  for (int i = 0; i < 3; ++i) t[rand() % L] = "ACGT"[rand() & 3];
  uint8_t blob[512];
  memcpy(blob, p, L); memcpy(blob + L, t, L);
  for (int full = 0; full < 2; ++full) {
    wfa_hip_config_t c;
    wfa_hip_config_default(&c);
    c.scope = full; c.span = full ? 1 : 0;
    if (getenv("MISMATCH")) c.mismatch = atoi(getenv("MISMATCH"));
    wfa_hip_aligner_t* al = wfa_hip_create(&c, 0);
    if (!al) { printf("create failed: %s\n", wfa_hip_global_error()); return 1; }
    int64_t p_off = 0, t_off = L, c_off[2] = {0, 2 * L}, c_begin = 0;
    int32_t p_len = L, t_len = L, score = 0, status = 0, c_len = 0;
    uint8_t ops[512];
    for (int i = 0; i < 20; ++i) wfa_hip_align_batch(al, 1, blob, &p_off, &p_len, &t_off, &t_len, &score, &status, full ? ops : NULL, full ? c_off : NULL, full ? &c_begin : NULL, full ? &c_len : NULL);
    double t0 = now_us(), worst = 0;
    for (int i = 0; i < calls; ++i) {
      const double a = now_us();
      wfa_hip_align_batch(al, 1, blob, &p_off, &p_len, &t_off, &t_len, &score, &status, full ? ops : NULL, full ? c_off : NULL, full ? &c_begin : NULL, full ? &c_len : NULL);
      const double d = now_us() - a;
      if (d > worst) worst = d;
    }
    printf("%s: %.2f us per call (worst %.1f), score %d status %d cigar_len %d\n", full ? "full CIGAR" : "score only", (now_us() - t0) / calls, worst, score, status, c_len);
    wfa_hip_destroy(al);
  }
  return 0;
}
