// host_cigar.cpp — host-only text helpers of the C ABI (include/wfa_hip.h): what pywfa prints through WFA2-lib's
// cigar_print_pretty (align.pyx:445-459 -> alignment/cigar.c:778-863), as a string a binding can write wherever it likes.
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <string.h>
#include "wfa_hip.h"

namespace {
// "{n}{op}" runs of ops[0..len) (cigar.c:705-739); with_matches = false leaves the M runs out
void append_runs(std::string& out, const uint8_t* ops, int64_t len, bool with_matches, bool fold_mismatches) {
  int64_t i = 0;
  char num[24];
  while (i < len) {
    const uint8_t op = (fold_mismatches && ops[i] == 'X') ? (uint8_t)'M' : ops[i];
    int64_t j = i + 1;
    while (j < len && ((fold_mismatches && ops[j] == 'X') ? (uint8_t)'M' : ops[j]) == op) ++j;
    if (with_matches || op != 'M') { snprintf(num, sizeof(num), "%lld", (long long)(j - i)); out += num; out += (char)op; }
    i = j;
  }
}
}  // namespace

extern "C" int64_t wfa_hip_cigar_sprint_pretty(const uint8_t* ops, int64_t ops_len, const uint8_t* pattern, int32_t plen,
                                               const uint8_t* text, int32_t tlen, char* out, int64_t cap) {
  if (ops_len < 0 || plen < 0 || tlen < 0 || (ops_len > 0 && !ops) || (plen > 0 && !pattern) || (tlen > 0 && !text) || (cap > 0 && !out)) return WFA_HIP_EINVAL;
  std::string rp, rg, rt;
  rp.reserve((size_t)ops_len + 16); rg.reserve((size_t)ops_len + 16); rt.reserve((size_t)ops_len + 16);
  int32_t pp = 0, tp = 0;
  for (int64_t i = 0; i < ops_len; ++i) {
    const uint8_t op = ops[i];
    const bool hp = pp < plen, ht = tp < tlen;
    if (op == 'M' || op == 'X') {
      if (!hp || !ht) break;   // (an op string that outruns its sequences: nothing more to draw)
      const bool same = pattern[pp] == text[tp];
      // a match is drawn '|', a mismatch ' '; an op that contradicts the sequences is marked 'X' (cigar.c:799-821)
      rg += (op == 'M') ? (same ? '|' : 'X') : (same ? 'X' : ' ');
      rp += (char)pattern[pp++]; rt += (char)text[tp++];
    } else if (op == 'I') {
      if (!ht) break;
      rp += '-'; rg += ' '; rt += (char)text[tp++];
    } else if (op == 'D') {
      if (!hp) break;
      rp += (char)pattern[pp++]; rg += ' '; rt += '-';
    }
  }
  // whatever the op string leaves unaligned follows, marked '?' (cigar.c:836-847)
  const int32_t rest_p = plen - pp, rest_t = tlen - tp;
  rp.append(reinterpret_cast<const char*>(pattern) + pp, (size_t)rest_p);
  rt.append(reinterpret_cast<const char*>(text) + tp, (size_t)rest_t);
  rg.append((size_t)(rest_p > rest_t ? rest_p : rest_t), '?');
  std::string s = "      ALIGNMENT ";
  append_runs(s, ops, ops_len, true, false);
  s += "\n      ETRACE    ";
  append_runs(s, ops, ops_len, false, false);
  s += "\n      CIGAR     ";
  append_runs(s, ops, ops_len, true, true);   // SAM style without '=' / 'X' (cigar_print_SAM_CIGAR(.., false))
  s += "\n      PATTERN    " + rp + "\n                 " + rg + "\n      TEXT       " + rt + "\n";
  if (cap > 0) {
    const size_t ncopy = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
    memcpy(out, s.data(), ncopy);
    out[ncopy] = '\0';
  }
  return (int64_t)s.size();
}
