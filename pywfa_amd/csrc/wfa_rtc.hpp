// wfa_rtc.hpp — run-time instantiation of the register kernels for penalty shapes the library was not built with (csrc/wfa_rtc.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace wfa {

// false under WFA_HIP_NO_RTC=1 (no compile: what an aligner records when it is created)
bool rtc_enabled();
// true when hipRTC can compile here (a probe kernel, compiled once per process the first time a shape without an instantiation asks)
bool rtc_available();
// failures of the run-time path so far in this process (compile, module load, launch): wfa_hip_batch_run compares the count before
// and after a run to tell a failed run-time shape from any other device error, and re-plans without them
unsigned rtc_failure_count();
// the kernel `name_expr` (a C++ name expression, e.g. "wfa::wfa_lane_kernel<5, 8, 2, false, false>") of the kernel header
// `header` (e.g. "wfa_lane.hpp") for the current device; compiled on first use, nullptr on failure (rtc_last_error())
hipFunction_t rtc_kernel(const char* header, const std::string& name_expr);
// launch it: `args` is the kernel's argument list laid out as the kernel sees it (each argument at its natural alignment)
int rtc_launch(const char* header, const std::string& name_expr, unsigned grid, unsigned block, size_t smem, hipStream_t stream,
               const void* args, size_t args_bytes);
const char* rtc_last_error();
// WFA_HIP_RTC_ALL=1 (tests): also the shapes the library has instantiations of take the run-time path
bool rtc_force_all();

// Index seg_shape() reports for a penalty shape without an instantiation that the run-time path takes.
#define WFA_SHAPE_RTC 100
// Shapes the run-time path accepts (penalties / gcd): the rings of the register kernels are registers, one per step of history —
// beyond these depths a kernel would spill or run at one wave per SIMD, and the general / tiled kernels are the better tools.
inline bool rtc_shape_ok(int X, int OE, int E, int OE2 = 0, int E2 = 0) {
  if (X < 1 || OE < 1 || E < 1 || E > OE) return false;
  const int dm = X > OE ? X : OE;
  if (dm > 24 || E > 8) return false;   // (match = -1 rescales pywfa's default 4/6/2 to 10/12/5: lags 10, 17, 5)
  if (OE2 > 0 && (E2 < 1 || E2 > 8 || OE2 <= dm || OE2 > 64)) return false;
  return true;
}
// the lane-per-pair kernel unrolls 16 diagonals x the M ring over packed registers: 8 registers per step of history
inline bool rtc_lane_shape_ok(int X, int OE, int E) { return (X > OE ? X : OE) <= 14 && E <= 4; }
inline std::string rtc_bool(bool b) { return b ? "true" : "false"; }

}  // namespace wfa
