// wfa_rtc_compat.hpp — lets the kernel headers compile both in the library build (hipcc) and at run time through hipRTC
// (csrc/wfa_rtc.cpp: kernels for penalty shapes the library has no instantiation of).  hipRTC defines __HIPCC_RTC__, brings its
// own reduced <hip/hip_runtime.h> and has no host standard library: the few constants the kernels use are defined here, and the
// host-side launch code of every header sits behind #ifndef __HIPCC_RTC__.
#pragma once
#ifdef __HIPCC_RTC__
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#ifndef INT_MAX
#define INT_MAX 2147483647
#endif
#ifndef INT_MIN
#define INT_MIN (-2147483647 - 1)
#endif
#ifndef UINT_MAX
#define UINT_MAX 4294967295u
#endif
#ifndef LLONG_MAX
#define LLONG_MAX 9223372036854775807ll
#endif
namespace wfa {
template <bool B, class T, class F> struct rtc_conditional { typedef T type; };
template <class T, class F> struct rtc_conditional<false, T, F> { typedef F type; };
}  // namespace wfa
#define WFA_CONDITIONAL wfa::rtc_conditional
#else
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#define WFA_CONDITIONAL std::conditional
#endif
