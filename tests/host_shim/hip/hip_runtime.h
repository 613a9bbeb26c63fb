// Test shim: lets g++ compile the DEVICE-side model math of csrc/models.h for the host (tests/test_device_math_on_host.py).
// The arithmetic templates there (dynamics on double / Dual / MDual, RK steps, error-state maps) contain nothing GPU-specific
// except two hardware estimate instructions, which are replaced by exact host equivalents: the Newton steps that follow them
// in rcp_fast / rsqrt_fast then change nothing.  Test infrastructure only.
#pragma once
#include <cmath>
#include <cstddef>
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__
#define TO_CONST_AS   // descriptor tables: plain pointers on the host (problem_dev.h)
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
using std::fabs;
using std::fma;
using std::fmax;
using std::fmin;
using std::rint;
using std::sqrt;
using std::log10;
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
#include <cstring>
inline long long __double_as_longlong(double x) { long long b; std::memcpy(&b, &x, sizeof(b)); return b; }
