// eco_common.h -- host-side helpers for the C-ABI entry points.
#pragma once
#include <stdarg.h>
#include <stdio.h>

#include "../../include/eco_hip.h"
#include "eco_device.h"

namespace eco {

// Thread-local error text; replaces the reference's glog CHECK / LOG(FATAL) output.
char* error_buffer();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// After a kernel launch: pick up launch-configuration errors without synchronising.
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ECO_ERR_RUNTIME, "%s: HIP launch failed: %s", what, hipGetErrorString(e));
  return ECO_OK;
}

inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

// Element offset of (image, channel 0, spatial index sp) in a strided output view (include/eco_hip.h).
__device__ __forceinline__ long view_base(const eco_view& v, int img, int sp) {
  const int b = img / v.t, t = img - b * v.t;
  return (long)b * v.stride_b + (long)t * v.stride_t + sp;
}

}  // namespace eco

#define ECO_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return eco::fail(ECO_ERR_INVALID, __VA_ARGS__); \
  } while (0)
