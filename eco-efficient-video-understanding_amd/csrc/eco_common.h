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

// Work-counter slot (0 .. 255) of a dynamic-share launch on `stream`, or -1 (the launch then takes static shares):
// eager launches get ONE slot per stream -- launches of a stream run in order, and a launch's last workgroup leaves the
// slot zero for the next -- for up to 64 distinct streams; launches recorded by a stream capture take slots 64 .. 255 in
// turn (a captured graph keeps its slots and may be replayed on any stream, next to eager launches).  include/eco_hip.h,
// "work counters", states the bound this leaves.
int counter_slot_index(void* stream);

// Compute units of the calling thread's current device (plans with num_cu = 0 are sized for it).
inline int current_device_num_cu() {
#ifdef ECO_EMU
  return 256;
#else
  int dev = 0, cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) return 256;
  return cu;
#endif
}

// Element offset of (image, channel 0, spatial index sp) in a strided output view (include/eco_hip.h).
__device__ __forceinline__ long view_base(const eco_view& v, int img, int sp) {
  const int b = v.t == 1 ? img : img / v.t, t = img - b * v.t;   // (t == 1, the plain tensor: no ~40-instruction divide)
  return (long)b * v.stride_b + (long)t * v.stride_t + sp;
}

}  // namespace eco

// Launches that need more than the default 64 KB of dynamic LDS opt in with hipFuncSetAttribute -- which applies to the
// CURRENT DEVICE only.  The macro remembers, per call site (= per kernel instantiation), thread and device ordinal,
// that the limit was raised: a thread that moves to another device (caffe.set_device, Net(device=...)) raises it there
// too (round-2 advisor finding: a per-thread flag left the second device at the default and its launches failed).
// (the limit leaves 8 KB of the CU's 160 KB to a kernel's static __shared__ arrays -- the blocked kernels' epilogue
// parameters --: the runtime rejects a dynamic limit that, with the static part, exceeds the hardware's; no launch of
// this library asks for more than 135 KB)
constexpr int kEcoMaxDynamicLds = 152 * 1024;
#ifdef ECO_EMU
#define ECO_RAISE_DYNAMIC_LDS(kernel, who) do { } while (0)
#else
#define ECO_RAISE_DYNAMIC_LDS(kernel, who)                                                                           \
  do {                                                                                                               \
    static thread_local unsigned long long eco_raised_ = 0;   /* one bit per device ordinal < 64 */                   \
    int eco_dev_ = -1;                                                                                               \
    if (hipGetDevice(&eco_dev_) != hipSuccess) eco_dev_ = -1;                                                        \
    if (eco_dev_ < 0 || eco_dev_ >= 64 || !((eco_raised_ >> eco_dev_) & 1ull)) {                                     \
      hipError_t eco_e_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                              kEcoMaxDynamicLds);                                                    \
      if (eco_e_ != hipSuccess)                                                                                      \
        return eco::fail(ECO_ERR_RUNTIME, "%s: cannot raise the dynamic LDS limit: %s", who, hipGetErrorString(eco_e_)); \
      if (eco_dev_ >= 0 && eco_dev_ < 64) eco_raised_ |= 1ull << eco_dev_;                                           \
    }                                                                                                                \
  } while (0)
#endif

#define ECO_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return eco::fail(ECO_ERR_INVALID, __VA_ARGS__); \
  } while (0)
