// eco_api.hip -- library-level entry points: error text, ABI version, device selection/query
// (Caffe::SetDevice / DeviceQuery, caffe_3d/src/caffe/common.cpp:140-190).
#include <string.h>

#include <mutex>

#include "eco_common.h"

namespace eco {

static thread_local char g_err[512] = {0};

char* error_buffer() { return g_err; }

void clear_error() { g_err[0] = 0; }

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int counter_slot_index(void* stream) {
  static std::mutex mu;
  static void* seen[64];
  static int nseen = 0;
  static unsigned cap_seq = 0;
  std::lock_guard<std::mutex> lock(mu);
#ifndef ECO_EMU
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive)
    return 64 + (int)(cap_seq++ % 192u);
#else
  (void)cap_seq;
#endif
  for (int i = 0; i < nseen; ++i)
    if (seen[i] == stream) return i;
  if (nseen == 64) return -1;
  seen[nseen] = stream;
  return nseen++;
}

}  // namespace eco

using namespace eco;

extern "C" int eco_abi_version(void) { return ECO_ABI_VERSION; }

extern "C" const char* eco_last_error(void) { return error_buffer(); }

#ifndef ECO_SRC_DIGEST
#define ECO_SRC_DIGEST "unknown"
#endif
extern "C" const char* eco_source_digest(void) { return ECO_SRC_DIGEST; }

#ifdef ECO_EMU

extern "C" int eco_is_device_build(void) { return 0; }
extern "C" int eco_device_count(int* count) {
  clear_error();
  ECO_REQUIRE(count != nullptr, "device_count: null argument");
  *count = 0;
  return ECO_OK;
}
extern "C" int eco_set_device(int device) {
  clear_error();
  return fail(ECO_ERR_RUNTIME, "set_device(%d): this is the CPU emulator build (tests only); no HIP device", device);
}
extern "C" int eco_device_info(int device, char* name, size_t name_len, int* num_cu, uint64_t* hbm_bytes) {
  clear_error();
  (void)name; (void)name_len; (void)num_cu; (void)hbm_bytes;
  return fail(ECO_ERR_RUNTIME, "device_info(%d): this is the CPU emulator build (tests only); no HIP device", device);
}
extern "C" int eco_device_pci_bus_id(int device, char* pci, size_t len) {
  clear_error();
  (void)pci; (void)len;
  return fail(ECO_ERR_RUNTIME, "device_pci_bus_id(%d): this is the CPU emulator build (tests only); no HIP device", device);
}

#else

extern "C" int eco_is_device_build(void) { return 1; }

extern "C" int eco_device_count(int* count) {
  clear_error();
  ECO_REQUIRE(count != nullptr, "device_count: null argument");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(ECO_ERR_RUNTIME, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  return ECO_OK;
}

extern "C" int eco_set_device(int device) {
  clear_error();
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(ECO_ERR_RUNTIME, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  return ECO_OK;
}

extern "C" int eco_device_info(int device, char* name, size_t name_len, int* num_cu, uint64_t* hbm_bytes) {
  clear_error();
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) return fail(ECO_ERR_RUNTIME, "hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
  if (name && name_len) {
    snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  }
  if (num_cu) *num_cu = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
  return ECO_OK;
}

extern "C" int eco_device_pci_bus_id(int device, char* pci, size_t len) {
  clear_error();
  ECO_REQUIRE(pci != nullptr && len >= 13, "device_pci_bus_id: buffer of at least 13 bytes needed");
  hipError_t e = hipDeviceGetPCIBusId(pci, (int)len, device);
  if (e != hipSuccess) return fail(ECO_ERR_RUNTIME, "hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e));
  for (char* c = pci; *c; ++c)
    if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');   // sysfs spells the address in lower case
  return ECO_OK;
}

#endif

// the two counter arrays (eco_blocked.hip, eco_stemb.hip)
namespace eco {
int spanp_counters_reset(void* stream);
int stemb_counters_reset(void* stream);
}
extern "C" int eco_counters_reset(void* stream) {
  clear_error();
  if (int rc = eco::spanp_counters_reset(stream)) return rc;
  return eco::stemb_counters_reset(stream);
}
