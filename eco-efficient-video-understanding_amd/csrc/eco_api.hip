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

// (round-5 advisor finding) Three ways the slot of a launch could be shared by launches that run concurrently, all closed:
//   * hipStreamPerThread is ONE handle value for a different real stream in every host thread: static shares (-1).
//     (The null stream and hipStreamLegacy name the same, process-wide, in-order stream: one slot.)
//   * the capture ring wrapped: two live graphs could hold the same slot.  Capture slots are now handed out once; when the
//     192 are gone, captured launches take static shares until eco_counters_release_capture_slots() says that every graph
//     captured so far has been destroyed.
//   * the 64-entry stream table never forgot a destroyed stream: a process that creates and destroys streams ended on
//     static shares for good.  A full table now recycles the entry of a stream that has no work in flight (destroyed
//     handles and idle streams alike: hipStreamQuery says anything but hipErrorNotReady).
static std::mutex g_slot_mu;
static void* g_seen[64];
static int g_nseen = 0;
static unsigned g_cap_next = 0;

int counter_slot_index(void* stream) {
  std::lock_guard<std::mutex> lock(g_slot_mu);
#ifndef ECO_EMU
  if ((hipStream_t)stream == hipStreamPerThread) return -1;
  if ((hipStream_t)stream == hipStreamLegacy) stream = nullptr;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive)
    return g_cap_next < 192u ? 64 + (int)(g_cap_next++) : -1;
#endif
  for (int i = 0; i < g_nseen; ++i)
    if (g_seen[i] == stream) return i;
  if (g_nseen < 64) {
    g_seen[g_nseen] = stream;
    return g_nseen++;
  }
#ifndef ECO_EMU
  for (int i = 0; i < 64; ++i) {
    if (g_seen[i] == nullptr) continue;                       // (the null stream is never evicted)
    const hipError_t q = hipStreamQuery((hipStream_t)g_seen[i]);
    (void)hipGetLastError();                                  // an invalid (destroyed) handle must not look like a launch failure
    if (q != hipErrorNotReady) {
      g_seen[i] = stream;
      return i;
    }
  }
#endif
  return -1;
}

void counter_release_capture_slots() {
  std::lock_guard<std::mutex> lock(g_slot_mu);
  g_cap_next = 0;
}

// test hook (tests/test_advice_r5.py through eco_counter_slot_probe): the slot a launch on `stream` would get
int counter_slot_probe(void* stream) { return counter_slot_index(stream); }

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

namespace eco {
void counter_release_capture_slots();
int counter_slot_probe(void* stream);
}
extern "C" int eco_counters_release_capture_slots(void) {
  clear_error();
  eco::counter_release_capture_slots();
  return ECO_OK;
}
extern "C" int eco_counter_slot_probe(void* stream) { return eco::counter_slot_probe(stream); }
