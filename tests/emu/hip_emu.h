// hip_emu.h -- CPU fiber emulator for the HIP kernels in csrc/ (TEST INFRASTRUCTURE ONLY).
//
// The authoring container has no GPU and GPU time is scarce, so the CPU test-suite
// compiles the *same* kernel sources (csrc/*.hip) as plain C++ against this header
// (`-DECO_EMU -x c++ -include hip_emu.h`) into tests/emu/libeco_emu.so.  One OS thread
// runs one workgroup at a time; every GPU thread of the workgroup is a ucontext fiber.
// `__syncthreads()` and the wave-level operations (MFMA, shuffles, readfirstlane)
// are rendezvous points between fibers, so tiling / index / barrier logic is
// exercised exactly as written; only timing and memory-model effects are not.
//
//   * MFMA 32x32x2 f32 follows the gfx950 lane layout documented in
//     /opt/skills/guides/cdna_hip_programming.md section 3:
//       A: lane l holds A[i = l&31][k = l>>5];  B: lane l holds B[k = l>>5][j = l&31];
//       C/D: lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31],
//     accumulated as a k-ordered fmaf chain (bitwise what the hardware does).
//   * MFMA 32x32x16 bf16 (the blocked bf16 path): same C/D map, eight consecutive k per lane (see below).
//   * global loads/stores made through eco::ld / eco::st are bounds-checked against
//     buffers registered with emu_register_buffer() -- an out-of-range access that
//     would fault (or silently corrupt) on the GPU aborts the test with a message.
//
// The product package never loads libeco_emu.so; it is not a fallback.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 {
  float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct float2 {
  float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct alignas(16) uint4 {
  unsigned x, y, z, w;
};
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct uint2 {
  unsigned x, y;
};
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));

namespace emu {
constexpr int kWave = 64;

struct ThreadCtx {
  dim3 tidx, bidx, bdim, gdim;
  int tid;   // linear thread id in the block
  int lane;  // tid % 64
  int wave;  // tid / 64
};
extern thread_local ThreadCtx* tls_cur;  // the fiber currently running on this OS thread

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body);
void syncthreads();
float wave_xchg_f32(float v, int src_lane);          // value of `v` held by src_lane
emu_f32x16 mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c);
// v_mfma_f32_32x32x16_bf16: lane l holds A[i = l&31][k = 8*(l>>5) + e] and B[k = 8*(l>>5) + e][j = l&31] as eight
// bf16 (e = 0..7, two per dword, low half first); C/D as for the f32 form.  Products are exact in fp32 and are
// accumulated here in k order with one rounding each (the hardware's internal order is unspecified: compare
// with a tolerance).
emu_f32x16 mfma_f32_32x32x16_bf16(uint4 a, uint4 b, emu_f32x16 c);
int readfirstlane(int v);
void* dyn_smem();
bool check_access(const void* p, size_t bytes, bool write);  // false = out of bounds (counted)
}  // namespace emu

#define threadIdx (emu::tls_cur->tidx)
#define blockIdx (emu::tls_cur->bidx)
#define blockDim (emu::tls_cur->bdim)
#define gridDim (emu::tls_cur->gdim)

static inline void __syncthreads() { emu::syncthreads(); }

// hipLaunchKernelGGL((kernel<...>), grid, block, dyn_smem, stream, args...)
#define ECO_EMU_STRIP_PARENS(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...)                     \
  do {                                                                                 \
    (void)(stream);                                                                    \
    emu::launch((grid), (block), (smem), [=]() { ECO_EMU_STRIP_PARENS kernel(__VA_ARGS__); }); \
  } while (0)

extern "C" {
// Registers a host buffer as "device memory" for bounds checking (tests call this via ctypes).
void emu_register_buffer(const void* p, size_t bytes);
void emu_clear_buffers(void);
// 0 = off (default when nothing is registered), 1 = abort on unregistered access.
void emu_set_strict(int on);
int emu_violation_count(void);
}

// HIP device code uses the global-namespace integer min/max overloads.
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
