// eco_device.h -- device-side helpers shared by the ECO kernels (gfx950 / CDNA4).
//
// Written for gfx950 only: wave = 64 lanes, fp32 MFMA 32x32x2.  The ECO_EMU branch
// is not a second GPU backend; it binds the same kernel source to the CPU fiber
// emulator in tests/emu/ so the CPU test-suite can run the kernels' index math.
#pragma once

#ifdef ECO_EMU
#include <sched.h>

#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

#include <stdint.h>
#include <string.h>

#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace eco {

constexpr int kWave = 64;

// D(32x32) += A(32x2) * B(2x32), fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32).
// Lane l supplies a = A[l&31][l>>5], b = B[l>>5][l&31]; reg r of the result is
// D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
#ifdef ECO_EMU
  return emu::mfma_f32_32x32x2f32(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// D(32x32) += A(32x16) * B(16x32), bf16 in / fp32 accumulate (v_mfma_f32_32x32x16_bf16, 16x the fp32 rate).
// Lane l supplies eight consecutive k of row / column l&31: A[l&31][8*(l>>5) + e], B[8*(l>>5) + e][l&31],
// e = 0..7, packed two per dword (low half = even e) in a 16-byte vector; C/D as above.
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(uint4 a, uint4 b, f32x16 c) {
#ifdef ECO_EMU
  return emu::mfma_f32_32x32x16_bf16(a, b, c);
#else
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// fp32 <-> bf16 (round to nearest even; the GPU form is v_cvt_pk_bf16_f32, the expression below gives the same
// bits for every finite input and for infinities).
__device__ __forceinline__ unsigned f32_to_bf16_bits(float f) {
#ifdef ECO_EMU
  unsigned u;
  memcpy(&u, &f, 4);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
#else
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
#endif
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned h) {
  const unsigned u = h << 16;
#ifdef ECO_EMU
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __builtin_bit_cast(float, u);
#endif
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
#ifdef ECO_EMU
  return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
#else
  // one v_cvt_pk_bf16_f32 (the scalar casts above compile to two conversions, a shift and an or)
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
#endif
}

// LDS-DMA: every lane copies 16 bytes from its own global address straight into LDS at
// `lds_wave_base + 16*lane` (global_load_lds_dwordx4: no VGPR staging, no ds_write; the LDS base is wave-uniform,
// in M0).  Asynchronous on the GPU: the bytes are in LDS once the issuing wave's vmcnt has counted the piece down
// (wait_dma_all_but) AND a workgroup barrier has been passed by the reader.  Issued from inline asm on purpose:
// hipcc orders ds_reads behind LDS-DMA it can see by alias analysis and, in a multi-buffer pipeline, ends up
// draining the whole queue (s_waitcnt vmcnt(0)) in front of fragment reads; the kernels that use this do their
// own counting, and exactly one instruction is issued per call, whatever the lanes' addresses.
__device__ __forceinline__ void glds16(const uint4* gptr, uint4* lds_wave_base) {
#ifdef ECO_EMU
  uint4 q = {0u, 0u, 0u, 0u};
  if (emu::check_access(gptr, 16, false)) memcpy(&q, (const void*)gptr, 16);   // the source may be only 4-byte aligned
  lds_wave_base[emu::tls_cur->lane] = q;
#else
#ifdef ECO_GLDS_READFIRSTLANE
  // Under SGPR pressure (eco_blocked.hip's kernels with their large argument structs) the register allocator has handed
  // the "s" operand a VGPR -- spilled SGPRs live in VGPR lanes -- which the assembler rejects; a readfirstlane pins it.
  // Opt-in per file: where the offset is not already in an SGPR it costs a VALU instruction per DMA piece.
  const unsigned lds_off = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
#else
  const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
#endif
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(gptr) : "memory", "m0");
#endif
}

// LDS-DMA through a buffer descriptor: `buffer_load_dwordx4 voff, srd, soff offen lds`.  Two things the flat form
// (glds16) cannot do: (1) the address is (SGPR descriptor base) + (SGPR soffset) + (VGPR 32-bit byte offset) -- the
// uniform part of an address costs SALU instructions, the per-lane part is one register that stays put for a whole tile,
// where glds16 needs a 64-bit VALU add per piece; (2) a lane whose voffset + soffset is not below the descriptor's
// num_records reads nothing and WRITES ZEROS to its 16 bytes of LDS (measured on gfx950, tools/ubench/buflds_check.hip:
// soffset takes part in the range check): zero padding is a per-lane select of kBufOob, no zero page, no second address.
// num_records is 32-bit and kBufOob + soffset must not wrap: tensors above 2 GB stay on the glds16 kernels.
constexpr unsigned kBufOob = 0x80000000u;
// 16-byte loads through a raw buffer descriptor (wave-uniform base / size, per-lane byte offset).  Range check as the
// hardware's (tools/ubench/bufld_check.hip, MI355X): per dword against num_records at the END of the range (a load that
// runs over it returns its in-range dwords and zeros), on the 32-bit offset without wrap-around (a "negative" offset is
// out of range as a whole, also the dwords that would land at 0 and above).  A compiler builtin: its vmcnt is tracked.
#ifdef ECO_EMU
struct BufRd { const char* base; unsigned bytes; };
__device__ __forceinline__ BufRd make_buf_rd(const void* p, unsigned bytes) { return BufRd{(const char*)p, bytes}; }
__device__ __forceinline__ uint4 gld16_buf(const BufRd& r, unsigned voff) {
  unsigned q[4] = {0u, 0u, 0u, 0u};
  for (int i = 0; i < 4; ++i)
    if ((unsigned long long)voff + 4 * i + 4 <= r.bytes && emu::check_access(r.base + voff + 4 * i, 4, false)) memcpy(&q[i], r.base + voff + 4 * i, 4);
  return make_uint4(q[0], q[1], q[2], q[3]);
}
#else
typedef __amdgpu_buffer_rsrc_t BufRd;
__device__ __forceinline__ BufRd make_buf_rd(const void* p, unsigned bytes) {   // (p, bytes: wave-uniform)
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 gld16_buf(const BufRd& r, unsigned voff) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
#endif
#ifdef ECO_EMU
struct BufRsrc { const char* base; unsigned bytes; };
__device__ __forceinline__ BufRsrc make_buf_rsrc(const void* p, unsigned bytes) { return BufRsrc{(const char*)p, bytes}; }
__device__ __forceinline__ void glds16_buf(const BufRsrc& r, unsigned voff, unsigned soff, uint4* lds_wave_base) {
  uint4 q = {0u, 0u, 0u, 0u};
  const unsigned long long off = (unsigned long long)voff + soff;
  if (off + 16 <= r.bytes && emu::check_access(r.base + off, 16, false)) memcpy(&q, r.base + off, 16);
  lds_wave_base[emu::tls_cur->lane] = q;
}
#else
typedef int BufRsrc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ BufRsrc make_buf_rsrc(const void* p, unsigned bytes) {   // (p, bytes: wave-uniform)
  const unsigned long long a = (unsigned long long)p;
  BufRsrc r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));   // stride 0: raw buffer
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
// (lds_wave_base: wave-uniform; one instruction per call whatever the lanes' offsets.  Counted by the caller:
// wait_dma_all_but.)
__device__ __forceinline__ void glds16_buf(const BufRsrc& r, unsigned voff, unsigned soff, uint4* lds_wave_base) {
  // (soff / lds_wave_base must be SGPR values to the compiler -- no readfirstlane here: a VALU-written SGPR read by a
  // VMEM instruction five states later is a hazard nobody pads inside an asm statement)
  const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_off), "v"(voff), "s"(r), "s"(soff) : "memory", "m0");
}
#endif

// 16-byte store through a buffer descriptor (`buffer_store_dwordx4 v, voff, srd, soff offen`): lanes whose voffset +
// soffset is not below num_records store nothing, and the instruction is issued whatever the lanes' predicates -- so a
// kernel that counts its own memory operations (s_waitcnt vmcnt(N) over LDS-DMA pieces) knows how many stores an epilogue
// put in the queue.  Not waited for by anybody: the data registers must not be reused before the store has read them,
// hence the two wait states inside the statement.
__device__ __forceinline__ void gst16_buf(const BufRsrc& r, unsigned voff, unsigned soff, uint4 v) {
#ifdef ECO_EMU
  const unsigned long long off = (unsigned long long)voff + soff;
  if (off + 16 <= r.bytes && emu::check_access(r.base + off, 16, true)) memcpy((void*)(r.base + off), &v, 16);
#else
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t q = {v.x, v.y, v.z, v.w};
  // s_nop 4 in front: the descriptor / soffset may have been written by a VALU instruction (v_readfirstlane) just ahead
  // of this statement -- VALU-writes-SGPR -> VMEM-reads-it needs five wait states and the compiler pads nothing inside asm
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(q), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
}

// v_permlane32_swap: the upper half-wave's `a` and the lower half-wave's `b` trade places (lanes 32..63 of a <-> lanes
// 0..31 of b).  Two results of a 32x32 MFMA tile that sit in lanes l and l + 32 end up side by side in one lane.
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {
#ifdef ECO_EMU
  const int l = emu::tls_cur->lane;
  float fa, fb;
  memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
  const float xa = emu::wave_xchg_f32(fa, l ^ 32), xb = emu::wave_xchg_f32(fb, l ^ 32);
  float na = l < 32 ? fa : xb, nb = l < 32 ? xa : fb;
  memcpy(&a, &na, 4); memcpy(&b, &nb, 4);
#else
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
#endif
}

// Division of n < 2^31 by a run-time constant d >= 1 without the ~40-instruction software divide: q = (t + ((n - t) >> s1))
// >> s2 with t = umulhi(n, m) (Granlund & Montgomery's round-up method; m, s1, s2 from fastdiv_make on the host).
struct FastDiv { unsigned m, s1, s2, d; };
inline FastDiv fastdiv_make(unsigned d) {
  FastDiv f;
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                       // l = ceil(log2 d)
  f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l < 1 ? 0 : l - 1;
  return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) {
#ifdef ECO_EMU
  const unsigned t = (unsigned)(((unsigned long long)n * f.m) >> 32);
#else
  const unsigned t = __umulhi(n, f.m);
#endif
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

// Pipelined LDS-DMA needs two things __syncthreads() cannot give: a wait for all but the newest N DMA pieces of
// this wave (s_waitcnt vmcnt(N); __syncthreads() drains everything) and a barrier that does not drain.  Both are
// no-ops / a plain barrier under the emulator, where the DMA is synchronous.
template <int N>
__device__ __forceinline__ void wait_dma_all_but() {
#ifndef ECO_EMU
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wg_barrier_nodrain() {
#ifdef ECO_EMU
  emu::syncthreads();
#else
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}

// Compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>).  For bodies whose
// tables / register-array indices must be constants: a `#pragma unroll` loop leaves them as run-time arithmetic until
// after the unroller has priced the body, and a body it then refuses to unroll indexes its register arrays through
// scratch memory (wfused_kernel<48, *> ran 4x slower that way for one build of round 3).
template <int... I, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// Issue priority of this wave among the waves of its SIMD (s_setprio 0..3; 0 is the launch default).
template <int P>
__device__ __forceinline__ void set_wave_priority() {
#ifndef ECO_EMU
  __builtin_amdgcn_s_setprio(P);
#endif
}

// Make a wave-uniform value provably uniform (SGPR) for the compiler.
__device__ __forceinline__ int uniform(int v) {
#ifdef ECO_EMU
  return emu::readfirstlane(v);
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}

// Scheduling fence: keeps the compiler from moving instructions across this point (used to pin
// the load / MFMA interleave of the conv main loop).  No code is emitted.
__device__ __forceinline__ void sched_fence() {
#ifndef ECO_EMU
  __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ int lane_id() {
#ifdef ECO_EMU
  return emu::tls_cur->lane;
#else
  return (int)(threadIdx.x & 63u);
#endif
}

__device__ __forceinline__ float shfl_xor(float v, int mask) {
#ifdef ECO_EMU
  return emu::wave_xchg_f32(v, emu::tls_cur->lane ^ mask);
#else
  return __shfl_xor(v, mask, 64);
#endif
}

// Butterfly reductions over the 64 lanes of a wave; every lane gets the result.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
  return v;
}

// Global-memory accessors.  On the GPU these are plain loads/stores; under the
// emulator they are bounds-checked against the registered "device" buffers.
template <typename T>
__device__ __forceinline__ T ld(const T* p) {
#ifdef ECO_EMU
  if (!emu::check_access(p, sizeof(T), false)) return T{};
#endif
  return *p;
}
template <typename T>
__device__ __forceinline__ void st(T* p, T v) {
#ifdef ECO_EMU
  if (!emu::check_access(p, sizeof(T), true)) return;
#endif
  *p = v;
}

// Load from (wave-uniform base) + (per-lane 32-bit byte offset): the `saddr` form of global_load -- SGPR base, VGPR
// offset, immediate -- so that a stream of loads off one uniform base costs no per-lane 64-bit address arithmetic.
// (Beside f32 MFMAs every VALU instruction is ~6 cycles of matrix-pipe time: profiles/r03_notes.md.)
template <typename T>
__device__ __forceinline__ T ld_su(const void* uniform_base, unsigned lane_byte_offset) {
  return ld((const T*)((const char*)uniform_base + lane_byte_offset));
}

// ---- hand-offs between workgroups of ONE launch (stream-K partial sums) -------------------------------------------
// Per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by other CUs' stores
// (MI355X_MICROARCH.md, "inter-workgroup visibility").  The payload therefore leaves as 16-byte WRITE-THROUGH stores
// (sc0 sc1: reaches memory, leaves no dirty line), the producing waves wait for theirs (vmcnt(0)), a workgroup barrier,
// then one lane publishes a flag with a device-scope store; the consumer polls the flag with device-scope loads and
// reads the payload with device-scope (sc1: L1-bypassing) loads.  No agent-scope release / acquire fences: a release
// writes back every dirty line of the XCD's L2 and an acquire invalidates L1 and the L2's non-coherent lines -- one per
// workgroup cost more than the whole kernel (profiles/r03_notes.md).
// (OFF: compile-time byte offset < 4096, the instruction's immediate: four stores off one address register pair)
template <int OFF>
__device__ __forceinline__ void st_writethrough16(float* p, float4 v) {
#ifdef ECO_EMU
  st((float4*)((char*)p + OFF), v);
#else
  static_assert(OFF >= 0 && OFF < 4096, "");
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t q = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc0 sc1" ::"v"(p), "v"(q), "n"(OFF) : "memory");
#endif
}
__device__ __forceinline__ void wait_own_stores() {
#ifndef ECO_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ float ld_device_scope(const float* p) {
#ifdef ECO_EMU
  return ld(p);
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// Sixteen 16-byte device-scope loads issued together and waited for once: four 32x32 tiles of a partial block (tile t
// at p + t*1024 floats, its four register groups 1 KB apart by immediate).  The wait's asm statement carries every
// destination as an in/out operand, so no use can move above it.  (With 4-byte device-scope loads, 16 in flight, the
// 128 KB of one workgroup's partial took ~16 us -- on the critical path of the workgroup that finishes the tile.)
__device__ __forceinline__ void ld_partial_tiles4(const float* p, float4 (&q)[4][4]) {
#ifdef ECO_EMU
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) q[t][g] = ld((const float4*)(p + t * 1024 + g * 256));
#else
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  f32x4_t r[16];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float* pt = p + t * 1024;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[4 * t]) : "v"(pt) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024 sc1" : "=v"(r[4 * t + 1]) : "v"(pt) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:2048 sc1" : "=v"(r[4 * t + 2]) : "v"(pt) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:3072 sc1" : "=v"(r[4 * t + 3]) : "v"(pt) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                 "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
               :
               : "memory");
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) q[t][g] = make_float4(r[4 * t + g][0], r[4 * t + g][1], r[4 * t + g][2], r[4 * t + g][3]);
#endif
}

// Device-scope fetch-and-add / store on a work counter (one lane calls it).
__device__ __forceinline__ unsigned counter_fetch_add(unsigned* p, unsigned v) {
#ifdef ECO_EMU
  return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);   // (workgroups run on several host threads)
#else
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// The same for a whole wave at once, on the scalar unit (s_atomic_add, returning form; p wave-uniform): ~600 cycles round
// trip for a wave on its own (tools/ubench/satomic_check.hip), nothing in the vector memory counter.  The wait is part of
// the statement: the compiler must not copy the result register before the value is in it.
__device__ __forceinline__ unsigned counter_draw_wave(unsigned* p) {
#ifdef ECO_EMU
  return __atomic_fetch_add(p, 1u, __ATOMIC_SEQ_CST);   // (called by one fiber of the workgroup)
#else
  unsigned t = 1u;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(p) : "memory");
  return t;
#endif
}
__device__ __forceinline__ void counter_store(unsigned* p, unsigned v) {
#ifdef ECO_EMU
  __atomic_store_n(p, v, __ATOMIC_SEQ_CST);
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

__device__ __forceinline__ void flag_publish(int* flag, int value) {
#ifdef ECO_EMU
  __atomic_store_n(flag, value, __ATOMIC_SEQ_CST);
#else
  __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// Spin (one lane, with s_sleep between polls) until *flag == value.
__device__ __forceinline__ void flag_wait(const int* flag, int value) {
#ifdef ECO_EMU
  while (__atomic_load_n(flag, __ATOMIC_SEQ_CST) != value) sched_yield();   // (the producer block runs on another host thread)
#else
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value) __builtin_amdgcn_s_sleep(8);
#endif
}

// Dynamic LDS of the launch as a float array (16-byte aligned; no static __shared__ may precede it).
#ifdef ECO_EMU
#define ECO_DYNAMIC_LDS(name) float* name = (float*)emu::dyn_smem()
#else
#define ECO_DYNAMIC_LDS(name)                                        \
  extern __shared__ __attribute__((aligned(16))) char name##_raw[]; \
  float* name = (float*)name##_raw
#endif

// Hide a per-lane integer from the optimiser.  Used on LDS fragment indices: two 8-byte reads off the same
// base register get merged into ds_read2_b64, which the LDS serves at half the rate of two ds_read_b64.
#ifdef ECO_EMU
#define ECO_OPAQUE(v) ((void)(v))
#define ECO_OPAQUE64(v) ((void)(v))
#else
#define ECO_OPAQUE(v) asm volatile("" : "+v"(v))
#define ECO_OPAQUE64(v) asm volatile("" : "+v"(v))
#endif

// Probe builds (-DECO_CLOCK_PROBE, tools/exp/clock_probe.sh): the shader clock a kernel actually ran at, from inside it --
// workgroup 0's first thread reads the shader-cycle counter (s_memtime) and the constant 100 MHz real-time counter
// (s_memrealtime) when it starts and when it ends and prints both differences.
#if defined(ECO_CLOCK_PROBE) && !defined(ECO_EMU)
struct ClockProbe {
  unsigned long long t0, r0;
  const char* name;
  bool on;
  // (ECO_CLOCK_PROBE = 2: also the first thread of the middle and the last workgroup, with absolute real-time stamps --
  // do the workgroups of a launch start and end together?)
  __device__ __forceinline__ ClockProbe(const char* n)
      : name(n), on(threadIdx.x == 0 && (blockIdx.x == 0 || (ECO_CLOCK_PROBE > 1 && (blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)))) {
    if (on) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  }
  __device__ __forceinline__ ~ClockProbe() {
    if (on) {
      const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
      const unsigned long long dt = __builtin_readcyclecounter() - t0, dr = r1 - r0;
      if (ECO_CLOCK_PROBE > 1) printf("CLKB %s %u %u %llu %llu %llu\n", name, (unsigned)blockIdx.x, (unsigned)gridDim.x, r0, r1, dt);
      else printf("CLK %s %llu %llu\n", name, dt, dr);
    }
  }
};
#define ECO_CLOCK(name) ClockProbe eco_clock_probe_(name)
#else
#define ECO_CLOCK(name) do { } while (0)
#endif

// XCD-aware workgroup remap (MI355X: 8 XCDs, hardware places block b on XCD b % 8, each
// XCD has a private 4 MiB L2).  Returns the logical tile id for hardware block `b` such
// that each XCD works on a contiguous range of logical tiles; bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  constexpr int kXcd = 8;
  const int q = nwg / kXcd, r = nwg % kXcd;
  const int xcd = b % kXcd, idx = b / kXcd;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace eco
