// eco_blocked.h -- element helpers of the channel-blocked bf16 path, shared by eco_blocked.hip (convolutions) and
// eco_blocked_ops.hip (pooling, the pool + fc tail): the eight channels of one position are ONE 16-byte vector of bf16.
#pragma once
#include "eco_common.h"

namespace eco {

__device__ __forceinline__ uint4 load_block(const void* base, long block) { return ld((const uint4*)base + block); }

__device__ __forceinline__ void block_to_f32(const uint4& b, float (&f)[8]) {
  const unsigned w[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = bf16_bits_to_f32(w[e] & 0xffffu); f[2 * e + 1] = bf16_bits_to_f32(w[e] >> 16); }
}

// Four consecutive channels (elements 4*half .. 4*half+3 of a block) to / from memory.
__device__ __forceinline__ void load_quad(const void* base, long block, int half, float (&v)[4]) {
  const uint2 q = ld((const uint2*)base + 2 * block + half);
  v[0] = bf16_bits_to_f32(q.x & 0xffffu); v[1] = bf16_bits_to_f32(q.x >> 16);
  v[2] = bf16_bits_to_f32(q.y & 0xffffu); v[3] = bf16_bits_to_f32(q.y >> 16);
}
__device__ __forceinline__ void store_quad(void* base, long block, int half, const float (&v)[4]) {
  st((uint2*)base + 2 * block + half, make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
}

inline int grid_for_b(long count) {
  long g = ceil_div(count, 256);
  if (g < 1) g = 1;
  if (g > 1048576) g = 1048576;
  return (int)g;
}

}  // namespace eco
