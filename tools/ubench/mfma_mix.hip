// What does one extra instruction of each class cost the f32 (and, MF16 = 2, the bf16 32x32x16) MFMA stream of the SAME wave on gfx950?
// One wave per SIMD (256-thread workgroup per CU, launch_bounds(256,1)) or two (512 threads); per iteration 16
// v_mfma_f32_32x32x2_f32 (= 1024 cycles of matrix pipe) or 32 v_mfma_f32_16x16x4_f32, plus N fillers issued from
// inline asm between them.  Reports cycles per iteration at the measured time (assuming 2.3 GHz) and the cost per filler.
//   hipcc --offload-arch=gfx950 -O3 mfma_mix.hip -o mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { F_NONE, F_VFMA, F_VMOV, F_DS32, F_DS64, F_DS128, F_SALU, F_DSW32, F_VFMA_DEP, F_PKFMA, F_PKADD, F_VADD, F_GLD16, F_2VADD, F_4VADD };
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FILL, int NF, int MF16>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = in[i & 4095];
  __syncthreads();
  f32x16 acc[4];
  f32x4 acc4[8];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc4[j][r] = 0.f;
  const float a = in[tid], b = in[tid + 64];
  bf16x8 ab, bb;
  for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)in[tid + e]; bb[e] = (__bf16)in[tid + 8 + e]; }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = in[tid + j];
  const float m = in[tid + 100], c = in[tid + 101];
  unsigned addr = (unsigned)(tid & 63) * 16u;
  float d0, d1; float2 d2; float4 d4;
  int sacc = iters;
  f32x2 pv[8], pm = {m, c}, pc = {c, m};
  for (int j = 0; j < 8; ++j) { pv[j][0] = in[tid + j]; pv[j][1] = in[tid + j + 9]; }
  f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
  const float* gp = in + (tid & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MF16 == 2) {
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[u & 3], 0, 0, 0);
      } else if (MF16) {
        acc4[(2 * u) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[(2 * u) & 7], 0, 0, 0);
        acc4[(2 * u + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[(2 * u + 1) & 7], 0, 0, 0);
      } else {
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
      }
      if (u < NF) {
        if (FILL == F_VFMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(m), "v"(c));
        if (FILL == F_VFMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(m), "v"(c));
        if (FILL == F_VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[u & 7]) : "v"(m));
        if (FILL == F_DS32) asm volatile("ds_read_b32 %0, %1" : "=v"(d0) : "v"(addr));
        if (FILL == F_DS64) asm volatile("ds_read_b64 %0, %1" : "=v"(d2) : "v"(addr));
        if (FILL == F_DS128) asm volatile("ds_read_b128 %0, %1" : "=v"(d4) : "v"(addr));
        if (FILL == F_DSW32) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(m));
        if (FILL == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[u & 7]) : "v"(pm), "v"(pc));
        if (FILL == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[u & 7]) : "v"(pm));
        if (FILL == F_VADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[u & 7]) : "v"(m));
        if (FILL == F_2VADD) asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(v[u & 7]), "+v"(v[(u + 1) & 7]) : "v"(m));
        if (FILL == F_4VADD) asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(v[u & 7]), "+v"(v[(u + 1) & 7]), "+v"(v[(u + 2) & 7]), "+v"(v[(u + 3) & 7]) : "v"(m));
        if (FILL == F_GLD16) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g4) : "v"(gp));
        if (FILL == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (FILL == F_DS32 || FILL == F_DS64 || FILL == F_DS128 || FILL == F_DSW32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (FILL == F_GLD16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float res = (float)sacc;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) res += acc[j][r];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) res += acc4[j][r];
  for (int j = 0; j < 8; ++j) res += v[j] + pv[j][0] + pv[j][1];
  res += g4[0];
  if (res == 12345.678f) out[blockIdx.x * 512 + tid] = res + d0 + d2.x + d4.x;
}

static float g_base[3][2];
template <int FILL, int NF, int MF16>
void run(const char* name, float* out, const float* in, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((k<FILL, NF, MF16>), dim3(256), dim3(threads), 0, 0, out, in, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<FILL, NF, MF16>), dim3(256), dim3(threads), 0, 0, out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int w = threads / 256 - 1;
  if (FILL == F_NONE) g_base[MF16][w] = ms;
  const double cyc = ms * 1e-3 * 2.3e9 / iters;   // cycles per iteration per SIMD (both waves of a SIMD together)
  const double per = NF ? (ms - g_base[MF16][w]) * 1e-3 * 2.3e9 / iters / (NF * (threads / 256)) : 0.0;
  printf("%-12s %s waves/SIMD %d  fillers/iter %2d: %.3f ms  %7.1f cyc/iter  -> %5.1f cycles per filler\n",
         MF16 == 2 ? "bf16 32x32x16" : MF16 ? "16x16x4" : "32x32x2", name, threads / 256, NF, ms, cyc, per);
}

template <int MF16>
void all(float* out, const float* in, int threads) {
  run<F_NONE, 0, MF16>("none      ", out, in, threads);
  run<F_VFMA, 16, MF16>("v_fma     ", out, in, threads);
  run<F_VFMA, 8, MF16>("v_fma     ", out, in, threads);
  run<F_VFMA_DEP, 16, MF16>("v_fma dep ", out, in, threads);
  run<F_VMOV, 16, MF16>("v_mov     ", out, in, threads);
  run<F_DS32, 16, MF16>("ds_read32 ", out, in, threads);
  run<F_DS64, 16, MF16>("ds_read64 ", out, in, threads);
  run<F_DS128, 16, MF16>("ds_read128", out, in, threads);
  run<F_DS128, 8, MF16>("ds_read128", out, in, threads);
  run<F_DSW32, 16, MF16>("ds_write32", out, in, threads);
  run<F_SALU, 16, MF16>("s_add     ", out, in, threads);
  run<F_VADD, 16, MF16>("v_add     ", out, in, threads);
  run<F_2VADD, 16, MF16>("2x v_add  ", out, in, threads);
  run<F_4VADD, 16, MF16>("4x v_add  ", out, in, threads);
  run<F_4VADD, 4, MF16>("4x v_add  ", out, in, threads);
  run<F_PKFMA, 16, MF16>("v_pk_fma  ", out, in, threads);
  run<F_PKADD, 16, MF16>("v_pk_add  ", out, in, threads);
  run<F_GLD16, 16, MF16>("gload x4  ", out, in, threads);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 512 * 4);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
  hipMemcpy(in, h, 8192 * 4, hipMemcpyHostToDevice);
  for (int threads = 256; threads <= 512; threads += 256) { all<0>(out, in, threads); all<1>(out, in, threads); all<2>(out, in, threads); }
  return 0;
}
