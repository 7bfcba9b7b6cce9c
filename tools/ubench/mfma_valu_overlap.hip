// Do v_mfma_f32_32x32x2_f32 (wave A) and plain fp32 VALU work (wave B) of the SAME SIMD overlap on gfx950?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a pure MFMA chain, waves 4-7 a VALU-only /
// LDS-only / bf16-MFMA loop.  Times: each role alone, then both.  sum => serialized, max => overlapped.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// roleB: 0 none, 1 VALU fma chain x8 independent, 2 ds_read_b32 stream, 3 bf16 MFMA, 4 f32 MFMA (two MFMA waves per SIMD)
template <int ROLE_B>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int itersA, int itersB, int prioB) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, wave = tid >> 6;
  lds[tid] = in[tid]; lds[tid + 512] = in[tid + 512];
  __syncthreads();
  float res = 0.f;
  if (wave < 4) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float a = in[tid], b = in[tid + 64];
    for (int it = 0; it < itersA; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) res += acc[j][r];
  } else {
    if (prioB) __builtin_amdgcn_s_setprio(3);
    if (ROLE_B == 1) {
      float v[8];
      for (int j = 0; j < 8; ++j) v[j] = in[tid + j];
      const float m = in[tid + 100], c = in[tid + 101];
      for (int it = 0; it < itersB; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], m, c);
      }
      for (int j = 0; j < 8; ++j) res += v[j];
    } else if (ROLE_B == 2) {
      int idx = tid & 1023;
      float s = 0.f;
      for (int it = 0; it < itersB; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) s += lds[(idx + 64 * u) & 4095];
        idx = (idx + 1) & 1023;
      }
      res = s;
    } else if (ROLE_B == 3) {
      f32x16 acc[4];
      for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[tid + e]; b[e] = (__bf16)in[tid + 8 + e]; }
      for (int it = 0; it < itersB; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
      for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) res += acc[j][r];
    } else if (ROLE_B == 4) {
      f32x16 acc[4];
      for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const float a = in[tid], b = in[tid + 64];
      for (int it = 0; it < itersB; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      }
      for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) res += acc[j][r];
    }
  }
  out[blockIdx.x * 512 + tid] = res;
}

template <int ROLE_B>
float run(float* out, const float* in, int itersA, int itersB, int prioB) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ROLE_B>), dim3(256), dim3(512), 0, 0, out, in, 10, 10, prioB);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ROLE_B>), dim3(256), dim3(512), 0, 0, out, in, itersA, itersB, prioB);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int ROLE_B>
void trio(const char* name, float* out, const float* in, int itA, int itB) {
  const float a = run<ROLE_B>(out, in, itA, 0, 0), b = run<ROLE_B>(out, in, 0, itB, 0), ab = run<ROLE_B>(out, in, itA, itB, 0),
              abp = run<ROLE_B>(out, in, itA, itB, 1);
  printf("%-28s f32-MFMA alone %.3f ms | partner alone %.3f ms | both %.3f ms | both, partner at prio 3 %.3f ms  (sum %.3f, max %.3f)\n",
         name, a, b, ab, abp, a + b, a > b ? a : b);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 512 * 4);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
  hipMemcpy(in, h, 8192 * 4, hipMemcpyHostToDevice);
  const int itA = 20000;   // 16 MFMAs x 64 cycles = 1024 cycles per iteration
  trio<1>("partner: 32 v_fma / iter", out, in, itA, 20000 * 4);   // 32 VALU x ~4 cycles = 128+ cycles per iteration
  trio<1>("partner: v_fma, half load", out, in, itA, 20000 * 2);
  trio<2>("partner: 32 ds_read_b32/iter", out, in, itA, 20000);
  trio<3>("partner: 32 bf16 MFMA/iter", out, in, itA, 20000);
  trio<4>("partner: 16 f32 MFMA/iter", out, in, itA, 20000);
  return 0;
}
