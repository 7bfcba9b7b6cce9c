// How fast does a NON-MFMA wave run next to a wave issuing back-to-back v_mfma_f32_32x32x2_f32 on the same SIMD?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) = MFMA chain, waves 4-7 = partner loop written in inline
// asm (so that the instruction mix is exactly what the name says).  Every wave reports its own elapsed s_memtime.
//   hipcc --offload-arch=gfx950 -O3 mfma_partner.hip -o mfma_partner
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { P_VALU = 1, P_DSREAD, P_DSWRITE, P_MIX, P_SALU, P_DSREAD_WAIT8 };

template <int ROLE_B>
__global__ __launch_bounds__(512) void k(unsigned long long* cyc, float* out, const float* in, int itersA, int itersB, int prioB) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 512) lds[i] = in[i & 4095];
  __syncthreads();
  float res = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
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
    float v0 = in[tid], v1 = in[tid + 1], v2 = in[tid + 2], v3 = in[tid + 3];
    const float m = in[tid + 100];
    unsigned addr = (unsigned)(tid & 63) * 4u;
    float d[8];
    int sacc = 0;
    for (int it = 0; it < itersB; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (ROLE_B == P_VALU)
          asm volatile("v_max3_f32 %0, %0, %4, %4\n v_max3_f32 %1, %1, %4, %4\n v_max3_f32 %2, %2, %4, %4\n v_max3_f32 %3, %3, %4, %4"
                       : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(m));
        if (ROLE_B == P_DSREAD || ROLE_B == P_DSREAD_WAIT8)
          asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768"
                       : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "v"(addr));
        if (ROLE_B == P_DSWRITE)
          asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768"
                       :: "v"(addr), "v"(m));
        if (ROLE_B == P_MIX)
          asm volatile("ds_read_b32 %0, %3\n v_max3_f32 %2, %2, %4, %4\n ds_read_b32 %1, %3 offset:256\n v_max3_f32 %2, %2, %4, %4"
                       : "=v"(d[0]), "=v"(d[1]), "+v"(v0) : "v"(addr), "v"(m));
        if (ROLE_B == P_SALU)
          asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(sacc));
        if (ROLE_B == P_DSREAD_WAIT8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    res = v0 + v1 + v2 + v3 + (float)sacc;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (res == 12345.678f) out[blockIdx.x * 512 + tid] = res;
}

template <int ROLE_B>
void pair(const char* name, unsigned long long* cyc, float* out, const float* in, int itA, int itB) {
  unsigned long long h[8];
  double r[4][2];
  const int cfg[4][3] = {{itA, 0, 0}, {0, itB, 0}, {itA, itB, 0}, {itA, itB, 1}};
  for (int c = 0; c < 4; ++c) {
    hipLaunchKernelGGL((k<ROLE_B>), dim3(256), dim3(512), 0, 0, cyc, out, in, cfg[c][0], cfg[c][1], cfg[c][2]);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc + 8 * 17, sizeof(h), hipMemcpyDeviceToHost);   // block 17
    r[c][0] = (h[0] + h[1] + h[2] + h[3]) / 4.0;
    r[c][1] = (h[4] + h[5] + h[6] + h[7]) / 4.0;
  }
  const double per = 32.0 * itB;   // partner instructions
  printf("%-34s MFMA wave: alone %.0f, paired %.0f (prio3: %.0f) cycles | partner: alone %.0f (%.1f cyc/instr), paired %.0f (%.1f cyc/instr), at prio 3 %.0f (%.1f)\n",
         name, r[0][0], r[2][0], r[3][0], r[1][1], r[1][1] / per, r[2][1], r[2][1] / per, r[3][1], r[3][1] / per);
}

int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
  hipMemcpy(in, h, 8192 * 4, hipMemcpyHostToDevice);
  const int itA = 4000;      // 16 MFMAs = 1024 cycles per iteration -> 4.1 M cycles
  pair<P_VALU>("partner: 32 v_max3 per iter", cyc, out, in, itA, 4000);
  pair<P_DSREAD>("partner: 32 ds_read_b32 per iter", cyc, out, in, itA, 4000);
  pair<P_DSREAD_WAIT8>("partner: ds_read_b32, wait per 4", cyc, out, in, itA, 4000);
  pair<P_DSWRITE>("partner: 32 ds_write_b32 per iter", cyc, out, in, itA, 4000);
  pair<P_MIX>("partner: 16 ds_read + 16 v_max3", cyc, out, in, itA, 4000);
  pair<P_SALU>("partner: 32 s_add per iter", cyc, out, in, itA, 4000);
  return 0;
}
