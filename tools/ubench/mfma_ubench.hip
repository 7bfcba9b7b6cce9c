// mfma_ubench.hip -- what does v_mfma_f32_32x32x2_f32 sustain on MI355X under the instruction mixes of the
// conv main loop?  (experiment tool, not part of the library)   hipcc --offload-arch=gfx950 -O3 mfma_ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(256, 3) void k(float* out, const float* in, int iters) {
  __shared__ float lds[2][16][128 + 128];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * 16 * 256; i += 256) ((float*)lds)[i] = in[i & 4095];
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768];
  float dummy = in[tid + 1024];
  const int half = lane >> 5, l31 = lane & 31;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (VARIANT & 1) {  // LDS fragment reads (no prefetch)
        a0 = lds[buf][2 * kk + half][l31]; a1 = lds[buf][2 * kk + half][32 + l31];
        b0 = lds[buf][2 * kk + half][128 + l31]; b1 = lds[buf][2 * kk + half][160 + l31];
      }
      if (VARIANT & 2) {  // a few VALU ops per MFMA group
        dummy = dummy * 1.0001f + 0.5f; dummy = dummy * 0.9999f - 0.5f; dummy += (float)kk; dummy *= 1.00001f;
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      if (VARIANT & 8) __builtin_amdgcn_sched_barrier(0);
    }
    if (VARIANT & 4) __syncthreads();
  }
  float s = dummy;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(int blocks_per_cu, int iters, float* out, const float* in) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, in, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * 32.0 * 4096.0;
  printf("variant %2d blocks/CU %d: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", V, blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
  std::vector<float> h(8192); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2 - 1;
  hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  for (int b = 1; b <= 4; ++b) {
    run<0>(b, iters, out, in); run<1>(b, iters, out, in); run<2>(b, iters, out, in); run<3>(b, iters, out, in);
    run<5>(b, iters, out, in); run<7>(b, iters, out, in); run<15>(b, iters, out, in);
  }
  return 0;
}
