// What does a MAX pool folded into a producer's epilogue through atomics cost on gfx950?  (tools/ubench)
// The pattern wfused_kernel would have for conv2_3x3 + pool2 (56 x 56 -> 28 x 28, 3x3 stride 2, values >= 0 after ReLU): a
// thread owns one 4 x 4 output tile of one channel and contributes to 3 x 3 pooled cells -- one of them alone (plain store),
// eight shared with neighbouring tiles (device-scope atomic max on the bit pattern).  Compared with: the sixteen-float tile
// stored as four float4 (what the kernel does today), and the nine cells stored plainly (a lower bound).  Correctness of the
// atomic form is checked against a host pooling of the same synthetic values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ __forceinline__ float val(int f, int c, int h, int w) {
  unsigned x = (unsigned)(((f * 192 + c) * 56 + h) * 56 + w) * 2654435761u;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
template <int MODE>   // 0: full tile as 4 x float4; 1: 9 cells, atomics where shared; 2: 9 plain stores (wrong, lower bound)
__global__ __launch_bounds__(256) void k(float* full, float* pooled, int frames) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)frames * 192 * 196;
  if (tid >= total) return;
  // as in wfused: 32 consecutive lanes = 32 consecutive tile columns r of one channel, 8 channels per workgroup
  const int lane32 = (int)(tid & 31);
  const long g = tid >> 5;
  const int cl = (int)(g % 8);
  const long g2 = g / 8;
  const long nblk = ((long)frames * 196 + 31) / 32;
  const long nb = g2 % nblk;
  const int cb = (int)(g2 / nblk);
  const long r = nb * 32 + lane32;
  if (r >= (long)frames * 196 || cb >= 24) return;
  const int c = cb * 8 + cl;
  const int f = (int)(r / 196), t = (int)(r % 196), th = t / 14, tw = t % 14;
  float y[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) y[i][j] = val(f, c, 4 * th + i, 4 * tw + j);
  if (MODE == 0) {
    float* o = full + (((long)f * 192 + c) * 56 + 4 * th) * 56 + 4 * tw;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(float4*)(o + i * 56) = make_float4(y[i][0], y[i][1], y[i][2], y[i][3]);
    return;
  }
  // row groups: {0} -> pooled row 2th-1, {0,1,2} -> 2th, {2,3} -> 2th+1; columns alike
  float cm[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cm[i][0] = y[i][0];
    cm[i][1] = fmaxf(fmaxf(y[i][0], y[i][1]), y[i][2]);
    cm[i][2] = fmaxf(y[i][2], y[i][3]);
  }
  float p[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    p[0][j] = cm[0][j];
    p[1][j] = fmaxf(fmaxf(cm[0][j], cm[1][j]), cm[2][j]);
    p[2][j] = fmaxf(cm[2][j], cm[3][j]);
  }
  float* o = pooled + ((long)f * 192 + c) * 784;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int ph = 2 * th - 1 + a, pw = 2 * tw - 1 + b;
      if (ph < 0 || pw < 0 || ph >= 28 || pw >= 28) continue;
      if (MODE == 2 || (a == 1 && b == 1)) o[ph * 28 + pw] = p[a][b];
      else __hip_atomic_fetch_max((unsigned*)(o + ph * 28 + pw), __float_as_uint(p[a][b]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
static float hval(int f, int c, int h, int w) {
  unsigned x = (unsigned)(((f * 192 + c) * 56 + h) * 56 + w) * 2654435761u;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
int main() {
  const int frames = 512;
  const long total = (long)frames * 192 * 196;
  float *full, *pooled;
  hipMalloc(&full, (size_t)frames * 192 * 3136 * 4);
  hipMalloc(&pooled, (size_t)frames * 192 * 784 * 4);
  const long nblk = ((long)frames * 196 + 31) / 32;
  const unsigned grid = (unsigned)((nblk * 24 * 8 * 32 + 255) / 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (mode == 1) hipMemsetAsync(pooled, 0, (size_t)frames * 192 * 784 * 4, 0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, full, pooled, frames);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, full, pooled, frames);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, full, pooled, frames);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    const char* what[3] = {"full tile, 4 x float4 stores (1.23 GB)", "9 cells: 1 store + 8 device-scope atomic max (+ memset of 0.31 GB)", "9 plain stores (lower bound)"};
    printf("mode %d  %-72s %.4f ms\n", mode, what[mode], best);
    if (mode == 1) {
      std::vector<float> h((size_t)4 * 192 * 784);
      hipMemcpy(h.data(), pooled + (size_t)(frames - 4) * 192 * 784, h.size() * 4, hipMemcpyDeviceToHost);
      long bad = 0;
      for (int f = 0; f < 4; ++f) for (int c = 0; c < 192; ++c) for (int ph = 0; ph < 28; ++ph) for (int pw = 0; pw < 28; ++pw) {
        float m = 0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { int hh = 2 * ph + i, ww = 2 * pw + j; if (hh < 56 && ww < 56) m = std::max(m, hval(frames - 4 + f, c, hh, ww)); }
        if (h[((size_t)f * 192 + c) * 784 + ph * 28 + pw] != m) ++bad;
      }
      printf("        atomic pooling vs host pooling on the last 4 frames: %ld wrong cells\n", bad);
    }
  }
  return 0;
}
