// Do scalar memory atomics work on gfx950, and what does one cost?  (tools/ubench: hipcc --offload-arch=gfx950 -O2)
// Every wave draws `n` tickets from one counter with s_atomic_add (returning form); the host checks that the tickets are
// a permutation of 0 .. waves*n-1 and prints the cycles per draw of wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned* ctr, unsigned* out, unsigned long long* cyc, int n) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    unsigned t = 1u;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(ctr) : "memory");
    if ((threadIdx.x & 63) == 0) out[wave * n + i] = t;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = (t1 - t0) / n;
}
int main() {
  const int blocks = 512, threads = 256, n = 64, waves = blocks * threads / 64;
  unsigned *ctr, *out; unsigned long long* cyc;
  hipMalloc(&ctr, 4); hipMalloc(&out, waves * n * 4); hipMalloc(&cyc, 8);
  hipMemset(ctr, 0, 4);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, ctr, out, cyc, n);
  std::vector<unsigned> h(waves * n); unsigned long long c; unsigned fin;
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&fin, ctr, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = fin == (unsigned)(waves * n);
  for (size_t i = 0; i < h.size(); ++i) ok = ok && h[i] == i;
  printf("s_atomic_add: final %u (expected %d), tickets %s, %llu cycles per draw (2048 waves drawing at once)\n", fin, waves * n, ok ? "a permutation" : "NOT a permutation", c);
  hipMemset(ctr, 0, 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ctr, out, cyc, n);
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("one wave alone: %llu cycles per draw\n", c);
  return ok ? 0 : 1;
}
