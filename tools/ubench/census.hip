// Which workgroups of a 2-per-CU persistent grid share a CU?  Every block records its XCC_ID and HW_ID
// (s_getreg_b32), spins ~50 us so that the whole grid is resident at once, and the host prints, per
// (xcc, se, sh, cu), the block indices that landed there.   hipcc --offload-arch=gfx950 -O2 census.hip -o census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void census(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, all 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  lds[threadIdx.x] = (float)hw;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);
  if (lds[(threadIdx.x + 1) & 255] == -1.0f) out[0] = 0;
}

int main(int argc, char** argv) {
  int grid = argc > 1 ? atoi(argv[1]) : 512;
  size_t lds = argc > 2 ? atoi(argv[2]) : 72000;
  unsigned* d;
  hipMalloc(&d, grid * 8);
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(census, dim3(grid), dim3(256), lds, 0, d, 16);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * grid);
  hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < grid; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    unsigned key = (xcc << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 0x7) << 8);   // cu_id[11:8] sh_id[12] se_id[15:13]
    cu[key].push_back(b);
  }
  printf("grid %d, lds %zu: %zu distinct (xcc, se, sh, cu)\n", grid, lds, cu.size());
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ < 40) {
      printf("xcc %u hwid[15:8] %02x:", kv.first >> 16, kv.first & 0xff);
      for (int b : kv.second) printf(" %d", b);
      printf("\n");
    }
  }
  // histogram of the difference between co-resident block ids
  std::map<int, int> diff;
  for (auto& kv : cu)
    for (size_t i = 1; i < kv.second.size(); ++i) diff[kv.second[i] - kv.second[i - 1]]++;
  for (auto& kv : diff) printf("delta %d: %d\n", kv.first, kv.second);
  return 0;
}
