// What buffer_load_dwordx4 (raw buffer, stride 0, offen) returns for a lane whose 16 bytes are partly outside
// num_records: per-dword range check, or the whole load?  (tools/ubench: hipcc --offload-arch=gfx950 -O2 -o /tmp/b bufld_check.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* x, unsigned* y, int bytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(x + 4), 0, bytes, 0x00020000);
  const int off = (int)threadIdx.x * 4 - 16;     // lane 0: 16 bytes before the base ... one dword per lane step
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  y[threadIdx.x * 4 + 0] = v.x; y[threadIdx.x * 4 + 1] = v.y; y[threadIdx.x * 4 + 2] = v.z; y[threadIdx.x * 4 + 3] = v.w;
}
int main() {
  unsigned h[64], *dx, *dy, o[64 * 4];
  for (int i = 0; i < 64; ++i) h[i] = 100 + i;
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dy, sizeof(o));
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, 32);   // records = dwords 104..111
  hipMemcpy(o, dy, sizeof(o), hipMemcpyDeviceToHost);
  for (int l = 0; l < 16; ++l) printf("lane %2d (byte offset %3d): %u %u %u %u\n", l, l * 4 - 16, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  return 0;
}
