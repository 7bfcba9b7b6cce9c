// buflds_check.hip -- what `buffer_load_dwordx4 ... offen lds` does with out-of-range lanes on gfx950 (the zero padding
// of the span kernel rides on it): in-range lanes must copy, lanes whose voffset is past num_records must write ZEROS to
// their LDS slot (not leave it), soffset must be added, and we record whether soffset takes part in the range check.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/buflds_check.hip -o tools/ubench/buflds_check && tools/ubench/buflds_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void glds16_buf(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_off), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
// mode 0: every third lane out of range by voffset; mode 1: voffset in range, soffset pushes the address past num_records
__global__ void k(const uint4* x, unsigned bytes, uint4* y, unsigned soff, int mode) {
  __shared__ uint4 L[256];
  L[threadIdx.x] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
  __syncthreads();
  i32x4 r = make_rsrc(x, bytes);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned voff = (unsigned)threadIdx.x * 16u;
  if (mode == 0 && threadIdx.x % 3 == 0) voff = 0xfffffff0u;
  glds16_buf(r, voff, soff, (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(L + wave * 64));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  y[threadIdx.x] = L[threadIdx.x];
}
int main() {
  const int n = 1024;   // 16 KB buffer of uint4
  std::vector<uint4> h(n);
  for (int i = 0; i < n; ++i) h[i] = make_uint4(i, i + 1000, i + 2000, i + 3000);
  uint4 *dx, *dy;
  hipMalloc(&dx, (n + 512) * 16);   // slack behind the buffer so that an unchecked soffset read stays mapped
  hipMalloc(&dy, 256 * 16);
  hipMemset(dx, 0x5a, (n + 512) * 16);
  hipMemcpy(dx, h.data(), n * 16, hipMemcpyHostToDevice);
  std::vector<uint4> out(256);
  int bad = 0;
  // mode 0, soffset 0 and soffset 4096
  for (unsigned soff : {0u, 4096u}) {
    k<<<1, 256>>>(dx, n * 16, dy, soff, 0);
    hipMemcpy(out.data(), dy, 256 * 16, hipMemcpyDeviceToHost);
    int zeros = 0, kept = 0, copied = 0, wrong = 0;
    for (int t = 0; t < 256; ++t) {
      const uint4 v = out[t];
      if (t % 3 == 0) {
        if (v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0) ++zeros;
        else if (v.x == 0xdeadbeefu) ++kept;
        else ++wrong;
      } else {
        const unsigned e = t + soff / 16;
        if (v.x == e && v.y == e + 1000 && v.z == e + 2000 && v.w == e + 3000) ++copied; else ++wrong;
      }
    }
    printf("mode 0 soffset %u: out-of-range lanes -> zeros %d, LDS left untouched %d; in-range copied %d; wrong %d\n", soff, zeros, kept, copied, wrong);
    bad += wrong + kept;
  }
  // mode 1: voffset < num_records but voffset + soffset >= num_records for the upper lanes (soffset = 14 KB: lanes >= 128)
  k<<<1, 256>>>(dx, n * 16, dy, 14336u, 1);
  hipMemcpy(out.data(), dy, 256 * 16, hipMemcpyDeviceToHost);
  int zeros = 0, read_through = 0, other = 0;
  for (int t = 128; t < 256; ++t) {
    const uint4 v = out[t];
    if (v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0) ++zeros;
    else if (v.x == 0x5a5a5a5au) ++read_through;
    else ++other;
  }
  printf("mode 1 (voffset in range, voffset + soffset past num_records): zeros %d, read through %d, other %d  -> soffset %s part of the range check\n",
         zeros, read_through, other, zeros == 128 ? "IS" : "is NOT");
  printf(bad ? "BUFLDS_FAIL\n" : "BUFLDS_OK\n");
  return bad ? 1 : 0;
}
