// unaligned_probe.hip — does a raw buffer dword load work at a 2-byte-aligned offset on gfx950?
// (NOT part of the product; decides how uint16 tap pairs can be fetched)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint16_t* src, unsigned* out, int n) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 2, 0x00020000);
  int i = threadIdx.x;
  out[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, i * 2, 0, 0);   // byte offset 2*i: odd i = unaligned
}
int main() {
  const int n = 256;
  uint16_t h[n]; for (int i = 0; i < n; i++) h[i] = (uint16_t)(1000 + i);
  uint16_t* d; unsigned* o; hipMalloc(&d, n * 2); hipMalloc(&o, 64 * 4);
  hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  unsigned r[64]; hipError_t e = hipMemcpy(r, o, 64 * 4, hipMemcpyDeviceToHost);
  printf("status %s\n", hipGetErrorString(e));
  int ok = 1;
  for (int i = 0; i < 64; i++) { unsigned want = (unsigned)h[i] | ((unsigned)h[i + 1] << 16); if (r[i] != want) { ok = 0; printf("lane %d got %08x want %08x\n", i, r[i], want); if (i > 6) break; } }
  printf("unaligned dword buffer loads %s\n", ok ? "WORK" : "DO NOT WORK");
  return 0;
}
