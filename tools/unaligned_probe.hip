// unaligned_probe.hip — do raw buffer loads work below their natural alignment on gfx950?
//   dword at 2-byte-aligned offsets (uint16 tap pairs), ushort at odd offsets (uint8 tap pairs)
// (NOT part of the product; decides how narrow-integer tap rows can be fetched)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k32(const uint16_t* src, unsigned* out, int n) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 2, 0x00020000);
  int i = threadIdx.x;
  out[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, i * 2, 0, 0);   // odd i: 2-byte aligned only
}
__global__ void k16(const uint8_t* src, unsigned* out, int n) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n, 0x00020000);
  int i = threadIdx.x;
  out[i] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, i, 0, 0);  // odd i: unaligned
}
int main() {
  const int n = 256;
  uint16_t h[n]; for (int i = 0; i < n; i++) h[i] = (uint16_t)(1000 + i);
  uint8_t b[n]; for (int i = 0; i < n; i++) b[i] = (uint8_t)(3 * i + 7);
  uint16_t* d; uint8_t* d8; unsigned* o; hipMalloc(&d, n * 2); hipMalloc(&d8, n); hipMalloc(&o, 64 * 4);
  hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice); hipMemcpy(d8, b, n, hipMemcpyHostToDevice);
  unsigned r[64];
  hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, d, o, n);
  hipError_t e = hipMemcpy(r, o, 64 * 4, hipMemcpyDeviceToHost);
  int ok = e == hipSuccess;
  for (int i = 0; i < 64; i++) ok &= r[i] == ((unsigned)h[i] | ((unsigned)h[i + 1] << 16));
  printf("dword loads at 2-byte offsets %s\n", ok ? "WORK" : "DO NOT WORK");
  hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, d8, o, n);
  e = hipMemcpy(r, o, 64 * 4, hipMemcpyDeviceToHost);
  ok = e == hipSuccess;
  for (int i = 0; i < 64; i++) ok &= r[i] == ((unsigned)b[i] | ((unsigned)b[i + 1] << 8));
  printf("ushort loads at odd offsets    %s\n", ok ? "WORK" : "DO NOT WORK");
  return 0;
}
