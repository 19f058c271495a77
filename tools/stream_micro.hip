// stream_micro.hip — HBM streaming-shape probe for gfx950 (NOT part of the product).
// Which launch shape lets a 3-reads + 1-write (16 B/px) or 1-read + 1-write (8 B/px) stream over
// 16 4K float frames reach the practical HBM rate?  Compares a linear float4 sweep with the
// wave-marching strip shape used by wave_stencil.hpp (256 px x strip_h rows per wave).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_micro.hip -o /tmp/stream_micro && /tmp/stream_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int W = 3840, H = 2160, F = 16;
constexpr long NPX = (long)W * H * F;

__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned n) {
  unsigned per = n / 8;
  if (per * 8 != n) return b;
  return (b % 8) * per + b / 8;
}

// NR read streams (1 or 3), one write stream; linear: thread -> one float4
template <int NR>
__global__ void __launch_bounds__(256) linear_kernel(const float4* a, const float4* b, const float4* c, float4* d, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = a[i];
  if (NR == 3) { float4 u = b[i], w = c[i]; v.x += u.x * w.x; v.y += u.y * w.y; v.z += u.z * w.z; v.w += u.w * w.w; }
  d[i] = v;
}

// persistent grid-stride version, U float4 per thread per iteration
template <int NR, int U>
__global__ void __launch_bounds__(256) gridstride_kernel(const float4* a, const float4* b, const float4* c, float4* d, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
    float4 v[U], u[U], w[U];
#pragma unroll
    for (int k = 0; k < U; k++) { long j = i + k * stride; if (j < n4) { v[k] = a[j]; if (NR == 3) { u[k] = b[j]; w[k] = c[j]; } } }
#pragma unroll
    for (int k = 0; k < U; k++) { long j = i + k * stride; if (j < n4) { if (NR == 3) { v[k].x += u[k].x * w[k].x; v[k].y += u[k].y * w[k].y; v[k].z += u[k].z * w[k].z; v[k].w += u[k].w * w[k].w; } d[j] = v[k]; } }
  }
}

// wave-marching strips: wave = 256 px x SH rows, D rows in flight
template <int NR, int D, bool SWZ>
__global__ void __launch_bounds__(256) strip_kernel(const float* a, const float* b, const float* c, float* d, int sh, int strips_x, unsigned strips) {
  const int lane = threadIdx.x & 63;
  unsigned blk = SWZ ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  unsigned sid = blk * 4 + (threadIdx.x >> 6);
  if (sid >= strips) return;
  int syi = sid / strips_x, sxi = sid - syi * strips_x;
  long base = (long)blockIdx.y * W * H + (long)syi * sh * W + sxi * 256 + lane * 4;
  int rows = H - syi * sh < sh ? H - syi * sh : sh;
#pragma unroll 1
  for (int r = 0; r < rows; r += D) {
    float4 v[D], u[D], w[D];
#pragma unroll
    for (int k = 0; k < D; k++) if (r + k < rows) {
      long o = base + (long)(r + k) * W;
      v[k] = *(const float4*)(a + o);
      if (NR == 3) { u[k] = *(const float4*)(b + o); w[k] = *(const float4*)(c + o); }
    }
#pragma unroll
    for (int k = 0; k < D; k++) if (r + k < rows) {
      long o = base + (long)(r + k) * W;
      if (NR == 3) { v[k].x += u[k].x * w[k].x; v[k].y += u[k].y * w[k].y; v[k].z += u[k].z * w[k].z; v[k].w += u[k].w * w[k].w; }
      *(float4*)(d + o) = v[k];
    }
  }
}

// as strip_kernel, but strips step 248 px (read 256 px from x = 248*sxi - 4, lanes 1..62 store):
// the geometry of wave_stencil.hpp with one halo lane per side
template <int NR, int D>
__global__ void __launch_bounds__(256) ostrip_kernel(const float* a, const float* b, const float* c, float* d, int sh, int strips_x, unsigned strips) {
  const int lane = threadIdx.x & 63;
  unsigned blk = xcd_swizzle(blockIdx.x, gridDim.x);
  unsigned sid = blk * 4 + (threadIdx.x >> 6);
  if (sid >= strips) return;
  int syi = sid / strips_x, sxi = sid - syi * strips_x;
  int x = sxi * 248 - 4 + lane * 4;
  bool in = x >= 0 && x + 4 <= W;
  bool writer = in && lane >= 1 && lane < 63;
  long base = (long)blockIdx.y * W * H + (long)syi * sh * W + x;
  int rows = H - syi * sh < sh ? H - syi * sh : sh;
#pragma unroll 1
  for (int r = 0; r < rows; r += D) {
    float4 v[D], u[D], w[D];
#pragma unroll
    for (int k = 0; k < D; k++) if (r + k < rows && in) {
      long o = base + (long)(r + k) * W;
      v[k] = *(const float4*)(a + o);
      if (NR == 3) { u[k] = *(const float4*)(b + o); w[k] = *(const float4*)(c + o); }
    }
#pragma unroll
    for (int k = 0; k < D; k++) if (r + k < rows && writer) {
      long o = base + (long)(r + k) * W;
      if (NR == 3) { v[k].x += u[k].x * w[k].x; v[k].y += u[k].y * w[k].y; v[k].z += u[k].z * w[k].z; v[k].w += u[k].w * w[k].w; }
      *(float4*)(d + o) = v[k];
    }
  }
}

template <int NR, int D> void run_ostrip(float* a, float* b, float* c, float* d, int sh) {
  int strips_x = (W + 247) / 248;
  unsigned strips = strips_x * ((H + sh - 1) / sh);
  dim3 grid((strips + 3) / 4, F);
  double us = 0;
  {
    hipLaunchKernelGGL((ostrip_kernel<NR, D>), grid, dim3(256), 0, 0, a, b, c, d, sh, strips_x, strips);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((ostrip_kernel<NR, D>), grid, dim3(256), 0, 0, a, b, c, d, sh, strips_x, strips);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us = ms * 100;
  }
  double bytes = (double)NPX * 4 * (NR + 1);
  printf("overlapped strip (248-px step) sh=%d D=%d      NR=%d  %8.1f us  %6.0f GB/s\n", sh, D, NR, us, bytes / us / 1e3);
}

template <typename F> double timeit(F f, int n = 10) {
  f(); f(); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < n; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / n;
}

template <int NR> void report(const char* name, double us) {
  double bytes = (double)NPX * 4 * (NR + 1);
  printf("%-44s NR=%d  %8.1f us  %6.0f GB/s\n", name, NR, us, bytes / us / 1e3);
}

template <int NR, int D, bool SWZ> void run_strip(const char* name, float* a, float* b, float* c, float* d, int sh) {
  int strips_x = W / 256;  // 15 full strips (3840 = 15 * 256)
  unsigned strips = strips_x * ((H + sh - 1) / sh);
  dim3 grid((strips + 3) / 4, F);
  double us = timeit([&] { hipLaunchKernelGGL((strip_kernel<NR, D, SWZ>), grid, dim3(256), 0, 0, a, b, c, d, sh, strips_x, strips); });
  char buf[96]; snprintf(buf, sizeof buf, "%s sh=%d D=%d swz=%d", name, sh, D, (int)SWZ);
  report<NR>(buf, us);
}

int main() {
  float *a, *b, *c, *d;
  size_t bytes = NPX * 4;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes));
  long n4 = NPX / 4;
  report<1>("hipMemcpy d2d", timeit([&] { CK(hipMemcpyAsync(d, a, bytes, hipMemcpyDeviceToDevice, 0)); }));
  report<1>("linear float4", timeit([&] { hipLaunchKernelGGL(linear_kernel<1>, dim3((n4 + 255) / 256), dim3(256), 0, 0, (float4*)a, (float4*)b, (float4*)c, (float4*)d, n4); }));
  report<3>("linear float4", timeit([&] { hipLaunchKernelGGL(linear_kernel<3>, dim3((n4 + 255) / 256), dim3(256), 0, 0, (float4*)a, (float4*)b, (float4*)c, (float4*)d, n4); }));
  for (int nb : {2048, 4096, 8192}) {
    char nm[64]; snprintf(nm, sizeof nm, "gridstride U=4 blocks=%d", nb);
    report<1>(nm, timeit([&] { hipLaunchKernelGGL((gridstride_kernel<1, 4>), dim3(nb), dim3(256), 0, 0, (float4*)a, (float4*)b, (float4*)c, (float4*)d, n4); }));
    report<3>(nm, timeit([&] { hipLaunchKernelGGL((gridstride_kernel<3, 4>), dim3(nb), dim3(256), 0, 0, (float4*)a, (float4*)b, (float4*)c, (float4*)d, n4); }));
  }
  for (int sh : {32}) { run_ostrip<1, 1>(a, b, c, d, sh); run_ostrip<1, 4>(a, b, c, d, sh); run_ostrip<3, 1>(a, b, c, d, sh); run_ostrip<3, 2>(a, b, c, d, sh); }
  for (int sh : {32}) {
    run_strip<1, 1, true>("strip", a, b, c, d, sh);
    run_strip<1, 4, true>("strip", a, b, c, d, sh);
    run_strip<1, 8, true>("strip", a, b, c, d, sh);
    run_strip<1, 8, false>("strip", a, b, c, d, sh);
    run_strip<3, 1, true>("strip", a, b, c, d, sh);
    run_strip<3, 2, true>("strip", a, b, c, d, sh);
    run_strip<3, 4, true>("strip", a, b, c, d, sh);
    run_strip<3, 4, false>("strip", a, b, c, d, sh);
  }
  return 0;
}
