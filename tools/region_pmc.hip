// region_pmc.hip — round-6 probe (NOT part of the product): the memory-side counters of the strip-shaped
// STORE stream on a FAST and on a SLOW block of the device memory, in one process (review of round 4, item 2b;
// round 5, item 1b).  24 blocks of 2025 MiB are drawn as the driver hands them out and classed by the
// store-only strip kernel of tools/region_micro.hip (240-px step of the product, frame groups a quarter at a
// time); the same kernels then run on the fastest and on the slowest block under DIFFERENT NAMES
// (template tag 0 = fast, 1 = slow), so that `rocprofv3 --pmc ... --kernel-trace` reports their counters apart:
//   strips_store<TAG>   the product's store stream (lanes 2 .. 61 of a 256-px window store a dwordx4 per row)
//   linear_store<TAG>   thread i <-> float4 i
//   strips_load<TAG>    the same strips, loads only
//   hipcc --offload-arch=gfx950 -O3 tools/region_pmc.hip -o tools/region_pmc.bin
//   tools/r06_pmc_region.sh runs it under one rocprofv3 pass per counter group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int W = 3840, H = 2160, F = 64;
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned n) {
  unsigned per = n / 8;
  if (per * 8 != n) return b;
  return (b % 8) * per + b / 8;
}

template <bool LOAD>
__device__ __forceinline__ void strips_body(float* p, float* sink, int sh, int strips_y) {
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  constexpr unsigned SX = 16;   // 240-px step: 16 strips per 3840-px row, window = 256 px starting at 240 s - 8
  const unsigned strips = SX * strips_y;
  const unsigned gc = F / 16, per = gc * strips, chunk = b / per, r = b - chunk * per;
  const unsigned frame = (chunk * gc + r % gc) * 4 + wave, sid = r / gc;
  if (frame >= (unsigned)F || sid >= strips) return;
  const unsigned syi = sid / SX, sxi = sid % SX;
  const unsigned lane = threadIdx.x & 63;
  const int xs = (int)sxi * 240 - 8;
  const bool writer = lane >= 2 && lane < 62;
  float* base = p + (long)frame * W * H + (long)syi * sh * W + xs;
  float acc = 0.f;
#pragma unroll 2
  for (int rr = 0; rr < sh; rr++) {
    v4f* q = (v4f*)(base + (long)rr * W + 4u * lane);
    if constexpr (LOAD) {
      if (writer) { v4f v = *q; acc += v.x + v.w; }
    } else {
      if (writer) __builtin_nontemporal_store(v4f{(float)rr, acc, 2.f, 3.f}, q);
    }
  }
  if (LOAD && acc == 1234.5f) sink[0] = acc;
}
template <int TAG> __global__ void __launch_bounds__(256) strips_store(float* p, float* sink, int sh, int sy) { strips_body<false>(p, sink, sh, sy); }
template <int TAG> __global__ void __launch_bounds__(256) strips_load(float* p, float* sink, int sh, int sy) { strips_body<true>(p, sink, sh, sy); }
template <int TAG> __global__ void __launch_bounds__(256) linear_store(float* p, float* sink, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  __builtin_nontemporal_store(v4f{(float)i, 1.f, 2.f, 3.f}, (v4f*)p + i);
}

template <typename Fn> static double timeit(Fn launch, int reps) {
  for (int i = 0; i < 2; i++) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 24;
  const int reps = argc > 2 ? atoi(argv[2]) : 6;
  const size_t bytes = (size_t)F * W * H * 4;
  const int sh = 72, sy = H / sh;
  std::vector<float*> blk(nblk);
  float* sink; CK(hipMalloc(&sink, 64));
  for (auto& b : blk) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes)); }
  CK(hipDeviceSynchronize());
  const dim3 grid(16 * sy * (F / 4)), block(256);
  // class the blocks with a THIRD name (tag 2) so that the classing launches stay out of both rows
  for (int i = 0; i < 100; i++) hipLaunchKernelGGL((strips_store<2>), grid, block, 0, 0, blk[0], sink, sh, sy);
  std::vector<std::pair<double, int>> cls;
  printf("blocks in allocation order: store-only 240-step strips, us\n");
  for (int i = 0; i < nblk; i++) {
    float* p = blk[i];
    const double t = timeit([&] { hipLaunchKernelGGL((strips_store<2>), grid, block, 0, 0, p, sink, sh, sy); }, 6);
    cls.push_back({t, i});
    printf("  %2d  %p  %.1f\n", i, (void*)p, t);
  }
  std::sort(cls.begin(), cls.end());
  float* fast = blk[cls.front().second];
  float* slow = blk[cls.back().second];
  printf("fast block %d (%.1f us) %p, slow block %d (%.1f us) %p\n", cls.front().second, cls.front().first, (void*)fast,
         cls.back().second, cls.back().first, (void*)slow);
  const long n4 = bytes / 16;
  const dim3 lgrid((unsigned)((n4 + 255) / 256));
  for (int pass = 0; pass < 2; pass++) {
    const double a0 = timeit([&] { hipLaunchKernelGGL((strips_store<0>), grid, block, 0, 0, fast, sink, sh, sy); }, reps);
    const double a1 = timeit([&] { hipLaunchKernelGGL((strips_store<1>), grid, block, 0, 0, slow, sink, sh, sy); }, reps);
    const double b0 = timeit([&] { hipLaunchKernelGGL((linear_store<0>), lgrid, block, 0, 0, fast, sink, n4); }, reps);
    const double b1 = timeit([&] { hipLaunchKernelGGL((linear_store<1>), lgrid, block, 0, 0, slow, sink, n4); }, reps);
    const double c0 = timeit([&] { hipLaunchKernelGGL((strips_load<0>), grid, block, 0, 0, fast, sink, sh, sy); }, reps);
    const double c1 = timeit([&] { hipLaunchKernelGGL((strips_load<1>), grid, block, 0, 0, slow, sink, sh, sy); }, reps);
    printf("pass %d  strips_store fast %.1f slow %.1f | linear_store fast %.1f slow %.1f | strips_load fast %.1f slow %.1f  (us)\n",
           pass, a0, a1, b0, b1, c0, c1);
  }
  return 0;
}
