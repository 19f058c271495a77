// region_micro.hip — round-5 probe (NOT part of the product): WHICH access patterns does a "slow"
// region of the device memory punish?  24 blocks of 2025 MiB (the 64 x 4K batch) are drawn as the
// driver hands them out and classed by a store-only strip kernel; then store-only and load-only
// kernels of several shapes run on the fastest and on the slowest block:
//   linear      thread i <-> float4 i
//   strips g0   256-px aligned strips, 72 rows, one dwordx4 per lane and row
//   strips g1   the 248-px step of the product's sampling kernels (lanes 1 .. 62 store)
//   order 0     workgroup = 4 FRAMES of one strip, frame groups fastest (the product's fused kernels)
//   order 1     workgroup = 4 consecutive strips of one frame, frame after frame
//   hipcc --offload-arch=gfx950 -O3 tools/region_micro.hip -o /tmp/region_micro && /tmp/region_micro
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

template <bool LOAD, bool NT>
__global__ void __launch_bounds__(256) linear_k(float* p, float* sink, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  if constexpr (LOAD) {
    v4f v = ((const v4f*)p)[i];
    if (v.x == 1234.5f) sink[0] = v.y;
  } else {
    v4f v = v4f{(float)i, 1.f, 2.f, 3.f};
    if constexpr (NT) __builtin_nontemporal_store(v, (v4f*)p + i);
    else ((v4f*)p)[i] = v;
  }
}

// GEO 0 aligned / 1 = 248 step; ORDER 0 frames of a strip / 1 strips of a frame
// ORDER 4: workgroup of 16 waves = 4 frames x 4 consecutive strips, frame groups fastest
// ORDER 5: order 0 with the frame groups a quarter at a time (the product's fused kernels since round 5)
// ORDER 6: order 4 with the frame groups a quarter at a time
template <bool LOAD, int GEO, int ORDER>
__global__ void __launch_bounds__(ORDER == 4 || ORDER == 6 ? 1024 : 256) strips_k(float* p, float* sink, int sh, int strips_y) {
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ORDER 2 / 3: orders 1 / 0 WITHOUT the XCD-contiguous block order (consecutive blocks on different XCDs)
  const unsigned b = ORDER >= 2 ? blockIdx.x : xcd_swizzle(blockIdx.x, gridDim.x);
  constexpr unsigned SX = GEO ? 16 : 15;
  const unsigned strips = SX * strips_y;
  unsigned frame, sid;
  if constexpr (ORDER == 0 || ORDER == 3) {
    const unsigned groups = F / 4;
    frame = (b % groups) * 4 + wave;
    sid = b / groups;
  } else if constexpr (ORDER == 5) {
    const unsigned gc = F / 16, per = gc * strips, chunk = b / per, r = b - chunk * per;
    frame = (chunk * gc + r % gc) * 4 + wave;
    sid = r / gc;
  } else if constexpr (ORDER == 4 || ORDER == 6) {
    const unsigned groups = F / 4, quads = (strips + 3) / 4;
    unsigned g, q;
    if constexpr (ORDER == 4) { g = b % groups; q = b / groups; }
    else { const unsigned gc = F / 16, per = gc * quads, chunk = b / per, r = b - chunk * per; g = chunk * gc + r % gc; q = r / gc; }
    frame = g * 4 + (wave & 3u);
    sid = q * 4 + (wave >> 2);
  } else {
    const unsigned g = b * 4 + wave;
    frame = g / strips;
    sid = g % strips;
  }
  if (frame >= (unsigned)F || sid >= strips) return;
  const unsigned syi = sid / SX, sxi = sid % SX;
  const unsigned lane = threadIdx.x & 63;
  const int xs = GEO ? (int)sxi * 248 - 4 : (int)sxi * 256;
  if (xs < 0 || xs + 256 > W) return;   // (the two rim strips of geo 1: skipped, the time is scaled)
  const bool writer = GEO ? (lane >= 1 && lane < 63) : true;
  float* base = p + (long)frame * W * H + (long)syi * sh * W + xs;
  float acc = 0.f;
#pragma unroll 2
  for (int r = 0; r < sh; r++) {
    v4f* q = (v4f*)(base + (long)r * W + 4u * lane);
    if constexpr (LOAD) {
      if (writer) { v4f v = *q; acc += v.x + v.w; }
    } else {
      if (writer) __builtin_nontemporal_store(v4f{(float)r, acc, 2.f, 3.f}, q);
    }
  }
  if (LOAD && acc == 1234.5f) sink[0] = acc;
}

template <typename Fn> static double timeit(Fn launch, int reps = 10) {
  for (int i = 0; i < 3; i++) launch();
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
  const size_t bytes = (size_t)F * W * H * 4;
  const int sh = 72, sy = H / sh;
  std::vector<float*> blk(nblk);
  float* sink; CK(hipMalloc(&sink, 64));
  for (auto& b : blk) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes)); }
  CK(hipDeviceSynchronize());
  auto st = [&](float* p, auto kern, unsigned sx) { return timeit([&] { hipLaunchKernelGGL(kern, dim3(sx * sy * (F / 4)), dim3(256), 0, 0, p, sink, sh, sy); }); };
  auto st16 = [&](float* p, auto kern, unsigned sx) { return timeit([&] { hipLaunchKernelGGL(kern, dim3((sx * sy + 3) / 4 * (F / 4)), dim3(1024), 0, 0, p, sink, sh, sy); }); };
  // settle the clocks, then class the blocks: store-only, aligned strips, order 0
  for (int i = 0; i < 200; i++) hipLaunchKernelGGL((strips_k<false, 0, 0>), dim3(15 * sy * (F / 4)), dim3(256), 0, 0, blk[0], sink, sh, sy);
  std::vector<std::pair<double, int>> cls;
  printf("blocks in allocation order: store-only aligned strips (order 0), us\n");
  for (int i = 0; i < nblk; i++) {
    const double t = st(blk[i], strips_k<false, 0, 0>, 15);
    cls.push_back({t, i});
    printf("  %2d  %p  %.1f\n", i, (void*)blk[i], t);
  }
  std::sort(cls.begin(), cls.end());
  const int fast = cls.front().second, slow = cls.back().second, mid = cls[nblk / 2].second;
  printf("fast block %d (%.1f), median block %d (%.1f), slow block %d (%.1f)\n", fast, cls.front().first, mid, cls[nblk / 2].first, slow, cls.back().first);
  const long n4 = bytes / 16;
  const double g1 = 3840.0 / (14 * 248);   // geo 1 skips 2 of its 16 strips
  for (int pass = 0; pass < 2; pass++) {
    printf("%-44s %10s %10s %10s   (us per 2.1 GB)\n", "", "fast", "median", "slow");
    auto row = [&](const char* name, auto fn) {
      printf("%-44s %10.1f %10.1f %10.1f\n", name, fn(blk[fast]), fn(blk[mid]), fn(blk[slow]));
      fflush(stdout);
    };
    row("store linear nt", [&](float* p) { return timeit([&] { hipLaunchKernelGGL((linear_k<false, true>), dim3((n4 + 255) / 256), dim3(256), 0, 0, p, sink, n4); }); });
    row("store linear plain", [&](float* p) { return timeit([&] { hipLaunchKernelGGL((linear_k<false, false>), dim3((n4 + 255) / 256), dim3(256), 0, 0, p, sink, n4); }); });
    row("store strips aligned, order 0", [&](float* p) { return st(p, strips_k<false, 0, 0>, 15); });
    row("store strips aligned, order 1", [&](float* p) { return st(p, strips_k<false, 0, 1>, 15); });
    row("store strips aligned, order 1, no XCD swizzle", [&](float* p) { return st(p, strips_k<false, 0, 2>, 15); });
    row("store strips aligned, order 0, no XCD swizzle", [&](float* p) { return st(p, strips_k<false, 0, 3>, 15); });
    row("store strips 248-step, order 0", [&](float* p) { return st(p, strips_k<false, 1, 0>, 16) * g1; });
    row("store strips 248-step, order 0, no XCD swizzle", [&](float* p) { return st(p, strips_k<false, 1, 3>, 16) * g1; });
    row("store strips 248-step, order 1, no XCD swizzle", [&](float* p) { return st(p, strips_k<false, 1, 2>, 16) * g1; });
    row("store strips 248-step, order 1", [&](float* p) { return st(p, strips_k<false, 1, 1>, 16) * g1; });
    row("store strips 248-step, order 0, groups a quarter at a time", [&](float* p) { return st(p, strips_k<false, 1, 5>, 16) * g1; });
    row("store strips 248-step, 4 frames x 4 strips per WG", [&](float* p) { return st16(p, strips_k<false, 1, 4>, 16) * g1; });
    row("store strips 248-step, 4 x 4 per WG, a quarter at a time", [&](float* p) { return st16(p, strips_k<false, 1, 6>, 16) * g1; });
    row("store strips aligned, order 0, a quarter at a time", [&](float* p) { return st(p, strips_k<false, 0, 5>, 15); });
    row("store strips aligned, 4 x 4 per WG, a quarter at a time", [&](float* p) { return st16(p, strips_k<false, 0, 6>, 15); });
    row("load linear", [&](float* p) { return timeit([&] { hipLaunchKernelGGL((linear_k<true, false>), dim3((n4 + 255) / 256), dim3(256), 0, 0, p, sink, n4); }); });
    row("load strips aligned, order 0", [&](float* p) { return st(p, strips_k<true, 0, 0>, 15); });
    row("load strips aligned, order 1", [&](float* p) { return st(p, strips_k<true, 0, 1>, 15); });
    row("load strips 248-step, order 0", [&](float* p) { return st(p, strips_k<true, 1, 0>, 16) * g1; });
  }
  return 0;
}
