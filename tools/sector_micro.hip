// sector_micro.hip — round-5 probe (NOT part of the product): does the WIDTH of the per-lane
// loads that bring a source row into the CU decide how fast the marching-strip shape streams?
// The fused kernel brings its rows in by dword gathers (64 lanes x 4 B = 256 contiguous bytes per
// instruction, 64-byte sector requests to L2: 39.6 M read requests per 64 x 4K launch against
// 16.6 M for a copy of the same bytes, profiles/r05_micro.txt), the plain filter by one dwordx4
// per lane.  Same strips, same order (4 frames of a strip per workgroup, frame groups fastest),
// same stores (one nt dwordx4 per lane and row); the row loads are
//   0  one dwordx4 per lane (1 KB per instruction)
//   1  four dwords per lane, lane-interleaved (pixel L + 64 k): 4 x 256 B
//   2  two dwordx2 per lane (pixels 2L, 2L+1 + 128 j): 2 x 512 B
//   3  as 1, the row read from an odd column offset (+13 px): every 256-byte piece straddles lines
//   4  as 2, from the odd offset
//   5  as 1 plus the dword at +4 bytes of every piece (the right-hand taps): 8 x 256 B
//   hipcc --offload-arch=gfx950 -O3 tools/sector_micro.hip -o /tmp/sector_micro && /tmp/sector_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int W = 3840, H = 2160;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned n) {
  unsigned per = n / 8;
  if (per * 8 != n) return b;
  return (b % 8) * per + b / 8;
}

// GEO 0: 256-px aligned strips (15 per 4K row, all lanes store whole lines); GEO 1: the 248-px step of
// the product's sampling kernels (16 strips per row, lanes 1 .. 62 store: 992-byte rows that start
// 16 bytes into a line); GEO 2: a 240-px step (16 strips per row, lanes 2 .. 61 store: 960-byte rows that
// start on a 64-byte sector and end on one - no sector of the result is written by two waves);
// MODE 6: no reads at all (the store stream alone)
template <int MODE, int D, int GEO = 0, int NT = 1>
__global__ void __launch_bounds__(256) strips(const float* a, float* d, int sh, int strips_y, int frames, int gchunk = 0) {
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned groups = frames / 4;
  unsigned frame = (b % groups) * 4 + wave;
  unsigned sid = b / groups;
  if (gchunk) {   // the frame groups gchunk at a time (the product's fused kernels since round 5)
    const unsigned nstr = (GEO ? 16u : 15u) * (unsigned)strips_y, per = (unsigned)gchunk * nstr;
    const unsigned chunk = b / per, r = b - chunk * per;
    frame = (chunk * gchunk + r % gchunk) * 4 + wave;
    sid = r / gchunk;
  }
  constexpr unsigned SX = GEO ? 16 : 15;
  const unsigned syi = sid / SX, sxi = sid % SX;
  if (syi >= (unsigned)strips_y) return;
  const unsigned lane = threadIdx.x & 63;
  const int xs = GEO == 1 ? (int)sxi * 248 - 4 : (GEO == 2 ? (int)sxi * 240 - 8 : (int)sxi * 256);
  const bool inside = xs >= 0 && xs + 256 + 16 <= W;   // (rim strips of GEO 1: skipped, 2 of 16)
  if (!inside) return;
  const bool writer = GEO == 1 ? (lane >= 1 && lane < 63) : (GEO == 2 ? (lane >= 2 && lane < 62) : true);
  const long base = (long)frame * W * H + (long)syi * sh * W + xs;
  constexpr int OFF = (MODE == 3 || MODE == 4) ? 13 : 0;
  const float* ap = a + base + OFF;
  float* dp = d + base;
  __shared__ float row[4][D][264];
#pragma unroll 1
  for (int r = 0; r < sh; r += D) {
    float v[D][8];
#pragma unroll
    for (int k = 0; k < D; k++) {
      const float* rp = ap + (long)(r + k) * W;
      if constexpr (MODE == 6) {
        v[k][0] = v[k][1] = v[k][2] = v[k][3] = (float)(r + k);
      } else if constexpr (MODE == 0) {
        v4f q = *(const v4f*)(rp + 4u * lane);
        v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = q.w;
      } else if constexpr (MODE == 1 || MODE == 3 || MODE == 5) {
#pragma unroll
        for (int j = 0; j < 4; j++) v[k][j] = __builtin_nontemporal_load(rp + lane + 64u * j) ;
        if constexpr (MODE == 5) {
#pragma unroll
          for (int j = 0; j < 4; j++) v[k][4 + j] = rp[lane + 64u * j + 1];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          v2f q = *(const v2f*)(rp + 2u * lane + 128u * j);
          v[k][2 * j] = q.x; v[k][2 * j + 1] = q.y;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < D; k++) {
      v4f o;
      if constexpr (MODE == 0 || MODE == 6) {
        o = v4f{v[k][0], v[k][1], v[k][2], v[k][3]};
      } else {
        // back into pixel order through the wave's LDS row (what the fused kernel does)
        float* xp = row[wave][k];
        if constexpr (MODE == 1 || MODE == 3 || MODE == 5) {
#pragma unroll
          for (int j = 0; j < 4; j++) xp[64u * j + lane] = v[k][j] + (MODE == 5 ? v[k][4 + j] * 1e-30f : 0.f);
        } else {
#pragma unroll
          for (int j = 0; j < 2; j++) *(v2f*)(xp + 128u * j + 2u * lane) = v2f{v[k][2 * j], v[k][2 * j + 1]};
        }
        __builtin_amdgcn_wave_barrier();
        o = *(const v4f*)(xp + 4u * lane);
        __builtin_amdgcn_wave_barrier();
      }
      // NT 1: every store non-temporal (the product's sampling kernels); 0: plain stores; 2: plain stores for
      // the lanes whose 16 bytes lie in a line this strip row does not cover whole, non-temporal for the rest
      v4f* op = (v4f*)(dp + (long)(r + k) * W + 4u * lane);
      bool plain = NT == 0;
      if constexpr (NT == 2) {
        const unsigned long a0 = (unsigned long)(dp + (long)(r + k) * W + 4) * 1, a1 = (unsigned long)(dp + (long)(r + k) * W + 252);
        const unsigned long me = (unsigned long)op;
        plain = (me & ~127ul) < a0 || (me & ~127ul) + 128 > a1;
      }
      if (writer) {
        if (plain) *op = o;
        else __builtin_nontemporal_store(o, op);
      }
    }
  }
}

static int g_chunk = 0;
template <int MODE, int D, int GEO = 0, int NT = 1> static void run(const char* name, const float* a, float* d, int frames, int sh) {
  const int strips_y = H / sh;
  dim3 grid((GEO ? 16 : 15) * strips_y * (frames / 4)), block(256);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((strips<MODE, D, GEO, NT>), grid, block, 0, 0, a, d, sh, strips_y, frames, g_chunk);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((strips<MODE, D, GEO, NT>), grid, block, 0, 0, a, d, sh, strips_y, frames, g_chunk);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  // (GEO 1 skips its two rim strips of 16: the time is scaled to the whole frame)
  if (GEO) ms *= 3840.0 / (14 * (GEO == 2 ? 240 : 248));
  printf("%-56s D=%d geo %d nt %d  %8.1f us  %6.0f GB/s\n", name, D, GEO, NT, ms * 1e3, (MODE == 6 ? 1.0 : 2.0) * frames * W * H * 4 / (ms * 1e-3) / 1e9);
  fflush(stdout);
}

__global__ void fill(float* p, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = (float)(i & 1023) * 0.001f;
}

int main(int argc, char** argv) {
  const int frames = 64, sh = 72;
  const long n = (long)frames * W * H + 4096;
  float *a, *d;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&d, n * 4));
  hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, a, n);
  CK(hipDeviceSynchronize());
  const bool one = argc > 1;   // counter mode: one pass
  if (argc > 3) {   // the three geometries under the product's dispatch order (frame groups 4 at a time), 144-row strips
    g_chunk = 4;
    for (int rep = 0; rep < 3; rep++) {
      run<3, 2, 0>("3 gather copy", a, d, frames, 144);
      run<3, 2, 1>("3 gather copy", a, d, frames, 144);
      run<3, 2, 2>("3 gather copy", a, d, frames, 144);
      run<6, 2, 0>("6 store stream alone", a, d, frames, 144);
      run<6, 2, 1>("6 store stream alone", a, d, frames, 144);
      run<6, 2, 2>("6 store stream alone", a, d, frames, 144);
    }
    return 0;
  }
  if (argc > 2) {   // chunk sweep: gather copy in both geometries by the chunk of frame groups
    for (int rep = 0; rep < 2; rep++)
      for (int gc : {0, 8, 4, 2, 1}) {
        g_chunk = gc;
        char nm[96];
        snprintf(nm, sizeof nm, "3 four dwords, odd offset, groups %d at a time", gc ? gc : 16);
        run<3, 2, 1>(nm, a, d, frames, 144);
        run<3, 2, 0>(nm, a, d, frames, 144);
        run<6, 2, 1>("6 store stream alone, same order", a, d, frames, 144);
      }
    return 0;
  }
  for (int rep = 0; rep < (one ? 1 : 2); rep++) {
    run<0, 2>("0 one dwordx4 per lane", a, d, frames, sh);
    run<1, 2>("1 four dwords per lane, interleaved", a, d, frames, sh);
    run<2, 2>("2 two dwordx2 per lane", a, d, frames, sh);
    run<3, 2>("3 four dwords, odd column offset", a, d, frames, sh);
    run<4, 2>("4 two dwordx2, odd column offset", a, d, frames, sh);
    run<5, 2>("5 four dwords + the dword at +4 bytes", a, d, frames, sh);
    run<6, 2>("6 no reads: the store stream alone", a, d, frames, sh);
    run<6, 2, 1>("6 no reads: the store stream alone", a, d, frames, sh);
    run<0, 2, 1>("0 one dwordx4 per lane", a, d, frames, sh);
    run<1, 2, 1>("1 four dwords per lane, interleaved", a, d, frames, sh);
    run<3, 2, 1>("3 four dwords, odd column offset", a, d, frames, sh);
    run<5, 2, 1>("5 four dwords + the dword at +4 bytes", a, d, frames, sh);
    run<6, 2, 1, 0>("6 no reads: the store stream alone, plain stores", a, d, frames, sh);
    run<6, 2, 1, 2>("6 no reads: the store stream alone, plain on partial lines", a, d, frames, sh);
    run<3, 2, 1, 0>("3 four dwords, odd column offset, plain stores", a, d, frames, sh);
    run<3, 2, 1, 2>("3 four dwords, odd column offset, plain on partial lines", a, d, frames, sh);
    run<3, 2, 0, 0>("3 four dwords, odd column offset, plain stores", a, d, frames, sh);
    run<6, 2, 0, 0>("6 no reads: the store stream alone, plain stores", a, d, frames, sh);
    run<6, 2>("6 no reads, 144-row strips", a, d, frames, 144);
    run<6, 2, 1>("6 no reads, 144-row strips", a, d, frames, 144);
    run<3, 2>("3 four dwords, odd column offset, 144-row strips", a, d, frames, 144);
    run<3, 2, 1>("3 four dwords, odd column offset, 144-row strips", a, d, frames, 144);
  }
  return 0;
}
