// pipe_micro.hip — round-3 probes for the marching-wave skeleton (NOT part of the product).
//
//  1. vmcnt rules the hand-scheduled pipeline of wave_stencil.hpp relies on:
//       order   a cold load L followed by a store S: does `s_waitcnt vmcnt(1)` guarantee L's data?
//               (i.e. do loads and stores of one wave retire in issue order on gfx950)
//       exec0   a cold load L followed by 4 loads issued with EXEC = 0: does `vmcnt(4)` still
//               guarantee L's data (are EXEC = 0 vector-memory instructions counted)?
//  2. the stream rate of launch shapes over 64 x 4K float32 (2.1 GB in, 2.1 GB out, random data):
//       linear float4 sweep, hipMemcpy, read-only, write-only,
//       wave strips (256 px x 72 rows per wave) with the compiler's waits (chunks of 8 rows) and
//       with a rolling software pipeline of P rows in flight (asm loads / stores / counted waits),
//       in four dispatch orders (frames of a strip in one workgroup, strips of a frame in one
//       workgroup, full-width row bands of 15 waves, ...).
//
//   hipcc --offload-arch=gfx950 -O3 tools/pipe_micro.hip -o gpurun_out/pipe_micro && gpurun_out/pipe_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int W = 3840, H = 2160;
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned n) {
  unsigned per = n / 8;
  if (per * 8 != n) return b;
  return (b % 8) * per + b / 8;
}
__device__ __host__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ void fill_kernel(unsigned* p, long n, int as_float) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h = hash32((unsigned)i);
  if (as_float) { float f = (float)(h >> 8) * (1.0f / 16777216.0f); p[i] = __float_as_uint(f); }
  else p[i] = h;
}

// ------------------------------------------------------------------ vmcnt probes --
// every lane: ITER cold loads from pseudo-random places of a 2 GB table whose word i holds
// hash32(i); a wrong register value after the counted wait = the wait did not cover the load.
template <int MODE>
__global__ void __launch_bounds__(256) probe_kernel(const unsigned* table, unsigned words_mask,
                                                    unsigned* scratch, unsigned* bad, int iters) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  unsigned seed = hash32(tid * 2654435761u + 12345u);
  unsigned nbad = 0;
  for (int it = 0; it < iters; it++) {
    seed = hash32(seed + it);
    const unsigned idx = seed & words_mask;
    const unsigned loff = idx * 4u;                 // byte offset < 2^31 * ... (mask keeps it < 2 GB)
    const unsigned soff = (tid * 4u) & 0xfffffffu;
    unsigned r, d1, d2, d3, d4;
    if constexpr (MODE == 0) {
      asm volatile(
          "v_mov_b32 %0, 0xdeadbeef\n\t"
          "s_nop 4\n\t"
          "global_load_dword %0, %1, %2\n\t"
          "global_store_dword %3, %4, %5\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          : "=&v"(r)
          : "v"(loff), "s"(table), "v"(soff), "v"(seed), "s"(scratch)
          : "memory");
    } else if constexpr (MODE == 1) {
      unsigned long long sv;
      asm volatile(
          "v_mov_b32 %0, 0xdeadbeef\n\t"
          "s_nop 4\n\t"
          "global_load_dword %0, %6, %7\n\t"
          "s_mov_b64 %5, exec\n\t"
          "s_mov_b64 exec, 0\n\t"
          "global_load_dword %1, %6, %7 offset:4\n\t"
          "global_load_dword %2, %6, %7 offset:8\n\t"
          "global_load_dword %3, %6, %7 offset:12\n\t"
          "global_load_dword %4, %6, %7 offset:16\n\t"
          "s_mov_b64 exec, %5\n\t"
          "s_waitcnt vmcnt(4)\n\t"
          : "=&v"(r), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&s"(sv)
          : "v"(loff), "s"(table)
          : "memory");
    } else {  // MODE 2: store first (older), then the load, wait vmcnt(0): reference (must be 0 bad)
      asm volatile(
          "v_mov_b32 %0, 0xdeadbeef\n\t"
          "s_nop 4\n\t"
          "global_store_dword %3, %4, %5\n\t"
          "global_load_dword %0, %1, %2\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          : "=&v"(r)
          : "v"(loff), "s"(table), "v"(soff), "v"(seed), "s"(scratch)
          : "memory");
    }
    nbad += (r != hash32(idx)) ? 1u : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (nbad) atomicAdd(bad, nbad);
}

// ------------------------------------------------------------------ stream shapes --
__global__ void __launch_bounds__(256) linear_copy(const v4f* a, v4f* d, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) d[i] = a[i];
}
__global__ void __launch_bounds__(256) linear_copy_nt(const v4f* a, v4f* d, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) __builtin_nontemporal_store(a[i], d + i);
}
__global__ void __launch_bounds__(256) linear_read(const v4f* a, float* sink, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  v4f v = a[i];
  if (v.x + v.y + v.z + v.w == 123456.789f) sink[0] = v.x;
}
__global__ void __launch_bounds__(256) linear_write(v4f* d, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) d[i] = v4f{(float)i, 1.f, 2.f, 3.f};
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// strip id / frame of this wave by dispatch order
//   0: workgroup = 4 FRAMES of one strip, frame groups fastest, then strips (frames_wg, the product's order)
//   1: workgroup = 4 consecutive strips of one frame, frame after frame
//   2: workgroup = 4 consecutive strips of one frame, the frames of a strip block neighbours (frames_inner)
//   3: workgroup = 15 waves = one full-width row band, frames fastest
//   4: workgroup = 15 waves = one full-width row band, frame after frame
template <int MODE>
__device__ __forceinline__ bool strip_of(unsigned& frame, unsigned& sxi, unsigned& syi, int strips_y, int frames) {
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned strips = 15u * strips_y;
  if constexpr (MODE == 0) {
    const unsigned groups = frames / 4;
    frame = (b % groups) * 4 + wave;
    unsigned sid = b / groups;
    syi = sid / 15; sxi = sid % 15;
    return sid < strips;
  } else if constexpr (MODE == 1) {
    unsigned g = b * 4 + wave;
    frame = g / strips;
    unsigned sid = g % strips;
    syi = sid / 15; sxi = sid % 15;
    return frame < (unsigned)frames;
  } else if constexpr (MODE == 2) {
    frame = b % frames;
    unsigned sid = (b / frames) * 4 + wave;
    syi = sid / 15; sxi = sid % 15;
    return sid < strips;
  } else if constexpr (MODE == 3) {
    frame = b % frames;
    syi = b / frames; sxi = wave;
    return syi < (unsigned)strips_y;
  } else {
    frame = b / strips_y;
    syi = b % strips_y; sxi = wave;
    return frame < (unsigned)frames;
  }
}

// compiler-managed waits: chunks of D rows (what wave_stencil.hpp's LoadRowSrc does today)
template <int D, int MODE>
__global__ void __launch_bounds__(MODE >= 3 ? 960 : 256)
strip_chunk(const float* a, float* d, int sh, int strips_y, int frames) {
  unsigned frame, sxi, syi;
  if (!strip_of<MODE>(frame, sxi, syi, strips_y, frames)) return;
  const unsigned lane = threadIdx.x & 63;
  const long base = (long)frame * W * H + (long)syi * sh * W + sxi * 256;
  const float* ap = a + base;
  float* dp = d + base;
#pragma unroll 1
  for (int r = 0; r < sh; r += D) {
    v4f v[D];
#pragma unroll
    for (int k = 0; k < D; k++) v[k] = *(const v4f*)(ap + (long)(r + k) * W + 4u * lane);
#pragma unroll
    for (int k = 0; k < D; k++) __builtin_nontemporal_store(v[k], (v4f*)(dp + (long)(r + k) * W + 4u * lane));
  }
}

__device__ __forceinline__ void gload4(v4f& x, unsigned voff, const float* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(x) : "v"(voff), "s"(sbase));
}
template <bool NT> __device__ __forceinline__ void gstore4(const v4f& x, unsigned voff, float* sbase) {
  if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" ::"v"(voff), "v"(x), "s"(sbase));
  else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 0" ::"v"(voff), "v"(x), "s"(sbase));
}
template <int N> __device__ __forceinline__ void wait_vm(v4f& x) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N));
}

// rolling pipeline: P rows in flight per wave; iteration r: wait row r, store row r, load row r + P
template <int P, int MODE, bool NT>
__global__ void __launch_bounds__(MODE >= 3 ? 960 : 256)
strip_pipe(const float* a, float* d, int sh, int strips_y, int frames) {
  unsigned frame, sxi, syi;
  if (!strip_of<MODE>(frame, sxi, syi, strips_y, frames)) return;
  const unsigned lane = threadIdx.x & 63;
  const long base = (long)frame * W * H + (long)syi * sh * W + sxi * 256;
  const float* ap = a + base;
  float* dp = d + base;
  const unsigned voff = 16u * lane;
  v4f buf[P];
  static_for<0, P>([&](auto K) { constexpr int k = decltype(K)::value; gload4(buf[k], voff, ap + (long)k * W); });
  // first chunk: loads younger than row k = (P - 1 - k) of the prologue + k reloads, stores k
  static_for<0, P>([&](auto K) {
    constexpr int k = decltype(K)::value;
    wait_vm<P - 1 + k>(buf[k]);
    gstore4<NT>(buf[k], voff, dp + (long)k * W);
    const int rn = P + k < sh ? P + k : sh - 1;  // past the strip: a dummy reload keeps the counts uniform
    gload4(buf[k], voff, ap + (long)rn * W);
  });
#pragma unroll 1
  for (int r = P; r < sh; r += P) {
    static_for<0, P>([&](auto K) {
      constexpr int k = decltype(K)::value;
      wait_vm<2 * P - 2>(buf[k]);
      gstore4<NT>(buf[k], voff, dp + (long)(r + k) * W);
      const int rn = r + P + k < sh ? r + P + k : sh - 1;
      gload4(buf[k], voff, ap + (long)rn * W);
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the geometry of wave_stencil.hpp's 5x5 and alternatives, as a copy with optional filter work.
// GEOM 0: 256-px aligned strips (15 per row), no halo rows                     [reference shape]
//      1: strips step 248 px: read 256 px from x = 248 sxi - 4, lanes 1..62 store, 76 rows in
//         per 72 rows out                                                       [wave_stencil.hpp today]
//      2: as 1 without the halo rows (72 in, 72 out)
//      4: 256-px aligned strips, 76 rows in per 72 out, the 2 + 2 halo pixels of a row by ONE
//         extra dwordx2 load of lanes 0 and 63 (EXEC-masked), all 64 lanes store   [proposal]
//      5: as 1 but loads aligned down to 128 B (stores unaligned 992 B)
//      6: as 1 but stores aligned down to 128 B, all lanes (loads unaligned)
// ORDER 0: workgroup = 4 frames of one strip, groups fastest (frames_wg); 1: workgroup = 4
//      adjacent strips of a frame, frame after frame; 2: as 0 within groups of 8 frames, group
//      after group
// WORK 0: copy only, 1: + the LDS row (ds_write_b128, 7 pair reads), 2: + 50 v_pk_fma_f32 per row
typedef float v2f __attribute__((ext_vector_type(2)));
template <int P, int WORK, int GEOM, int ORDER, bool NT = true>
__global__ void __launch_bounds__(256)
conv_like(const float* a, float* d, int sh, int strips_y, int frames, float w0) {
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  constexpr bool kAligned = GEOM == 0 || GEOM == 4;
  constexpr int strips_x = kAligned ? 15 : 16;
  unsigned frame, sid;
  if constexpr (ORDER == 0) {
    const unsigned groups = frames / 4;
    frame = (b % groups) * 4 + wave;
    sid = b / groups;
  } else if constexpr (ORDER == 1) {
    const unsigned g = b * 4 + wave, strips = strips_x * strips_y;
    frame = g / strips;
    sid = g % strips;
    if (frame >= (unsigned)frames) return;
  } else {
    const unsigned per = strips_x * strips_y * 2;  // blocks per group of 8 frames
    const unsigned grp = b / per, r = b % per;
    frame = grp * 8 + (r % 2) * 4 + wave;
    sid = r / 2;
    if (frame >= (unsigned)frames) return;
  }
  const unsigned syi = sid / strips_x, sxi = sid % strips_x;
  if (syi >= (unsigned)strips_y) return;
  const unsigned lane = threadIdx.x & 63;
  int xs = kAligned ? (int)sxi * 256 : (int)sxi * 248 - 4;
  if (xs < 0) xs = 0;
  if (xs > W - 256) xs = W - 256;
  int xl = xs, xst = xs;
  if (GEOM == 5) xl = xs & ~31;
  if (GEOM == 6) xst = xs & ~31;
  const bool writer = kAligned || GEOM == 6 || (lane >= 1 && lane < 63);
  const int y0 = (int)syi * sh;
  constexpr bool kHaloRows = GEOM == 1 || GEOM >= 4;
  const int T = kHaloRows ? sh + 4 : sh;
  const int yin = !kHaloRows ? y0 : (y0 - 2 < 0 ? 0 : (y0 - 2 + T > H ? H - T : y0 - 2));
  const float* ap = a + (long)frame * W * H + (long)yin * W + xl;
  float* dp = d + (long)frame * W * H + (long)y0 * W + xst;
  const unsigned voff = 16u * lane;
  // halo pair of GEOM 4: lane 0 -> the 2 px left of the strip, lane 63 -> the 2 px right of it
  const int hx = lane == 0 ? (xs >= 2 ? -2 : 0) : (xs + 258 <= W ? 256 : 254);
  const unsigned hoff = (unsigned)((hx + 2) * 4);   // relative to ap - 2 floats
  const unsigned long long hmask = 0x8000000000000001ull;
  __shared__ __attribute__((aligned(16))) float lds[4][272];
  float* xp = lds[wave];
  v4f buf[P];
  v2f hb[P];
  v2f acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; i++) acc[i][0] = acc[i][1] = v2f{0.f, 0.f};
  constexpr int OPS = GEOM == 4 ? 2 : 1;  // loads per row
  auto issue = [&](v4f& x, v2f& h, int row) {
    gload4(x, voff, ap + (long)row * W);
    if constexpr (GEOM == 4) {
      unsigned long long sv;
      asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\t"
                   "global_load_dwordx2 %0, %2, %3\n\ts_mov_b64 exec, %1"
                   : "=v"(h), "=&s"(sv)
                   : "v"(hoff), "s"(ap + (long)row * W - 2), "s"(hmask));
    }
  };
  static_for<0, P>([&](auto K) { constexpr int k = decltype(K)::value; issue(buf[k], hb[k], k); });
  const int lag = kHaloRows ? 4 : 0;
#pragma unroll 1
  for (int r = 0; r < T; r += P) {
    static_for<0, P>([&](auto K) {
      constexpr int k = decltype(K)::value;
      const int t = r + k;
      if (t < T) {
        // younger than row t's loads: the loads of rows t+1 .. t+P-1 and the stores of iterations
        // t-P .. t-1 (iteration j stores when j >= lag)
        if (t >= P + lag) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS * (P - 1) + P));
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS * (P - 1)));
        asm volatile("; pin %0" : "+v"(buf[k]));
        if constexpr (GEOM == 4) asm volatile("; pin %0" : "+v"(hb[k]));
        v4f q = buf[k];
        if constexpr (WORK >= 1) {
          *reinterpret_cast<v4f*>(xp + 4 + 4u * lane) = buf[k];
          if constexpr (GEOM == 4) {
            if (lane == 0) *reinterpret_cast<v2f*>(xp + 2) = hb[k];
            if (lane == 63) *reinterpret_cast<v2f*>(xp + 260) = hb[k];
          }
        }
        const int tn = t + P < T ? t + P : T - 1;
        issue(buf[k], hb[k], tn);
        if constexpr (WORK >= 1) {
          __builtin_amdgcn_wave_barrier();
          v2f pair[7];
#pragma unroll
          for (int m = 0; m < 7; m++) pair[m] = v2f{xp[2 + 4u * lane + m], xp[3 + 4u * lane + m]};
          if constexpr (WORK >= 2) {
#pragma unroll
            for (int i = 4; i >= 0; i--)
#pragma unroll
              for (int j = 0; j < 5; j++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                  const v2f w2 = v2f{w0 + (float)(i * 5 + j), w0 + (float)(i * 5 + j)};
                  acc[i][h] = __builtin_elementwise_fma(w2, pair[j + 2 * h], (j == 0 && i > 0) ? acc[i - 1][h] : acc[i][h]);
                }
            q = v4f{acc[4][0].x, acc[4][0].y, acc[4][1].x, acc[4][1].y};
          } else {
            q = v4f{pair[2].x, pair[2].y, pair[4].x, pair[4].y};
          }
          __builtin_amdgcn_wave_barrier();
        }
        const int o = t - lag;
        if (o >= 0) {
          if (writer) gstore4<NT>(q, voff, dp + (long)o * W);
        }
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// vector-memory issue cost by EXEC mask: 16 dword loads per iteration from an L1-resident window,
// MODE 0: all 64 lanes, 1: lanes 0..3, 2: EXEC = 0, 3: no loads (loop overhead)
template <int MODE>
__global__ void __launch_bounds__(256) exec_probe(const float* base, int iters, float* sink) {
  const unsigned lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* w = base + wave * 4096;
  const unsigned long long m = MODE == 0 ? ~0ull : (MODE == 1 ? 0xfull : 0ull);
  float acc = 0.f;
  const unsigned voff = 4u * lane;
  for (int it = 0; it < iters; it++) {
    float r[16];
    unsigned long long sv;
    if constexpr (MODE != 3) {
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(sv) : "s"(m));
#pragma unroll
      for (int k = 0; k < 16; k++)
        asm volatile("global_load_dword %0, %1, %2" : "=v"(r[k]) : "v"(voff), "s"(w + 256 * (k & 7) + (it & 3)));
      asm volatile("s_mov_b64 exec, %0\n\ts_waitcnt vmcnt(0)" ::"s"(sv));
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) acc += r[k];
      }
    } else {
      asm volatile("s_nop 0");
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

// ------------------------------------------------------------------ host --
template <typename F> double timeit(F f, int n = 6) {
  f(); f(); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < n; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3 / n;
}
static double g_bytes;
static void report(const char* name, double us, double bytes = -1) {
  if (bytes < 0) bytes = g_bytes;
  printf("%-64s %8.1f us  %6.0f GB/s\n", name, us, bytes / us / 1e3);
  fflush(stdout);
}

template <int D, int MODE> void run_chunk(const float* a, float* d, int sh, int frames) {
  const int strips_y = H / sh;
  unsigned blocks = MODE >= 3 ? strips_y * frames : (15u * strips_y * frames + 3) / 4;
  if (MODE == 2) blocks = ((15u * strips_y + 3) / 4) * frames;
  char nm[96]; snprintf(nm, sizeof nm, "strip chunks (compiler waits) D=%d order=%d sh=%d", D, MODE, sh);
  report(nm, timeit([&] { hipLaunchKernelGGL((strip_chunk<D, MODE>), dim3(blocks), dim3(MODE >= 3 ? 960 : 256), 0, 0, a, d, sh, strips_y, frames); }));
}
template <int P, int MODE, bool NT> void run_pipe(const float* a, float* d, int sh, int frames) {
  const int strips_y = H / sh;
  unsigned blocks = MODE >= 3 ? strips_y * frames : (15u * strips_y * frames + 3) / 4;
  if (MODE == 2) blocks = ((15u * strips_y + 3) / 4) * frames;
  char nm[96]; snprintf(nm, sizeof nm, "strip pipeline (asm, counted waits) P=%d order=%d sh=%d nt=%d", P, MODE, sh, (int)NT);
  report(nm, timeit([&] { hipLaunchKernelGGL((strip_pipe<P, MODE, NT>), dim3(blocks), dim3(MODE >= 3 ? 960 : 256), 0, 0, a, d, sh, strips_y, frames); }));
}

__global__ void check_copy(const unsigned* a, const unsigned* d, long n, unsigned* bad) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n && a[i] != d[i]) atomicAdd(bad, 1u);
}

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 64;
  const bool pmc = argc > 2;   // any second argument: counter mode (one launch per shape)
  const long npx = (long)W * H * frames;
  const size_t bytes = npx * 4;
  g_bytes = 2.0 * bytes;
  float *a, *d; unsigned* bad;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&bad, 64));
  hipLaunchKernelGGL(fill_kernel, dim3((npx + 255) / 256), dim3(256), 0, 0, (unsigned*)a, npx, 0);
  CK(hipMemset(d, 0, bytes));
  CK(hipDeviceSynchronize());

  if (pmc) {
    hipLaunchKernelGGL(fill_kernel, dim3((npx + 255) / 256), dim3(256), 0, 0, (unsigned*)a, npx, 1);
    const long n4 = npx / 4;
    const int sh = 72, strips_y = H / sh;
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(linear_copy, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const v4f*)a, (v4f*)d, n4);
      hipLaunchKernelGGL((conv_like<4, 2, 0, 0>), dim3(15u * strips_y * (frames / 4)), dim3(256), 0, 0, a, d, sh, strips_y, frames, 0.01f);
      hipLaunchKernelGGL((conv_like<4, 2, 1, 0>), dim3(16u * strips_y * (frames / 4)), dim3(256), 0, 0, a, d, sh, strips_y, frames, 0.01f);
      hipLaunchKernelGGL((conv_like<4, 2, 4, 0>), dim3(15u * strips_y * (frames / 4)), dim3(256), 0, 0, a, d, sh, strips_y, frames, 0.01f);
      hipLaunchKernelGGL((conv_like<4, 2, 4, 1>), dim3(15u * strips_y * (frames / 4)), dim3(256), 0, 0, a, d, sh, strips_y, frames, 0.01f);
    }
    CK(hipDeviceSynchronize());
    return 0;
  }
  // ---- probes (table = a, words hash32(i)); 2^29 words = 2 GB
  {
    unsigned mask = (1u << 28) - 1;  // 1 GB window
    while (((long)mask + 1) * 4 > (long)bytes) mask >>= 1;
    const char* names[3] = {"order  (load, store, vmcnt(1))", "exec0  (load, 4 loads with EXEC=0, vmcnt(4))", "ref    (store, load, vmcnt(0))"};
    for (int m = 0; m < 3; m++) {
      CK(hipMemset(bad, 0, 4));
      const int blocks = 256 * 8, iters = 400;
      if (m == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(blocks), dim3(256), 0, 0, (const unsigned*)a, mask, (unsigned*)d, bad, iters);
      if (m == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(blocks), dim3(256), 0, 0, (const unsigned*)a, mask, (unsigned*)d, bad, iters);
      if (m == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(blocks), dim3(256), 0, 0, (const unsigned*)a, mask, (unsigned*)d, bad, iters);
      CK(hipDeviceSynchronize());
      unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      printf("probe %-50s wrong values: %u of %ld\n", names[m], hb, (long)blocks * 256 * iters);
    }
  }
  {
    int clk_khz; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    const int blocks = 256 * 4, iters = 4000;   // 16 waves per CU
    const char* nm[4] = {"all 64 lanes", "lanes 0..3", "EXEC = 0", "no loads"};
    for (int mode = 0; mode < 4; mode++) {
      auto launch = [&](int n) {
        if (mode == 0) hipLaunchKernelGGL(exec_probe<0>, dim3(blocks), dim3(256), 0, 0, a, n, (float*)bad + 8);
        if (mode == 1) hipLaunchKernelGGL(exec_probe<1>, dim3(blocks), dim3(256), 0, 0, a, n, (float*)bad + 8);
        if (mode == 2) hipLaunchKernelGGL(exec_probe<2>, dim3(blocks), dim3(256), 0, 0, a, n, (float*)bad + 8);
        if (mode == 3) hipLaunchKernelGGL(exec_probe<3>, dim3(blocks), dim3(256), 0, 0, a, n, (float*)bad + 8);
      };
      launch(10); CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_cu = (double)blocks * 4 * iters * 16 / 256;
      printf("dword load issue cost, %-14s %8.3f ms  %6.2f clk per wave-instruction per CU (at %d MHz)\n",
             nm[mode], ms, ms * 1e-3 * clk_khz * 1e3 / instr_per_cu, clk_khz / 1000);
    }
  }
  // float data for the streams
  hipLaunchKernelGGL(fill_kernel, dim3((npx + 255) / 256), dim3(256), 0, 0, (unsigned*)a, npx, 1);
  CK(hipDeviceSynchronize());
  const long n4 = npx / 4;
  // clock settling
  for (int i = 0; i < 100; i++) hipLaunchKernelGGL(linear_copy, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const v4f*)a, (v4f*)d, n4);
  CK(hipDeviceSynchronize());
  printf("== %d frames of 4K float32: %.2f GB in, %.2f GB out ==\n", frames, bytes / 1e9, bytes / 1e9);
  report("hipMemcpyAsync d2d", timeit([&] { CK(hipMemcpyAsync(d, a, bytes, hipMemcpyDeviceToDevice, 0)); }));
  report("linear float4 copy", timeit([&] { hipLaunchKernelGGL(linear_copy, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const v4f*)a, (v4f*)d, n4); }));
  report("linear float4 copy, nt stores", timeit([&] { hipLaunchKernelGGL(linear_copy_nt, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const v4f*)a, (v4f*)d, n4); }));
  report("linear float4 read only", timeit([&] { hipLaunchKernelGGL(linear_read, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const v4f*)a, (float*)bad + 8, n4); }), (double)bytes);
  report("linear float4 write only", timeit([&] { hipLaunchKernelGGL(linear_write, dim3((n4 + 255) / 256), dim3(256), 0, 0, (v4f*)d, n4); }), (double)bytes);

  const int sh = 72;
  run_chunk<8, 0>(a, d, sh, frames);
  run_chunk<8, 1>(a, d, sh, frames);
  run_chunk<8, 2>(a, d, sh, frames);
  run_chunk<8, 3>(a, d, sh, frames);
  run_chunk<8, 4>(a, d, sh, frames);
  run_chunk<4, 0>(a, d, sh, frames);

  run_pipe<2, 0, true>(a, d, sh, frames);
  run_pipe<4, 0, true>(a, d, sh, frames);
  run_pipe<8, 0, true>(a, d, sh, frames);
  run_pipe<12, 0, true>(a, d, sh, frames);
  run_pipe<8, 0, false>(a, d, sh, frames);
  run_pipe<8, 1, true>(a, d, sh, frames);
  run_pipe<8, 2, true>(a, d, sh, frames);
  run_pipe<4, 3, true>(a, d, sh, frames);
  run_pipe<8, 3, true>(a, d, sh, frames);
  run_pipe<8, 4, true>(a, d, sh, frames);
  run_pipe<4, 4, true>(a, d, sh, frames);
  {
    const int strips_y = H / sh;
    auto run = [&](const char* nm, auto kern, int sx) {
      const unsigned blocks = (unsigned)sx * strips_y * (frames / 4);
      report(nm, timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, a, d, sh, strips_y, frames, 0.01f); }));
    };
    for (int rep = 0; rep < 2; rep++) {
      run("G0 aligned 256 strips, 72 rows, copy, order 0", conv_like<4, 0, 0, 0>, 15);
      run("G0 aligned 256 strips, 72 rows, 5x5 work, order 0", conv_like<4, 2, 0, 0>, 15);
      run("G1 248-step, 76 rows (today), copy, order 0", conv_like<4, 0, 1, 0>, 16);
      run("G1 248-step, 76 rows (today), 5x5 work, order 0", conv_like<4, 2, 1, 0>, 16);
      run("G2 248-step, 72 rows, copy, order 0", conv_like<4, 0, 2, 0>, 16);
      run("G4 aligned + halo pair load, 76 rows, copy, order 0", conv_like<4, 0, 4, 0>, 15);
      run("G4 aligned + halo pair load, 76 rows, 5x5 work, order 0", conv_like<4, 2, 4, 0>, 15);
      run("G5 248-step, loads aligned down, copy, order 0", conv_like<4, 0, 5, 0>, 16);
      run("G6 248-step, stores aligned down, copy, order 0", conv_like<4, 0, 6, 0>, 16);
      run("G0 order 1 (strips of a frame, frame after frame), copy", conv_like<4, 0, 0, 1>, 15);
      run("G1 order 1, 5x5 work", conv_like<4, 2, 1, 1>, 16);
      run("G4 order 1, 5x5 work", conv_like<4, 2, 4, 1>, 15);
      run("G1 order 0, 5x5 work, plain (not nt) stores", conv_like<4, 2, 1, 0, false>, 16);
      run("G1 order 1, 5x5 work, plain (not nt) stores", conv_like<4, 2, 1, 1, false>, 16);
      run("G4 order 1, 5x5 work, plain (not nt) stores", conv_like<4, 2, 4, 1, false>, 15);
      run("G0 order 2 (groups of 8 frames), copy", conv_like<4, 0, 0, 2>, 15);
      run("G1 order 2, 5x5 work", conv_like<4, 2, 1, 2>, 16);
      run("G4 order 2, 5x5 work", conv_like<4, 2, 4, 2>, 15);
    }
  }
  // the pipelined copy really copies
  CK(hipMemset(d, 0, bytes));
  {
    const int strips_y = H / sh;
    hipLaunchKernelGGL((strip_pipe<8, 0, true>), dim3((15u * strips_y * frames + 3) / 4), dim3(256), 0, 0, a, d, sh, strips_y, frames);
    CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(check_copy, dim3((npx + 255) / 256), dim3(256), 0, 0, (const unsigned*)a, (const unsigned*)d, npx, bad);
    unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("pipelined copy P=8 order 0: %u wrong words of %ld\n", hb, npx);
    CK(hipMemset(d, 0, bytes));
    hipLaunchKernelGGL((strip_pipe<8, 3, true>), dim3(strips_y * frames), dim3(960), 0, 0, a, d, sh, strips_y, frames);
    CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(check_copy, dim3((npx + 255) / 256), dim3(256), 0, 0, (const unsigned*)a, (const unsigned*)d, npx, bad);
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("pipelined copy P=8 order 3: %u wrong words of %ld\n", hb, npx);
  }
  return 0;
}
