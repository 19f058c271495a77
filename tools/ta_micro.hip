// ta_micro.hip — vector-memory issue-rate probe for gfx950 (NOT part of the product).
// Measures clocks per wave64 memory instruction per CU for the access patterns the remap
// gathers can be built from, with the data L1/L2-resident, so the address-processing rate
// of the texture-addresser path is visible apart from HBM bandwidth.
//   hipcc --offload-arch=gfx950 -O3 tools/ta_micro.hip -o gpurun_out/ta_micro && ./ta_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ base, int window_floats,
                                             int iters, int row_pitch, float* sink) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* w = base + wave * (long)window_floats;   // wave-private window
  __shared__ float lds[4][2048];
  float* l = lds[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
  if (MODE == 7 || MODE == 10)
    for (int i = lane; i < 2048; i += 64) l[i] = w[i];
  __builtin_amdgcn_s_waitcnt(0);
  float acc = 0.f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, window_floats * 4, 0x00020000);
  for (int it = 0; it < iters; it++) {
    // 8 independent loads per iteration, rows r = 0..7 of the window
    const int shift = (it & 3);   // runtime-dependent misalignment
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const float* row = w + r * row_pitch;
      if (MODE == 0) acc += row[lane + shift];
      else if (MODE == 1) { v2f v = *(const v2f*)(row + 2 * lane + 2 * shift); acc += v.x + v.y; }
      else if (MODE == 2) { v4f v = *(const v4f*)(row + 4 * lane + 4 * shift); acc += v.x + v.w; }
      else if (MODE == 3) { v2u v = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((row - w) + 4 * lane + 1 + shift) * 4, 0, 0)); acc += __uint_as_float(v.x) + __uint_as_float(v.y); }
      else if (MODE == 4) { v2u v = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((row - w) + lane + shift) * 4, 0, 0)); acc += __uint_as_float(v.x) + __uint_as_float(v.y); }
      else if (MODE == 5) acc += row[4 * lane + 1 + shift];
      else if (MODE == 6) acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((row - w) + lane + shift) * 4, 0, 0));
      else if (MODE == 7) { int o = (r * 256 + lane + it) & 2046; acc += l[o] + l[o + 1]; }
      else if (MODE == 8) { // dwordx2 step 4B, 4 different rows inside the wave (curved source row)
        v2u v = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((row - w) + lane + shift + (lane >> 4) * row_pitch) * 4, 0, 0)); acc += __uint_as_float(v.x) + __uint_as_float(v.y); }
      else if (MODE == 9) { // global (flat) dwordx2 step 4 B — needs 8-byte alignment? no: dword aligned ok
        v2f v; __builtin_memcpy(&v, row + lane + shift, 8); acc += v.x + v.y; }
      else if (MODE == 10) { v4f v = *(const v4f*)(l + ((r * 256 + 4 * lane + 4 * it) & 2044)); acc += v.x + v.w; }
      else if (MODE == 11) { // dwordx4 step 4 B overlapping (4 taps of a cubic row)
        v4i v = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((row - w) + lane + shift) * 4, 0, 0)); acc += __int_as_float(v.x) + __int_as_float(v.w); }
      else if (MODE == 12) { // dwordx2 stride 8 B aligned, coalesced via buffer
        v2u v = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((row - w) + 2 * lane + 2 * shift) * 4, 0, 0)); acc += __uint_as_float(v.x) + __uint_as_float(v.y); }
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE> void run(const char* name, const float* d, long window, int row_pitch, float* sink, int cus) {
  const int blocks = cus * 8, iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, (int)window, 10, row_pitch, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, (int)window, iters, row_pitch, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double instr_per_cu = (double)blocks * 4 * iters * 8 / cus;
  int clk_khz; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  double clks = ms * 1e-3 * clk_khz * 1e3;
  printf("%-46s window %6ld fl pitch %5d: %8.3f ms  %6.2f clk/wave-instr/CU (clock %d MHz)\n", name, window, row_pitch, ms, clks / instr_per_cu, clk_khz / 1000);
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  int cus = pr.multiProcessorCount;
  printf("%s, %d CUs\n", pr.name, cus);
  const long waves = (long)cus * 8 * 4;
  for (int pass = 0; pass < 2; pass++) {
    // pass 0: 8 rows inside a 2.6K-float window (L1 resident); pass 1: rows 4096 floats apart (L2)
    long window = pass == 0 ? 8 * 256 + 1024 : 8 * 4096 + 1024;
    int pitch = pass == 0 ? 256 : 4096;
    float* d; CK(hipMalloc(&d, waves * window * 4 + 4096));
    CK(hipMemset(d, 0, waves * window * 4 + 4096));
    float* sink; CK(hipMalloc(&sink, 64));
    printf("--- %s\n", pass == 0 ? "L1-resident windows" : "L2-resident windows (1 MB per 8 waves... per wave 132 KB)");
    run<0>("dword   coalesced (global)", d, window, pitch, sink, cus);
    run<1>("dwordx2 coalesced (global)", d, window, pitch, sink, cus);
    run<12>("dwordx2 coalesced (buffer)", d, window, pitch, sink, cus);
    run<2>("dwordx4 coalesced (global)", d, window, pitch, sink, cus);
    run<5>("dword   stride 16 B", d, window, pitch, sink, cus);
    run<3>("dwordx2 stride 16 B unaligned (old gather)", d, window, pitch, sink, cus);
    run<4>("dwordx2 step 4 B overlapping (interleaved)", d, window, pitch, sink, cus);
    run<9>("dwordx2 step 4 B overlapping (global)", d, window, pitch, sink, cus);
    run<8>("dwordx2 step 4 B, 4 source rows per wave", d, window, pitch, sink, cus);
    run<6>("dword   step 4 B (buffer)", d, window, pitch, sink, cus);
    run<11>("dwordx4 step 4 B overlapping (buffer)", d, window, pitch, sink, cus);
    run<7>("LDS 2 x ds_read_b32 step 4 B", d, window, pitch, sink, cus);
    run<10>("LDS ds_read_b128 stride 16 B", d, window, pitch, sink, cus);
    CK(hipFree(d)); CK(hipFree(sink));
  }
  return 0;
}
