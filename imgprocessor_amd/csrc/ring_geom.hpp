// ring_geom.hpp — geometry, wave reductions and the register-staged source row of the LDS-ring
// remap kernels (ring_stencil.hpp: planning, ring_remap.hpp: the remap loop).
//
// (Round 2 also built a frame-group kernel and fused ring / pair kernels on these pieces; they
// measured slower than the gather loop - DESIGN.md section 5 - and left the tree in round 4.)
#pragma once

#include <limits.h>

#include "wave_stencil.hpp"

namespace ipa {

constexpr int kSW = 128;                      // strip width (2 px per lane in the filter stage)
constexpr int kRR = 8;                        // ring rows (power of two)
constexpr int kRW = 160;                      // ring row length (pixels)
constexpr int kRingFloats = (kRR + 1) * kRW;  // slot kRR mirrors slot 0
constexpr int kXRow = kSW + 2 * kRowPad;      // staged sample row (+ pad on both sides)

template <int K> struct group_geom {
  static constexpr int H = K / 2;
  static constexpr int HL = (H + 1) / 2;      // halo lanes per side (2 px per lane)
  static constexpr int OW = kSW - 4 * HL;     // output pixels per strip row
};

// min / max over the 64 lanes (all active), result wave-uniform
template <bool MAX> __device__ __forceinline__ int wave_minmax(int v) {
#define IPA_MM(a, b) (MAX ? ((a) > (b) ? (a) : (b)) : ((a) < (b) ? (a) : (b)))
  int t;
  t = __builtin_amdgcn_update_dpp(v, v, 0x111 /*row_shr:1*/, 0xf, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x112 /*row_shr:2*/, 0xf, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x114 /*row_shr:4*/, 0xf, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x118 /*row_shr:8*/, 0xf, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false); v = IPA_MM(v, t);
#undef IPA_MM
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void wave_span(int& xmn, int& xmx, int& ymn, int& ymx) {
  xmn = wave_minmax<false>(xmn);
  xmx = wave_minmax<true>(xmx);
  ymn = wave_minmax<false>(ymn);
  ymx = wave_minmax<true>(ymx);
}

// LDS-only workgroup barrier: __syncthreads() would also drain the vector-memory counter,
// i.e. wait for the prefetched source rows and the output stores at every chunk
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// one source row of the ring window (kRW px from element offset eo) in registers:
// 2 px per lane (128 px) + 1 px (32 px; lanes 32..63 duplicate the lanes 0..31)
template <typename ST> struct PendRow;
template <> struct PendRow<float> {
  float a[2];
  float b;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int eo, unsigned lane) {
    // range-checked per dword: columns left / right of the frame read neighbouring rows or 0,
    // never used (only footprints wholly inside the frame sample from the ring)
    auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (eo + 2 * (int)lane) << 2, 0, 0);
    a[0] = u2f(r[0]); a[1] = u2f(r[1]);
    b = u2f(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (eo + 128 + (int)(lane & 31u)) << 2, 0, 0));
  }
  __device__ __forceinline__ void write(float* row, unsigned lane) const {
    *reinterpret_cast<float2*>(row + 2u * lane) = float2{a[0], a[1]};
    row[128u + (lane & 31u)] = b;  // both halves of the wave store the same value
  }
  // column c at row[2 c]: one row of a row-pair-interleaved ring (ring_remap.hpp, Lanczos4)
  __device__ __forceinline__ void write_every_other(float* row, unsigned lane) const {
    row[4u * lane] = a[0];
    row[4u * lane + 2u] = a[1];
    row[2u * (128u + (lane & 31u))] = b;
  }
};
template <> struct PendRow<uint16_t> {
  unsigned a;
  unsigned b;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int eo, unsigned lane) {
    a = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (eo + 2 * (int)lane) << 1, 0, 0);
    b = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
        rsrc, (eo + 128 + (int)(lane & 31u)) << 1, 0, 0);
  }
  __device__ __forceinline__ void write(float* row, unsigned lane) const {
    *reinterpret_cast<float2*>(row + 2u * lane) = float2{(float)(a & 0xffffu), (float)(a >> 16)};
    row[128u + (lane & 31u)] = (float)b;
  }
};

}  // namespace ipa
