// conv_tile.hpp — LDS-tiled dense KH x KW correlation core shared by the plain
// filter (conv.hip) and the fused remap->filter kernels (fused_impl.hpp).
//
// Geometry (one 256-thread workgroup = 4 wave64):
//   output tile 128 x 32 px; thread (tx 0..31, ty 0..7) owns a 4 x 4 micro-tile.
//   LDS tile: (32 + KH - 1) rows x (128 + 2*HX) columns, HX = KW/2 rounded up
//   to 4 so that every global load / LDS access is a 16-byte vector.
//   A thread slides over its 4 + KH - 1 input rows once; each row is read from
//   LDS as aligned ds_read_b128 chunks into registers (next row prefetched
//   while the current one is consumed) and feeds up to 4 output rows:
//   register-level reuse 4x vertically, KW-fold horizontally.
//   Weights stay in the kernarg segment and are fetched with scalar loads per
//   (input row, output row) pair -> v_fma with an SGPR operand, no VGPRs spent
//   on the KH*KW coefficients.  No MFMA: 2*K*K flop/px against 8 B/px is
//   memory-bound.
//
// Summation order per output: rows i = 0..KH-1, within a row j = 0..KW-1, one
// fma chain in the compute type (float for f32 images, double for f64).
#pragma once

#include "common.hpp"

namespace ipa {

constexpr int kTileW = 128;
constexpr int kTileH = 32;

template <typename CT, int N> struct Weights {
  CT w[N];
};

template <int KW> struct conv_geom {
  static constexpr int HX = ((KW / 2 + 3) / 4) * 4;   // aligned horizontal halo
  static constexpr int LW = kTileW + 2 * HX;          // LDS row length (elements)
  static constexpr int NW = 4 + 2 * HX;               // window elements a thread reads per row
  static constexpr int OFF = HX - KW / 2;             // window index of tap j=0 for output ox=0
};

template <typename CT> struct vec16;
template <> struct vec16<float> {
  using type = float4;
  static constexpr int n = 4;
};
template <> struct vec16<double> {
  using type = double2;
  static constexpr int n = 2;
};

template <typename CT, int NW>
__device__ __forceinline__ void load_window(const CT* __restrict__ p, CT (&win)[NW]) {
  using V = typename vec16<CT>::type;
  constexpr int VN = vec16<CT>::n;
  const V* src = reinterpret_cast<const V*>(p);
#pragma unroll
  for (int c = 0; c < NW / VN; c++) {
    V v = src[c];
    if constexpr (VN == 4) {
      win[c * 4 + 0] = v.x; win[c * 4 + 1] = v.y; win[c * 4 + 2] = v.z; win[c * 4 + 3] = v.w;
    } else {
      win[c * 2 + 0] = v.x; win[c * 2 + 1] = v.y;
    }
  }
}

// one input row (window `win`, tile-relative row r) feeds the up-to-4 output
// rows oy with kernel row i = r - oy; r is wave-uniform -> scalar weight loads
template <typename CT, int KH, int KW>
__device__ __forceinline__ void conv_row(const CT (&win)[conv_geom<KW>::NW], int r,
                                         const Weights<CT, KH * KW>& wts, CT (&acc)[4][4]) {
  using G = conv_geom<KW>;
#pragma unroll
  for (int oy = 0; oy < 4; oy++) {
    const int i = r - oy;
    if (i >= 0 && i < KH) {
#pragma unroll
      for (int j = 0; j < KW; j++) {
        CT w = wts.w[i * KW + j];
#pragma unroll
        for (int ox = 0; ox < 4; ox++)
          acc[oy][ox] = ipa_fma(w, win[G::OFF + ox + j], acc[oy][ox]);
      }
    }
  }
}

// Correlate the LDS tile; acc[oy][ox] for the thread's 4x4 micro-tile.
template <typename CT, int KH, int KW>
__device__ __forceinline__ void conv_from_lds(const CT* __restrict__ tile, int tx, int ty,
                                              const Weights<CT, KH * KW>& wts, CT (&acc)[4][4]) {
  using G = conv_geom<KW>;
#pragma unroll
  for (int oy = 0; oy < 4; oy++)
#pragma unroll
    for (int ox = 0; ox < 4; ox++) acc[oy][ox] = (CT)0;

  static_assert((4 + KH - 1) % 2 == 0, "odd KH only: the row loop is unrolled by two");
  const CT* base = tile + (ty * 4) * G::LW + tx * 4;
  CT wa[G::NW], wb[G::NW];  // ping-pong row windows: no register copies between rows
  load_window<CT, G::NW>(base, wa);
#pragma unroll 1
  for (int r = 0; r < 4 + KH - 1; r += 2) {
    // prefetch the next input row while the current one is consumed (the tile
    // has one spare row of slack for the last prefetch: see lds_rows)
    load_window<CT, G::NW>(base + (r + 1) * G::LW, wb);
    conv_row<CT, KH, KW>(wa, r, wts, acc);
    load_window<CT, G::NW>(base + (r + 2) * G::LW, wa);
    conv_row<CT, KH, KW>(wb, r + 1, wts, acc);
  }
}

// rows to allocate for the LDS tile: one spare row so the prefetch of row
// (4 + KH - 1) by the last micro-row stays inside the allocation
template <int KH> constexpr int lds_rows() { return kTileH + KH - 1 + 1; }

}  // namespace ipa
