// fused_sep_impl.hpp — remap -> separable K+K filter in ONE kernel (BASELINE config C3:
// PerspectiveCorrection remap + separable 9-tap Gaussian; the reference obtains such chains as
// cv2.warpPerspective / cv2.remap followed by scipy.ndimage.gaussian_filter).
//
// wave_sep_kernel with a sampling row source: the remapped image never exists in HBM
// (8 B/px analytic, 16 B/px map-based instead of 16 / 24 for two launches).  Built for
// float32 -> float32 (and uint16 -> float32 with maps or a homography), K = 3, 5, 7, 9 taps on both axes; fused.hip routes bilinear here and
// every other combination (bicubic included: measured slower fused) through two launches.
#pragma once

#include "fused_impl.hpp"
#include "wave_sep.hpp"

namespace ipa {

struct FusedSep {
  const double* ky;
  const double* kx;
  int n;        // taps per axis
  float xcval;  // constant x border of the intermediate (scipy pads the intermediate)
};

template <typename ST, int INTERP, typename Coord, int K>
static void fused_sep_one(ipa_ctx* ctx, const FusedCall& f, const Coord& c, const FusedSep& q) {
  using Src = SampleRowSrc<ST, INTERP, Coord>;
  if constexpr (std::is_same<Coord, HomographyCoord>::value && sep_shared<Src, K>::value) {
    if (fused_wants_stored_coords(ctx, f)) {   // (see fused_impl.hpp)
      StoredCoord<double> sc;
      if (stored_coords_prepare<Coord>(ctx, c, f.p.dh, f.p.dw, &sc) == 0) {
        fused_sep_one<ST, INTERP, StoredCoord<double>, K>(ctx, f, sc, q);
        return;
      }
    }
  }
  if (sep_shared<Src, K>::value && fused_split_tail<Src, 5>(ctx, f)) {   // (see fused_impl.hpp)
    FusedCall head = f, tail = f;
    head.n_frames = f.n_frames - f.n_frames % IPA_WPB;
    tail.n_frames = IPA_WPB;
    tail.src = f.src + (long)(f.n_frames - IPA_WPB) * f.src_frame_bytes;
    tail.p.dst = f.p.dst + (long)(f.n_frames - IPA_WPB) * f.p.dst_frame_elems * 4;
    fused_sep_one<ST, INTERP, Coord, K>(ctx, head, c, q);
    fused_sep_one<ST, INTERP, Coord, K>(ctx, tail, c, q);
    return;
  }
  Src s;
  s.coord = c;
  s.src = f.src; s.src_frame_bytes = f.src_frame_bytes; s.src_bytes = f.src_bytes;
  s.sh = f.sh; s.sw = f.sw; s.spitch = f.spitch;
  s.border = f.border; s.q5 = f.q5; s.cubic_a = f.cubic_a; s.lanczos = nullptr;
  s.cval = (float)f.cval; s.ccval = (float)f.conv_cval; s.map_vec = f.map_vec;
  launch_sep<Src, K>(ctx, f.p, s, q.ky, q.kx, f.n_frames, q.xcval);
}

template <typename Coord, int K>
static void fused_sep_interp(ipa_ctx* ctx, const FusedCall& f, const Coord& c, const FusedSep& q) {
  // uint16 frames (camera frames as toFloatArray ingests them): maps and homographies - fused.hip sends the
  // lens model by value through its cached map or two launches (round 6: the chain was two launches for every
  // uint16 batch)
  if constexpr (std::is_same<Coord, MapCoord>::value || std::is_same<Coord, HomographyCoord>::value) {
    if (f.src_dt == IPA_U16) {
      fused_sep_one<uint16_t, kLinear, Coord, K>(ctx, f, c, q);
      return;
    }
  }
  // uint8 frames into float32 (8-bit cameras through toFloatArray), with maps: the remap alone and + separable filter
  if constexpr (std::is_same<Coord, MapCoord>::value) {
    if (f.src_dt == IPA_U8) {
      fused_sep_one<uint8_t, kLinear, Coord, K>(ctx, f, c, q);
      return;
    }
  }
  // K = 1 (the remap alone, fused_sep_c.hip) is built for uint16 frames only: float32 frames have the tile kernel
  // (level on maps, 15 - 19 % faster under a homography), and fused.hip does not send them here
  if constexpr (K != 1) fused_sep_one<float, kLinear, Coord, K>(ctx, f, c, q);
}

template <int K> static void fused_sep_k(ipa_ctx* ctx, const FusedCall& f, const FusedSep& q) {
  switch (f.coord_kind) {
    case 0: fused_sep_interp<MapCoord, K>(ctx, f, f.map, q); break;
    case 1: fused_sep_interp<UndistortCoord, K>(ctx, f, f.und, q); break;
    default: fused_sep_interp<HomographyCoord, K>(ctx, f, f.hom, q); break;
  }
}

}  // namespace ipa
