// fused_impl.hpp — remap -> K x K filter in ONE kernel (the headline chain:
// LensDistortion.correct / PerspectiveCorrection.correct followed by a dense
// filter; in-tree archetype camera/lens/estimateSystematicErrorLensCorrection.py:199-207).
//
// The kernel is the wave-marching stencil of wave_stencil.hpp with a row
// source that samples the remapped image on the fly (positions outside the
// remapped image are resolved with the FILTER's border mode first, i.e.
// exactly what filtering the materialised remap result would see).  The
// intermediate image never exists in HBM: 16 B/px map-based, 8 B/px analytic,
// instead of 24 / 16 for two launches.
//
// This header is compiled once per K (fused_k*.hip define IPA_FUSED_K) so the
// translation units build in parallel.
#pragma once

#include "common.hpp"
#include "wave_stencil.hpp"
#include "stored_coords.hpp"

namespace ipa {

// host-side description of one fused call (coordinate source by kind)
struct FusedCall {
  WaveParams p;
  // remap source
  const char* src;
  long src_frame_bytes;
  unsigned src_bytes;
  int sh, sw, spitch;
  int border, q5;
  float cubic_a;
  double cval, conv_cval;
  int coord_kind;  // 0 map, 1 undistort, 2 homography
  int map_vec;
  MapCoord map;
  UndistortCoord und;
  HomographyCoord hom;
  int src_dt, dst_dt, interp_base, n_frames;
  const double* kernel;
};

// Batches whose frame count is no multiple of the workgroup's IPA_WPB frames cannot share
// footprint records in every workgroup and used to fall back to the per-frame loop as a whole
// (15 x 4K frames: 0.398 ms against 0.268 for 16).  They run as TWO launches of the shared loop
// instead: the first n - n % IPA_WPB frames, then the LAST IPA_WPB frames - up to three of those a
// second time, with the same bits.  (Not when source and result may overlap: the second launch
// would then read what the first wrote.)
// which strip-height table a fused launch takes (wave_strip_height's `piped`): the tall strips of
// the shared-record loop only where that loop runs - the frames of a workgroup are frames of one
// strip (knobs frames_wg / frames_inner, n a multiple of IPA_WPB); the per-frame loop, whose rim
// strips are on the chunked path, keeps the short ones (64 x 4K with frames_wg = 0: 1.53 ms on
// 144-row strips)
template <typename Src, int K> static int fused_strip_piped(const ipa_ctx* ctx, int n_frames) {
  const bool shared_run = shared_capable<Src, K>::value && IPA_PIPE && IPA_PIPE_SHARED &&
                          ctx->tune.frames_wg != 0 && ctx->tune.frames_inner != 0 &&
                          n_frames % IPA_WPB == 0;
  return shared_run ? 2 : 0;
}

template <typename Src, int K> static bool fused_split_tail(const ipa_ctx* ctx, const FusedCall& f) {
  if (!(shared_capable<Src, K>::value && IPA_PIPE && IPA_PIPE_SHARED)) return false;
  if (!ctx->tune.frames_wg || !ctx->tune.frames_inner) return false;
  // (from 7 frames on: 5 and 6 frames measure faster on the per-frame loop - 0.124 against 0.158 ms)
  if (f.n_frames < 2 * IPA_WPB - 1 || f.n_frames % IPA_WPB == 0) return false;
  const char* s0 = f.src;
  const char* s1 = f.src + (long)f.n_frames * f.src_frame_bytes;
  const char* d0 = f.p.dst;
  const char* d1 = f.p.dst + (long)f.n_frames * f.p.dst_frame_elems * 4;
  return s1 <= d0 || d1 <= s0;
}

// batches on the shared-record loop whose coordinates come from the homography: its double
// coordinates (two fused-multiply-add chains and a division per pixel) are evaluated ONCE per
// (matrix, geometry) into the plan buffer (stored_coords.hpp) and the record producers read them
// as a table - the C3 chain spent a third of its time evaluating them once per four frames
static inline bool fused_wants_stored_coords(const ipa_ctx* ctx, const FusedCall& f) {
  const int smin = ctx->tune.stored_coords;
  return smin > 0 && f.n_frames >= smin && ctx->tune.pipe != 0 && ctx->tune.frames_wg != 0 &&
         ctx->tune.frames_inner != 0 && (f.n_frames % IPA_WPB == 0 || f.n_frames >= 2 * IPA_WPB - 1);
}

template <typename ST, int INTERP, typename Coord, int K>
static void fused_launch_one(ipa_ctx* ctx, const FusedCall& f, const Coord& c) {
  using Src = SampleRowSrc<ST, INTERP, Coord>;
  if constexpr (std::is_same<Coord, HomographyCoord>::value && shared_capable<Src, K>::value) {
    if (fused_wants_stored_coords(ctx, f)) {
      StoredCoord<double> sc;
      if (stored_coords_prepare<Coord>(ctx, c, f.p.dh, f.p.dw, &sc) == 0) {
        fused_launch_one<ST, INTERP, StoredCoord<double>, K>(ctx, f, sc);
        return;
      }
    }
  }
  if (fused_split_tail<Src, K>(ctx, f)) {
    FusedCall head = f, tail = f;
    head.n_frames = f.n_frames - f.n_frames % IPA_WPB;
    tail.n_frames = IPA_WPB;
    tail.src = f.src + (long)(f.n_frames - IPA_WPB) * f.src_frame_bytes;
    tail.p.dst = f.p.dst + (long)(f.n_frames - IPA_WPB) * f.p.dst_frame_elems * 4;
    fused_launch_one<ST, INTERP, Coord, K>(ctx, head, c);
    fused_launch_one<ST, INTERP, Coord, K>(ctx, tail, c);
    return;
  }
  Weights<float, K * K> w;
  for (int i = 0; i < K * K; i++) w.w[i] = (float)f.kernel[i];
  Src s;
  s.coord = c;
  s.src = f.src; s.src_frame_bytes = f.src_frame_bytes; s.src_bytes = f.src_bytes;
  s.sh = f.sh; s.sw = f.sw; s.spitch = f.spitch;
  s.border = f.border; s.q5 = f.q5; s.cubic_a = f.cubic_a; s.lanczos = nullptr;
  s.cval = (float)f.cval; s.ccval = (float)f.conv_cval; s.map_vec = f.map_vec;
  WaveParams p = f.p;
  using G = wave_geom<K, geom_halo<Src, K, false>::value>;
  p.strips_x = (p.dw + G::OW - 1) / G::OW;
  p.strip_h = wave_strip_height(ctx, p.dh, p.dw, f.n_frames, K, false,
                                fused_strip_piped<Src, K>(ctx, f.n_frames), p.strips_x);
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  // frames of one strip block run together: map-based remaps share their map rows between
  // frames (L2 fetch traffic -62 % on 16 x 4K), and even without shared rows the order measured
  // ~5 % faster than frame-after-frame
  dim3 grid = wave_grid(ctx, p, f.n_frames, IPA_WPB, true,
                        coord_is_table<Coord>::value || shared_capable<Src, K>::value, false, K);
  dim3 block(64 * IPA_WPB);
  hipLaunchKernelGGL((wave_stencil_kernel<Src, K>), grid, block, 0, ctx->stream, p, s, w);
}

template <typename ST, typename Coord, int K>
static int fused_launch_interp(ipa_ctx* ctx, const FusedCall& f, const Coord& c) {
  switch (f.interp_base) {
    case IPA_INTER_LINEAR: fused_launch_one<ST, kLinear, Coord, K>(ctx, f, c); break;
    case IPA_INTER_CUBIC_CV:
    case IPA_INTER_CUBIC_KEYS: fused_launch_one<ST, kCubic, Coord, K>(ctx, f, c); break;
    default:
      IPA_UNSUPPORTED(ctx, "fused remap+filter supports INTER_LINEAR and the two bicubics");
  }
  return IPA_OK;
}

template <typename ST, int K> static int fused_launch_coord(ipa_ctx* ctx, const FusedCall& f) {
  switch (f.coord_kind) {
    case 0: return fused_launch_interp<ST, MapCoord, K>(ctx, f, f.map);
    case 1: return fused_launch_interp<ST, UndistortCoord, K>(ctx, f, f.und);
    default: return fused_launch_interp<ST, HomographyCoord, K>(ctx, f, f.hom);
  }
}

template <int K> static int fused_launch_k(ipa_ctx* ctx, const FusedCall& f) {
  if (f.src_dt == IPA_F32 && f.dst_dt == IPA_F32) return fused_launch_coord<float, K>(ctx, f);
  if (f.src_dt == IPA_U16 && f.dst_dt == IPA_F32) {
    // camera frames (toFloatArray ingest): bilinear undistort, map-based or analytic
    if (f.interp_base == IPA_INTER_LINEAR && f.coord_kind == 0) {
      fused_launch_one<uint16_t, kLinear, MapCoord, K>(ctx, f, f.map);
      return IPA_OK;
    }
    if (f.interp_base == IPA_INTER_LINEAR && f.coord_kind == 1) {
      fused_launch_one<uint16_t, kLinear, UndistortCoord, K>(ctx, f, f.und);
      return IPA_OK;
    }
    IPA_UNSUPPORTED(ctx, "fused remap+filter on uint16 frames is built for INTER_LINEAR with "
                         "maps or the analytic lens model; use ipa_remap_dev + ipa_conv2d_dev");
  }
  if (f.src_dt == IPA_U8 && f.dst_dt == IPA_F32 && f.interp_base == IPA_INTER_LINEAR && f.coord_kind == 0) {
    // 8-bit camera frames (round 6): bilinear, map-based
    fused_launch_one<uint8_t, kLinear, MapCoord, K>(ctx, f, f.map);
    return IPA_OK;
  }
  IPA_UNSUPPORTED(ctx, "fused remap+filter: src dtype %d -> dst dtype %d not supported "
                       "(float32->float32 and uint16->float32 are)", f.src_dt, f.dst_dt);
}

}  // namespace ipa

#ifdef IPA_FUSED_K
#define IPA_CAT2(a, b) a##b
#define IPA_CAT(a, b) IPA_CAT2(a, b)
int IPA_CAT(ipa_fused_launch_k, IPA_FUSED_K)(ipa_ctx* ctx, const ipa::FusedCall& f) {
  return ipa::fused_launch_k<IPA_FUSED_K>(ctx, f);
}
// the plain float32 filter on the same skeleton (rows straight from memory)
int IPA_CAT(ipa_wave_conv_launch_k, IPA_FUSED_K)(ipa_ctx* ctx, const ipa::WaveParams& p0,
                                                 const ipa::LoadRowSrc& src, const double* kernel,
                                                 int n_frames) {
  using namespace ipa;
  constexpr int K = IPA_FUSED_K;
  Weights<float, K * K> w;
  for (int i = 0; i < K * K; i++) w.w[i] = (float)kernel[i];
  WaveParams p = p0;
  using G = wave_geom<K, geom_halo<LoadRowSrc, K, false>::value>;
  p.strips_x = (p.dw + G::OW - 1) / G::OW;
  p.strip_h = wave_strip_height(ctx, p.dh, p.dw, n_frames, K, false,
                                pipe_capable<LoadRowSrc, K>::value && IPA_PIPE ? 1 : 0, p.strips_x);
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  dim3 grid = wave_grid(ctx, p, n_frames, IPA_WPB, true, false, true), block(64 * IPA_WPB);
  hipLaunchKernelGGL((wave_stencil_kernel<LoadRowSrc, K>), grid, block, 0, ctx->stream, p, src, w);
  return IPA_OK;
}
#endif
