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
  // coord_kind 0: the maps are the context's own lens maps (ipa_lens_map_cached), identified by
  // ctx->lens_serial - strip plans made for them can be reused; 0: the caller's maps, planned
  // anew on every call (their contents may have changed)
  unsigned long map_static = 0;
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


// ---- the LDS-ring kernel (wave_lring.hpp): clean strips take their source rows through LDS ----
static inline int lring_coord_key(const MapCoord& c, const FusedCall& f, double* k) {
  k[0] = (double)reinterpret_cast<uintptr_t>(c.mx);
  k[1] = (double)reinterpret_cast<uintptr_t>(c.my);
  k[2] = (double)c.pitch;
  k[3] = (double)f.map_static;
  return 4;
}
static inline int lring_coord_key(const UndistortCoord& c, const FusedCall&, double* k) {
  for (int i = 0; i < 9; i++) k[i] = c.ir[i];
  const double v[10] = {c.fx, c.fy, c.cx, c.cy, c.k1, c.k2, c.p1, c.p2, c.k3, (double)c.affine};
  for (int i = 0; i < 10; i++) k[9 + i] = v[i];
  return 19;
}
static inline int lring_coord_key(const HomographyCoord& c, const FusedCall&, double* k) {
  for (int i = 0; i < 9; i++) k[i] = c.m[i];
  return 9;
}

// 0 = launched, 1 = not covered (the caller runs the gather kernel), < 0 = error
template <typename ST, int INTERP, typename Coord, int K>
static int lring_try(ipa_ctx* ctx, const FusedCall& f, const Coord& c, const Weights<float, K * K>& w,
                     const SampleRowSrc<ST, INTERP, Coord>& s) {
  using Src = SampleRowSrc<ST, INTERP, Coord>;
  if constexpr (!lring_capable<Src, K>::value) {
    return 1;
  } else {
    if (!ctx->tune.lring || f.q5 || f.n_frames % IPA_WPB != 0 || f.n_frames < ctx->tune.lring_min) return 1;
    if (!ctx->tune.frames_wg || !ctx->tune.frames_inner) return 1;
    if (!f.p.vec_out || (f.p.dw & 3) != 0 || (Src::kMap && !f.map_vec)) return 1;
    WaveParams p = f.p;
    p.strips_x = (p.dw + 255) / 256;
    p.strip_h = wave_strip_height(ctx, p.dh, p.dw, f.n_frames, K, false, 2);
    if (p.strip_h > kLrMaxRows - (K - 1)) p.strip_h = kLrMaxRows - (K - 1);
    p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
    if ((unsigned long)p.strips * (unsigned long)(f.n_frames / IPA_WPB) >= (1ul << 31)) return 1;
    p.frames_inner = f.n_frames;
    p.frames_wg = 1;
    p.frame_major = 0;
    // key of the plan: coordinate source + everything the footprints and the schedule depend on
    double key[48];
    int kn = lring_coord_key(c, f, key);
    const double g[12] = {(double)p.dh, (double)p.dw, (double)f.sh, (double)f.sw, (double)K,
                          (double)p.strip_h, (double)p.cbx, (double)p.cby, (double)kLrR, (double)kLrD,
                          (double)kLrPitch, (double)sizeof(typename Coord::coord_t)};
    for (int i = 0; i < 12; i++) key[kn++] = g[i];
    const bool reusable = !std::is_same<Coord, MapCoord>::value || f.map_static != 0;
    const bool same = ctx->lring_hint_n == kn &&
                      memcmp(ctx->lring_hint_key, key, (size_t)kn * sizeof(double)) == 0;
    // what the last planning pass for this key found (read back without waiting: possibly one
    // call old): a source that leaves most strips to the gather loop - a strong rotation,
    // footprints outside the frame - stays on the gather kernel, whose strips overlap less LDS.
    // Same bits either way.  Every 16th skipped call of a source that is planned per call plans
    // again (its maps may have been rewritten in place).
    if (ctx->tune.lring < 2 && same && ctx->lring_hint && ctx->lring_hint[0] != 0xffffffffu &&
        ctx->lring_hint_strips == p.strips && 2u * ctx->lring_hint[0] < p.strips &&
        (reusable || (++ctx->lring_skips & 15u) != 0))
      return 1;
    const bool hit = reusable && ctx->lplan_key_n == kn &&
                     memcmp(ctx->lplan_key, key, (size_t)kn * sizeof(double)) == 0;
    const size_t plan_b = (((size_t)p.strips * kLrPlanWords * sizeof(unsigned)) + 255) & ~(size_t)255;
    if (!hit) {
      ctx->lplan_key_n = 0;
      if (ctx->lplan_bytes < plan_b + 256) {
        IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->lplan) {
          IPA_HIP(ctx, hipFree(ctx->lplan));
          ctx->lplan = nullptr;
          ctx->lplan_bytes = 0;
        }
        IPA_HIP(ctx, hipMalloc(&ctx->lplan, 2 * plan_b + 256));
        ctx->lplan_bytes = 2 * plan_b + 256;
      }
      if (!ctx->lring_hint)
        IPA_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->lring_hint), 2 * sizeof(unsigned)));
      if (!same) {
        ctx->lring_hint[0] = 0xffffffffu;   // nothing known about this source yet
        ctx->lring_skips = 0;
        memcpy(ctx->lring_hint_key, key, (size_t)kn * sizeof(double));
        ctx->lring_hint_n = kn;
      }
      ctx->lring_hint_strips = p.strips;
      unsigned* stats = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ctx->lplan) + plan_b);
      IPA_HIP(ctx, hipMemsetAsync(stats, 0, sizeof(unsigned), ctx->stream));
      hipLaunchKernelGGL((lring_plan_kernel<Coord, K>), dim3(p.strips), dim3(1024), 0, ctx->stream, p, c,
                         f.sh, f.sw, f.q5, reinterpret_cast<unsigned*>(ctx->lplan), stats);
      IPA_HIP(ctx, hipMemcpyAsync(ctx->lring_hint, stats, sizeof(unsigned), hipMemcpyDeviceToHost,
                                  ctx->stream));
      if (reusable) {
        memcpy(ctx->lplan_key, key, (size_t)kn * sizeof(double));
        ctx->lplan_key_n = kn;
      }
    }
    dim3 grid(p.strips * (unsigned)(f.n_frames / IPA_WPB), 1), block(64 * IPA_WPB);
    hipLaunchKernelGGL((lring_stencil_kernel<Src, K>), grid, block, 0, ctx->stream, p, s, w,
                       reinterpret_cast<const unsigned*>(ctx->lplan));
    return 0;
  }
}

template <typename ST, int INTERP, typename Coord, int K>
static void fused_launch_one(ipa_ctx* ctx, const FusedCall& f, const Coord& c) {
  using Src = SampleRowSrc<ST, INTERP, Coord>;
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
  if (lring_try<ST, INTERP, Coord, K>(ctx, f, c, w, s) == 0) return;
  WaveParams p = f.p;
  using G = wave_geom<K, geom_halo<Src, K, false>::value>;
  p.strips_x = (p.dw + G::OW - 1) / G::OW;
  p.strip_h = wave_strip_height(ctx, p.dh, p.dw, f.n_frames, K, false,
                                fused_strip_piped<Src, K>(ctx, f.n_frames));
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  // frames of one strip block run together: map-based remaps share their map rows between
  // frames (L2 fetch traffic -62 % on 16 x 4K), and even without shared rows the order measured
  // ~5 % faster than frame-after-frame
  dim3 grid = wave_grid(ctx, p, f.n_frames, IPA_WPB, true,
                        coord_is_table<Coord>::value || shared_capable<Src, K>::value);
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
                                pipe_capable<LoadRowSrc, K>::value && IPA_PIPE ? 1 : 0);
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  dim3 grid = wave_grid(ctx, p, n_frames, IPA_WPB, true, false, true), block(64 * IPA_WPB);
  hipLaunchKernelGGL((wave_stencil_kernel<LoadRowSrc, K>), grid, block, 0, ctx->stream, p, src, w);
  return IPA_OK;
}
#endif
