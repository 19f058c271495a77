// fused_ring_big.hip - bilinear / bicubic remap -> dense 7x7 / 9x9 / 11x11 filter of float32
// batches in one kernel, taps from an LDS ring (ring_big.hpp).  BASELINE configuration C5.
// Reference call chain: camera/PerspectiveCorrection.py:401-405 / camera/LensDistortion.py:323-326
// followed by filters/maskedConvolve.py:24-43.
#include "fused_impl.hpp"
#include "ring_big.hpp"

namespace ipa {

template <int INTERP, typename Coord, int K>
static int ring_big_launch(ipa_ctx* ctx, const FusedCall& f, const Coord& c) {
  using G = group_geom<K>;
  using RK = RingBigKernel<INTERP, Coord, K>;
  RingBigArgs<INTERP, Coord, K> a;
  RingGeom& gm = a.gm;
  gm.dh = f.p.dh; gm.dw = f.p.dw;
  gm.strips_x = (gm.dw + G::OW - 1) / G::OW;
  gm.pairs_x = (gm.strips_x + 1) / 2;
  // the K*K fmas per sample bound the kernel, halo rows included: tall strips
  gm.strip_h = wave_strip_height(ctx, gm.dh, 2 * gm.dw, f.n_frames, K, true);
  if (gm.strip_h + K - 1 > 16 * kPlanWords) return 1;
  const int rows = (gm.dh + gm.strip_h - 1) / gm.strip_h;
  gm.strips = gm.strips_x * rows;
  gm.pairs = gm.pairs_x * rows;
  RingTaps tp;
  tp.nt = ntaps<INTERP>::value;
  tp.q5 = f.q5;
  tp.rr = RK::RR;
  int rc = ring_plan_prepare<Coord, K>(ctx, gm, c, f.sh, f.sw, tp, &a.plan, &a.g.kc);
  if (rc) return rc;
  a.p = f.p;
  a.g.coord = c;
  a.g.src = f.src; a.g.src_frame_bytes = f.src_frame_bytes; a.g.src_bytes = f.src_bytes;
  a.g.sh = f.sh; a.g.sw = f.sw; a.g.spitch = f.spitch;
  a.g.border = f.border; a.g.q5 = f.q5; a.g.cubic_a = f.cubic_a;
  a.g.cval = (float)f.cval; a.g.ccval = (float)f.conv_cval;
  a.g.n_frames = f.n_frames;
  for (int i = 0; i < K; i++)
    for (int j = 0; j < 12; j++) a.wrows[i][j] = j < K ? (float)f.kernel[i * K + j] : 0.f;
  const unsigned groups = ((unsigned)f.n_frames + RK::kWaves - 1) / RK::kWaves;
  hipLaunchKernelGGL((ring_big_kernel<INTERP, Coord, K>), dim3((unsigned)gm.strips * groups),
                     dim3(64 * RK::kWaves), 0, ctx->stream, a);
  return 0;
}

template <int INTERP, int K> static int ring_big_coord(ipa_ctx* ctx, const FusedCall& f) {
  switch (f.coord_kind) {
    case 0: return ring_big_launch<INTERP, MapCoord, K>(ctx, f, f.map);
    case 2: return ring_big_launch<INTERP, HomographyCoord, K>(ctx, f, f.hom);
    default: return 1;  // the lens model arrives here as its cached maps (fused.hip)
  }
}

template <int INTERP> static int ring_big_k(ipa_ctx* ctx, const FusedCall& f, int K) {
  switch (K) {
    case 7: return ring_big_coord<INTERP, 7>(ctx, f);
    case 9: return ring_big_coord<INTERP, 9>(ctx, f);
    case 11: return ring_big_coord<INTERP, 11>(ctx, f);
    default: return 1;
  }
}

}  // namespace ipa

// returns 1 when the call is not covered
int ipa_fused_ring_big_launch(ipa_ctx* ctx, const ipa::FusedCall& f, int K) {
  using namespace ipa;
  if (f.dst_dt != IPA_F32 || f.src_dt != IPA_F32) return 1;
  if ((unsigned long)f.p.dh * f.p.dw * (unsigned long)f.n_frames >= (1ul << 40)) return 1;
  switch (f.interp_base) {
    case IPA_INTER_LINEAR: return ring_big_k<kLinear>(ctx, f, K);
    case IPA_INTER_CUBIC_CV:
    case IPA_INTER_CUBIC_KEYS: return ring_big_k<kCubic>(ctx, f, K);
    default: return 1;
  }
}
