// fused_ring.hip - bilinear remap -> K x K filter for batches, clean strips on the ring kernel
// (ring_stencil.hpp): a planning launch decides per strip, the ring kernel computes the clean
// strips from LDS, the per-frame kernel (fused_impl.hpp) the rest.
// Reference call chain: camera/LensDistortion.py:323-326 / camera/PerspectiveCorrection.py:401-405
// followed by filters/maskedConvolve.py:24-43.
#include "fused_impl.hpp"
#include "ring_stencil.hpp"

namespace ipa {

template <typename ST, typename Coord, int K>
static int ring_launch(ipa_ctx* ctx, FusedCall& f, const Coord& c) {
  using G = group_geom<K>;
  static_assert(2 * G::OW == wave_geom<K>::OW, "two ring strips = one strip of the per-frame kernel");
  RingGeom gm;
  gm.dh = f.p.dh; gm.dw = f.p.dw;
  gm.strips_x = (gm.dw + G::OW - 1) / G::OW;
  gm.pairs_x = (gm.strips_x + 1) / 2;
  // the per-frame kernel's strip height for this launch (fused_launch_one computes the same)
  using Src = SampleRowSrc<ST, kLinear, Coord>;
  gm.strip_h = wave_strip_height(ctx, gm.dh, gm.dw, f.n_frames, K, false,
                                 fused_strip_piped<Src, K>(ctx, f.n_frames));
  const int rows = (gm.dh + gm.strip_h - 1) / gm.strip_h;
  gm.strips = gm.strips_x * rows;
  gm.pairs = gm.pairs_x * rows;
  if (gm.strip_h + K - 1 > 16 * kPlanWords) return 1;

  const size_t info_b = (size_t)gm.strips * sizeof(int4);
  const size_t cnts_b = (size_t)gm.strips * kPlanWords * sizeof(unsigned);
  const size_t pair_b = (size_t)gm.pairs * sizeof(unsigned);
  int rc = ipa_plan_reserve(ctx, info_b + cnts_b + pair_b);
  if (rc) return rc;
  RingPlan plan;
  plan.info = reinterpret_cast<int4*>(ctx->plan);
  plan.cnts = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ctx->plan) + info_b);
  plan.pair_clean = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ctx->plan) + info_b + cnts_b);
  plan.stats = nullptr;

  hipLaunchKernelGGL((ring_plan_kernel<Coord, K>), dim3(gm.pairs), dim3(128), 0, ctx->stream, gm,
                     c, f.sh, f.sw, RingTaps{2, 0, kRR}, plan, nullptr, nullptr);

  RingSrc<ST, Coord> g;
  g.coord = c;
  g.src = f.src; g.src_frame_bytes = f.src_frame_bytes; g.src_bytes = f.src_bytes;
  g.sh = f.sh; g.sw = f.sw; g.spitch = f.spitch;
  g.n_frames = f.n_frames;
  g.ablate = ctx->tune.ring_ablate;
  Weights<float, K * K> w;
  for (int i = 0; i < K * K; i++) w.w[i] = (float)f.kernel[i];
  const unsigned groups = ((unsigned)f.n_frames + 3u) / 4u;
  hipLaunchKernelGGL((ring_kernel<ST, Coord, K>), dim3((unsigned)gm.strips * groups), dim3(256), 0,
                     ctx->stream, f.p, gm, g, plan, w);
  f.p.skip = plan.pair_clean;
  return 0;
}

template <typename ST, int K> static int ring_launch_coord(ipa_ctx* ctx, FusedCall& f) {
  switch (f.coord_kind) {
    case 0: return ring_launch<ST, MapCoord, K>(ctx, f, f.map);
    case 1: return ring_launch<ST, UndistortCoord, K>(ctx, f, f.und);
    default: return ring_launch<ST, HomographyCoord, K>(ctx, f, f.hom);
  }
}

}  // namespace ipa

int ipa_fused_ring_launch(ipa_ctx* ctx, ipa::FusedCall& f, int K) {
  using namespace ipa;
  // exact-coordinate bilinear of float32 frames into aligned float32 rows
  if (f.dst_dt != IPA_F32 || f.src_dt != IPA_F32 || f.interp_base != IPA_INTER_LINEAR || f.q5 ||
      !f.p.vec_out)
    return 1;
  if ((unsigned long)f.p.dh * f.p.dw * (unsigned long)f.n_frames >= (1ul << 40)) return 1;
  // destination frames are addressed through a 32-bit buffer descriptor
  if (((unsigned long)(f.p.dh - 1) * f.p.dpitch + f.p.dw) * 4ul >= 0xff000000ul) return 1;
  switch (K) {
    case 3: return ring_launch_coord<float, 3>(ctx, f);
    case 5: return ring_launch_coord<float, 5>(ctx, f);
    default: return 1;
  }
}
