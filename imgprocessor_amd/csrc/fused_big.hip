// fused_big.hip - map-based bilinear remap (float32) -> 9x9 / 11x11 filter in one kernel: the
// sampling row source of wave_stencil.hpp on wave_stencil_big_kernel (coefficient rows streamed
// through SGPRs).  The remap alone is bound by the vector-memory front end, the filter alone by
// its K*K fmas; in one kernel the two overlap across the waves of a CU and the intermediate
// image never exists in HBM: 16 x 4K, 9x9 673 -> 501 us, 11x11 748 -> 591 us against two
// launches.  Bicubic was built and measured slower in one kernel (185-193 VGPRs, occupancy 2:
// 858 vs 768 us, 953 vs 880 us) and stays on two launches.
// 7x7 (float32 and uint16 frames, BASELINE configuration C4) runs here too: next to the
// sampling source's scalar state 49 resident coefficients overflow the SGPR file (331 spills
// in wave_stencil_kernel); streamed, the 4K chain measured 489 -> 449 us / 493 -> 460 us.
// Reference call chain: camera/PerspectiveCorrection.py:401-405 / camera/LensDistortion.py:323-326
// followed by filters/maskedConvolve.py:24-43.
#include "fused_impl.hpp"

namespace ipa {

template <typename ST, int INTERP, int K>
static void fused_big_launch_one(ipa_ctx* ctx, const FusedCall& f) {
  using Src = SampleRowSrc<ST, INTERP, MapCoord>;
  WaveBigArgs<Src, K> a;
  for (int i = 0; i < K; i++)
    for (int j = 0; j < 12; j++) a.wrows[i][j] = j < K ? (float)f.kernel[i * K + j] : 0.f;
  Src& s = a.src;
  s.coord = f.map;
  s.src = f.src; s.src_frame_bytes = f.src_frame_bytes; s.src_bytes = f.src_bytes;
  s.sh = f.sh; s.sw = f.sw; s.spitch = f.spitch;
  s.border = f.border; s.q5 = f.q5; s.cubic_a = f.cubic_a; s.lanczos = nullptr;
  s.cval = (float)f.cval; s.ccval = (float)f.conv_cval; s.map_vec = f.map_vec;
  a.p = f.p;
  using G = wave_geom<K>;
  a.p.strips_x = (a.p.dw + G::OW - 1) / G::OW;
  a.p.strip_h = wave_strip_height(ctx, a.p.dh, a.p.dw, f.n_frames, K, true, 0, a.p.strips_x);
  a.p.strips = (unsigned)a.p.strips_x * (unsigned)((a.p.dh + a.p.strip_h - 1) / a.p.strip_h);
  dim3 grid = wave_grid(ctx, a.p, f.n_frames, IPA_WPB, true, coord_is_table<typename Src::coord_type>::value, false, K),
       block(64 * IPA_WPB);
  hipLaunchKernelGGL((wave_stencil_big_kernel<Src, K>), grid, block, 0, ctx->stream, a);
}

}  // namespace ipa

// returns 1 when the call is not covered (the caller then runs remap and filter as two launches)
int ipa_fused_big_launch(ipa_ctx* ctx, const ipa::FusedCall& f, int K) {
  using namespace ipa;
  if (f.dst_dt != IPA_F32 || f.coord_kind != 0 || f.interp_base != IPA_INTER_LINEAR) return 1;
  if (K == 7 && f.src_dt == IPA_U16) {
    fused_big_launch_one<uint16_t, kLinear, 7>(ctx, f);
    return 0;
  }
  if (f.src_dt != IPA_F32) return 1;
  if (K == 7) fused_big_launch_one<float, kLinear, 7>(ctx, f);
  else if (K == 9) fused_big_launch_one<float, kLinear, 9>(ctx, f);
  else if (K == 11) fused_big_launch_one<float, kLinear, 11>(ctx, f);
  else return 1;
  return 0;
}
