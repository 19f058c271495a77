// fused_pair.hip - bilinear map-based remap -> 3x3 / 5x5 filter of float32 batches with one wave
// per strip of a frame PAIR (wave_pair.hpp).
// Reference call chain: camera/LensDistortion.py:323-326 + filters/maskedConvolve.py:24-43.
#include "fused_impl.hpp"
#include "wave_pair.hpp"
#include "wave_split.hpp"

int ipa_fused_launch_k3(ipa_ctx*, const ipa::FusedCall&);  // fused_k3.hip
int ipa_fused_launch_k5(ipa_ctx*, const ipa::FusedCall&);  // fused_k5.hip

namespace ipa {

template <int K> static void pair_launch(ipa_ctx* ctx, const FusedCall& f) {
  using Src = SampleRowSrc<float, kLinear, MapCoord>;
  Weights<float, K * K> w;
  for (int i = 0; i < K * K; i++) w.w[i] = (float)f.kernel[i];
  Src s;
  s.coord = f.map;
  s.src = f.src; s.src_frame_bytes = f.src_frame_bytes; s.src_bytes = f.src_bytes;
  s.sh = f.sh; s.sw = f.sw; s.spitch = f.spitch;
  s.border = f.border; s.q5 = f.q5; s.cubic_a = f.cubic_a; s.lanczos = nullptr;
  s.cval = (float)f.cval; s.ccval = (float)f.conv_cval; s.map_vec = f.map_vec;
  WaveParams p = f.p;
  using G = wave_geom<K>;
  p.strips_x = (p.dw + G::OW - 1) / G::OW;
  p.strip_h = wave_strip_height(ctx, p.dh, p.dw, f.n_frames, K);
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  p.frames_inner = 0;
  const unsigned blocks = (p.strips + IPA_WPB - 1) / IPA_WPB;
  const unsigned pairs = ((unsigned)f.n_frames + 1u) / 2u;
  hipLaunchKernelGGL((wave_pair_kernel<K>), dim3(blocks * pairs), dim3(64 * IPA_WPB), 0, ctx->stream,
                     p, s, w, f.n_frames);
}

// sampler wave + filter wave per strip (wave_split.hpp), knob pair = 2
template <int K> static void split_launch(ipa_ctx* ctx, const FusedCall& f) {
  using Src = SampleRowSrc<float, kLinear, MapCoord>;
  Weights<float, K * K> w;
  for (int i = 0; i < K * K; i++) w.w[i] = (float)f.kernel[i];
  Src s;
  s.coord = f.map;
  s.src = f.src; s.src_frame_bytes = f.src_frame_bytes; s.src_bytes = f.src_bytes;
  s.sh = f.sh; s.sw = f.sw; s.spitch = f.spitch;
  s.border = f.border; s.q5 = f.q5; s.cubic_a = f.cubic_a; s.lanczos = nullptr;
  s.cval = (float)f.cval; s.ccval = (float)f.conv_cval; s.map_vec = f.map_vec;
  WaveParams p = f.p;
  using G = wave_geom<K>;
  p.strips_x = (p.dw + G::OW - 1) / G::OW;
  p.strip_h = wave_strip_height(ctx, p.dh, p.dw, f.n_frames, K);
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  p.frames_inner = f.n_frames;
  hipLaunchKernelGGL((wave_split_kernel<K>), dim3(p.strips * (unsigned)f.n_frames), dim3(128), 0,
                     ctx->stream, p, s, w);
  // the rim strips on the per-frame kernel
  FusedCall rim = f;
  rim.p.rim_only = 1;
  if (K == 3) ipa_fused_launch_k3(ctx, rim);
  else ipa_fused_launch_k5(ctx, rim);
}

}  // namespace ipa

// returns 1 when the call is not covered
int ipa_fused_pair_launch(ipa_ctx* ctx, const ipa::FusedCall& f, int K) {
  using namespace ipa;
  if (f.dst_dt != IPA_F32 || f.src_dt != IPA_F32 || f.interp_base != IPA_INTER_LINEAR ||
      f.coord_kind != 0 || f.n_frames < 2)
    return 1;
  if ((unsigned long)((f.p.strips_x ? f.p.strips_x : 1)) * (unsigned long)f.n_frames >= (1ul << 30))
    return 1;
  if (ctx->tune.pair == 2) {
    if ((unsigned long)f.p.dh * (unsigned long)f.n_frames >= (1ul << 30)) return 1;
    switch (K) {
      case 3: split_launch<3>(ctx, f); return 0;
      case 5: split_launch<5>(ctx, f); return 0;
      default: return 1;
    }
  }
  switch (K) {
    case 3: pair_launch<3>(ctx, f); return 0;
    case 5: pair_launch<5>(ctx, f); return 0;
    default: return 1;
  }
}
