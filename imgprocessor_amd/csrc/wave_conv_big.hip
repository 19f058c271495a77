// wave_conv_big.hip - the 9x9 and 11x11 float32 filter on the wave-marching skeleton
// (wave_stencil.hpp): 81 / 121 coefficients are re-read per input row from the kernel-argument
// segment with scalar loads instead of living in SGPRs (wave_stencil_big_kernel).
// Reference semantics: filters/maskedConvolve.py:24-43 + scipy.ndimage.correlate.
#include "wave_stencil.hpp"

namespace ipa {

template <int K>
static int launch_wave_conv_big(ipa_ctx* ctx, const WaveParams& p0, const LoadRowSrc& src,
                                const double* kernel, int n_frames) {
  WaveBigArgs<LoadRowSrc, K> a;
  a.p = p0;
  a.src = src;
  for (int i = 0; i < K; i++)
    for (int j = 0; j < 12; j++) a.wrows[i][j] = j < K ? (float)kernel[i * K + j] : 0.f;
  using G = wave_geom<K>;
  a.p.strips_x = (a.p.dw + G::OW - 1) / G::OW;
  a.p.strip_h = wave_strip_height(ctx, a.p.dh, a.p.dw, n_frames, K, true, 0, a.p.strips_x);
  a.p.strips = (unsigned)a.p.strips_x * (unsigned)((a.p.dh + a.p.strip_h - 1) / a.p.strip_h);
  dim3 grid = wave_grid(ctx, a.p, n_frames, IPA_WPB, true, false, true), block(64 * IPA_WPB);
  hipLaunchKernelGGL((wave_stencil_big_kernel<LoadRowSrc, K>), grid, block, 0, ctx->stream, a);
  return IPA_OK;
}

}  // namespace ipa

int ipa_wave_conv_launch_k9(ipa_ctx* ctx, const ipa::WaveParams& p, const ipa::LoadRowSrc& src,
                            const double* kernel, int n_frames) {
  return ipa::launch_wave_conv_big<9>(ctx, p, src, kernel, n_frames);
}
int ipa_wave_conv_launch_k11(ipa_ctx* ctx, const ipa::WaveParams& p, const ipa::LoadRowSrc& src,
                             const double* kernel, int n_frames) {
  return ipa::launch_wave_conv_big<11>(ctx, p, src, kernel, n_frames);
}
