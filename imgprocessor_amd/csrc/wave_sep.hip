// wave_sep.hip - the plain separable filter on the wave-marching skeleton (wave_sep.hpp)
#include "wave_sep.hpp"

// returns 1 when the shape is not covered (caller falls back to the LDS kernel)
int ipa_wave_sep_launch(ipa_ctx* ctx, const ipa::WaveParams& p, const ipa::LoadRowSrc& src,
                        const double* ky, int nky, const double* kx, int nkx, int n_frames,
                        float xcval) {
  using namespace ipa;
  if (nky != nkx) return 1;
  switch (nky) {
    case 3: launch_sep<LoadRowSrc, 3>(ctx, p, src, ky, kx, n_frames, xcval); return 0;
    case 5: launch_sep<LoadRowSrc, 5>(ctx, p, src, ky, kx, n_frames, xcval); return 0;
    case 7: launch_sep<LoadRowSrc, 7>(ctx, p, src, ky, kx, n_frames, xcval); return 0;
    case 9: launch_sep<LoadRowSrc, 9>(ctx, p, src, ky, kx, n_frames, xcval); return 0;
  }
  return 1;
}
