// fused_sep_c.hip - the remap ALONE on the marching strips: wave_sep_kernel with K = 1 (no filter, no halo;
// fused_sep_impl.hpp, wave_sep.hpp::sep_geom<1>) - the standalone bilinear remap of frame batches (knob strip_remap)
#include "fused_sep_impl.hpp"

void ipa_fused_sep_launch_c(ipa_ctx* ctx, const ipa::FusedCall& f, const ipa::FusedSep& q) {
  ipa::fused_sep_k<1>(ctx, f, q);
}
