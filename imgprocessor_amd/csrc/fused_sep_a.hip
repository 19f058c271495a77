// fused_sep_a.hip — remap -> separable filter, 3 and 5 taps (fused_sep_impl.hpp)
#include "fused_sep_impl.hpp"

void ipa_fused_sep_launch_a(ipa_ctx* ctx, const ipa::FusedCall& f, const ipa::FusedSep& q) {
  if (q.n == 3) ipa::fused_sep_k<3>(ctx, f, q);
  else ipa::fused_sep_k<5>(ctx, f, q);
}
