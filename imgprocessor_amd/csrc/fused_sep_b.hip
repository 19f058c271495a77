// fused_sep_b.hip — remap -> separable filter, 7 and 9 taps (fused_sep_impl.hpp)
#include "fused_sep_impl.hpp"

void ipa_fused_sep_launch_b(ipa_ctx* ctx, const ipa::FusedCall& f, const ipa::FusedSep& q) {
  if (q.n == 7) ipa::fused_sep_k<7>(ctx, f, q);
  else ipa::fused_sep_k<9>(ctx, f, q);
}
