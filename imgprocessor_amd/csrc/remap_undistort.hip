// remap_undistort.hip — analytic lens-undistort remap kernels (no map arrays)
#include "remap_impl.hpp"
int ipa_remap_launch_undistort(ipa_ctx* ctx, const RemapCall& a, const UndistortCoord& c) {
  return remap_dispatch<UndistortCoord>(ctx, a, c, 0);
}
