// remap_homography.hip — perspective-warp kernels (cv2.warpPerspective)
#include "remap_impl.hpp"
int ipa_remap_launch_homography(ipa_ctx* ctx, const RemapCall& a, const HomographyCoord& c) {
  return remap_dispatch<HomographyCoord>(ctx, a, c, 0);
}
