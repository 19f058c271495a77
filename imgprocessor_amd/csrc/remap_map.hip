// remap_map.hip — map-based remap kernels (cv2.remap with CV_32FC1 maps)
#include "remap_impl.hpp"
int ipa_remap_launch_map(ipa_ctx* ctx, const RemapCall& a, const MapCoord& c, int map_vec) {
  return remap_dispatch<MapCoord>(ctx, a, c, map_vec);
}
