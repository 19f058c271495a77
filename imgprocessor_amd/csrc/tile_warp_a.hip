// tile_warp_a.hip - the tile warp kernels (tile_warp.hpp) on 64 x 32 output tiles
#include "common.hpp"
#include "tile_warp.hpp"

namespace ipa {
void tile_warp_run_a(hipStream_t stream, const TileWarpArgs& t, const HomographyCoord& coord, int interp,
                     bool u16, int /*shape*/, unsigned grid, size_t lds) {
  tile_warp_run_shape<64, 32>(stream, t, coord, interp, u16, grid, lds);
}
void tile_warp_run_map(hipStream_t stream, const TileWarpArgs& t, const MapCoord& coord, int interp,
                       unsigned grid, size_t lds) {
  if (interp == kLanczos4)
    hipLaunchKernelGGL((tile_warp_kernel<kLanczos4, float, 64, 32, MapCoord>), dim3(grid), dim3(256), lds, stream, t, coord);
  else if (interp == kCubic)
    hipLaunchKernelGGL((tile_warp_kernel<kCubic, float, 64, 32, MapCoord>), dim3(grid), dim3(256), lds, stream, t, coord);
  else
    hipLaunchKernelGGL((tile_warp_kernel<kLinear, float, 64, 32, MapCoord>), dim3(grid), dim3(256), lds, stream, t, coord);
}
}  // namespace ipa
