// tile_warp_a.hip - the tile warp kernels (tile_warp.hpp) on 64 x 32 output tiles
#include "common.hpp"
#include "tile_warp.hpp"

namespace ipa {
void tile_warp_run_a(hipStream_t stream, const TileWarpArgs& t, const HomographyCoord& coord, int interp,
                     bool u16, int /*shape*/, unsigned grid, size_t lds) {
  tile_warp_run_shape<64, 32>(stream, t, coord, interp, u16, grid, lds);
}
}  // namespace ipa
