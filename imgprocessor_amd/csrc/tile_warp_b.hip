// tile_warp_b.hip - the tile warp kernels (tile_warp.hpp) on 32 x 32 and 32 x 16 output tiles:
// homographies that shrink parts of the picture, whose 64 x 32 tiles would span too much source
#include "common.hpp"
#include "tile_warp.hpp"

namespace ipa {
void tile_warp_run_b(hipStream_t stream, const TileWarpArgs& t, const HomographyCoord& coord, int interp,
                     bool u16, int shape, unsigned grid, size_t lds) {
  if (shape == 1) tile_warp_run_shape<32, 32>(stream, t, coord, interp, u16, grid, lds);
  else tile_warp_run_shape<32, 16>(stream, t, coord, interp, u16, grid, lds);
}
}  // namespace ipa
