// stored_coords.hpp — the coordinates of a source given by value (homography, lens model),
// evaluated ONCE per (parameters, geometry) into the context's plan buffer in the source's own
// coordinate type and read by every frame of a batch (and by the next call with the same key):
// the gather remap kernels for bicubic / Lanczos4 (remap_impl.hpp) and, since round 4, the
// shared-record loop of the fused bilinear chains (wave_pipe.hpp) - the homography's double
// coordinates with their division were a third of the C3 chain.  Same bits: the table holds what
// the per-pixel evaluation yields.
// Reference call sites: camera/PerspectiveCorrection.py:380-406 (cv2.warpPerspective).
#pragma once

#include "common.hpp"
#include "sampler.hpp"

namespace ipa {

// the parameters of a coordinate source given by value, for the plan buffer's reuse key
static inline int coord_key(const MapCoord& c, double* k) {
  k[0] = (double)reinterpret_cast<uintptr_t>(c.mx);
  k[1] = (double)reinterpret_cast<uintptr_t>(c.my);
  k[2] = (double)c.pitch;
  return 3;
}
static inline int coord_key(const UndistortCoord& c, double* k) {
  for (int i = 0; i < 9; i++) k[i] = c.ir[i];
  const double v[10] = {c.fx, c.fy, c.cx, c.cy, c.k1, c.k2, c.p1, c.p2, c.k3, (double)c.affine};
  for (int i = 0; i < 10; i++) k[9 + i] = v[i];
  return 19;
}
static inline int coord_key(const HomographyCoord& c, double* k) {
  for (int i = 0; i < 9; i++) k[i] = c.m[i];
  return 9;
}

// ------------------------------------------------------------ stored coordinates --
// A coordinate source given by value (homography: double coordinates, ~40 float64 operations
// per pixel; lens model) evaluated ONCE per (source, geometry) into the context's plan buffer, in
// the source's own type - the frames of a batch then sample through remap_kernel<StoredCoord>
// exactly what the per-pixel evaluation would give (same coordinate bits), in the
// lane-interleaved order of the map-based kernel.  16 x 4K, 7 degrees + perspective (no clean
// strips for the ring kernel): bicubic 0.648 -> 0.600 ms, Lanczos4 1.853 -> 1.467 ms (the same warp
// from float32 maps, i.e. other coordinate bits: 0.502 / 1.344).
template <typename Coord>
__global__ void __launch_bounds__(256)
store_coords_kernel(Coord c, int dh, int dw, typename Coord::coord_t* __restrict__ ox,
                    typename Coord::coord_t* __restrict__ oy) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
  if (u >= dw) return;
  typename Coord::coord_t sx, sy;
  c.get(u, v, sx, sy);
  ox[(long)v * dw + u] = sx;
  oy[(long)v * dw + u] = sy;
}

template <typename Coord>
static int stored_coords_prepare(ipa_ctx* ctx, const Coord& coord, int dh, int dw,
                                 StoredCoord<typename Coord::coord_t>* sc) {
  using CT = typename Coord::coord_t;
  if (dh > 65535) return 1;
  double key[40];
  int kn = coord_key(coord, key);
  key[kn++] = (double)dh; key[kn++] = (double)dw; key[kn++] = (double)sizeof(CT);
  key[kn++] = 7777.0;   // (not a ring plan: those keys are longer)
  const size_t plane = (((size_t)dh * dw * sizeof(CT)) + 255) & ~(size_t)255;
  const bool hit = ctx->plan_key_n == kn &&
                   memcmp(ctx->plan_key, key, (size_t)kn * sizeof(double)) == 0 &&
                   ctx->plan_bytes >= 2 * plane;
  if (!hit) {
    int rc = ipa_plan_reserve(ctx, 2 * plane);
    if (rc) return rc;
  }
  CT* ox = reinterpret_cast<CT*>(ctx->plan);
  CT* oy = reinterpret_cast<CT*>(reinterpret_cast<char*>(ctx->plan) + plane);
  if (!hit) {
    hipLaunchKernelGGL((store_coords_kernel<Coord>), dim3((unsigned)((dw + 255) / 256), (unsigned)dh),
                       dim3(256), 0, ctx->stream, coord, dh, dw, ox, oy);
    memcpy(ctx->plan_key, key, (size_t)kn * sizeof(double));
    ctx->plan_key_n = kn;
  }
  *sc = StoredCoord<CT>{ox, oy, (long)dw};
  return 0;
}

}  // namespace ipa
