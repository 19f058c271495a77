// tile_chain.hip - perspective warp + separable filter in one launch (tile_chain.hpp): the kernels'
// instantiations, the host's walk over the regions (the LDS box of a launch) and the launch
#include "common.hpp"
#include "tile_chain.hpp"

namespace ipa {

template <int INTERP, int K>
static int chain_launch(hipStream_t stream, const TileChainArgs& t, const HomographyCoord& coord,
                        unsigned grid, size_t lds) {
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)tile_chain_kernel<INTERP, K>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return 1;
  }
  hipLaunchKernelGGL((tile_chain_kernel<INTERP, K>), dim3(grid), dim3(256), lds, stream, t, coord);
  return 0;
}

int tile_chain_run(hipStream_t stream, const TileChainArgs& t, const HomographyCoord& coord, int interp,
                   int K, unsigned grid, size_t lds) {
  if (interp == kLinear) {
    switch (K) {
      case 3: return chain_launch<kLinear, 3>(stream, t, coord, grid, lds);
      case 5: return chain_launch<kLinear, 5>(stream, t, coord, grid, lds);
      case 7: return chain_launch<kLinear, 7>(stream, t, coord, grid, lds);
      case 9: return chain_launch<kLinear, 9>(stream, t, coord, grid, lds);
    }
  } else if (interp == kCubic) {
    switch (K) {
      case 3: return chain_launch<kCubic, 3>(stream, t, coord, grid, lds);
      case 5: return chain_launch<kCubic, 5>(stream, t, coord, grid, lds);
      case 7: return chain_launch<kCubic, 7>(stream, t, coord, grid, lds);
      case 9: return chain_launch<kCubic, 9>(stream, t, coord, grid, lds);
    }
  }
  return 1;
}

}  // namespace ipa

using namespace ipa;

// 0: launched; 1: not a chain for this kernel (the caller's two launches take it); < 0: error.
// M maps output to source pixels (cv2.WARP_INVERSE_MAP form, as ipa_warp_perspective_dev takes it)
int ipa_tile_chain_launch(ipa_ctx* ctx, const void* d_src, int sh, int sw, long src_pitch, const double* M,
                          const double* ky, const double* kx, int K, void* d_dst, int dh, int dw,
                          long dst_pitch, int n_frames, long src_frame_stride, long dst_frame_stride,
                          int interp, int border_mode, double border_value, int cby, int cbx) {
  const int base = interp & 0xff;
  const bool cubic = base == IPA_INTER_CUBIC_CV || base == IPA_INTER_CUBIC_KEYS;
  if (base != IPA_INTER_LINEAR && !cubic) return 1;
  if (K != 3 && K != 5 && K != 7 && K != 9) return 1;
  if (dh < 1 || dw < 1 || sh < 1 || sw < 1 || n_frames < 1) return 1;
  if (src_pitch >= (1l << 23) || sh >= (1 << 23)) return 1;
  const size_t frame_bytes = ((size_t)(sh - 1) * src_pitch + sw) * 4;
  if (frame_bytes >= (1ull << 31)) return 1;
  const int H = K / 2;
  int steps = ctx->tune.chain_steps > 0 ? ctx->tune.chain_steps : 4;
  while (steps > 1 && kChainRows * (steps - 1) - 2 * H >= dh) steps--;   // (short pictures: one segment)
  if (kChainRows * steps - 2 * H < 1) return 1;
  HomographyCoord coord;
  for (int i = 0; i < 9; i++) coord.m[i] = M[i];
  const int NT = cubic ? 4 : 2;
  TileChainArgs t;
  {
    double key[14] = {(double)(64 + (cubic ? 1 : 0) + 2 * K + 32 * steps), (double)dh, (double)dw, (double)sh, (double)sw};
    for (int k = 0; k < 9; k++) key[5 + k] = coord.m[k];
    ipa_ctx::TileWarpPlan* pl = nullptr;
    for (auto& q : ctx->tile_warp_plans)
      if (q.valid && memcmp(key, q.key, sizeof key) == 0) pl = &q;
    if (!pl) {
      pl = &ctx->tile_warp_plans[0];
      for (auto& q : ctx->tile_warp_plans)
        if (!q.valid || q.used < pl->used) { pl = &q; if (!q.valid) break; }
      int pitch = 0, rows = 0;
      const bool ok = cubic ? tile_chain_box<4>(coord.m, dh, dw, sh, sw, H, steps, &pitch, &rows)
                            : tile_chain_box<2>(coord.m, dh, dw, sh, sw, H, steps, &pitch, &rows);
      pl->ok = ok ? 1 : 0;
      pl->shape = 0;
      pl->rows = rows;
      pl->pitch = 0;
      if (ok) pl->pitch = cubic ? tile_warp_pitch<4>(coord.m, dh, dw, pitch, rows)
                                : tile_warp_pitch<2>(coord.m, dh, dw, pitch, rows);
      pl->drift = pl->step = pl->fetch = 0;
      memcpy(pl->key, key, sizeof key);
      pl->valid = 1;
    }
    pl->used = ++ctx->tile_warp_clock;
    if (!pl->ok) return 1;
    t.pitch = pl->pitch;
    t.rows = pl->rows;
  }
  (void)NT;
  t.dst = (char*)d_dst; t.dst_frame_elems = dst_frame_stride; t.dpitch = dst_pitch;
  t.src = (const char*)d_src; t.src_frame_bytes = src_frame_stride * 4; t.src_bytes = (unsigned)frame_bytes;
  t.sh = sh; t.sw = sw; t.spitch = (int)src_pitch; t.dh = dh; t.dw = dw;
  t.n_frames = n_frames;
  t.border = border_mode; t.q5 = (interp & IPA_INTER_Q5) ? 1 : 0;
  t.cubic_a = base == IPA_INTER_CUBIC_KEYS ? -0.5f : -0.75f;
  t.cval = (float)border_value;
  t.cby = cby; t.cbx = cbx; t.ccval = 0.0f;
  t.steps = steps;
  t.seg_rows = kChainRows * steps - 2 * H;
  t.strips_x = (dw + 63) / 64;
  t.segs = (dh + t.seg_rows - 1) / t.seg_rows;
  for (int i = 0; i < 9; i++) {
    t.ky[i] = i < K ? (float)ky[i] : 0.f;
    t.kx[i] = i < K ? (float)kx[i] : 0.f;
  }
  t.vec_out = ((uintptr_t)d_dst % 16 == 0 && dst_pitch % 4 == 0 && dst_frame_stride % 4 == 0) ? 1 : 0;
  // frames a workgroup walks through with one evaluation of a step's coordinates
  const long units = (long)t.strips_x * t.segs;
  t.frames_wg = ctx->tune.chain_frames > 0 ? ctx->tune.chain_frames : kChainMaxFrames;
  if (t.frames_wg > kChainMaxFrames) t.frames_wg = kChainMaxFrames;
  if (ctx->tune.chain_frames <= 0)
    while (t.frames_wg > 1 && units * ((n_frames + t.frames_wg - 1) / t.frames_wg) < 3072) t.frames_wg >>= 1;
  if (t.frames_wg > n_frames) t.frames_wg = n_frames;
  const unsigned groups = ((unsigned)n_frames + t.frames_wg - 1) / (unsigned)t.frames_wg;
  if ((unsigned long)units * groups >= (1ul << 31)) return 1;
  {   // (wave_grid's rule for the frame groups of a launch: a quarter of them at a time)
    int gc = ctx->tune.group_chunk;
    if (gc < 0) gc = groups >= 4 ? (int)groups / 4 : 0;
    t.group_chunk = (gc > 0 && gc < (int)groups && groups % (unsigned)gc == 0) ? gc : 0;
  }
  size_t lds = 0;
  switch (K) {
    case 3: lds = tile_chain_lds_bytes<3>(t.pitch, t.rows, t.frames_wg); break;
    case 5: lds = tile_chain_lds_bytes<5>(t.pitch, t.rows, t.frames_wg); break;
    case 7: lds = tile_chain_lds_bytes<7>(t.pitch, t.rows, t.frames_wg); break;
    default: lds = tile_chain_lds_bytes<9>(t.pitch, t.rows, t.frames_wg); break;
  }
  if (lds > 150 * 1024) return 1;
  if (hipSetDevice(ctx->device) != hipSuccess) return 1;
  const int rc = tile_chain_run(ctx->stream, t, coord, cubic ? kCubic : kLinear, K, (unsigned)units * groups, lds);
  if (rc == 0) ctx->chain_launches++;
  return rc;
}
