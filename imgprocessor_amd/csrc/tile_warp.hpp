// tile_warp.hpp — cv2.warpPerspective (camera/PerspectiveCorrection.py:377-378, 401-405) for
// homographies that ROTATE the picture: the source box of an output tile staged in LDS.
//
// The gather kernel (remap_impl.hpp) and the ring kernel (ring_remap.hpp) both assume that an
// output row walks along a source row: a wave's gather then touches 2-3 cache lines and a strip's
// footprints advance row by row.  Under a rotation of a few degrees neither holds - a gather of 64
// lanes crosses a source row every 1 / sin(angle) pixels and pays the texture addresser per line
// touched, the ring's strips stop being "clean" - and the time of a warp doubles to triples
// (tools/angle_sweep.py, profiles/r04_micro.txt).  Here a workgroup owns a 64 x 32 output tile:
//   * the tile's corners through the homography give the bounding box of its footprints in the
//     source (a projective map takes the tile to a convex quadrilateral: the extremes are at the
//     corners);
//   * the box is read row by row with coalesced dword loads - every cell through the border mode,
//     so the samples below need no border logic - into LDS; the next frame's box is requested
//     while the current frame is being sampled;
//   * every pixel's NT x NT taps come from LDS: the weights of axis_split(), the products and sums
//     of sample() in its order - results identical to remap_kernel;
//   * footprints the box does not hold (outside the source altogether, not finite, or a box
//     clipped to the LDS the launch reserved) go through a rolled-loop restatement of sample().
// The coordinates (a double division per pixel) are evaluated once per tile and kept in registers
// for the frames_wg frames the workgroup walks through.
// uint16 frames (ST = uint16_t; bicubic and Lanczos4 at 1/32-px coordinates): OpenCV's 16U
// arithmetic of sampler.hpp::sample_u16_cv - float32 table weights, tap * (wy * wx) products added
// without fma -, the box clipped to the source, footprints on its border tap by tap (slow_u16).
#pragma once

#include "sampler.hpp"
#include "ring_remap.hpp"   // pk_fma_half / pk_mul_half, lds_read_b64, lds_wait_all

namespace ipa {

// Tile shapes (output pixels per workgroup of 256 threads): 64 x 32 wherever its source boxes fit,
// 32 x 32 and 32 x 16 for homographies that shrink parts of the picture (PerspectiveCorrection's
// uncorrect / distort, strong trapezoids: a 64 x 32 tile of the far side can span 180 x 96 source
// pixels).  TW = 64: lane = column, wave + 4 j = row; TW = 32: lanes 0-31 / 32-63 = two rows.
#ifndef IPA_TILE_SLOW_INSIDE
#define IPA_TILE_SLOW_INSIDE 1    // 0 (round 6, measured with IPA_TILE_RECOMPUTE = 1: slower, off): the rare tap-by-tap footprints of
                                  // float32 frames in a frame loop of their own behind the main one
#endif
#ifndef IPA_TILE_CUBIC_READS_FIRST
#define IPA_TILE_CUBIC_READS_FIRST 1   // 0: the taps of a pixel pair read and summed row by row
#endif
#ifndef IPA_TILE_PITCH_ODD
#define IPA_TILE_PITCH_ODD 0      // 1: the LDS box pitch among the odd candidates only (rounds 4 - 5)
#endif
#ifndef IPA_LZ_TABLE_STRIDE12
#define IPA_LZ_TABLE_STRIDE12 1   // 0: the Lanczos4 weight table with rows of 8 floats (rounds 4 - 5)
#endif
#ifndef IPA_TILE_RECOMPUTE
#define IPA_TILE_RECOMPUTE 0      // 1 (round 6, measured slower, off): the frame loop's scalar invariants recomputed per frame on
                                  // the scalar unit instead of hoisted and restored with v_readlane_b32 - see the frame loop
#endif
#ifndef IPA_TILE_CUBIC_PACKED
#define IPA_TILE_CUBIC_PACKED 1   // 0: the scalar bicubic loop of rounds 4 - 5 (same bits; tuning A/B builds)
#endif
constexpr int kWarpShapes = 3;
constexpr int kWarpTileWs[kWarpShapes] = {64, 32, 32}, kWarpTileHs[kWarpShapes] = {32, 32, 16};
constexpr int kWarpTileLdsBytes = 40960;  // the box of one tile (4 workgroups per CU at the most)

struct TileWarpArgs {
  char* dst;
  long dst_frame_elems, dpitch;
  const char* src;
  long src_frame_bytes;
  unsigned src_bytes, dst_bytes;   // of one frame (descriptor ranges)
  int sh, sw, spitch, dh, dw;
  int n_frames, frames_wg;
  int group_chunk;   // frame groups walked this many at a time (0: all groups of a tile together), as WaveParams::group_chunk
  int border, q5;
  float cubic_a;
  const float* lanczos;
  float cval;
  int tiles_x, tiles;
  int pitch, rows;   // the LDS box: floats per row (chosen against bank conflicts: tile_warp_pitch), rows
  // coordinate tables: pixels of the first frame group whose footprints the box did not hold are
  // counted here (the host cannot know a map's boxes: remap_impl.hpp reads the count back
  // without waiting and keeps the next call with this map away if they were many); else null
  unsigned* slow_count;
};

// first and last source index the footprints of coordinates in [lo, hi] can touch, clipped to what
// a footprint with at least one tap inside [0, n) reaches.  (1/32-px rounding moves a coordinate
// by up to 1/64: 0.02 covers it and the rounding of the corner arithmetic.)
// INSIDE: clipped to [0, n) instead - the uint16 kernel's fast path takes footprints that lie
// inside the source only.
template <int NT, bool INSIDE = false>
__host__ __device__ inline void tile_axis_box(double lo, double hi, int n, int& first, int& count) {
  const double lim = 1.0e6;
  lo = lo < -lim ? -lim : (lo > lim ? lim : lo);
  hi = hi < -lim ? -lim : (hi > lim ? lim : hi);
  int a = (int)floor(lo - 0.02) - (NT / 2 - 1);
  int b = (int)floor(hi + 0.02) + NT / 2;
  if (a < (INSIDE ? 0 : -(NT - 1))) a = INSIDE ? 0 : -(NT - 1);
  if (b > (INSIDE ? n - 1 : n + NT - 2)) b = INSIDE ? n - 1 : n + NT - 2;
  first = a;
  count = b - a + 1 > 0 ? b - a + 1 : 0;
}

// sample() of sampler.hpp tap by tap (rolled loops, a handful of registers): the same weights,
// products and sums in the same order - for the few footprints a tile's box does not hold
template <int INTERP, typename C>
__device__ __forceinline__ float tile_slow_sample(const SrcView& s, C sx, C sy, float cval, int lz_stride = 8) {
  constexpr int NT = ntaps<INTERP>::value;
  if (!(sx > (C)-kCoordLimit && sx < (C)kCoordLimit && sy > (C)-kCoordLimit && sy < (C)kCoordLimit)) {
    if (s.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cval;
    sx = sx < (C)-kCoordLimit ? (C)-kCoordLimit : (sx > (C)kCoordLimit ? (C)kCoordLimit : sx);
    sy = sy < (C)-kCoordLimit ? (C)-kCoordLimit : (sy > (C)kCoordLimit ? (C)kCoordLimit : sy);
  }
  int ix0, iy0;
  float wxs[4], wys[4];   // bilinear / bicubic weights (indexed with constants only)
  const float *wxr = s.lanczos, *wyr = s.lanczos;   // Lanczos4: rows of the table (the LDS copy)
  if constexpr (INTERP == kLanczos4) {
    const int qx = (int)ipa_rint(sx * (C)32), qy = (int)ipa_rint(sy * (C)32);
    ix0 = (qx >> 5) - 3;
    iy0 = (qy >> 5) - 3;
    wxr += (qx & 31) * lz_stride;
    wyr += (qy & 31) * lz_stride;
  } else {
    float wx[NT], wy[NT];
    axis_split<INTERP, float, C>(s, sx, ix0, wx);
    axis_split<INTERP, float, C>(s, sy, iy0, wy);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      wxs[k] = wx[k % NT];
      wys[k] = wy[k % NT];
    }
  }
  if (s.border == IPA_BORDER_CONSTANT &&
      (ix0 >= s.w || ix0 + NT <= 0 || iy0 >= s.h || iy0 + NT <= 0))
    return cval;
  auto weight = [](const float* row, const float (&w)[4], int k) {
    if constexpr (INTERP == kLanczos4) return row[k];
    else return k == 0 ? w[0] : (k == 1 ? w[1] : (k == 2 ? w[2] : w[3]));
  };
  float out = 0.f;
#pragma unroll 1
  for (int r = 0; r < NT; r++) {
    const int yy = resolve_idx(iy0 + r, s.h, s.border);
    float rs = 0.f;
#pragma unroll 1
    for (int c = 0; c < NT; c++) {
      const int xx = resolve_idx(ix0 + c, s.w, s.border);
      const float v = (yy < 0 || xx < 0)
                          ? cval
                          : u2f(__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, (yy * s.pitch + xx) << 2, 0, 0));
      const float w = weight(wxr, wxs, c);
      rs = c == 0 ? w * v : ipa_fma(w, v, rs);
    }
    const float w = weight(wyr, wys, r);
    out = r == 0 ? w * rs : ipa_fma(w, rs, out);
  }
  return out;
}

template <int INTERP, typename ST, int TW, int TH, typename Coord = HomographyCoord>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(INTERP == kCubic || (INTERP == kLanczos4 && sizeof(ST) == 4) ? 4 : 3, 8)))
tile_warp_kernel(TileWarpArgs a, Coord coord) {
  // Coord: the homography (its tile boxes bounded by the tile's corners, the LDS sized by the
  // host's walk over the tiles) or a coordinate TABLE - cv2.remap's map pair
  // (camera/LensDistortion.py:323-326): the box of a tile is then the span of its own footprints
  // (a reduction over the workgroup), clamped to the fixed LDS the launch reserves
  using C = typename Coord::coord_t;
  constexpr bool kHom = std::is_same<Coord, HomographyCoord>::value;
  static_assert(kHom || !std::is_same<ST, uint16_t>::value, "coordinate tables: float32 frames");
  static_assert((TW == 64 || TW == 32) && (TW * TH) % 256 == 0, "tile shapes of tile_warp.hpp");
  constexpr int kWarpTileW = TW, kWarpTileH = TH;
  constexpr int kWarpTilePx = TW * TH / 256;   // pixels per thread
  constexpr int kRowsPass = 256 / TW;          // output rows the workgroup covers per pixel index j
  constexpr int NT = ntaps<INTERP>::value;
  constexpr bool kLz = INTERP == kLanczos4;
  // uint16 frames: OpenCV's 16U arithmetic (float32 table weights, no fma).  The box is clipped to
  // the source: footprints inside it take the fast path, the ones on the border slow_u16 below
  constexpr bool kU16 = std::is_same<ST, uint16_t>::value;
  static_assert(!kU16 || INTERP == kCubic || INTERP == kLanczos4, "uint16: bicubic and Lanczos4");
  constexpr int kEsh = kU16 ? 1 : 2;   // log2 of the element size
  // LDS index of box cell (row r, column c).  Lanczos4 keeps the rows in interleaved pairs - pair
  // p = {row 2p, row 2p + 1}, column c of both at float 2c - so that ONE aligned ds_read_b64
  // (256 B/clk, twice ds_read2_b32) fetches two tap rows of a column: a footprint is 5 pairs x 8
  // columns = 40 reads instead of 64 dwords in 32 (the scheme of ring_remap.hpp's Lanczos4 ring)
  auto cell = [&](int r, int c) {
    if constexpr (kLz) return __mul24(r >> 1, 2 * a.pitch) + 2 * c + (r & 1);
    else return __mul24(r, a.pitch) + c;
  };
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  __shared__ double corner[8];   // (homography: the tile's corners; tables: the waves' footprint spans)
  // float32 Lanczos4: the 32 x 8 weight table with its rows 12 floats apart (round 6).  A sample reads four 16-byte
  // pieces of it - two rows picked by the lanes' own fractions -; at 8 floats per row the rows r, r + 8, r + 16, r + 24
  // start on the same bank and the 16 lanes of a ds_read_b128 group pile up 2 - 4 deep on them (a fifth of the
  // kernel's LDS cycles were bank conflicts, and its LDS arrays are busy 77 % of the launch); at 12 only r and r + 16 meet.
  // (uint16 frames, Lanczos4: the same table, the same reads; uint16 bicubic keeps its 4-float rows behind 256 floats)
  constexpr int kLzStride = (kLz && IPA_LZ_TABLE_STRIDE12) ? 12 : 8;
  __shared__ __attribute__((aligned(16))) float lz[kLz ? 32 * kLzStride : (kU16 ? 384 : 4)];
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (kLz || kU16) lz[(tid >> 3) * kLzStride + (tid & 7u)] = a.lanczos[tid];
  if constexpr (kU16 && !kLz)
    if (tid < 128u) lz[256 + tid] = a.lanczos[256 + tid];

  const unsigned groups = ((unsigned)a.n_frames + a.frames_wg - 1) / (unsigned)a.frames_wg;
  const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  unsigned grp = b % groups, tile = b / groups;
  if (a.group_chunk) {
    const unsigned gc = (unsigned)a.group_chunk, per = gc * (unsigned)a.tiles;
    const unsigned chunk = b / per, r = b - chunk * per;
    grp = chunk * gc + r % gc;
    tile = r / gc;
  }
  const int tyi = (int)(tile / (unsigned)a.tiles_x), txi = (int)tile - tyi * a.tiles_x;
  const int x0 = txi * kWarpTileW;
  int y0 = tyi * kWarpTileH;   // (not const: made opaque per frame below)
  const int x1 = x0 + kWarpTileW - 1 < a.dw ? x0 + kWarpTileW - 1 : a.dw - 1;
  const int y1 = y0 + kWarpTileH - 1 < a.dh ? y0 + kWarpTileH - 1 : a.dh - 1;
  int bx0 = 0, by0 = 0, bw = 0, bh = 0;
  if constexpr (kHom) {
    if (tid < 4u) {
      double sx, sy;
      coord.get((tid & 1u) ? x1 : x0, (tid & 2u) ? y1 : y0, sx, sy);
      corner[2 * tid] = sx;
      corner[2 * tid + 1] = sy;
    }
    __syncthreads();
    double lox = corner[0], hix = corner[0], loy = corner[1], hiy = corner[1];
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double cx = corner[2 * k], cy = corner[2 * k + 1];
      fin = fin && ipa_abs(cx) < 1.0e6 && ipa_abs(cy) < 1.0e6;   // false for NaN
      lox = cx < lox ? cx : lox; hix = cx > hix ? cx : hix;
      loy = cy < loy ? cy : loy; hiy = cy > hiy ? cy : hiy;
    }
    tile_axis_box<NT, kU16>(lox, hix, a.sw, bx0, bw);
    tile_axis_box<NT, kU16>(loy, hiy, a.sh, by0, bh);
    if (!fin) bw = bh = 0;
    bw = bw < a.pitch ? bw : a.pitch;
    bw = bw < 128 ? bw : 128;   // (the fill covers columns 0 .. 127; tile_warp_pitch may pad the pitch past them)
    bh = bh < a.rows ? bh : a.rows;
    bx0 = __builtin_amdgcn_readfirstlane(bx0); by0 = __builtin_amdgcn_readfirstlane(by0);
    bw = __builtin_amdgcn_readfirstlane(bw); bh = __builtin_amdgcn_readfirstlane(bh);
  }

  // the tile's footprints, once: LDS index of the first tap (-1: through sample()), fractions
  SrcView s;
  s.h = a.sh; s.w = a.sw; s.pitch = a.spitch;
  s.border = a.border; s.q5 = a.q5; s.cubic_a = a.cubic_a;
  s.lanczos = lz;
  int ad[kWarpTilePx];
  float tx[kWarpTilePx], ty[kWarpTilePx];   // Lanczos4: the table rows' float offsets, as ints
  const int x = x0 + (int)(lane & (unsigned)(TW - 1));
  // row of pixel j of this thread: yl + kRowsPass j (TW = 64: wave-uniform)
  const int yl = TW == 64 ? (int)wave : 2 * (int)wave + (int)(lane >> 5);
  unsigned slow = 0, costly = 0;   // costly: a footprint sample() has to walk tap by tap
  // footprint of pixel j: first tap, fractions (or table rows); false for coordinates that are
  // not finite / far outside
  auto footprint = [&](int j, int& ix0, int& iy0) -> bool {
    const int y = y0 + yl + kRowsPass * j;
    C sx, sy;
    coord.get(x < a.dw ? x : a.dw - 1, y < a.dh ? y : a.dh - 1, sx, sy);
    const bool ok = ipa_abs(sx) < (C)kCoordLimit && ipa_abs(sy) < (C)kCoordLimit;
    if (!ok) sx = sy = (C)0;
    if constexpr (kLz) {
      const int qx = (int)ipa_rint(sx * (C)32), qy = (int)ipa_rint(sy * (C)32);
      ix0 = (qx >> 5) - 3;
      iy0 = (qy >> 5) - 3;
      tx[j] = __int_as_float(((qx & 31) * kLzStride) | (((qy & 31) * kLzStride) << 16));   // float offsets of both table rows
      ty[j] = 0.f;
    } else if constexpr (kU16) {   // bicubic: rows of the float32 table at 256, 1/32-px coordinates
      const int qx = (int)ipa_rint(sx * (C)32), qy = (int)ipa_rint(sy * (C)32);
      ix0 = (qx >> 5) - 1;
      iy0 = (qy >> 5) - 1;
      tx[j] = __int_as_float((256 + ((qx & 31) << 2)) | ((256 + ((qy & 31) << 2)) << 16));
      ty[j] = 0.f;
    } else {
      axis_frac<INTERP, float, C>(s, sx, ix0, tx[j]);
      axis_frac<INTERP, float, C>(s, sy, iy0, ty[j]);
    }
    return ok;
  };
  auto place = [&](int j, bool ok, int ix0, int iy0) {
    const int cx = ix0 - bx0, cy = iy0 - by0;
    const bool in = ok && cx >= 0 && cy >= 0 && cx + NT <= bw && cy + NT <= bh;
    // (Lanczos4: first pair and column of the footprint in front, its row parity in bit 0)
    ad[j] = !in ? -1 : (kLz ? ((__mul24(cy >> 1, 2 * a.pitch) + 2 * cx) << 1) | (cy & 1) : __mul24(cy, a.pitch) + cx);
    slow |= in ? 0u : 1u << j;
  };
  if constexpr (kHom) {
#pragma unroll
    for (int j = 0; j < kWarpTilePx; j++) {
      int ix0, iy0;
      const bool ok = footprint(j, ix0, iy0);
      place(j, ok, ix0, iy0);
      __builtin_amdgcn_sched_barrier(0);   // one pixel's double arithmetic at a time (registers)
    }
  } else {
    // coordinate table: all footprints first, their span over the workgroup = the box
    int fx[kWarpTilePx], fy[kWarpTilePx];
    unsigned okm = 0;
    int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
#pragma unroll
    for (int j = 0; j < kWarpTilePx; j++) {
      const bool ok = footprint(j, fx[j], fy[j]);
      okm |= ok ? 1u << j : 0u;
      // (footprints wholly outside the source do not stretch the box: sample() gives them the
      // border value or resolves them tap by tap)
      const bool use = ok && fx[j] > -NT && fx[j] < a.sw && fy[j] > -NT && fy[j] < a.sh;
      costly |= (use || (ok && a.border != IPA_BORDER_CONSTANT)) ? 1u << j : 0u;
      xmn = use && fx[j] < xmn ? fx[j] : xmn; xmx = use && fx[j] > xmx ? fx[j] : xmx;
      ymn = use && fy[j] < ymn ? fy[j] : ymn; ymx = use && fy[j] > ymx ? fy[j] : ymx;
    }
    wave_span(xmn, xmx, ymn, ymx);
    int* wbox = reinterpret_cast<int*>(corner);
    if (lane == 0) {
      wbox[4 * wave + 0] = xmn; wbox[4 * wave + 1] = xmx;
      wbox[4 * wave + 2] = ymn; wbox[4 * wave + 3] = ymx;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      xmn = wbox[4 * k] < xmn ? wbox[4 * k] : xmn; xmx = wbox[4 * k + 1] > xmx ? wbox[4 * k + 1] : xmx;
      ymn = wbox[4 * k + 2] < ymn ? wbox[4 * k + 2] : ymn; ymx = wbox[4 * k + 3] > ymx ? wbox[4 * k + 3] : ymx;
    }
    if (xmn <= xmx && ymn <= ymx) {
      bx0 = xmn; by0 = ymn;
      bw = xmx + NT - xmn; bh = ymx + NT - ymn;
      bw = bw < a.pitch ? bw : a.pitch;
      bw = bw < 128 ? bw : 128;
      bh = bh < a.rows ? bh : a.rows;
    }
    bx0 = __builtin_amdgcn_readfirstlane(bx0); by0 = __builtin_amdgcn_readfirstlane(by0);
    bw = __builtin_amdgcn_readfirstlane(bw); bh = __builtin_amdgcn_readfirstlane(bh);
#pragma unroll
    for (int j = 0; j < kWarpTilePx; j++) place(j, (okm >> j) & 1u, fx[j], fy[j]);
  }
  const bool inside = bx0 >= 0 && by0 >= 0 && bx0 + bw <= a.sw && by0 + bh <= a.sh;
  if constexpr (!kHom) {
    if (a.slow_count && grp == 0u) {
      // (pixels outside the output do not count, nor the ones sample() answers with the border
      // value at once: footprints wholly outside the source in the constant border mode)
      unsigned cnt = 0;
#pragma unroll
      for (int j = 0; j < kWarpTilePx; j++)
        cnt += ((slow & costly) >> j) & 1u && x < a.dw && y0 + yl + kRowsPass * j < a.dh ? 1u : 0u;
      // wave sum by ballot per bit would be 8 ballots; one atomic per lane that has any is rare
      if (cnt) atomicAdd(a.slow_count, cnt);
    }
  }

  ST* dst0 = reinterpret_cast<ST*>(a.dst);
  // one source element at byte offset voffset + soffset of a frame's descriptor, as float
  auto load_px = [&](const __amdgpu_buffer_rsrc_t& rs, int vo, int so) -> float {
    if constexpr (kU16) return (float)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, vo, so, 0);
    else return u2f(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
  };
  const int cells = bw * bh;
  const float inv_bw = 1.0f / (float)(bw > 0 ? bw : 1);
  const unsigned f0 = grp * (unsigned)a.frames_wg;
  const unsigned f1 = f0 + (unsigned)a.frames_wg < (unsigned)a.n_frames ? f0 + (unsigned)a.frames_wg
                                                                        : (unsigned)a.n_frames;

  // The box of a tile: wave w takes rows w, w + 4, ..., lane l column l - the row offset is a
  // scalar, nothing but the LDS address is computed per load; the columns past 64 go 64 >> esh
  // rows per load (lane = row : column, esh bits of column).  The first kRowsFly rows of a wave
  // and its first 4 loads of the far columns are REQUESTED while the previous frame is being
  // sampled and written to LDS after it (box_issue / box_commit): the ~1 us of an HBM round trip
  // is then behind the samples and stores of a frame, not in front of them.
  // A box INSIDE the source, or ANY box in the constant border mode (`direct`): there the rows and
  // columns that are not the source's - the rim of a rotated picture, of an alpha = 1
  // undistortion - load from a clamped place and are written as the border value.  The other
  // border modes on the rim go cell by cell (below).
  // (a box of 35 / 37 / 41 rows at no rotation: 9 / 10 / 11 rows per wave.  Lanczos4 on float32 frames
  // keeps exactly its 11: with a twelfth prefetch register the 64 x 32 instantiation spilled one
  // register to scratch memory under its 128-register cap - tools/check_pipe_asm.py)
  constexpr int kRowsFly = INTERP == kCubic ? 10 : (kLz && !kU16 ? 11 : 12);
  // (bicubic only: in one process, 16 x 4K at 15 degrees, direct against cell by cell: bicubic
  // 0.392 / 0.414 ms, but bilinear 0.371 / 0.310 and Lanczos4 0.869 / 0.841 - the larger loop body
  // costs those two more than their rim tiles gain; profiles/r04_micro.txt)
  constexpr bool kDirectRim = INTERP == kCubic;
  const bool direct = inside || (kDirectRim && a.border == IPA_BORDER_CONSTANT);
  const bool c0 = (int)lane < bw;                                               // a cell of the box
  const bool c0v = c0 && (unsigned)(bx0 + (int)lane) < (unsigned)a.sw;          // ... inside the source
  const int voff = c0v ? (bx0 + (int)lane) << kEsh : 0;
  const int e = bw - 64;
  const int esh = e <= 4 ? 2 : (e <= 8 ? 3 : (e <= 16 ? 4 : (e <= 32 ? 5 : 6)));
  const int rstep = 64 >> esh;   // rows per load of the far columns
  const int lr = (int)lane >> esh, lc = 64 + ((int)lane & ((1 << esh) - 1));
  const bool cl = lc < bw;
  const bool clv = cl && (unsigned)(bx0 + lc) < (unsigned)a.sw;
  const int lcoff = clv ? bx0 + lc : 0;
  // byte offset of box row r in a frame (scalar; rows that are not the source's: row 0, not used)
  auto row_off = [&](int r) -> int {
    const int yy = by0 + r;
    return (unsigned)yy < (unsigned)a.sh ? __mul24(yy, a.spitch) << kEsh : 0;
  };
  auto row_live = [&](int r) -> bool { return (unsigned)(by0 + r) < (unsigned)a.sh; };
  float v0[kRowsFly], v1[4];
  // (addresses stepped from one value the compiler cannot carry across the frame loop: held as
  // loop invariants they are 40 registers)
  // (RIM: the box has rows / columns that are not the source's; a box inside the source skips the
  // checks - they cost the bilinear kernel 15 %)
  auto box_issue = [&](auto rim_, const __amdgpu_buffer_rsrc_t& rs) {
    constexpr bool RIM = decltype(rim_)::value;
#pragma unroll
    for (int u = 0; u < kRowsFly; u++) {
      const int r = (int)wave + 4 * u < bh ? (int)wave + 4 * u : bh - 1;
      v0[u] = load_px(rs, voff, RIM ? row_off(r) : __mul24(by0 + r, a.spitch) << kEsh);
    }
    if (e > 0) {
      int r = (int)wave * rstep + lr;
      asm volatile("" : "+v"(r));
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool live = clv && r < bh && (!RIM || (unsigned)(by0 + r) < (unsigned)a.sh);
        v1[u] = load_px(rs, live ? (__mul24(by0 + r, a.spitch) + lcoff) << kEsh : 0, 0);
        r += 4 * rstep;
      }
    }
  };
  auto box_commit = [&](auto rim_) {
    constexpr bool RIM = decltype(rim_)::value;
    int la = cell((int)wave, (int)lane);   // rows wave + 4 u: 4 rows = 2 pairs further each
    asm volatile("" : "+v"(la));
    const int lstep = 4 * a.pitch;
#pragma unroll
    for (int u = 0; u < kRowsFly; u++) {
      const int r = (int)wave + 4 * u;
      if (r < bh && c0) tile_lds[la] = (!RIM || (c0v && row_live(r))) ? v0[u] : a.cval;
      la += lstep;
    }
    if (e > 0) {
      int r = (int)wave * rstep + lr;
      asm volatile("" : "+v"(r));
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (cl && r < bh) tile_lds[cell(r, lc)] = (!RIM || (clv && (unsigned)(by0 + r) < (unsigned)a.sh)) ? v1[u] : a.cval;
        r += 4 * rstep;
      }
    }
  };
  // (boxes taller than 4 kRowsFly rows - Lanczos4 under a rotation of 30 degrees and more - or
  // with many far columns: the rest, read when the frame's turn has come)
  auto box_rest = [&](auto rim_, const __amdgpu_buffer_rsrc_t& rs) {
    constexpr bool RIM = decltype(rim_)::value;
    constexpr int kMore = 8;   // rows in flight
#pragma unroll 1
    for (int r0 = (int)wave + 4 * kRowsFly; r0 < bh; r0 += 4 * kMore) {
      float t[kMore];
#pragma unroll
      for (int u = 0; u < kMore; u++) {
        const int r = r0 + 4 * u < bh ? r0 + 4 * u : bh - 1;
        t[u] = load_px(rs, voff, RIM ? row_off(r) : __mul24(by0 + r, a.spitch) << kEsh);
      }
#pragma unroll
      for (int u = 0; u < kMore; u++)
        if (r0 + 4 * u < bh && c0) tile_lds[cell(r0 + 4 * u, (int)lane)] = (!RIM || (c0v && row_live(r0 + 4 * u))) ? t[u] : a.cval;
    }
    if (e > 0) {
#pragma unroll 1
      for (int r0 = ((int)wave + 16) * rstep; r0 < bh; r0 += 4 * rstep) {
        const int r = r0 + lr;
        const bool live = clv && r < bh && (!RIM || (unsigned)(by0 + r) < (unsigned)a.sh);
        const float t = load_px(rs, live ? (__mul24(by0 + r, a.spitch) + lcoff) << kEsh : 0, 0);
        if (cl && r < bh) tile_lds[cell(r, lc)] = live ? t : a.cval;
      }
    }
  };
  // uint16: a footprint the box does not hold whole - on the border of the source, outside it, not
  // finite - in the arithmetic of sampler.hpp::sample_u16_cv, tap by tap (rolled loops, a handful of
  // registers); the taps that exist come from the box where it has them (it holds what lies inside
  // the source of every footprint of the tile), from memory otherwise (wrap / reflect borders)
  auto slow_u16 = [&](double sx, double sy) -> uint16_t {
#pragma clang fp contract(off)
    constexpr int ks = NT;
    const double rv = rint((double)a.cval);
    const uint16_t cv16 = (uint16_t)(rv > 0 ? (rv < 65535 ? rv : 65535) : 0);
    if (!(sx > (double)-kCoordLimit && sx < (double)kCoordLimit && sy > (double)-kCoordLimit &&
          sy < (double)kCoordLimit)) {
      if (a.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cv16;
      sx = sx < (double)-kCoordLimit ? (double)-kCoordLimit : (sx > (double)kCoordLimit ? (double)kCoordLimit : sx);
      sy = sy < (double)-kCoordLimit ? (double)-kCoordLimit : (sy > (double)kCoordLimit ? (double)kCoordLimit : sy);
    }
    const int qx = (int)ipa_rint(sx * 32.0), qy = (int)ipa_rint(sy * 32.0);
    const int ix0 = (qx >> 5) - (ks / 2 - 1), iy0 = (qy >> 5) - (ks / 2 - 1);
    if (a.border == IPA_BORDER_CONSTANT && (ix0 >= a.sw || ix0 + ks <= 0 || iy0 >= a.sh || iy0 + ks <= 0))
      return cv16;
    const float* wx = lz + (INTERP == kCubic ? 256 : 0) + (qx & 31) * (INTERP == kCubic ? ks : kLzStride);
    const float* wy = lz + (INTERP == kCubic ? 256 : 0) + (qy & 31) * (INTERP == kCubic ? ks : kLzStride);
    const float cv = (float)cv16;
    const bool whole = ix0 >= 0 && iy0 >= 0 && ix0 + ks <= a.sw && iy0 + ks <= a.sh;
    auto tap = [&](int yy, int xx) -> float {
      const int r = yy - by0, c = xx - bx0;
      if ((unsigned)r < (unsigned)bh && (unsigned)c < (unsigned)bw) return tile_lds[cell(r, c)];
      return (float)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(s.rsrc, (__mul24(yy, a.spitch) + xx) << 1, 0, 0);
    };
    float sum;
    if (whole) {
      sum = 0.f;
#pragma unroll 1
      for (int r = 0; r < ks; r++) {
        float rs = 0.f;
#pragma unroll 1
        for (int c = 0; c < ks; c++) {
          const float pr = tap(iy0 + r, ix0 + c) * (wy[r] * wx[c]);
          if constexpr (INTERP == kCubic) sum = (r == 0 && c == 0) ? pr : sum + pr;
          else rs = c == 0 ? pr : rs + pr;
        }
        if constexpr (INTERP != kCubic) sum = sum + rs;
      }
    } else {
      sum = cv;
#pragma unroll 1
      for (int r = 0; r < ks; r++) {
        const int yy = resolve_idx(iy0 + r, a.sh, a.border);
        if (yy < 0) continue;
#pragma unroll 1
        for (int c = 0; c < ks; c++) {
          const int xx = resolve_idx(ix0 + c, a.sw, a.border);
          if (xx >= 0) sum = sum + (tap(yy, xx) - cv) * (wy[r] * wx[c]);
        }
      }
    }
    float q = rintf(sum);
    q = q > 0.f ? q : 0.f;
    q = q < 65535.f ? q : 65535.f;
    return (uint16_t)q;
  };
  s.rsrc = make_rsrc(a.src + (long)f0 * a.src_frame_bytes, a.src_bytes);
  using RimT = std::true_type;
  using InT = std::false_type;
  // (ONE issue statement for both kinds of box: two alternative ones would meet at a join, and
  // the compiler drains the loads there)
  // (uint16 Lanczos4 - ~200 vector instructions per sample, every register taken - loads its box
  // when the frame's turn has come: without the 16 prefetch registers 0.862 -> 0.838 ms per 16 x 4K)
  constexpr bool kPrefetch = !(kU16 && kLz);
  if (kPrefetch && direct && f0 < f1) box_issue(RimT{}, s.rsrc);
#pragma unroll 1
  for (unsigned f = f0; f < f1; f++) {
#if IPA_TILE_RECOMPUTE
    // Round 6: everything the frame loop derives from the box's first row, its height and the tile's first row -
    // the 10 + 4 row offsets of the box loads, the per-row store predicates of the fill, the 8 row offsets of the
    // result stores - is loop-invariant, so the compiler forms it once, runs out of scalar registers (102) and
    // restores ~70 of those values per frame with v_readlane_b32: VECTOR instructions in a kernel whose SIMDs issue
    // 70 - 88 % of the launch.  Opaque copies make it recompute them on the scalar unit instead - MEASURED SLOWER
    // (16 x 4K: bicubic 0.303 -> 0.324 ms, under 15 degrees 0.409 -> 0.468, bilinear 0.304 -> 0.381; Lanczos4 level):
    // the recomputed multiplies and selects sit in front of the loads that need them, the restores did not.  Off.
    asm volatile("" : "+s"(by0), "+s"(bh), "+s"(y0));
#endif
    s.rsrc = make_rsrc(a.src + (long)f * a.src_frame_bytes, a.src_bytes);
    __syncthreads();   // the previous frame's taps are read (first pass: the Lanczos table is written)
    // 1. the box
    if (inside) {
      if (!kPrefetch) box_issue(RimT{}, s.rsrc);
      box_commit(InT{});
      box_rest(InT{}, s.rsrc);
    } else if (kDirectRim && direct) {
      box_commit(RimT{});
      box_rest(RimT{}, s.rsrc);
    } else {
      // on the rim of the source in the other border modes: cell by cell (i = row i / bw, column
      // i % bw) through the border mode
#pragma unroll 1
      for (int i = (int)tid; i < cells; i += 256) {
        const int row = (int)(((float)i + 0.5f) * inv_bw), col = i - row * bw;
        const int yy = resolve_idx(by0 + row, a.sh, a.border);
        const int xx = resolve_idx(bx0 + col, a.sw, a.border);
        const bool live = yy >= 0 && xx >= 0;
        const int off = live ? (__mul24(yy, a.spitch) + xx) << kEsh : 0;
        const float t = load_px(s.rsrc, off, 0);
        tile_lds[cell(row, col)] = live ? t : a.cval;
      }
    }
    __syncthreads();
    if (f + 1 < f1) {
      const __amdgpu_buffer_rsrc_t nrs = make_rsrc(a.src + (long)(f + 1) * a.src_frame_bytes, a.src_bytes);
      if (kPrefetch && direct) box_issue(RimT{}, nrs);
    }
    // (the weights are formed anew for every frame: kept across the frame loop they are 8 registers
    // per pixel the compiler would hold - bicubic 178 registers, 2 workgroups per CU)
    if constexpr (INTERP != kLinear) {   // (Lanczos4: the table rows read anew - 16 registers per pixel)
#pragma unroll
      for (int j = 0; j < kWarpTilePx; j++) asm volatile("" : "+v"(tx[j]), "+v"(ty[j]));
    }
    // (... and what derives from a footprint's address - LDS byte addresses, the row parity)
#pragma unroll
    for (int j = 0; j < kWarpTilePx; j++) asm volatile("" : "+v"(ad[j]));
    // 2. the samples, kGroup at a time (their taps in flight together)
    const __amdgpu_buffer_rsrc_t drs = make_rsrc(dst0 + (long)f * a.dst_frame_elems, a.dst_bytes);
    // (uint16: `o` is the float32 sum; cv::saturate_cast<ushort> = round half to even, clamp)
    auto store_px = [&](float o, int y) {
      // (TW = 64: the row is wave-uniform and its offset a scalar; TW = 32: two rows per wave)
      const int so = TW == 64 ? (int)((long)y * a.dpitch) << kEsh : 0;
      const int vo = TW == 64 ? x << kEsh : (int)((long)y * a.dpitch + x) << kEsh;
      if constexpr (kU16) {
        float q = rintf(o);
        q = q > 0.f ? q : 0.f;   // NaN -> 0
        q = q < 65535.f ? q : 65535.f;
        __builtin_amdgcn_raw_buffer_store_b16((short)(unsigned short)q, drs, vo, so, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), drs, vo, so, 0);
      }
    };
    constexpr int kGroup = INTERP == kLinear ? 4 : (INTERP == kCubic ? 2 : 1);
    if constexpr (kLz) {
      const int lbase = lds_address(tile_lds);
      const int pstep = a.pitch << 3;   // bytes from pair to pair
#pragma unroll
      for (int j = 0; j < kWarpTilePx; j++) {
        const int y = y0 + yl + kRowsPass * j;
        const float4* rx = reinterpret_cast<const float4*>(lz + (__float_as_int(tx[j]) & 0xffff));
        const float4* ry = reinterpret_cast<const float4*>(lz + (__float_as_int(tx[j]) >> 16));
        const float4 a0 = rx[0], a1 = rx[1], b0 = ry[0], b1 = ry[1];
        const v2f wp[4] = {v2f{a0.x, a0.y}, v2f{a0.z, a0.w}, v2f{a1.x, a1.y}, v2f{a1.z, a1.w}};
        const float uy[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const bool odd = (ad[j] & 1) != 0;
        int ra = lbase + ((ad[j] < 0 ? 0 : ad[j] >> 1) << 2);
        // 5 pairs x 8 columns: rows 2 pb .. 2 pb + 9, of which the sample uses 8 from row `odd` on.
        // A pair at a time (all 40 reads at once are 80 registers: 210 in all, 2 waves per SIMD),
        // their row sums added as they come to BOTH column sums - the one from the even and
        // the one from the odd row on; one select at the end.  (Selecting the eight row sums by
        // `odd` first is as many instructions, and the compiler turns that form into an array in
        // scratch memory indexed by `odd`.)
        float oe = 0.f, oo = 0.f;
        auto row_sum = [&](auto k_, float R) {   // row k of the 10: row k of the even, k - 1 of the odd footprint
          constexpr int k = decltype(k_)::value;
          if constexpr (kU16) {
#pragma clang fp contract(off)
            if constexpr (k < 8) oe = oe + R;         // sum = 0; sum = sum + row sum, top to bottom
            if constexpr (k >= 1 && k <= 8) oo = oo + R;
          } else {
            if constexpr (k == 0) oe = uy[0] * R;
            else if constexpr (k < 8) oe = ipa_fma(uy[k], R, oe);
            if constexpr (k == 1) oo = uy[0] * R;
            else if constexpr (k >= 2 && k <= 8) oo = ipa_fma(uy[k - 1], R, oo);
          }
        };
        // The pairs go through two register sets: pair p + 1 is in flight while pair p is summed
        // (counted s_waitcnt lgkmcnt(8): the 8 reads of the older pair have returned - LDS reads
        // return in order, anything else in flight only makes the wait stricter).
        auto issue = [&](v2f (&t)[8]) {
#ifdef IPA_DEBUG_LZ_HALF_READS   // measurement only (WRONG results): every other sample re-uses the taps in the registers
          if (j & 1) return;
#endif
          static_for<0, 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            t[c] = lds_read_b64<c * 8>(ra);
          });
          ra += pstep;
        };
        auto sum_pair = [&](auto pp_, v2f (&t)[8]) {
          constexpr int k0 = 2 * decltype(pp_)::value;   // rows k0, k0 + 1 of the 10
#pragma unroll
          for (int c = 0; c < 8; c++) asm volatile("" : "+v"(t[c]));
          v2f rs;
          if constexpr (kU16) {
#pragma clang fp contract(off)
            // tap * (wy[r] * wx[c]), the products of a row added left to right - two rows at a
            // time in the halves of packed multiplies and adds (each half rounded on its own).
            // The row weights against the rows of this pair: of the even or of the odd
            // footprint (0 on a row that is not the footprint's: that sum is not used)
            const float we0 = k0 < 8 ? uy[k0 % 8] : 0.f, wo0 = k0 >= 1 ? uy[(k0 + 7) % 8] : 0.f;
            const float we1 = k0 + 1 < 8 ? uy[(k0 + 1) % 8] : 0.f, wo1 = uy[k0 % 8];
            const v2f wyp = v2f{odd ? wo0 : we0, odd ? wo1 : we1};
#pragma unroll
            for (int c = 0; c < 8; c++) {
              const float wxc = c & 1 ? wp[c >> 1].y : wp[c >> 1].x;
              const v2f pr = t[c] * (wyp * v2f{wxc, wxc});
              rs = c == 0 ? pr : rs + pr;
            }
          } else {
            rs = pk_mul_half<0>(wp[0], t[0]);
#pragma unroll
            for (int c = 1; c < 8; c++) {
              if (c & 1) rs = pk_fma_half<1>(wp[c >> 1], t[c], rs);
              else rs = pk_fma_half<0>(wp[c >> 1], t[c], rs);
            }
          }
          row_sum(std::integral_constant<int, k0>{}, rs.x);
          row_sum(std::integral_constant<int, k0 + 1>{}, rs.y);
        };
        v2f ta[8], tb[8];
        issue(ta);
        issue(tb);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        sum_pair(std::integral_constant<int, 0>{}, ta);
        __builtin_amdgcn_sched_barrier(0);
        issue(ta);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        sum_pair(std::integral_constant<int, 1>{}, tb);
        __builtin_amdgcn_sched_barrier(0);
        issue(tb);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        sum_pair(std::integral_constant<int, 2>{}, ta);
        __builtin_amdgcn_sched_barrier(0);
        issue(ta);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        sum_pair(std::integral_constant<int, 3>{}, tb);
        __builtin_amdgcn_sched_barrier(0);
        lds_wait_all();
        sum_pair(std::integral_constant<int, 4>{}, ta);
        const float o = odd ? oo : oe;
        if (ad[j] >= 0 && x < a.dw && y < a.dh)
          store_px(o, y);
        __builtin_amdgcn_sched_barrier(0);   // one sample's taps in flight
      }
    } else if constexpr (INTERP == kCubic && !kU16 && IPA_TILE_CUBIC_PACKED != 0 && kWarpTilePx % 2 == 0) {
      // bicubic on float32 frames, two pixels of the thread at a time in the halves of packed instructions (round
      // 6): the Keys weights of both (cubic_weights<v2f>), then per tap row one packed multiply and three packed
      // fmas on (tap of pixel a, tap of pixel b) - 20 vector instructions for the 2 x 16 taps instead of 40, 14
      // for the weights instead of 28.  Every half is the scalar operation of sample() on the same operands in
      // the same order: the bits of the gather kernel (tests/test_gpu_tile_warp.py, tools/fuzz_tile_warp.py).
#pragma unroll
      for (int j0 = 0; j0 < kWarpTilePx; j0 += 2) {
        const int ya = y0 + yl + kRowsPass * j0, yb = ya + kRowsPass;
        v2f wx[4], wy[4];
        cubic_weights<v2f>(v2f{tx[j0], tx[j0 + 1]}, v2f{s.cubic_a, s.cubic_a}, wx);
        cubic_weights<v2f>(v2f{ty[j0], ty[j0 + 1]}, v2f{s.cubic_a, s.cubic_a}, wy);
        const float* tpa = tile_lds + (ad[j0] < 0 ? 0 : ad[j0]);
        const float* tpb = tile_lds + (ad[j0 + 1] < 0 ? 0 : ad[j0 + 1]);
        v2f o = v2f{0.f, 0.f};
        // (volatile: one ds_read_b32 per tap, each straight into its half of a register pair - merged into
        // ds_read2_b32 the taps of ONE pixel share a pair and 16 moves per pixel pair put them apart again.
        // All 32 reads of the pixel pair first, then the sums: read and used row by row, only 8 reads were in flight
        // and their latency stood in front of every tap row - the kernel neither issued (0.70) nor kept its LDS
        // arrays busy (0.47))
        typedef const volatile __attribute__((address_space(3))) float* lds_vf;
#if IPA_TILE_CUBIC_READS_FIRST
        v2f tp[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          lds_vf tra = (lds_vf)(tpa + r * a.pitch);
          lds_vf trb = (lds_vf)(tpb + r * a.pitch);
#pragma unroll
          for (int c = 0; c < 4; c++) tp[r][c] = v2f{tra[c], trb[c]};
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          v2f rs = wx[0] * tp[r][0];
#pragma unroll
          for (int c = 1; c < 4; c++) rs = ipa_fma(wx[c], tp[r][c], rs);
          o = r == 0 ? wy[0] * rs : ipa_fma(wy[r], rs, o);
        }
#else
#pragma unroll
        for (int r = 0; r < 4; r++) {
          lds_vf tra = (lds_vf)(tpa + r * a.pitch);
          lds_vf trb = (lds_vf)(tpb + r * a.pitch);
          v2f rs = wx[0] * v2f{tra[0], trb[0]};
#pragma unroll
          for (int c = 1; c < 4; c++) rs = ipa_fma(wx[c], v2f{tra[c], trb[c]}, rs);
          o = r == 0 ? wy[0] * rs : ipa_fma(wy[r], rs, o);
        }
#endif
        if (ad[j0] >= 0 && x < a.dw && ya < a.dh) store_px(o.x, ya);
        if (ad[j0 + 1] >= 0 && x < a.dw && yb < a.dh) store_px(o.y, yb);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int j0 = 0; j0 < kWarpTilePx; j0 += kGroup) {
#pragma unroll
      for (int j = j0; j < j0 + kGroup; j++) {
        const int y = y0 + yl + kRowsPass * j;
        float wx[NT], wy[NT];
        const float* tp = tile_lds + (ad[j] < 0 ? 0 : ad[j]);
        float o = 0.f;
        if constexpr (kU16) {
#pragma clang fp contract(off)
          // bicubic on uint16 frames: the rows of the float32 table, the 16 products tap *
          // (wy[r] * wx[c]) added left to right in row-major order
          const float4 p4 = *reinterpret_cast<const float4*>(lz + (__float_as_int(tx[j]) & 0xffff));
          const float4 q4 = *reinterpret_cast<const float4*>(lz + (__float_as_int(tx[j]) >> 16));
          wx[0] = p4.x; wx[1] = p4.y; wx[2 % NT] = p4.z; wx[3 % NT] = p4.w;
          wy[0] = q4.x; wy[1] = q4.y; wy[2 % NT] = q4.z; wy[3 % NT] = q4.w;
#pragma unroll
          for (int r = 0; r < NT; r++) {
            const float* tr = tp + r * a.pitch;
#pragma unroll
            for (int c = 0; c < NT; c++) {
              const float pr = tr[c] * (wy[r] * wx[c]);
              o = (r == 0 && c == 0) ? pr : o + pr;
            }
          }
        } else {
          weights_from_frac<INTERP, float>(s, tx[j], wx);
          weights_from_frac<INTERP, float>(s, ty[j], wy);
#pragma unroll
          for (int r = 0; r < NT; r++) {
            const float* tr = tp + r * a.pitch;
            float rs = wx[0] * tr[0];
#pragma unroll
            for (int c = 1; c < NT; c++) rs = ipa_fma(wx[c], tr[c], rs);
            o = r == 0 ? wy[0] * rs : ipa_fma(wy[r], rs, o);
          }
        }
        if (ad[j] >= 0 && x < a.dw && y < a.dh)
          store_px(o, y);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    // 3. rare: footprints the box does not hold, through the gather kernel's sample().  uint16 frames here (their
    //    border footprints take what the frame's box holds from LDS); float32 frames in a frame loop of their own
    //    below, so that the coordinate source and the sampler's state are not live through THIS loop (round 6: 134 of
    //    the bicubic kernel's 747 vector instructions per frame were v_readlane_b32 restores of spilled scalars)
    if constexpr (kU16 || IPA_TILE_SLOW_INSIDE != 0) {
    if (slow) {
#pragma unroll 1
      for (int j = 0; j < kWarpTilePx; j++) {
        if (!((slow >> j) & 1u)) continue;
        const int y = y0 + yl + kRowsPass * j;
        if (x >= a.dw || y >= a.dh) continue;
        C sx, sy;
        coord.get(x, y, sx, sy);
        if constexpr (kU16)
          __builtin_amdgcn_raw_buffer_store_b16((short)slow_u16(sx, sy), drs,
                                                (int)((long)y * a.dpitch + x) << kEsh, 0, 0);
        else
          store_px(tile_slow_sample<INTERP, C>(s, sx, sy, a.cval, kLzStride), y);
      }
    }
    }
  }
  if constexpr (!kU16 && IPA_TILE_SLOW_INSIDE == 0) {
    if (slow) {
#pragma unroll 1
      for (unsigned f = f0; f < f1; f++) {
        s.rsrc = make_rsrc(a.src + (long)f * a.src_frame_bytes, a.src_bytes);
        const __amdgpu_buffer_rsrc_t drs = make_rsrc(dst0 + (long)f * a.dst_frame_elems, a.dst_bytes);
#pragma unroll 1
        for (int j = 0; j < kWarpTilePx; j++) {
          if (!((slow >> j) & 1u)) continue;
          const int y = y0 + yl + kRowsPass * j;
          if (x >= a.dw || y >= a.dh) continue;
          C sx, sy;
          coord.get(x, y, sx, sy);
          const float o = tile_slow_sample<INTERP, C>(s, sx, sy, a.cval, kLzStride);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), drs, (int)((long)y * a.dpitch + x) << kEsh, 0, 0);
        }
      }
    }
  }
}

// The LDS box a launch needs: the largest source box over all tiles.  Returns false when the
// warp is not one for this kernel (the plane's horizon crosses the picture, or a tile's box is
// wider than 128 columns or exceeds the LDS budget: strong minification).
// LDS bytes of a box (Lanczos4: whole row pairs, and the fifth pair of a footprint in the last rows)
template <int NT>
static inline long tile_warp_lds_bytes(int pitch, int rows) {
  return NT == 8 ? (long)((rows + 3) / 2) * 2 * pitch * 4 : (long)pitch * rows * 4;
}

template <int NT, bool INSIDE = false>
static inline bool tile_warp_box(const double* m, int dh, int dw, int sh, int sw, int kWarpTileW,
                                 int kWarpTileH, int* pitch, int* rows) {
  // w = m6 u + m7 v + m8 keeps one sign over the picture when it does at the four corners
  const double wc[4] = {m[8], m[6] * (dw - 1) + m[8], m[7] * (dh - 1) + m[8],
                        m[6] * (dw - 1) + m[7] * (dh - 1) + m[8]};
  for (int k = 0; k < 4; k++)
    if (!(wc[k] * wc[0] > 0.0) || !(fabs(wc[k]) > 1e-12)) return false;
  auto at = [&](int u, int v, double& sx, double& sy) {
    const double du = u, dv = v;
    const double X = m[0] * du + m[1] * dv + m[2], Y = m[3] * du + m[4] * dv + m[5];
    const double W = m[6] * du + m[7] * dv + m[8], iw = 1.0 / W;
    sx = X * iw;
    sy = Y * iw;
  };
  int mw = 0, mh = 0;
  for (int y0 = 0; y0 < dh; y0 += kWarpTileH)
    for (int x0 = 0; x0 < dw; x0 += kWarpTileW) {
      const int x1 = x0 + kWarpTileW - 1 < dw ? x0 + kWarpTileW - 1 : dw - 1;
      const int y1 = y0 + kWarpTileH - 1 < dh ? y0 + kWarpTileH - 1 : dh - 1;
      double lox = 0, hix = 0, loy = 0, hiy = 0;
      for (int k = 0; k < 4; k++) {
        double cx, cy;
        at((k & 1) ? x1 : x0, (k & 2) ? y1 : y0, cx, cy);
        if (!(fabs(cx) < 1.0e6) || !(fabs(cy) < 1.0e6)) return false;
        if (k == 0) { lox = hix = cx; loy = hiy = cy; }
        lox = cx < lox ? cx : lox; hix = cx > hix ? cx : hix;
        loy = cy < loy ? cy : loy; hiy = cy > hiy ? cy : hiy;
      }
      int f, bw, bh;
      tile_axis_box<NT, INSIDE>(lox, hix, sw, f, bw);
      tile_axis_box<NT, INSIDE>(loy, hiy, sh, f, bh);
      mw = bw > mw ? bw : mw;
      mh = bh > mh ? bh : mh;
    }
  if (mw < NT || mh < NT) return false;   // nothing of the source in sight: the gather kernel's cval fill
  if (mw > 128) return false;   // the fill reads columns 0-63 and 64-127 of a box
  mw |= 1;
  if (tile_warp_lds_bytes<NT>(mw, mh) > kWarpTileLdsBytes) return false;
  *pitch = mw;
  *rows = mh;
  return true;
}

// What decides between this kernel and the row-walking ones (remap_impl.hpp), from the homography:
//   drift  rows of the source an output row crosses per pixel, |d sy / d u| (largest of 9 probes);
//   step   source pixels per output pixel along either axis (> 1: the picture shrinks);
//   fetch  cells of the largest box per pixel of a tile (what a tile reads over what it writes).
static inline void tile_warp_measure(const double* m, int dh, int dw, int pitch, int rows,
                                     int kWarpTileW, int kWarpTileH, double* drift, double* step,
                                     double* fetch) {
  auto at = [&](double u, double v, double& sx, double& sy) {
    const double W = m[6] * u + m[7] * v + m[8], iw = W != 0.0 ? 1.0 / W : 0.0;
    sx = (m[0] * u + m[1] * v + m[2]) * iw;
    sy = (m[3] * u + m[4] * v + m[5]) * iw;
  };
  double d = 0, st = 0;
  for (int py = 0; py < 3; py++)
    for (int px = 0; px < 3; px++) {
      const double u = (dw - 2) * 0.5 * px, v = (dh - 2) * 0.5 * py;
      double x0, y0, x1, y1, x2, y2;
      at(u, v, x0, y0);
      at(u + 1, v, x1, y1);
      at(u, v + 1, x2, y2);
      const double a = hypot(x1 - x0, y1 - y0), b = hypot(x2 - x0, y2 - y0);
      d = fabs(y1 - y0) > d ? fabs(y1 - y0) : d;
      st = a > st ? a : st;
      st = b > st ? b : st;
    }
  *drift = d;
  *step = st;
  *fetch = (double)pitch * rows / (kWarpTileW * kWarpTileH);
}

// The LDS row pitch: the tap reads of a wave are 64 lanes walking along the output row, i.e. along
// a slanted line of the box - lane l at row floor(y + l dy), column floor(x + l dx) - and
// ds_read_b32 / ds_read2_b32 serve 32 lanes per cycle from 32 banks.  Some residues of the pitch
// mod 32 put every row step of that line back on the banks just used (Lanczos4, 16 x 4K: 1.0 ->
// 7.7 ms at 45 degrees with pitch = 31 mod 32, profiles/r04_micro.txt).  Counted here on the
// lines of a few output rows for the odd pitches from the box width up; the cheapest wins.
template <int NT>
static inline int tile_warp_pitch(const double* m, int dh, int dw, int min_pitch, int rows) {
  auto at = [&](double u, double v, double& sx, double& sy) {
    const double W = m[6] * u + m[7] * v + m[8], iw = W != 0.0 ? 1.0 / W : 0.0;
    sx = (m[0] * u + m[1] * v + m[2]) * iw;
    sy = (m[3] * u + m[4] * v + m[5]) * iw;
  };
  // (rounds 4 - 5 tried the odd pitches only.  A line that advances about one column per lane puts a 32-lane group on 32
  // consecutive columns, and then a pitch that is a MULTIPLE of 32 is conflict-free whatever rows the line crosses - the
  // bank is the column; round 6 counts every pitch: C5's bicubic warp under 7 degrees read its taps at 46 % conflict cycles)
  int best = min_pitch | 1;
  long best_cost = -1;
  for (int P = IPA_TILE_PITCH_ODD ? (min_pitch | 1) : min_pitch; P < (min_pitch | 1) + (IPA_TILE_PITCH_ODD ? 32 : 33);
       P += IPA_TILE_PITCH_ODD ? 2 : 1) {
    if (tile_warp_lds_bytes<NT>(P, rows) > kWarpTileLdsBytes) break;
    long cost = 0;
    for (int py = 0; py < 3; py++)
      for (int px = 0; px < 3; px++)
        for (int sub = 0; sub < 4; sub++) {
          const double u0 = (dw - 64) * (0.1 + 0.4 * px), v0 = (dh - 1) * (0.1 + 0.4 * py) + sub;
          double ox, oy;
          at(u0, v0, ox, oy);
          for (int g = 0; g < 64; g += 32) {
            long addr[32];
            int n = 0;
            for (int l = g; l < g + 32; l++) {
              double sx, sy;
              at(u0 + l, v0, sx, sy);
              // (Lanczos4: 8-byte reads of row pairs - the banks of a dword read at half the row)
              const long row = (long)floor(sy - oy + 0.37 * sub + 1024.0);
              const long aa = (NT == 8 ? row >> 1 : row) * P + (long)floor(sx - ox + 0.21 * sub + 1024.0);
              bool dup = false;
              for (int k = 0; k < n; k++) dup = dup || addr[k] == aa;
              if (!dup) addr[n++] = aa;
            }
            int cnt[32] = {0}, worst = 0;
            for (int k = 0; k < n; k++) {
              const int b = (int)(addr[k] & 31);
              worst = ++cnt[b] > worst ? cnt[b] : worst;
            }
            cost += worst;
          }
        }
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = P;
    }
  }
  return best;
}

// the launches, by tile shape (tile_warp_a.hip: 64 x 32; tile_warp_b.hip: 32 x 32, 32 x 16)
void tile_warp_run_a(hipStream_t stream, const TileWarpArgs& t, const HomographyCoord& coord, int interp,
                     bool u16, int shape, unsigned grid, size_t lds);
void tile_warp_run_b(hipStream_t stream, const TileWarpArgs& t, const HomographyCoord& coord, int interp,
                     bool u16, int shape, unsigned grid, size_t lds);

// coordinate tables (cv2.remap's map pair), float32 frames, 64 x 32 tiles (tile_warp_a.hip)
void tile_warp_run_map(hipStream_t stream, const TileWarpArgs& t, const MapCoord& coord, int interp,
                       unsigned grid, size_t lds);

// (for the two translation units above)
template <int TW, int TH>
static inline void tile_warp_run_shape(hipStream_t stream, const TileWarpArgs& t, const HomographyCoord& coord,
                                       int interp, bool u16, unsigned grid, size_t lds) {
  if (u16) {
    if (interp == kLanczos4)
      hipLaunchKernelGGL((tile_warp_kernel<kLanczos4, uint16_t, TW, TH>), dim3(grid), dim3(256), lds, stream, t, coord);
    else
      hipLaunchKernelGGL((tile_warp_kernel<kCubic, uint16_t, TW, TH>), dim3(grid), dim3(256), lds, stream, t, coord);
  } else if (interp == kLinear) {
    hipLaunchKernelGGL((tile_warp_kernel<kLinear, float, TW, TH>), dim3(grid), dim3(256), lds, stream, t, coord);
  } else if (interp == kLanczos4) {
    hipLaunchKernelGGL((tile_warp_kernel<kLanczos4, float, TW, TH>), dim3(grid), dim3(256), lds, stream, t, coord);
  } else {
    hipLaunchKernelGGL((tile_warp_kernel<kCubic, float, TW, TH>), dim3(grid), dim3(256), lds, stream, t, coord);
  }
}

}  // namespace ipa
