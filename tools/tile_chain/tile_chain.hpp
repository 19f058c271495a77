// tile_chain.hpp — cv2.warpPerspective followed by a separable K + K filter in ONE launch on the
// tile skeleton of tile_warp.hpp: PerspectiveCorrection.correct (camera/PerspectiveCorrection.py:
// 401-405) then scipy.ndimage.gaussian_filter (filters/fastFilter.py:42; the archetype of the chain
// is camera/lens/estimateSystematicErrorLensCorrection.py:199-207).
//
// The row-walking fused kernels (wave_sep.hpp) cover upright bilinear warps; bicubic warps and warps
// that rotate the picture ran as two launches through the context workspace - tile warp, filter -,
// i.e. 16 B/px of HBM traffic for an 8 B/px workload.  Here a workgroup of 256 threads owns a
// 64-px-wide column of the output and walks down it in steps of 32 rows:
//   * a step's REGION is 32 rows x (64 + 2 H) columns of the warped picture (H = K / 2): the 64
//     columns the workgroup writes plus the H columns either side its x pass needs; the region's
//     corners through the homography give its source box, staged in LDS row by row as in
//     tile_warp.hpp; every thread samples 8 pixels of the 64 main columns (lane = column) and one of
//     the 2 H x 32 halo pixels - at K = 9 that is exactly one each - into the LDS image W;
//   * the y pass runs down W into the LDS image V (the float32 intermediate scipy stores between
//     its passes), the x pass along V; 4 px x 2 rows per thread, stored as float4 rows of 256 bytes;
//   * the last 2 H rows of W are what the next step's y pass starts with: they are kept per frame
//     (the frames of a group share a step's coordinates - a double division per pixel - so the
//     loop order is step, then frame) in LDS;
//   * a workgroup walks `steps` steps = 32 steps - 2 H output rows (its first step has no rows above
//     it and writes 32 - 2 H rows): sampling overhead (64 + 2 H) / 64 x 32 steps / (32 steps - 2 H),
//     1.2 at K = 9 and 4 steps, against 1.41 for a free-standing 64 x 32 tile with its halo.
// Pixels of the region outside the picture take the filter's border mode: the warped pixel at the
// resolved index (reflect, mirror, nearest, wrap), or the filter's cval.  Arithmetic and order are
// those of the two launches - sample() of sampler.hpp, then wave_sep.hpp's ascending fma chains
// with the intermediate rounded to float32 - so the results have the same bits
// (tests/test_gpu_tile_chain.py, tools/fuzz_chain.py).
//
// Status (round 5): behind the knob tile_chain, OFF.  16 x 4K + 9 + 9: bicubic 0.80 ms against 0.55
// for the two launches, bilinear under 15 degrees 0.87 against 0.56 - the tile warp is bound by its
// vector work, not by the workspace traffic this kernel saves, and the filter passes run in the same
// waves at 3 workgroups per CU instead of at stream rate in wave_sep.hpp (profiles/r05_micro.txt).
#pragma once

#include "tile_warp.hpp"

namespace ipa {

constexpr int kChainRows = 32;    // rows of the warped picture a step samples
constexpr int kChainWP = 72;      // floats per row of W and V (64 + 2 H at K = 9; rows 16-byte aligned)
constexpr int kChainMaxFrames = 8;

struct TileChainArgs {
  char* dst;
  long dst_frame_elems, dpitch;
  const char* src;
  long src_frame_bytes;
  unsigned src_bytes;
  int sh, sw, spitch, dh, dw;
  int n_frames, frames_wg, group_chunk;
  int border, q5;        // of the warp
  float cubic_a, cval;
  int cby, cbx;          // the filter's border modes
  float ccval;           // ... and its constant
  int strips_x, segs, steps, seg_rows;   // seg_rows = 32 steps - 2 H
  int pitch, rows;       // the LDS box
  int vec_out;           // rows of the result 16-byte aligned
  float ky[9], kx[9];
};

// rows (columns) of the picture the pixels first .. first + count - 1 of a region resolve to, as far
// as the filter needs them (H beyond the picture): what the region's source box has to hold.
// reflect / mirror / nearest stay within H + 1 of the edge they pass; wrap lands on the far side
// (those footprints then miss the box and are sampled tap by tap).
__host__ __device__ inline void chain_axis_range(int first, int count, int n, int H, int& lo, int& hi) {
  int a = first, b = first + count - 1;
  lo = a < 0 ? 0 : (a > n - 1 ? n - 1 : a);
  hi = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
  if (a < 0) { const int r = H < n - 1 ? H : n - 1; hi = hi > r ? hi : r; }
  if (b > n - 1) { const int r = n - 1 - H > 0 ? n - 1 - H : 0; lo = lo < r ? lo : r; }
}

template <int INTERP, int K>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
tile_chain_kernel(TileChainArgs a, HomographyCoord coord) {
  using C = double;
  static_assert(INTERP == kLinear || INTERP == kCubic, "bilinear and bicubic warps");
  static_assert(K == 3 || K == 5 || K == 7 || K == 9, "3 .. 9 taps per axis");
  constexpr int NT = ntaps<INTERP>::value;
  constexpr int H = K / 2, TS = kChainRows, WP = kChainWP;
  constexpr int NPX = 9;                 // 8 of the main columns, 1 of the halo columns
  constexpr int kHaloPx = 2 * H * TS;    // <= 256
  constexpr int kKeep = 2 * H * WP;      // floats of W a step hands to the next one
  extern __shared__ __attribute__((aligned(16))) float chain_lds[];
  float* const box = chain_lds;
  float* const Wb = box + ((a.pitch * a.rows + 3) & ~3);
  float* const Vb = Wb + (TS + 2 * H) * WP;
  float* const Hb = Vb + TS * WP;
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const unsigned groups = ((unsigned)a.n_frames + a.frames_wg - 1) / (unsigned)a.frames_wg;
  const unsigned units = (unsigned)a.strips_x * (unsigned)a.segs;
  const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  unsigned grp = b % groups, unit = b / groups;
  if (a.group_chunk) {
    const unsigned gc = (unsigned)a.group_chunk, per = gc * units;
    const unsigned chunk = b / per, r = b - chunk * per;
    grp = chunk * gc + r % gc;
    unit = r / gc;
  }
  const int segi = (int)(unit / (unsigned)a.strips_x), sxi = (int)unit - segi * a.strips_x;
  const int x0 = sxi * 64, Y0 = segi * a.seg_rows;
  const int yend = Y0 + a.seg_rows < a.dh ? Y0 + a.seg_rows : a.dh;
  const unsigned f0 = grp * (unsigned)a.frames_wg;
  const unsigned f1 = f0 + (unsigned)a.frames_wg < (unsigned)a.n_frames ? f0 + (unsigned)a.frames_wg
                                                                        : (unsigned)a.n_frames;
  const int nf = (int)(f1 - f0);
  // steps that have rows to write: the rows of step k end at Y0 - 2 H + 32 k + 31
  int nsteps = (yend - Y0 + 2 * H + TS - 1) / TS;
  nsteps = nsteps < a.steps ? nsteps : a.steps;
  const int items = nsteps * nf;   // (step, frame) pairs, frame fastest
  SrcView s;
  s.h = a.sh; s.w = a.sw; s.pitch = a.spitch;
  s.border = a.border; s.q5 = a.q5; s.cubic_a = a.cubic_a;
  s.lanczos = nullptr;
  float* dst0 = reinterpret_cast<float*>(a.dst);

  // this thread's pixels of a region: 8 of the main columns (rows wave + 4 j, column H + lane) and
  // one of the halo columns (row tid / 2H; columns 0 .. H - 1 left, 64 + H .. 64 + 2 H - 1 right)
  const int hrow = (int)tid / (2 * H), hc = (int)tid - hrow * (2 * H);
  const int hcol = hc < H ? hc : 64 + hc;
  const bool has_halo = (int)tid < kHaloPx;
  // the columns of the picture behind them, through the filter's border mode (-1: its constant)
  const int ix_main = x0 + (int)lane, ix_halo = x0 - H + hcol;
  const int ox_main = ix_main <= a.dw - 1 + H ? resolve_idx(ix_main, a.dw, a.cbx) : -1;
  const int ox_halo = ix_halo <= a.dw - 1 + H ? resolve_idx(ix_halo, a.dw, a.cbx) : -1;
  // (columns of the intermediate outside the picture in the constant mode: scipy pads the
  // INTERMEDIATE with cval)
  const bool xc_main = a.cbx == IPA_BORDER_CONSTANT && ix_main >= a.dw;
  const bool xc_halo = a.cbx == IPA_BORDER_CONSTANT && (unsigned)ix_halo >= (unsigned)a.dw;

  // The source box of the region a step samples (first row R0 of the picture): the region's corners
  // through the homography - lanes 0-3 of every wave evaluate one each, the rest read them from
  // there (no LDS, no barrier).
  struct Geo { int bx0, by0, bw, bh; };
  auto bcast = [](double v, int l) -> double {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)__double2loint(v), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double((int)hi, (int)lo);
  };
  auto geometry = [&](int R0) -> Geo {
    int xlo, xhi, ylo, yhi;
    chain_axis_range(x0 - H, 64 + 2 * H, a.dw, H, xlo, xhi);
    chain_axis_range(R0, TS, a.dh, H, ylo, yhi);
    double sx = 0.0, sy = 0.0;
    if (lane < 4u) coord.get((lane & 1u) ? xhi : xlo, (lane & 2u) ? yhi : ylo, sx, sy);
    double lox = 0, hix = 0, loy = 0, hiy = 0;
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double cx = bcast(sx, k), cy = bcast(sy, k);
      fin = fin && ipa_abs(cx) < 1.0e6 && ipa_abs(cy) < 1.0e6;   // false for NaN
      if (k == 0) { lox = hix = cx; loy = hiy = cy; }
      lox = cx < lox ? cx : lox; hix = cx > hix ? cx : hix;
      loy = cy < loy ? cy : loy; hiy = cy > hiy ? cy : hiy;
    }
    Geo g;
    tile_axis_box<NT>(lox, hix, a.sw, g.bx0, g.bw);
    tile_axis_box<NT>(loy, hiy, a.sh, g.by0, g.bh);
    if (!fin) g.bw = g.bh = 0;
    g.bw = g.bw < a.pitch ? g.bw : a.pitch;
    g.bw = g.bw < 128 ? g.bw : 128;
    g.bh = g.bh < a.rows ? g.bh : a.rows;
    g.bx0 = __builtin_amdgcn_readfirstlane(g.bx0); g.by0 = __builtin_amdgcn_readfirstlane(g.by0);
    g.bw = __builtin_amdgcn_readfirstlane(g.bw); g.bh = __builtin_amdgcn_readfirstlane(g.bh);
    return g;
  };
  // a box whose cells are the source's or the border constant is read row by row (wave w rows w,
  // w + 4, ...; lane = column; the columns past 64 go 64 >> esh rows per load), its first 4 kFly
  // rows REQUESTED while the previous frame is being sampled (box_issue) and written to LDS after it
  // (box_commit), the rest when the frame's turn has come (box_rest); the other border modes on the
  // rim of the source go cell by cell
  auto is_direct = [&](const Geo& g) -> bool {
    const bool inside = g.bx0 >= 0 && g.by0 >= 0 && g.bx0 + g.bw <= a.sw && g.by0 + g.bh <= a.sh;
    return inside || a.border == IPA_BORDER_CONSTANT;
  };
  constexpr int kFly = 10;
  float v0[kFly], v1[4];
  auto far_shift = [](int bw) -> int {
    const int e = bw - 64;
    return e <= 4 ? 2 : (e <= 8 ? 3 : (e <= 16 ? 4 : (e <= 32 ? 5 : 6)));
  };
  auto cell_live = [&](const Geo& g, int r, int c) -> bool {
    return r < g.bh && c < g.bw && (unsigned)(g.by0 + r) < (unsigned)a.sh && (unsigned)(g.bx0 + c) < (unsigned)a.sw;
  };
  auto cell_load = [&](const __amdgpu_buffer_rsrc_t& rs, const Geo& g, int r, int c) -> float {
    const bool live = cell_live(g, r, c);
    return u2f(__builtin_amdgcn_raw_buffer_load_b32(rs, live ? (__mul24(g.by0 + r, a.spitch) + g.bx0 + c) << 2 : 0, 0, 0));
  };
  auto box_issue = [&](const Geo& g, const __amdgpu_buffer_rsrc_t& rs) {
#pragma unroll
    for (int u = 0; u < kFly; u++) v0[u] = cell_load(rs, g, (int)wave + 4 * u, (int)lane);
    if (g.bw > 64) {
      const int esh = far_shift(g.bw), rstep = 64 >> esh;
      const int lr = (int)lane >> esh, lc = 64 + ((int)lane & ((1 << esh) - 1));
#pragma unroll
      for (int u = 0; u < 4; u++) v1[u] = cell_load(rs, g, ((int)wave + 4 * u) * rstep + lr, lc);
    }
  };
  auto box_commit = [&](const Geo& g) {
#pragma unroll
    for (int u = 0; u < kFly; u++) {
      const int r = (int)wave + 4 * u;
      if (r < g.bh && (int)lane < g.bw) box[__mul24(r, a.pitch) + (int)lane] = cell_live(g, r, (int)lane) ? v0[u] : a.cval;
    }
    if (g.bw > 64) {
      const int esh = far_shift(g.bw), rstep = 64 >> esh;
      const int lr = (int)lane >> esh, lc = 64 + ((int)lane & ((1 << esh) - 1));
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int r = ((int)wave + 4 * u) * rstep + lr;
        if (r < g.bh && lc < g.bw) box[__mul24(r, a.pitch) + lc] = cell_live(g, r, lc) ? v1[u] : a.cval;
      }
    }
  };
  auto box_rest = [&](const Geo& g, const __amdgpu_buffer_rsrc_t& rs) {
    constexpr int kMore = 8;
#pragma unroll 1
    for (int r0 = (int)wave + 4 * kFly; r0 < g.bh; r0 += 4 * kMore) {
      float t[kMore];
#pragma unroll
      for (int u = 0; u < kMore; u++) t[u] = cell_load(rs, g, r0 + 4 * u, (int)lane);
#pragma unroll
      for (int u = 0; u < kMore; u++) {
        const int r = r0 + 4 * u;
        if (r < g.bh && (int)lane < g.bw) box[__mul24(r, a.pitch) + (int)lane] = cell_live(g, r, (int)lane) ? t[u] : a.cval;
      }
    }
    if (g.bw > 64) {
      const int esh = far_shift(g.bw), rstep = 64 >> esh;
      const int lr = (int)lane >> esh, lc = 64 + ((int)lane & ((1 << esh) - 1));
#pragma unroll 1
      for (int r0 = ((int)wave + 16) * rstep; r0 < g.bh; r0 += 4 * rstep) {
        const int r = r0 + lr;
        const float t = cell_load(rs, g, r, lc);
        if (r < g.bh && lc < g.bw) box[__mul24(r, a.pitch) + lc] = cell_live(g, r, lc) ? t : a.cval;
      }
    }
  };
  auto box_cells = [&](const Geo& g, const __amdgpu_buffer_rsrc_t& rs) {
    const int cells = g.bw * g.bh;
    const float inv_bw = 1.0f / (float)(g.bw > 0 ? g.bw : 1);
#pragma unroll 1
    for (int i = (int)tid; i < cells; i += 256) {
      const int row = (int)(((float)i + 0.5f) * inv_bw), col = i - row * g.bw;
      const int yy = resolve_idx(g.by0 + row, a.sh, a.border);
      const int xx = resolve_idx(g.bx0 + col, a.sw, a.border);
      const bool live = yy >= 0 && xx >= 0;
      const float t = u2f(__builtin_amdgcn_raw_buffer_load_b32(rs, live ? (__mul24(yy, a.spitch) + xx) << 2 : 0, 0, 0));
      box[__mul24(row, a.pitch) + col] = live ? t : a.cval;
    }
  };
  auto frame_rsrc = [&](int fi) { return make_rsrc(a.src + (long)(f0 + (unsigned)fi) * a.src_frame_bytes, a.src_bytes); };

  // the footprints of this thread's pixels, once per step for the frames of the group:
  // ad >= 0: LDS index of the first tap; -1: through sample() tap by tap; -2: the filter's constant
  int ad[NPX];
  float tx[NPX], ty[NPX];
  unsigned slow = 0;
  // (rows more than H outside the picture feed no row that is written)
  auto row_of = [&](int R0, int rrow) -> int {
    const int iy = R0 + rrow;
    return (iy >= -H && iy <= a.dh - 1 + H) ? resolve_idx(iy, a.dh, a.cby) : -1;
  };
  auto footprints = [&](const Geo& g, int R0) {
    slow = 0;
    auto one = [&](int j, int ox, int oy) {
      if (ox < 0 || oy < 0) {
        ad[j] = -2; tx[j] = ty[j] = 0.f;
        return;
      }
      C sx, sy;
      coord.get(ox, oy, sx, sy);
      const bool ok = ipa_abs(sx) < (C)kCoordLimit && ipa_abs(sy) < (C)kCoordLimit;
      if (!ok) sx = sy = (C)0;
      int ix0, iy0;
      axis_frac<INTERP, float, C>(s, sx, ix0, tx[j]);
      axis_frac<INTERP, float, C>(s, sy, iy0, ty[j]);
      const int cx = ix0 - g.bx0, cy = iy0 - g.by0;
      const bool in = ok && cx >= 0 && cy >= 0 && cx + NT <= g.bw && cy + NT <= g.bh;
      ad[j] = in ? __mul24(cy, a.pitch) + cx : -1;
      slow |= in ? 0u : 1u << j;
    };
#pragma unroll
    for (int j = 0; j < 8; j++) {
      one(j, ox_main, row_of(R0, (int)wave + 4 * j));
      __builtin_amdgcn_sched_barrier(0);   // one pixel's double arithmetic at a time (registers)
    }
    if (has_halo) one(8, ox_halo, row_of(R0, hrow));
    else { ad[8] = -2; tx[8] = ty[8] = 0.f; }
  };
  auto wslot = [&](int j) -> int {
    return j < 8 ? (2 * H + (int)wave + 4 * j) * WP + H + (int)lane : (2 * H + hrow) * WP + hcol;
  };
  auto sample_one = [&](int j) {
    float wx[NT], wy[NT];
    const float* tp = box + (ad[j] < 0 ? 0 : ad[j]);
    weights_from_frac<INTERP, float>(s, tx[j], wx);
    weights_from_frac<INTERP, float>(s, ty[j], wy);
    float o = 0.f;
#pragma unroll
    for (int r = 0; r < NT; r++) {
      const float* tr = tp + r * a.pitch;
      float rs = wx[0] * tr[0];
#pragma unroll
      for (int c = 1; c < NT; c++) rs = ipa_fma(wx[c], tr[c], rs);
      o = r == 0 ? wy[0] * rs : ipa_fma(wy[r], rs, o);
    }
    if (ad[j] >= 0) Wb[wslot(j)] = o;
    else if (ad[j] == -2) Wb[wslot(j)] = a.ccval;
  };
  // the samples of a frame into W (rows 2 H ..)
  auto sample = [&](int R0) {
#pragma unroll
    for (int j = 0; j < NPX; j++) asm volatile("" : "+v"(tx[j]), "+v"(ty[j]), "+v"(ad[j]));
    constexpr int kGroup = INTERP == kLinear ? 4 : 2;
#pragma unroll
    for (int j0 = 0; j0 < 8; j0 += kGroup) {
#pragma unroll
      for (int j = j0; j < j0 + kGroup; j++) sample_one(j);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_halo) sample_one(8);
    // (rare: footprints the box does not hold, through the gather kernel's sample())
    if (slow) {
#pragma unroll 1
      for (int j = 0; j < NPX; j++) {
        if (!((slow >> j) & 1u)) continue;
        const int ox = j < 8 ? ox_main : ox_halo;
        const int oy = row_of(R0, j < 8 ? (int)wave + 4 * j : hrow);
        C sx, sy;
        coord.get(ox, oy, sx, sy);
        Wb[wslot(j)] = tile_slow_sample<INTERP, C>(s, sx, sy, a.cval);
      }
    }
  };
  // y pass: wave w takes rows 8 w .. 8 w + 7 of V in the main columns (8 + 2 H rows of W down a
  // column), every thread one pixel of the halo columns; ascending fma chains as wave_sep.hpp.  The
  // last 2 H rows of W go to the frame's store for the next step.
  auto y_pass = [&](int fi) {
    float col[8 + 2 * H];
    const float* wc = Wb + (8 * (int)wave) * WP + H + (int)lane;
#pragma unroll
    for (int i = 0; i < 8 + 2 * H; i++) col[i] = wc[i * WP];
    float* vc = Vb + (8 * (int)wave) * WP + H + (int)lane;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      float acc = a.ky[0] * col[q];
#pragma unroll
      for (int i = 1; i < K; i++) acc = fmaf(a.ky[i], col[q + i], acc);
      vc[q * WP] = xc_main ? a.ccval : acc;
    }
    if (has_halo) {
      const float* hw = Wb + hrow * WP + hcol;
      float acc = a.ky[0] * hw[0];
#pragma unroll
      for (int i = 1; i < K; i++) acc = fmaf(a.ky[i], hw[i * WP], acc);
      Vb[hrow * WP + hcol] = xc_halo ? a.ccval : acc;
    }
    float* hb = Hb + fi * kKeep;
#pragma unroll
    for (int i = (int)tid; i < kKeep; i += 256) hb[i] = Wb[TS * WP + i];
  };
  // x pass and store: 4 px x 2 rows per thread (lanes 0-15 along a row: 256 bytes)
  auto x_pass = [&](int fi, int R0) {
    const int cg = (int)(lane & 15u), rs = (int)(lane >> 4);
    float* dfr = dst0 + (long)(f0 + (unsigned)fi) * a.dst_frame_elems;
    const int x = x0 + 4 * cg;
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
      const int r = 8 * (int)wave + 4 * h2 + rs;
      const float4* vr = reinterpret_cast<const float4*>(Vb + r * WP + 4 * cg);
      float win[12];
#pragma unroll
      for (int m = 0; m < (4 + 2 * H + 3) / 4; m++) {
        const float4 q = vr[m];
        win[4 * m] = q.x; win[4 * m + 1] = q.y; win[4 * m + 2] = q.z; win[4 * m + 3] = q.w;
      }
      float out[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float acc = a.kx[0] * win[q];
#pragma unroll
        for (int jj = 1; jj < K; jj++) acc = fmaf(a.kx[jj], win[q + jj], acc);
        out[q] = acc;
      }
      const int y = R0 - H + r;
      if (y >= Y0 && y < yend && x < a.dw) {
        float* row = dfr + (long)y * a.dpitch + x;
        if (a.vec_out && x + 3 < a.dw) {
          *reinterpret_cast<float4*>(row) = float4{out[0], out[1], out[2], out[3]};
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (x + q < a.dw) row[q] = out[q];
        }
      }
    }
  };

  // The items (step, frame) in two phases with a barrier behind each, the filter one item behind the
  // warp: phase 1 writes item i's box to LDS and runs the y pass of item i - 1 (W -> V); phase 2
  // requests item i + 1's box, samples item i (box -> W) and runs the x pass of item i - 1
  // (V -> memory).  Every phase has loads or stores in flight under its LDS and vector work.
  Geo gc = geometry(Y0 - H);
  if (is_direct(gc)) box_issue(gc, frame_rsrc(0));
  int pfi = 0, pR0 = 0;
#pragma unroll 1
  for (int i = 0; i <= items; i++) {
    const int st = i / nf, fi = i - st * nf;
    const int R0 = Y0 - H + TS * st;   // first row of the picture this item samples
    // phase 1
    if (i < items) {
      s.rsrc = frame_rsrc(fi);
      if (is_direct(gc)) {
        box_commit(gc);
        box_rest(gc, s.rsrc);
      } else {
        box_cells(gc, s.rsrc);
      }
    }
    if (i > 0) y_pass(pfi);
    __syncthreads();
    // phase 2
    Geo gn = gc;
    if (i + 1 < items) {
      const int st1 = (i + 1) / nf, fi1 = (i + 1) - st1 * nf;
      if (st1 != st) gn = geometry(Y0 - H + TS * st1);
      if (is_direct(gn)) box_issue(gn, frame_rsrc(fi1));
    }
    if (i < items) {
      if (fi == 0) footprints(gc, R0);
      if (st > 0) {   // the rows the previous step kept for this frame: W rows 0 .. 2 H - 1
        const float* hb = Hb + fi * kKeep;
#pragma unroll
        for (int k = (int)tid; k < kKeep; k += 256) Wb[k] = hb[k];
      }
      sample(R0);
    }
    if (i > 0) x_pass(pfi, pR0);
    __syncthreads();
    pfi = fi; pR0 = R0;
    gc = gn;
  }
}

// LDS floats of a launch: the box, W, V and the rows kept per frame
template <int K>
static inline size_t tile_chain_lds_bytes(int pitch, int rows, int frames_wg) {
  constexpr int H = K / 2;
  return (size_t)(((pitch * rows + 3) & ~3) + (kChainRows + 2 * H) * kChainWP + kChainRows * kChainWP +
                  frames_wg * 2 * H * kChainWP) * 4;
}

// The largest source box over all regions of a launch.  false: not a warp for this kernel (the
// plane's horizon crosses the picture, or a region's box is wider than 128 columns / larger than
// the LDS reserve of tile_warp.hpp)
template <int NT>
static inline bool tile_chain_box(const double* m, int dh, int dw, int sh, int sw, int H, int steps,
                                  int* pitch, int* rows) {
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
  const int seg_rows = kChainRows * steps - 2 * H;
  int mw = 0, mh = 0;
  for (int Y0 = 0; Y0 < dh; Y0 += seg_rows)
    for (int st = 0; st < steps; st++) {
      const int R0 = Y0 - H + kChainRows * st;
      const int yend = Y0 + seg_rows < dh ? Y0 + seg_rows : dh;
      if (R0 - H >= yend) break;
      int ylo, yhi;
      chain_axis_range(R0, kChainRows, dh, H, ylo, yhi);
      for (int x0 = 0; x0 < dw; x0 += 64) {
        int xlo, xhi;
        chain_axis_range(x0 - H, 64 + 2 * H, dw, H, xlo, xhi);
        double lox = 0, hix = 0, loy = 0, hiy = 0;
        for (int k = 0; k < 4; k++) {
          double cx, cy;
          at((k & 1) ? xhi : xlo, (k & 2) ? yhi : ylo, cx, cy);
          if (!(fabs(cx) < 1.0e6) || !(fabs(cy) < 1.0e6)) return false;
          if (k == 0) { lox = hix = cx; loy = hiy = cy; }
          lox = cx < lox ? cx : lox; hix = cx > hix ? cx : hix;
          loy = cy < loy ? cy : loy; hiy = cy > hiy ? cy : hiy;
        }
        int f, bw, bh;
        tile_axis_box<NT>(lox, hix, sw, f, bw);
        tile_axis_box<NT>(loy, hiy, sh, f, bh);
        mw = bw > mw ? bw : mw;
        mh = bh > mh ? bh : mh;
      }
    }
  if (mw < NT || mh < NT) return false;
  if (mw > 128) return false;
  mw |= 1;
  if ((long)mw * mh * 4 > kWarpTileLdsBytes) return false;
  *pitch = mw;
  *rows = mh;
  return true;
}

// tile_chain.hip
int tile_chain_run(hipStream_t stream, const TileChainArgs& t, const HomographyCoord& coord, int interp,
                   int K, unsigned grid, size_t lds);

}  // namespace ipa
