// ring_stencil.hpp — planning pass of the LDS-ring remap kernel (ring_remap.hpp): which strips of
// a batch remap can take their source rows through a wave-private LDS ring, decided ONCE per call
// (and kept for sources given by value) instead of inside the hot loop.
//
//   ring_plan_kernel   one wave per 128-px strip walks the strip's coordinates once (the same
//                      maps / lens model / homography every frame of the batch uses) and
//                      decides whether the strip is CLEAN: every footprint wholly inside the
//                      source, the footprints of every two-row step within the ring's source
//                      rows, of the whole strip within kRW columns, source rows only moving
//                      forward and at most kRingMaxNew new rows per step.  For a clean strip it
//                      records the ring schedule: window origin, first source row, new rows per
//                      step (4 bits each); for sources given by value it also stores the
//                      coordinates it evaluated.
//
// The strips that are not clean (frame rim, footprints on the source border, strong rotation ...)
// run on the gather kernel behind the `pair_clean` skip mask.  Arithmetic, rounding and summation
// order are those of sample(): the two kernels produce identical bits.
//
// Reference semantics: camera/LensDistortion.py:323-326 (cv2.remap), camera/PerspectiveCorrection.py
// :377-378, :401-405 (cv2.warpPerspective).
#pragma once

#include "ring_geom.hpp"
#include "stored_coords.hpp"

namespace ipa {

constexpr int kRingMaxNew = 4;   // source rows a step may add (steps after the first)
constexpr int kPlanWords = 8;    // 4-bit row counts of up to 64 steps

struct RingPlan {
  int4* info;            // per strip: x = clean, y = xlo, z = first source row, w = rows of step 0
  unsigned* cnts;        // per strip: kPlanWords words of 4-bit counts
  unsigned* pair_clean;  // per strip pair (= one strip of wave_stencil_kernel): 1 = both clean
  unsigned* stats;       // [0] = clean pairs of the call (zeroed before the planning launch)
};

// footprint of the interpolation the plan is made for
struct RingTaps {
  int nt;                // taps per axis (2 bilinear, 4 bicubic, 8 Lanczos4)
  int q5;                // coordinates rounded to 1/32 px first (cv2's rule; always for Lanczos4)
  int rr;                // ring rows the kernel keeps (power of two)
};

struct RingGeom {
  int dh, dw;            // output (= filter domain)
  int strips_x;          // 128-px strips per strip row
  int pairs_x;           // (strips_x + 1) / 2
  int strip_h;           // output rows per strip
  int strips;            // strips_x * strip rows
  int pairs;             // pairs_x * strip rows
};

// coordinates of the 4 samples of a step of an interior strip (k = 2 * row + column group)
template <typename Coord>
__device__ __forceinline__ void ring_coords(const Coord& coord, int xs, int v0, int v1,
                                            typename Coord::coord_t (&sx)[4],
                                            typename Coord::coord_t (&sy)[4]) {
  const int lane = threadIdx.x & 63;
  if constexpr (coord_is_table<Coord>::value) {
    const int v[2] = {v0, v1};
#pragma unroll
    for (int r = 0; r < 2; r++) {
      // 32-bit element offsets (maps are far below 2^32 elements): scalar base + lane offset
      const unsigned o = (unsigned)v[r] * (unsigned)coord.pitch + (unsigned)xs;  // scalar
#pragma unroll
      for (int q = 0; q < 2; q++) {
        sx[2 * r + q] = coord.mx[o + (unsigned)lane + 64u * q];
        sy[2 * r + q] = coord.my[o + (unsigned)lane + 64u * q];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) coord.get(xs + lane + 64 * (k & 1), (k >> 1) ? v1 : v0, sx[k], sy[k]);
  }
}

// ------------------------------------------------------------------------- planning --
template <typename Coord, int K>
__global__ void __launch_bounds__(128)
ring_plan_kernel(RingGeom gm, Coord coord, int sh, int sw, RingTaps tp, RingPlan plan,
                 typename Coord::coord_t* outx, typename Coord::coord_t* outy) {
  using C = typename Coord::coord_t;
  using G = group_geom<K>;
  __shared__ int flag[2];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = (int)blockIdx.x;
  const int syi = pair / gm.pairs_x, px = pair - syi * gm.pairs_x;
  const int sxi = 2 * px + (int)wave;
  const bool exists = sxi < gm.strips_x;
  const int sid = syi * gm.strips_x + sxi;

  const int xs = sxi * G::OW - 2 * G::HL;
  const int y0 = syi * gm.strip_h;
  const int nrows = gm.dh - y0 < gm.strip_h ? gm.dh - y0 : gm.strip_h;
  const int T = nrows + K - 1;
  const int nsteps = (T + 1) / 2;
  bool clean = exists && xs >= 0 && xs + kSW <= gm.dw && y0 - G::H >= 0 &&
               y0 - G::H + T <= gm.dh && nsteps <= 8 * kPlanWords;

  SrcView s;  // only what axis_frac reads
  s.q5 = tp.q5;
  const int nt = tp.nt, back = tp.nt / 2 - 1;  // first tap = floor(coordinate) - back
  int sxmin = INT_MAX, sxmax = INT_MIN, ybase = 0, hrun = 0, cnt0 = 0;
  unsigned words[kPlanWords];
#pragma unroll
  for (int i = 0; i < kPlanWords; i++) words[i] = 0u;

  if (clean) {
#pragma unroll 1
    for (int st = 0; st < nsteps; st++) {
      const int v0 = y0 - G::H + 2 * st;
      const bool live1 = 2 * st + 1 < T;
      const int v1 = live1 ? v0 + 1 : v0;
      C sx[4], sy[4];
      ring_coords<Coord>(coord, xs, v0, v1, sx, sy);
      if (outx) {  // coordinates computed in the kernel: kept for the frames of the batch
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned o = (unsigned)((k >> 1) ? v1 : v0) * (unsigned)gm.dw +
                             (unsigned)(xs + (int)lane + 64 * (k & 1));
          outx[o] = sx[k];
          outy[o] = sy[k];
        }
      }
      int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
      bool bad = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool live = k < 2 || live1;
        const bool ok = sx[k] > (C)-kCoordLimit && sx[k] < (C)kCoordLimit &&
                        sy[k] > (C)-kCoordLimit && sy[k] < (C)kCoordLimit;
        int ix0, iy0;
        float tx, ty;
        axis_frac<kLinear, float, C, -1>(s, ok ? sx[k] : (C)0, ix0, tx);
        axis_frac<kLinear, float, C, -1>(s, ok ? sy[k] : (C)0, iy0, ty);
        ix0 -= back;
        iy0 -= back;
        const bool inside = ok && ix0 >= 0 && iy0 >= 0 && ix0 + nt <= sw && iy0 + nt <= sh;
        bad = bad || (live && !inside);
        const bool use = live && inside;
        const int xl = use ? ix0 : INT_MAX, xh = use ? ix0 : INT_MIN;
        const int yl = use ? iy0 : INT_MAX, yh = use ? iy0 : INT_MIN;
        xmn = xl < xmn ? xl : xmn;
        xmx = xh > xmx ? xh : xmx;
        ymn = yl < ymn ? yl : ymn;
        ymx = yh > ymx ? yh : ymx;
      }
      wave_span(xmn, xmx, ymn, ymx);
      if (__builtin_amdgcn_ballot_w64(bad) != 0 || ymx - ymn + nt > tp.rr) {
        clean = false;
        break;
      }
      sxmin = xmn < sxmin ? xmn : sxmin;
      sxmax = xmx > sxmax ? xmx : sxmax;
      const int need = ymx + nt;
      int cnt;
      if (st == 0) {
        ybase = ymn;
        hrun = need;
        cnt0 = need - ymn;  // <= rr
        cnt = 0;
      } else {
        cnt = need > hrun ? need - hrun : 0;
        hrun = need > hrun ? need : hrun;
      }
      // the ring holds rows [max(ybase, hrun - rr), hrun)
      const int lowest = ybase > hrun - tp.rr ? ybase : hrun - tp.rr;
      if (cnt > kRingMaxNew || ymn < lowest) {
        clean = false;
        break;
      }
#pragma unroll
      for (int i = 0; i < kPlanWords; i++)
        if (i == (st >> 3)) words[i] |= (unsigned)cnt << (4 * (st & 7));
    }
  }
  const int xlo = (sxmin & ~3) - 4;
  if (clean && sxmax + nt > xlo + kRW) clean = false;
  if (exists && lane == 0) {
    plan.info[sid] = int4{clean ? 1 : 0, xlo, ybase, cnt0};
#pragma unroll
    for (int i = 0; i < kPlanWords; i++) plan.cnts[sid * kPlanWords + i] = words[i];
  }
  if (lane == 0) flag[wave] = (clean || !exists) ? 1 : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned pc = (flag[0] && flag[1]) ? 1u : 0u;
    plan.pair_clean[pair] = pc;
    if (pc && plan.stats) atomicAdd(plan.stats, 1u);
  }
}

// ------------------------------------------------------------------ host: plan + reuse --
// the coordinate source the ring kernels read: maps as they are, sources given by value
// through the coordinates the planning pass stored
template <typename Coord> struct ring_kernel_coord {
  using type = StoredCoord<typename Coord::coord_t>;
};
template <> struct ring_kernel_coord<MapCoord> { using type = MapCoord; };

// Plan of a call in ctx->plan: strip records, step counts, pair flags and - for sources given
// by value - the coordinates of the clean strips, in the source's own type (so the frames
// sample exactly what the gather / per-frame kernels compute).  The planning pass evaluates
// every such coordinate anyway; the next call with the same source and geometry skips the pass.
template <typename Coord, int K>
static int ring_plan_prepare(ipa_ctx* ctx, const RingGeom& gm, const Coord& coord, int sh, int sw,
                             const RingTaps& tp, RingPlan* plan,
                             typename ring_kernel_coord<Coord>::type* kc) {
  using CT = typename Coord::coord_t;
  constexpr bool kByValue = !std::is_same<Coord, MapCoord>::value;
  if (kByValue && (size_t)gm.dh * gm.dw >= (1ull << 32)) return 1;  // 32-bit coordinate offsets
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t info_b = up((size_t)gm.strips * sizeof(int4));
  const size_t cnts_b = up((size_t)gm.strips * kPlanWords * sizeof(unsigned));
  const size_t pair_b = up((size_t)gm.pairs * sizeof(unsigned));
  const size_t stat_b = 256;
  const size_t coord_b = kByValue ? up((size_t)gm.dh * gm.dw * sizeof(CT)) : 0;
  double key[40];
  int kn = coord_key(coord, key);
  const double g[10] = {(double)gm.dh, (double)gm.dw, (double)sh, (double)sw, (double)tp.nt,
                        (double)tp.q5, (double)tp.rr, (double)sizeof(CT), (double)K,
                        (double)gm.strip_h};
  for (int i = 0; i < 10; i++) key[kn++] = g[i];
  // How many strips the last call with this source and geometry found clean (read back without
  // waiting, so possibly one call old): a call that would leave most strips to the gather
  // kernel anyway - strong rotation, footprints outside the source - skips the ring path: under
  // half of them clean, or under 85 % for bilinear, whose ring is only 6 % ahead of the gather
  // kernel (16 x 4K, alpha = 1 lens maps, 80 % clean: ring + rest 0.323 ms, gather alone 0.311).  A
  // hint only: the results are the same bits either way.
  const bool same = ctx->ring_hint_n == kn &&
                    memcmp(ctx->ring_hint_key, key, (size_t)kn * sizeof(double)) == 0;
  // (a stale "few clean strips" - a read-back that landed after the key changed, maps rewritten
  // at the same address - would otherwise keep the ring path off for as long as the key stays:
  // every 16th skipped call plans again and refreshes the hint)
  if (same && ctx->ring_hint && ctx->ring_hint[1] == (unsigned)gm.pairs &&
      ctx->ring_hint[0] != 0xffffffffu &&
      100ul * ctx->ring_hint[0] < (tp.nt == 2 ? 85ul : 50ul) * (unsigned long)gm.pairs &&
      (++ctx->ring_hint_skips & 15u) != 0)
    return 1;
  const bool hit = kByValue && ctx->plan_key_n == kn &&
                   memcmp(ctx->plan_key, key, (size_t)kn * sizeof(double)) == 0;
  if (!hit) {
    int rc = ipa_plan_reserve(ctx, info_b + cnts_b + pair_b + stat_b + 2 * coord_b);
    if (rc) return rc;
  }
  char* pb = reinterpret_cast<char*>(ctx->plan);
  plan->info = reinterpret_cast<int4*>(pb);
  plan->cnts = reinterpret_cast<unsigned*>(pb + info_b);
  plan->pair_clean = reinterpret_cast<unsigned*>(pb + info_b + cnts_b);
  plan->stats = reinterpret_cast<unsigned*>(pb + info_b + cnts_b + pair_b);
  char* cb = pb + info_b + cnts_b + pair_b + stat_b;
  CT* outx = kByValue ? reinterpret_cast<CT*>(cb) : nullptr;
  CT* outy = kByValue ? reinterpret_cast<CT*>(cb + coord_b) : nullptr;
  if (!hit) {
    if (!ctx->ring_hint)
      IPA_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->ring_hint), 2 * sizeof(unsigned)));
    if (!same) {
      // another source / geometry: nothing known yet (a read-back of the old one still in
      // flight may land here once - a wrong hint, corrected by the next planning pass)
      ctx->ring_hint[0] = 0xffffffffu;
      ctx->ring_hint_skips = 0;
      memcpy(ctx->ring_hint_key, key, (size_t)kn * sizeof(double));
      ctx->ring_hint_n = kn;
    }
    IPA_HIP(ctx, hipMemsetAsync(plan->stats, 0, 2 * sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL((ring_plan_kernel<Coord, K>), dim3(gm.pairs), dim3(128), 0, ctx->stream, gm,
                       coord, sh, sw, tp, *plan, outx, outy);
    IPA_HIP(ctx, hipMemcpyAsync(ctx->ring_hint, plan->stats, sizeof(unsigned),
                                hipMemcpyDeviceToHost, ctx->stream));
    ctx->ring_hint[1] = (unsigned)gm.pairs;
    if (kByValue) {
      memcpy(ctx->plan_key, key, (size_t)kn * sizeof(double));
      ctx->plan_key_n = kn;
    }
  }
  if constexpr (kByValue) *kc = StoredCoord<CT>{outx, outy, (long)gm.dw};
  else *kc = coord;
  return 0;
}

}  // namespace ipa
