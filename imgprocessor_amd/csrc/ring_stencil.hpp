// ring_stencil.hpp — bilinear remap -> K x K filter with the taps read from LDS, for the
// strips where that is possible without any special case ("clean" strips), decided ONCE per
// call by a small planning kernel instead of inside the hot loop.
//
//   ring_plan_kernel   one wave per 128-px strip walks the strip's coordinates once (the same
//                      maps / lens model / homography every frame of the batch uses) and
//                      decides whether the strip is CLEAN: interior of the output (no filter
//                      border), every bilinear footprint wholly inside the source, the
//                      footprints of every two-row step within kRR source rows, of the whole
//                      strip within kRW columns, source rows only moving forward and at most
//                      kRingMaxNew new rows per step.  For a clean strip it records the ring
//                      schedule: window origin, first source row, new rows per step (4 bits
//                      each).  Two neighbouring strips form one 256-px strip of the per-frame
//                      kernels; `pair_clean` is what both kernels consult.
//   ring_kernel        the hot loop, clean strips only: one wave per (strip, frame), no
//                      workgroup barrier, no border or validity test, no planning.  Per step
//                      (2 rows x 128 px = 4 samples per lane): the scheduled source rows
//                      arrive by coalesced row loads issued one step ahead and go into a wave-
//                      private LDS ring (slot kRR mirrors slot 0); the coordinates of the step
//                      after next are prefetched; a sample is two ds_read2_b32 from one
//                      computed address.  Per 256 samples the vector-memory path sees 8 map
//                      dwords, ~2 row loads and 2 stores instead of 8 + 16 gathers + 1.
//   wave_stencil_kernel (wave_stencil.hpp) runs the remaining strips (frame rim, footprints on
//                      the source border, strong rotation ...) exactly as before; it skips
//                      the strips marked in `pair_clean`.
//
// Arithmetic, rounding and summation order are those of sample() / wave_run_strip(): the two
// kernels produce identical bits, so the split is invisible in the result.
//
// Reference semantics: camera/LensDistortion.py:323-326 (cv2.remap INTER_LINEAR,
// BORDER_CONSTANT), camera/PerspectiveCorrection.py:377-378 followed by a dense K x K filter
// (filters/maskedConvolve.py:24-43 / scipy.ndimage.correlate).
#pragma once

#include "group_stencil.hpp"

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

template <typename ST, typename Coord> struct RingSrc {
  Coord coord;
  const char* src;       // frame 0 of the remap source
  long src_frame_bytes;
  unsigned src_bytes;
  int sh, sw, spitch;
  int n_frames;
  int ablate;            // measurement only (context knob ring_ablate): 1 no stores, 8 no filter
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
  // kernel anyway - strong rotation, footprints outside the source - skips the ring path.  A
  // hint only: the results are the same bits either way.
  const bool same = ctx->ring_hint_n == kn &&
                    memcmp(ctx->ring_hint_key, key, (size_t)kn * sizeof(double)) == 0;
  // (a stale "few clean strips" - a read-back that landed after the key changed, maps rewritten
  // at the same address - would otherwise keep the ring path off for as long as the key stays:
  // every 16th skipped call plans again and refreshes the hint)
  if (same && ctx->ring_hint && ctx->ring_hint[1] == (unsigned)gm.pairs &&
      ctx->ring_hint[0] != 0xffffffffu && 2u * ctx->ring_hint[0] < (unsigned)gm.pairs &&
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

// --------------------------------------------------------------------------- hot loop --
template <typename ST, typename Coord, int K> struct RingKernel {
  using C = typename Coord::coord_t;
  using G = group_geom<K>;
  using Src = RingSrc<ST, Coord>;
  static constexpr int kWaves = 4;
  static constexpr int kLead = G::H > kRowPad ? 4 : 0;

  struct Shared {
    float ring[kWaves][kRingFloats];
    float xrow[kLead + kWaves * 2 * kXRow + kLead];
  };

  // the K x K step on one staged row (one pixel pair per lane): the chain of wave_run_strip
  static __device__ __forceinline__ void filter_row(v2f (&acc)[K], const float* xr, unsigned lane,
                                                    const Weights<float, K * K>& wts) {
    const float* wp = xr + kRowPad - G::H + 2u * lane;
    v2f pair[K];
#pragma unroll
    for (int m = 0; m < K; m++) pair[m] = v2f{wp[m], wp[m + 1]};
    static_for<0, K>([&](auto Ii) {
      constexpr int i = K - 1 - decltype(Ii)::value;
      static_for<0, K>([&](auto Jj) {
        constexpr int j = decltype(Jj)::value;
        constexpr int n = i * K + j;
        constexpr int n0 = n & ~1, n1 = n0 + 1 < K * K ? n0 + 1 : n0;
        const v2f wp2 = v2f{wts.w[n0], wts.w[n1]};
        if constexpr (i == 0 && j == 0) acc[0] = pk_mul_coef<(n & 1)>(wp2, pair[0]);
        else if constexpr (j == 0) acc[i] = pk_fma_coef<(n & 1)>(wp2, pair[0], acc[i - 1]);
        else acc[i] = pk_fma_coef<(n & 1)>(wp2, pair[j], acc[i]);
      });
    });
  }

  static __device__ __forceinline__ void body(const WaveParams& p, const RingGeom& gm,
                                              const Src& g, const RingPlan& plan,
                                              const Weights<float, K * K>& wts) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the frames of one strip are neighbours in the XCD-contiguous block order and share a
    // workgroup: they read the same map rows at the same time
    const unsigned groups = ((unsigned)g.n_frames + kWaves - 1) / kWaves;
    const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned grp = b % groups;
    const unsigned sid = b / groups;
    const unsigned frame = grp * kWaves + wave;
    if (frame >= (unsigned)g.n_frames) return;
    const int syi = (int)(sid / (unsigned)gm.strips_x), sxi = (int)sid - syi * gm.strips_x;
    if (!plan.pair_clean[syi * gm.pairs_x + (sxi >> 1)]) return;  // wave_stencil_kernel's strip
    const int4 info = plan.info[sid];
    // the plan is wave-uniform: keep it in scalar registers (scalar branches, not exec masks)
    const int xlo = __builtin_amdgcn_readfirstlane(info.y);
    const unsigned* cwp = plan.cnts + (size_t)sid * kPlanWords;

    const int xs = sxi * G::OW - 2 * G::HL;
    const int y0 = syi * gm.strip_h;
    const int nrows = gm.dh - y0 < gm.strip_h ? gm.dh - y0 : gm.strip_h;
    const int T = nrows + K - 1;
    const int nsteps = (T + 1) / 2;
    const bool writer = lane >= (unsigned)G::HL && lane < 64u - G::HL;
    // output row pointer of this strip (advanced row by row: no 64-bit multiply per store)
    float* dst = reinterpret_cast<float*>(p.dst) + (long)frame * p.dst_frame_elems +
                 ((long)y0 * p.dpitch + xs);
    const __amdgpu_buffer_rsrc_t rsrc =
        make_rsrc(g.src + (long)frame * g.src_frame_bytes, g.src_bytes);
    SrcView s;  // only what axis_frac reads
    s.q5 = 0;
    float* xp = sh.xrow + kLead + wave * 2 * kXRow;
    float* ringw = sh.ring[wave];
    auto put_row = [&](const PendRow<ST>& r, int y) {
      r.write(ringw + (y & (kRR - 1)) * kRW, lane);
      if ((y & (kRR - 1)) == 0) r.write(ringw + kRR * kRW, lane);
    };

    // ring rows of step 0: resident before the loop
    int hres = __builtin_amdgcn_readfirstlane(info.z);  // rows [.., hres) are in the ring
    {
      const int n0 = __builtin_amdgcn_readfirstlane(info.w);
#pragma unroll 1
      for (int j = 0; j < n0; j += 2) {
        PendRow<ST> a, bq;
        a.load(rsrc, __mul24(hres + j, g.spitch) + xlo, lane);
        if (j + 1 < n0) bq.load(rsrc, __mul24(hres + j + 1, g.spitch) + xlo, lane);
        put_row(a, hres + j);
        if (j + 1 < n0) put_row(bq, hres + j + 1);
      }
      hres += n0;
    }
    // coordinates of step 0
    C cx[4], cy[4];
    auto step_coords = [&](int st) {
      const int sc = st < nsteps ? st : nsteps - 1;
      const int v0 = y0 - G::H + 2 * sc;
      const int v1 = 2 * sc + 1 < T ? v0 + 1 : v0;
      ring_coords<Coord>(g.coord, xs, v0, v1, cx, cy);
    };
    step_coords(0);
    // the strip's row counts: 8 words, made scalar once
    unsigned words[kPlanWords];
#pragma unroll
    for (int i = 0; i < kPlanWords; i++) words[i] = __builtin_amdgcn_readfirstlane(cwp[i]);
    auto word_of = [&](int i) -> unsigned {
      unsigned wv = words[0];
#pragma unroll
      for (int q = 1; q < kPlanWords; q++) wv = i == q ? words[q] : wv;
      return wv;
    };
    PendRow<ST> pend[kRingMaxNew];
    int cnt = 0;  // rows in pend (for the step about to run)

    // Every vector-memory result a step needs (its source rows, its coordinates) was requested
    // at the start of the PREVIOUS step, and the previous step's output rows are stored at the
    // start of this one: whatever the step waits for at its top was issued a whole step ago.
    // (Loads and stores share the vmcnt counter but complete out of order with respect to each
    // other, so with a store in flight the compiler can only wait with vmcnt(0): deeper request
    // queues in registers do not help a wave that also stores.)
    v2f acc[K];
    v2f hold[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};  // output rows of the previous step
    int hold_o = INT_MIN;  // output row of hold[0] (INT_MIN: nothing held)
    int hold_n = 0;
    auto flush = [&]() {
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int o = hold_o + r;
        if (r < hold_n && o >= 0) {
          if (writer && !(g.ablate & 1)) {
            __builtin_nontemporal_store(hold[r].x, dst + 2u * lane);
            __builtin_nontemporal_store(hold[r].y, dst + 2u * lane + 1);
          }
          dst += p.dpitch;  // rows are flushed in order, o = 0, 1, 2, ...
        }
      }
      hold_n = 0;
    };
#pragma unroll 1
    for (int st = 0; st < nsteps; st++) {
      // 1. the rows requested during the previous step go into the ring; footprints of this step
      // (cnt / hres are wave-uniform; say so, or the tests below become exec-mask sequences)
      cnt = __builtin_amdgcn_readfirstlane(cnt);
      hres = __builtin_amdgcn_readfirstlane(hres);
#pragma unroll
      for (int j = 0; j < kRingMaxNew; j++)
        if (j < cnt) put_row(pend[j], hres + j);
      hres += cnt;
      float tx[4], ty[4];
      int ad[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int ix0, iy0;
        axis_frac<kLinear, float, C, 0>(s, cx[k], ix0, tx[k]);
        axis_frac<kLinear, float, C, 0>(s, cy[k], iy0, ty[k]);
        ad[k] = (__mul24(iy0 & (kRR - 1), kRW) + (ix0 - xlo)) << 2;
      }
      // (the compiler must not sink the footprint arithmetic below the requests of step 3: its
      // wait for the coordinates would then also wait for those requests)
#pragma unroll
      for (int k = 0; k < 4; k++)
        asm volatile("" : "+v"(ad[k]), "+v"(tx[k]), "+v"(ty[k]) : : "memory");
      // 2. output rows of the previous step
      flush();
      // 3. requests for the next step: its source rows and its coordinates
      const int sn = st + 1;
      cnt = sn < nsteps ? (int)((word_of(sn >> 3) >> (4 * (sn & 7))) & 15u) : 0;
      cnt = __builtin_amdgcn_readfirstlane(cnt);
#pragma unroll
      for (int j = 0; j < kRingMaxNew; j++)
        if (j < cnt) pend[j].load(rsrc, __mul24(hres + j, g.spitch) + xlo, lane);
      step_coords(sn);
      __builtin_amdgcn_sched_barrier(0);

      // 4. taps from the ring, blend (the chain of sample())
      __builtin_amdgcn_wave_barrier();
      float cur[4];
      const char* rb = reinterpret_cast<const char*>(ringw);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float* tp = reinterpret_cast<const float*>(rb + ad[k]);
        const float v00 = tp[0], v01 = tp[1], v10 = tp[kRW], v11 = tp[kRW + 1];
        const float wx0 = 1.f - tx[k], wy0 = 1.f - ty[k];
        float r0 = wx0 * v00;
        r0 = ipa_fma(tx[k], v01, r0);
        float r1 = wx0 * v10;
        r1 = ipa_fma(tx[k], v11, r1);
        const float o = wy0 * r0;
        cur[k] = ipa_fma(ty[k], r1, o);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) xp[(k >> 1) * kXRow + kRowPad + 64u * (k & 1) + lane] = cur[k];
      __builtin_amdgcn_wave_barrier();

      // 5. K x K steps; the finished rows are held for the next step's flush
      hold_o = 2 * st - (K - 1);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (2 * st + r >= T) break;
        if (!(g.ablate & 8)) filter_row(acc, xp + r * kXRow, lane, wts);
        hold[r] = acc[K - 1];
        hold_n = r + 1;
      }
      __builtin_amdgcn_wave_barrier();
    }
    flush();
  }
};

template <typename ST, typename Coord, int K>
__global__ void __launch_bounds__(256)
ring_kernel(WaveParams p, RingGeom gm, RingSrc<ST, Coord> g, RingPlan plan,
            Weights<float, K * K> wts) {
  RingKernel<ST, Coord, K>::body(p, gm, g, plan, wts);
}

}  // namespace ipa
