// group_stencil.hpp — remap -> K x K filter for BATCHES: one workgroup = one strip of
// kGW frames (one wave per frame).  Two things the per-frame marching wave of
// wave_stencil.hpp pays once per frame are paid once per GROUP here, and the bilinear taps stop
// going through the vector-memory path altogether:
//
//   * the frames of a batch share their geometry (one map / one lens model / one homography;
//     reference: LensDistortion.py:344-345 caches the maps, PerspectiveCorrection keeps one
//     homography).  A strip is 128 px wide and is walked in STEPS of two rows (256 samples,
//     four per lane).  For a chunk of kGW steps each wave turns ONE step of coordinates into
//     footprint records (first-tap address, x / y fraction) in LDS; after a barrier all waves
//     read every step's records.  Map loads, the f64 lens model, floor / fraction / bounds
//     arithmetic: 1/kGW per frame.
//   * the footprints of a step cover a few, nearly contiguous source rows.  Each wave keeps
//     the last kRR source rows of ITS frame (kRW px from a window origin xlo) in a wave-
//     private LDS ring, filled by coalesced row loads that are issued a chunk ahead (the
//     producers publish the row / column span of every step; lanes 0..3 of every wave turn
//     the four spans of a chunk into the ring schedule).  A bilinear sample is then two
//     ds_read2_b32 from ONE computed address (slot kRR mirrors slot 0, so the row below is
//     always + kRW) instead of four 64-lane gathers through the texture addresser - the unit
//     that bounds the per-frame kernel (profiles/r01_micro.txt).
//     Steps whose footprints do not fit the ring (row span > kRR - 2, column span > kRW - 8:
//     strong rotations, large magnification) carry global element offsets in their records and
//     are gathered as before; steps the ring cannot serve at that moment (rows running
//     backwards) rebuild the element offsets from the ring addresses; footprints touching the
//     source border are redone tap by tap through sample().  Same arithmetic and summation
//     order as wave_stencil.hpp / sampler.hpp: bit-identical to the per-frame kernels.
//
// Reference semantics: camera/LensDistortion.py:323-326 (cv2.remap INTER_LINEAR,
// BORDER_CONSTANT), camera/PerspectiveCorrection.py:377-378 followed by a dense K x K filter
// (filters/maskedConvolve.py:24-43 / scipy.ndimage.correlate), archetype
// camera/lens/estimateSystematicErrorLensCorrection.py:199-207.
#pragma once

#include <limits.h>

#include "wave_stencil.hpp"

namespace ipa {

constexpr int kGW = 4;                        // waves (= frames) per workgroup = steps per chunk
constexpr int kSW = 128;                      // strip width (2 px per lane in the filter stage)
constexpr int kRR = 8;                        // ring rows (power of two)
constexpr int kRW = 160;                      // ring row length (pixels)
constexpr int kRingFloats = (kRR + 1) * kRW;  // slot kRR mirrors slot 0
constexpr int kRecFloats = 3 * 256;           // per step: 256 slots, x fractions, y fractions
constexpr int kXRow = kSW + 2 * kRowPad;      // staged sample row (+ pad on both sides)
#ifndef IPA_GROUP_PEND
#define IPA_GROUP_PEND 10
#endif
constexpr int kPend = IPA_GROUP_PEND;         // source rows prefetched per chunk (registers)

// step flags (meta word 0)
enum : int {
  kStepSkip = 1,     // past the strip
  kStepRing = 4,     // records hold ring addresses (else element offsets)
  kStepSlow = 8,     // some footprints touch the source border
  kStepAny = 16,     // some footprint lies wholly inside the source
  kRow0Const = 32,   // row 0 / 1 of the step is a constant filter-border row
  kRow1Const = 64,
  kRow1Skip = 128,   // odd number of input rows: row 1 of the last step does not exist
  kStepServ = 256    // (planner) the ring serves this step
};

template <int K> struct group_geom {
  static constexpr int H = K / 2;
  static constexpr int HL = (H + 1) / 2;      // halo lanes per side (2 px per lane)
  static constexpr int OW = kSW - 4 * HL;     // output pixels per strip row
};

template <typename ST, typename Coord> struct GroupSrc {
  Coord coord;
  const char* src;       // frame 0 of the remap source
  long src_frame_bytes;
  unsigned src_bytes;
  int sh, sw, spitch;
  int border, q5;
  float cval;            // remap border value
  float ccval;           // filter border value
  int n_frames;
  int use_ring;          // 0: every step gathers (tuning / A-B knob)
};

// min / max over the 64 lanes (all active), result wave-uniform
template <bool MAX> __device__ __forceinline__ int wave_minmax(int v) {
#define IPA_MM(a, b) (MAX ? ((a) > (b) ? (a) : (b)) : ((a) < (b) ? (a) : (b)))
  int t;
  t = __builtin_amdgcn_update_dpp(v, v, 0x111 /*row_shr:1*/, 0xf, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x112 /*row_shr:2*/, 0xf, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x114 /*row_shr:4*/, 0xf, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x118 /*row_shr:8*/, 0xf, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false); v = IPA_MM(v, t);
  t = __builtin_amdgcn_update_dpp(v, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false); v = IPA_MM(v, t);
#undef IPA_MM
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void wave_span(int& xmn, int& xmx, int& ymn, int& ymx) {
  xmn = wave_minmax<false>(xmn);
  xmx = wave_minmax<true>(xmx);
  ymn = wave_minmax<false>(ymn);
  ymx = wave_minmax<true>(ymx);
}

// LDS-only workgroup barrier: __syncthreads() would also drain the vector-memory counter,
// i.e. wait for the prefetched source rows and the output stores at every chunk
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// one source row of the ring window (kRW px from element offset eo) in registers:
// 2 px per lane (128 px) + 1 px (32 px; lanes 32..63 duplicate the lanes 0..31)
template <typename ST> struct PendRow;
template <> struct PendRow<float> {
  float a[2];
  float b;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int eo, unsigned lane) {
    // range-checked per dword: columns left / right of the frame read neighbouring rows or 0,
    // never used (only footprints wholly inside the frame sample from the ring)
    auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (eo + 2 * (int)lane) << 2, 0, 0);
    a[0] = u2f(r[0]); a[1] = u2f(r[1]);
    b = u2f(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (eo + 128 + (int)(lane & 31u)) << 2, 0, 0));
  }
  __device__ __forceinline__ void write(float* row, unsigned lane) const {
    *reinterpret_cast<float2*>(row + 2u * lane) = float2{a[0], a[1]};
    row[128u + (lane & 31u)] = b;  // both halves of the wave store the same value
  }
  // column c at row[2 c]: one row of a row-pair-interleaved ring (ring_remap.hpp, Lanczos4)
  __device__ __forceinline__ void write_every_other(float* row, unsigned lane) const {
    row[4u * lane] = a[0];
    row[4u * lane + 2u] = a[1];
    row[2u * (128u + (lane & 31u))] = b;
  }
};
template <> struct PendRow<uint16_t> {
  unsigned a;
  unsigned b;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int eo, unsigned lane) {
    a = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (eo + 2 * (int)lane) << 1, 0, 0);
    b = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
        rsrc, (eo + 128 + (int)(lane & 31u)) << 1, 0, 0);
  }
  __device__ __forceinline__ void write(float* row, unsigned lane) const {
    *reinterpret_cast<float2*>(row + 2u * lane) = float2{(float)(a & 0xffffu), (float)(a >> 16)};
    row[128u + (lane & 31u)] = (float)b;
  }
};

template <typename ST, typename Coord, int K, bool STREAM> struct GroupKernel {
  using C = typename Coord::coord_t;
  using G = group_geom<K>;
  using Src = GroupSrc<ST, Coord>;
  static constexpr bool kMap = std::is_same<Coord, MapCoord>::value;
  static constexpr int kLead = G::H > kRowPad ? 4 : 0;

  struct Shared {
    float rec[2][kGW][kRecFloats];
    int meta[2][kGW][8];
    float ring[kGW][kRingFloats];
    float xrow[kLead + kGW * 2 * kXRow + kLead];
  };

  // columns of the strip a lane samples (lane-interleaved: xs + lane + 64 q), border-resolved
  struct SCols {
    int xs;       // first column of the strip (scalar)
    int uq[2];    // resolved sample columns, -1 = constant filter border (rim strips only)
  };

  // ---- coordinates of the 4 samples of a step (k = 2 * row + column group) ---------------
  template <bool FAST>
  static __device__ __forceinline__ void coords_of_step(const Src& g, const SCols& c, int vv0,
                                                        int vv1, C (&sx)[4], C (&sy)[4]) {
    const int lane = threadIdx.x & 63;
    const int v[2] = {vv0 < 0 ? 0 : vv0, vv1 < 0 ? 0 : vv1};
    if constexpr (kMap) {
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if constexpr (FAST) {
          const long o = (long)v[r] * g.coord.pitch + c.xs;  // scalar
          const float* rx = g.coord.mx + o;
          const float* ry = g.coord.my + o;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            sx[2 * r + q] = rx[(unsigned)lane + 64u * q];
            sy[2 * r + q] = ry[(unsigned)lane + 64u * q];
          }
        } else {
          const float* rx = g.coord.mx + (long)v[r] * g.coord.pitch;
          const float* ry = g.coord.my + (long)v[r] * g.coord.pitch;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const unsigned u = (unsigned)(c.uq[q] < 0 ? 0 : c.uq[q]);
            sx[2 * r + q] = rx[u];
            sy[2 * r + q] = ry[u];
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int r = k >> 1, q = k & 1;
        if constexpr (FAST) g.coord.get(c.xs + lane + 64 * q, v[r], sx[k], sy[k]);
        else g.coord.get(c.uq[q] < 0 ? 0 : c.uq[q], v[r], sx[k], sy[k]);
      }
    }
  }

  // ---- producer: one step of coordinates -> footprint records + span -----------------------
  // rowflags: kStepSkip / kRow0Const / kRow1Const / kRow1Skip of this step
  template <bool FAST>
  static __device__ __forceinline__ void produce_step(const Src& g, const SrcView& s,
                                                      const SCols& c, int rowflags,
                                                      const C (&sx)[4], const C (&sy)[4],
                                                      float* rec, int* meta) {
    const unsigned lane = threadIdx.x & 63u;
    if (rowflags & kStepSkip) {
      if (lane == 0) meta[0] = kStepSkip;
      return;
    }
    int ix0[4], iy0[4];
    float tx[4], ty[4];
    bool in[4];
    int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
    bool slow = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int r = k >> 1, q = k & 1;
      const bool ok = sx[k] > (C)-kCoordLimit && sx[k] < (C)kCoordLimit &&
                      sy[k] > (C)-kCoordLimit && sy[k] < (C)kCoordLimit;
      axis_frac<kLinear, float, C, 0>(s, ok ? sx[k] : (C)0, ix0[k], tx[k]);
      axis_frac<kLinear, float, C, 0>(s, ok ? sy[k] : (C)0, iy0[k], ty[k]);
      const bool inside = ok && ix0[k] >= 0 && iy0[k] >= 0 && ix0[k] + 2 <= s.w && iy0[k] + 2 <= s.h;
      // constant filter-border rows / columns are replaced by the consumer: not sampled
      bool live = !(rowflags & (r == 0 ? kRow0Const : (kRow1Const | kRow1Skip)));
      if constexpr (!FAST) live = live && c.uq[q] >= 0;
      in[k] = inside && live;
      slow = slow || (!inside && live);
      const int xl = in[k] ? ix0[k] : INT_MAX, xh = in[k] ? ix0[k] : INT_MIN;
      const int yl = in[k] ? iy0[k] : INT_MAX, yh = in[k] ? iy0[k] : INT_MIN;
      xmn = xl < xmn ? xl : xmn;
      xmx = xh > xmx ? xh : xmx;
      ymn = yl < ymn ? yl : ymn;
      ymx = yh > ymx ? yh : ymx;
    }
    wave_span(xmn, xmx, ymn, ymx);
    const bool any = xmn <= xmx;
    const bool any_slow = __builtin_amdgcn_ballot_w64(slow) != 0;
    const bool ring = g.use_ring &&
                      (!any || (xmx - xmn + 2 <= kRW - 8 && ymx - ymn + 2 <= kRR));
    float sl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int v;
      if (!in[k]) v = -1;
      else if (ring) v = ((iy0[k] & (kRR - 1)) * kRW + ix0[k]) << 2;  // byte address in the ring
      else v = __mul24(iy0[k], s.pitch) + ix0[k];                      // element offset in the frame
      sl[k] = __int_as_float(v);
    }
    *reinterpret_cast<float4*>(rec + 4u * lane) = float4{sl[0], sl[1], sl[2], sl[3]};
    *reinterpret_cast<float4*>(rec + 256 + 4u * lane) = float4{tx[0], tx[1], tx[2], tx[3]};
    *reinterpret_cast<float4*>(rec + 512 + 4u * lane) = float4{ty[0], ty[1], ty[2], ty[3]};
    if (lane == 0) {
      meta[0] = rowflags | (ring ? kStepRing : 0) | (any ? kStepAny : 0) | (any_slow ? kStepSlow : 0);
      meta[1] = xmn; meta[2] = xmx; meta[3] = ymn; meta[4] = ymx;
    }
  }

  // ---- K x K step on one staged row of 128 px: the chain of wave_run_strip, one pair / lane ----
  static __device__ __forceinline__ void filter_row(v2f (&acc)[K], const float* xr, unsigned lane,
                                                    const Weights<float, K * K>& wts,
                                                    kernarg_f32 wk) {
    const float* wp = xr + kRowPad - G::H + 2u * lane;
    v2f pair[K];
#pragma unroll
    for (int m = 0; m < K; m++) pair[m] = v2f{wp[m], wp[m + 1]};

#define IPA_LOAD_COEF_ROW(r)                                                                  \
  asm volatile("s_load_dwordx4 %0, %3, %4\n\ts_load_dwordx4 %1, %3, %5\n\t"                    \
               "s_load_dwordx4 %2, %3, %6"                                                     \
               : "=&s"(cc[r][0]), "=&s"(cc[r][1]), "=&s"(cc[r][2])                             \
               : "s"(wk), "n"((r) * 48), "n"((r) * 48 + 16), "n"((r) * 48 + 32))
    v4f cc[K][3];
    if constexpr (STREAM) IPA_LOAD_COEF_ROW(K - 1);
#undef IPA_LOAD_COEF_ROW

    static_for<0, K>([&](auto Ii) {
      constexpr int i = K - 1 - decltype(Ii)::value;
      if constexpr (STREAM) {
        if constexpr (i == K - 1)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cc[i][0]), "+s"(cc[i][1]), "+s"(cc[i][2]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+s"(cc[i][0]), "+s"(cc[i][1]), "+s"(cc[i][2]), "+v"(acc[i + 1]));
        if constexpr (i > 0) {
          asm volatile("s_load_dwordx4 %0, %4, %5\n\ts_load_dwordx4 %1, %4, %6\n\t"
                       "s_load_dwordx4 %2, %4, %7"
                       : "=&s"(cc[i - 1][0]), "=&s"(cc[i - 1][1]), "=&s"(cc[i - 1][2]),
                         "+v"(acc[i - 1])
                       : "s"(wk), "n"((i - 1) * 48), "n"((i - 1) * 48 + 16),
                         "n"((i - 1) * 48 + 32));
        }
      }
      if constexpr (STREAM) {
#pragma unroll
        for (int j = 0; j < K; j++) {
          const float w = cc[i][j >> 2][j & 3];
          const v2f w2 = v2f{w, w};
          if constexpr (i == 0) {
            acc[0] = j == 0 ? w2 * pair[0] : __builtin_elementwise_fma(w2, pair[j], acc[0]);
          } else {
            acc[i] = __builtin_elementwise_fma(w2, pair[j], j == 0 ? acc[i - 1] : acc[i]);
          }
        }
      } else {
        static_for<0, K>([&](auto Jj) {
          constexpr int j = decltype(Jj)::value;
          constexpr int n = i * K + j;
          constexpr int n0 = n & ~1, n1 = n0 + 1 < K * K ? n0 + 1 : n0;
          const v2f wp2 = v2f{wts.w[n0], wts.w[n1]};
          if constexpr (i == 0 && j == 0) acc[0] = pk_mul_coef<(n & 1)>(wp2, pair[0]);
          else if constexpr (j == 0) acc[i] = pk_fma_coef<(n & 1)>(wp2, pair[0], acc[i - 1]);
          else acc[i] = pk_fma_coef<(n & 1)>(wp2, pair[j], acc[i]);
        });
      }
      if constexpr (STREAM) __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- one strip of one group ---------------------------------------------------------------
  // Pipeline (chunk = kGW steps of 2 rows, ci = chunk being consumed):
  //   a. produce the records of chunk ci + 1 (this wave: one step; its coordinates were loaded
  //      two iterations ago)                                                   | barrier X
  //   b. consume chunk ci: steps software-pipelined by one (the taps of step d + 1 are issued
  //      before step d is blended and filtered); waits for no vector-memory result issued in
  //      this iteration
  //   c. plan: lanes 0..3 turn the spans of chunk ci + 1 into the ring schedule (which
  //      prefetched row goes in before which step)
  //   d. issue the coordinate loads of chunk ci + 3 and the source-row loads of chunk ci + 1
  //      (in this order: waiting for the rows never waits for younger loads)    | barrier Y
  template <bool FAST>
  static __device__ __forceinline__ void run_strip(const WaveParams& p, const Src& g,
                                                   const SrcView& s,
                                                   const Weights<float, K * K>& wts, kernarg_f32 wk,
                                                   const SCols& c, int xo, int y0, int nrows,
                                                   bool writer, bool active, float* dst,
                                                   Shared& sh, unsigned wave) {
    const int T = nrows + K - 1;            // input rows of the strip
    const int nsteps = (T + 1) / 2;
    const int nchunks = (nsteps + kGW - 1) / kGW;
    const unsigned lane = threadIdx.x & 63u;
    float* xp = sh.xrow + kLead + wave * 2 * kXRow;
    float* ringw = sh.ring[wave];

    auto row_of = [&](int t) -> int {
      if constexpr (FAST) return y0 - G::H + t;
      else return resolve_idx(y0 - G::H + t, p.dh, p.cby);
    };
    // rows / flags of step st (clamped to the strip)
    auto step_rows = [&](int st, int& vv0, int& vv1) -> int {
      int flags = 0;
      if (st >= nsteps) {
        flags = kStepSkip;
        st = nsteps - 1;
      }
      const int t0 = 2 * st;
      vv0 = row_of(t0);
      if (t0 + 1 < T) {
        vv1 = row_of(t0 + 1);
      } else {
        vv1 = vv0;
        flags |= kRow1Skip;
      }
      if constexpr (!FAST) {
        if (vv0 < 0) flags |= kRow0Const;
        if (vv1 < 0) flags |= kRow1Const;
      }
      return flags;
    };

    C cxa[4], cya[4];  // coordinates of the step this wave produces next
    C cxb[4], cyb[4];  // ... and of the one after it (in flight)
    PendRow<ST> pend[kPend];
    // ring state (uniform): rows [max(ringL, ringH - kRR), ringH) resident with window xlo
    int xlo = 0, ringH = 0, ringL = 0;
    bool ring_empty = true;
    // schedule of the chunk being consumed / of the next one
    int wmC[kGW], flC[kGW];
    int pbaseC = 0, pnC = 0;
#pragma unroll
    for (int d = 0; d < kGW; d++) { wmC[d] = 0; flC[d] = kStepSkip; }

    auto load_coords = [&](int ci, C (&ox)[4], C (&oy)[4]) {
      int vv0, vv1;
      step_rows(kGW * ci + (int)wave, vv0, vv1);
      coords_of_step<FAST>(g, c, vv0, vv1, ox, oy);
    };
    auto produce = [&](int ci) {
      int vv0, vv1;
      const int rf = step_rows(kGW * ci + (int)wave, vv0, vv1);
      if constexpr (!kMap) coords_of_step<FAST>(g, c, vv0, vv1, cxa, cya);
      produce_step<FAST>(g, s, c, rf, cxa, cya, sh.rec[ci & 1][wave], sh.meta[ci & 1][wave]);
    };
    // lanes 0..3 = the steps of chunk ci: spans -> ring schedule (all waves compute the same)
    auto plan = [&](int ci) {
      const unsigned dl = lane < (unsigned)kGW ? lane : (unsigned)kGW - 1u;
      const int* m = sh.meta[ci & 1][dl];
      const int f = m[0], xmin = m[1], xmax = m[2], ymin = m[3], ymax = m[4];
      const bool ring = (f & (kStepRing | kStepAny | kStepSkip)) == (kStepRing | kStepAny);
      bool fit = xmin >= xlo && xmax + 2 <= xlo + kRW;
      const unsigned rmask = (unsigned)__builtin_amdgcn_ballot_w64(ring) & 0xfu;
      const unsigned bad = (unsigned)__builtin_amdgcn_ballot_w64(ring && !fit) & 0xfu;
      if (rmask && (bad || ring_empty)) {
        // new column window (16-byte aligned origin, 4-7 px of slack on the left) and a fresh
        // start at the first ring step of this chunk
        int xm = INT_MAX;
#pragma unroll
        for (int d = 0; d < kGW; d++) {
          const int v = __builtin_amdgcn_readlane(xmin, d);
          if ((rmask >> d) & 1u) xm = v < xm ? v : xm;
        }
        xlo = (xm & ~3) - 4;
        const int fd = __builtin_ctz(rmask);
        int y = 0;
#pragma unroll
        for (int d = 0; d < kGW; d++) {
          const int v = __builtin_amdgcn_readlane(ymin, d);
          if (d == fd) y = v;
        }
        ringH = ringL = y;
        ring_empty = false;
        fit = xmin >= xlo && xmax + 2 <= xlo + kRW;
      }
      // rows each step needs up to (exclusive), running maximum over the chunk
      const int need = (ring && fit) ? ymax + 2 : INT_MIN;
      int hd = need > ringH ? need : ringH;
      int t;
      t = __builtin_amdgcn_update_dpp(hd, hd, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
      hd = t > hd ? t : hd;
      t = __builtin_amdgcn_update_dpp(hd, hd, 0x112 /*row_shr:2*/, 0xf, 0xf, false);
      hd = t > hd ? t : hd;
      int hprev = __builtin_amdgcn_update_dpp(hd, hd, 0x111, 0xf, 0xf, false);
      hprev = lane == 0 ? ringH : hprev;
      const int first = hprev > hd - kRR ? hprev : hd - kRR;  // a jump ahead loads the last kRR only
      const int cnt = hd - first;
      const int shift = first - ringH;
      const int lowest = ringL > hd - kRR ? ringL : hd - kRR;
      const bool serv = ring && fit && ymin >= lowest && hd - ringH <= 24;
      const int wm = (cnt > 0 && shift < 24) ? (int)(((1u << cnt) - 1u) << shift) : 0;
      const int fl = f | (serv ? kStepServ : 0);
      pbaseC = ringH;
#pragma unroll
      for (int d = 0; d < kGW; d++) {
        wmC[d] = __builtin_amdgcn_readlane(wm, d);
        flC[d] = __builtin_amdgcn_readlane(fl, d);
      }
      const int hnew = __builtin_amdgcn_readlane(hd, kGW - 1);
      if (hnew - ringH > 24) ring_empty = true;  // absurd jump: served by gathers, start over
      pnC = hnew - ringH < kPend ? hnew - ringH : kPend;
      ringH = hnew;
    };
    auto ring_slot = [&](int y) -> float* { return ringw + (y & (kRR - 1)) * kRW; };

    if constexpr (kMap) {
      load_coords(0, cxa, cya);
      if (nchunks > 1) load_coords(1, cxb, cyb);
    }

    v2f acc[K];
#pragma unroll 1
    for (int ci = -1; ci < nchunks; ci++) {
      // a. records of the next chunk (other buffer)
      if (ci + 1 < nchunks) produce(ci + 1);
      lds_barrier();  // X: records + spans of chunk ci + 1 visible

      // b. consume chunk ci (window xlo, rows pend[j] = pbaseC + j)
      if (active && ci >= 0) {
        const int cb = ci & 1;
        float4 rs4[kGW], rx4[kGW], ry4[kGW];  // records of the steps in flight
        float v[2][4][2][2];                  // taps of two steps (even / odd d)

        auto load_rec = [&](auto Dd) {
          constexpr int d = decltype(Dd)::value;
          if (flC[d] & kStepSkip) return;
          const float* r = sh.rec[cb][d];
          rs4[d] = *reinterpret_cast<const float4*>(r + 4u * lane);
          rx4[d] = *reinterpret_cast<const float4*>(r + 256 + 4u * lane);
          ry4[d] = *reinterpret_cast<const float4*>(r + 512 + 4u * lane);
        };
        // stage A of a step: make its source rows resident, issue its taps
        auto stage_a = [&](auto Dd) {
          constexpr int d = decltype(Dd)::value;
          const int fl = flC[d];
          if (fl & kStepSkip) return;
          int sl[4] = {__float_as_int(rs4[d].x), __float_as_int(rs4[d].y),
                       __float_as_int(rs4[d].z), __float_as_int(rs4[d].w)};
          // source rows scheduled in front of this step (also when it is not served itself:
          // later steps count on them)
          const int wm = wmC[d];
          if (wm) {
#pragma unroll
            for (int j = 0; j < kPend; j++) {
              if ((wm >> j) & 1) {
                const int y = pbaseC + j;
                pend[j].write(ring_slot(y), lane);
                if ((y & (kRR - 1)) == 0) pend[j].write(ringw + kRR * kRW, lane);
              }
            }
            if (wm >> kPend) {  // more rows than the prefetch holds: load them now
#pragma unroll 1
              for (int j = kPend; (wm >> j) != 0; j++) {
                if ((wm >> j) & 1) {
                  const int y = pbaseC + j;
                  PendRow<ST> q;
                  q.load(s.rsrc, __mul24(y, s.pitch) + xlo, lane);
                  q.write(ring_slot(y), lane);
                  if ((y & (kRR - 1)) == 0) q.write(ringw + kRR * kRW, lane);
                }
              }
            }
          }
          if (fl & kStepServ) {
            // taps: two 2-dword reads per footprint row from one computed address; reads beyond
            // the allocation return 0, so unused (-1) slots need no guard
            const char* rb = reinterpret_cast<const char*>(ringw) - 4 * xlo;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const float* tp = reinterpret_cast<const float*>(rb + sl[k]);
              v[d & 1][k][0][0] = tp[0];
              v[d & 1][k][0][1] = tp[1];
              v[d & 1][k][1][0] = tp[kRW];
              v[d & 1][k][1][1] = tp[kRW + 1];
            }
          } else {
            if (fl & kStepRing) {
              // ring addresses the ring cannot serve now: back to element offsets
              const int* m = sh.meta[cb][d];
              const int xmin = __builtin_amdgcn_readfirstlane(m[1]);
              const int ymin = __builtin_amdgcn_readfirstlane(m[3]);
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const unsigned u = (unsigned)((sl[k] >> 2) - xmin);   // slot * kRW + ix0 - xmin
                static_assert(kRW == 160, "the reciprocal below is 1 / kRW");
                const unsigned slot = (unsigned)(((unsigned long)u * 52429ul) >> 23);  // u / 160
                const int ix = (int)(u - slot * kRW) + xmin;
                const int iy = ymin + (int)((slot - (unsigned)ymin) & (kRR - 1));
                sl[k] = sl[k] < 0 ? -1 : __mul24(iy, s.pitch) + ix;
              }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int e = sl[k] < 0 ? 0 : sl[k];
              TapLoad<ST, float>::template row<2>(s, e, v[d & 1][k][0]);
              TapLoad<ST, float>::template row<2>(s, e + s.pitch, v[d & 1][k][1]);
            }
          }
        };
        // stage B of a step: blend, stage the two rows, K x K steps, stores
        auto stage_b = [&](auto Dd) {
          constexpr int d = decltype(Dd)::value;
          const int fl = flC[d];
          if (fl & kStepSkip) return;
          const int st = kGW * ci + d;
          const float tx[4] = {rx4[d].x, rx4[d].y, rx4[d].z, rx4[d].w};
          const float ty[4] = {ry4[d].x, ry4[d].y, ry4[d].z, ry4[d].w};
          float cur[4];
          // the chain of sample() / batch_blend_one()
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float wx0 = 1.f - tx[k], wy0 = 1.f - ty[k];
            float r0 = wx0 * v[d & 1][k][0][0];
            r0 = ipa_fma(tx[k], v[d & 1][k][0][1], r0);
            float r1 = wx0 * v[d & 1][k][1][0];
            r1 = ipa_fma(tx[k], v[d & 1][k][1][1], r1);
            float o = wy0 * r0;
            cur[k] = ipa_fma(ty[k], r1, o);
          }
          if (fl & kStepSlow) {
            // footprints touching the source border (rare): tap by tap
            const int sl[4] = {__float_as_int(rs4[d].x), __float_as_int(rs4[d].y),
                               __float_as_int(rs4[d].z), __float_as_int(rs4[d].w)};
            int vv0, vv1;
            step_rows(st, vv0, vv1);
            C sx[4], sy[4];
            coords_of_step<FAST>(g, c, vv0, vv1, sx, sy);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              bool live = !(fl & ((k >> 1) == 0 ? kRow0Const : (kRow1Const | kRow1Skip)));
              if constexpr (!FAST) live = live && c.uq[k & 1] >= 0;
              if (sl[k] < 0 && live) cur[k] = sample<ST, kLinear, C>(s, sx[k], sy[k], g.cval);
            }
          }
          if constexpr (!FAST) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const bool cst = (fl & ((k >> 1) == 0 ? kRow0Const : kRow1Const)) || c.uq[k & 1] < 0;
              cur[k] = cst ? g.ccval : cur[k];
            }
          }
#pragma unroll
          for (int k = 0; k < 4; k++)
            xp[(k >> 1) * kXRow + kRowPad + 64u * (k & 1) + lane] = cur[k];
          __builtin_amdgcn_wave_barrier();

#pragma unroll
          for (int r = 0; r < 2; r++) {
            if (r == 1 && (fl & kRow1Skip)) break;
            filter_row(acc, xp + r * kXRow, lane, wts, wk);
            const int o = 2 * st + r - (K - 1);
            if (o >= 0 && o < nrows && writer) {
              const v2f q = acc[K - 1];
              if constexpr (FAST) {
                float* rows_ = dst + ((long)(y0 + o) * p.dpitch + c.xs);  // scalar base
                __builtin_nontemporal_store(q.x, rows_ + 2u * lane);
                __builtin_nontemporal_store(q.y, rows_ + 2u * lane + 1);
              } else {
                float* orow = dst + (long)(y0 + o) * p.dpitch + xo;
                orow[0] = q.x;
                if (xo + 1 < p.dw) orow[1] = q.y;
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        };

        load_rec(std::integral_constant<int, 0>{});
        if constexpr (kGW > 1) load_rec(std::integral_constant<int, 1>{});
        stage_a(std::integral_constant<int, 0>{});
        static_for<0, kGW>([&](auto Dd) {
          constexpr int d = decltype(Dd)::value;
          if constexpr (d + 2 < kGW) load_rec(std::integral_constant<int, d + 2>{});
          if constexpr (d + 1 < kGW) stage_a(std::integral_constant<int, d + 1>{});
          stage_b(Dd);
        });
      }

      // c. ring schedule of the next chunk (its spans became visible at X)
#pragma unroll
      for (int d = 0; d < kGW; d++) { wmC[d] = 0; flC[d] = kStepSkip; }
      pnC = 0;
      if (active && ci + 1 < nchunks) plan(ci + 1);

      // d. coordinates two chunks ahead, then the source rows of the next chunk
      if constexpr (kMap) {
#pragma unroll
        for (int k = 0; k < 4; k++) { cxa[k] = cxb[k]; cya[k] = cyb[k]; }
        if (ci + 3 < nchunks) load_coords(ci + 3, cxb, cyb);
      }
      if (active && ci + 1 < nchunks) {
#pragma unroll
        for (int j = 0; j < kPend; j++)
          if (j < pnC) pend[j].load(s.rsrc, __mul24(pbaseC + j, s.pitch) + xlo, lane);
      }
      lds_barrier();  // Y: everyone is done with the records of chunk ci
    }
  }

  static __device__ __forceinline__ void body(const WaveParams& p, const Src& g,
                                              const Weights<float, K * K>& wts, kernarg_f32 wk) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // frame groups of one strip are neighbours in the XCD-contiguous block order: they read the
    // same map rows at the same time (one L2 fetch per strip, not per group)
    const unsigned groups = ((unsigned)g.n_frames + kGW - 1) / kGW;
    unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned grp = b % groups;
    const unsigned sid = b / groups;
    const unsigned frame = grp * kGW + wave;
    const bool active = frame < (unsigned)g.n_frames;
    const int syi = (int)(sid / (unsigned)p.strips_x), sxi = (int)sid - syi * p.strips_x;

    SrcView s;
    s.rsrc = make_rsrc(g.src + (long)(active ? frame : 0u) * g.src_frame_bytes, g.src_bytes);
    s.h = g.sh; s.w = g.sw; s.pitch = g.spitch;
    s.border = g.border; s.q5 = g.q5; s.cubic_a = 0.f; s.lanczos = nullptr;

    const int xs = sxi * G::OW - 2 * G::HL;
    SCols c;
    c.xs = xs;
    const int xo = xs + lane * 2;  // first of the lane's two output columns
    const int y0 = syi * p.strip_h;
    const int nrows = p.dh - y0 < p.strip_h ? p.dh - y0 : p.strip_h;
    const bool writer = lane >= G::HL && lane < 64 - G::HL && xo < p.dw;
    float* dst = reinterpret_cast<float*>(p.dst) + (long)(active ? frame : 0u) * p.dst_frame_elems;

    // vec_out: 16-byte aligned rows and frames -> the 8-byte stores of a fast strip are aligned
    const bool fast = p.vec_out && xs >= 0 && xs + kSW <= p.dw && y0 - G::H >= 0 &&
                      y0 - G::H + nrows + K - 1 <= p.dh;
    if (fast) {
      c.uq[0] = xs + lane;
      c.uq[1] = xs + lane + 64;
      run_strip<true>(p, g, s, wts, wk, c, xo, y0, nrows, writer, active, dst, sh, wave);
    } else {
      c.uq[0] = resolve_idx(xs + lane, p.dw, p.cbx);
      c.uq[1] = resolve_idx(xs + lane + 64, p.dw, p.cbx);
      run_strip<false>(p, g, s, wts, wk, c, xo, y0, nrows, writer, active, dst, sh, wave);
    }
  }
};

}  // namespace ipa
