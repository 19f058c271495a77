// group_stencil.hpp — remap -> K x K filter for BATCHES: one workgroup = one strip of
// kGW frames (one wave per frame).  Two things the per-frame marching wave of
// wave_stencil.hpp pays once per frame are paid once per GROUP here, and the bilinear taps stop
// going through the vector-memory path altogether:
//
//   * the frames of a batch share their geometry (one map / one lens model / one homography,
//     reference: LensDistortion.py:344-345 caches the maps, PerspectiveCorrection keeps one
//     homography).  For a chunk of kGW rows each wave turns ONE row of coordinates into
//     footprint records (first-tap address, x/y fraction) and puts them into LDS; after a
//     barrier all waves read every row's records.  Map loads, the f64 lens model, floor /
//     fraction / bounds arithmetic: 1/kGW per frame.
//   * the footprints of a 256-px output row cover a few, nearly contiguous source rows.  Each
//     wave keeps the last kRR source rows of ITS frame (kRW px from a window origin xlo) in a
//     wave-private LDS ring, filled with coalesced 16-byte row loads that are issued one chunk
//     ahead (the producing waves also publish the row span of every output row, so all waves
//     know which source rows the next chunk needs).  A bilinear sample is then four
//     ds_read_b32 at immediate offsets from ONE address (slot 8 of the ring mirrors slot 0, so
//     the row below is always +kRW) instead of four 64-lane gathers through the texture
//     addresser - the unit that bounds the per-frame kernel (profiles/r01_micro.txt).
//     Rows whose footprints do not fit the ring (span > kRR - 2 rows or > kRW - 8 columns,
//     e.g. strong rotations) carry global element offsets in their records instead and are
//     gathered as before; footprints touching the source border are redone tap by tap through
//     sample().  Same arithmetic and summation order as wave_stencil.hpp / sampler.hpp:
//     results are bit-identical to the per-frame kernels.
//
// Reference semantics: camera/LensDistortion.py:323-326 (cv2.remap INTER_LINEAR,
// BORDER_CONSTANT), camera/PerspectiveCorrection.py:377-378 followed by a dense K x K filter
// (filters/maskedConvolve.py:24-43 / scipy.ndimage.correlate), archetype
// camera/lens/estimateSystematicErrorLensCorrection.py:199-207.
#pragma once

#include <limits.h>

#include "wave_stencil.hpp"

namespace ipa {

constexpr int kGW = 4;                    // waves (= frames) per workgroup = rows per chunk
constexpr int kRR = 8;                    // ring rows (power of two)
constexpr int kRW = 320;                  // ring row length (pixels)
constexpr int kRingFloats = (kRR + 1) * kRW;  // slot kRR mirrors slot 0
constexpr int kRecFloats = 3 * 256;       // per row: 256 slots, 256 x-fractions, 256 y-fractions
#ifndef IPA_GROUP_PEND
#define IPA_GROUP_PEND 5
#endif
constexpr int kPend = IPA_GROUP_PEND;     // source rows prefetched per chunk (registers)

enum : int { kRowSkip = 1, kRowConst = 2, kRowRing = 4, kRowSlow = 8, kRowAny = 16 };

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
  int use_ring;          // 0: every row gathers (tuning / A-B knob)
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

// LDS-only workgroup barrier: __syncthreads() would also drain the vector-memory counter,
// i.e. wait for the prefetched source rows and the output stores at every chunk
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// one source row of the ring window (kRW px from element offset eo) in registers
template <typename ST> struct PendRow;
template <> struct PendRow<float> {
  float a[4];
  float b;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int eo, unsigned lane) {
    // range-checked per dword: columns left / right of the frame read neighbouring rows or 0,
    // never used (only footprints wholly inside the frame sample from the ring)
    auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (eo + 4 * (int)lane) << 2, 0, 0);
    a[0] = u2f(r[0]); a[1] = u2f(r[1]); a[2] = u2f(r[2]); a[3] = u2f(r[3]);
    b = u2f(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (eo + 256 + (int)lane) << 2, 0, 0));
  }
  __device__ __forceinline__ void write(float* row, unsigned lane) const {
    *reinterpret_cast<float4*>(row + 4u * lane) = float4{a[0], a[1], a[2], a[3]};
    row[256u + lane] = b;
  }
};
template <> struct PendRow<uint16_t> {
  unsigned a[2];
  unsigned b;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int eo, unsigned lane) {
    auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (eo + 4 * (int)lane) << 1, 0, 0);
    a[0] = r[0]; a[1] = r[1];
    b = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (eo + 256 + 2 * (int)(lane & 31u)) << 1, 0, 0);
  }
  __device__ __forceinline__ void write(float* row, unsigned lane) const {
    *reinterpret_cast<float4*>(row + 4u * lane) =
        float4{(float)(a[0] & 0xffffu), (float)(a[0] >> 16), (float)(a[1] & 0xffffu),
               (float)(a[1] >> 16)};
    if (lane < 32u)
      *reinterpret_cast<float2*>(row + 256u + 2u * lane) =
          float2{(float)(b & 0xffffu), (float)(b >> 16)};
  }
};

// state of a wave's ring: rows [lo, hi) of the frame are resident, columns [xlo, xlo + kRW)
struct RingState {
  int xlo, lo, hi;
};

// The ring rule, shared by the consumer and by the planner that runs one chunk ahead of it:
// a row needs source rows [ymin, ymax + 2) and columns [xmin, xmax + 1].  Returns the first row
// that has to be loaded (rows [first, ymax + 2) are then appended).
__device__ __forceinline__ int ring_advance(RingState& r, int xmin, int xmax, int ymin, int ymax,
                                            bool& restarted) {
  restarted = false;
  if (r.lo == r.hi || xmin < r.xlo || xmax + 2 > r.xlo + kRW) {
    // new column window (16-byte aligned origin, 4-7 px of slack on the left): start over
    r.xlo = (xmin & ~3) - 4;
    r.lo = r.hi = ymin;
    restarted = true;
  } else if (ymin < r.lo || ymin > r.hi) {
    r.lo = r.hi = ymin;
    restarted = true;
  }
  const int first = r.hi;
  const int need_hi = ymax + 2;
  if (need_hi > r.hi) {
    r.hi = need_hi;
    r.lo = r.lo > r.hi - kRR ? r.lo : r.hi - kRR;
  }
  return first;
}

template <typename ST, typename Coord, int K, bool STREAM> struct GroupKernel {
  using C = typename Coord::coord_t;
  using G = wave_geom<K>;
  using Src = GroupSrc<ST, Coord>;
  static constexpr bool kMap = std::is_same<Coord, MapCoord>::value;
  static constexpr int kLead = G::H > kRowPad ? 4 : 0;

  struct Shared {
    float rec[2][kGW][kRecFloats];
    int meta[2][kGW][8];
    float ring[kGW][kRingFloats];
    float xrow[kLead + kGW * kRowStride + kLead];
  };

  // ---- coordinates of the 4 lane-interleaved samples of one row -------------------------
  template <bool FAST>
  static __device__ __forceinline__ void coords_of_row(const Src& g, const Cols& c, int vv,
                                                       C (&sx)[4], C (&sy)[4]) {
    const int lane = threadIdx.x & 63;
    const int v = (!FAST && vv < 0) ? 0 : vv;
    if constexpr (kMap) {
      if constexpr (FAST) {
        const long o = (long)v * g.coord.pitch + c.xs;  // scalar
        const float* rx = g.coord.mx + o;
        const float* ry = g.coord.my + o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          sx[k] = rx[(unsigned)lane + 64u * k];
          sy[k] = ry[(unsigned)lane + 64u * k];
        }
      } else {
        const float* rx = g.coord.mx + (long)v * g.coord.pitch;
        const float* ry = g.coord.my + (long)v * g.coord.pitch;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned u = (unsigned)(c.uq[k] < 0 ? 0 : c.uq[k]);
          sx[k] = rx[u];
          sy[k] = ry[u];
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if constexpr (FAST) g.coord.get(c.xs + lane + 64 * k, v, sx[k], sy[k]);
        else g.coord.get(c.uq[k] < 0 ? 0 : c.uq[k], v, sx[k], sy[k]);
      }
    }
  }

  // ---- producer: one row of coordinates -> footprint records + row span ------------------
  template <bool FAST>
  static __device__ __forceinline__ void produce_row(const Src& g, const SrcView& s, const Cols& c,
                                                     int vv, bool skip, const C (&sx)[4],
                                                     const C (&sy)[4], float* rec, int* meta) {
    const unsigned lane = threadIdx.x & 63u;
    if (skip) {
      if (lane == 0) meta[0] = kRowSkip;
      return;
    }
    if (!FAST && vv < 0) {  // constant filter border: the whole row is the border value
      if (lane == 0) meta[0] = kRowConst;
      return;
    }
    int ix0[4], iy0[4];
    float tx[4], ty[4];
    bool in[4];
    int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
    bool slow = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool ok = sx[k] > (C)-kCoordLimit && sx[k] < (C)kCoordLimit &&
                      sy[k] > (C)-kCoordLimit && sy[k] < (C)kCoordLimit;
      axis_frac<kLinear, float, C, -1>(s, ok ? sx[k] : (C)0, ix0[k], tx[k]);
      axis_frac<kLinear, float, C, -1>(s, ok ? sy[k] : (C)0, iy0[k], ty[k]);
      const bool inside = ok && ix0[k] >= 0 && iy0[k] >= 0 && ix0[k] + 2 <= s.w && iy0[k] + 2 <= s.h;
      // columns of a constant filter border are replaced by the consumer: not sampled at all
      const bool live = FAST ? true : c.uq[k] >= 0;
      in[k] = inside && live;
      slow = slow || (!inside && live);
      if (in[k]) {
        xmn = ix0[k] < xmn ? ix0[k] : xmn;
        xmx = ix0[k] > xmx ? ix0[k] : xmx;
        ymn = iy0[k] < ymn ? iy0[k] : ymn;
        ymx = iy0[k] > ymx ? iy0[k] : ymx;
      }
    }
    xmn = wave_minmax<false>(xmn);
    xmx = wave_minmax<true>(xmx);
    ymn = wave_minmax<false>(ymn);
    ymx = wave_minmax<true>(ymx);
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
      meta[0] = (ring ? kRowRing : 0) | (any ? kRowAny : 0) | (any_slow ? kRowSlow : 0);
      meta[1] = xmn; meta[2] = xmx; meta[3] = ymn; meta[4] = ymx;
    }
  }

  // ---- K x K step on one staged row (same chain as wave_run_strip) -------------------------
  static __device__ __forceinline__ void filter_step(v2f (&acc)[K][2], const float* xp,
                                                     unsigned lane, unsigned lane4_opaque,
                                                     const Weights<float, K * K>& wts,
                                                     kernarg_f32 wk) {
    const float* wp = xp + kRowPad - G::H + 4u * lane;
    const float* wq = xp + kRowPad - G::H + lane4_opaque;
    v2f pair[K + 2];
#pragma unroll
    for (int m = 0; m < K + 2; m++)
      pair[m] = (m & 1) ? v2f{wq[m], wq[m + 1]} : v2f{wp[m], wp[m + 1]};

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
                       : "+s"(cc[i][0]), "+s"(cc[i][1]), "+s"(cc[i][2]), "+v"(acc[i + 1][0]),
                         "+v"(acc[i + 1][1]));
        if constexpr (i > 0) {
          asm volatile("s_load_dwordx4 %0, %5, %6\n\ts_load_dwordx4 %1, %5, %7\n\t"
                       "s_load_dwordx4 %2, %5, %8"
                       : "=&s"(cc[i - 1][0]), "=&s"(cc[i - 1][1]), "=&s"(cc[i - 1][2]),
                         "+v"(acc[i - 1][0]), "+v"(acc[i - 1][1])
                       : "s"(wk), "n"((i - 1) * 48), "n"((i - 1) * 48 + 16),
                         "n"((i - 1) * 48 + 32));
        }
      }
#pragma unroll
      for (int j = 0; j < K; j++) {
        float w;
        if constexpr (STREAM) w = cc[i][j >> 2][j & 3];
        else w = wts.w[i * K + j];
        const v2f w2 = v2f{w, w};
#pragma unroll
        for (int h = 0; h < 2; h++) {
          if constexpr (i == 0) {
            acc[0][h] = j == 0 ? w2 * pair[2 * h]
                               : __builtin_elementwise_fma(w2, pair[j + 2 * h], acc[0][h]);
          } else {
            acc[i][h] = __builtin_elementwise_fma(w2, pair[j + 2 * h],
                                                  j == 0 ? acc[i - 1][h] : acc[i][h]);
          }
        }
      }
      if constexpr (STREAM) __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- one strip of one group ---------------------------------------------------------------
  struct RowMeta {
    int flags, xmin, xmax, ymin, ymax;
  };

  // Pipeline of a strip (chunk = kGW rows, ci = chunk being consumed):
  //   a. produce the records of chunk ci + 1 (this wave: one row; coordinates were loaded
  //      during the previous iteration)                                      | barrier X
  //   b. read the spans of chunk ci + 1, plan its ring traffic, ISSUE those source-row loads
  //      and the coordinate loads of chunk ci + 2 - nothing waits for them here
  //   c. consume chunk ci: rows software-pipelined by one (taps of row d + 1 are issued
  //      before row d is blended and filtered); needs no vector-memory result of b.
  //   d. the prefetched rows / spans become the current ones                 | barrier Y
  // The loop starts at ci = -1 (nothing to consume) so that the rows a chunk consumes always
  // come out of step d: no vector-memory wait lands inside step c.
  template <bool FAST>
  static __device__ __forceinline__ void run_strip(const WaveParams& p, const Src& g,
                                                   const SrcView& s,
                                                   const Weights<float, K * K>& wts, kernarg_f32 wk,
                                                   const Cols& c, int y0, int nrows, bool writer,
                                                   bool active, float* dst, Shared& sh,
                                                   unsigned wave) {
    const int T = nrows + K - 1;  // input rows of the strip
    const int nchunks = (T + kGW - 1) / kGW;
    const unsigned lane = threadIdx.x & 63u;
    unsigned lane4_opaque = 4u * lane;
    asm volatile("" : "+v"(lane4_opaque));
    float* xp = sh.xrow + kLead + wave * kRowStride;
    float* ringw = sh.ring[wave];

    auto row_of = [&](int t) -> int {
      if constexpr (FAST) return y0 - G::H + t;
      else return resolve_idx(y0 - G::H + t, p.dh, p.cby);
    };
    // the row this wave turns into records in chunk ci (clamped: rows past the strip are skipped)
    auto prod_row = [&](int ci) -> int {
      int t = kGW * ci + (int)wave;
      return row_of(t < T ? t : T - 1);
    };

    C cx[4], cy[4];          // coordinates of the row this wave produces next
    RingState rs{0, 0, 0};   // what the ring holds now
    RingState ps{0, 0, 0};   // the same state one chunk ahead (planner)
    PendRow<ST> pend[kPend];
    int pbase = 0, pn = 0, pxlo = 0;
    RowMeta mcur[kGW];
#pragma unroll
    for (int d = 0; d < kGW; d++) mcur[d] = RowMeta{kRowSkip, 0, 0, 0, 0};

    auto produce = [&](int ci) {
      const int t = kGW * ci + (int)wave;
      const int vv = prod_row(ci);
      if constexpr (!kMap) coords_of_row<FAST>(g, c, vv, cx, cy);
      produce_row<FAST>(g, s, c, vv, t >= T, cx, cy, sh.rec[ci & 1][wave], sh.meta[ci & 1][wave]);
    };
    // spans of chunk ci (its records are visible) -> scalars; plan the ring traffic and issue
    // the row loads
    auto plan = [&](int ci, PendRow<ST> (&pr)[kPend], RowMeta (&mm)[kGW], int& nbase, int& nn,
                    int& nxlo) {
      nn = 0; nbase = 0; nxlo = 0;
      bool stop = false;
#pragma unroll
      for (int d = 0; d < kGW; d++) {
        const int* m = sh.meta[ci & 1][d];
        mm[d].flags = __builtin_amdgcn_readfirstlane(m[0]);
        mm[d].xmin = __builtin_amdgcn_readfirstlane(m[1]);
        mm[d].xmax = __builtin_amdgcn_readfirstlane(m[2]);
        mm[d].ymin = __builtin_amdgcn_readfirstlane(m[3]);
        mm[d].ymax = __builtin_amdgcn_readfirstlane(m[4]);
      }
#pragma unroll
      for (int d = 0; d < kGW; d++) {
        if ((mm[d].flags & (kRowRing | kRowAny | kRowSkip)) == (kRowRing | kRowAny)) {
          bool restarted;
          const int first = ring_advance(ps, mm[d].xmin, mm[d].xmax, mm[d].ymin, mm[d].ymax,
                                         restarted);
          if (restarted && nn > 0) stop = true;
          const int add = mm[d].ymax + 2 - first;
          if (add > 0 && !stop) {
            if (nn == 0) { nbase = first; nxlo = ps.xlo; }
            if (nbase + nn == first && nxlo == ps.xlo) {
              const int take = add < kPend - nn ? add : kPend - nn;
              nn += take;
              if (take < add) stop = true;
            } else {
              stop = true;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kPend; j++)
        if (j < nn) pr[j].load(s.rsrc, __mul24(nbase + j, s.pitch) + nxlo, lane);
    };

    if constexpr (kMap) coords_of_row<FAST>(g, c, prod_row(0), cx, cy);

    v2f acc[K][2];
#pragma unroll 1
    for (int ci = -1; ci < nchunks; ci++) {
      // a. records of the next chunk (other buffer)
      if (ci + 1 < nchunks) produce(ci + 1);
      lds_barrier();  // X: records of chunk ci + 1 visible

      // b. spans + ring rows of the next chunk, coordinates of the one after it: in flight
      //    while this chunk is consumed
      PendRow<ST> pnext[kPend];
      RowMeta mnext[kGW];
#pragma unroll
      for (int d = 0; d < kGW; d++) mnext[d] = RowMeta{kRowSkip, 0, 0, 0, 0};
      int nbase = 0, nn = 0, nxlo = 0;
      if (active && ci + 1 < nchunks) plan(ci + 1, pnext, mnext, nbase, nn, nxlo);
      if constexpr (kMap)
        if (ci + 2 < nchunks) coords_of_row<FAST>(g, c, prod_row(ci + 2), cx, cy);

      // c. consume chunk ci
      if (active && ci >= 0) {
        const int cb = ci & 1;
        float4 rs4[kGW], rx4[kGW], ry4[kGW];  // records of the rows in flight
        float v[2][4][2][2];                  // taps of two rows (even / odd d)

        auto load_rec = [&](auto Dd) {
          constexpr int d = decltype(Dd)::value;
          if (mcur[d].flags & (kRowSkip | kRowConst)) return;
          const float* r = sh.rec[cb][d];
          rs4[d] = *reinterpret_cast<const float4*>(r + 4u * lane);
          rx4[d] = *reinterpret_cast<const float4*>(r + 256 + 4u * lane);
          ry4[d] = *reinterpret_cast<const float4*>(r + 512 + 4u * lane);
        };
        // stage A of a row: make its source rows resident, issue its taps
        auto stage_a = [&](auto Dd) {
          constexpr int d = decltype(Dd)::value;
          const RowMeta m = mcur[d];
          if (m.flags & (kRowSkip | kRowConst)) return;
          const int sl[4] = {__float_as_int(rs4[d].x), __float_as_int(rs4[d].y),
                             __float_as_int(rs4[d].z), __float_as_int(rs4[d].w)};
          if (m.flags & kRowRing) {
            if (m.flags & kRowAny) {
              bool restarted;
              int y = ring_advance(rs, m.xmin, m.xmax, m.ymin, m.ymax, restarted);
              const int need_hi = m.ymax + 2;
              // rows prefetched for this chunk, in order
#pragma unroll
              for (int j = 0; j < kPend; j++) {
                if (j < pn && pxlo == rs.xlo && pbase + j == y && y < need_hi) {
                  pend[j].write(ringw + (y & (kRR - 1)) * kRW, lane);
                  if ((y & (kRR - 1)) == 0) pend[j].write(ringw + kRR * kRW, lane);
                  y++;
                }
              }
              // whatever the prefetch did not cover (restarts, tall spans)
#pragma unroll 1
              for (; y < need_hi; y++) {
                PendRow<ST> q;
                q.load(s.rsrc, __mul24(y, s.pitch) + rs.xlo, lane);
                q.write(ringw + (y & (kRR - 1)) * kRW, lane);
                if ((y & (kRR - 1)) == 0) q.write(ringw + kRR * kRW, lane);
              }
            }
            // taps: 4 reads at immediate offsets from one address; reads beyond the
            // allocation return 0, so unused (-1) slots need no guard
            const char* rb = reinterpret_cast<const char*>(ringw) - 4 * rs.xlo;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const float* tp = reinterpret_cast<const float*>(rb + sl[k]);
              v[d & 1][k][0][0] = tp[0];
              v[d & 1][k][0][1] = tp[1];
              v[d & 1][k][1][0] = tp[kRW];
              v[d & 1][k][1][1] = tp[kRW + 1];
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int e = sl[k] < 0 ? 0 : sl[k];
              TapLoad<ST, float>::template row<2>(s, e, v[d & 1][k][0]);
              TapLoad<ST, float>::template row<2>(s, e + s.pitch, v[d & 1][k][1]);
            }
          }
        };
        // stage B of a row: blend, K x K step, store
        auto stage_b = [&](auto Dd) {
          constexpr int d = decltype(Dd)::value;
          const RowMeta m = mcur[d];
          if (m.flags & kRowSkip) return;
          const int t = kGW * ci + d;
          float cur[4];
          if (m.flags & kRowConst) {
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = g.ccval;
          } else {
            const float tx[4] = {rx4[d].x, rx4[d].y, rx4[d].z, rx4[d].w};
            const float ty[4] = {ry4[d].x, ry4[d].y, ry4[d].z, ry4[d].w};
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
            if (m.flags & kRowSlow) {
              // footprints touching the source border (rare): tap by tap
              const int sl[4] = {__float_as_int(rs4[d].x), __float_as_int(rs4[d].y),
                                 __float_as_int(rs4[d].z), __float_as_int(rs4[d].w)};
              const int vv = row_of(t);
              C sx[4], sy[4];
              coords_of_row<FAST>(g, c, vv, sx, sy);
#pragma unroll
              for (int k = 0; k < 4; k++)
                if (sl[k] < 0 && (FAST || c.uq[k] >= 0))
                  cur[k] = sample<ST, kLinear, C>(s, sx[k], sy[k], g.cval);
            }
            if constexpr (!FAST) {
#pragma unroll
              for (int k = 0; k < 4; k++) cur[k] = c.uq[k] < 0 ? g.ccval : cur[k];
            }
          }
          float* row = xp + kRowPad;
#pragma unroll
          for (int k = 0; k < 4; k++) row[64u * k + lane] = cur[k];
          __builtin_amdgcn_wave_barrier();

          filter_step(acc, xp, lane, lane4_opaque, wts, wk);

          const int o = t - (K - 1);
          if (o >= 0 && o < nrows && writer) {
            const float4 q = float4{acc[K - 1][0].x, acc[K - 1][0].y, acc[K - 1][1].x, acc[K - 1][1].y};
            if constexpr (FAST) {
              float* rows_ = dst + ((long)(y0 + o) * p.dpitch + c.xs);  // scalar base
              __builtin_nontemporal_store(q.x, rows_ + 4u * lane);
              __builtin_nontemporal_store(q.y, rows_ + 4u * lane + 1);
              __builtin_nontemporal_store(q.z, rows_ + 4u * lane + 2);
              __builtin_nontemporal_store(q.w, rows_ + 4u * lane + 3);
            } else {
              float* orow = dst + (long)(y0 + o) * p.dpitch + c.xo;
              const int n = p.dw - c.xo < 4 ? p.dw - c.xo : 4;
              if (p.vec_out && n == 4) {
                *reinterpret_cast<float4*>(orow) = q;
              } else {
                const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                  if (k < n) orow[k] = e[k];
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
      // d. the prefetched rows / spans become the current ones
#pragma unroll
      for (int j = 0; j < kPend; j++) pend[j] = pnext[j];
      pbase = nbase; pn = nn; pxlo = nxlo;
#pragma unroll
      for (int d = 0; d < kGW; d++) mcur[d] = mnext[d];
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

    const int xs = sxi * G::OW - 4 * G::HL;
    Cols c;
    c.xs = xs;
    c.xo = xs + lane * 4;
    const int y0 = syi * p.strip_h;
    const int nrows = p.dh - y0 < p.strip_h ? p.dh - y0 : p.strip_h;
    const bool writer = lane >= G::HL && lane < 64 - G::HL && c.xo < p.dw;
    float* dst = reinterpret_cast<float*>(p.dst) + (long)(active ? frame : 0u) * p.dst_frame_elems;

    const bool fast = p.vec_out && xs >= 0 && xs + 256 <= p.dw &&
                      y0 - G::H >= 0 && y0 - G::H + nrows + K - 1 <= p.dh;
    if (fast) {
#pragma unroll
      for (int k = 0; k < 4; k++) c.uu[k] = c.xo + k;
      run_strip<true>(p, g, s, wts, wk, c, y0, nrows, writer, active, dst, sh, wave);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        c.uu[k] = resolve_idx(c.xo + k, p.dw, p.cbx);
        c.uq[k] = resolve_idx(xs + lane + 64 * k, p.dw, p.cbx);
      }
      run_strip<false>(p, g, s, wts, wk, c, y0, nrows, writer, active, dst, sh, wave);
    }
  }
};

}  // namespace ipa
