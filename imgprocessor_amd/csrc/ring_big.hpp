// ring_big.hpp - bilinear / bicubic remap -> dense K x K filter (K = 7, 9, 11) of float32
// batches in ONE kernel with the taps read from LDS: BASELINE configuration C5 (bicubic
// perspective warp -> 11x11) and its relatives, which ran as two launches through a workspace
// image because the sampling gathers plus 9 / 11 running rows did not fit one wave's registers.
//
// One wave per (128-px strip, frame); a step is 2 rows of samples (4 per lane):
//   clean strips (ring_plan_kernel: interior of the output, every footprint inside the source
//   and inside the ring's reach)   the step's source rows arrive by coalesced row loads issued
//   one step ahead and go into a wave-private LDS ring; a sample's taps are ds_read2_b32 of
//   {row r, row r + 1} pairs, summed by v_pk_fma_f32 (ring_remap.hpp) - no gather instruction,
//   the vector-memory path only sees row loads, coordinate loads and the output stores;
//   the other strips (output rim, footprints on the source border, strong rotation)
//   sample() per pixel at the border-resolved position, as the per-frame kernels do.
// Either way the two sample rows go to a wave-private staging row and the K x K running sums
// advance by two rows: kernel row i streams through 12 SGPRs while row i + 1 is accumulated
// (the scheme of wave_run_strip, wave_stencil.hpp), 2 pixels per lane.
//
// Arithmetic, rounding and summation order are those of sample() and wave_run_strip(): the
// result has the bits of remap kernel -> filter kernel through a float32 workspace image.
//
// Reference call chain: camera/PerspectiveCorrection.py:401-405 (cv2.warpPerspective) /
// camera/LensDistortion.py:323-326 (cv2.remap) followed by filters/maskedConvolve.py:24-43.
#pragma once
#include "ring_remap.hpp"

namespace ipa {

template <typename Coord> struct RingBigSrc {
  Coord coord;                                   // as given (rim strips evaluate it)
  typename ring_kernel_coord<Coord>::type kc;    // what the clean strips read
  const char* src;
  long src_frame_bytes;
  unsigned src_bytes;
  int sh, sw, spitch;
  int border, q5;
  float cubic_a;
  float cval;            // remap border value
  float ccval;           // filter border value
  int n_frames;
};

template <int INTERP, typename Coord, int K> struct RingBigArgs {
  WaveParams p;
  RingGeom gm;
  RingBigSrc<Coord> g;
  RingPlan plan;
  alignas(16) float wrows[K][12];  // kernel row i, taps 0..K-1, zero padded
};

template <int INTERP, typename Coord, int K> struct RingBigKernel {
  using KCoord = typename ring_kernel_coord<Coord>::type;
  using C = typename Coord::coord_t;
  using G = group_geom<K>;
  using Args = RingBigArgs<INTERP, Coord, K>;
  static constexpr int NT = ntaps<INTERP>::value;
  static constexpr int RR = ring_rows<INTERP>::value;
  static constexpr int kSlots = RR + NT - 1;
  static constexpr int kWaves = 4;
  static constexpr int kLead = G::H > kRowPad ? 4 : 0;

  struct Shared {
    float ring[kWaves][kSlots * kRW];
    float xrow[kLead + kWaves * 2 * kXRow + kLead];
  };

  // the K x K step on one staged row, 2 px per lane; returns with acc[K - 1] = a finished row
  static __device__ __forceinline__ void filter_row(v2f (&acc)[K], const float* xr, unsigned lane,
                                                    unsigned lane2_opaque, kernarg_f32 wk) {
    // pair[m] = (px 2L-H+m, px 2L-H+m+1); even and odd m through two offsets the compiler
    // cannot relate, so that every pair is its own aligned register pair (wave_run_strip)
    const float* wp = xr + kRowPad - G::H + 2u * lane;
    const float* wq = xr + kRowPad - G::H + lane2_opaque;
    v2f pair[K];
#pragma unroll
    for (int m = 0; m < K; m++) pair[m] = (m & 1) ? v2f{wq[m], wq[m + 1]} : v2f{wp[m], wp[m + 1]};
    v4f cc[K][3];
    asm volatile("s_load_dwordx4 %0, %3, %4\n\ts_load_dwordx4 %1, %3, %5\n\t"
                 "s_load_dwordx4 %2, %3, %6"
                 : "=&s"(cc[K - 1][0]), "=&s"(cc[K - 1][1]), "=&s"(cc[K - 1][2])
                 : "s"(wk), "n"((K - 1) * 48), "n"((K - 1) * 48 + 16), "n"((K - 1) * 48 + 32));
    static_for<0, K>([&](auto Ii) {
      constexpr int i = K - 1 - decltype(Ii)::value;
      if constexpr (i == K - 1)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cc[i][0]), "+s"(cc[i][1]), "+s"(cc[i][2]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+s"(cc[i][0]), "+s"(cc[i][1]), "+s"(cc[i][2]), "+v"(acc[i + 1]));
      if constexpr (i > 0) {
        // in front of row i's fmas (they read acc[i - 1]): a row of fmas covers the latency
        asm volatile("s_load_dwordx4 %0, %4, %5\n\ts_load_dwordx4 %1, %4, %6\n\t"
                     "s_load_dwordx4 %2, %4, %7"
                     : "=&s"(cc[i - 1][0]), "=&s"(cc[i - 1][1]), "=&s"(cc[i - 1][2]),
                       "+v"(acc[i - 1])
                     : "s"(wk), "n"((i - 1) * 48), "n"((i - 1) * 48 + 16), "n"((i - 1) * 48 + 32));
      }
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
      // keep the next kernel row's scalar loads below this row's fmas
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // Two staged rows at once (rows t and t + 1 of the strip): running sum i after both rows is
  //   A_i'' = A_(i-2) + sum_j w[i-1][j] x_t[j] + sum_j w[i][j] x_(t+1)[j]
  // evaluated as ONE chain in exactly that order - the order two calls of filter_row produce -
  // so a kernel row is fetched once per step instead of once per staged row and every wait for
  // it is followed by up to 2 K fmas.  Chain K is the row finished by row t alone (out0:
  // A_(K-2) + w[K-1] x_t); acc[K - 1] afterwards is the row finished by row t + 1.
  static __device__ __forceinline__ void filter_rows2(v2f (&acc)[K], v2f& out0, const float* xr0,
                                                      const float* xr1, unsigned lane,
                                                      unsigned lane2_opaque, kernarg_f32 wk) {
    v2f x0[K], x1[K];
    {
      const float* wp = xr0 + kRowPad - G::H + 2u * lane;
      const float* wq = xr0 + kRowPad - G::H + lane2_opaque;
#pragma unroll
      for (int m = 0; m < K; m++) x0[m] = (m & 1) ? v2f{wq[m], wq[m + 1]} : v2f{wp[m], wp[m + 1]};
      const float* up = xr1 + kRowPad - G::H + 2u * lane;
      const float* uq = xr1 + kRowPad - G::H + lane2_opaque;
#pragma unroll
      for (int m = 0; m < K; m++) x1[m] = (m & 1) ? v2f{uq[m], uq[m + 1]} : v2f{up[m], up[m + 1]};
    }
    v4f cc[K][3];
    asm volatile("s_load_dwordx4 %0, %3, %4\n\ts_load_dwordx4 %1, %3, %5\n\t"
                 "s_load_dwordx4 %2, %3, %6"
                 : "=&s"(cc[K - 1][0]), "=&s"(cc[K - 1][1]), "=&s"(cc[K - 1][2])
                 : "s"(wk), "n"((K - 1) * 48), "n"((K - 1) * 48 + 16), "n"((K - 1) * 48 + 32));
    v2f prev = v2f{0.f, 0.f};  // result of the previous chain: pins the order of the statements
    static_for<0, K + 1>([&](auto Ii) {
      constexpr int i = K - decltype(Ii)::value;  // K, K-1, ..., 0
      constexpr int rn = i >= 1 ? i - 1 : 0;       // the row that arrived last
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+s"(cc[rn][0]), "+s"(cc[rn][1]), "+s"(cc[rn][2]), "+v"(prev));
      if constexpr (i >= 2) {
        // row i - 2 for the next chain, in front of this chain's fmas: they cover its latency
        asm volatile("s_load_dwordx4 %0, %4, %5\n\ts_load_dwordx4 %1, %4, %6\n\t"
                     "s_load_dwordx4 %2, %4, %7"
                     : "=&s"(cc[i - 2][0]), "=&s"(cc[i - 2][1]), "=&s"(cc[i - 2][2]), "+v"(prev)
                     : "s"(wk), "n"((i - 2) * 48), "n"((i - 2) * 48 + 16), "n"((i - 2) * 48 + 32));
      }
      v2f r = v2f{0.f, 0.f};
      if constexpr (i >= 2) r = acc[i - 2];
      if constexpr (i >= 1) {
#pragma unroll
        for (int j = 0; j < K; j++) {
          const float w = cc[i - 1][j >> 2][j & 3];
          const v2f w2 = v2f{w, w};
          if (i < 2 && j == 0) r = w2 * x0[0];
          else r = __builtin_elementwise_fma(w2, x0[j], r);
        }
      }
      if constexpr (i <= K - 1) {
#pragma unroll
        for (int j = 0; j < K; j++) {
          const float w = cc[i][j >> 2][j & 3];
          const v2f w2 = v2f{w, w};
          if (i < 1 && j == 0) r = w2 * x1[0];
          else r = __builtin_elementwise_fma(w2, x1[j], r);
        }
      }
      if constexpr (i == K) out0 = r;
      else acc[i] = r;
      prev = r;
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  static __device__ __forceinline__ void body(const Args& a, kernarg_f32 wk) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    const WaveParams& p = a.p;
    const RingGeom& gm = a.gm;
    const RingBigSrc<Coord>& g = a.g;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned lane2_opaque = 2u * lane;
    asm volatile("" : "+v"(lane2_opaque));
    // the frames of one strip share a workgroup: they read the same coordinates at the same time
    const unsigned groups = ((unsigned)g.n_frames + kWaves - 1) / kWaves;
    const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned grp = b % groups;
    const unsigned sid = b / groups;
    const unsigned frame = grp * kWaves + wave;
    if (frame >= (unsigned)g.n_frames) return;
    const int syi = (int)(sid / (unsigned)gm.strips_x), sxi = (int)sid - syi * gm.strips_x;
    const int4 info = a.plan.info[sid];
    const bool clean = __builtin_amdgcn_readfirstlane(info.x) != 0;

    const int xs = sxi * G::OW - 2 * G::HL;
    const int y0 = syi * gm.strip_h;
    const int nrows = gm.dh - y0 < gm.strip_h ? gm.dh - y0 : gm.strip_h;
    const int T = nrows + K - 1;
    const int nsteps = (T + 1) / 2;
    // output pixels of the lane: columns xs + 2 lane, + 1
    const int xo = xs + 2 * (int)lane;
    const bool wlane = lane >= (unsigned)G::HL && lane < 64u - G::HL;
    const bool w0 = wlane && xo < gm.dw, w1 = wlane && xo + 1 < gm.dw;
    float* dst = reinterpret_cast<float*>(p.dst) + (long)frame * p.dst_frame_elems +
                 ((long)y0 * p.dpitch + xs);
    SrcView s;
    s.rsrc = make_rsrc(g.src + (long)frame * g.src_frame_bytes, g.src_bytes);
    s.w = g.sw; s.h = g.sh; s.pitch = g.spitch;
    s.border = g.border; s.q5 = g.q5; s.cubic_a = g.cubic_a; s.lanczos = nullptr;
    s.pair_split = 1;
    float* xp = sh.xrow + kLead + wave * 2 * kXRow;
    float* ringw = sh.ring[wave];
    v2f acc[K];
#pragma unroll
    for (int i = 0; i < K; i++) acc[i] = v2f{0.f, 0.f};

    // the two staged rows of step st -> running sums -> output rows
    auto filter_step = [&](int st) {
      const int t = 2 * st;
      float* row = dst + (long)(t - (K - 1)) * p.dpitch;  // output row of staged row t
      if (t + 1 < T) {
        v2f out0;
        filter_rows2(acc, out0, xp, xp + kXRow, lane, lane2_opaque, wk);
        if (t - (K - 1) >= 0) {
          if (w0) __builtin_nontemporal_store(out0.x, row + 2u * lane);
          if (w1) __builtin_nontemporal_store(out0.y, row + 2u * lane + 1);
        }
        if (t + 1 - (K - 1) >= 0) {
          row += p.dpitch;
          if (w0) __builtin_nontemporal_store(acc[K - 1].x, row + 2u * lane);
          if (w1) __builtin_nontemporal_store(acc[K - 1].y, row + 2u * lane + 1);
        }
      } else {  // odd number of staged rows: the last one alone
        filter_row(acc, xp, lane, lane2_opaque, wk);
        if (t - (K - 1) >= 0) {
          if (w0) __builtin_nontemporal_store(acc[K - 1].x, row + 2u * lane);
          if (w1) __builtin_nontemporal_store(acc[K - 1].y, row + 2u * lane + 1);
        }
      }
    };

    if (!clean) {
      // ---- rim / border strips: one gathered sample per pixel at the resolved position ----
      int uq[2];
#pragma unroll
      for (int q = 0; q < 2; q++) uq[q] = resolve_idx(xs + (int)lane + 64 * q, gm.dw, p.cbx);
#pragma unroll 1
      for (int st = 0; st < nsteps; st++) {
        float cur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int t = 2 * st + (k >> 1);
          const int v = resolve_idx(y0 - G::H + (t < T ? t : T - 1), gm.dh, p.cby);
          const int u = uq[k & 1];
          float val = g.ccval;
          if (u >= 0 && v >= 0) {
            C sx, sy;
            g.coord.get(u, v, sx, sy);
            val = sample<float, INTERP, C>(s, sx, sy, g.cval);
          }
          cur[k] = val;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) xp[(k >> 1) * kXRow + kRowPad + 64u * (k & 1) + lane] = cur[k];
        __builtin_amdgcn_wave_barrier();
        filter_step(st);
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }

    // ---- clean strips: taps from the ring ---------------------------------------------------
    const int xlo = __builtin_amdgcn_readfirstlane(info.y);
    const unsigned* cwp = a.plan.cnts + (size_t)sid * kPlanWords;
    auto put_row = [&](const PendRow<float>& r, int y) {
      const int slot = y & (RR - 1);
      r.write(ringw + slot * kRW, lane);
      if (slot < NT - 1) r.write(ringw + (slot + RR) * kRW, lane);  // mirror: no footprint wraps
    };
    int hres = __builtin_amdgcn_readfirstlane(info.z);  // rows [.., hres) are in the ring
    {
      const int n0 = __builtin_amdgcn_readfirstlane(info.w);
#pragma unroll 1
      for (int j = 0; j < n0; j += 2) {
        PendRow<float> p0, p1;
        p0.load(s.rsrc, __mul24(hres + j, g.spitch) + xlo, lane);
        if (j + 1 < n0) p1.load(s.rsrc, __mul24(hres + j + 1, g.spitch) + xlo, lane);
        put_row(p0, hres + j);
        if (j + 1 < n0) put_row(p1, hres + j + 1);
      }
      hres += n0;
    }
    C cx[4], cy[4];
    auto step_coords = [&](int st) {
      const int sc = st < nsteps ? st : nsteps - 1;
      const int v0 = y0 - G::H + 2 * sc;
      const int v1 = 2 * sc + 1 < T ? v0 + 1 : v0;
      ring_coords<KCoord>(g.kc, xs, v0, v1, cx, cy);
    };
    step_coords(0);
    unsigned words[kPlanWords];
#pragma unroll
    for (int i = 0; i < kPlanWords; i++) words[i] = __builtin_amdgcn_readfirstlane(cwp[i]);
    PendRow<float> pend[kRingMaxNew];
    int cnt = 0;
    const int rbase = lds_address(ringw);

#pragma unroll 1
    for (int st = 0; st < nsteps; st++) {
      // 1. the rows requested during the previous step go into the ring
      cnt = __builtin_amdgcn_readfirstlane(cnt);
      hres = __builtin_amdgcn_readfirstlane(hres);
#pragma unroll
      for (int j = 0; j < kRingMaxNew; j++)
        if (j < cnt) put_row(pend[j], hres + j);
      hres += cnt;
      // 2. footprints and weights of this step (the arithmetic of sample())
      int ad[4];
      float wx[4][NT], wy[4][NT];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int ix0, iy0;
        axis_split<INTERP, float, C>(s, cx[k], ix0, wx[k]);
        axis_split<INTERP, float, C>(s, cy[k], iy0, wy[k]);
        ad[k] = (__mul24(iy0 & (RR - 1), kRW) + (ix0 - xlo)) << 2;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) asm volatile("" : "+v"(ad[k]) : : "memory");
      // 3. requests for the next step: its source rows and its coordinates
      const int sn = st + 1;
      {
        unsigned wv = words[0];
#pragma unroll
        for (int q = 1; q < kPlanWords; q++) wv = (sn >> 3) == q ? words[q] : wv;
        cnt = sn < nsteps ? (int)((wv >> (4 * (sn & 7))) & 15u) : 0;
      }
      cnt = __builtin_amdgcn_readfirstlane(cnt);
#pragma unroll
      for (int j = 0; j < kRingMaxNew; j++)
        if (j < cnt) pend[j].load(s.rsrc, __mul24(hres + j, g.spitch) + xlo, lane);
      step_coords(sn);
      __builtin_amdgcn_sched_barrier(0);

      // 4. taps from the ring: row pairs by ds_read2_b32, column weights by v_pk_fma_f32
      __builtin_amdgcn_wave_barrier();
      v2f t[4][NT / 2][NT];
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int rp = 0; rp < NT / 2; rp++) {
          const int ra = rbase + ad[k] + rp * (2 * kRW * 4);
          static_for<0, NT>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            t[k][rp][c] = lds_read2<c, c + kRW>(ra);
          });
        }
      lds_wait_all();
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int rp = 0; rp < NT / 2; rp++)
#pragma unroll
          for (int c = 0; c < NT; c++) asm volatile("" : "+v"(t[k][rp][c]));
      float cur[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        v2f wp[NT / 2];
#pragma unroll
        for (int j = 0; j < NT / 2; j++) wp[j] = v2f{wx[k][2 * j], wx[k][2 * j + 1]};
        float o = 0.f;
#pragma unroll
        for (int rp = 0; rp < NT / 2; rp++) {
          v2f rs = pk_mul_half<0>(wp[0], t[k][rp][0]);
#pragma unroll
          for (int c = 1; c < NT; c++) {
            if (c & 1) rs = pk_fma_half<1>(wp[c >> 1], t[k][rp][c], rs);
            else rs = pk_fma_half<0>(wp[c >> 1], t[k][rp][c], rs);
          }
          o = rp == 0 ? wy[k][0] * rs.x : ipa_fma(wy[k][2 * rp], rs.x, o);
          o = ipa_fma(wy[k][2 * rp + 1], rs.y, o);
        }
        cur[k] = o;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) xp[(k >> 1) * kXRow + kRowPad + 64u * (k & 1) + lane] = cur[k];
      __builtin_amdgcn_wave_barrier();

      // 5. K x K steps and output rows
      filter_step(st);
      __builtin_amdgcn_wave_barrier();
    }
  }
};

template <int INTERP, typename Coord, int K>
__global__ void __launch_bounds__(256)
ring_big_kernel(RingBigArgs<INTERP, Coord, K> a) {
  static_assert(K <= 12, "padded coefficient rows of 12");
  typedef const char __attribute__((address_space(4)))* kernarg_bytes;
  kernarg_bytes base = (kernarg_bytes)__builtin_amdgcn_kernarg_segment_ptr();
  using Args = RingBigArgs<INTERP, Coord, K>;
  kernarg_f32 wk = (kernarg_f32)(base + offsetof(Args, wrows));
  RingBigKernel<INTERP, Coord, K>::body(a, wk);
}

}  // namespace ipa
