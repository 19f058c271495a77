// wave_lring.hpp — the fused bilinear remap -> filter loop with the SOURCE ROWS of a strip staged
// through a wave-private LDS ring by LDS-DMA (round 4).  Included at the end of wave_stencil.hpp.
//
// wave_run_strip_shared (wave_pipe.hpp) takes the four taps of a bilinear footprint with per-lane
// gathers: per 256-px row a wave issues ~13 dword gathers, and a vector-memory instruction costs
// the CU's vector-memory path ~4.6 clocks whatever it fetches.  A lens undistortion is a smooth
// map: the footprints of one output row of a 256-px strip lie in a band of a few source rows, and
// the band moves down by at most one row per output row.  Here
//
//   * a planning launch (lring_plan_kernel, once per call / cached per coordinate source) walks
//     the footprints of every strip and decides whether it is CLEAN: every footprint inside the
//     source, all footprints of a row within kLrPitch columns of the strip's leftmost one, the
//     rows a strip row needs available in a ring of kLrR source rows that advances by AT MOST ONE
//     row per output row.  For a clean strip it records the column origin, the first source row
//     and one schedule bit per output row: "load the next source row" / "stall";
//   * the hot loop of a clean strip issues per output row exactly TWO vector-memory instructions
//     for the source - `buffer_load_dwordx4 ... lds` (1 KiB) + a 4-lane tail: the scheduled source
//     row, kLrD rows ahead of its first use, straight into the ring (no registers, no ds_write) -
//     and one store; a stall re-loads the row it loaded last (same bytes to the same place), so
//     that the hand-counted `s_waitcnt vmcnt(N)` do not depend on the schedule;
//   * the footprint RECORDS of a row (ring byte offsets of the top and the bottom tap pair, the
//     two fractions) are worked out once per workgroup = four frames of one strip, as in
//     wave_run_strip_shared, in blocks of TWO rows (waves 0, 1 produce the even blocks, waves 2, 3
//     the odd ones; one s_barrier per two rows), which keeps the record ring at 20 KB;
//   * a tap pair is one ds_read2_b32 from the recorded offset.  Same footprint arithmetic
//     (axis_frac), same words, same blend order as sample() / wave_run_strip_shared: identical
//     bits.
//
// The aligned 256-px strip geometry (all 64 lanes store whole 128-byte lines) comes for free: the
// 2 H halo pixels of a row are a fifth sample of lanes 0 .. 2H-1 that costs LDS reads, not six
// vector-memory instructions.  Rim strips (columns / rows resolved through the filter's border
// mode) take the same loop - the plan walks the resolved footprints.  Strips that are not clean
// (footprints leaving the source, a band wider than the ring: far corners, strong rotations) run
// wave_run_strip_shared in the same kernel.
//
// LDS per workgroup: 20 KB records + 4 sample rows + 4 rings of 13 x 1088 B = 79.4 KB: two
// workgroups = 8 waves per CU.  tools/ring_micro.hip measured the shape first (64 x 4K, one box,
// same buffers): plain rows 843 - 863 us, this skeleton with fake records 917 - 930 us for 2 - 4
// rows ahead, 8 - 16 waves per CU alike; the gather loop runs ~1.3 x the plain rows.
//
// Reference semantics: camera/LensDistortion.py:323-326 (cv2.remap INTER_LINEAR, BORDER_CONSTANT)
// followed by a K x K filter (filters/maskedConvolve.py:24-43 / scipy.ndimage.correlate).
#pragma once

namespace ipa {

#ifndef IPA_LRING_ROWS
#define IPA_LRING_ROWS 13    // source rows a wave's ring holds
#endif
#ifndef IPA_LRING_AHEAD
#define IPA_LRING_AHEAD 3    // iterations between a row's load and its first use
#endif
constexpr int kLrR = IPA_LRING_ROWS, kLrD = IPA_LRING_AHEAD;
constexpr int kLrPitch = 272, kLrPitchB = 4 * kLrPitch;   // floats / bytes of a ring row
constexpr int kLrTail = (kLrPitch - 256) / 4;             // lanes of a row load's second piece
constexpr int kLrRec = 1280;                              // floats of a record row
// plan of a strip: [0] clean, [1] column origin, [2] first source row, [3] rows loaded up front,
// [4..11] schedule bits: bit 4 + t = iteration t loads a NEW row (t = -kLrD .. : the kLrD loads
// issued before row 0)
constexpr int kLrPlanWords = 12;
constexpr int kLrMaxRows = 252;                           // 4 + rows of a strip <= 256 bits
static_assert(kLrD >= 1 && kLrD <= 4, "the virtual iterations take the first nibble of the schedule");
static_assert(kLrR >= kLrD + 3, "ring: a footprint's two rows + the rows in flight");

// LDS of a workgroup (bytes): records of 4 rows | sample rows of the 4 waves | the 4 rings
constexpr int kLrRecBytes = 4 * kLrRec * 4;
constexpr int kLrXpBytes = IPA_WPB * kRowStride * 4;
constexpr int kLrRingBytes = kLrR * kLrPitchB;
constexpr int kLrLdsBytes = kLrRecBytes + kLrXpBytes + IPA_WPB * kLrRingBytes;
static_assert(kLrLdsBytes <= 80 * 1024, "two workgroups per CU");

// footprints of NS samples: fractions, top-left tap (column, row), inside bits - the arithmetic
// of batch_footprint_linear
template <int NS, int QM, typename C>
__device__ __forceinline__ void lring_footprint(const SrcView& s, const C (&sx)[NS], const C (&sy)[NS],
                                                float (&tx)[NS], float (&ty)[NS], int (&ix)[NS],
                                                int (&iy)[NS], unsigned& interior) {
  interior = 0;
  const unsigned xlim = s.w - 1 > 0 ? (unsigned)(s.w - 1) : 0u;
  const unsigned ylim = s.h - 1 > 0 ? (unsigned)(s.h - 1) : 0u;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const bool ok = ipa_abs(sx[k]) < (C)kCoordLimit && ipa_abs(sy[k]) < (C)kCoordLimit;
    axis_frac<kLinear, float, C, QM>(s, ok ? sx[k] : (C)0, ix[k], tx[k]);
    axis_frac<kLinear, float, C, QM>(s, ok ? sy[k] : (C)0, iy[k], ty[k]);
    const bool in = ok && (unsigned)ix[k] < xlim && (unsigned)iy[k] < ylim;
    interior |= in ? (1u << k) : 0u;
  }
}

// ------------------------------------------------------------------------- planning --
// one workgroup of 16 waves per strip: wave w walks the strip rows t = w, w + 16, ... (the
// coordinates of the 260 samples of a row, resolved through the filter's border mode on the rim),
// the bounds of every row go to LDS; thread 0 then lays out the schedule.
template <typename Coord, int K>
__global__ void __launch_bounds__(1024)
lring_plan_kernel(WaveParams p, Coord coord, int sh, int sw, int q5, unsigned* plan, unsigned* stats) {
  using G = wave_geom<K, true>;
  using C = typename Coord::coord_t;
  constexpr int NS = 5, H = G::H, R = kLrR, D = kLrD;
  __shared__ int rymin[kLrMaxRows + 4], rneed[kLrMaxRows + 4], rxmin[kLrMaxRows + 4], rxmax[kLrMaxRows + 4];
  __shared__ int rbad[kLrMaxRows + 4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const unsigned sid = blockIdx.x;
  const int syi = (int)(sid / (unsigned)p.strips_x), sxi = (int)sid - syi * p.strips_x;
  const int xs = sxi * 256;
  const int y0 = syi * p.strip_h;
  const int nrows = p.dh - y0 < p.strip_h ? p.dh - y0 : p.strip_h;
  const int T = nrows + K - 1, yb = y0 - H;
  const bool fast = xs - H >= 0 && xs + 256 + H <= p.dw && yb >= 0 && yb + T <= p.dh;
  // columns of the lane's samples (k < 4: xs + lane + 64 k; k = 4: the halo pixel of lanes < 2H)
  int col[NS];
#pragma unroll
  for (int k = 0; k < 4; k++) col[k] = fast ? xs + lane + 64 * k : resolve_idx(xs + lane + 64 * k, p.dw, p.cbx);
  {
    const int hc = xs - H + (int)halo_pos<H>(lane < 2 * H ? (unsigned)lane : 0u);
    col[4] = fast ? hc : resolve_idx(hc, p.dw, p.cbx);
  }
  SrcView s;
  s.h = sh; s.w = sw; s.q5 = q5;
  for (int t = wave; t < T && T <= kLrMaxRows; t += 16) {
    const int rowt = fast ? yb + t : resolve_idx(yb + t, p.dh, p.cby);
    C sx[NS], sy[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) coord.get(col[k] < 0 ? 0 : col[k], rowt < 0 ? 0 : rowt, sx[k], sy[k]);
    float tx[NS], ty[NS];
    int ix[NS], iy[NS];
    unsigned interior;
    if (q5) lring_footprint<NS, 1>(s, sx, sy, tx, ty, ix, iy, interior);
    else lring_footprint<NS, 0>(s, sx, sy, tx, ty, ix, iy, interior);
    int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      // samples the filter's constant border supplies are never taken from the source
      const bool used = rowt >= 0 && col[k] >= 0 && (k < 4 || lane < 2 * H);
      const bool in = (interior >> k) & 1u;
      bad = bad || (used && !in);
      if (used && in) {
        xmn = ix[k] < xmn ? ix[k] : xmn;
        xmx = ix[k] > xmx ? ix[k] : xmx;
        ymn = iy[k] < ymn ? iy[k] : ymn;
        ymx = iy[k] > ymx ? iy[k] : ymx;
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const int a = __shfl_xor(xmn, m), b = __shfl_xor(xmx, m), c2 = __shfl_xor(ymn, m), d = __shfl_xor(ymx, m);
      xmn = a < xmn ? a : xmn;
      xmx = b > xmx ? b : xmx;
      ymn = c2 < ymn ? c2 : ymn;
      ymx = d > ymx ? d : ymx;
    }
    const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (lane == 0) {
      rymin[t] = ymn;
      rneed[t] = ymx == INT_MIN ? INT_MIN : ymx + 1;   // the bottom tap row
      rxmin[t] = xmn;
      rxmax[t] = xmx;
      rbad[t] = anybad ? 1 : 0;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  unsigned* ps = plan + (size_t)sid * kLrPlanWords;
  unsigned words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool clean = T <= kLrMaxRows;
  int x0 = INT_MAX, xhi = INT_MIN, row0 = INT_MAX, n0 = 0;
  if (clean) {
    for (int t = 0; t < T; t++) {
      if (rbad[t]) clean = false;
      x0 = rxmin[t] < x0 ? rxmin[t] : x0;
      xhi = rxmax[t] > xhi ? rxmax[t] : xhi;
    }
    // lowest[t] = the lowest source row needed by strip rows >= t (suffix minimum, in place)
    int low = INT_MAX;
    for (int t = T - 1; t >= 0; t--) {
      low = rymin[t] < low ? rymin[t] : low;
      rymin[t] = low;
    }
    row0 = low;
    // (a strip without a single sample from the source has nothing to plan)
    if (row0 == INT_MAX || xhi + 2 - x0 > kLrPitch) clean = false;
  }
  if (clean) {
    // the row loaded at iteration j (j = -D ..) is complete at iteration j + D: hi(t - D) must
    // reach need(t); the load of iteration j overwrites row hi(j) - R, which no row after j
    // may need
    const int Tb = (T + 3) & ~3;
    int idx = 0, pm = rneed[0];
    int hi = pm > row0 + 1 ? pm : row0 + 1;
    n0 = hi - row0 + 1;
    if (n0 > R) clean = false;
    for (int j = -D; j < Tb && clean; j++) {
      const int upto = j + D < T - 1 ? j + D : T - 1;
      while (idx < upto) {
        idx++;
        pm = rneed[idx] > pm ? rneed[idx] : pm;
      }
      const int bit = hi < pm ? 1 : 0;
      hi += bit;
      if (hi < pm) clean = false;
      const int lw = j + 1 < T ? rymin[j + 1 < 0 ? 0 : j + 1] : INT_MAX;
      if (hi - R >= lw) clean = false;
      const int bi = 4 + j;
      if (bit) words[bi >> 5] |= 1u << (bi & 31);
    }
  }
  ps[0] = clean ? 1u : 0u;
  ps[1] = clean ? (unsigned)x0 : 0u;
  ps[2] = clean ? (unsigned)row0 : 0u;
  ps[3] = clean ? (unsigned)n0 : 0u;
  for (int i = 0; i < 8; i++) ps[4 + i] = words[i];
  if (clean && stats) atomicAdd(stats, 1u);
}

// ------------------------------------------------------------------------- hot loop --
// younger vector-memory operations than the row load of iteration t - D when iteration t waits
// for it: the store of iteration t - D, then per iteration t - D + 1 .. t - 1 the row load (2),
// the store and - on the producer class's step - the map loads (issued in front of the row load)
template <int D, int STEP, int CLS, int MAPOPS, int STORES> struct lring_younger {
  static constexpr int maps_at(int step) {
    return (CLS == 0 ? (step & 3) == 0 : (step & 3) == 2) ? MAPOPS : 0;
  }
  static constexpr int sum() {
    int n = STORES;
    for (int i = 1; i <= D - 1; i++) n += 2 + STORES + maps_at(STEP - i + 8);
    return n;
  }
  // the first block of a strip: everything before it is drained, no stores counted
  static constexpr int first() {
    int n = 0;
    for (int i = 1; i <= D - 1 && i <= STEP; i++) n += 2 + maps_at(STEP - i + 8);
    return n;
  }
  static constexpr int value = sum(), value0 = first();
};

template <int K, bool EDGE, typename Filter, typename Coord>
__device__ __forceinline__ void wave_run_strip_lring(const WaveParams& p,
                                                     const SampleRowSrc<float, kLinear, Coord>& src,
                                                     Filter& filt, float* xp, float* rec, char* ringw,
                                                     unsigned wave, const Cols& c, int y0, int nrows,
                                                     bool writer, float* dst, const unsigned* ps) {
  using G = wave_geom<K, true>;
  using C = typename Coord::coord_t;
  constexpr int NS = 5, R = kLrR, D = kLrD, RR = kLrRec, H = G::H;
  constexpr bool kTable = coord_is_table<Coord>::value;
  constexpr int kMapOps = kTable ? 2 * NS : 0;
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  const unsigned voff = 16u * lane, moff = 4u * lane;
  float* outs = dst + ((long)y0 * p.dpitch + c.xs);   // scalar: output row 0
  const int yb = y0 - H;
  const unsigned cls = wave >> 1, sub = wave & 1u;    // producer class and row of its block
  const int X0 = (int)ps[1], row0 = (int)ps[2], n0 = (int)ps[3];
  const float* mxr = nullptr;
  const float* myr = nullptr;
  if constexpr (kTable) {
    mxr = src.coord.mx + (EDGE ? 0 : (long)yb * src.coord.pitch + c.xs);
    myr = src.coord.my + (EDGE ? 0 : (long)yb * src.coord.pitch + c.xs);
  }
  auto row_of = [&](int t) -> int {   // resolved input row of strip row t (-1 = constant border)
    if constexpr (EDGE) return resolve_idx(yb + t, p.dh, p.cby);
    else return yb + t;
  };
  const SrcView& s = src.s;
  const unsigned long long fb = (unsigned long long)src.fbase;
  const v4i rs = v4i{(int)(unsigned)fb, (int)((unsigned)(fb >> 32) & 0xffffu), (int)src.src_bytes,
                     0x00020000};
  const unsigned pitch_b = (unsigned)s.pitch * 4u;
  const unsigned hcol = halo_pos<H>(lane);
  const unsigned long long hmask = (1ull << (2 * H)) - 1ull;
  float* rlane = rec + 4u * lane;

  // ---- producer (see wave_run_strip_shared): the coordinates of this wave's row of a block
  float pm[2 * NS] = {};
  auto issue_coords = [&](int r) {
    if constexpr (kTable) {
      if constexpr (EDGE) {
        const int rr = row_of(r < T ? r : T - 1);
        const long o = (long)(rr < 0 ? 0 : rr) * src.coord.pitch;  // scalar
#pragma unroll
        for (int k = 0; k < 4; k++) pipe_load1<0>(pm[k], 4u * (unsigned)(c.uq[k] < 0 ? 0 : c.uq[k]), mxr + o);
        pipe_load1_masked<0>(pm[4], 4u * (unsigned)(c.uh < 0 ? 0 : c.uh), mxr + o, hmask);
#pragma unroll
        for (int k = 0; k < 4; k++) pipe_load1<0>(pm[NS + k], 4u * (unsigned)(c.uq[k] < 0 ? 0 : c.uq[k]), myr + o);
        pipe_load1_masked<0>(pm[NS + 4], 4u * (unsigned)(c.uh < 0 ? 0 : c.uh), myr + o, hmask);
      } else {
        const long o = (long)(r < T ? r : T - 1) * src.coord.pitch;  // scalar
        static_for<0, 4>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          pipe_load1<256 * k>(pm[k], moff, mxr + o);
        });
        pipe_load1_masked<0>(pm[4], 4u * hcol, mxr + o - H, hmask);
        static_for<0, 4>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          pipe_load1<256 * k>(pm[NS + k], moff, myr + o);
        });
        pipe_load1_masked<0>(pm[NS + 4], 4u * hcol, myr + o - H, hmask);
      }
    }
  };
  // record row r: ring byte offsets of the top / bottom tap pair, fractions
  auto publish = [&](int r) {
    C sx[NS], sy[NS];
    const int rc = r < T ? r : T - 1;
    const int rr = row_of(rc);
    if constexpr (kTable) {
#pragma unroll
      for (int k = 0; k < 2 * NS; k++) vm_pin(pm[k]);
#pragma unroll
      for (int k = 0; k < NS; k++) { sx[k] = pm[k]; sy[k] = pm[NS + k]; }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if constexpr (EDGE) src.coord.get(c.uq[k] < 0 ? 0 : c.uq[k], rr < 0 ? 0 : rr, sx[k], sy[k]);
        else src.coord.get(c.xs + (int)lane + 64 * k, rr, sx[k], sy[k]);
      }
      if constexpr (EDGE) src.coord.get(c.uh < 0 ? 0 : c.uh, rr < 0 ? 0 : rr, sx[4], sy[4]);
      else src.coord.get(c.xs - H + (int)(lane < 2u * H ? hcol : (unsigned)H), rr, sx[4], sy[4]);
    }
    float tx[NS], ty[NS];
    int ix[NS], iy[NS];
    unsigned interior;
    lring_footprint<NS, 0>(s, sx, sy, tx, ty, ix, iy, interior);
    unsigned ot[NS], ob[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
      // (a clean strip: every sample that is used lies inside; the others read offset 0)
      bool used = (interior >> k) & 1u;
      if constexpr (EDGE) used = used && rr >= 0 && (k < 4 ? c.uq[k] : c.uh) >= 0;
      if (k == 4) used = used && lane < 2u * H;
      const unsigned y = used ? (unsigned)iy[k] : 0u;
      const unsigned q = __umulhi(y, (unsigned)((0x100000000ull + R - 1) / R));
      const unsigned st = y - q * R;                 // y mod R
      const unsigned sb = st + 1 == R ? 0u : st + 1;
      const unsigned cx = used ? (unsigned)(ix[k] - X0) * 4u : 0u;
      ot[k] = used ? st * kLrPitchB + cx : 0u;
      ob[k] = used ? sb * kLrPitchB + cx : 0u;
    }
    float* slot = rlane + (unsigned)(r & 3) * RR;
    *reinterpret_cast<v4i*>(slot) = v4i{(int)ot[0], (int)ot[1], (int)ot[2], (int)ot[3]};
    *reinterpret_cast<v4i*>(slot + 256) = v4i{(int)ob[0], (int)ob[1], (int)ob[2], (int)ob[3]};
    *reinterpret_cast<v4f*>(slot + 512) = v4f{tx[0], tx[1], tx[2], tx[3]};
    *reinterpret_cast<v4f*>(slot + 768) = v4f{ty[0], ty[1], ty[2], ty[3]};
    float* rowp = rec + (unsigned)(r & 3) * RR;
    reinterpret_cast<unsigned*>(rowp + 1024)[lane] = ot[4];
    reinterpret_cast<unsigned*>(rowp + 1088)[lane] = ob[4];
    rowp[1152 + lane] = tx[4];
    rowp[1216 + lane] = ty[4];
  };

  // ---- the ring: source row n lives in slot n mod R; `slot` / `rowoff` belong to the row
  // loaded last
  const unsigned ringbase = (unsigned)(unsigned long long)ringw;   // LDS byte address
  unsigned rb = ringbase;
  asm volatile("" : "+s"(rb));
  unsigned slot = (unsigned)row0 % (unsigned)R;
  unsigned rowoff = ((unsigned)row0 * (unsigned)s.pitch + (unsigned)X0) * 4u;
  auto dma_row = [&]() {
    const unsigned m0v = ringbase + slot * kLrPitchB;
    unsigned long long sv;
    asm volatile(IPA_SGPR_HAZARD "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen offset:1024 lds\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(sv)
                 : "s"(m0v), "v"(voff), "s"(rs), "s"(rowoff), "n"((1 << kLrTail) - 1)
                 : "memory");
  };
  auto advance = [&](unsigned bit) {
    slot += bit;
    slot = slot == (unsigned)R ? 0u : slot;
    rowoff += bit * pitch_b;
  };

  // prologue: this wave's record row of blocks 0 / 1, the first n0 source rows and the D loads
  // that precede row 0; everything drained once
  issue_coords((int)wave);
  dma_row();
  for (int i = 1; i < n0; i++) {
    advance(1u);
    dma_row();
  }
  unsigned long long sched = *reinterpret_cast<const unsigned long long*>(ps + 4);
  sched >>= 4 - D;
#pragma unroll
  for (int j = 0; j < D; j++) {
    advance((unsigned)sched & 1u);
    sched >>= 1;
    dma_row();
  }
  int nibs = 15, word = 1;   // nibbles left in `sched`, next 64-bit word of the schedule
  vm_wait<0>();
  publish((int)wave);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto step = [&](auto St, int tb, unsigned bit) {
    constexpr int STEP = decltype(St)::value;
    const int t = tb + STEP;
    if constexpr ((STEP & 1) == 0) {
      // block barrier: behind it the records of this block are in the ring and nobody reads
      // the block before the previous one any more
#ifndef IPA_LR_NO_BARRIER   // (measurement only: racy without it)
      if (tb != 0 || STEP != 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    }
    // 1. the source row loaded at iteration t - D (and everything older)
    const bool full = t - D >= K - 1 && t - 1 < T;
    if (tb == 0) {
      if (cls == 0) vm_wait<lring_younger<D, STEP, 0, kMapOps, 0>::value0>();
      else vm_wait<lring_younger<D, STEP, 1, kMapOps, 0>::value0>();
    } else if (cls == 0) {
      if (full) vm_wait<lring_younger<D, STEP, 0, kMapOps, 1>::value>();
      else vm_wait<lring_younger<D, STEP, 0, kMapOps, 0>::value>();
    } else {
      if (full) vm_wait<lring_younger<D, STEP, 1, kMapOps, 1>::value>();
      else vm_wait<lring_younger<D, STEP, 1, kMapOps, 0>::value>();
    }
    asm volatile("" ::: "memory");
    // 2. producers: the records of the next block of this wave's class (its map loads were
    //    issued three iterations ago: younger = three row loads and up to three stores)
    if constexpr (STEP == 3 || STEP == 1) {
      if (cls == (STEP == 3 ? 0u : 1u) && (STEP == 3 || tb != 0)) {
        if constexpr (kTable && D > 3) {
          if (t - 3 >= K - 1 && t - 1 < T) vm_wait<9>();
          else vm_wait<6>();
        }
#ifndef IPA_LR_NO_PUBLISH   // (measurement only: WRONG results)
        publish(STEP == 3 ? tb + 4 + (int)sub : tb + 2 + (int)sub);
#endif
      }
    }
    // 3. the footprints of row t and their taps from the ring
    const float* slotp = rlane + (unsigned)STEP * RR;
    const float* rowp = rec + (unsigned)STEP * RR;
    const v4i qt = *reinterpret_cast<const v4i*>(slotp);
    const v4i qb = *reinterpret_cast<const v4i*>(slotp + 256);
    const v4f qx = *reinterpret_cast<const v4f*>(slotp + 512);
    const v4f qy = *reinterpret_cast<const v4f*>(slotp + 768);
    const unsigned ot[NS] = {(unsigned)qt.x, (unsigned)qt.y, (unsigned)qt.z, (unsigned)qt.w,
                             reinterpret_cast<const unsigned*>(rowp + 1024)[lane]};
    const unsigned ob[NS] = {(unsigned)qb.x, (unsigned)qb.y, (unsigned)qb.z, (unsigned)qb.w,
                             reinterpret_cast<const unsigned*>(rowp + 1088)[lane]};
    const float tx[NS] = {qx.x, qx.y, qx.z, qx.w, rowp[1152 + lane]};
    const float ty[NS] = {qy.x, qy.y, qy.z, qy.w, rowp[1216 + lane]};
    float cur[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
      // (one add per address: the ring's LDS address is ONE scalar the compiler cannot take apart)
      typedef const __attribute__((address_space(3))) float* lds_f32;
      lds_f32 pt = (lds_f32)(unsigned long)(rb + ot[k]);
      lds_f32 pb = (lds_f32)(unsigned long)(rb + ob[k]);
#ifdef IPA_LR_NO_TAPS      // (measurement only: WRONG results)
      const float v00 = __uint_as_float(ot[k] | 0x3f000000u), v01 = v00, v10 = __uint_as_float(ob[k] | 0x3f000000u), v11 = v10;
      (void)pt; (void)pb;
#else
      const float v00 = pt[0], v01 = pt[1], v10 = pb[0], v11 = pb[1];
#endif
      const float wx0 = 1.f - tx[k], wx1 = tx[k], wy0 = 1.f - ty[k], wy1 = ty[k];
      float r0 = wx0 * v00;
      r0 = ipa_fma(wx1, v01, r0);
      float o = wy0 * r0;
      float r1 = wx0 * v10;
      r1 = ipa_fma(wx1, v11, r1);
      cur[k] = ipa_fma(wy1, r1, o);
    }
    if constexpr (EDGE) {
      // positions the filter's constant border supplies
      const int rowt = row_of(t < T ? t : T - 1);
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = (rowt < 0 || c.uq[k] < 0) ? src.ccval : cur[k];
      cur[4] = (rowt < 0 || c.uh < 0) ? src.ccval : cur[4];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) xp[kRowPad + 64u * k + lane] = cur[k];
    if (lane < 2u * H) xp[kRowPad - H + hcol] = cur[4];
    // 4. the taps are in registers (the blend consumed them): this iteration's row load may
    //    overwrite the ring's oldest row.  Producers request their next row in front of it.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (STEP == 0 || STEP == 2) {
#ifndef IPA_LR_NO_PUBLISH
      if (cls == (STEP == 0 ? 0u : 1u)) issue_coords(STEP == 0 ? tb + 4 + (int)sub : tb + 6 + (int)sub);
#endif
    }
    advance(bit);
#ifndef IPA_LR_NO_DMA      // (measurement only: WRONG results)
    dma_row();
#endif
    __builtin_amdgcn_wave_barrier();
    // 5. filter + store
#ifdef IPA_LR_NO_FILTER    // (measurement only: WRONG results)
    const v4f q = *reinterpret_cast<const v4f*>(xp + kRowPad + 4u * lane);
#else
    const v4f q = filt.template row<EDGE>(xp, lane, c);
#endif
    const int o = t - (K - 1);
    if (o >= 0 && o < nrows) {
      if (writer) pipe_store4<true>(q, voff, outs + (long)o * p.dpitch);
    }
    __builtin_amdgcn_wave_barrier();
  };

  const int Tb = (T + 3) & ~3;
  int tb = 0;
#pragma unroll 1
  do {
    if (nibs == 0) {
      // (a scalar load by hand: the compiler's own choice here was a vector load, which would
      // take part in the counted vmcnt waits)
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)"
                   : "=s"(sched) : "s"(ps + 4 + 2 * word) : "memory");
      word++;
      nibs = 16;
    }
    const unsigned nib = (unsigned)sched & 15u;
    sched >>= 4;
    nibs--;
    static_for<0, 4>([&](auto St) { step(St, tb, (nib >> decltype(St)::value) & 1u); });
    tb += 4;
  } while (tb < Tb);
  // LDS-DMA still in flight would land in the LDS of the workgroup that comes after this one
  vm_wait<0>();
}

// kernel: the waves of a workgroup are IPA_WPB frames of one strip (WaveParams::frames_wg);
// clean strips on the ring loop, the others on wave_run_strip_shared (aligned geometry)
template <typename Src, int K>
__global__ void __launch_bounds__(64 * IPA_WPB, 2)
lring_stencil_kernel(WaveParams p, Src src, Weights<float, K * K> wts, const unsigned* plan) {
  static_assert(IPA_WPB == 4, "two producer classes of two waves");
  using G = wave_geom<K, true>;
  __shared__ __attribute__((aligned(16))) char lds[kLrLdsBytes];
  const int lane = threadIdx.x & 63;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned groups = (unsigned)p.frames_inner / IPA_WPB;
  const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned frame = (b % groups) * IPA_WPB + wave;
  const unsigned sid = b / groups;
  if (sid >= p.strips) return;
  float* rec = reinterpret_cast<float*>(lds);
  float* xp = reinterpret_cast<float*>(lds + kLrRecBytes) + wave * kRowStride;
  char* ringw = lds + kLrRecBytes + kLrXpBytes + wave * kLrRingBytes;
  // (the gather loop's record ring of 8 x 1024 floats lies over the rings)
  float* mapring = reinterpret_cast<float*>(lds + kLrRecBytes + kLrXpBytes);
  static_assert(2 * IPA_WPB * ring_row<true>::value * 4 <= IPA_WPB * kLrRingBytes, "gather loop's records");
  const int syi = (int)(sid / (unsigned)p.strips_x), sxi = (int)sid - syi * p.strips_x;
  src.set_frame(frame);
  const int xs = sxi * 256;
  Cols c;
  c.xs = xs;
  c.xo = xs + lane * 4;
  const int y0 = syi * p.strip_h;
  const int nrows = p.dh - y0 < p.strip_h ? p.dh - y0 : p.strip_h;
  const bool writer = c.xo < p.dw;
  float* dst = reinterpret_cast<float*>(p.dst) + (long)frame * p.dst_frame_elems;
  const int T = nrows + K - 1;
  const bool fast = xs - G::H >= 0 && xs + 256 + G::H <= p.dw && y0 - G::H >= 0 && y0 - G::H + T <= p.dh;
  const unsigned* ps = plan + (size_t)sid * kLrPlanWords;
  const bool clean = ps[0] != 0u;
  DenseFilter<K> filt(wts);
  if (fast) {
#pragma unroll
    for (int k = 0; k < 4; k++) c.uu[k] = c.xo + k;
    if (clean) wave_run_strip_lring<K, false>(p, src, filt, xp, rec, ringw, wave, c, y0, nrows, writer, dst, ps);
    else wave_run_strip_shared<K, 0, false, true>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      c.uu[k] = resolve_idx(c.xo + k, p.dw, p.cbx);
      c.uq[k] = resolve_idx(xs + lane + 64 * k, p.dw, p.cbx);
    }
    c.uh = resolve_idx(xs - G::H + (int)halo_pos<G::H>(lane < 2 * G::H ? (unsigned)lane : 0u), p.dw, p.cbx);
    if (clean) wave_run_strip_lring<K, true>(p, src, filt, xp, rec, ringw, wave, c, y0, nrows, writer, dst, ps);
    else wave_run_strip_shared<K, 0, true, true>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
  }
}

// which launches take the ring kernel: float32 frames, bilinear, exact coordinates, 3x3 / 5x5
template <typename Src, int K> struct lring_capable : std::false_type {};
template <typename Coord, int K> struct lring_capable<SampleRowSrc<float, kLinear, Coord>, K> {
  static constexpr bool value = (K == 3 || K == 5) && IPA_PIPE != 0 && IPA_PIPE_SHARED != 0 && IPA_WPB == 4;
};

}  // namespace ipa
