// wave_pipe.hpp — the FAST strips of wave_stencil.hpp with a HAND-SCHEDULED memory pipeline
// (round 3).  Included at the end of wave_stencil.hpp.
//
// Why: hipcc treats a wave's loads and stores as completing out of order with respect to each
// other (both count in vmcnt), so as soon as a store is in flight every wait for a load becomes
// `s_waitcnt vmcnt(0)` - a wave that stores can never have a load in flight ACROSS a use of an
// older one, and the chunked form of wave_run_strip (issue D rows, drain, consume D rows) is
// the best the compiler can do.  On gfx950 vector-memory operations of one wave retire in issue
// order, loads and stores alike, and instructions issued with EXEC = 0 are counted too
// (tools/pipe_micro.hip: 0 wrong values in 2 x 2.1e8 counted waits).  Here every vector-memory
// instruction of the strip loop is inline asm, invisible to the compiler's scoreboard, and the
// waits are counted by hand:
//
//   * the wait for an operation X is `s_waitcnt vmcnt(N)` with N <= the number of operations
//     issued after X (a smaller N only waits longer, never too short);
//   * each wait statement carries the registers it releases as in/out operands, so no use can be
//     scheduled above it;
//   * loads past the strip's last input row are clamped to that row (dummy re-loads) so that the
//     counts do not depend on the position in the strip; the only run-time part of a count is
//     whether the output stores have started (the first K - 1 rows of a strip store nothing).
//
// plain rows (LoadRowSrc): P rows in flight per wave, rolling (iteration t: wait row t, issue
// row t + P, filter, store row t - K + 1).
//
// sampling source (float32 frames, bilinear, coordinate table): per output row a wave issues
//   8 coalesced map dwords (row t + 3), 16 dword gathers (row t + 1), 1 store (row t - K + 1)
// and consumes what it issued one iteration earlier.  VERTICAL TAP REUSE: the top tap row of
// sample row t + 1 is the bottom tap row of sample row t wherever the footprint moved straight
// down (element offset e' == e + pitch: the same two memory words).  The bottom-row registers of
// row t become the top-row registers of row t + 1 and the top gather is issued under an EXEC mask
// of the lanes where that fails (often none: the instruction is then issued with EXEC = 0 and
// costs an issue slot, no address or cache work).  Same words, same blend: identical bits.
//
// Reference semantics: as wave_stencil.hpp (camera/LensDistortion.py:323-326 followed by a K x K
// filter; filters/maskedConvolve.py:24-43 for the plain filter).
#pragma once

namespace ipa {

typedef int v4i __attribute__((ext_vector_type(4)));

#ifndef IPA_PIPE
#define IPA_PIPE 1
#endif
#ifndef IPA_PIPE_ROWS
#define IPA_PIPE_ROWS 4   // plain rows in flight per wave
#endif
#ifndef IPA_PIPE_REUSE
#define IPA_PIPE_REUSE 1  // vertical tap reuse of the sampling source
#endif

// ------------------------------------------------------------------ asm primitives --
template <int N> __device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
// releases registers after a wait: no instruction, only the dependence
// (the comment names the registers for tools/check_pipe_asm.py)
__device__ __forceinline__ void vm_pin(float (&a)[8]) {
  asm volatile("; pin %0 %1 %2 %3 %4 %5 %6 %7"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),
                 "+v"(a[6]), "+v"(a[7]));
}
__device__ __forceinline__ void vm_pin(v4f& a) { asm volatile("; pin %0" : "+v"(a)); }

__device__ __forceinline__ void pipe_load4(v4f& x, unsigned voff, const float* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(x) : "v"(voff), "s"(sbase));
}
template <int OFF> __device__ __forceinline__ void pipe_load1(float& x, unsigned voff,
                                                              const float* sbase) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(x) : "v"(voff), "s"(sbase), "n"(OFF));
}
template <bool NT> __device__ __forceinline__ void pipe_store4(const v4f& x, unsigned voff,
                                                               float* sbase) {
  // (the s_nop: a VALU write of the data registers directly behind a 128-bit store needs one
  // wait state the compiler's hazard pass cannot see through the asm)
  if constexpr (NT)
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" ::"v"(voff), "v"(x), "s"(sbase));
  else
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 0" ::"v"(voff), "v"(x), "s"(sbase));
}
// the two dwords of a bilinear tap row at byte offset `off` of the frame (range-checked)
__device__ __forceinline__ void pipe_gather2(float& a, float& b, unsigned off, v4i rs) {
#ifdef IPA_DEBUG_NO_GATHER   // measurement only (WRONG results): taps from arithmetic
  a = __uint_as_float(off | 0x3f000000u);
  b = a;
  return;
#endif
#ifdef IPA_DEBUG_ONE_DWORD   // measurement only (WRONG results): one dword per tap row
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(a) : "v"(off), "s"(rs));
  b = a;
  return;
#endif
  asm volatile("buffer_load_dword %0, %2, %3, 0 offen\n\t"
               "buffer_load_dword %1, %2, %3, 0 offen offset:4"
               : "=v"(a), "=v"(b)
               : "v"(off), "s"(rs));
}
// the same under an EXEC mask: lanes outside `m` keep a and b
__device__ __forceinline__ void pipe_gather2_masked(float& a, float& b, unsigned off, v4i rs,
                                                    unsigned long long m) {
  unsigned long long sv;
#ifdef IPA_DEBUG_NO_GATHER
  a = __uint_as_float(off | 0x3f000000u);
  b = a;
  return;
#endif
#ifdef IPA_DEBUG_ONE_DWORD
  asm volatile("s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, %4\n\t"
               "buffer_load_dword %0, %2, %3, 0 offen\n\t"
               "s_mov_b64 exec, %1"
               : "+v"(a), "=&s"(sv)
               : "v"(off), "s"(rs), "s"(m));
  b = a;
  return;
#endif
  asm volatile("s_mov_b64 %2, exec\n\t"
               "s_mov_b64 exec, %5\n\t"
               "buffer_load_dword %0, %3, %4, 0 offen\n\t"
               "buffer_load_dword %1, %3, %4, 0 offen offset:4\n\t"
               "s_mov_b64 exec, %2"
               : "+v"(a), "+v"(b), "=&s"(sv)
               : "v"(off), "s"(rs), "s"(m));
}

// ------------------------------------------------------------------ the K x K row step --
// sample row (LDS, natural pixel order) -> the K running rows; returns the completed row
template <int K>
__device__ __forceinline__ v4f pipe_filter_row(const Weights<float, K * K>& wts, const float* xp,
                                               unsigned lane, unsigned lane4_opaque,
                                               v2f (&acc)[K][2]) {
  using G = wave_geom<K>;
  const float* wp = xp + kRowPad - G::H + 4u * lane;
  const float* wq = xp + kRowPad - G::H + lane4_opaque;
  v2f pair[K + 2];
#pragma unroll
  for (int m = 0; m < K + 2; m++) pair[m] = (m & 1) ? v2f{wq[m], wq[m + 1]} : v2f{wp[m], wp[m + 1]};
  // coefficient n = i K + j is one half of the SGPR pair {w[n & ~1], w[(n & ~1) + 1]}, broadcast
  // by op_sel: K K / 2 scalar pairs instead of the K K {w, w} pairs the compiler forms (which
  // spill to VGPR lanes: 45 v_readlane per row in the first build of this loop)
  static_for<0, K>([&](auto Ii) {
    constexpr int i = K - 1 - decltype(Ii)::value;
    static_for<0, K>([&](auto Jj) {
      constexpr int j = decltype(Jj)::value;
      constexpr int n = i * K + j, n0 = n & ~1, n1 = n0 + 1 < K * K ? n0 + 1 : n0;
      const v2f wp2 = v2f{wts.w[n0], wts.w[n1]};
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if constexpr (i == 0) {
          if constexpr (j == 0) acc[0][h] = pk_mul_coef<0>(wp2, pair[2 * h]);
          else acc[0][h] = pk_fma_coef<(n & 1)>(wp2, pair[j + 2 * h], acc[0][h]);
        } else {
          if constexpr (j == 0) acc[i][h] = pk_fma_coef<(n & 1)>(wp2, pair[2 * h], acc[i - 1][h]);
          else acc[i][h] = pk_fma_coef<(n & 1)>(wp2, pair[j + 2 * h], acc[i][h]);
        }
      }
    });
  });
  return v4f{acc[K - 1][0].x, acc[K - 1][0].y, acc[K - 1][1].x, acc[K - 1][1].y};
}

// ------------------------------------------------------------------ plain rows --
template <int K>
__device__ __forceinline__ void wave_run_strip_pipe(const WaveParams& p, const LoadRowSrc& src,
                                                    const Weights<float, K * K>& wts, float* xp,
                                                    const Cols& c, int y0, int nrows, bool writer,
                                                    float* dst) {
  using G = wave_geom<K>;
  constexpr int P = IPA_PIPE_ROWS;
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  unsigned lane4_opaque = 4u * lane;
  asm volatile("" : "+v"(lane4_opaque));
  const unsigned voff = 16u * lane;
  const float* rows = src.base + ((long)(y0 - G::H) * src.pitch + c.xs);  // scalar: input row 0
  float* outs = dst + ((long)y0 * p.dpitch + c.xs);                      // scalar: output row 0
  v4f buf[P];
  static_for<0, P>([&](auto U) {
    constexpr int u = decltype(U)::value;
    pipe_load4(buf[u], voff, rows + (long)(u < T ? u : T - 1) * src.pitch);
  });
  v2f acc[K][2];
  int tb = 0;
#pragma unroll 1
  do {
    static_for<0, P>([&](auto U) {
      constexpr int u = decltype(U)::value;
      const int t = tb + u;
      if (t < T) {
        // younger than load(t): loads t+1 .. t+P-1, and the stores of iterations t-P .. t-1
        // (iteration j stores when j >= K - 1)
        if (t >= P + K - 1) vm_wait<2 * P - 1>();
        else vm_wait<P - 1>();
        vm_pin(buf[u]);
        *reinterpret_cast<v4f*>(xp + kRowPad + 4u * lane) = buf[u];
        const int tn = t + P < T ? t + P : T - 1;
        pipe_load4(buf[u], voff, rows + (long)tn * src.pitch);
        __builtin_amdgcn_wave_barrier();
        const v4f q = pipe_filter_row<K>(wts, xp, lane, lane4_opaque, acc);
        const int o = t - (K - 1);
        if (o >= 0) {
          if (writer) pipe_store4<false>(q, voff, outs + (long)o * p.dpitch);
        }
        __builtin_amdgcn_wave_barrier();
      }
    });
    tb += P;
  } while (tb < T);
}

// ------------------------------------------------------------------ sampling source --
// float32 frames, bilinear, coordinates from a table (the maps of LensDistortion.correct)
template <typename Src, int K> struct pipe_capable : std::false_type {};
template <int K> struct pipe_capable<LoadRowSrc, K> : std::true_type {};
template <typename Coord, int K> struct pipe_capable<SampleRowSrc<float, kLinear, Coord>, K> {
  static constexpr bool value = SampleRowSrc<float, kLinear, Coord>::template depth<K>::kPiped;
};

template <int K, int QM, typename Coord>
__device__ __forceinline__ void wave_run_strip_pipe(const WaveParams& p,
                                                    const SampleRowSrc<float, kLinear, Coord>& src,
                                                    const Weights<float, K * K>& wts, float* xp,
                                                    const Cols& c, int y0, int nrows, bool writer,
                                                    float* dst) {
  using G = wave_geom<K>;
  using C = typename Coord::coord_t;
  static_assert(sizeof(C) == 4, "float32 coordinate tables");
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  unsigned lane4_opaque = 4u * lane;
  asm volatile("" : "+v"(lane4_opaque));
  const unsigned voff = 16u * lane, moff = 4u * lane;
  float* outs = dst + ((long)y0 * p.dpitch + c.xs);  // scalar: output row 0
#ifdef IPA_DEBUG_ALIGN_STORES   // measurement only (WRONG results): every store a whole 128-byte line run
  outs = (float*)((unsigned long long)outs & ~127ull);
  writer = true;
#endif
  const int yb = y0 - G::H;                          // first input row of the strip
  const float* mxr = src.coord.mx + ((long)yb * src.coord.pitch + c.xs);
  const float* myr = src.coord.my + ((long)yb * src.coord.pitch + c.xs);
  // raw buffer descriptor of the frame (what make_rsrc builds), as four scalars for the asm
  const SrcView& s = src.s;
  const unsigned long long fb = (unsigned long long)src.fbase;
  const v4i rs = v4i{(int)(unsigned)fb, (int)((unsigned)(fb >> 32) & 0xffffu), (int)src.src_bytes,
                     0x00020000};
  const unsigned pitch_b = (unsigned)s.pitch * 4u;

  // map row r (clamped to the strip) -> m[0..3] = x of pixels lane + 64 k, m[4..7] = y
  auto issue_map = [&](float (&m)[8], int r) {
#ifdef IPA_DEBUG_NO_MAP   // measurement only (WRONG results): coordinates from arithmetic, no map traffic
    static_for<0, 4>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      const float fx = (float)(c.xs + (int)lane + 64 * k), fy = (float)(yb + (r < T ? r : T - 1));
      m[k] = fx * 0.97f + 40.f + fy * 0.004f;
      m[4 + k] = fy * 0.97f + 30.f + fx * 0.002f;
      asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
    });
    // (8 no-ops stand in for the 8 loads so that the counted waits stay valid: nothing is
    // counted for them, so the waits below are merely more conservative)
    return;
#endif
    const long o = (long)(r < T ? r : T - 1) * src.coord.pitch;  // scalar
    static_for<0, 4>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      pipe_load1<256 * k>(m[k], moff, mxr + o);
    });
    static_for<0, 4>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      pipe_load1<256 * k>(m[4 + k], moff, myr + o);
    });
  };
  // footprints of a map row: fractions, byte offsets of the top-left taps, interior bits
  auto footprint = [&](const float (&m)[8], float (&tx)[4], float (&ty)[4], unsigned (&off)[4],
                       unsigned& interior) {
    const float sx[4] = {m[0], m[1], m[2], m[3]}, sy[4] = {m[4], m[5], m[6], m[7]};
    int e[4];
    batch_footprint_linear<4, QM>(s, sx, sy, tx, ty, e, interior);
#pragma unroll
    for (int k = 0; k < 4; k++) off[k] = (unsigned)e[k] << 2;
  };

  // vector-memory operations per row (the counted waits below)
#ifdef IPA_DEBUG_NO_MAP
  constexpr int kMapOps = 0;
#else
  constexpr int kMapOps = 8;
#endif
#if defined(IPA_DEBUG_NO_GATHER)
  constexpr int kTapOps = 0;
#elif defined(IPA_DEBUG_ONE_DWORD)
  constexpr int kTapOps = 8;
#else
  constexpr int kTapOps = 16;
#endif
  float m[8];            // map row in flight / being consumed
  float ga[8], gb[8];    // tap rows: [2k], [2k+1] = the two dwords of footprint k
  float txa[4], tya[4], txb[4], tyb[4];
  unsigned offa[4], offb[4], ina, inb;
  v2f acc[K][2];

  // prologue: rows 0 and 1 resolved, gathers of row 0 and map row 2 in flight
  issue_map(m, 0);
  vm_wait<0>();
  vm_pin(m);
  footprint(m, txa, tya, offa, ina);
  issue_map(m, 1);
#pragma unroll
  for (int k = 0; k < 4; k++) pipe_gather2(ga[2 * k], ga[2 * k + 1], offa[k], rs);
#pragma unroll
  for (int k = 0; k < 4; k++) pipe_gather2(gb[2 * k], gb[2 * k + 1], offa[k] + pitch_b, rs);
  vm_wait<kTapOps>();
  vm_pin(m);
  footprint(m, txb, tyb, offb, inb);
  issue_map(m, 2);
  // from here on, in issue order: ... gathers(t) [16], store(t-1)?, map(t+2) [8] | iteration t

  // one iteration: TOP / BOT = tap-row registers of row t (top, bottom); the bottom registers
  // become the top registers of row t + 1
  auto step = [&](int t, float (&top)[8], float (&bot)[8], const float (&tx)[4],
                  const float (&ty)[4], const unsigned (&off)[4], unsigned interior,
                  float (&txn)[4], float (&tyn)[4], unsigned (&offn)[4], unsigned& interiorn,
                  float (&txnn)[4], float (&tynn)[4], unsigned (&offnn)[4], unsigned& interiornn) {
    // 1. the gathers of row t: younger = [store of iteration t-1] + map(t+2)
    // (the branch holds operand-less waits only: with the registers as operands of two
    // alternative statements the compiler merges them through copies, and a copy of a register
    // whose load is still in flight reads garbage)
    if (t >= K) vm_wait<kMapOps + 1>();
    else vm_wait<kMapOps>();
    vm_pin(top);
    vm_pin(bot);
    // 2. blend (the arithmetic and order of batch_blend_one) -> LDS row, natural pixel order
    float cur[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float wx0 = 1.f - tx[k], wx1 = tx[k], wy0 = 1.f - ty[k], wy1 = ty[k];
      float r0 = wx0 * top[2 * k];
      r0 = ipa_fma(wx1, top[2 * k + 1], r0);
      float o = wy0 * r0;
      float r1 = wx0 * bot[2 * k];
      r1 = ipa_fma(wx1, bot[2 * k + 1], r1);
      cur[k] = ipa_fma(wy1, r1, o);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) xp[kRowPad + 64u * k + lane] = cur[k];
    if (__builtin_amdgcn_ballot_w64(interior != 0xfu)) {
      // footprints touching the source border (rare): redo them tap by tap, straight into the
      // LDS row - ONE copy of the border-aware sampler per step (a loop, not unrolled)
#pragma unroll 1
      for (int k = 0; k < 4; k++) {
        if (!((interior >> k) & 1u)) {
          float sx, sy;
          src.coord.get(c.xs + (int)lane + 64 * k, yb + t, sx, sy);
          xp[kRowPad + 64u * k + lane] = sample<float, kLinear, float>(s, sx, sy, src.cval);
        }
      }
    }
    // 3. gathers of row t + 1: its top row into `bot` under the mask of the lanes whose
    //    footprint did not move straight down, its bottom row into `top`
#pragma unroll
    for (int k = 0; k < 4; k++) {
#if IPA_PIPE_REUSE
      const unsigned long long need = __builtin_amdgcn_ballot_w64(offn[k] != off[k] + pitch_b);
      pipe_gather2_masked(bot[2 * k], bot[2 * k + 1], offn[k], rs, need);
#else
      pipe_gather2(bot[2 * k], bot[2 * k + 1], offn[k], rs);
#endif
    }
#pragma unroll
    for (int k = 0; k < 4; k++) pipe_gather2(top[2 * k], top[2 * k + 1], offn[k] + pitch_b, rs);
    __builtin_amdgcn_wave_barrier();
    // 4. filter + store
#ifdef IPA_DEBUG_NO_FILTER   // measurement only (WRONG results): the sample row goes straight out
    const v4f q = *reinterpret_cast<const v4f*>(xp + kRowPad + 4u * lane);
#else
    const v4f q = pipe_filter_row<K>(wts, xp, lane, lane4_opaque, acc);
#endif
    const int o = t - (K - 1);
#ifdef IPA_DEBUG_NO_STORE   // measurement only: one store per strip (keeps the work alive)
    if (o == 0) {
#else
    if (o >= 0) {
#endif
      if (writer) pipe_store4<true>(q, voff, outs + (long)o * p.dpitch);
    }
    __builtin_amdgcn_wave_barrier();
    // 5. map row t + 2: younger = gathers(t+1) [16] + this iteration's store
    if (t >= K - 1) vm_wait<kTapOps + 1>();
    else vm_wait<kTapOps>();
    vm_pin(m);
    footprint(m, txnn, tynn, offnn, interiornn);
    issue_map(m, t + 3);
  };

  // rows rotate through three footprint sets (t, t+1, t+2) and two tap-register roles
  float txc[4], tyc[4];
  unsigned offc[4], inc = 0xfu;
  // (a do-while: T >= K, and with no path around the loop the prologue's loads provably flow
  // into it - tools/check_pipe_asm.py follows the control-flow graph)
  int tb = 0;
#pragma unroll 1
  do {
    // t = tb: top ga, bottom gb; sets a (t), b (t+1), c (t+2)
    if (tb + 0 < T) step(tb + 0, ga, gb, txa, tya, offa, ina, txb, tyb, offb, inb, txc, tyc, offc, inc);
    if (tb + 1 < T) step(tb + 1, gb, ga, txb, tyb, offb, inb, txc, tyc, offc, inc, txa, tya, offa, ina);
    if (tb + 2 < T) step(tb + 2, ga, gb, txc, tyc, offc, inc, txa, tya, offa, ina, txb, tyb, offb, inb);
    if (tb + 3 < T) step(tb + 3, gb, ga, txa, tya, offa, ina, txb, tyb, offb, inb, txc, tyc, offc, inc);
    if (tb + 4 < T) step(tb + 4, ga, gb, txb, tyb, offb, inb, txc, tyc, offc, inc, txa, tya, offa, ina);
    if (tb + 5 < T) step(tb + 5, gb, ga, txc, tyc, offc, inc, txa, tya, offa, ina, txb, tyb, offb, inb);
    tb += 6;
  } while (tb < T);
}

}  // namespace ipa
