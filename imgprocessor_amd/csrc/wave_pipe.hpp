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
#ifndef IPA_PIPE_STAY
#define IPA_PIPE_STAY 0   // ... the footprint-did-not-move case from registers too (measured: no gain, 1.035 vs 1.034 ms)
#endif
#ifndef IPA_SHARE_TAPS
#define IPA_SHARE_TAPS 0   // 9: measurement only (WRONG results): no right-hand tap gathers at all, the
                           // right tap of a footprint = the left tap of the neighbouring lane (DPP)
#endif
#ifndef IPA_DEBUG_HALO_LEVEL
#define IPA_DEBUG_HALO_LEVEL 0   // measurement only (WRONG results): 1 = no halo gathers, 2 = no halo work at all
#endif
#ifndef IPA_LANE_NATURAL
#define IPA_LANE_NATURAL 0   // shared-record loop: sample k of lane L is strip pixel 4 L + k instead of L + 64 k
#endif

// ------------------------------------------------------------------ asm primitives --
// A gfx9-family hazard the compiler cannot guard for us: a vector-memory instruction that reads
// an SGPR (row base, buffer descriptor) a VALU instruction has written within the last 5 wait
// states sees the OLD value.  hipcc inserts the no-ops for the vector-memory instructions it
// emits itself; inline asm is opaque to its hazard recognizer - and under SGPR pressure it
// restores exactly these scalars with v_readlane_b32 (VALU writes SGPR) right in front of our
// statements.  Symptom (round 3): wrong first rows of strips in the 7x7 kernels (800 spilled
// SGPRs), non-deterministic, while kernels with fewer spills ran clean.  Every statement that
// carries a vector-memory instruction therefore opens with the five wait states.
#define IPA_SGPR_HAZARD "s_nop 4\n\t"

template <int N> __device__ __forceinline__ void vm_wait() {
#ifdef IPA_DEBUG_WAIT0   // debugging: every counted wait drains the queue
  asm volatile("s_waitcnt vmcnt(0)" ::"n"(N));
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
#endif
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
  asm volatile(IPA_SGPR_HAZARD "global_load_dwordx4 %0, %1, %2" : "=v"(x) : "v"(voff), "s"(sbase));
}
template <int OFF> __device__ __forceinline__ void pipe_load1(float& x, unsigned voff,
                                                              const float* sbase) {
  asm volatile(IPA_SGPR_HAZARD "global_load_dword %0, %1, %2 offset:%3" : "=v"(x) : "v"(voff), "s"(sbase), "n"(OFF));
}
// one dword of the lanes in `m` (the halo lanes); the others keep x
template <int OFF> __device__ __forceinline__ void pipe_load1_masked(float& x, unsigned voff,
                                                                     const float* sbase,
                                                                     unsigned long long m) {
  unsigned long long sv;
  asm volatile(IPA_SGPR_HAZARD "s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, %4\n\t"
               "global_load_dword %0, %2, %3 offset:%5\n\t"
               "s_mov_b64 exec, %1"
               : "+v"(x), "=&s"(sv)
               : "v"(voff), "s"(sbase), "s"(m), "n"(OFF));
}
__device__ __forceinline__ void vm_pin(float& a) { asm volatile("; pin %0" : "+v"(a)); }
// the same for tables of 8-byte entries (double coordinates stored by stored_coords.hpp)
template <int OFF> __device__ __forceinline__ void pipe_load1(double& x, unsigned voff,
                                                              const double* sbase) {
  asm volatile(IPA_SGPR_HAZARD "global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(x) : "v"(voff), "s"(sbase), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void pipe_load1_masked(double& x, unsigned voff,
                                                                     const double* sbase,
                                                                     unsigned long long m) {
  unsigned long long sv;
  asm volatile(IPA_SGPR_HAZARD "s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, %4\n\t"
               "global_load_dwordx2 %0, %2, %3 offset:%5\n\t"
               "s_mov_b64 exec, %1"
               : "+v"(x), "=&s"(sv)
               : "v"(voff), "s"(sbase), "s"(m), "n"(OFF));
}
__device__ __forceinline__ void vm_pin(double& a) { asm volatile("; pin %0" : "+v"(a)); }
template <bool NT> __device__ __forceinline__ void pipe_store4(const v4f& x, unsigned voff,
                                                               float* sbase) {
  // (the s_nop: a VALU write of the data registers directly behind a 128-bit store needs one
  // wait state the compiler's hazard pass cannot see through the asm)
#ifndef IPA_STORE_FLAVOUR
#define IPA_STORE_FLAVOUR "nt"   // cache policy of the sampling kernels' output stores (A/B: "", "sc1", "sc0 sc1", "nt sc1")
#endif
  if constexpr (NT)
    asm volatile(IPA_SGPR_HAZARD "global_store_dwordx4 %0, %1, %2 " IPA_STORE_FLAVOUR "\n\ts_nop 0" ::"v"(voff), "v"(x), "s"(sbase));
  else
    asm volatile(IPA_SGPR_HAZARD "global_store_dwordx4 %0, %1, %2\n\ts_nop 0" ::"v"(voff), "v"(x), "s"(sbase));
}
// four uint16 results of a lane (two dwords) - the strip remap of uint16 frames into uint16 (CV16 below)
__device__ __forceinline__ void pipe_store2(unsigned lo, unsigned hi, unsigned voff, uint16_t* sbase) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const v2u x = v2u{lo, hi};
  asm volatile(IPA_SGPR_HAZARD "global_store_dwordx2 %0, %1, %2 " IPA_STORE_FLAVOUR "\n\ts_nop 0" ::"v"(voff), "v"(x), "s"(sbase));
}
// cv2.remap's 16U bilinear sum (sampler.hpp::sample_u16_cv, operation for operation): the 2-D weight of a tap is the
// float32 PRODUCT wy * wx, the four products are summed left to right, every product and sum rounded (no fma)
__device__ __forceinline__ float cv16_sum(float v00, float v01, float v10, float v11, float wx0, float wx1,
                                          float wy0, float wy1) {
#pragma clang fp contract(off)
  float sum = v00 * (wy0 * wx0);
  sum = sum + v01 * (wy0 * wx1);
  sum = sum + v10 * (wy1 * wx0);
  sum = sum + v11 * (wy1 * wx1);
  return sum;
}
// cv2.remap's 8U bilinear (sampler.hpp::sample_u8_fixed, integer for integer): 15-bit weights from the 1/32-px
// fractions, rounded shift; tx / ty are k / 32 exactly
__device__ __forceinline__ float fix8_sum(int v00, int v01, int v10, int v11, float tx, float ty) {
  const int fx = (int)(tx * 32.f), fy = (int)(ty * 32.f);
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  const int acc = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
  const int o = (acc + (1 << 14)) >> 15;
  return (float)(o < 0 ? 0 : (o > 255 ? 255 : o));
}
__device__ __forceinline__ void pipe_store1(unsigned x, unsigned voff, uint8_t* sbase) {
  asm volatile(IPA_SGPR_HAZARD "global_store_dword %0, %1, %2 " IPA_STORE_FLAVOUR "\n\ts_nop 0" ::"v"(voff), "v"(x), "s"(sbase));
}
// saturate_cast<ushort>: round half to even, clamp (NaN -> 0)
__device__ __forceinline__ unsigned cv16_round(float v) {
  float r = rintf(v);
  r = r > 0.f ? r : 0.f;
  r = r < 65535.f ? r : 65535.f;
  return (unsigned)r;
}
// the two dwords of a bilinear tap row at byte offset `off` of the frame (range-checked)
__device__ __forceinline__ void pipe_gather2(float& a, float& b, unsigned off, v4i rs) {
#ifdef IPA_DEBUG_NO_GATHER   // measurement only (WRONG results): taps from arithmetic
  a = __uint_as_float(off | 0x3f000000u);
  b = a;
  return;
#endif
#ifdef IPA_DEBUG_ONE_DWORD   // measurement only (WRONG results): one dword per tap row
  asm volatile(IPA_SGPR_HAZARD "buffer_load_dword %0, %1, %2, 0 offen" : "=v"(a) : "v"(off), "s"(rs));
  b = a;
  return;
#endif
  asm volatile(IPA_SGPR_HAZARD "buffer_load_dword %0, %2, %3, 0 offen\n\t"
               "buffer_load_dword %1, %2, %3, 0 offen offset:4"
               : "=v"(a), "=v"(b)
               : "v"(off), "s"(rs));
}
// one dword (uint16 frames: both taps of a bilinear tap row)
__device__ __forceinline__ void pipe_gather1(float& a, unsigned off, v4i rs) {
  asm volatile(IPA_SGPR_HAZARD "buffer_load_dword %0, %1, %2, 0 offen" : "=v"(a) : "v"(off), "s"(rs));
}
// two bytes (uint8 frames: both taps of a bilinear tap row, zero-extended; any byte offset)
__device__ __forceinline__ void pipe_gather1_b16(float& a, unsigned off, v4i rs) {
  asm volatile(IPA_SGPR_HAZARD "buffer_load_ushort %0, %1, %2, 0 offen" : "=v"(a) : "v"(off), "s"(rs));
}
__device__ __forceinline__ void pipe_gather1_b16_masked(float& a, unsigned off, v4i rs, unsigned long long m) {
  unsigned long long sv;
  asm volatile(IPA_SGPR_HAZARD "s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, %4\n\t"
               "buffer_load_ushort %0, %2, %3, 0 offen\n\t"
               "s_mov_b64 exec, %1"
               : "+v"(a), "=&s"(sv)
               : "v"(off), "s"(rs), "s"(m));
}
__device__ __forceinline__ void pipe_gather1_masked(float& a, unsigned off, v4i rs,
                                                    unsigned long long m) {
  unsigned long long sv;
  asm volatile(IPA_SGPR_HAZARD "s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, %4\n\t"
               "buffer_load_dword %0, %2, %3, 0 offen\n\t"
               "s_mov_b64 exec, %1"
               : "+v"(a), "=&s"(sv)
               : "v"(off), "s"(rs), "s"(m));
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
  asm volatile(IPA_SGPR_HAZARD "s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, %4\n\t"
               "buffer_load_dword %0, %2, %3, 0 offen\n\t"
               "s_mov_b64 exec, %1"
               : "+v"(a), "=&s"(sv)
               : "v"(off), "s"(rs), "s"(m));
  b = a;
  return;
#endif
  asm volatile(IPA_SGPR_HAZARD "s_mov_b64 %2, exec\n\t"
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
  v2f pair[K + 2];
  window_pairs<K>(xp, lane, lane4_opaque, pair);
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
// EDGE: a strip on the rim of the image - its columns are resolved through the filter's border
// mode per lane (c.uu / c.uh, -1 = constant border) and loaded as dwords, its rows per row on
// the scalar unit; interior strips load a row as one float4 per lane (+ the halo dword).
template <int K, bool HALO, bool EDGE>
__device__ __forceinline__ void wave_run_strip_pipe(const WaveParams& p, const LoadRowSrc& src,
                                                    const Weights<float, K * K>& wts, float* xp,
                                                    const Cols& c, int y0, int nrows, bool writer,
                                                    float* dst) {
  using G = wave_geom<K, HALO>;
  constexpr int P = IPA_PIPE_ROWS;
  // loads per row: the row (one float4, EDGE: four dwords) + (HALO) its 2 H halo pixels
  constexpr int OPS = (EDGE ? 4 : 1) + (HALO ? 1 : 0);
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  unsigned lane4_opaque = 4u * lane;
  asm volatile("" : "+v"(lane4_opaque));
  const unsigned voff = 16u * lane;
  const int yb = y0 - G::H;
  const float* rows = src.base + (EDGE ? 0 : (long)yb * src.pitch + c.xs);  // scalar
  float* outs = dst + ((long)y0 * p.dpitch + c.xs);                          // scalar: output row 0
  // HALO: lane j < 2H supplies the pixel H - j left of the strip / j - H right of it
  const unsigned hoff = EDGE ? 4u * (unsigned)(c.uh < 0 ? 0 : c.uh) : 4u * halo_pos<G::H>(lane);
  const unsigned long long hmask = (1ull << (2 * G::H)) - 1ull;
  auto row_of = [&](int t) -> int {   // resolved input row of strip row t (-1 = constant border)
    if constexpr (EDGE) return resolve_idx(yb + t, p.dh, p.cby);
    else return t;
  };
  float buf[P][4];   // EDGE: the row as four dwords
  v4f bufv[P];       // interior: the row as one float4 (only one of the two forms is live)
  float hb[P] = {};
  auto issue = [&](float (&b)[4], v4f& bv, float& h, int t) {
    const int rr = row_of(t < T ? t : T - 1);
    const float* r = rows + (long)(rr < 0 ? 0 : rr) * src.pitch;
    if constexpr (EDGE) {
#pragma unroll
      for (int k = 0; k < 4; k++) pipe_load1<0>(b[k], 4u * (unsigned)(c.uu[k] < 0 ? 0 : c.uu[k]), r);
      if constexpr (HALO) pipe_load1_masked<0>(h, hoff, r, hmask);
    } else {
      pipe_load4(bv, voff, r);
      if constexpr (HALO) pipe_load1_masked<0>(h, hoff, r - G::H, hmask);
    }
  };
  static_for<0, P>([&](auto U) { constexpr int u = decltype(U)::value; issue(buf[u], bufv[u], hb[u], u); });
  v2f acc[K][2];
  int tb = 0;
#pragma unroll 1
  do {
    static_for<0, P>([&](auto U) {
      constexpr int u = decltype(U)::value;
      const int t = tb + u;
      if (t < T) {
        // younger than the loads of row t: the loads of rows t+1 .. t+P-1, and the stores of
        // iterations t-P .. t-1 (iteration j stores when j >= K - 1)
        if (t >= P + K - 1) vm_wait<OPS * (P - 1) + P>();
        else vm_wait<OPS * (P - 1)>();
        float v[4];
        if constexpr (EDGE) {
#pragma unroll
          for (int k = 0; k < 4; k++) { vm_pin(buf[u][k]); v[k] = buf[u][k]; }
        } else {
          vm_pin(bufv[u]);
          v[0] = bufv[u].x; v[1] = bufv[u].y; v[2] = bufv[u].z; v[3] = bufv[u].w;
        }
        float hv = 0.f;
        if constexpr (HALO) { vm_pin(hb[u]); hv = hb[u]; }
        if constexpr (EDGE) {   // positions the constant border supplies
          const int rr = row_of(t);
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = (rr < 0 || c.uu[k] < 0) ? src.cval : v[k];
          hv = (rr < 0 || c.uh < 0) ? src.cval : hv;
        }
        *reinterpret_cast<v4f*>(xp + kRowPad + 4u * lane) = v4f{v[0], v[1], v[2], v[3]};
        if constexpr (HALO) {
          if (lane < 2u * G::H) xp[kRowPad - G::H + halo_pos<G::H>(lane)] = hv;
        }
        issue(buf[u], bufv[u], hb[u], t + P);
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
template <typename ST, typename Coord, int K> struct pipe_capable<SampleRowSrc<ST, kLinear, Coord>, K> {
  static constexpr bool value = SampleRowSrc<ST, kLinear, Coord>::template depth<K>::kPiped;
};
// kernels whose batches share footprint records through LDS (wave_run_strip_shared): bilinear
// sampling of float32 / uint16 frames from ANY coordinate source
template <typename Src, int K> struct shared_capable : std::false_type {};
template <typename ST, typename Coord, int K> struct shared_capable<SampleRowSrc<ST, kLinear, Coord>, K> {
  static constexpr bool value = SampleRowSrc<ST, kLinear, Coord>::template depth<K>::kShared;
};
// ... of which the FAST strips of launches that do NOT share map rows (single frames, batches
// that are no multiple of IPA_WPB) run on wave_run_strip_pipe (float32 frames only)
template <typename Src, int K> struct pipe_unshared : std::false_type {};
template <int K> struct pipe_unshared<LoadRowSrc, K> : std::true_type {};
template <typename Coord, int K> struct pipe_unshared<SampleRowSrc<float, kLinear, Coord>, K> {
  static constexpr bool value = pipe_capable<SampleRowSrc<float, kLinear, Coord>, K>::value;
};
// which kernels take the 256-px aligned strip geometry with a halo pass: the plain filter
// (one extra EXEC-masked load per row: 64 x 4K 5x5 0.97 -> 0.90 ms, 0.85 with 32-row strips).
// NOT the sampling source: there the halo costs a fifth sample per lane and row - six more
// vector-memory instructions, and an instruction costs the CU's vector-memory path ~4.6 clocks
// WHATEVER its EXEC mask is, EXEC = 0 included (tools/pipe_micro.hip) - 1.30 -> 1.42 ms on the
// unshared loop, 0.969 -> 1.015 ms on the shared-record loop (same bits; IPA_HALO_SAMPLE=1
// builds both).  Round 5 rebuilt the shared-record loop's halo as ONE gather per row for all
// halo pixels (quads of lanes, see wave_run_strip_shared) - 128 registers instead of 128 + 5
// spilled, same bits - and measured it as a flavour of its own on batches: 1.08 against 0.98 ms,
// 0.974 even with the halo work left out (the aligned geometry streams 15 - 20 % faster as a
// plain copy, but this loop's compute floor is too close under it; profiles/r05_micro.txt).
// The sampling kernels keep the overlapping 248-px strips.
#ifndef IPA_HALO_SAMPLE
#define IPA_HALO_SAMPLE 0
#endif
template <typename Src, int K, bool STREAM> struct geom_halo {
  static constexpr bool value = (IPA_PIPE != 0) && (IPA_HALO != 0) && !STREAM && K <= 9 &&
                                pipe_capable<Src, K>::value && (IPA_HALO_SAMPLE != 0 || !Src::kHasQ5);
};

// ---- footprints on the border of the source, constant border mode (the reference's: cv2.remap
// with BORDER_CONSTANT, camera/LensDistortion.py:323-326 - and with its getOptimalNewCameraMatrix
// (alpha = 1) call every undistorted picture has such a rim).  sample() redoes them tap by tap
// with dependent loads: a strip that has one in every row ran 25 - 50 % over the others (64 x 4K,
// alpha = 1: 1.42 ms against 0.94).  Their taps are in the registers already - the gathers are
// range-checked and never fault; a tap outside the source is just the wrong pixel - so: whoever
// forms the records of a row marks, for every footprint that is not interior, which of its 4
// taps lie inside the source (bits 8 + 4 k .. 11 + 4 k of the interior word; rows that have none
// pay nothing), and the blend replaces the others by the border value: sample()'s arithmetic in
// sample()'s order, no tap inside = the border value itself.
// The one footprint whose INSIDE tap the gathers cannot deliver: the left tap in column -1 of row 0 (ix0 = -1,
// iy0 = 0 or -1).  The byte offset of that tap row is -4; uint16 frames hold both taps of a row in ONE dword at
// that offset and the range check drops it whole, and float32 frames fetch the right tap - pixel (0, 0) - as
// `offen offset:4` from the same -4: the hardware range-checks the UNSIGNED sum without wrapping it, so the load
// returns 0 for a pixel that is there (found by tools/fuzz_paths.py seed 63 in round 6: one or two samples per
// frame where the map crosses the source's top-left corner).  Bit 31 sends such a lane through sample().
constexpr unsigned kBorderSlow = 1u << 31;
template <int NS, int QM, bool PACKED, typename C>
__device__ __forceinline__ unsigned border_tap_bits(const SrcView& s, const C (&sx)[NS],
                                                    const C (&sy)[NS], unsigned interior) {
  unsigned vb = 0;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    if (!((interior >> k) & 1u)) {
      const bool ok = ipa_abs(sx[k]) < (C)kCoordLimit && ipa_abs(sy[k]) < (C)kCoordLimit;
      int ix0, iy0;
      float t0, t1;
      axis_frac<kLinear, float, C, QM>(s, ok ? sx[k] : (C)0, ix0, t0);
      axis_frac<kLinear, float, C, QM>(s, ok ? sy[k] : (C)0, iy0, t1);
      const bool x0 = (unsigned)ix0 < (unsigned)s.w, x1 = (unsigned)(ix0 + 1) < (unsigned)s.w;
      const bool y0 = (unsigned)iy0 < (unsigned)s.h, y1 = (unsigned)(iy0 + 1) < (unsigned)s.h;
      const unsigned b = ((y0 && x0) ? 1u : 0u) | ((y0 && x1) ? 2u : 0u) | ((y1 && x0) ? 4u : 0u) |
                         ((y1 && x1) ? 8u : 0u);
      vb |= (ok ? b : 0u) << (8 + 4 * k);
#ifdef IPA_DEBUG_CORNER_AS_ROUND5   // (test builds only: the rules as they were, to see the tests fail)
      if constexpr (PACKED)
        if (ok && ix0 == -1 && (iy0 == 0 || iy0 == -1) && s.w > 0 && s.h > 0) vb |= kBorderSlow;
#else
      if (ok && ix0 == -1 && (iy0 == 0 || iy0 == -1) && s.w > 0 && s.h > 0) vb |= kBorderSlow;
      // PACKED, the opposite corner: the left tap in the LAST column of the LAST row - the dword that holds it
      // ends two bytes past the frame and is dropped whole (tools/fuzz_corners.py, round 6)
      if constexpr (PACKED)
        if (ok && ix0 == s.w - 1 && (iy0 == s.h - 1 || iy0 == s.h - 2) && s.w > 0 && s.h > 0) vb |= kBorderSlow;
#endif
    }
  }
  return vb;
}
__device__ __forceinline__ float border_blend(float v00, float v01, float v10, float v11, float tx,
                                              float ty, unsigned vb, float cval) {
  v00 = (vb & 1u) ? v00 : cval;
  v01 = (vb & 2u) ? v01 : cval;
  v10 = (vb & 4u) ? v10 : cval;
  v11 = (vb & 8u) ? v11 : cval;
  const float wx0 = 1.f - tx, wx1 = tx, wy0 = 1.f - ty, wy1 = ty;
  float r0 = wx0 * v00;
  r0 = ipa_fma(wx1, v01, r0);
  float o = wy0 * r0;
  float r1 = wx0 * v10;
  r1 = ipa_fma(wx1, v11, r1);
  o = ipa_fma(wy1, r1, o);
  return vb ? o : cval;
}

template <int K, int QM, bool HALO, typename Coord>
__device__ __forceinline__ void wave_run_strip_pipe(const WaveParams& p,
                                                    const SampleRowSrc<float, kLinear, Coord>& src,
                                                    const Weights<float, K * K>& wts, float* xp,
                                                    const Cols& c, int y0, int nrows, bool writer,
                                                    float* dst) {
  using G = wave_geom<K, HALO>;
  using C = typename Coord::coord_t;
  static_assert(sizeof(C) == 4, "float32 coordinate tables");
  // samples per lane and row: pixels lane + 64 k (k = 0..3) and, in the HALO geometry, the halo
  // pixel of lanes 0 .. 2H-1 as a fifth sample under an EXEC mask
  constexpr int NS = HALO ? 5 : 4;
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  unsigned lane4_opaque = 4u * lane;
  asm volatile("" : "+v"(lane4_opaque));
  // (4 lane doubles as the map rows' byte offset and as the opaque LDS window offset: every
  // per-lane constant costs a register, and this kernel sits at the 128-register step)
  const unsigned voff = 16u * lane, moff = lane4_opaque;
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
  // the halo pixel of lane j < 2H: column xs - H + halo_pos(j); its LDS slot; the lanes' mask
  const unsigned hcol = halo_pos<G::H>(lane);
#define hoff (4u * hcol)                             /* bytes from (map row - H floats) */
  const unsigned long long hmask = (1ull << (2 * G::H)) - 1ull;
  const unsigned allin = lane < 2u * G::H ? 0u : 0x10u;   // lanes without a halo pixel: "inside"

  // vector-memory operations per row (the counted waits below)
#ifdef IPA_DEBUG_NO_MAP
  constexpr int kMapOps = 0;
#else
  constexpr int kMapOps = 2 * NS;
#endif
#if defined(IPA_DEBUG_NO_GATHER)
  constexpr int kTapOps = 0;
#elif defined(IPA_DEBUG_ONE_DWORD)
  constexpr int kTapOps = 2 * NS;
#else
  constexpr int kTapOps = 4 * NS;
#endif

  // map row r (clamped to the strip) -> m[k] = x, m[NS + k] = y of sample k
  auto issue_map = [&](float (&m)[2 * NS], int r) {
#ifdef IPA_DEBUG_NO_MAP   // measurement only (WRONG results): coordinates from arithmetic, no map traffic
    static_for<0, NS>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      const float fx = (float)(c.xs + (int)lane + 64 * k), fy = (float)(yb + (r < T ? r : T - 1));
      m[k] = fx * 0.97f + 40.f + fy * 0.004f;
      m[NS + k] = fy * 0.97f + 30.f + fx * 0.002f;
    });
    return;
#endif
    const long o = (long)(r < T ? r : T - 1) * src.coord.pitch;  // scalar
    static_for<0, 4>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      pipe_load1<256 * k>(m[k], moff, mxr + o);
    });
    if constexpr (HALO) pipe_load1_masked<0>(m[4], hoff, mxr + o - G::H, hmask);
    static_for<0, 4>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      pipe_load1<256 * k>(m[NS + k], moff, myr + o);
    });
    if constexpr (HALO) pipe_load1_masked<0>(m[NS + 4], hoff, myr + o - G::H, hmask);
  };
  auto pin_map = [&](float (&m)[2 * NS]) {
#pragma unroll
    for (int k = 0; k < 2 * NS; k++) vm_pin(m[k]);
  };
  // footprints of a map row: fractions, byte offsets of the top-left taps, interior bits
  auto footprint = [&](const float (&m)[2 * NS], float (&tx)[NS], float (&ty)[NS],
                       unsigned (&off)[NS], unsigned& interior) {
    float sx[NS], sy[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) { sx[k] = m[k]; sy[k] = m[NS + k]; }
    int e[NS];
    batch_footprint_linear<NS, QM>(s, sx, sy, tx, ty, e, interior);
    if constexpr (HALO) interior |= allin;
    if (s.border == IPA_BORDER_CONSTANT && __builtin_amdgcn_ballot_w64(interior != (1u << NS) - 1u))
      interior |= border_tap_bits<NS, QM, false, float>(s, sx, sy, interior);
#pragma unroll
    for (int k = 0; k < NS; k++) off[k] = (unsigned)e[k] << 2;
  };
  // tap rows of sample k: full wave for the strip's pixels, the halo lanes for sample 4
  auto gather_row = [&](float (&g)[2 * NS], const unsigned (&off)[NS], unsigned add) {
#pragma unroll
    for (int k = 0; k < 4; k++) pipe_gather2(g[2 * k], g[2 * k + 1], off[k] + add, rs);
    if constexpr (HALO) pipe_gather2_masked(g[8], g[9], off[4] + add, rs, hmask);
  };
  auto pin_taps = [&](float (&g)[2 * NS]) {
#pragma unroll
    for (int k = 0; k < 2 * NS; k++) vm_pin(g[k]);
  };

  float m[2 * NS] = {};      // map row in flight / being consumed
  float ga[2 * NS] = {}, gb[2 * NS] = {};   // tap rows: [2k], [2k+1] = the two dwords of footprint k
  float txa[NS], tya[NS], txb[NS], tyb[NS];
  unsigned offa[NS], offb[NS], ina, inb;
  v2f acc[K][2];

  // prologue: rows 0 and 1 resolved, gathers of row 0 and map row 2 in flight
  issue_map(m, 0);
  vm_wait<0>();
  pin_map(m);
  footprint(m, txa, tya, offa, ina);
  issue_map(m, 1);
  gather_row(ga, offa, 0u);
  gather_row(gb, offa, pitch_b);
  vm_wait<kTapOps>();
  pin_map(m);
  footprint(m, txb, tyb, offb, inb);
  issue_map(m, 2);
  // from here on, in issue order: ... gathers(t) [kTapOps], store(t-1)?, map(t+2) [kMapOps] | iteration t

  // one iteration: TOP / BOT = tap-row registers of row t (top, bottom); the bottom registers
  // become the top registers of row t + 1
  auto step = [&](int t, float (&top)[2 * NS], float (&bot)[2 * NS], const float (&tx)[NS],
                  const float (&ty)[NS], const unsigned (&off)[NS], unsigned interior,
                  float (&txn)[NS], float (&tyn)[NS], unsigned (&offn)[NS], unsigned& interiorn,
                  float (&txnn)[NS], float (&tynn)[NS], unsigned (&offnn)[NS], unsigned& interiornn) {
    // 1. the gathers of row t: younger = [store of iteration t-1] + map(t+2)
    // (the branch holds operand-less waits only: with the registers as operands of two
    // alternative statements the compiler merges them through copies, and a copy of a register
    // whose load is still in flight reads garbage)
    if (t >= K) vm_wait<kMapOps + 1>();
    else vm_wait<kMapOps>();
    pin_taps(top);
    pin_taps(bot);
    // 2. blend (the arithmetic and order of batch_blend_one) -> LDS row, natural pixel order
    float cur[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
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
    if constexpr (HALO) {
      if (lane < 2u * G::H) xp[kRowPad - G::H + hcol] = cur[4];
    }
    if (__builtin_amdgcn_ballot_w64(interior != (1u << NS) - 1u)) {
      if (s.border == IPA_BORDER_CONSTANT) {
        // footprints touching the source border, constant border mode: the taps are here, the
        // ones outside the source replaced by the border value (border_tap_bits / border_blend)
#pragma unroll
        for (int k = 0; k < NS; k++) {
          if (!((interior >> k) & 1u) && !(interior & kBorderSlow))
            xp[k < 4 ? kRowPad + 64u * k + lane : kRowPad - G::H + hcol] =
                border_blend(top[2 * k], top[2 * k + 1], bot[2 * k], bot[2 * k + 1], tx[k], ty[k],
                             (interior >> (8 + 4 * k)) & 15u, src.cval);
        }
      }
      if (__builtin_amdgcn_ballot_w64(s.border != IPA_BORDER_CONSTANT || (interior & kBorderSlow) != 0u)) {
        // the other border modes, and the lanes border_tap_bits left to it (the source's top-left
        // corner; rare): redo them tap by tap, straight into the LDS row - ONE copy of the
        // border-aware sampler per step (a loop, not unrolled)
#pragma unroll 1
        for (int k = 0; k < NS; k++) {
          if (!((interior >> k) & 1u) && (s.border != IPA_BORDER_CONSTANT || (interior & kBorderSlow))) {
            const int col = k < 4 ? c.xs + (int)lane + 64 * k : c.xs - G::H + (int)hcol;
            float sx, sy;
            // (t < T: the loop runs a step or two past the strip - their records are row T - 1's, and the map row
            //  past a strip that ends with the frame does not exist: found as a page fault in round 6)
            src.coord.get(col, yb + (t < T ? t : T - 1), sx, sy);
            xp[k < 4 ? kRowPad + 64u * k + lane : kRowPad - G::H + hcol] =
                sample<float, kLinear, float>(s, sx, sy, src.cval);
          }
        }
      }
    }
    // 3. gathers of row t + 1: its top row into `bot` under the mask of the lanes whose
    //    footprint did not move straight down, its bottom row into `top`
#pragma unroll
    for (int k = 0; k < NS; k++) {
#if IPA_PIPE_REUSE
      unsigned long long need = __builtin_amdgcn_ballot_w64(offn[k] != off[k] + pitch_b);
#else
      unsigned long long need = ~0ull;
#endif
      if (k == 4) need &= hmask;
      pipe_gather2_masked(bot[2 * k], bot[2 * k + 1], offn[k], rs, need);
    }
    gather_row(top, offn, pitch_b);
    __builtin_amdgcn_wave_barrier();
    // 4. filter + store
#ifdef IPA_DEBUG_NO_FILTER   // measurement only (WRONG results): the sample row goes straight out
    const v4f q = *reinterpret_cast<const v4f*>(xp + kRowPad + 4u * lane);
#else
    const v4f q = pipe_filter_row<K>(wts, xp, lane, lane4_opaque, acc);
#endif
    const int o = t - (K - 1);
    if (o >= 0) {
      if (writer) pipe_store4<true>(q, voff, outs + (long)o * p.dpitch);
    }
    __builtin_amdgcn_wave_barrier();
    // 5. map row t + 2: younger = gathers(t+1) [kTapOps] + this iteration's store
    if (t >= K - 1) vm_wait<kTapOps + 1>();
    else vm_wait<kTapOps>();
    pin_map(m);
    footprint(m, txnn, tynn, offnn, interiornn);
    issue_map(m, t + 3);
  };

  // rows rotate through three footprint sets (t, t+1, t+2) and two tap-register roles
  // (a do-while: T >= K, and with no path around the loop the prologue's loads provably flow
  // into it - tools/check_pipe_asm.py follows the control-flow graph)
  float txc[NS], tyc[NS];
  unsigned offc[NS], inc = (1u << NS) - 1u;
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
#undef hoff
}

// ------------------------------------------------------------------ sampling source, footprints
// shared by the frames of a workgroup --
// WaveParams::frames_wg: the IPA_WPB waves of a workgroup work on IPA_WPB frames of ONE strip, so
// their samples have the SAME footprints.  A vector-memory instruction costs the CU's
// vector-memory path ~4.6 clocks whatever it fetches (tools/pipe_micro.hip), and the 8 map
// dwords per row and wave were a third of the kernel's vector-memory instructions (13 % of its
// time: -DIPA_DEBUG_NO_MAP); the homography's double coordinates a third of the C3 chain.  Here
// the footprints of a row - byte offset of the top-left tap, the two fractions, the "inside the
// source" bits - are worked out ONCE per workgroup, row r by wave r mod IPA_WPB (from 8 map
// dwords, or from the lens model / homography in double), and handed to the other waves as a
// RECORD row in a ring of 2 IPA_WPB rows in LDS:
//
//   rows are grouped in blocks of W = IPA_WPB; between the barriers of block b-1 and block b
//   (one s_barrier per W rows, at the top of the block's last step) every wave READS the rows
//   of block b and the producers WRITE the rows of block b + 1 into the other half of the ring;
//   a producer of a table source issues its row's 8 loads right behind the barrier and turns
//   them into the record W - 1 iterations later.
//
// With the map loads issued BEFORE the row's gathers, the only operation younger than the
// gathers a wave waits for is its latest store: every wait is vmcnt(1) (0 before the first
// store), whatever was issued - so a top-row gather whose EXEC mask is empty is branched over
// instead of issued (43 % of them on the 4K lens map).
// The filter is a policy: dense K x K (DenseFilter) or separable K + K (SepFilter, wave_sep.hpp).
// Same footprints, same words, same blend as wave_run_strip_pipe / the chunked loop: identical bits.
// floats per record row: offsets, tx, ty (256 each), inside bits (64); HALO geometry: + one 16-byte
// entry per HALO LANE (see wave_run_strip_shared: 4 lanes per halo pixel, at most 8 halo pixels)
template <bool HALO> struct ring_row { static constexpr int value = HALO ? 832 + 128 : 832; };

template <int K> struct DenseFilter {
  static constexpr int kTaps = K;
  const Weights<float, K * K>& wts;
  unsigned lane4_opaque;
  v2f acc[K][2];
  __device__ __forceinline__ DenseFilter(const Weights<float, K * K>& w) : wts(w) {
    lane4_opaque = 4u * (threadIdx.x & 63u);
    asm volatile("" : "+v"(lane4_opaque));
  }
  // sample row (LDS, natural pixel order) -> the output row that completes with it
  template <bool EDGE> __device__ __forceinline__ v4f row(const float* xp, unsigned lane, const Cols&) {
    return pipe_filter_row<K>(wts, xp, lane, lane4_opaque, acc);
  }
};

// CV16 (round 6, the strip remap of uint16 frames INTO uint16 - what LensDistortion.correct returns for camera
// frames, camera/LensDistortion.py:323-326): K = 1, the blend is cv2's 16U arithmetic (cv16_sum; the slow path is
// sample_u16_cv itself), `dst` is a uint16 image and a lane's four results leave as two dwords.
// uint8 frames (ST = uint8_t, the same CV16 flag): into uint8 with cv2's 8U fixed-point arithmetic (fix8_sum /
// sample_u8_fixed); a tap row is one 16-bit load, a lane's four results leave as one dword.
template <int K, int QM, bool EDGE, bool HALO, typename Filter, typename ST, typename Coord, bool CV16 = false>
__device__ __forceinline__ void wave_run_strip_shared(const WaveParams& p,
                                                      const SampleRowSrc<ST, kLinear, Coord>& src,
                                                      Filter& filt, float* xp, float* ring,
                                                      unsigned wave, const Cols& c, int y0,
                                                      int nrows, bool writer, float* dst) {
  using G = wave_geom<K, HALO>;
  using C = typename Coord::coord_t;
  // samples per lane and row: pixels lane + 64 k (k = 0..3).
  // HALO geometry (round 5: 256-px ALIGNED strips, all 64 lanes store whole 128-byte lines - the
  // 248-px step of the overlapping strips costs a plain copy 16 %, tools/pipe_micro.hip G1 / G0):
  // the HP = 2 H halo pixels of a row (H left of the strip, H right of it) are sampled by QUADS of
  // lanes: lane 4 j + tap holds tap `tap` (0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right)
  // of halo pixel j - ONE gather instruction and ONE register per row for all of them; the blend
  // picks the four taps of a quad with DPP quad_perm broadcasts (sample()'s arithmetic in its
  // order: identical bits) and lane 4 j writes the sample into the pad of the LDS row.  The
  // producer of a record row forms the halo footprints as a fifth sample of its lanes 0 .. HP-1
  // and publishes them per halo lane: {byte offset of the lane's tap, tx, ty, flags}.  (The
  // first form of this geometry - the halo pixel as a fifth sample of lanes 0 .. HP-1 with its
  // own tap rows, fractions and need masks - took 128 VGPRs + 5 spilled and measured 5 % slower
  // than the overlapping strips, profiles/r03_micro.txt.)
  constexpr int NS = 4;
  constexpr int NSP = HALO ? 5 : 4;           // producer: footprints per lane
  constexpr int HP = 2 * G::H;                // halo pixels per row
  static_assert(!HALO || HP <= 8, "at most 8 halo pixels (32 halo lanes, 128 record floats)");
  constexpr int RR = ring_row<HALO>::value;   // floats per record row
  constexpr bool kTable = coord_is_table<Coord>::value;   // coordinates from a float32 table
  // registers per tap row of a footprint: float32 frames two dwords, uint16 frames ONE dword that
  // holds both taps (any byte offset, as TapLoad<uint16_t, float> loads it)
  constexpr int NR = sizeof(ST) == 4 ? 2 : 1;
  constexpr int SH = sizeof(ST) == 4 ? 2 : (sizeof(ST) == 2 ? 1 : 0);   // log2 of the element size
  constexpr bool kU8 = sizeof(ST) == 1;         // uint8 frames: a tap row is one 16-bit load (both taps, zero-extended)
  constexpr bool FIX8 = kU8 && CV16;            // ... into uint8: cv2's integer arithmetic (into float32: the float blend)
  static_assert(std::is_same<ST, float>::value || std::is_same<ST, uint16_t>::value || std::is_same<ST, uint8_t>::value,
                "float32 / uint16 / uint8 frames");
  constexpr int W = IPA_WPB;        // rows per block = waves per workgroup
  // lane order of the samples: interleaved (sample k of lane L = strip pixel L + 64 k: the 64
  // gathers of an instruction walk along the source row) or natural (pixel 4 L + k; measurement
  // only: 55 % slower - every gather then touches the 8 - 9 lines of the whole row segment)
  constexpr bool NAT = IPA_LANE_NATURAL != 0 && !HALO;
  constexpr int R = 2 * W;          // ring rows
  constexpr int kMapOps = kTable ? 2 * NSP : 0;
  static_assert(W == 2 || W == 4 || W == 8, "steps of a block alternate the tap-register roles");
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  // (tables of float32 coordinates - maps - or of the 8-byte ones stored_coords.hpp keeps)
  constexpr unsigned CB = sizeof(C);
  const unsigned voff = 16u * lane, moff = NAT ? 4u * CB * lane : CB * lane;
  auto ucol = [&](int k) -> int { return NAT ? c.uu[k] : c.uq[k]; };            // EDGE: resolved column of sample k
  auto pxcol = [&](int k) -> int { return NAT ? 4 * (int)lane + k : (int)lane + 64 * k; };   // strip pixel of sample k
  static_assert(!CV16 || (K == 1 && QM == 1 && !HALO && sizeof(ST) <= 2), "CV16: integer frames, 1/32-px coordinates, no filter");
  float* outs = dst + ((long)y0 * p.dpitch + c.xs);  // scalar: output row 0
  uint16_t* outs16 = reinterpret_cast<uint16_t*>(dst) + ((long)y0 * p.dpitch + c.xs);   // (CV16)
  uint8_t* outs8 = reinterpret_cast<uint8_t*>(dst) + ((long)y0 * p.dpitch + c.xs);      // (CV16, uint8 frames)
  const int yb = y0 - G::H;                          // first input row of the strip
  // EDGE: a strip on the rim of the filter domain - its columns are resolved through the
  // filter's border mode per lane (c.uq, -1 = constant border), its rows per row on the scalar
  // unit; interior strips address the map rows as base + 4 lane + 256 k
  const C* mxr = nullptr;
  const C* myr = nullptr;
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
  const unsigned pitch_b = (unsigned)s.pitch << SH;
  // a lane's four values of a record array are one aligned 16-byte word (conflict-free b128)
  float* rlane = ring + 4u * lane;
  // HALO, producer side: the halo pixel of lane j < HP is column xs - H + halo_pos(j)
  const unsigned hcol = halo_pos<G::H>(lane);
  const unsigned long long pmask = (1ull << HP) - 1ull;           // the producer's halo lanes
  const unsigned long long qmask = (1ull << (4 * HP)) - 1ull;     // the halo lanes of a row (quads)
  // HALO, consumer side: this lane's halo pixel (lanes past the quads mirror the last one: their
  // values are never stored) and its place in the LDS row
  const unsigned hq = lane < 4u * HP ? lane >> 2 : (unsigned)HP - 1u;
  const unsigned hpos = kRowPad - G::H + halo_pos<G::H>(hq);
  // ... EDGE: its resolved column (-1: the filter's constant border supplies the pixel)
  int uhq = 0;
  if constexpr (HALO && EDGE) uhq = resolve_idx(c.xs - G::H + (int)halo_pos<G::H>(hq), p.dw, p.cbx);
  // halo flags (record word 3)
  constexpr unsigned kHTapIn = 1u, kHAnyIn = 2u, kHSlow = 4u, kHInterior = 8u;

  // ---- producer: this wave's row of a block.  Table sources: the 8 map dwords (clamped to the
  // strip) into pm; the record is formed when they have arrived.
  C pm[2 * NSP] = {};
  auto issue_coords = [&](int r) {
    if constexpr (kTable) {
      if constexpr (EDGE) {
        const int rr = row_of(r < T ? r : T - 1);
        const long o = (long)(rr < 0 ? 0 : rr) * src.coord.pitch;  // scalar
#pragma unroll
        for (int k = 0; k < 4; k++) pipe_load1<0>(pm[k], CB * (unsigned)(ucol(k) < 0 ? 0 : ucol(k)), mxr + o);
        if constexpr (HALO) pipe_load1_masked<0>(pm[4], CB * (unsigned)(c.uh < 0 ? 0 : c.uh), mxr + o, pmask);
#pragma unroll
        for (int k = 0; k < 4; k++) pipe_load1<0>(pm[NSP + k], CB * (unsigned)(ucol(k) < 0 ? 0 : ucol(k)), myr + o);
        if constexpr (HALO) pipe_load1_masked<0>(pm[NSP + 4], CB * (unsigned)(c.uh < 0 ? 0 : c.uh), myr + o, pmask);
      } else {
        const long o = (long)(r < T ? r : T - 1) * src.coord.pitch;  // scalar
        static_for<0, 4>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          pipe_load1<(NAT ? 1 : 64) * (int)CB * k>(pm[k], moff, mxr + o);
        });
        if constexpr (HALO) pipe_load1_masked<0>(pm[4], CB * hcol, mxr + o - G::H, pmask);
        static_for<0, 4>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          pipe_load1<(NAT ? 1 : 64) * (int)CB * k>(pm[NSP + k], moff, myr + o);
        });
        if constexpr (HALO) pipe_load1_masked<0>(pm[NSP + 4], CB * hcol, myr + o - G::H, pmask);
      }
    }
  };
  auto publish = [&](int r) {   // (table sources: after a wait that covers pm)
    C sx[NSP], sy[NSP];
    if constexpr (kTable) {
#pragma unroll
      for (int k = 0; k < 2 * NSP; k++) vm_pin(pm[k]);
#pragma unroll
      for (int k = 0; k < NSP; k++) { sx[k] = pm[k]; sy[k] = pm[NSP + k]; }
    } else {
      const int rc = r < T ? r : T - 1;
      const int rr = row_of(rc);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if constexpr (EDGE) src.coord.get(ucol(k) < 0 ? 0 : ucol(k), rr < 0 ? 0 : rr, sx[k], sy[k]);
        else src.coord.get(c.xs + pxcol(k), rr, sx[k], sy[k]);
      }
      if constexpr (HALO) {
        if constexpr (EDGE) src.coord.get(c.uh < 0 ? 0 : c.uh, rr < 0 ? 0 : rr, sx[4], sy[4]);
        else src.coord.get(c.xs - G::H + (int)(lane < (unsigned)HP ? hcol : (unsigned)G::H), rr, sx[4], sy[4]);
      }
    }
    float tx[NSP], ty[NSP];
    int e[NSP];
    unsigned interior;
    batch_footprint_linear<NSP, QM>(s, sx, sy, tx, ty, e, interior);
    if constexpr (HALO) interior |= lane < (unsigned)HP ? 0u : 0x10u;   // lanes without a halo pixel
    if (s.border == IPA_BORDER_CONSTANT && __builtin_amdgcn_ballot_w64(interior != (1u << NSP) - 1u))
      interior |= border_tap_bits<NSP, QM, NR == 1, C>(s, sx, sy, interior);
    float* slot = rlane + (unsigned)(r % R) * RR;
    *reinterpret_cast<v4i*>(slot) = v4i{e[0] << SH, e[1] << SH, e[2] << SH, e[3] << SH};
    *reinterpret_cast<v4f*>(slot + 256) = v4f{tx[0], tx[1], tx[2], tx[3]};
    *reinterpret_cast<v4f*>(slot + 512) = v4f{ty[0], ty[1], ty[2], ty[3]};
    float* rowp = ring + (unsigned)(r % R) * RR;
    reinterpret_cast<unsigned*>(rowp + 768)[lane] = interior & ~0x0f000010u;   // (the halo's bits stay here)
    if constexpr (HALO) {
      if (lane < (unsigned)HP) {
        const bool in4 = (interior >> 4) & 1u;
        const unsigned vb = (interior >> 24) & 15u;   // taps of the halo footprint inside the source
        // what the blend cannot do from the taps: the other border modes, and (packed frames) the
        // one corner border_tap_bits leaves to sample()
        const bool slow = !in4 && (s.border != IPA_BORDER_CONSTANT || (interior & kBorderSlow) != 0u);
        float* hrow = rowp + 832 + 16u * lane;
#pragma unroll
        for (int tap = 0; tap < 4; tap++) {
          const unsigned o = ((unsigned)e[4] << SH) + (NR == 2 ? (unsigned)(tap & 1) * 4u : 0u) +
                             (unsigned)(tap >> 1) * pitch_b;
          const unsigned fl = ((in4 || ((vb >> tap) & 1u)) ? kHTapIn : 0u) | ((in4 || vb) ? kHAnyIn : 0u) |
                              (slow ? kHSlow : 0u) | (in4 ? kHInterior : 0u);
          *reinterpret_cast<v4f*>(hrow + 4 * tap) = v4f{__uint_as_float(o), tx[4], ty[4], __uint_as_float(fl)};
        }
      }
    }
  };
  // ---- consumer: the footprints of ring row r
  auto footprint = [&](int r, float (&tx)[NS], float (&ty)[NS], unsigned (&off)[NS],
                       unsigned& interior) {
    const float* slot = rlane + (unsigned)(r % R) * RR;
    const float* rowp = ring + (unsigned)(r % R) * RR;
    const v4i qo = *reinterpret_cast<const v4i*>(slot);
    const v4f qx = *reinterpret_cast<const v4f*>(slot + 256), qy = *reinterpret_cast<const v4f*>(slot + 512);
    interior = reinterpret_cast<const unsigned*>(rowp + 768)[lane];
    off[0] = (unsigned)qo.x; off[1] = (unsigned)qo.y; off[2] = (unsigned)qo.z; off[3] = (unsigned)qo.w;
    tx[0] = qx.x; tx[1] = qx.y; tx[2] = qx.z; tx[3] = qx.w;
    ty[0] = qy.x; ty[1] = qy.y; ty[2] = qy.z; ty[3] = qy.w;
  };
  // ... and the halo lane's entry of it: byte offset of its tap, the fractions, the flags
  auto halo_record = [&](int r, unsigned& off, float& tx, float& ty, unsigned& fl) {
    const float* rowp = ring + (unsigned)(r % R) * RR;
    const v4f q = *reinterpret_cast<const v4f*>(rowp + 832 + 4u * (lane < 4u * HP ? lane : 4u * HP - 1u));
    off = __float_as_uint(q.x); tx = q.y; ty = q.z; fl = __float_as_uint(q.w);
  };
  // the tap row of footprint k at byte offset o -> g[NR k .. NR k + NR - 1]
  auto gather = [&](float (&g)[NS * NR], int k, unsigned o) {
#if IPA_SHARE_TAPS == 9
    if constexpr (NR == 2) { pipe_gather1(g[2 * k], o, rs); return; }
#endif
    if constexpr (NR == 2) pipe_gather2(g[2 * k], g[2 * k + 1], o, rs);
    else if constexpr (kU8) pipe_gather1_b16(g[k], o, rs);
    else pipe_gather1(g[k], o, rs);
  };
  auto gather_masked = [&](float (&g)[NS * NR], int k, unsigned o, unsigned long long m) {
#if IPA_SHARE_TAPS == 9
    if constexpr (NR == 2) { pipe_gather1_masked(g[2 * k], o, rs, m); return; }
#endif
    if constexpr (NR == 2) pipe_gather2_masked(g[2 * k], g[2 * k + 1], o, rs, m);
    else if constexpr (kU8) pipe_gather1_b16_masked(g[k], o, rs, m);
    else pipe_gather1_masked(g[k], o, rs, m);
  };
  auto pin_taps = [&](float (&g)[NS * NR]) {
#pragma unroll
    for (int k = 0; k < NS * NR; k++)
      if (IPA_SHARE_TAPS != 9 || NR != 2 || (k & 1) == 0) vm_pin(g[k]);
  };
  // a full tap row of the strip's pixels
  auto gather_row = [&](float (&g)[NS * NR], const unsigned (&off)[NS], unsigned add) {
#pragma unroll
    for (int k = 0; k < 4; k++) gather(g, k, off[k] + add);
  };
  auto taps_of = [&](const float (&g)[NS * NR], int k, float& v0, float& v1) {
#if IPA_SHARE_TAPS == 9
    if constexpr (NR == 2) { v0 = g[2 * k]; v1 = from_lane_above(v0); return; }
#endif
    if constexpr (NR == 2) { v0 = g[2 * k]; v1 = g[2 * k + 1]; }
    else if constexpr (kU8) { v0 = (float)(__float_as_uint(g[k]) & 0xffu); v1 = (float)((__float_as_uint(g[k]) >> 8) & 0xffu); }
    else TapLoad<uint16_t, float>::unpack(__float_as_uint(g[k]), v0, v1);
  };

  float ga[NS * NR] = {}, gb[NS * NR] = {};    // tap rows of the footprints (see NR)
  float txa[NS], tya[NS], txb[NS], tyb[NS];
  unsigned offa[NS], offb[NS], ina, inb;
  // HALO: the halo lane's tap dword of the row in flight / being consumed, its fractions and flags
  float hga = 0.f, hgb = 0.f, htxa = 0.f, htya = 0.f, htxb = 0.f, htyb = 0.f;
  unsigned hfla = 0u, hflb = 0u;

#ifdef IPA_DEBUG_STAMPS   // measurement only (the first output row of a strip receives the sums)
  // cycles (s_memtime) a wave spends per phase of a step, summed over the strip:
  // 0 barrier, 1 wait for the gathers, 2 blend + LDS row (+ publish), 3 footprints + gather issue,
  // 4 filter, 5 store issue, 6 the whole loop, 7 steps
  unsigned stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned stamp_prev = 0;
#define IPA_STAMP0() stamp_prev = (unsigned)__builtin_amdgcn_s_memtime()
#define IPA_STAMP(i) { const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_prev; stamp_prev = now_; }
#else
#define IPA_STAMP0()
#define IPA_STAMP(i)
#endif
  // prologue: the records of block 0 into the ring (row `wave` by this wave), barrier, the
  // loads of block 1's row issued, row 0's gathers in flight
  issue_coords((int)wave);
  vm_wait<0>();
  publish((int)wave);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring rows are written
  issue_coords(W + (int)wave);
  footprint(0, txa, tya, offa, ina);
  gather_row(ga, offa, 0u);
  gather_row(gb, offa, pitch_b);
  if constexpr (HALO && IPA_DEBUG_HALO_LEVEL < 2) {
    unsigned ho;
    halo_record(0, ho, htxa, htya, hfla);
    if (IPA_DEBUG_HALO_LEVEL < 1) pipe_gather1_masked(hga, ho, rs, qmask);
  }

  // one iteration (row t, step STEP = t mod W of its block); TOP / BOT = tap-row registers of
  // row t; the bottom registers become the top registers of row t + 1
  auto step = [&](auto St, int t, float (&top)[NS * NR], float (&bot)[NS * NR], const float (&tx)[NS],
                  const float (&ty)[NS], const unsigned (&off)[NS], unsigned interior,
                  float (&txn)[NS], float (&tyn)[NS], unsigned (&offn)[NS], unsigned& interiorn,
                  float& hg, float htx, float hty, unsigned hfl,
                  float& hgn, float& htxn, float& htyn, unsigned& hfln) {
    constexpr int STEP = decltype(St)::value;
    IPA_STAMP0();
    if constexpr (STEP == W - 1) {
      // the block's barrier: behind it the records of the next block are in the ring and nobody
      // reads this block's half any more; the next row of this wave is requested at once
#ifndef IPA_DEBUG_NO_BARRIER   // (measurement only: racy without it)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      IPA_STAMP(0);
      issue_coords(t + 1 + W + (int)wave);
    }
    // 1. the gathers of row t (and everything older): the only younger operations are the
    //    store of iteration t-1 and, on the last step of a block, the map loads above
    // (rows past the strip - the last block is filled up - store nothing either)
    const bool stored = t >= K && t <= T;
    if constexpr (STEP == W - 1) {
      if (stored) vm_wait<kMapOps + 1>();
      else vm_wait<kMapOps>();
    } else {
      if (stored) vm_wait<1>();
      else vm_wait<0>();
    }
    pin_taps(top);
    pin_taps(bot);
    IPA_STAMP(1);
    if constexpr (HALO && IPA_DEBUG_HALO_LEVEL < 1) vm_pin(hg);
    if constexpr (STEP == W - 2) publish(t + 2 + (int)wave);  // (its loads: W - 1 iterations ago)
    // 2. blend (the arithmetic and order of batch_blend_one) -> LDS row, natural pixel order
    float cur[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const float wx0 = 1.f - tx[k], wx1 = tx[k], wy0 = 1.f - ty[k], wy1 = ty[k];
      float v00, v01, v10, v11;
      taps_of(top, k, v00, v01);
      taps_of(bot, k, v10, v11);
      if constexpr (FIX8) {
        cur[k] = fix8_sum((int)v00, (int)v01, (int)v10, (int)v11, tx[k], ty[k]);
      } else if constexpr (CV16) {
        cur[k] = cv16_sum(v00, v01, v10, v11, wx0, wx1, wy0, wy1);
      } else {
        float r0 = wx0 * v00;
        r0 = ipa_fma(wx1, v01, r0);
        float o = wy0 * r0;
        float r1 = wx0 * v10;
        r1 = ipa_fma(wx1, v11, r1);
        cur[k] = ipa_fma(wy1, r1, o);
      }
    }
    int rowt = 0;   // EDGE: resolved row of this strip row
    if constexpr (EDGE) {
      rowt = row_of(t < T ? t : T - 1);
      // positions the filter's constant border supplies
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = (rowt < 0 || ucol(k) < 0) ? src.ccval : cur[k];
    }
    if constexpr (NAT) {
      *reinterpret_cast<v4f*>(xp + kRowPad + 4u * lane) = v4f{cur[0], cur[1], cur[2], cur[3]};
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) xp[kRowPad + 64u * k + lane] = cur[k];
    }
    if constexpr (HALO && IPA_DEBUG_HALO_LEVEL < 2) {
      // the halo pixels: tap `lane & 3` of halo pixel `lane >> 2` is in hg; a tap outside the
      // source is the border value (constant border mode; border_blend's rule), the blend takes
      // the quad's four taps by DPP broadcasts - sample()'s arithmetic in its order
      float v = hg;
      if constexpr (NR == 1) {
        float lo, hi;
        TapLoad<uint16_t, float>::unpack(__float_as_uint(hg), lo, hi);
        v = (lane & 1u) ? hi : lo;
      }
      v = (hfl & kHTapIn) ? v : src.cval;
      const int vi = __float_as_int(v);
      const float v00 = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0x00, 0xf, 0xf, false));
      const float v01 = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0x55, 0xf, 0xf, false));
      const float v10 = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0xaa, 0xf, 0xf, false));
      const float v11 = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0xff, 0xf, 0xf, false));
      const float wx0 = 1.f - htx, wx1 = htx, wy0 = 1.f - hty, wy1 = hty;
      float r0 = wx0 * v00;
      r0 = ipa_fma(wx1, v01, r0);
      float o = wy0 * r0;
      float r1 = wx0 * v10;
      r1 = ipa_fma(wx1, v11, r1);
      o = ipa_fma(wy1, r1, o);
      o = (hfl & kHAnyIn) ? o : src.cval;
      if (__builtin_amdgcn_ballot_w64((hfl & kHSlow) != 0u && lane < 4u * HP)) {
        // the other border modes and the lanes border_tap_bits left to it (rare): tap by tap
        if (hfl & kHSlow) {
          C sx, sy;
          const int col = c.xs - G::H + (int)halo_pos<G::H>(hq);
          if constexpr (EDGE) src.coord.get(uhq < 0 ? 0 : uhq, rowt < 0 ? 0 : rowt, sx, sy);
          else src.coord.get(col, yb + (t < T ? t : T - 1), sx, sy);   // (see below)
          o = sample<ST, kLinear, C>(s, sx, sy, src.cval);
        }
      }
      if constexpr (EDGE) o = (rowt < 0 || uhq < 0) ? src.ccval : o;
      if ((lane & 3u) == 0u && lane < 4u * HP) xp[hpos] = o;
    }
    if (s.border == IPA_BORDER_CONSTANT) {
      if (__builtin_amdgcn_ballot_w64(interior != (1u << NS) - 1u)) {
        // footprints touching the source border, constant border mode: the taps are here, the
        // ones outside the source replaced by the border value (border_tap_bits / border_blend)
#pragma unroll
        for (int k = 0; k < NS; k++) {
          if (!((interior >> k) & 1u) && !(interior & kBorderSlow)) {
            float v00, v01, v10, v11;
            taps_of(top, k, v00, v01);
            taps_of(bot, k, v10, v11);
            float o;
            if constexpr (FIX8) {
              const unsigned vb = (interior >> (8 + 4 * k)) & 15u;
              const int cv = (int)src.cval;
              o = fix8_sum((vb & 1u) ? (int)v00 : cv, (vb & 2u) ? (int)v01 : cv, (vb & 4u) ? (int)v10 : cv,
                           (vb & 8u) ? (int)v11 : cv, tx[k], ty[k]);
              o = vb ? o : src.cval;
            } else if constexpr (CV16) {   // taps outside the source are the border value, none inside = the border value
              const unsigned vb = (interior >> (8 + 4 * k)) & 15u;
              o = cv16_sum((vb & 1u) ? v00 : src.cval, (vb & 2u) ? v01 : src.cval, (vb & 4u) ? v10 : src.cval,
                           (vb & 8u) ? v11 : src.cval, 1.f - tx[k], tx[k], 1.f - ty[k], ty[k]);
              o = vb ? o : src.cval;
            } else {
              o = border_blend(v00, v01, v10, v11, tx[k], ty[k], (interior >> (8 + 4 * k)) & 15u, src.cval);
            }
            bool keep = true;   // EDGE: positions the filter's constant border supplies stay
            if constexpr (EDGE) keep = !(rowt < 0 || ucol(k) < 0);
            if (keep) xp[kRowPad + (unsigned)pxcol(k)] = o;
          }
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(s.border == IPA_BORDER_CONSTANT ? (interior & kBorderSlow) != 0u
                                                                     : interior != (1u << NS) - 1u)) {
      // the other border modes, and the lanes border_tap_bits left to it (rare): redo them tap by
      // tap, straight into the LDS row - ONE copy of the border-aware sampler per step (a loop,
      // not unrolled)
#pragma unroll 1
      for (int k = 0; k < NS; k++) {
        if (!((interior >> k) & 1u) && (s.border != IPA_BORDER_CONSTANT || (interior & kBorderSlow))) {
          C sx, sy;
          // pixel column of sample k and its place in the LDS row
          const int col = c.xs + pxcol(k);
          const unsigned pos = kRowPad + (unsigned)pxcol(k);
          if constexpr (EDGE) {
            // (the column is resolved again: c.uq[k] with a run-time k would live in scratch)
            const int uqk = resolve_idx(col, p.dw, p.cbx);
            src.coord.get(uqk < 0 ? 0 : uqk, rowt < 0 ? 0 : rowt, sx, sy);
            if (!(rowt < 0 || uqk < 0)) {
              if constexpr (FIX8) xp[pos] = (float)sample_u8_fixed<C>(s, sx, sy, (uint8_t)src.cval);
              else if constexpr (CV16) xp[pos] = (float)sample_u16_cv<kLinear, C>(s, nullptr, sx, sy, (uint16_t)src.cval);
              else xp[pos] = sample<ST, kLinear, C>(s, sx, sy, src.cval);
            }
          } else {
            // (t < T: the last block is filled up to a whole number of steps with the records of row T - 1; the
            //  coordinate row itself must be clamped too - a table source has no row past a strip that ends with
            //  the frame: a page fault on a map that ended with its allocation, round 6)
            src.coord.get(col, yb + (t < T ? t : T - 1), sx, sy);
            if constexpr (FIX8) xp[pos] = (float)sample_u8_fixed<C>(s, sx, sy, (uint8_t)src.cval);
            else if constexpr (CV16) xp[pos] = (float)sample_u16_cv<kLinear, C>(s, nullptr, sx, sy, (uint16_t)src.cval);
            else xp[pos] = sample<ST, kLinear, C>(s, sx, sy, src.cval);
          }
        }
      }
    }
    IPA_STAMP(2);
    // 3. row t + 1: footprints from the ring, its top tap row into `bot` for the lanes whose
    //    footprint did not move straight down (no instruction at all when there is none), its
    //    bottom row into `top`
#ifdef IPA_DEBUG_NO_FOOTPRINT   // measurement only (WRONG results): the footprint moves straight down
#pragma unroll
    for (int k = 0; k < NS; k++) { txn[k] = tx[k]; tyn[k] = ty[k]; offn[k] = off[k] + pitch_b; }
    interiorn = (1u << NS) - 1u;
#else
    footprint(t + 1 < T ? t + 1 : T - 1, txn, tyn, offn, interiorn);
#endif
#pragma unroll
    for (int k = 0; k < NS; k++) {
#if IPA_PIPE_REUSE
      // three cases per lane: the footprint moved straight down (its top row = this row's
      // bottom row, already in `bot`); it did NOT move (vertical scale < 1: both its tap rows
      // are this row's - the top one is copied over from `top` before the gathers below
      // overwrite it); anything else: gathered under the lanes' EXEC mask
      const bool stay = IPA_PIPE_STAY && offn[k] == off[k];
      if (__builtin_amdgcn_ballot_w64(stay)) {
#pragma unroll
        for (int j = 0; j < NR; j++) bot[NR * k + j] = stay ? top[NR * k + j] : bot[NR * k + j];
      }
      unsigned long long need =
          __builtin_amdgcn_ballot_w64(offn[k] != off[k] + pitch_b && !stay);
      if (need) gather_masked(bot, k, offn[k], need);
#else
      gather(bot, k, offn[k]);
#endif
    }
    gather_row(top, offn, pitch_b);
    if constexpr (HALO && IPA_DEBUG_HALO_LEVEL < 2) {
      unsigned ho;
      halo_record(t + 1 < T ? t + 1 : T - 1, ho, htxn, htyn, hfln);
      if (IPA_DEBUG_HALO_LEVEL < 1) pipe_gather1_masked(hgn, ho, rs, qmask);
      else hgn = __uint_as_float(ho | 0x3f000000u);
    }
    __builtin_amdgcn_wave_barrier();
    IPA_STAMP(3);
    // 4. filter + store
#ifdef IPA_DEBUG_NO_FILTER   // measurement only (WRONG results): the sample row goes straight out
    const v4f q = *reinterpret_cast<const v4f*>(xp + kRowPad + 4u * lane);
#else
    const v4f q = filt.template row<EDGE>(xp, lane, c);
#endif
    const int o = t - (K - 1);
#ifdef IPA_DEBUG_STAMPS
    { float qx = q.x; asm volatile("" : "+v"(qx)); }   // (the filter's result is due here)
#endif
    IPA_STAMP(4);
    if (o >= 0 && o < nrows) {
      if constexpr (FIX8) {   // (the samples are whole numbers 0 .. 255 already)
        const unsigned px = (unsigned)q.x | ((unsigned)q.y << 8) | ((unsigned)q.z << 16) | ((unsigned)q.w << 24);
        if (writer) pipe_store1(px, 4u * lane, outs8 + (long)o * p.dpitch);
      } else if constexpr (CV16) {
        const unsigned lo = cv16_round(q.x) | (cv16_round(q.y) << 16), hi = cv16_round(q.z) | (cv16_round(q.w) << 16);
        if (writer) pipe_store2(lo, hi, 8u * lane, outs16 + (long)o * p.dpitch);
      } else {
        if (writer) pipe_store4<true>(q, voff, outs + (long)o * p.dpitch);
      }
    }
    __builtin_amdgcn_wave_barrier();
    IPA_STAMP(5);
#ifdef IPA_DEBUG_STAMPS
    stamp_acc[7] += 1;
#endif
  };

  // the loop body is one block (W steps: the tap registers swap roles every step, W is even);
  // every wave of the workgroup runs the same T, so the barriers match
  const int Tb = (T + W - 1) / W * W;
  int tb = 0;
#pragma unroll 1
  do {
    static_for<0, W>([&](auto St) {
      constexpr int st = decltype(St)::value;
      if constexpr (st % 2 == 0)
        step(St, tb + st, ga, gb, txa, tya, offa, ina, txb, tyb, offb, inb, hga, htxa, htya, hfla, hgb, htxb, htyb, hflb);
      else
        step(St, tb + st, gb, ga, txb, tyb, offb, inb, txa, tya, offa, ina, hgb, htxb, htyb, hflb, hga, htxa, htya, hfla);
    });
    tb += W;
  } while (tb < Tb);
#ifdef IPA_DEBUG_STAMPS
  stamp_acc[6] = stamp_acc[0] + stamp_acc[1] + stamp_acc[2] + stamp_acc[3] + stamp_acc[4] + stamp_acc[5];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!EDGE) {
    float* o0 = outs + 4u * lane;
    if (lane == 1) *reinterpret_cast<v4f*>(o0) = v4f{(float)stamp_acc[0], (float)stamp_acc[1], (float)stamp_acc[2], (float)stamp_acc[3]};
    if (lane == 2) *reinterpret_cast<v4f*>(o0) = v4f{(float)stamp_acc[4], (float)stamp_acc[5], (float)stamp_acc[6], (float)stamp_acc[7]};
  }
#endif
#undef IPA_STAMP0
#undef IPA_STAMP
}

}  // namespace ipa
