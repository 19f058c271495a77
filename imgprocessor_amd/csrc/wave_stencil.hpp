// wave_stencil.hpp — the K x K stencil skeleton tuned for CDNA4: a wave64
// marches down a column strip; no workgroup barrier, LDS only wave-private.
//
//   * one wave owns a strip 256 px wide (64 lanes x 4 px: every global access
//     is a coalesced 16-byte-per-lane vector) and `strip_h` output rows tall;
//     everything derived from the strip id is wave-uniform and lives in SGPRs
//     (the wave index is read with readfirstlane), rows are addressed as a
//     scalar base + a 32-bit lane offset;
//   * it walks the strip's strip_h + K - 1 input rows top to bottom in CHUNKS
//     of D rows.  All memory traffic of a chunk is issued up front in straight-
//     line code (D row loads; for the fused kernels D map-row loads, then the
//     4 x D tap gathers), only then are the rows consumed one by one — so the
//     waitcnt scoreboard sees counted waits; 16-20 waves per CU cover the
//     latency without any workgroup-level staging phase;
//   * every arriving row goes through a wave-private LDS row (256 px + pad).
//     Each lane reads the K+2 overlapping PAIRS of its 4 + 2H pixel window
//     from it: the horizontal halo needs no cross-lane exchange, and one
//     v_pk_fma_f32 per coefficient advances two output pixels.  The wave's own
//     ds_write / ds_read execute in order: no barrier;
//   * vertical reuse is in registers: an arriving input row is scattered into
//     the K output rows it contributes to (K x 2 running pair sums per lane,
//     shifted by one row per step inside the fma chain itself); the oldest row
//     is complete after each step and is stored as one float4.
//     Strips overlap by 8 px instead of loading ragged halos: lanes 0 and 63
//     only supply neighbours.
//
// Input rows are produced by a row source: plain image rows (the filters/
// convolution) or remapped rows sampled on the fly (undistort / perspective
// warp fused with the filter: the intermediate image never exists in HBM).
// Interior strips of the sampling source take their footprints in lane-
// interleaved order (footprint k of lane L = strip pixel L + 64 k), so the 64
// gathers of one instruction walk along the source row; the LDS row puts the
// samples back into pixel order for free.
//
// Two code paths per kernel, chosen per strip (wave-uniform):
//   FAST  every column and row the strip touches lies inside the image and
//         vector alignment holds: unconditional 16-byte accesses; the remap's
//         coordinate rule (exact / 1/32-px) is a compile-time parameter here;
//   rim   strips touching the image border: columns / rows are resolved
//         through the FILTER's border mode (per lane once, per row on the
//         scalar unit), element accesses with selects instead of branches.  The
//         sampling source keeps its lane-interleaved order here too (resolved
//         columns Cols::uq): 15 % of a 4K frame's strips are rim strips.
//
// Dispatch order: the frames of one strip block are neighbours in the
// XCD-contiguous block order (wave_grid / frames_inner), so a batch's frames
// march through the same map rows together and fetch them into L2 once.
//
// Summation order per output pixel is identical to conv_tile.hpp: kernel rows
// i = 0..K-1, taps j = 0..K-1, one float fma chain -> bit-identical results.
//
// Reference semantics: filters/maskedConvolve.py:24-43 + scipy.ndimage.correlate
// (filter), camera/LensDistortion.py:323-326 and
// camera/PerspectiveCorrection.py:401-405 followed by a K x K filter (fused).
#pragma once

#include <type_traits>

#include "common.hpp"
#include "conv_tile.hpp"
#include "sampler.hpp"

#ifndef IPA_PIPE
#define IPA_PIPE 1   // FAST strips on the hand-scheduled memory pipeline of wave_pipe.hpp
#endif
#ifndef IPA_PIPE_MAX_K
#define IPA_PIPE_MAX_K 7   // largest K of the hand-scheduled sampling kernels (resident coefficients)
#endif
#ifndef IPA_PIPE_MIN_WAVES
#define IPA_PIPE_MIN_WAVES 4
#endif
#ifndef IPA_WPB
#define IPA_WPB 4   // waves per workgroup
#endif
#ifndef IPA_PIPE_EDGE
#define IPA_PIPE_EDGE 1     // ... the rim strips of such a launch too (columns / rows resolved)
#endif
#ifndef IPA_PIPE_SHARED
#define IPA_PIPE_SHARED 1   // frames of a workgroup share their map rows through LDS
#endif

namespace ipa {

// lane i <- lane i-1 (lane 0 keeps its own value)
__device__ __forceinline__ float from_lane_below(float v) {
  int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
// lane i <- lane i+1 (lane 63 keeps its own value)
__device__ __forceinline__ float from_lane_above(float v) {
  int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Two strip geometries:
//   HALO = false  strips step 256 - 8 HL px and overlap: lanes 0 .. HL-1 and 64-HL .. 63 only
//                 supply the horizontal halo (every kernel of rounds 1 and 2);
//   HALO = true   (round 3) strips are 256 px wide and 256-px ALIGNED, all 64 lanes store, and
//                 the 2 H halo pixels of a row are fetched / sampled by an extra pass of the
//                 lanes 0 .. 2H-1 into the pads of the LDS row.  A strip row is then written as
//                 whole 128-byte lines (the 992-byte rows of the overlapping strips end in
//                 partial lines shared with the neighbour strip) and a 4K row takes 15 strips,
//                 not 16: the plain 5x5 ran 1.01 -> 0.85 ms per 64 x 4K with this geometry,
//                 the fused undistort + 5x5 1.22 -> 1.12 (tools/ab_libs.py, one box).
#ifndef IPA_MIN_HL
#define IPA_MIN_HL 2   // (1: the 248-px step of rounds 1 - 4)
#endif
#ifndef IPA_MIN_HL_K7
#define IPA_MIN_HL_K7 1
#endif
#ifndef IPA_MIN_HL_K9
#define IPA_MIN_HL_K9 2
#endif
template <int K, bool HALO = false> struct wave_geom {
  static constexpr int H = K / 2;
  static constexpr bool kHalo = HALO;
  // halo lanes per side: what the window needs, and two for the short kernels (IPA_MIN_HL): their
  // strips then step 240 px and a strip row is 960 bytes that start and end on a 64-byte sector -
  // no sector of the result is written by two waves (the 992-byte rows of a 248-px step start 16
  // bytes into a line).  64 x 4K undistort + 5x5: -1.2 % on a box of the fast memory class, -2.7 /
  // -3.9 % on slow ones, same bits; the 7x7 kernels are bound by their vector work and pay 1 % for
  // the 3 % more samples; 9x9: -1.6 % fused, -0.8 % plain (profiles/r05_micro.txt)
  static constexpr int kMinHL = K <= 5 ? IPA_MIN_HL : (K == 7 ? IPA_MIN_HL_K7 : (K == 9 ? IPA_MIN_HL_K9 : 1));
  static constexpr int HL = HALO ? 0 : ((H + 3) / 4 > kMinHL ? (H + 3) / 4 : kMinHL);
  static constexpr int OW = 256 - 8 * HL;            // output pixels per strip row
  static constexpr int NW = 4 + 2 * H;               // window a lane needs per row
};
#ifndef IPA_HALO
#define IPA_HALO 1   // the hand-scheduled kernels (wave_pipe.hpp) use the aligned geometry
#endif

struct WaveParams {
  char* dst;
  long dst_frame_elems;
  int dh, dw;          // filter domain == output size
  long dpitch;
  int cbx, cby;        // filter border mode per axis
  int strips_x, strip_h;
  unsigned strips;     // per frame
  int vec_out;
  int frames_inner;    // 0: grid.y = frame; n: 1-D grid, frame index fastest (see the kernel)
  // strips another kernel computes (ring_stencil.hpp): skip[strip] != 0 -> nothing to do here
  const unsigned* skip = nullptr;
  int rim_only = 0;    // 1: the interior (FAST) strips belong to another kernel (wave_split.hpp)
  unsigned frame_major = 0;  // n > 0: 1-D grid of n blocks per frame, frame after frame - with the
                             // XCD-contiguous block order every XCD then streams through whole
                             // frames of its own (plain filters: no rows shared between frames;
                             // tools/pipe_micro.hip order 1: 835 -> 775 us per 64 x 4K copy)
  int no_pipe = 0;     // 1 (context knob pipe = 0): no strip takes the hand-scheduled loops of
                       // wave_pipe.hpp - every strip runs the compiler-scheduled chunked loop with
                       // its columns and rows resolved (the fallback and cross-check of the
                       // hand-counted waits; same bits, slower)
  int group_chunk = 0; // frames_wg launches: n > 0 = the frame groups are walked n at a time - all strips of n
                       // groups (4 n frames), then the next n groups - instead of all groups of a strip
                       // together (set by wave_grid; knob group_chunk)
  // SHORT strips at the end of every XCD's share of a chunked frames_wg launch (round 6).  A 64 x 4K launch pays ~95 us
  // - a tenth of its duration - over the marginal cost of 64 more frames, and that cost scales with the height of its
  // strips (2 T(64) - T(128) on the same memory: 95 / 61 / 47 / 41 us at 144 / 108 / 72 / 48 rows, tools/drain_probe.py):
  // the last round of workgroups drains the machine.  Short strips everywhere pay that back in halo rows, and a
  // block order that moves whole chunks to the end breaks the lock-step of the XCD pairs through the map rows
  // (+4 %: each pair then fetches its own map rows from HBM instead of finding the other pairs' in the Infinity
  // Cache).  So the geometry itself is non-uniform, identically for every chunk: a frame is `seg_count` segments of
  // seg_rows rows - one per XCD of a chunk's XCD group -, each seg_tall strips of strip_h rows followed by seg_short
  // strips of short_h rows; every XCD still owns a contiguous block range = one segment of one chunk, walks it in
  // lock-step with the others, and ends on short workgroups.  seg_count = 0: uniform strips.
  int seg_count = 0, seg_rows = 0, seg_tall = 0, seg_short = 0, short_h = 0;
  int frames_wg = 0;   // 1: the waves of a workgroup are consecutive FRAMES of one strip - the strip's
                       // map rows then reach the CU's L1 once per workgroup instead of once per
                       // frame (64 x 4K fused 5x5: 1.361 -> 1.335 ms); set by wave_grid
};

// grid for a launch over n_frames; fills p.frames_inner
// share_maps: the row source reads a coordinate table the frames of a batch share (MapCoord)
static inline dim3 wave_grid(ipa_ctx* ctx, WaveParams& p, int n_frames, int waves_per_block,
                             bool frames_inner, bool share_maps = false,
                             bool may_frame_major = false, int taps = 0) {
  unsigned blocks = (p.strips + waves_per_block - 1) / waves_per_block;
  frames_inner = frames_inner && ctx->tune.frames_inner != 0;
  p.frames_wg = 0;
  p.frame_major = 0;
  p.no_pipe = ctx->tune.pipe == 0;
  if (may_frame_major && !share_maps && ctx->tune.frame_major != 0 && n_frames > 1 &&
      (unsigned long)blocks * n_frames < (1ul << 31)) {
    p.frames_inner = 0;
    p.frame_major = blocks;
    return dim3(blocks * (unsigned)n_frames, 1);
  }
  if (frames_inner && share_maps && ctx->tune.frames_wg != 0 && !p.no_pipe &&
      n_frames % waves_per_block == 0 &&
      (unsigned long)p.strips * n_frames < (1ul << 31)) {
    p.frames_inner = n_frames;
    p.frames_wg = 1;
    const int groups = n_frames / waves_per_block;
    // the frame groups a quarter at a time (knob group_chunk: -1 = this rule, 0 = all groups of a strip
    // together as in rounds 2 - 4, n = chunks of n groups).  Measured, same bits (profiles/r05_micro.txt):
    // 64 x 4K undistort + 5x5 1.097 -> 1.044 ms with chunks of 4 of the 16 groups (2: 1.064, 8: 1.077, 1:
    // 1.068), 128 frames 2.01 -> 1.90 with 8 of 32, 16 frames 0.297 -> 0.284 with 1 of 4.  (Tried because
    // the strip-shaped STORE stream is what the slow regions of the device memory punish - loads and
    // linear stores are level everywhere, tools/region_micro.hip -; a store-only probe does not gain from
    // this order, the whole kernel does on every box: reads, shared map rows and stores of a chunk meet
    // in the same L2s)
    // A chunk size that does not divide the group count (20, 28, 36 ... frames) goes to the NEAREST divisor
    // (ties: the smaller) - it fell back to 0 without a trace until round 5; the value chosen is
    // read back as "group_chunk_used" (ipa_ctx_get_tuning).
    {
      int gc = ctx->tune.group_chunk;
      if (gc < 0) gc = groups >= 4 ? groups / 4 : 0;
      int best = 0;
      if (gc > 0 && gc < groups)
        for (int d = 1; d < groups; d++)
          if (groups % d == 0 && (best == 0 || abs(d - gc) < abs(best - gc))) best = d;
      p.group_chunk = best;
      ctx->group_chunk_used = best;
    }
    p.seg_count = 0;
    ctx->tail_rows_used = 0;
    if (p.group_chunk > 0 && ctx->tune.tail_rows != 0 && !p.skip && !p.rim_only) {
      // knob tail_rows: n > 0 = short strips of that many rows, 0 = uniform strips, -1 = the measured rule (4K frames,
      // in-process A/Bs, profiles/r06_micro.txt): what the shorter drain gives is roughly fixed per launch, what the short
      // strips cost - K - 1 more halo rows and one more prologue per strip - grows with the kernel and the launch.
      //   up to 16 frames: a quarter of the strip height for every kernel (16 x 4K: maps / homography + 9 + 9 -5.1 / -3.6 %,
      //     maps + 7 + 7 -6.0 %, C4 -0.6 %, headline +0.5 %);
      //   more: 3 / 5 taps a quarter (64 frames: headline 0.947 -> 0.924 ms, -2.3 .. -2.8 % on four boxes; 3 + 3 -2.6 %;
      //     32 frames -0.7 %), from 64 frames the dense 7 x 7 half the strip height (C4 -1.0 .. -1.4 %; a quarter: level;
      //     at 32 frames +1.0 %), separable 7 + 7 and every 9-tap kernel uniform (+1.3 .. +1.7 % otherwise);
      //   128 frames: level either way.
      // (taps > 0: a dense taps x taps filter, < 0: a separable one, 0: unknown = the short-kernel rule)
      int hs = ctx->tune.tail_rows;
      if (hs < 0) {
        const int k = taps < 0 ? -taps : taps;
        if (n_frames <= 16 || k <= 5) hs = p.strip_h / 4;
        else if (taps == 7 && n_frames >= 64) hs = p.strip_h / 2;
        else hs = 0;
        if (hs > 0 && hs < 24) hs = 24;
      }
      const int chunks = groups / p.group_chunk;
      // XCDs per chunk (the contiguous block ranges of xcd_swizzle): 8 / chunks; 8 chunks and more: whole chunks per XCD
      const int segs = chunks >= kXcds ? (chunks % kXcds == 0 ? 1 : 0) : (kXcds % chunks == 0 ? kXcds / chunks : 0);
      const int rows = segs ? (p.dh + segs - 1) / segs : 0;
      if (segs && hs > 0 && hs < p.strip_h && rows >= 2 * p.strip_h) {
        // about a fifth of a segment's rows on short strips; the tall ones cover the rest (the last of them clipped)
        int nshort = (rows / 5 + hs - 1) / hs;
        if (nshort < 1) nshort = 1;
        const int tall_rows = rows - nshort * hs;
        const int ntall = (tall_rows + p.strip_h - 1) / p.strip_h;
        const unsigned long strips = (unsigned long)segs * (ntall + nshort) * p.strips_x;
        if (tall_rows > 0 && strips * n_frames < (1ul << 31)) {
          p.seg_count = segs; p.seg_rows = rows; p.seg_tall = ntall; p.seg_short = nshort; p.short_h = hs;
          p.strips = (unsigned)strips;
          ctx->tail_rows_used = hs;
        }
      }
    }
    return dim3(p.strips * (unsigned)groups, 1);
  }
  if (frames_inner && n_frames > 1 && (unsigned long)blocks * n_frames < (1ul << 31)) {
    p.frames_inner = n_frames;
    return dim3(blocks * (unsigned)n_frames, 1);
  }
  p.frames_inner = 0;
  return dim3(blocks, (unsigned)n_frames);
}

// rows [y0, y0 + nrows) of strip row syi; false: an empty strip (the clipped end of a segment)
__device__ __forceinline__ bool wave_strip_rows(const WaveParams& p, int syi, int& y0, int& nrows) {
  if (!p.seg_count) {
    y0 = syi * p.strip_h;
    nrows = p.dh - y0 < p.strip_h ? p.dh - y0 : p.strip_h;
    return nrows > 0;
  }
  const int per = p.seg_tall + p.seg_short, seg = syi / per, idx = syi - seg * per;
  const int base = seg * p.seg_rows;
  const int rows = p.dh - base < p.seg_rows ? p.dh - base : p.seg_rows;       // of this segment
  int t0 = rows - p.seg_short * p.short_h;                                     // where its short strips begin
  t0 = t0 > 0 ? t0 : 0;
  if (idx < p.seg_tall) {
    const int o = idx * p.strip_h;
    y0 = base + o;
    nrows = t0 - o < p.strip_h ? t0 - o : p.strip_h;
  } else {
    const int o = t0 + (idx - p.seg_tall) * p.short_h;
    y0 = base + o;
    nrows = rows - o < p.short_h ? rows - o : p.short_h;
  }
  return nrows > 0;
}

// columns of the filter domain a lane covers, resolved once per strip
struct Cols {
  int xs;       // first column of the strip (lane 0) - wave-uniform, lives in an SGPR
  int xo;       // first column of the lane = xs + 4*lane (may be < 0 or >= dw at the rim)
  int uu[4];    // border-resolved column per pixel, -1 = constant border
  int uq[4];    // the same for the lane-interleaved pixels xs + lane + 64 k (sampling sources)
  int uh;       // HALO geometry: border-resolved halo column of lanes 0 .. 2H-1 (-1 = constant)
};
// position (floats from the start of a wave's LDS row) of the halo pixel lane j < 2H supplies:
// H pixels left of the strip, H pixels right of it
template <int H> __device__ __forceinline__ unsigned halo_pos(unsigned lane) {
  return lane < (unsigned)H ? lane : 256u + lane;   // + kRowPad - H
}

// ---------------------------------------------------------------- row sources --
// load_chunk<FAST, D>() issues the memory traffic of D consecutive rows
// (vv[d] = border-resolved row, -1 = constant border; never -1 when FAST);
// row<FAST, D>(d) turns row d of the chunk into the lane's 4 pixels.

// LDS row of a wave: 256 strip pixels + 4 pad floats on each side (the halo lanes' windows
// reach past the strip; what they read there is never used)
constexpr int kRowPad = 4, kRowStride = 256 + 2 * kRowPad;

// plain float32 image rows
struct LoadRowSrc {
#ifndef IPA_LOAD_DEPTH
#define IPA_LOAD_DEPTH 8
#endif
  // K = 7 holds 9 window pairs + 14 sum pairs per lane: a shallow chunk keeps occupancy 5
#ifndef IPA_K7_DEPTH
#define IPA_K7_DEPTH 2
#endif
  // (with the hand-scheduled FAST path of wave_pipe.hpp the chunked loop only runs the rim
  // strips of K <= 7: a shallow chunk keeps its registers and LDS rows out of the kernel's way)
  template <int K> struct depth {
    static constexpr int value = K >= 7 ? IPA_K7_DEPTH : (IPA_PIPE ? 2 : IPA_LOAD_DEPTH);
  };
  template <int D> struct Chunk { float v[D][4]; };
  const float* base;   // frame 0
  long frame_elems, pitch;
  int vec_in;
  float cval;          // filter border value
  __device__ __forceinline__ void set_frame(unsigned f) { base += (long)f * frame_elems; }
  __device__ __forceinline__ bool vectors_ok() const { return vec_in != 0; }

  static constexpr bool kHasQ5 = false;
  template <bool FAST, int D, int QM = -1>
  __device__ __forceinline__ void load_chunk(const Cols& c, const int (&vv)[D],
                                             Chunk<D>& ch) const {
#pragma unroll
    for (int d = 0; d < D; d++) {
      if constexpr (FAST) {
        // scalar row pointer + unsigned 32-bit lane offset: global_load with an SGPR base
        const float* rowp = base + ((long)vv[d] * pitch + c.xs);
        float4 q = *reinterpret_cast<const float4*>(rowp + 4u * (threadIdx.x & 63u));
        ch.v[d][0] = q.x; ch.v[d][1] = q.y; ch.v[d][2] = q.z; ch.v[d][3] = q.w;
      } else {
        const float* row = base + (long)(vv[d] < 0 ? 0 : vv[d]) * pitch;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float v = row[c.uu[k] < 0 ? 0 : c.uu[k]];
          ch.v[d][k] = (vv[d] < 0 || c.uu[k] < 0) ? cval : v;
        }
      }
    }
  }
  // rows of the chunk into the wave's LDS rows (natural pixel order, see wave_run_strip)
  template <bool FAST, int D>
  __device__ __forceinline__ void stage_rows(const Cols&, const int (&)[D], const Chunk<D>& ch,
                                             float* xp) const {
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 0; d < D; d++)
      *reinterpret_cast<float4*>(xp + d * kRowStride + kRowPad + 4u * lane) =
          float4{ch.v[d][0], ch.v[d][1], ch.v[d][2], ch.v[d][3]};
  }
  // HALO geometry, chunked loop (rim strips): the 2 H halo pixels of every row of the chunk
  template <int D, int H, int QM = -1>
  __device__ __forceinline__ void issue_halo(const Cols& c, const int (&vv)[D], Chunk<D>&) const {}
  template <int D, int H>
  __device__ __forceinline__ void stage_halo(const Cols& c, const int (&vv)[D], const Chunk<D>&,
                                             float* xp) const {
    const unsigned lane = threadIdx.x & 63u;
    if (lane < 2u * H) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        const float* row = base + (long)(vv[d] < 0 ? 0 : vv[d]) * pitch;
        const float v = row[c.uh < 0 ? 0 : c.uh];
        xp[d * kRowStride + kRowPad - H + halo_pos<H>(lane)] = (vv[d] < 0 || c.uh < 0) ? cval : v;
      }
    }
  }
};

// rows of the remapped image, sampled on the fly
template <typename ST, int INTERP, typename Coord> struct SampleRowSrc {
  using C = typename Coord::coord_t;
  using coord_type = Coord;
  using sample_type = ST;
  static constexpr bool kMap = std::is_same<Coord, MapCoord>::value;
#ifndef IPA_SAMPLE_DEPTH
#define IPA_SAMPLE_DEPTH 2
#endif
  template <int K> struct depth {
#ifndef IPA_SAMPLE_DEPTH_BIG
#define IPA_SAMPLE_DEPTH_BIG IPA_SAMPLE_DEPTH
#endif
    // float32 frames sampled bilinearly from a coordinate table run their FAST strips on
    // wave_pipe.hpp for K <= 5 (K = 7 streams its coefficients: wave_stencil_big_kernel); the chunked loop is then the rim strips only: depth 1
    // kShared: batches share footprint records through LDS (any coordinate source);
    // kPiped: table sources, whose single frames run the hand-scheduled loop too
    // (not float32 frames + 7x7 from a map pair: with 49 resident coefficients next to the
    // two-dword tap rows and the map pointers that kernel overflowed its SGPR spill lanes into
    // scratch memory - and scratch loads count in the hand-counted vmcnt waits; such batches run
    // the streamed kernel of fused_big.hip anyway, knob stream_k)
    static constexpr bool kShared = IPA_PIPE && K <= IPA_PIPE_MAX_K && INTERP == kLinear &&
                                    ((std::is_same<ST, float>::value &&
                                      (K <= 5 || !std::is_same<Coord, MapCoord>::value)) ||
                                     std::is_same<ST, uint16_t>::value || std::is_same<ST, uint8_t>::value);
    static constexpr bool kPiped = kShared && coord_is_table<Coord>::value && sizeof(C) == 4;
    static constexpr int value =
        INTERP == kLinear ? (K >= 9 ? IPA_SAMPLE_DEPTH_BIG : (kShared ? 1 : IPA_SAMPLE_DEPTH)) : 1;
  };
  template <int D> struct Chunk {
    BatchTaps<ST, INTERP, 4> t[D];
    BatchTaps<ST, INTERP, 1> th[D];   // HALO geometry: the halo sample of lanes 0 .. 2H-1
  };

  Coord coord;
  const char* src;       // frame 0 of the remap source
  long src_frame_bytes;
  unsigned src_bytes;
  int sh, sw, spitch;
  int border, q5;
  float cubic_a;
  const float* lanczos;
  float cval;            // remap border value
  float ccval;           // filter border value
  int map_vec;
  SrcView s;             // built by set_frame
  const char* fbase;     // this wave's frame (set_frame)

  __device__ __forceinline__ void set_frame(unsigned f) {
    fbase = src + (long)f * src_frame_bytes;
    s.rsrc = make_rsrc(fbase, src_bytes);
    s.h = sh; s.w = sw; s.pitch = spitch;
    s.border = border; s.q5 = q5; s.cubic_a = cubic_a; s.lanczos = lanczos;
    s.pair_split = 1;  // strips sample lane-interleaved
  }
  __device__ __forceinline__ bool vectors_ok() const { return !kMap || map_vec != 0; }

  // Strips sample in LANE-INTERLEAVED order: footprint k of lane L is strip pixel L + 64 k,
  // so the 64 gathers of one instruction walk along the source row (neighbouring lanes hit
  // the same cache lines) instead of striding 4 px; stage_rows writes the blended samples to
  // the wave's LDS row at their pixel positions.  Rim strips do the same through the
  // border-resolved columns c.uq (a constant filter border substitutes its value afterwards).
  template <bool FAST>
  __device__ __forceinline__ void coords_of_row(const Cols& c, int vv, C (&sx)[4],
                                                C (&sy)[4]) const {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if constexpr (FAST) coord.get(c.xs + lane + 64 * k, vv, sx[k], sy[k]);
      else coord.get(c.uq[k] < 0 ? 0 : c.uq[k], vv < 0 ? 0 : vv, sx[k], sy[k]);
    }
  }

  static constexpr bool kHasQ5 = true;  // FAST strips pick the coordinate rule per strip
  template <bool FAST, int D, int QM = -1>
  __device__ __forceinline__ void load_chunk(const Cols& c, const int (&vv)[D],
                                             Chunk<D>& ch) const {
    C sx[D][4], sy[D][4];
    if constexpr (kMap && FAST) {
      // every map row of the chunk first (4 coalesced dword loads per map row)
      const unsigned lane = threadIdx.x & 63u;
#pragma unroll
      for (int d = 0; d < D; d++) {
        const long o = (long)vv[d] * coord.pitch + c.xs;  // scalar
        const float* rx = coord.mx + o;
        const float* ry = coord.my + o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          sx[d][k] = rx[lane + 64u * k];
          sy[d][k] = ry[lane + 64u * k];
        }
      }
    } else if constexpr (kMap) {
      // rim strips: the same interleaved dword loads through the resolved columns
#pragma unroll
      for (int d = 0; d < D; d++) {
        const float* rx = coord.mx + (long)(vv[d] < 0 ? 0 : vv[d]) * coord.pitch;  // scalar
        const float* ry = coord.my + (long)(vv[d] < 0 ? 0 : vv[d]) * coord.pitch;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned u = (unsigned)(c.uq[k] < 0 ? 0 : c.uq[k]);
          sx[d][k] = rx[u];
          sy[d][k] = ry[u];
        }
      }
    } else {
#pragma unroll
      for (int d = 0; d < D; d++) coords_of_row<FAST>(c, vv[d], sx[d], sy[d]);
    }
    // stage B: footprints + all tap gathers of the chunk
#pragma unroll
    for (int d = 0; d < D; d++) batch_issue<ST, INTERP, 4, QM>(s, sx[d], sy[d], ch.t[d]);
  }

  // blend the chunk's samples and put the rows into the wave's LDS rows in natural pixel order
  template <bool FAST, int D>
  __device__ __forceinline__ void stage_rows(const Cols& c, const int (&vv)[D],
                                             const Chunk<D>& ch, float* xp) const {
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 0; d < D; d++) {
      float cur[4];
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = batch_blend_one<ST, INTERP, 4>(s, ch.t[d], k);
      if (ch.t[d].interior != 0xfu) {
        // footprints touching the source border (rare): from the taps issued where they can give
        // them (sampler.hpp::batch_blend_border), else redone tap by tap
        bool left = false;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (!((ch.t[d].interior >> k) & 1u))
            left = !batch_blend_border<ST, INTERP, 4>(s, ch.t[d], k, cval, cur[k]) || left;
        if (left) {
          C sx[4], sy[4];
          coords_of_row<FAST>(c, vv[d], sx, sy);
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (!((ch.t[d].interior >> k) & 1u))
              cur[k] = sample<ST, INTERP, C>(s, sx[k], sy[k], cval);
        }
      }
      float* row = xp + d * kRowStride + kRowPad;
      // lane-interleaved samples: pixel L + 64 k
      if constexpr (!FAST) {
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = (vv[d] < 0 || c.uq[k] < 0) ? ccval : cur[k];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) row[64u * k + lane] = cur[k];
    }
  }
  // HALO geometry, chunked loop (rim strips): the halo sample of lanes 0 .. 2H-1 through the
  // same batch machinery (taps issued with the chunk's other loads, blended with its rows)
  template <int D, int H, int QM = -1>
  __device__ __forceinline__ void issue_halo(const Cols& c, const int (&vv)[D], Chunk<D>& ch) const {
#pragma unroll
    for (int d = 0; d < D; d++) {
      C hx[1], hy[1];
      coord.get(c.uh < 0 ? 0 : c.uh, vv[d] < 0 ? 0 : vv[d], hx[0], hy[0]);
      batch_issue<ST, INTERP, 1, QM>(s, hx, hy, ch.th[d]);
    }
  }
  template <int D, int H>
  __device__ __forceinline__ void stage_halo(const Cols& c, const int (&vv)[D], const Chunk<D>& ch,
                                             float* xp) const {
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 0; d < D; d++) {
      float cur = batch_blend_one<ST, INTERP, 1>(s, ch.th[d], 0);
      if (!(ch.th[d].interior & 1u)) {
        if (!batch_blend_border<ST, INTERP, 1>(s, ch.th[d], 0, cval, cur)) {
          C hx, hy;
          coord.get(c.uh < 0 ? 0 : c.uh, vv[d] < 0 ? 0 : vv[d], hx, hy);
          cur = sample<ST, INTERP, C>(s, hx, hy, cval);
        }
      }
      cur = (vv[d] < 0 || c.uh < 0) ? ccval : cur;
      if (lane < 2u * H) xp[d * kRowStride + kRowPad - H + halo_pos<H>(lane)] = cur;
    }
  }
};

// -------------------------------------------------------------------- kernel --
// Row windows come from LDS and the K x K sums run in packed fp32:
// every arriving row is first written to a wave-private LDS row (natural pixel order), then
// each lane reads the K+2 overlapping PAIRS (px m, px m+1) of its 4 + 2H pixel window - the
// horizontal halo comes out of the same reads, no DPP exchange - and one v_pk_fma_f32 per
// coefficient and pixel pair advances two output pixels.  Summation order per pixel is
// unchanged (rows i, taps j ascending).
typedef float v2f __attribute__((ext_vector_type(2)));

// Coefficients: K <= 7 keeps all K*K in SGPRs (kernel arguments loaded once).  81 / 121 do not
// fit the scalar register file; wave_stencil_big_kernel re-reads them for every input row from
// its kernel-argument segment, one kernel row (12 SGPRs) at a time (`wk` = the padded rows of
// WaveBigArgs; the scalar cache serves them and the scalar unit is idle anyway).
typedef const float __attribute__((address_space(4)))* kernarg_f32;
typedef float v4f __attribute__((ext_vector_type(4)));

// v_pk_fma_f32 / v_pk_mul_f32 with ONE coefficient of an SGPR pair broadcast to both halves
// (op_sel): the K * K coefficients then take K * K scalar registers, not 2 * K * K as the
// {w, w} pairs the compiler forms on its own
template <int HI> __device__ __forceinline__ v2f pk_fma_coef(v2f wp, v2f x, v2f c) {
  v2f d;
  if constexpr (HI)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(wp), "v"(x), "v"(c));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(wp), "v"(x), "v"(c));
  return d;
}
template <int HI> __device__ __forceinline__ v2f pk_mul_coef(v2f wp, v2f x) {
  v2f d;
  if constexpr (HI)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "s"(wp), "v"(x));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "s"(wp), "v"(x));
  return d;
}

// The K + 2 overlapping pairs of a lane's window out of its wave's LDS row.
// Rounds 1 - 5 read every pair with a ds_read_b64 of its own: lanes at a 16-byte stride reading 8 bytes use half
// of the 64 banks, and lanes L and L + 16 of a 32-lane group meet on the same ones - a 2-way conflict on every
// read (SQ_LDS_BANK_CONFLICT = 38 % of the C4 launch's cycles, profiles/r05_micro.txt).  IPA_WINDOW_B128 (round
// 6): the 12 floats 4 L - 4 .. 4 L + 7 that hold every window up to 9 taps arrive as THREE aligned 16-byte reads
// (ds_read_b128: 16 lanes x 16 bytes cover the 64 banks exactly - conflict-free, 12 LDS cycles per row instead of
// 4 (K + 2)); the pairs that start on an odd float are formed from two register pairs with one v_pk_mov_b32
// each.  Same values in the same pairs: same bits.
#ifndef IPA_WINDOW_B128
#define IPA_WINDOW_B128 1
#endif
// (a.y, b.x): D.lo = src0.hi, D.hi = src1.lo
__device__ __forceinline__ v2f pk_mov_hi_lo(v2f a, v2f b) {
  v2f d;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// row = the wave's LDS row (float 0 = its left pad); pair[m] = (px 4 L - H + m, px 4 L - H + m + 1)
template <int K>
__device__ __forceinline__ void window_pairs(const float* row, unsigned lane, unsigned lane4_opaque,
                                             v2f (&pair)[K + 2]) {
  constexpr int H = K / 2;
  if constexpr (IPA_WINDOW_B128 != 0 && H <= 4 && kRowPad == 4) {
    const v4f* wv = reinterpret_cast<const v4f*>(row + 4u * lane);
    const v4f q0 = wv[0], q1 = wv[1], q2 = wv[2];
    const v2f e[6] = {v2f{q0.x, q0.y}, v2f{q0.z, q0.w}, v2f{q1.x, q1.y},
                      v2f{q1.z, q1.w}, v2f{q2.x, q2.y}, v2f{q2.z, q2.w}};
    static_for<0, K + 2>([&](auto M) {
      constexpr int m = decltype(M)::value, i = 4 - H + m;   // pair[m] = floats i, i + 1 of the twelve
      if constexpr ((i & 1) == 0) pair[m] = e[i / 2];
      else pair[m] = pk_mov_hi_lo(e[i / 2], e[i / 2 + 1]);
    });
  } else if constexpr (IPA_WINDOW_B128 != 0 && H == 5 && kRowPad == 4) {
    // 11 taps: the window 4 L - 5 .. 4 L + 8 lies in the 20 floats 4 L - 8 .. 4 L + 11: five aligned 16-byte reads (the first
    // starts 4 floats in front of the row for lane 0, the last ends 4 floats behind it for lane 63: the lead-in / lead-out
    // floats the kernels with H > kRowPad reserve around their rows, wave_stencil_body::kLead).  The plain 11 x 11 spent 72 %
    // of its LDS cycles on the conflicts of the thirteen 8-byte reads (LDS arrays busy 62 % of the launch).
    const v4f* wv = reinterpret_cast<const v4f*>(row + 4u * lane) - 1;
    const v4f q0 = wv[0], q1 = wv[1], q2 = wv[2], q3 = wv[3], q4 = wv[4];
    const v2f e[10] = {v2f{q0.x, q0.y}, v2f{q0.z, q0.w}, v2f{q1.x, q1.y}, v2f{q1.z, q1.w}, v2f{q2.x, q2.y},
                       v2f{q2.z, q2.w}, v2f{q3.x, q3.y}, v2f{q3.z, q3.w}, v2f{q4.x, q4.y}, v2f{q4.z, q4.w}};
    static_for<0, K + 2>([&](auto M) {
      constexpr int m = decltype(M)::value, i = 8 - H + m;   // pair[m] = floats i, i + 1 of the twenty
      if constexpr ((i & 1) == 0) pair[m] = e[i / 2];
      else pair[m] = pk_mov_hi_lo(e[i / 2], e[i / 2 + 1]);
    });
  } else {
    // Pairs at even and at odd m are read through two offsets the compiler cannot relate:
    // read once, the odd pairs would straddle register pairs and fall back to scalar fmas.
    const float* wp = row + kRowPad - H + 4u * lane;
    const float* wq = row + kRowPad - H + lane4_opaque;
#pragma unroll
    for (int m = 0; m < K + 2; m++) pair[m] = (m & 1) ? v2f{wq[m], wq[m + 1]} : v2f{wp[m], wp[m + 1]};
  }
}

template <bool FAST, typename Src, int K, int QM = -1, bool STREAM = false, bool HALO = false>
__device__ __forceinline__ void wave_run_strip(const WaveParams& p, const Src& src,
                                                   const Weights<float, K * K>& wts, float* xp,
                                                   const Cols& c, int y0, int nrows, bool writer,
                                                   float* dst, kernarg_f32 wk = nullptr) {
  using G = wave_geom<K, HALO>;
  constexpr int D = Src::template depth<K>::value;
  const int T = nrows + K - 1;  // input rows of this strip
  const unsigned lane = threadIdx.x & 63u;

  unsigned lane4_opaque = 4u * lane;
  asm volatile("" : "+v"(lane4_opaque));

  // acc[i][h] = running sums of output row (t - i), pixel pair h, after input row t was added
  v2f acc[K][2];
#pragma unroll 1
  for (int tb = 0; tb < T; tb += D) {
    int vv[D];  // wave-uniform row indices of the chunk (border-resolved on the rim path)
#pragma unroll
    for (int d = 0; d < D; d++) {
      if constexpr (FAST) vv[d] = y0 - G::H + tb + d;
      else vv[d] = resolve_idx(y0 - G::H + tb + d, p.dh, p.cby);
    }
    typename Src::template Chunk<D> ch;
    src.template load_chunk<FAST, D, QM>(c, vv, ch);
    if constexpr (HALO) src.template issue_halo<D, G::H, QM>(c, vv, ch);
    src.template stage_rows<FAST, D>(c, vv, ch, xp);
    if constexpr (HALO) src.template stage_halo<D, G::H>(c, vv, ch, xp);
    // wave-private LDS rows: the wave's own ds_write / ds_read execute in order
    __builtin_amdgcn_wave_barrier();

    static_for<0, D>([&](auto Dd) {
      constexpr int d = decltype(Dd)::value;
      const int t = tb + d;
      // pair[m] = (px 4L-H+m, px 4L-H+m+1), m = 0 .. K+1
      v2f pair[K + 2];
      window_pairs<K>(xp + d * kRowStride, lane, lane4_opaque, pair);

      // STREAM: kernel row i (padded to 12 floats, 16-byte aligned) arrives in 12 SGPRs while
      // row i + 1 is being accumulated: load(i - 1) is issued before the fmas of row i, the
      // wait for it stands in front of the fmas of row i - 1.  The asm statements carry the
      // running sums as in/out operands, which pins them between the fma groups (left to the
      // compiler, all K*K scalar loads are issued up front and spilled to VGPR lanes).
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
            // in front of row i's fmas (they read acc[i - 1]): a row of fmas covers the latency
            asm volatile("s_load_dwordx4 %0, %5, %6\n\ts_load_dwordx4 %1, %5, %7\n\t"
                         "s_load_dwordx4 %2, %5, %8"
                         : "=&s"(cc[i - 1][0]), "=&s"(cc[i - 1][1]), "=&s"(cc[i - 1][2]),
                           "+v"(acc[i - 1][0]), "+v"(acc[i - 1][1])
                         : "s"(wk), "n"((i - 1) * 48), "n"((i - 1) * 48 + 16),
                           "n"((i - 1) * 48 + 32));
          }
        }
        static_for<0, K>([&](auto Jj) {
          constexpr int j = decltype(Jj)::value;
          if constexpr (STREAM) {
            // the coefficient is one half of an SGPR pair, broadcast by op_sel: no scalar move
            // (and its wait state) per odd tap
            const v2f wp2 = v2f{cc[i][j >> 2][j & 2], cc[i][j >> 2][(j & 2) + 1]};
#pragma unroll
            for (int h = 0; h < 2; h++) {
              if constexpr (i == 0) {
                if constexpr (j == 0) acc[0][h] = pk_mul_coef<0>(wp2, pair[2 * h]);
                else acc[0][h] = pk_fma_coef<(j & 1)>(wp2, pair[j + 2 * h], acc[0][h]);
              } else {
                if constexpr (j == 0) acc[i][h] = pk_fma_coef<0>(wp2, pair[2 * h], acc[i - 1][h]);
                else acc[i][h] = pk_fma_coef<(j & 1)>(wp2, pair[j + 2 * h], acc[i][h]);
              }
            }
          } else {
            const float w = wts.w[i * K + j];
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
        });
        // keep the next kernel row's scalar loads below this row's fmas
        if constexpr (STREAM) __builtin_amdgcn_sched_barrier(0);
      });

      // output row t - (K-1) is complete
      const int o = t - (K - 1);
      if (o >= 0 && o < nrows && writer) {
        const float4 q = float4{acc[K - 1][0].x, acc[K - 1][0].y, acc[K - 1][1].x, acc[K - 1][1].y};
        float* row = dst + (long)(y0 + o) * p.dpitch + c.xo;
        const int n = p.dw - c.xo < 4 ? p.dw - c.xo : 4;
        if constexpr (FAST) {  // FAST strips: whole 16-byte-aligned chunks inside the image
          float* rows_ = dst + ((long)(y0 + o) * p.dpitch + c.xs);  // scalar base
          if constexpr (Src::kHasQ5) {
            // sampling sources: stream the output past the caches (nt), they are better spent
            // on the source lines adjacent rows and waves re-read (-2.5 % on the 4K chain)
            __builtin_nontemporal_store(q.x, rows_ + 4u * lane);
            __builtin_nontemporal_store(q.y, rows_ + 4u * lane + 1);
            __builtin_nontemporal_store(q.z, rows_ + 4u * lane + 2);
            __builtin_nontemporal_store(q.w, rows_ + 4u * lane + 3);
          } else {
            *reinterpret_cast<float4*>(rows_ + 4u * lane) = q;
          }
        } else if (p.vec_out && n == 4) {
          *reinterpret_cast<float4*>(row) = q;
        } else {
          const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (k < n) row[k] = e[k];
        }
      }
    });
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace ipa
#include "wave_pipe.hpp"  // FAST strips with a hand-scheduled memory pipeline (round 3)
namespace ipa {

#ifndef IPA_WAVE_MIN_WAVES
#define IPA_WAVE_MIN_WAVES 1
#endif
template <typename Src, int K, bool STREAM = false>
__device__ __forceinline__ void wave_stencil_body(const WaveParams& p, Src src,
                                                  const Weights<float, K * K>& wts,
                                                  kernarg_f32 wk) {
  constexpr bool HALO = geom_halo<Src, K, STREAM>::value;
  using G = wave_geom<K, HALO>;
  constexpr int D = Src::template depth<K>::value;
  const int lane = threadIdx.x & 63;
  // Two dispatch orders.  grid = (strip blocks, frames): frame after frame.  grid = (strip
  // blocks * frames, 1) with p.frames_inner = n_frames: the frames of ONE strip block are
  // neighbours in the (XCD-contiguous) order, so they run at the same time on the same XCD and
  // the read-only rows they share (the remap's map rows) are fetched into that L2 once.
  unsigned b = xcd_swizzle(blockIdx.x, gridDim.x), frame = blockIdx.y;
  if (p.frame_major) {
    frame = b / p.frame_major;
    b -= frame * p.frame_major;
  } else if (p.frames_inner) {
    frame = b % (unsigned)p.frames_inner;
    b /= (unsigned)p.frames_inner;
  }
  // the wave index as a SCALAR: everything derived from the strip id (rows, row addresses,
  // the FAST decision) then lives in SGPRs and is computed on the scalar unit
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned sid = b * IPA_WPB + wave;
  if (p.frames_wg) {  // b = (strip, group of IPA_WPB frames), groups fastest
    const unsigned groups = (unsigned)p.frames_inner / IPA_WPB;
    b = xcd_swizzle(blockIdx.x, gridDim.x);
    if (p.group_chunk) {   // the groups a chunk at a time: (chunk, strip, group of the chunk)
      const unsigned gc = (unsigned)p.group_chunk, per = gc * p.strips;
      const unsigned chunk = b / per, r = b - chunk * per;
      frame = (chunk * gc + r % gc) * IPA_WPB + wave;
      sid = r / gc;
    } else {
      frame = (b % groups) * IPA_WPB + wave;
      sid = b / groups;
    }
  }
  constexpr int kXp = kRowStride * D;  // LDS floats per wave
  // the windows of lane 0 start H px left of the row: H - kRowPad floats of lead-in for K = 11
  constexpr int kLead = G::H > kRowPad ? 4 : 0;
  __shared__ __attribute__((aligned(16))) float xpose[kLead + IPA_WPB * kXp + kLead];
  float* xp = xpose + kLead + wave * kXp;
  // map rows shared by the frames of a workgroup (wave_run_strip_shared): a ring of 2 IPA_WPB rows
  constexpr bool kShared = (IPA_PIPE != 0) && (IPA_PIPE_SHARED != 0) && !STREAM &&
                           shared_capable<Src, K>::value;
#ifndef IPA_DEBUG_LDS_PAD
#define IPA_DEBUG_LDS_PAD 0   // measurement only: extra LDS floats per workgroup (lowers the occupancy)
#endif
  __shared__ __attribute__((aligned(16))) float mapring[(kShared ? 2 * IPA_WPB * ring_row<HALO>::value : 4) + IPA_DEBUG_LDS_PAD];
  if (sid >= p.strips) return;  // whole wave
  if (p.skip && p.skip[sid]) return;
  const int syi = (int)(sid / (unsigned)p.strips_x), sxi = (int)sid - syi * p.strips_x;
  src.set_frame(frame);

  const int xs = sxi * G::OW - 4 * G::HL;  // first column of the strip (lane 0)
  Cols c;
  c.xs = xs;
  c.xo = xs + lane * 4;
  int y0, nrows;
  if (!wave_strip_rows(p, syi, y0, nrows)) return;
  const bool writer = lane >= G::HL && lane < 64 - G::HL && c.xo < p.dw;
  float* dst = reinterpret_cast<float*>(p.dst) + (long)frame * p.dst_frame_elems;

  // rows touched incl. the overshoot of the last chunk
  const int rows_touched = ((nrows + K - 1 + D - 1) / D) * D;
  // (HALO geometry: the halo columns of a FAST strip lie inside the image too)
  const int hx = HALO ? G::H : 0;
  const bool fast = !p.no_pipe && src.vectors_ok() && p.vec_out && xs - hx >= 0 &&
                    xs + 256 + hx <= p.dw && y0 - G::H >= 0 && y0 - G::H + rows_touched <= p.dh;
  if (fast && p.rim_only) return;
  if (fast) {
#pragma unroll
    for (int k = 0; k < 4; k++) c.uu[k] = c.xo + k;
    if constexpr (kShared) {
      // the waves of this workgroup are frames of ONE strip: every wave takes this branch
      if (p.frames_wg) {
        DenseFilter<K> filt(wts);
        if (src.q5) wave_run_strip_shared<K, 1, false, HALO>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
        else wave_run_strip_shared<K, 0, false, HALO>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
        return;
      }
    }
    if constexpr (IPA_PIPE && !STREAM && pipe_unshared<Src, K>::value) {
      if constexpr (Src::kHasQ5) {
        if (src.q5) wave_run_strip_pipe<K, 1, HALO>(p, src, wts, xp, c, y0, nrows, writer, dst);
        else wave_run_strip_pipe<K, 0, HALO>(p, src, wts, xp, c, y0, nrows, writer, dst);
      } else {
        wave_run_strip_pipe<K, HALO, false>(p, src, wts, xp, c, y0, nrows, writer, dst);
      }
    } else if constexpr (Src::kHasQ5) {
      // wave-uniform choice hoisted out of the per-sample code
      if (src.q5) wave_run_strip<true, Src, K, 1, STREAM>(p, src, wts, xp, c, y0, nrows, writer, dst, wk);
      else wave_run_strip<true, Src, K, 0, STREAM>(p, src, wts, xp, c, y0, nrows, writer, dst, wk);
    } else {
      wave_run_strip<true, Src, K, -1, STREAM>(p, src, wts, xp, c, y0, nrows, writer, dst, wk);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      c.uu[k] = resolve_idx(c.xo + k, p.dw, p.cbx);
      c.uq[k] = resolve_idx(xs + lane + 64 * k, p.dw, p.cbx);
    }
    if constexpr (HALO)
      c.uh = resolve_idx(xs - G::H + (int)halo_pos<G::H>(lane < 2 * G::H ? (unsigned)lane : 0u), p.dw, p.cbx);
    if constexpr (kShared) {
      // (vector alignment holds: the strip runs on the shared-map loop with its columns and
      // rows resolved through the filter's border mode; 15 % of a 4K frame's strips)
      if (p.frames_wg && !p.no_pipe && src.vectors_ok() && p.vec_out && (p.dw & 3) == 0 && IPA_PIPE_EDGE) {
        DenseFilter<K> filt(wts);
        if (src.q5) wave_run_strip_shared<K, 1, true, HALO>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
        else wave_run_strip_shared<K, 0, true, HALO>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
        return;
      }
    }
    if constexpr (HALO && !Src::kHasQ5 && IPA_PIPE_EDGE) {
      // plain rows: the rim strips on the hand-scheduled loop too (resolved columns and rows)
      if (!p.no_pipe && p.vec_out && (p.dw & 3) == 0) {
        wave_run_strip_pipe<K, HALO, true>(p, src, wts, xp, c, y0, nrows, writer, dst);
        return;
      }
    }
    wave_run_strip<false, Src, K, -1, STREAM, HALO>(p, src, wts, xp, c, y0, nrows, writer, dst, wk);
  }
}

// waves per SIMD the register allocation must leave room for: the hand-scheduled sampling
// kernels sit right at the 128-register step (4 waves per SIMD)
template <typename Src, int K> struct wave_min_waves {
  static constexpr int value = (pipe_capable<Src, K>::value && Src::kHasQ5 && IPA_PIPE && K <= 5) ? IPA_PIPE_MIN_WAVES : IPA_WAVE_MIN_WAVES;   // (7x7: 46 filter registers more - no cap, no spills)
};
template <typename Src, int K>
__global__ void __launch_bounds__(64 * IPA_WPB, (wave_min_waves<Src, K>::value))
wave_stencil_kernel(WaveParams p, Src src, Weights<float, K * K> wts) {
  wave_stencil_body<Src, K>(p, src, wts, nullptr);
}

// K = 9, 11: one argument struct, so that the coefficients' place in the kernel-argument
// segment is offsetof(WaveBigArgs, wts) whatever the argument layout rules are
template <typename Src, int K> struct WaveBigArgs {
  WaveParams p;
  Src src;
  alignas(16) float wrows[K][12];  // kernel row i, taps 0..K-1, zero padded
};
template <typename Src, int K>
__global__ void __launch_bounds__(64 * IPA_WPB, IPA_WAVE_MIN_WAVES)
wave_stencil_big_kernel(WaveBigArgs<Src, K> a) {
  static_assert(K <= 12, "padded coefficient rows of 12");
  typedef const char __attribute__((address_space(4)))* kernarg_bytes;
  kernarg_bytes base = (kernarg_bytes)__builtin_amdgcn_kernarg_segment_ptr();
  using Args = WaveBigArgs<Src, K>;
  kernarg_f32 wk = (kernarg_f32)(base + offsetof(Args, wrows));
  Weights<float, K * K> unused;  // the K <= 7 form of the coefficients, never read here
  wave_stencil_body<Src, K, true>(a.p, a.src, unused, wk);
}

// strip height: tall strips amortise the K-1 halo rows, short ones give small
// problems enough waves to fill 256 CUs
// strips_x: the strips per row of the caller's geometry (wave_geom / sep_geom: 240- or 248-px steps); 0 = estimated
static inline int wave_strip_height(const ipa_ctx* ctx, int dh, int dw, int n_frames, int K,
                                    bool fma_bound = false, int piped = 0, int strips_x = 0) {
  if (ctx->tune.strip_h > 0) return ctx->tune.strip_h;  // tuning knob
  // the hand-scheduled kernels of round 3 (wave_pipe.hpp; their rim strips run on the fast loop
  // too).  Plain filters (piped = 1) want many short strips - the tail of the launch's last
  // round of waves weighs more than the K - 1 halo rows: 64 x 4K 5x5, one box: 72 rows 0.738,
  // 48 0.740, 24 0.729 ms; 16 x 4K: 72 0.208, 36 0.199, 24 0.195.  The fused undistort + filter
  // on shared map rows (piped = 2) wants tall ones (every strip start costs a barrier-separated
  // prologue of dependent loads): 64 x 4K: 36 rows 1.066, 72 1.042, 108 1.031, 144 1.024 ms;
  // 16 x 4K: 24 0.296, 48 0.286, 72 0.283
  if (piped) {
    const long sx = strips_x > 0 ? strips_x : (dw + 255) / 256;
    const int cand2[4] = {144, 72, 48, 32};
    const long need2[4] = {12288, 6144, 4096, 0};
    if (piped == 2) {
      for (int i = 0; i < 4; i++)
        if (sx * ((dh + cand2[i] - 1) / cand2[i]) * n_frames >= need2[i]) return cand2[i];
    } else if (sx * ((dh + 23) / 24) * n_frames >= 8192) {
      return 24;
    }
  }
  int ow = 256 - 8 * ((K / 2 + 3) / 4);
  long sx = strips_x > 0 ? strips_x : (dw + ow - 1) / ow;
  // measured on 4K frames (MI355X, 4096 resident waves): with 16 frames 16-48 rows are within
  // noise of each other and 8 / 128+ clearly slower; with 64-128 frames per launch 48 rows
  // beat 32 by 2.3 % and 72 by 2.9 % (fewer halo rows sampled per output row), 90-135 fall
  // back -> the tallest strip that still leaves about 6 / 4 / 2 rounds of waves
  // 9x9 / 11x11 are bound by their K*K fmas per sample, halo rows included: 64 rows as soon
  // as they give two rounds of waves (16 x 4K: 11x11 461 -> 428 us, 9x9 334 -> 322)
  const int cand[5] = {fma_bound ? 64 : 72, 48, 32, 16, 8};
  const long need[5] = {fma_bound ? 8192 : 24576, fma_bound ? 8192 : 16384, 8192, 8192, 0};
  for (int i = 0; i < 5; i++) {
    long waves = sx * ((dh + cand[i] - 1) / cand[i]) * n_frames;
    if (waves >= need[i]) return cand[i];
  }
  return 8;
}

}  // namespace ipa
