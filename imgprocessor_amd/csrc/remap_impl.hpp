// remap_impl.hpp — the gather half of the hot path on gfx950:
//   ipa_build_undistort_map*   (cv2.initUndistortRectifyMap, LensDistortion.py:355-357)
//   ipa_remap*                 (cv2.remap,               LensDistortion.py:323-326)
//   ipa_undistort*             (the two above fused: no map arrays in HBM)
//   ipa_warp_perspective*      (cv2.warpPerspective,     PerspectiveCorrection.py:377-378,401-405)
//
// Kernel shape: one wave64 owns 256 consecutive output pixels of one row
// (4 px per lane -> 16-byte map loads and 16-byte stores, fully coalesced);
// a 256-thread workgroup owns a 256 x 4 output tile; tile ids are remapped so
// every XCD walks a contiguous band of the image (its L2 then holds the
// source rows neighbouring tiles share).  Source taps are gathered through a
// raw buffer descriptor per frame.  HBM-bound: 16 B/px map-based, 8 B/px
// analytic (f32).
#pragma once

#include <math.h>
#include <type_traits>

#include "common.hpp"
#include "sampler.hpp"
#include "ring_remap.hpp"
#include "stored_coords.hpp"
#include "tile_warp.hpp"

namespace ipa {

struct RemapParams {
  const char* src;
  char* dst;
  long src_frame_bytes;  // byte distance between consecutive source frames
  long dst_frame_elems;
  unsigned src_bytes;    // bytes of one source frame (descriptor range)
  int sh, sw, spitch;
  int dh, dw;
  long dpitch;
  int border, q5;
  float cubic_a;
  const float* lanczos;
  double cval;
  unsigned tiles_x, tiles;
  int dst_vec, map_vec;
  int frames_inner;  // n_frames when the grid is 1-D with the frame index fastest, else 0
  // tiles the ring kernel computes (ring_remap.hpp): skip[strip row * tiles_x + tile column]
  const int* tab2d;    // uint8 -> uint8 bicubic (a = -0.75): OpenCV's short weights, 1024 x 8 dwords
  const unsigned* skip;
  unsigned tile_rows;  // groups of 4 rows per workgroup (1; a strip of the skip mask behind it)
};

template <typename Coord>
__device__ __forceinline__ void coords4(const Coord& c, int u0, int v, int n, int /*vec*/,
                                        typename Coord::coord_t (&sx)[4],
                                        typename Coord::coord_t (&sy)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    sx[k] = 0;
    sy[k] = 0;
    if (k < n) c.get(u0 + k, v, sx[k], sy[k]);
  }
}

template <>
__device__ __forceinline__ void coords4<MapCoord>(const MapCoord& c, int u0, int v, int n, int vec,
                                                  float (&sx)[4], float (&sy)[4]) {
  long o = (long)v * c.pitch + u0;
  if (vec && n == 4) {
    float4 a = *reinterpret_cast<const float4*>(c.mx + o);
    float4 b = *reinterpret_cast<const float4*>(c.my + o);
    sx[0] = a.x; sx[1] = a.y; sx[2] = a.z; sx[3] = a.w;
    sy[0] = b.x; sy[1] = b.y; sy[2] = b.z; sy[3] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      sx[k] = 0;
      sy[k] = 0;
      if (k < n) {
        sx[k] = c.mx[o + k];
        sy[k] = c.my[o + k];
      }
    }
  }
}

template <typename DT>
__device__ __forceinline__ void store4(DT* row, int x0, const DT (&v)[4], int n, int vec) {
  if (vec && n == 4) {
    if constexpr (sizeof(DT) == 4) {
      *reinterpret_cast<float4*>(row + x0) = *reinterpret_cast<const float4*>(v);
    } else if constexpr (sizeof(DT) == 8) {
      reinterpret_cast<double2*>(row + x0)[0] = reinterpret_cast<const double2*>(v)[0];
      reinterpret_cast<double2*>(row + x0)[1] = reinterpret_cast<const double2*>(v)[1];
    } else if constexpr (sizeof(DT) == 2) {
      *reinterpret_cast<uint2*>(row + x0) = *reinterpret_cast<const uint2*>(v);
    } else {
      *reinterpret_cast<unsigned*>(row + x0) = *reinterpret_cast<const unsigned*>(v);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < n) row[x0 + k] = v[k];
  }
}

// REST: the launch behind a ring kernel's skip mask (float32 batches): tile_rows groups of rows
// per workgroup.  A separate instantiation - the loop costs the plain launch 10-15 %.
template <typename ST, typename DT, int INTERP, typename Coord, bool FIXED, bool REST = false>
__global__ void __launch_bounds__(256) remap_kernel(RemapParams p, Coord coord) {
  using CT = typename compute_of<ST>::type;
  // frames_inner: the frames of one tile are neighbours in the XCD-contiguous block order, so
  // a batch's frames read a map tile (and write neighbouring rows) at the same time on one XCD
  unsigned t = xcd_swizzle(blockIdx.x, gridDim.x), frame = blockIdx.y;
  if (p.frames_inner) {
    frame = t % (unsigned)p.frames_inner;
    t /= (unsigned)p.frames_inner;
  }
  // a workgroup = tile_rows groups of 4 rows x 256 px.  Behind a ring kernel's skip mask
  // tile_rows spans one strip of the mask: a launch that mostly has nothing to do is then
  // bound by far fewer workgroup dispatches (16 x 4K: 130 k -> 16 k, 55 -> 12 us)
  const unsigned tyb = t / p.tiles_x, txi = t - tyb * p.tiles_x;
  if (REST && p.skip[tyb * p.tiles_x + txi]) return;
  // Lanczos4: the 32 x 8 weight table is read four float4 per sample - from LDS, not through
  // the vector-memory path the 16 tap-row gathers of the sample already load
  // uint8 Lanczos4: OpenCV's short weights formed per sample from the float32 1-D table;
  // uint8 bicubic: the whole 2-D short table (32 KB) in LDS, a workgroup then works on
  // p.tile_rows groups of rows
  // uint16 -> uint16 with FIXED: OpenCV's 16U float-table arithmetic (sample_u16_cv); bicubic
  // reads the float32 bicubic rows of the same table
  constexpr bool kU16Cv = FIXED && std::is_same<ST, uint16_t>::value;
  constexpr bool kTabFixed = FIXED && (INTERP == kLanczos4 || (kU16Cv && INTERP == kCubic));
  constexpr bool kTabLds = FIXED && INTERP == kCubic && !kU16Cv;
  __shared__ __attribute__((aligned(16)))
  float lz_tab[kTabFixed ? 384 : (INTERP == kLanczos4 ? 256 : 4)];
  __shared__ __attribute__((aligned(16))) int4 tab_lds[kTabLds ? 1024 * kU8CubicRow : 1];
  if constexpr (INTERP == kLanczos4 || kTabFixed) {
    const unsigned tid = threadIdx.y * 64 + threadIdx.x;
    lz_tab[tid] = p.lanczos[tid];
    if constexpr (kTabFixed)
      if (tid < 128u) lz_tab[256 + tid] = p.lanczos[256 + tid];
    __syncthreads();
  }
  if constexpr (kTabLds) {
    const unsigned tid = threadIdx.y * 64 + threadIdx.x;
    const int4* gt = reinterpret_cast<const int4*>(p.tab2d);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const unsigned e = tid + 256u * i;
      tab_lds[(e >> 1) * kU8CubicRow + (e & 1u)] = gt[e];
    }
    __syncthreads();
  }
  SrcView s;
  s.rsrc = make_rsrc(p.src + (long)frame * p.src_frame_bytes, p.src_bytes);
  s.h = p.sh; s.w = p.sw; s.pitch = p.spitch;
  s.border = p.border; s.q5 = p.q5; s.cubic_a = p.cubic_a;
  s.lanczos = INTERP == kLanczos4 ? lz_tab : p.lanczos;
  const int x0 = (int)((txi * 64 + threadIdx.x) * 4);
  if (x0 >= p.dw) return;
  const int n = p.dw - x0 < 4 ? p.dw - x0 : 4;
  const unsigned tile_rows = (REST || kTabLds) ? p.tile_rows : 1u;
  for (unsigned sub = 0; sub < tile_rows; sub++) {
  const int y = (int)((tyb * tile_rows + sub) * 4 + threadIdx.y);
  if (y >= p.dh) return;

  // Row segments that lie wholly inside the output row sample in LANE-INTERLEAVED order
  // (footprint k of lane L = segment pixel L + 64 k): the 64 gathers of one instruction then
  // walk along the source row instead of striding 4 px (20-22 -> 16 clocks of the texture
  // addresser per dwordx2 gather, tools/ta_micro.hip), and the map values are four coalesced
  // dword loads.  A wave-private LDS row puts the results back into 4-px-per-lane order for
  // the 16-byte store.  Same samples, same arithmetic: identical results.
  // (map-based remaps only: the analytic coordinate sources measured 3-13 % slower this way)
  // (float destinations only: integer ones - uint8 / uint16, bound by their arithmetic - measured
  // level (uint8) to 8-20 % slower (uint16) this way)
  constexpr bool kIlv = !FIXED && sizeof(DT) == 4 && coord_is_table<Coord>::value;
  if constexpr (kIlv) {
    __shared__ __attribute__((aligned(16))) DT xpose[4][256];
    const int xw = (int)(txi * 256u);  // wave-uniform (threadIdx.y = wave)
    if (xw + 256 <= p.dw && p.dst_vec && p.map_vec) {
      SrcView si = s;  // neighbouring lanes, neighbouring pixels: dword pairs
      si.pair_split = 1;
      const unsigned lane = threadIdx.x;
      typename Coord::coord_t qx[4], qy[4];
      const long o = (long)y * coord.pitch + xw;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        qx[k] = coord.mx[o + lane + 64u * k];
        qy[k] = coord.my[o + lane + 64u * k];
      }
      CT cval = (CT)p.cval;
      constexpr int BN = batch_of<INTERP>::value;
      DT* xp = xpose[threadIdx.y];
#pragma unroll
      for (int b = 0; b < 4; b += BN) {
        typename Coord::coord_t bx[BN], by[BN];
        CT ov[BN];
#pragma unroll
        for (int j = 0; j < BN; j++) {
          bx[j] = qx[b + j];
          by[j] = qy[b + j];
        }
        sample_batch<ST, INTERP, BN>(si, bx, by, cval, ov);
#pragma unroll
        for (int j = 0; j < BN; j++) xp[64u * (b + j) + lane] = store_cast<DT, CT>(ov[j]);
      }
      __builtin_amdgcn_wave_barrier();  // wave-private row: in-order ds_write / ds_read
      const float4 q = *reinterpret_cast<const float4*>(xp + 4u * lane);
      DT* row = reinterpret_cast<DT*>(p.dst) + (long)frame * p.dst_frame_elems + (long)y * p.dpitch;
      *reinterpret_cast<float4*>(row + xw + 4u * lane) = q;
      __builtin_amdgcn_wave_barrier();  // the row is reused by the next group of rows
      continue;
    }
  }

  typename Coord::coord_t sx[4], sy[4];
  coords4<Coord>(coord, x0, y, n, p.map_vec, sx, sy);

  alignas(16) DT out[4];
  if constexpr (std::is_integral<DT>::value && !FIXED) {
    // integer destination: plain double arithmetic, then round-half-even + saturate
#pragma unroll
    for (int k = 0; k < 4; k++)
      out[k] = k < n ? store_cast<DT, double>(
                           sample_exact<ST, INTERP, typename Coord::coord_t>(s, sx[k], sy[k], p.cval))
                     : (DT)0;
  } else if constexpr (kU16Cv) {
    const double r16 = rint(p.cval);
    const uint16_t cv16 = (uint16_t)(r16 > 0 ? (r16 < 65535 ? r16 : 65535) : 0);
#pragma unroll
    for (int k = 0; k < 4; k++)
      out[k] = k < n ? sample_u16_cv<INTERP>(s, lz_tab, sx[k], sy[k], cv16) : (DT)0;
  } else if constexpr (FIXED) {
    double r = rint(p.cval);
    uint8_t cv8 = (uint8_t)(r > 0 ? (r < 255 ? r : 255) : 0);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if constexpr (INTERP == kLinear) out[k] = k < n ? sample_u8_fixed(s, sx[k], sy[k], cv8) : 0;
      else if constexpr (kTabLds) out[k] = k < n ? sample_u8_cubic_lds(s, tab_lds, sx[k], sy[k], cv8) : 0;
      else out[k] = k < n ? sample_u8_tab<INTERP>(s, sx[k], sy[k], cv8) : 0;
    }
  } else {
    CT cval = (CT)p.cval;
    constexpr int BN = batch_of<INTERP>::value;
#pragma unroll
    for (int b = 0; b < 4; b += BN) {
      typename Coord::coord_t bx[BN], by[BN];
      CT o[BN];
#pragma unroll
      for (int j = 0; j < BN; j++) {
        bx[j] = sx[b + j];  // lanes past the row end carry (0,0): a harmless interior sample
        by[j] = sy[b + j];
      }
      sample_batch<ST, INTERP, BN>(s, bx, by, cval, o);
#pragma unroll
      for (int j = 0; j < BN; j++) out[b + j] = store_cast<DT, CT>(o[j]);
    }
  }
  DT* row = reinterpret_cast<DT*>(p.dst) + (long)frame * p.dst_frame_elems + (long)y * p.dpitch;
  store4<DT>(row, x0, out, n, p.dst_vec);
  }
}

#ifdef IPA_REMAP_API_TU  // only remap.hip carries the map builder
__global__ void __launch_bounds__(256)
build_map_kernel(UndistortCoord c, int h, int w, float* mapx, float* mapy, long pitch, int vec) {
  int x0 = (int)((blockIdx.x * 64 + threadIdx.x) * 4);
  int y = (int)(blockIdx.y * 4 + threadIdx.y);
  if (y >= h || x0 >= w) return;
  int n = w - x0 < 4 ? w - x0 : 4;
  alignas(16) float sx[4], sy[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    sx[k] = 0;
    sy[k] = 0;
    if (k < n) c.get(x0 + k, y, sx[k], sy[k]);
  }
  store4<float>(mapx + (long)y * pitch, x0, sx, n, vec);
  store4<float>(mapy + (long)y * pitch, x0, sy, n, vec);
}

#endif

}  // namespace ipa



using namespace ipa;

int ipa_check_interp_border(ipa_ctx* ctx, int interp, int border);  // remap.hip
int ipa_lanczos_table(ipa_ctx* ctx, const float** out);               // remap.hip
int ipa_u8_cubic_tab2d(ipa_ctx* ctx, const int** out);                // remap.hip
int ipa_u8_lanczos_tab2d(ipa_ctx* ctx, const int** out);              // remap.hip

struct RemapCall {
  const void* src; int src_dt; int sh, sw; long spitch;
  void* dst; int dst_dt; int dh, dw; long dpitch;
  int n_frames; long src_fs, dst_fs;
  int interp, border; double cval;
};

template <typename ST, typename DT, typename Coord, bool FIXED>
static void launch_interp(ipa_ctx* ctx, const RemapParams& p, const Coord& c, int base, dim3 grid) {
  dim3 block(64, 4);
  switch (base) {
    case IPA_INTER_NEAREST:
      hipLaunchKernelGGL((remap_kernel<ST, DT, kNearest, Coord, false>), grid, block, 0,
                         ctx->stream, p, c);
      break;
    case IPA_INTER_LINEAR:
      hipLaunchKernelGGL((remap_kernel<ST, DT, kLinear, Coord, FIXED>), grid, block, 0,
                         ctx->stream, p, c);
      break;
    case IPA_INTER_CUBIC_CV:  // (uint8 -> uint8: cv2's fixed-point tables)
      hipLaunchKernelGGL((remap_kernel<ST, DT, kCubic, Coord, FIXED>), grid, block, 0,
                         ctx->stream, p, c);
      break;
    case IPA_INTER_CUBIC_KEYS:
      hipLaunchKernelGGL((remap_kernel<ST, DT, kCubic, Coord, false>), grid, block, 0,
                         ctx->stream, p, c);
      break;
    default:
      hipLaunchKernelGGL((remap_kernel<ST, DT, kLanczos4, Coord, FIXED>), grid, block, 0,
                         ctx->stream, p, c);
      break;
  }
}

// uint8 -> uint8 Lanczos4 (cv2.warpPerspective / remap on the camera's 8-bit frames, the default of
// PerspectiveCorrection.correct, camera/PerspectiveCorrection.py:401-405): OpenCV's whole 8U weight
// table (1024 fraction pairs x 64 shorts = 128 KB) in the LDS of a 1024-thread workgroup that works
// on p.tile_rows groups of 16 rows x 256 px.  Full 256-px segments sample lane-interleaved (lane L
// = pixels L + 64 k: the tap dwords of neighbouring lanes share cache lines) and go through a
// wave-private byte row back to 4 pixels per lane for the store.
template <typename Coord>
__global__ void __launch_bounds__(1024) remap_u8_lz_kernel(RemapParams p, Coord coord) {
  __shared__ __attribute__((aligned(16))) int4 tab[1024 * kU8LzRow];  // 147 KB
  __shared__ __attribute__((aligned(16))) uint8_t xrow[16][256];
  const unsigned tid = threadIdx.x;
  {
    const int4* gt = reinterpret_cast<const int4*>(p.tab2d);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const unsigned e = tid + 1024u * i;  // int4 e of the packed table = row e / 8, part e % 8
      tab[(e >> 3) * kU8LzRow + (e & 7u)] = gt[e];
    }
    __syncthreads();
  }
  const unsigned lane = tid & 63u, wave = tid >> 6;
  // the frames of a tile are neighbours in the grid: their workgroups read the tile's map rows
  // at about the same time (placing them on one XCD as well measured the same)
  const unsigned nf = (unsigned)p.frames_inner, t = blockIdx.x / nf, frame = blockIdx.x - t * nf;
  const unsigned tyb = t / p.tiles_x, txi = t - tyb * p.tiles_x;
  SrcView s;
  const char* fbase = p.src + (long)frame * p.src_frame_bytes;
  const unsigned mis = (unsigned)(uintptr_t)fbase & 3u;   // odd h x w: every other frame
  s.rsrc = make_rsrc(fbase - mis, p.src_bytes + mis);
  s.org = (int)mis;
  s.h = p.sh; s.w = p.sw; s.pitch = p.spitch;
  s.border = p.border; s.q5 = 1; s.cubic_a = p.cubic_a; s.lanczos = nullptr;
  const double rv = rint(p.cval);
  const uint8_t cv8 = (uint8_t)(rv > 0 ? (rv < 255 ? rv : 255) : 0);
  const int xw = (int)(txi * 256u);
  // rows that are not dword aligned (an odd width) and the last, partial strip of a row keep the
  // lane-interleaved sampling and store bytes (16 x 2160 x 3838: 1423 -> 937 us with the tap
  // dwords aligned in memory, sampler.hpp; profiles/r03_micro.txt)
  const bool whole = xw + 256 <= p.dw;
  uint8_t* dst = reinterpret_cast<uint8_t*>(p.dst) + (long)frame * p.dst_frame_elems;
  for (unsigned it = 0; it < p.tile_rows; it++) {
    const int y = (int)((tyb * p.tile_rows + it) * 16u + wave);
    if (y >= p.dh) return;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {  // one sample at a time: 16 tap dwords + 32 weight dwords live
      const int x = xw + (int)lane + 64 * k;
      if (whole || x < p.dw) {     // (the last strip of a row samples the same way, lanes past it idle)
        typename Coord::coord_t sx, sy;
        coord.get(x, y, sx, sy);
        xrow[wave][64u * k + lane] = sample_u8_lanczos_lds(s, tab, sx, sy, cv8);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (whole && p.dst_vec) {
      const unsigned q = *reinterpret_cast<const unsigned*>(&xrow[wave][4u * lane]);
      *reinterpret_cast<unsigned*>(dst + (long)y * p.dpitch + xw + 4u * lane) = q;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {   // (lane-interleaved: 64 consecutive bytes per store)
        const int x = xw + 64 * k + (int)lane;
        if (x < p.dw) dst[(long)y * p.dpitch + x] = xrow[wave][64u * k + lane];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// the strips a ring kernel left (float32 -> float32, p.skip / p.tile_rows set)
template <typename Coord>
static void launch_rest(ipa_ctx* ctx, const RemapParams& p, const Coord& c, int base, dim3 grid) {
  dim3 block(64, 4);
  switch (base) {
    case IPA_INTER_LINEAR:
      hipLaunchKernelGGL((remap_kernel<float, float, kLinear, Coord, false, true>), grid, block, 0,
                         ctx->stream, p, c);
      break;
    case IPA_INTER_CUBIC_CV:
    case IPA_INTER_CUBIC_KEYS:
      hipLaunchKernelGGL((remap_kernel<float, float, kCubic, Coord, false, true>), grid, block, 0,
                         ctx->stream, p, c);
      break;
    default:
      hipLaunchKernelGGL((remap_kernel<float, float, kLanczos4, Coord, false, true>), grid, block, 0,
                         ctx->stream, p, c);
      break;
  }
}

static inline bool aligned_rows(const void* base, long pitch_elems, long frame_elems, int n_frames,
                         size_t elem, size_t vec_bytes) {
  if (((uintptr_t)base) % vec_bytes) return false;
  if ((pitch_elems * (long)elem) % (long)vec_bytes) return false;
  if (n_frames > 1 && (frame_elems * (long)elem) % (long)vec_bytes) return false;
  return true;
}

// plan + ring kernel for the clean strips of a batch; sets p.skip for remap_kernel.
// Returns 1 when the call is not covered.
template <int INTERP, typename Coord>
static void ring_remap_launch_one(ipa_ctx* ctx, const RingGeom& gm, const RingRemapArgs& ra,
                                  const Coord& coord, const RingPlan& plan) {
  using RK = RingRemapKernel<INTERP, Coord>;
  const unsigned groups = ((unsigned)ra.n_frames + RK::kWaves - 1) / RK::kWaves;
  hipLaunchKernelGGL((ring_remap_kernel<INTERP, Coord>), dim3((unsigned)gm.strips * groups),
                     dim3(64 * RK::kWaves), 0, ctx->stream, gm, ra, coord, plan);
}

template <typename Coord>
static int ring_remap_launch(ipa_ctx* ctx, RemapParams& p, const Coord& coord, int base,
                             int n_frames) {
  RingGeom gm;
  gm.dh = p.dh; gm.dw = p.dw;
  gm.strips_x = (p.dw + kSW - 1) / kSW;
  gm.pairs_x = (gm.strips_x + 1) / 2;  // == tiles_x of remap_kernel (256-px tiles)
  gm.strip_h = 32;
  const int rows = (p.dh + gm.strip_h - 1) / gm.strip_h;
  gm.strips = gm.strips_x * rows;
  gm.pairs = gm.pairs_x * rows;
  if ((unsigned)gm.pairs_x != p.tiles_x) return 1;
  RingTaps tp;
  tp.nt = base == IPA_INTER_LINEAR ? 2 : (base == IPA_INTER_LANCZOS4 ? 8 : 4);
  tp.q5 = (p.q5 || base == IPA_INTER_LANCZOS4) ? 1 : 0;
  tp.rr = base == IPA_INTER_LANCZOS4 ? ring_rows<kLanczos4>::value : ring_rows<kLinear>::value;
  RingPlan plan;
  typename ring_kernel_coord<Coord>::type kc;
  int rc = ring_plan_prepare<Coord, 1>(ctx, gm, coord, p.sh, p.sw, tp, &plan, &kc);
  if (rc) return rc;
  using KCoord = typename ring_kernel_coord<Coord>::type;
  RingRemapArgs ra;
  ra.dst = p.dst; ra.dst_frame_elems = p.dst_frame_elems; ra.dpitch = p.dpitch;
  ra.src = p.src; ra.src_frame_bytes = p.src_frame_bytes; ra.src_bytes = p.src_bytes;
  ra.spitch = p.spitch; ra.n_frames = n_frames; ra.q5 = p.q5; ra.cubic_a = p.cubic_a;
  ra.lanczos = p.lanczos;
  switch (base) {
    case IPA_INTER_LINEAR: ring_remap_launch_one<kLinear, KCoord>(ctx, gm, ra, kc, plan); break;
    case IPA_INTER_CUBIC_CV:
    case IPA_INTER_CUBIC_KEYS: ring_remap_launch_one<kCubic, KCoord>(ctx, gm, ra, kc, plan); break;
    default: ring_remap_launch_one<kLanczos4, KCoord>(ctx, gm, ra, kc, plan); break;
  }
  p.skip = plan.pair_clean;
  p.tile_rows = (unsigned)gm.strip_h / 4u;
  return 0;
}

// float32 perspective warps on the tile kernel (tile_warp.hpp); 1: not covered / does not pay.
// Where it pays (tile_warp = 1), measured against the ring + gather kernels on 480p ... 8K frames,
// batches of 1 ... 16, rotations, perspective quadrilaterals and zooms (profiles/r04_micro.txt):
//   bilinear  batches of 8+ frames and 64+ Mpx (4+ frames from 100 Mpx) of a picture that is
//             enlarged (fetch <= 1.05: small boxes), or under a perspective / rotation whose rows
//             drift a little (0.01+ rows per pixel) with boxes up to fetch 1.2 (the bench's
//             quadrilateral: 0.37 -> 0.23 ms), or whose rows drift by 0.1+ (the gathers then pay
//             per cache line) while the boxes stay moderate (fetch <= 2.6); at scale 1 without
//             rotation the two are level;
//   bicubic   batches of 4+ frames with 16+ Mpx (up to 2x), single 4K frames from a drift of 0.3;
//             on small frames the gather kernel is 25 % ahead, and where the picture shrinks so
//             much that a 32 x 16 tile reads 8+ times its pixels (fetch > 8) too;
//   Lanczos4  whenever the homography fits: up to 3 times faster (16 x 4K at scale 1: 1.19 ->
//             0.62 ms), level with the ring kernel on small batches of an enlarged picture.
// (tools/warp_policy_matrix.py; re-measured after the row-walking kernels stopped paying for
// footprints on the source border)
// what of that does not depend on the boxes: false = the tile kernel does not pay whatever they are
// (tested BEFORE the host walks over the tiles: 0.4 ms per new matrix on a 4K frame)
static inline bool tile_warp_may_pay(int base, int n_frames, long px) {
  const double work = (double)n_frames * (double)px;
  if (base == IPA_INTER_LINEAR) return (n_frames >= 8 && work >= 64e6) || (n_frames >= 4 && work >= 100e6);
  if (base == IPA_INTER_LANCZOS4) return true;
  return (n_frames >= 4 && work >= 16e6) || work >= 8e6;
}
static inline bool tile_warp_pays(const ipa_ctx::TileWarpPlan& pl, int base, int n_frames, long px) {
  const double d = pl.drift, g = pl.fetch;   // (step: recorded, not used)
  const double work = (double)n_frames * (double)px;
  if (base == IPA_INTER_LINEAR)
    return ((n_frames >= 8 && work >= 64e6) || (n_frames >= 4 && work >= 100e6)) &&
           (g <= 1.05 || (d >= 0.01 && g <= 1.2) || (d >= 0.1 && g <= 2.6));
  if (base == IPA_INTER_LANCZOS4) return true;
  return ((n_frames >= 4 && work >= 16e6) || (d >= 0.3 && work >= 8e6)) && g <= 8.0;
}

template <int INTERP, typename ST = float>
static int tile_warp_launch(ipa_ctx* ctx, const RemapParams& p, const HomographyCoord& coord,
                            int n_frames, int base) {
  constexpr bool kU16 = std::is_same<ST, uint16_t>::value;
  constexpr int NT = ntaps<INTERP>::value;
  TileWarpArgs t;
  int shape;
  {
    // (the conditions that do not depend on the boxes first: a call they reject pays no walk)
    if (ctx->tune.tile_warp < 2) {
      if (kU16 ? (double)n_frames * p.dh * p.dw < 2e6 : !tile_warp_may_pay(base, n_frames, (long)p.dh * p.dw))
        return 1;
    }
    double key[14] = {(double)(INTERP + (kU16 ? 16 : 0)), (double)p.dh, (double)p.dw, (double)p.sh, (double)p.sw};
    for (int k = 0; k < 9; k++) key[5 + k] = coord.m[k];
    ipa_ctx::TileWarpPlan* pl = nullptr;
    for (auto& q : ctx->tile_warp_plans)
      if (q.valid && memcmp(key, q.key, sizeof key) == 0) pl = &q;
    if (!pl) {
      pl = &ctx->tile_warp_plans[0];
      for (auto& q : ctx->tile_warp_plans)
        if (!q.valid || q.used < pl->used) { pl = &q; if (!q.valid) break; }
      // the largest tile shape whose source boxes fit (tile_warp.hpp)
      int pitch = 0, rows = 0, sh = 0;
      bool ok = false;
      for (sh = 0; sh < kWarpShapes && !ok; sh++)
        ok = tile_warp_box<NT, kU16>(coord.m, p.dh, p.dw, p.sh, p.sw, kWarpTileWs[sh], kWarpTileHs[sh],
                                     &pitch, &rows);
      sh -= 1;
      pl->ok = ok ? 1 : 0;
      pl->shape = sh;
      pl->rows = rows;
      pl->pitch = ok ? tile_warp_pitch<NT>(coord.m, p.dh, p.dw, pitch, rows) : 0;
      if (ok)
        tile_warp_measure(coord.m, p.dh, p.dw, pitch, rows, kWarpTileWs[sh], kWarpTileHs[sh],
                          &pl->drift, &pl->step, &pl->fetch);
      memcpy(pl->key, key, sizeof key);
      pl->valid = 1;
    }
    pl->used = ++ctx->tile_warp_clock;
    if (!pl->ok) return 1;
    if constexpr (!kU16) {
      if (ctx->tune.tile_warp < 2 && !tile_warp_pays(*pl, base, n_frames, (long)p.dh * p.dw)) return 1;
    }
    t.pitch = pl->pitch;
    t.rows = pl->rows;
    shape = pl->shape;
  }
  const int TW = kWarpTileWs[shape], TH = kWarpTileHs[shape];
  t.slow_count = nullptr;
  t.dst = p.dst; t.dst_frame_elems = p.dst_frame_elems; t.dpitch = p.dpitch;
  t.src = p.src; t.src_frame_bytes = p.src_frame_bytes; t.src_bytes = p.src_bytes;
  t.sh = p.sh; t.sw = p.sw; t.spitch = p.spitch; t.dh = p.dh; t.dw = p.dw;
  t.n_frames = n_frames;
  t.border = p.border; t.q5 = p.q5; t.cubic_a = p.cubic_a; t.lanczos = p.lanczos;
  t.cval = (float)p.cval;
  t.tiles_x = (p.dw + TW - 1) / TW;
  t.tiles = t.tiles_x * ((p.dh + TH - 1) / TH);
  // frames a workgroup walks through with one evaluation of its tile's coordinates: as many as
  // still leave the launch four rounds of workgroups (4 per CU)
  t.frames_wg = 8;
  while (t.frames_wg > 1 && (long)t.tiles * ((n_frames + t.frames_wg - 1) / t.frames_wg) < 4096) t.frames_wg >>= 1;
  if (t.frames_wg > n_frames) t.frames_wg = n_frames;
  const size_t dbytes = ((size_t)(p.dh - 1) * p.dpitch + p.dw) * sizeof(ST);
  if (dbytes >= (1ull << 31)) return 1;
  t.dst_bytes = (unsigned)dbytes;
  const unsigned groups = ((unsigned)n_frames + t.frames_wg - 1) / (unsigned)t.frames_wg;
  if ((unsigned long)t.tiles * groups >= (1ul << 31)) return 1;
  {   // (wave_grid's rule for the frame groups of a launch: a quarter of them at a time)
    int gc = ctx->tune.group_chunk;
    if (gc < 0) gc = groups >= 4 ? (int)groups / 4 : 0;
    t.group_chunk = (gc > 0 && gc < (int)groups && groups % (unsigned)gc == 0) ? gc : 0;
  }
  const size_t lds = (size_t)tile_warp_lds_bytes<NT>(t.pitch, t.rows);
  if (shape == 0) tile_warp_run_a(ctx->stream, t, coord, INTERP, kU16, shape, (unsigned)t.tiles * groups, lds);
  else tile_warp_run_b(ctx->stream, t, coord, INTERP, kU16, shape, (unsigned)t.tiles * groups, lds);
  return 0;
}

// map-based remaps of float32 frames on the tile kernel: the box of a tile is the span of its own
// footprints, the LDS a fixed reserve (what does not fit goes tap by tap).  knob tile_warp as above
template <int INTERP>
static int tile_warp_launch_map(ipa_ctx* ctx, const RemapParams& p, const MapCoord& coord, int n_frames) {
  constexpr int NT = ntaps<INTERP>::value;
  TileWarpArgs t;
  // (the registers limit every instantiation to 4 workgroups per CU: 39 KB of box per workgroup
  // cost no occupancy - 127 columns x 72 rows hold a 64 x 32 tile of a map that shrinks the picture
  // up to 1.8 x 2 times)
  // (bilinear fits 5 workgroups per CU into its registers: a reserve of 97 x 48 - a lens map's
  // tile with its halo and some drift - keeps them; 16 x 4K: 0.228 against 0.281 ms)
  // bicubic: beyond a zoom of 1.4 its boxes make it slower than the gather kernel (1.7: 0.66
  // against 0.48 ms) - a reserve that does not hold them hands such maps back through the count
  // (pitch: the homography's boxes get theirs from a conflict count over its tap lines, tile_warp_pitch; a map is device
  // data.  Measured on the lens maps, 16 x 4K, 97 / 127 against 96 / 128: bilinear 0.2547 / 0.2565 ms, bicubic 0.3510 /
  // 0.3494, Lanczos4 0.6157 / 0.6086 - a multiple of 32 pays for the kernel whose LDS arrays are the bound)
  t.pitch = NT == 8 ? 128 : 97;
  t.rows = NT == 2 ? 48 : (NT == 8 ? 72 : 56);
  // A map whose tiles need more - the host cannot know: the map is device data - sends the pixels
  // outside the box tap by tap, at 10 - 100 times the cost (16 x 4K zoomed out 2.5 x: Lanczos4
  // 4.7 ms against 1.2 on the gather kernel).  The kernel counts them for its first frame group;
  // the count comes back without a wait (so: one call late), and from the second call on a map
  // pair with more than 0.5 % such pixels takes the ring / gather kernels (every 64th call tries
  // again: the map's contents may have changed behind the same pointers)
  ipa_ctx::TileSlowHint* hint = nullptr;
  int hslot = 0;
  {
    double key[10] = {(double)(uintptr_t)coord.mx, (double)(uintptr_t)coord.my, (double)coord.pitch,
                      (double)p.dh, (double)p.dw, (double)p.sh, (double)p.sw, (double)INTERP,
                      (double)p.border, (double)p.q5};
    if (!ctx->tile_slow_dev) {
      // all or nothing: the words and the events are published to the context only once every one of them
      // exists - a failure half way frees what it made and leaves the hint state absent, so that the next
      // call tries again; this call goes to the ring / gather kernels (return 1: "not taken")
      unsigned *dev = nullptr, *host = nullptr;
      hipEvent_t evs[ipa_ctx::kTileSlowHints] = {};
      bool ok = hipMalloc((void**)&dev, ipa_ctx::kTileSlowHints * sizeof(unsigned)) == hipSuccess &&
                hipHostMalloc((void**)&host, ipa_ctx::kTileSlowHints * sizeof(unsigned)) == hipSuccess;
      for (int k = 0; ok && k < ipa_ctx::kTileSlowHints; k++)
        ok = hipEventCreateWithFlags(&evs[k], hipEventDisableTiming) == hipSuccess;
      if (!ok) {
        for (hipEvent_t e : evs)
          if (e) (void)hipEventDestroy(e);
        if (host) (void)hipHostFree(host);
        if (dev) (void)hipFree(dev);
        (void)hipGetLastError();
        return 1;
      }
      for (int k = 0; k < ipa_ctx::kTileSlowHints; k++) {
        host[k] = 0;
        ctx->tile_slow[k].copied = evs[k];
      }
      ctx->tile_slow_host = host;
      ctx->tile_slow_dev = dev;
    }
    for (int k = 0; k < ipa_ctx::kTileSlowHints; k++)
      if (ctx->tile_slow[k].valid && memcmp(key, ctx->tile_slow[k].key, sizeof key) == 0) { hint = &ctx->tile_slow[k]; hslot = k; }
    if (!hint) {
      // a new map pair: the least recently used slot.  Its word may still receive the old key's
      // last copy - `launches` = 0 keeps it from being read as this key's until a copy of THIS
      // key has passed its event (copies of one stream land in order)
      hslot = 0;
      for (int k = 0; k < ipa_ctx::kTileSlowHints; k++) {
        if (!ctx->tile_slow[k].valid) { hslot = k; break; }
        if (ctx->tile_slow[k].used < ctx->tile_slow[hslot].used) hslot = k;
      }
      hint = &ctx->tile_slow[hslot];
      memcpy(hint->key, key, sizeof key);
      hint->valid = 1;
      hint->skips = 0;
      hint->launches = 0;
    }
    hint->used = ++ctx->tile_slow_clock;
    // the word is this key's count once a launch for this key has been copied back
    const bool known = hint->launches > 0 && hipEventQuery(hint->copied) == hipSuccess;
    if (known && ctx->tune.tile_warp < 2 &&
        (double)ctx->tile_slow_host[hslot] > 0.005 * (double)p.dh * p.dw && (++hint->skips & 63u) != 0)
      return 1;
    IPA_HIP(ctx, hipMemsetAsync(ctx->tile_slow_dev + hslot, 0, sizeof(unsigned), ctx->stream));
  }
  t.slow_count = ctx->tile_slow_dev + hslot;
  t.dst = p.dst; t.dst_frame_elems = p.dst_frame_elems; t.dpitch = p.dpitch;
  t.src = p.src; t.src_frame_bytes = p.src_frame_bytes; t.src_bytes = p.src_bytes;
  t.sh = p.sh; t.sw = p.sw; t.spitch = p.spitch; t.dh = p.dh; t.dw = p.dw;
  t.n_frames = n_frames;
  t.border = p.border; t.q5 = p.q5; t.cubic_a = p.cubic_a; t.lanczos = p.lanczos;
  t.cval = (float)p.cval;
  t.tiles_x = (p.dw + 63) / 64;
  t.tiles = t.tiles_x * ((p.dh + 31) / 32);
  t.frames_wg = 8;
  while (t.frames_wg > 1 && (long)t.tiles * ((n_frames + t.frames_wg - 1) / t.frames_wg) < 4096) t.frames_wg >>= 1;
  if (t.frames_wg > n_frames) t.frames_wg = n_frames;
  const size_t dbytes = ((size_t)(p.dh - 1) * p.dpitch + p.dw) * sizeof(float);
  if (dbytes >= (1ull << 31)) return 1;
  t.dst_bytes = (unsigned)dbytes;
  const unsigned groups = ((unsigned)n_frames + t.frames_wg - 1) / (unsigned)t.frames_wg;
  if ((unsigned long)t.tiles * groups >= (1ul << 31)) return 1;
  {   // (wave_grid's rule for the frame groups of a launch: a quarter of them at a time)
    int gc = ctx->tune.group_chunk;
    if (gc < 0) gc = groups >= 4 ? (int)groups / 4 : 0;
    t.group_chunk = (gc > 0 && gc < (int)groups && groups % (unsigned)gc == 0) ? gc : 0;
  }
  tile_warp_run_map(ctx->stream, t, coord, INTERP, (unsigned)t.tiles * groups,
                    (size_t)tile_warp_lds_bytes<NT>(t.pitch, t.rows));
  IPA_HIP(ctx, hipMemcpyAsync(ctx->tile_slow_host + hslot, ctx->tile_slow_dev + hslot, sizeof(unsigned),
                              hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipEventRecord(hint->copied, ctx->stream));
  hint->launches++;
  return 0;
}

template <typename Coord>
static int remap_dispatch(ipa_ctx* ctx, const RemapCall& a, const Coord& coord, int map_vec) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, a.src && a.dst, "null image pointer");
  IPA_REQUIRE(ctx, a.sh > 0 && a.sw > 0 && a.dh > 0 && a.dw > 0, "empty image (%dx%d -> %dx%d)",
              a.sh, a.sw, a.dh, a.dw);
  IPA_REQUIRE(ctx, a.spitch >= a.sw && a.dpitch >= a.dw, "pitch smaller than width");
  IPA_REQUIRE(ctx, a.spitch < (1l << 23), "source pitch must be below 2^23 elements");  // mul24
  IPA_REQUIRE(ctx, a.n_frames >= 1 && a.n_frames <= 65535, "n_frames must be in [1,65535]");
  int rc = ipa_check_interp_border(ctx, a.interp, a.border);
  if (rc) return rc;
  size_t ss = ipa_dtype_size(a.src_dt), ds = ipa_dtype_size(a.dst_dt);
  IPA_REQUIRE(ctx, ss && ds, "unknown dtype");
  size_t frame_bytes = ((size_t)(a.sh - 1) * a.spitch + a.sw) * ss;
  IPA_REQUIRE(ctx, frame_bytes < (1ull << 31), "source frame too large for 32-bit offsets");
  int base = a.interp & 0xff;

  RemapParams p;
  p.src = (const char*)a.src;
  p.dst = (char*)a.dst;
  p.src_frame_bytes = a.src_fs * (long)ss;
  p.dst_frame_elems = a.dst_fs;
  p.src_bytes = (unsigned)frame_bytes;
  p.sh = a.sh; p.sw = a.sw; p.spitch = (int)a.spitch;
  p.dh = a.dh; p.dw = a.dw; p.dpitch = a.dpitch;
  p.border = a.border;
  p.q5 = (a.interp & IPA_INTER_Q5) ? 1 : 0;
  p.cubic_a = base == IPA_INTER_CUBIC_KEYS ? -0.5f : -0.75f;
  p.lanczos = nullptr;
  p.tab2d = nullptr;
  const bool u8_cubic_tab = base == IPA_INTER_CUBIC_CV && a.src_dt == IPA_U8 && a.dst_dt == IPA_U8;
  if (u8_cubic_tab) {
    rc = ipa_u8_cubic_tab2d(ctx, &p.tab2d);
    if (rc) return rc;
  }
  const bool u8_lz_tab = base == IPA_INTER_LANCZOS4 && a.src_dt == IPA_U8 && a.dst_dt == IPA_U8 &&
                         ctx->tune.u8_lz_lds;
  if (u8_lz_tab) {
    rc = ipa_u8_lanczos_tab2d(ctx, &p.tab2d);
    if (rc) return rc;
  }
  // uint16 -> uint16 in a cv2 mode (1/32-px coordinates: linear_cv_q5, cubic_cv_q5, lanczos4):
  // OpenCV's 16U float-table arithmetic (sampler.hpp::sample_u16_cv); the exact-coordinate modes
  // keep the double sums of sample_exact
  const bool u16_cv = a.src_dt == IPA_U16 && a.dst_dt == IPA_U16 &&
                      (base == IPA_INTER_LANCZOS4 ||
                       (p.q5 && (base == IPA_INTER_LINEAR || base == IPA_INTER_CUBIC_CV)));
  if (base == IPA_INTER_LANCZOS4 || (u16_cv && base == IPA_INTER_CUBIC_CV)) {
    rc = ipa_lanczos_table(ctx, &p.lanczos);
    if (rc) return rc;
  }
  p.cval = a.cval;
  p.tiles_x = (unsigned)((a.dw + 255) / 256);
  unsigned tiles_y = (unsigned)((a.dh + 3) / 4);
  p.tiles = p.tiles_x * tiles_y;
  p.dst_vec = aligned_rows(a.dst, a.dpitch, a.dst_fs, a.n_frames, ds,
                           4 * ds > IPA_VEC_ALIGN ? (size_t)IPA_VEC_ALIGN : 4 * ds);
  p.map_vec = map_vec;
  // frame index fastest only where frames share data and the gathers are light: map-based
  // nearest / bilinear (16 x 4K: nearest 281 -> 260 us, bilinear level; bicubic level,
  // Lanczos4 and the analytic sources 5 % slower)
  bool inner = a.n_frames > 1 && (unsigned long)p.tiles * a.n_frames < (1ul << 31) &&
               std::is_same<Coord, MapCoord>::value &&
               (base == IPA_INTER_NEAREST || base == IPA_INTER_LINEAR);
  inner = inner && ctx->tune.frames_inner != 0;  // tuning knob
  p.frames_inner = inner ? a.n_frames : 0;
  dim3 grid = inner ? dim3(p.tiles * (unsigned)a.n_frames, 1) : dim3(p.tiles, (unsigned)a.n_frames);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  p.skip = nullptr;
  p.tile_rows = 1;
  if (u8_cubic_tab) {
    // the 32 KB table is staged per workgroup: 16 row groups (64 rows x 256 px) each
    p.tile_rows = 16;
    p.tiles = p.tiles_x * ((tiles_y + p.tile_rows - 1) / p.tile_rows);
    grid = inner ? dim3(p.tiles * (unsigned)a.n_frames, 1) : dim3(p.tiles, (unsigned)a.n_frames);
  }
  // batches of float32 frames: the clean strips on the ring kernel (taps from LDS), the rest
  // below behind the skip mask
  // (ring_remap = 1: where it measured faster - 16 x 4K frames, gather -> ring + rest (+ plan):
  //   Lanczos4  maps 1156 -> 712 us, lens model 1111 -> 676, homography 1112 -> 707
  //   bicubic   maps  477 -> 378,    lens model  461 -> 356, homography  464 -> 423
  //   bilinear  maps  351 -> 297,    lens model  343 -> 261, homography  344 -> 355 (stays)
  // the lens model and the homography read the coordinates the planning pass stored; the
  // homography's are doubles, 16 B per pixel and frame.  ring_remap = 2: every covered case)
  constexpr bool kHom = std::is_same<Coord, HomographyCoord>::value;
  if constexpr (kHom) {
    // uint16 frames in a cv2 mode (bicubic / Lanczos4 at 1/32-px coordinates: what
    // PerspectiveCorrection.correct does to the camera's frames)
    if (ctx->tune.tile_warp && u16_cv && (base == IPA_INTER_LANCZOS4 || base == IPA_INTER_CUBIC_CV) &&
        (unsigned long)p.tiles * a.n_frames < (1ul << 30)) {
      int trc = base == IPA_INTER_LANCZOS4
                    ? tile_warp_launch<kLanczos4, uint16_t>(ctx, p, coord, a.n_frames, base)
                    : tile_warp_launch<kCubic, uint16_t>(ctx, p, coord, a.n_frames, base);
      if (trc < 0) return trc;
      if (trc == 0) {
        IPA_HIP(ctx, hipGetLastError());
        return IPA_OK;
      }
    }
    if (ctx->tune.tile_warp && a.src_dt == IPA_F32 && a.dst_dt == IPA_F32 &&
        base != IPA_INTER_NEAREST && (unsigned long)p.tiles * a.n_frames < (1ul << 30)) {
      int trc = base == IPA_INTER_LINEAR ? tile_warp_launch<kLinear>(ctx, p, coord, a.n_frames, base)
                : base == IPA_INTER_LANCZOS4 ? tile_warp_launch<kLanczos4>(ctx, p, coord, a.n_frames, base)
                                             : tile_warp_launch<kCubic>(ctx, p, coord, a.n_frames, base);
      if (trc == 0) {
        IPA_HIP(ctx, hipGetLastError());
        return IPA_OK;
      }
    }
  }
  // ... and from which batch size (4K frames, ring against gather kernel, profiles/r03_micro.txt):
  // a map pair is planned anew on every call (its contents may have changed) - ~50 us that 2
  // bilinear frames do not earn back (79 against 30 us; level at 16) -, a source given by value
  // only on its first call
  constexpr bool kMapSrc = std::is_same<Coord, MapCoord>::value;
  if constexpr (kMapSrc) {
    // map remaps of float32 frames on the tile kernel (knob tile_warp; 2: always).  16 x 4K lens
    // maps, ring + gather kernels -> tile kernel: bilinear 0.279 -> 0.228 ms, bicubic 0.367 -> 0.318,
    // Lanczos4 0.637 -> 0.617; with the reference's alpha = 1 maps (a rim outside the source)
    // 0.314 / 0.406 / 0.851 -> 0.313 / 0.347 / 0.694.  Batches of 4: only Lanczos4 gains
    // (0.193 -> 0.182, single frames 0.068 -> 0.060); bilinear / bicubic from 8 frames and 64 Mpx
    const double work = (double)a.n_frames * a.dh * a.dw;
    const bool pays = base == IPA_INTER_LANCZOS4 ? work >= 8e6 : (a.n_frames >= 8 && work >= 64e6);
    if (ctx->tune.tile_warp && a.src_dt == IPA_F32 && a.dst_dt == IPA_F32 && base != IPA_INTER_NEAREST &&
        (ctx->tune.tile_warp > 1 || pays)) {
      int trc = base == IPA_INTER_LINEAR ? tile_warp_launch_map<kLinear>(ctx, p, coord, a.n_frames)
                : base == IPA_INTER_LANCZOS4 ? tile_warp_launch_map<kLanczos4>(ctx, p, coord, a.n_frames)
                                             : tile_warp_launch_map<kCubic>(ctx, p, coord, a.n_frames);
      if (trc == 0) {
        IPA_HIP(ctx, hipGetLastError());
        return IPA_OK;
      }
    }
  }
  const int ring_from = base == IPA_INTER_LINEAR ? (kMapSrc ? 16 : 4)
                        : base == IPA_INTER_LANCZOS4 ? (kMapSrc ? 3 : 2)
                                                     : (kMapSrc ? 8 : 4);
  const bool ring_pays = !(base == IPA_INTER_LINEAR && kHom) && a.n_frames >= ring_from;
  if ((ctx->tune.ring_remap > 1 || (ctx->tune.ring_remap == 1 && ring_pays)) &&
      a.n_frames >= ctx->tune.ring_min && a.src_dt == IPA_F32 && a.dst_dt == IPA_F32 &&
      base != IPA_INTER_NEAREST) {
    rc = ring_remap_launch<Coord>(ctx, p, coord, base, a.n_frames);
    if (rc < 0) return rc;
    if (p.skip) {
      p.tiles = p.tiles_x * ((tiles_y + p.tile_rows - 1) / p.tile_rows);
      grid = inner ? dim3(p.tiles * (unsigned)a.n_frames, 1) : dim3(p.tiles, (unsigned)a.n_frames);
    }
  }

  if (u8_lz_tab) {
    // 16 rows per pass and workgroup, 8 passes: 128 rows x 256 px per staging of the table
    p.tile_rows = 8;
    p.frames_inner = a.n_frames;
    const unsigned bands = ((unsigned)a.dh + 16u * p.tile_rows - 1u) / (16u * p.tile_rows);
    hipLaunchKernelGGL((remap_u8_lz_kernel<Coord>), dim3(p.tiles_x * bands * (unsigned)a.n_frames),
                       dim3(1024), 0, ctx->stream, p, coord);
    IPA_HIP(ctx, hipGetLastError());
    return IPA_OK;
  }
  if constexpr (!coord_is_table<Coord>::value) {
    const int smin = ctx->tune.stored_coords;
    if (!p.skip && smin > 0 && a.n_frames >= smin && a.src_dt == IPA_F32 && a.dst_dt == IPA_F32 &&
        (base == IPA_INTER_CUBIC_CV || base == IPA_INTER_CUBIC_KEYS || base == IPA_INTER_LANCZOS4)) {
      StoredCoord<typename Coord::coord_t> sc;
      rc = stored_coords_prepare<Coord>(ctx, coord, a.dh, a.dw, &sc);
      if (rc < 0) return rc;
      if (rc == 0) {
        RemapParams q = p;
        q.map_vec = 1;   // rows of dw coordinates, 256-byte aligned planes
        launch_interp<float, float, StoredCoord<typename Coord::coord_t>, false>(ctx, q, sc, base, grid);
        IPA_HIP(ctx, hipGetLastError());
        return IPA_OK;
      }
    }
  }
  int s = a.src_dt, d = a.dst_dt;
  if (p.skip) {
    launch_rest<Coord>(ctx, p, coord, base, grid);
  } else if (s == IPA_F32 && d == IPA_F32) {
    launch_interp<float, float, Coord, false>(ctx, p, coord, base, grid);
  } else if (s == IPA_F64 && d == IPA_F64) {
    launch_interp<double, double, Coord, false>(ctx, p, coord, base, grid);
  } else if (s == IPA_U16 && d == IPA_F32) {
    launch_interp<uint16_t, float, Coord, false>(ctx, p, coord, base, grid);
  } else if (s == IPA_U16 && d == IPA_U16) {
    if (u16_cv) launch_interp<uint16_t, uint16_t, Coord, true>(ctx, p, coord, base, grid);
    else launch_interp<uint16_t, uint16_t, Coord, false>(ctx, p, coord, base, grid);
  } else if (s == IPA_U8 && d == IPA_F32) {
    launch_interp<uint8_t, float, Coord, false>(ctx, p, coord, base, grid);
  } else if (s == IPA_U8 && d == IPA_U8) {
    launch_interp<uint8_t, uint8_t, Coord, true>(ctx, p, coord, base, grid);
  } else if (s == IPA_F32 && d == IPA_U8) {
    launch_interp<float, uint8_t, Coord, false>(ctx, p, coord, base, grid);
  } else if (s == IPA_F32 && d == IPA_U16) {
    launch_interp<float, uint16_t, Coord, false>(ctx, p, coord, base, grid);
  } else {
    IPA_UNSUPPORTED(ctx, "remap: src dtype %d -> dst dtype %d not supported", s, d);
  }
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

