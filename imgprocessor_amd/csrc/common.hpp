// common.hpp — context, error plumbing and device helpers shared by the gfx950 kernels.
// Written for MI355X (CDNA4, wave64) only; no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include <mutex>
#include <string>

#include "../../include/imgproc_hip.h"

// Byte alignment of rows and frames that the 16-byte row loads / stores of the fast paths ask for.
// 4: gfx950 serves a global dwordx4 at any dword address (the HSA ABI runs the memory unit in its
// unaligned mode), so float32 images of ANY width take the fast paths - until round 3 they had to
// be 16-byte aligned, and a width that is no multiple of 4 cost 1.4 - 1.7 x (2160 x 3838: fused
// undistort + 5x5 550 -> 429 us, plain 5x5 370 -> 252; the rest is the last partial lane per row).
#ifndef IPA_VEC_ALIGN
#define IPA_VEC_ALIGN 4
#endif

// Launch-shape knobs of a context.  Defaults are the measured optima;
// ipa_ctx_create() reads the IPA_* environment variables ONCE into this struct and
// ipa_ctx_set_tuning() changes a field afterwards - nothing in a launch path calls getenv().
struct ipa_tuning {
  int strip_h = 0;        // rows per strip of the marching kernels (0: by launch size)
  int frames_inner = 1;   // batches dispatched strip by strip (frames of a strip adjacent)
  int frames_wg = 1;      // map-based fused kernels: the waves of a workgroup are frames of ONE strip
  int pipe = 1;           // 0: no strip on the hand-scheduled loops of wave_pipe.hpp (hand-counted vmcnt
                          // waits) - the compiler-scheduled chunked loops everywhere: the fallback
                          // and cross-check of that scheme (same bits, slower)
  int group_chunk = -1;   // batches on the shared-record loop: frame groups walked this many at a time (-1: a
                          // quarter of them, 0: all groups of a strip together); see wave_grid
#ifndef IPA_WITH_TILE_CHAIN
#define IPA_WITH_TILE_CHAIN 0   // 1: experiment builds that carry tools/tile_chain (make TILE_CHAIN=1); not the product
#endif
#if IPA_WITH_TILE_CHAIN
  int tile_chain = 0;     // 1: perspective warp + separable filter in ONE launch on the tile skeleton (tile_chain.hpp)
                          // for the chains that take two launches (bicubic; bilinear under a rotation); 2: wherever
                          // that kernel covers the chain.  Off: built, bit-identical to the two launches through the
                          // workspace, and 25 - 50 % slower (profiles/r05_micro.txt)
  int chain_steps = 0;    // ... steps of 32 rows a workgroup walks down its column (0: 4)
  int chain_frames = 0;   // ... frames per workgroup (0: up to 8, by launch size)
#endif
  int pipe7 = 1;          // 7x7 after a bilinear map remap of a batch: resident coefficients on the shared-map loop
  int frame_major = 1;    // kernels whose frames share nothing (plain filters): frame after frame, every
                          // XCD streaming through frames of its own
  int big_wave = 1;       // 9x9 / 11x11 filter on the marching wave (0: LDS-tiled kernel)
  int big_fused = 1;      // remap -> 7x7 / 9x9 / 11x11 in one kernel
  int stream_k = 7;       // smallest K whose coefficients are streamed through SGPRs
  int ring_min = 2;       // smallest batch the ring remap kernel takes
  int u8_lz_lds = 1;      // uint8 Lanczos4: OpenCV's 128 KB weight table in LDS (0: weights formed per sample)
  int lens_cache = 1;     // fused undistort + filter: lens model evaluated once per (K, dist, newK, size)
  int ring_remap = 1;     // standalone remap of batches on the ring kernel: 1 where it pays, 2 always
  int tile_warp = 1;      // perspective warps of float32 frames with the tile's source box in LDS (tile_warp.hpp):
                          // 0 never, 1 where it pays, 2 whenever the homography is covered
  int tail_rows = -1;     // chunked batches on the shared-record loop: every XCD's share of the launch ends on strips of this
                          // many rows (-1: a quarter of the strip height, at least 24; 0: uniform strips) - the workgroups
                          // that run while the launch drains (WaveParams::seg_count)
  int rank1_sep = 3;      // dense K x K kernels that are an exact outer product ky (x) kx (how the reference obtains its
                          // Gaussians: scipy.ndimage.gaussian_filter, filters/fastFilter.py:42) run on the separable K + K
                          // loops wherever those cover the call: bit 0 the remap -> filter chains, bit 1 the plain filter
  int strip_remap = 1;    // bilinear remaps of uint16 frames into float32 on the marching strips of the chains, no filter
                          // (remap.hip::strip_remap_takes; 0: the gather kernels of rounds 1 - 5)
  int sep_u16 = 1;        // integer frames (uint16: maps, homographies; uint8: maps): bilinear remap -> separable 3 / 5 / 7 / 9-tap filter on uint16 frames in ONE kernel
                          // (float32 frames always were; 0: two launches through the workspace, as in rounds 1 - 5)
  int stored_coords = 4;  // bicubic / Lanczos4 remaps of >= this many float32 frames from a coordinate source given
                          // by value (homography, lens model) that the ring kernel does not take: the coordinates
                          // are evaluated ONCE into the plan buffer and the gather kernel reads them (0: never)
};

struct ipa_ctx {
  ipa_tuning tune;
  int device = -1;
  hipStream_t stream = nullptr;

  int cu_count = 0;
  std::string last_error;
  // grow-only device workspace used by the host-pointer entry points
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // small device scratch for per-call tables (IDW weights, Lanczos table, ...)
  void* tab = nullptr;
  size_t tab_bytes = 0;
  size_t tab_valid = 0;        // bytes of the last upload (the same table again is not re-sent)
  unsigned long tab_serial = 0;   // counts uploads: a caller's tables are still there while it stands
  long resize_key[5] = {0, 0, 0, 0, 0};   // sw, dw, sh, dh, interp of the resize tables in `tab`
  unsigned long resize_serial = 0;
  int resize_xmax = 0;
  void* tab_pinned = nullptr;  // pinned staging so the H2D is truly stream-ordered
  // per-call strip plans of the ring kernels (device only, stream-ordered reuse)
  void* plan = nullptr;
  size_t plan_bytes = 0;
  // what the plan buffer holds when it can be reused by the next call (coordinate sources given
  // by value: lens model, homography): the source's parameters + geometry; n = 0: nothing.
  // ipa_plan_reserve() clears it, the user that wants reuse sets it after filling the buffer.
  double plan_key[40];
  int plan_key_n = 0;
  // the LDS boxes of the tile warp kernel (tile_warp.hpp) for the last few homographies +
  // geometries (least recently used of kTileWarpPlans replaced): a host-side walk over the tiles
  // that a repeated call - or two matrices used in turn - does not pay again (0.4 ms per 4K call)
  static constexpr int kTileWarpPlans = 4;
  struct TileWarpPlan {
    double key[14];
    int valid = 0, pitch = 0, rows = 0, ok = 0, shape = 0;
    double drift = 0, step = 0, fetch = 0;   // see tile_warp_pays()
    unsigned long used = 0;
  } tile_warp_plans[kTileWarpPlans];
  unsigned long tile_warp_clock = 0;
  unsigned long strip_remaps = 0;     // standalone uint16 -> float32 remaps the strip kernel took (remap.hip; read like rank1_routed)
  unsigned long rank1_routed = 0;     // dense calls sent to the separable loops so far (read through ipa_ctx_get_tuning)
  int tail_rows_used = 0;             // height of the short strips of the last such launch (0: uniform strips)
  int group_chunk_used = 0;           // groups per chunk of the last launch on the shared-record loop (0: all together)
#if IPA_WITH_TILE_CHAIN
  unsigned long chain_launches = 0;   // launches of tile_chain.hpp's kernel (read through ipa_ctx_get_tuning: the tests' evidence of the path taken)
#endif
  // map remaps on the tile kernel: per (map pair, geometry) - least recently used of kTileSlowHints
  // replaced - a device word the kernel counts its tap-by-tap pixels in, the page-locked word it is
  // read back to by an asynchronous copy, and the event behind that copy: the word is the hint of
  // ITS key once the event has passed (the host never writes it)
  static constexpr int kTileSlowHints = 4;
  struct TileSlowHint {
    double key[10];
    int valid = 0;
    unsigned skips = 0;
    unsigned long used = 0, launches = 0;
    hipEvent_t copied = nullptr;
  } tile_slow[kTileSlowHints];
  unsigned long tile_slow_clock = 0;
  unsigned* tile_slow_dev = nullptr;    // kTileSlowHints words
  unsigned* tile_slow_host = nullptr;   // kTileSlowHints words
  // clean strip pairs / pairs of the last planning pass (page-locked, written by an async copy)
  // and the source + geometry it belongs to: ring_plan_prepare's hint
  unsigned* ring_hint = nullptr;
  double ring_hint_key[40];
  int ring_hint_n = 0;
  unsigned ring_hint_skips = 0;   // calls that skipped the planning pass on the hint's word
  // float32 coordinate maps of the last lens model a fused undistort + filter call was made
  // with (what LensDistortion.getUndistortRectifyMap caches, camera/LensDistortion.py:344-345):
  // the next call with the same K, distortion, newK and size reads them instead of evaluating
  // the model per pixel and frame
  void* lens_map = nullptr;
  size_t lens_map_bytes = 0;
  double lens_key[25];
  int lens_key_n = 0;
  // No lock here: a context (stream + workspaces) belongs to ONE host thread at a time
  // (INTEGRATION.md section 4); the Python layer hands every thread its own default context.
};

struct ipa_event {
  hipEvent_t ev;
};

void ipa_set_error(ipa_ctx* ctx, const char* fmt, ...);
int ipa_ws_reserve(ipa_ctx* ctx, size_t bytes);                         // ctx->ws >= bytes
int ipa_lens_map_cached(ipa_ctx* ctx, const double* K, const double* dist5, const double* newK,
                        int h, int w, float** mx, float** my);
// remap.hip -> fused.hip: the strip remap of integer frames into their own type (1: not a call it covers)
int ipa_strip_remap_int(ipa_ctx* ctx, int dtype, const void* d_src, int sh, int sw, long src_pitch, const float* d_mapx,
                        const float* d_mapy, long map_pitch, void* d_dst, int dh, int dw, long dst_pitch, int n_frames,
                        long src_frame_stride, long dst_frame_stride, int interp, int border_mode, double border_value);
int ipa_plan_reserve(ipa_ctx* ctx, size_t bytes);                       // ctx->plan >= bytes
int ipa_tab_upload(ipa_ctx* ctx, const void* host, size_t bytes, void** d);  // stream-ordered

#define IPA_HIP(ctx, call)                                                            \
  do {                                                                                \
    hipError_t e__ = (call);                                                          \
    if (e__ != hipSuccess) {                                                          \
      ipa_set_error(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),      \
                    __FILE__, __LINE__);                                              \
      return e__ == hipErrorOutOfMemory ? IPA_ERR_OOM : IPA_ERR_HIP;                  \
    }                                                                                 \
  } while (0)

#define IPA_REQUIRE(ctx, cond, ...)          \
  do {                                       \
    if (!(cond)) {                           \
      ipa_set_error(ctx, __VA_ARGS__);       \
      return IPA_ERR_BAD_ARG;                \
    }                                        \
  } while (0)

#define IPA_UNSUPPORTED(ctx, ...)            \
  do {                                       \
    ipa_set_error(ctx, __VA_ARGS__);         \
    return IPA_ERR_UNSUPPORTED;              \
  } while (0)

// Is the kh x kw kernel an outer product ky (x) kx to within 1e-12 of its largest entry (five orders below what a
// float32 coefficient resolves)?  Pivot on that entry: ky = its column, kx = its row / pivot.  Zero and non-finite
// kernels are not.  Host code; the dense and the separable loops then differ by summation order only.
static inline bool ipa_rank1_factor(const double* k, int kh, int kw, double* ky, double* kx) {
  int i0 = 0, j0 = 0;
  double piv = 0.0;
  for (int i = 0; i < kh; i++)
    for (int j = 0; j < kw; j++) {
      const double v = k[i * kw + j];
      if (!(v - v == 0.0)) return false;  // inf / nan
      if ((v < 0 ? -v : v) > (piv < 0 ? -piv : piv)) { piv = v; i0 = i; j0 = j; }
    }
  if (piv == 0.0) return false;
  for (int i = 0; i < kh; i++) ky[i] = k[i * kw + j0];
  for (int j = 0; j < kw; j++) kx[j] = k[i0 * kw + j] / piv;
  const double tol = 1e-12 * (piv < 0 ? -piv : piv);
  for (int i = 0; i < kh; i++)
    for (int j = 0; j < kw; j++) {
      const double d = ky[i] * kx[j] - k[i * kw + j];
      if ((d < 0 ? -d : d) > tol) return false;
    }
  return true;
}

static inline size_t ipa_dtype_size(int dt) {
  switch (dt) {
    case IPA_U8: return 1;
    case IPA_U16: return 2;
    case IPA_F32: return 4;
    case IPA_F64: return 8;
  }
  return 0;
}

// -------------------------------------------------------------------------
// device helpers
// -------------------------------------------------------------------------
namespace ipa {

constexpr int kWave = 64;
constexpr int kXcds = 8;  // MI355X: 8 XCDs, blocks are dealt round-robin over them

// Bijective XCD-aware remap of a linear workgroup id: workgroups that land on
// the same XCD (id % 8) get a CONTIGUOUS range of tiles, so tiles that share
// halo / source rows share that XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned nwg) {
  unsigned q = nwg / kXcds, r = nwg % kXcds;
  unsigned k = bid % kXcds, j = bid / kXcds;
  unsigned start = k * q + (k < r ? k : r);
  return start + j;
}

// index resolution shared by every border-aware load; returns -1 for "cval"
__device__ __forceinline__ int resolve_idx(int i, int n, int mode) {
  if ((unsigned)i < (unsigned)n) return i;
  switch (mode) {
    case IPA_BORDER_REPLICATE: return i < 0 ? 0 : n - 1;
    case IPA_BORDER_REFLECT: {
      if (n == 1) return 0;
      int p = 2 * n;
      int m = i % p;
      if (m < 0) m += p;
      return m < n ? m : p - 1 - m;
    }
    case IPA_BORDER_REFLECT101: {
      if (n == 1) return 0;
      int p = 2 * n - 2;
      int m = i % p;
      if (m < 0) m += p;
      return m < n ? m : p - m;
    }
    case IPA_BORDER_WRAP: {
      int m = i % n;
      if (m < 0) m += n;
      return m;
    }
    default: return -1;
  }
}

__device__ __forceinline__ float ipa_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double ipa_fma(double a, double b, double c) { return fma(a, b, c); }

template <typename T> struct compute_of { using type = float; };
template <> struct compute_of<double> { using type = double; };

// cv::saturate_cast semantics for integer outputs: round-half-even then clamp
template <typename DT, typename CT> __device__ __forceinline__ DT store_cast(CT v);
template <> __device__ __forceinline__ float store_cast<float, float>(float v) { return v; }
template <> __device__ __forceinline__ double store_cast<double, double>(double v) { return v; }
template <> __device__ __forceinline__ float store_cast<float, double>(double v) { return (float)v; }
template <> __device__ __forceinline__ uint8_t store_cast<uint8_t, float>(float v) {
  float r = rintf(v);
  r = r > 0.f ? r : 0.f;  // NaN -> 0
  r = r < 255.f ? r : 255.f;
  return (uint8_t)r;
}
template <> __device__ __forceinline__ uint16_t store_cast<uint16_t, float>(float v) {
  float r = rintf(v);
  r = r > 0.f ? r : 0.f;
  r = r < 65535.f ? r : 65535.f;
  return (uint16_t)r;
}
// integer destinations computed in double (sample_exact): cv::saturate_cast rounding
template <> __device__ __forceinline__ uint8_t store_cast<uint8_t, double>(double v) {
  double r = rint(v);
  r = r > 0.0 ? r : 0.0;  // NaN -> 0
  r = r < 255.0 ? r : 255.0;
  return (uint8_t)r;
}
template <> __device__ __forceinline__ uint16_t store_cast<uint16_t, double>(double v) {
  double r = rint(v);
  r = r > 0.0 ? r : 0.0;
  r = r < 65535.0 ? r : 65535.0;
  return (uint16_t)r;
}

}  // namespace ipa
