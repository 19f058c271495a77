// conv.hip — the filters/ half of the hot path on gfx950:
//   ipa_conv2d*     dense KH x KW centred correlation with per-axis border mode and
//                   optional mask (filters/maskedConvolve.py:24-43 +
//                   filters/_extendArrayForConvolution.py:5-97; scipy.ndimage.correlate)
//   ipa_sepconv2d*  separable correlation in scipy.ndimage.gaussian_filter order
//                   (filters/standardDeviation.py:23, filters/fastFilter.py:42)
//
// Both are single-pass over HBM (8 B/px for f32): the input tile incl. halo is
// staged once in LDS with 16-byte loads, border pixels are resolved while
// staging (no padded copy like extendArrayForConvolution makes), the stencil
// runs out of LDS/registers.  See conv_tile.hpp for the tile geometry.
#include <vector>

#include "common.hpp"
#include "conv_tile.hpp"
#include "wave_stencil.hpp"

// float32 K x K filter on the wave-marching skeleton (instantiated per K in fused_k*.hip)
int ipa_wave_conv_launch_k3(ipa_ctx*, const ipa::WaveParams&, const ipa::LoadRowSrc&, const double*, int);
int ipa_wave_conv_launch_k5(ipa_ctx*, const ipa::WaveParams&, const ipa::LoadRowSrc&, const double*, int);
int ipa_wave_conv_launch_k7(ipa_ctx*, const ipa::WaveParams&, const ipa::LoadRowSrc&, const double*, int);
int ipa_wave_conv_launch_k9(ipa_ctx*, const ipa::WaveParams&, const ipa::LoadRowSrc&, const double*, int);
int ipa_wave_conv_launch_k11(ipa_ctx*, const ipa::WaveParams&, const ipa::LoadRowSrc&, const double*, int);
// float32 separable K+K filter on the same skeleton (wave_sep.hip); returns 1 if not covered
int ipa_wave_sep_launch(ipa_ctx*, const ipa::WaveParams&, const ipa::LoadRowSrc&, const double* ky,
                        int nky, const double* kx, int nkx, int n_frames, float xcval);

namespace ipa {

struct ConvParams {
  const char* src;
  char* dst;
  const uint8_t* mask;
  long mask_pitch;
  long src_frame_elems, dst_frame_elems;
  int h, w;
  long spitch, dpitch;
  int bx, by;
  double cval;
  unsigned tiles_x, tiles;
  int vec_in, vec_out;
};

// stage rows [y0-KH/2, ...) x cols [x0-HX, x0+128+HX) of the frame into LDS
#ifndef IPA_RANK1_PLAIN_MIN
#define IPA_RANK1_PLAIN_MIN 9
#endif

template <typename T, int KH, int KW>
__device__ __forceinline__ void fill_tile_global(T* __restrict__ tile, const T* __restrict__ src,
                                                 const ConvParams& p, int x0, int y0) {
  using G = conv_geom<KW>;
  using V = typename vec16<T>::type;
  constexpr int VN = vec16<T>::n;
  constexpr int ROWS = kTileH + KH - 1;
  constexpr int CHUNKS = G::LW / VN;
  const int tid = threadIdx.y * 32 + threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const T cval = (T)p.cval;
  // One tile row per wave per trip; the trips are unrolled and the LDS writes follow the loads
  // of all rows, so the scheduler keeps several row loads in flight (9x9: 436 -> 386 us for
  // 16 4K frames).  Measured slower: a branch-free variant with every load issued strictly up
  // front (681 us), and dealing the tile's chunks linearly over all 256 lanes (421 vs 312 us,
  // 5x5).
  constexpr int NR = (ROWS + 3) / 4;          // rows per wave
  constexpr int NC = (CHUNKS + 63) / 64;      // 16-byte chunks per lane and row
  V v[NR][NC];
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int lr = wave + 4 * q;
    if (lr >= ROWS) break;
    const int yy = resolve_idx(y0 - KH / 2 + lr, p.h, p.by);
    const T* srow = src + (long)(yy < 0 ? 0 : yy) * p.spitch;
#pragma unroll
    for (int cc = 0; cc < NC; cc++) {
      const int c = lane + 64 * cc;
      if (c >= CHUNKS) break;
      const int gx = x0 - G::HX + c * VN;
      if (yy >= 0 && p.vec_in && gx >= 0 && gx + VN <= p.w) {
        v[q][cc] = *reinterpret_cast<const V*>(srow + gx);
      } else {
        T e[VN];
#pragma unroll
        for (int k = 0; k < VN; k++) {
          int xx = resolve_idx(gx + k, p.w, p.bx);
          e[k] = (yy < 0 || xx < 0) ? cval : srow[xx];
        }
        if constexpr (VN == 4) v[q][cc] = V{e[0], e[1], e[2], e[3]};
        else v[q][cc] = V{e[0], e[1]};
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int lr = wave + 4 * q;
    if (lr >= ROWS) break;
#pragma unroll
    for (int cc = 0; cc < NC; cc++) {
      const int c = lane + 64 * cc;
      if (c >= CHUNKS) break;
      *reinterpret_cast<V*>(tile + lr * G::LW + c * VN) = v[q][cc];
    }
  }
}

template <typename T, int KH, int KW>
__global__ void __launch_bounds__(256) conv_kernel(ConvParams p, Weights<T, KH * KW> wts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);
  unsigned t = xcd_swizzle(blockIdx.x, p.tiles);
  unsigned tyi = t / p.tiles_x, txi = t - tyi * p.tiles_x;
  int x0 = (int)txi * kTileW, y0 = (int)tyi * kTileH;
  unsigned frame = blockIdx.y;
  const T* src = reinterpret_cast<const T*>(p.src) + (long)frame * p.src_frame_elems;
  T* dst = reinterpret_cast<T*>(p.dst) + (long)frame * p.dst_frame_elems;

  fill_tile_global<T, KH, KW>(tile, src, p, x0, y0);
  __syncthreads();

  T acc[4][4];
  conv_from_lds<T, KH, KW>(tile, threadIdx.x, threadIdx.y, wts, acc);

  int ox = x0 + threadIdx.x * 4;
#pragma unroll
  for (int oy = 0; oy < 4; oy++) {
    int y = y0 + threadIdx.y * 4 + oy;
    if (y >= p.h || ox >= p.w) continue;
    int n = p.w - ox < 4 ? p.w - ox : 4;
    if (p.mask) {
      const uint8_t* m = p.mask + (long)y * p.mask_pitch + ox;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < n && !m[k]) acc[oy][k] = (T)0;
    }
    T* row = dst + (long)y * p.dpitch + ox;
    if (p.vec_out && n == 4) {
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(row) = float4{acc[oy][0], acc[oy][1], acc[oy][2], acc[oy][3]};
      } else {
        reinterpret_cast<double2*>(row)[0] = double2{acc[oy][0], acc[oy][1]};
        reinterpret_cast<double2*>(row)[1] = double2{acc[oy][2], acc[oy][3]};
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < n) row[k] = acc[oy][k];
    }
  }
}

// Any kernel shape (rectangular, even sizes, > 11): weights in device memory,
// one output pixel per lane, taps through L1/L2.  Completeness path, not the
// tuned one.
template <typename T>
__global__ void __launch_bounds__(256)
conv_generic_kernel(ConvParams p, const T* __restrict__ wts, int kh, int kw) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= p.w || y >= p.h) return;
  unsigned frame = blockIdx.z;
  const T* src = reinterpret_cast<const T*>(p.src) + (long)frame * p.src_frame_elems;
  T* dst = reinterpret_cast<T*>(p.dst) + (long)frame * p.dst_frame_elems;
  T out = (T)0;
  if (!p.mask || p.mask[(long)y * p.mask_pitch + x]) {
    const T cval = (T)p.cval;
    for (int i = 0; i < kh; i++) {
      int yy = resolve_idx(y + i - kh / 2, p.h, p.by);
      for (int j = 0; j < kw; j++) {
        int xx = resolve_idx(x + j - kw / 2, p.w, p.bx);
        T v = (yy < 0 || xx < 0) ? cval : src[(long)yy * p.spitch + xx];
        out = ipa_fma(wts[i * kw + j], v, out);
      }
    }
  }
  dst[(long)y * p.dpitch + x] = out;
}

// ------------------------------------------------------------- separable --
constexpr int kSepMaxTaps = 63;

template <typename T> struct SepWeights {
  T ky[kSepMaxTaps];
  T kx[kSepMaxTaps];
};

struct SepParams {
  const char* src;
  char* dst;
  long src_frame_elems, dst_frame_elems;
  int h, w;
  long spitch, dpitch;
  int bx, by;
  double cval;
  unsigned tiles_x, tiles;
  int nky, nkx;   // 0 = skip axis
  int hy, hx;     // nky/2, nkx/2 (0 when skipped)
  int hxa;        // hx rounded up to 4
  int vec_in, vec_out;
};

// One workgroup: 128 x 32 output tile.  LDS A: (32+nky-1) x (128+2*hxa) input,
// LDS B: 32 x (128+2*hxa) after the y pass (values rounded to T exactly like
// scipy's per-axis intermediate array), then the x pass writes the output.
template <typename T>
__global__ void __launch_bounds__(256) sepconv_kernel(SepParams p, SepWeights<T> wts) {
  using V = typename vec16<T>::type;
  constexpr int VN = vec16<T>::n;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int LW = kTileW + 2 * p.hxa;
  const int rowsA = kTileH + 2 * p.hy;
  T* A = reinterpret_cast<T*>(smem);
  T* B = A + rowsA * LW;
  unsigned t = xcd_swizzle(blockIdx.x, p.tiles);
  unsigned tyi = t / p.tiles_x, txi = t - tyi * p.tiles_x;
  int x0 = (int)txi * kTileW, y0 = (int)tyi * kTileH;
  unsigned frame = blockIdx.y;
  const T* src = reinterpret_cast<const T*>(p.src) + (long)frame * p.src_frame_elems;
  T* dst = reinterpret_cast<T*>(p.dst) + (long)frame * p.dst_frame_elems;
  const int tid = threadIdx.y * 32 + threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const T cval = (T)p.cval;
  const int chunks = LW / VN;

  // stage input (columns outside the image are resolved with bx NOW; rows with by)
  for (int lr = wave; lr < rowsA; lr += 4) {
    int yy = resolve_idx(y0 - p.hy + lr, p.h, p.by);
    const T* srow = src + (long)(yy < 0 ? 0 : yy) * p.spitch;
    for (int c = lane; c < chunks; c += 64) {
      int gx = x0 - p.hxa + c * VN;
      V v;
      if (yy >= 0 && p.vec_in && gx >= 0 && gx + VN <= p.w) {
        v = *reinterpret_cast<const V*>(srow + gx);
      } else {
        T e[VN];
#pragma unroll
        for (int k = 0; k < VN; k++) {
          int xx = resolve_idx(gx + k, p.w, p.bx);
          // a constant-border COLUMN must stay cval after the y pass too; rows
          // outside are cval only for the y pass.  Both handled by value here
          // because sum(ky)*cval is what scipy computes for such a column.
          e[k] = (yy < 0 || xx < 0) ? cval : srow[xx];
        }
        if constexpr (VN == 4) v = V{e[0], e[1], e[2], e[3]};
        else v = V{e[0], e[1]};
      }
      *reinterpret_cast<V*>(A + lr * LW + c * VN) = v;
    }
  }
  __syncthreads();

  // y pass: B[r][c] = sum_i ky[i] * A[r+i][c]
  for (int r = wave; r < kTileH; r += 4) {
    for (int c = lane; c < chunks; c += 64) {
      T acc[VN];
#pragma unroll
      for (int k = 0; k < VN; k++) acc[k] = (T)0;
      if (p.nky > 0) {
        for (int i = 0; i < p.nky; i++) {
          V v = *reinterpret_cast<const V*>(A + (r + i) * LW + c * VN);
          T w = wts.ky[i];
          if constexpr (VN == 4) {
            acc[0] = ipa_fma(w, v.x, acc[0]); acc[1] = ipa_fma(w, v.y, acc[1]);
            acc[2] = ipa_fma(w, v.z, acc[2]); acc[3] = ipa_fma(w, v.w, acc[3]);
          } else {
            acc[0] = ipa_fma(w, v.x, acc[0]); acc[1] = ipa_fma(w, v.y, acc[1]);
          }
        }
      } else {
        V v = *reinterpret_cast<const V*>(A + r * LW + c * VN);
        if constexpr (VN == 4) { acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w; }
        else { acc[0] = v.x; acc[1] = v.y; }
      }
      V o;
      if constexpr (VN == 4) o = V{acc[0], acc[1], acc[2], acc[3]};
      else o = V{acc[0], acc[1]};
      *reinterpret_cast<V*>(B + r * LW + c * VN) = o;
    }
  }
  __syncthreads();

  // constant border in x: scipy pads the INTERMEDIATE with cval, not with the
  // filtered cval column -> overwrite out-of-image columns of B
  if (p.bx == IPA_BORDER_CONSTANT && p.nkx > 0 && p.nky > 0) {
    for (int r = wave; r < kTileH; r += 4)
      for (int c = lane; c < LW; c += 64) {
        int gx = x0 - p.hxa + c;
        if (gx < 0 || gx >= p.w) B[r * LW + c] = cval;
      }
    __syncthreads();
  }

  // x pass: 4 px per thread, 4 rows per thread
  int ox = x0 + threadIdx.x * 4;
#pragma unroll 1
  for (int oy = 0; oy < 4; oy++) {
    int lr = threadIdx.y * 4 + oy;
    int y = y0 + lr;
    if (y >= p.h || ox >= p.w) continue;
    T acc[4] = {(T)0, (T)0, (T)0, (T)0};
    const T* brow = B + lr * LW + threadIdx.x * 4 + p.hxa - p.hx;
    if (p.nkx > 0) {
      for (int j = 0; j < p.nkx; j++) {
        T w = wts.kx[j];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = ipa_fma(w, brow[j + k], acc[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) acc[k] = brow[k];
    }
    int n = p.w - ox < 4 ? p.w - ox : 4;
    T* row = dst + (long)y * p.dpitch + ox;
    if (p.vec_out && n == 4) {
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(row) = float4{acc[0], acc[1], acc[2], acc[3]};
      } else {
        reinterpret_cast<double2*>(row)[0] = double2{acc[0], acc[1]};
        reinterpret_cast<double2*>(row)[1] = double2{acc[2], acc[3]};
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < n) row[k] = acc[k];
    }
  }
}

// padded copy (filters/_extendArrayForConvolution.py:5-97): pure index remap
template <typename T>
__global__ void __launch_bounds__(256)
extend_kernel(const T* __restrict__ src, int h, int w, long spitch, int px, int py, int bx, int by,
              T* __restrict__ dst, long dpitch) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= w + 2 * px || y >= h + 2 * py) return;
  int sx = resolve_idx(x - px, w, bx), sy = resolve_idx(y - py, h, by);
  dst[(long)y * dpitch + x] = (sx < 0 || sy < 0) ? (T)0 : src[(long)sy * spitch + sx];
}

}  // namespace ipa

using namespace ipa;

static bool rows_aligned16(const void* base, long pitch, long frame, int n_frames, size_t es) {
  if (((uintptr_t)base) & (IPA_VEC_ALIGN - 1)) return false;
  if ((pitch * (long)es) & (IPA_VEC_ALIGN - 1)) return false;
  if (n_frames > 1 && ((frame * (long)es) & (IPA_VEC_ALIGN - 1))) return false;
  return true;
}

template <typename T, int K>
static void launch_conv(ipa_ctx* ctx, const ConvParams& p, const double* kernel, int n_frames) {
  Weights<T, K * K> w;
  for (int i = 0; i < K * K; i++) w.w[i] = (T)kernel[i];
  using G = conv_geom<K>;
  size_t lds = (size_t)lds_rows<K>() * G::LW * sizeof(T);
  dim3 grid(p.tiles, (unsigned)n_frames), block(32, 8);
  hipLaunchKernelGGL((conv_kernel<T, K, K>), grid, block, lds, ctx->stream, p, w);
}

template <typename T>
static int conv_typed(ipa_ctx* ctx, ConvParams& p, const double* kernel, int kh, int kw,
                      int n_frames) {
  bool fast = (kh == kw) && (kh == 3 || kh == 5 || kh == 7 || kh == 9 || kh == 11);
  if (fast && sizeof(T) == 8 && kh > 7) fast = false;  // f64: tuned path instantiated to 7x7
  if constexpr (sizeof(T) == 4) {
    // float32: the wave-marching stencil (wave_stencil.hpp)
    // (masked filtering stays on the LDS-tiled kernel; so do 9x9 / 11x11 with the context knob big_wave = 0,
    // the tuning knob that A/Bs wave_conv_big.hip against it: 334 vs 375 us, 428 vs 487 us)
    const bool big_wave = ctx->tune.big_wave != 0;
    if (fast && (kh <= 7 || big_wave) && !p.mask) {
      WaveParams wp;
      wp.dst = p.dst; wp.dst_frame_elems = p.dst_frame_elems;
      wp.dh = p.h; wp.dw = p.w; wp.dpitch = p.dpitch;
      wp.cbx = p.bx; wp.cby = p.by;
      wp.vec_out = p.vec_out;
      LoadRowSrc src;
      src.base = (const float*)p.src; src.frame_elems = p.src_frame_elems; src.pitch = p.spitch;
      src.vec_in = p.vec_in; src.cval = (float)p.cval;
      switch (kh) {
        case 3: ipa_wave_conv_launch_k3(ctx, wp, src, kernel, n_frames); break;
        case 5: ipa_wave_conv_launch_k5(ctx, wp, src, kernel, n_frames); break;
        // (7x7 with streamed coefficients measured slower here: 285 vs 278 us)
        case 7: ipa_wave_conv_launch_k7(ctx, wp, src, kernel, n_frames); break;
        case 9: ipa_wave_conv_launch_k9(ctx, wp, src, kernel, n_frames); break;
        default: ipa_wave_conv_launch_k11(ctx, wp, src, kernel, n_frames); break;
      }
      IPA_HIP(ctx, hipGetLastError());
      return IPA_OK;
    }
  }
  if (fast) {
    switch (kh) {
      case 3: launch_conv<T, 3>(ctx, p, kernel, n_frames); break;
      case 5: launch_conv<T, 5>(ctx, p, kernel, n_frames); break;
      case 7: launch_conv<T, 7>(ctx, p, kernel, n_frames); break;
      case 9:
        if constexpr (sizeof(T) == 4) launch_conv<T, 9>(ctx, p, kernel, n_frames);
        break;
      case 11:
        if constexpr (sizeof(T) == 4) launch_conv<T, 11>(ctx, p, kernel, n_frames);
        break;
    }
    IPA_HIP(ctx, hipGetLastError());
    return IPA_OK;
  }
  // generic: weights through the table arena
  IPA_REQUIRE(ctx, (long)kh * kw <= 65536, "kernel too large (%dx%d)", kh, kw);
  IPA_REQUIRE(ctx, n_frames <= 65535, "n_frames too large");
  std::vector<T> hw((size_t)kh * kw);
  for (size_t i = 0; i < hw.size(); i++) hw[i] = (T)kernel[i];
  void* dw = nullptr;
  int rc = ipa_tab_upload(ctx, hw.data(), hw.size() * sizeof(T), &dw);
  if (rc) return rc;
  dim3 grid((p.w + 63) / 64, (p.h + 3) / 4, (unsigned)n_frames), block(64, 4);
  hipLaunchKernelGGL((conv_generic_kernel<T>), grid, block, 0, ctx->stream, p, (const T*)dw, kh, kw);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

static int check_border(ipa_ctx* ctx, int b) {
  IPA_REQUIRE(ctx, b >= IPA_BORDER_CONSTANT && b <= IPA_BORDER_REFLECT101, "unknown border mode %d", b);
  return IPA_OK;
}

// ---------------------------------------------------- channel layout --
// (H, W, C) images as cv2 hands them to remap / warpPerspective (camera/LensDistortion.py:323-326,
// camera/PerspectiveCorrection.py:401-405; the reference's own demo warps a colour PNG, :858-900)
// against the library's planes: C frames of (H, W).  One thread per pixel, its C elements in a
// row: the plane accesses are coalesced, the interleaved ones contiguous over the wave.
template <typename T, bool TO_PLANES>
__global__ void __launch_bounds__(256) channels_kernel(const T* __restrict__ src, T* __restrict__ dst, int h,
                                                       int w, int ch, long ipitch, long ppitch,
                                                       long pstride) {
  // src / dst: the interleaved image (row pitch ipitch elements) and the planes (row pitch ppitch,
  // plane stride pstride), in the direction TO_PLANES says
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const long io = (long)y * ipitch + (long)x * ch, po = (long)y * ppitch + x;
  for (int c = 0; c < ch; c++) {
    if constexpr (TO_PLANES) dst[po + c * pstride] = src[io + c];
    else dst[io + c] = src[po + c * pstride];
  }
}

template <bool TO_PLANES>
static int channels_launch(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, int ch, long ipitch,
                           long ppitch, long pstride, void* d_dst) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_src && d_dst && h > 0 && w > 0 && ch > 0, "bad arguments");
  IPA_REQUIRE(ctx, ipitch >= (long)w * ch && ppitch >= w && pstride >= (long)(h - 1) * ppitch + w,
              "pitch / plane stride smaller than the image");
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
#define IPA_CH(T)                                                                                   \
  hipLaunchKernelGGL((channels_kernel<T, TO_PLANES>), grid, block, 0, ctx->stream, (const T*)d_src, \
                     (T*)d_dst, h, w, ch, ipitch, ppitch, pstride)
  switch (dtype) {
    case IPA_U8: IPA_CH(uint8_t); break;
    case IPA_U16: IPA_CH(uint16_t); break;
    case IPA_F32: IPA_CH(float); break;
    case IPA_F64: IPA_CH(double); break;
    default: IPA_REQUIRE(ctx, false, "unknown dtype %d", dtype);
  }
#undef IPA_CH
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}


extern "C" {

int ipa_conv2d_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                   const double* kernel, int kh, int kw, const uint8_t* d_mask, long mask_pitch,
                   void* d_dst, long dst_pitch, int n_frames, long src_frame_stride,
                   long dst_frame_stride, int border_x, int border_y, double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_src && d_dst && kernel, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && kh > 0 && kw > 0, "empty image or kernel");
  IPA_REQUIRE(ctx, src_pitch >= w && dst_pitch >= w, "pitch smaller than width");
  IPA_REQUIRE(ctx, n_frames >= 1 && n_frames <= 65535, "n_frames must be in [1,65535]");
  IPA_REQUIRE(ctx, !d_mask || mask_pitch >= w, "mask pitch smaller than width");
  IPA_REQUIRE(ctx, d_src != d_dst, "conv2d cannot run in place");
  int rc = check_border(ctx, border_x);
  if (rc) return rc;
  rc = check_border(ctx, border_y);
  if (rc) return rc;
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "conv2d supports float32/float64 images (got dtype %d)", dtype);
  // an outer product ky (x) kx on the separable K + K loop (wave_sep.hpp) - knob rank1_sep bit 1.  A constant
  // border with a non-zero value does not factor (the second pass would pad the first pass's output with
  // cval instead of cval * sum(ky)): those calls stay dense.
  // (measured, 64 x 4K, ms dense / separable: 3 taps 0.82 / 1.02, 5: 0.82 / 1.00, 7: 0.94 / 0.99, 9: 1.14 / 0.97 -
  // the dense loops keep their rows in registers at stream rate up to 7 x 7: only 9 x 9 goes)
  if ((ctx->tune.rank1_sep & 2) && dtype == IPA_F32 && !d_mask && kh == kw && kh >= IPA_RANK1_PLAIN_MIN &&
      (kh == 3 || kh == 5 || kh == 7 || kh == 9) &&
      (border_value == 0.0 || (border_x != IPA_BORDER_CONSTANT && border_y != IPA_BORDER_CONSTANT))) {
    double ky[9], kx[9];
    if (ipa_rank1_factor(kernel, kh, kw, ky, kx)) {
      ctx->rank1_routed++;
      return ipa_sepconv2d_dev(ctx, d_src, dtype, h, w, src_pitch, ky, kh, kx, kw, d_dst, dst_pitch, n_frames,
                               src_frame_stride, dst_frame_stride, border_y, border_x, border_value);
    }
  }
  size_t es = ipa_dtype_size(dtype);
  ConvParams p;
  p.src = (const char*)d_src; p.dst = (char*)d_dst;
  p.mask = d_mask; p.mask_pitch = mask_pitch;
  p.src_frame_elems = src_frame_stride; p.dst_frame_elems = dst_frame_stride;
  p.h = h; p.w = w; p.spitch = src_pitch; p.dpitch = dst_pitch;
  p.bx = border_x; p.by = border_y; p.cval = border_value;
  p.tiles_x = (unsigned)((w + kTileW - 1) / kTileW);
  p.tiles = p.tiles_x * (unsigned)((h + kTileH - 1) / kTileH);
  p.vec_in = rows_aligned16(d_src, src_pitch, src_frame_stride, n_frames, es);
  p.vec_out = rows_aligned16(d_dst, dst_pitch, dst_frame_stride, n_frames, es);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  if (dtype == IPA_F32) return conv_typed<float>(ctx, p, kernel, kh, kw, n_frames);
  return conv_typed<double>(ctx, p, kernel, kh, kw, n_frames);
}

int ipa_sepconv2d_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                      const double* ky, int nky, const double* kx, int nkx, void* d_dst,
                      long dst_pitch, int n_frames, long src_frame_stride, long dst_frame_stride,
                      int border_y, int border_x, double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_src && d_dst, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0, "empty image");
  IPA_REQUIRE(ctx, nky >= 0 && nkx >= 0 && (nky == 0 || ky) && (nkx == 0 || kx), "bad kernel args");
  {
    // long kernels (e.g. sigma = 11 -> 89 taps, filters/standardDeviation.py:23) do not fit
    // the kernarg table / LDS planes: run the two axes as two launches of the generic
    // correlation with the intermediate (rounded to the image dtype, like scipy) in a
    // temporary device buffer
    size_t es0 = ipa_dtype_size(dtype);
    int hxa0 = ((nkx / 2 + 3) / 4) * 4;
    size_t lds0 = (size_t)(2 * kTileH + 2 * (nky / 2)) * (kTileW + 2 * hxa0) * es0;
    if (nky > kSepMaxTaps || nkx > kSepMaxTaps || lds0 > 150 * 1024) {
      IPA_REQUIRE(ctx, d_src != d_dst, "sepconv2d cannot run in place");
      if (nky == 0 || nkx == 0) {
        const double* k = nky ? ky : kx;
        return ipa_conv2d_dev(ctx, d_src, dtype, h, w, src_pitch, k, nky ? nky : 1, nkx ? nkx : 1,
                              nullptr, 0, d_dst, dst_pitch, n_frames, src_frame_stride,
                              dst_frame_stride, border_x, border_y, border_value);
      }
      void* tmp = nullptr;
      IPA_HIP(ctx, hipSetDevice(ctx->device));
      IPA_HIP(ctx, hipMalloc(&tmp, (size_t)n_frames * h * w * es0));
      int rc = ipa_conv2d_dev(ctx, d_src, dtype, h, w, src_pitch, ky, nky, 1, nullptr, 0, tmp, w,
                              n_frames, src_frame_stride, (long)h * w, border_x, border_y,
                              border_value);
      if (!rc)
        rc = ipa_conv2d_dev(ctx, tmp, dtype, h, w, w, kx, 1, nkx, nullptr, 0, d_dst, dst_pitch,
                            n_frames, (long)h * w, dst_frame_stride, border_x, border_y,
                            border_value);
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipFree(tmp);
      return rc;
    }
  }
  IPA_REQUIRE(ctx, (nky == 0 || (nky & 1)) && (nkx == 0 || (nkx & 1)), "tap counts must be odd");
  IPA_REQUIRE(ctx, src_pitch >= w && dst_pitch >= w, "pitch smaller than width");
  IPA_REQUIRE(ctx, n_frames >= 1 && n_frames <= 65535, "n_frames must be in [1,65535]");
  IPA_REQUIRE(ctx, d_src != d_dst, "sepconv2d cannot run in place");
  int rc = check_border(ctx, border_x);
  if (rc) return rc;
  rc = check_border(ctx, border_y);
  if (rc) return rc;
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "sepconv2d supports float32/float64 images (got dtype %d)", dtype);
  size_t es = ipa_dtype_size(dtype);
  if (dtype == IPA_F32 && nky == nkx && (nky == 3 || nky == 5 || nky == 7 || nky == 9)) {
    // float32, equal short tap counts: the wave-marching separable kernel (wave_sep.hip)
    WaveParams wp;
    wp.dst = (char*)d_dst; wp.dst_frame_elems = dst_frame_stride;
    wp.dh = h; wp.dw = w; wp.dpitch = dst_pitch;
    wp.cbx = border_x; wp.cby = border_y;
    wp.vec_out = rows_aligned16(d_dst, dst_pitch, dst_frame_stride, n_frames, es);
    LoadRowSrc src;
    src.base = (const float*)d_src; src.frame_elems = src_frame_stride; src.pitch = src_pitch;
    src.vec_in = rows_aligned16(d_src, src_pitch, src_frame_stride, n_frames, es);
    src.cval = (float)border_value;
    IPA_HIP(ctx, hipSetDevice(ctx->device));
    if (ipa_wave_sep_launch(ctx, wp, src, ky, nky, kx, nkx, n_frames, (float)border_value) == 0) {
      IPA_HIP(ctx, hipGetLastError());
      return IPA_OK;
    }
  }
  SepParams p;
  p.src = (const char*)d_src; p.dst = (char*)d_dst;
  p.src_frame_elems = src_frame_stride; p.dst_frame_elems = dst_frame_stride;
  p.h = h; p.w = w; p.spitch = src_pitch; p.dpitch = dst_pitch;
  p.bx = border_x; p.by = border_y; p.cval = border_value;
  p.tiles_x = (unsigned)((w + kTileW - 1) / kTileW);
  p.tiles = p.tiles_x * (unsigned)((h + kTileH - 1) / kTileH);
  p.nky = nky; p.nkx = nkx; p.hy = nky / 2; p.hx = nkx / 2;
  p.hxa = ((p.hx + 3) / 4) * 4;
  p.vec_in = rows_aligned16(d_src, src_pitch, src_frame_stride, n_frames, es);
  p.vec_out = rows_aligned16(d_dst, dst_pitch, dst_frame_stride, n_frames, es);
  int LW = kTileW + 2 * p.hxa;
  size_t lds = (size_t)(kTileH + 2 * p.hy + kTileH) * LW * es;
  IPA_REQUIRE(ctx, lds <= 160 * 1024, "separable kernel too large for LDS");
  dim3 grid(p.tiles, (unsigned)n_frames), block(32, 8);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  if (dtype == IPA_F32) {
    SepWeights<float> sw;
    for (int i = 0; i < nky; i++) sw.ky[i] = (float)ky[i];
    for (int i = 0; i < nkx; i++) sw.kx[i] = (float)kx[i];
    if (lds > 64 * 1024)
      IPA_HIP(ctx, hipFuncSetAttribute((const void*)sepconv_kernel<float>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((sepconv_kernel<float>), grid, block, lds, ctx->stream, p, sw);
  } else {
    SepWeights<double> sw;
    for (int i = 0; i < nky; i++) sw.ky[i] = ky[i];
    for (int i = 0; i < nkx; i++) sw.kx[i] = kx[i];
    if (lds > 64 * 1024)
      IPA_HIP(ctx, hipFuncSetAttribute((const void*)sepconv_kernel<double>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((sepconv_kernel<double>), grid, block, lds, ctx->stream, p, sw);
  }
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_extend_array_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                         int kx, int ky, int modex, int modey, void* d_dst, long dst_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_src && d_dst && h > 0 && w > 0 && kx >= 1 && ky >= 1, "bad arguments");
  int px = kx / 2, py = ky / 2;
  // the reference asserts ky//2 < rows and kx//2 < cols (:32-33)
  IPA_REQUIRE(ctx, py < h && px < w, "kernel half-size must be smaller than the array");
  IPA_REQUIRE(ctx, src_pitch >= w && dst_pitch >= w + 2 * px, "pitch smaller than width");
  int rc = check_border(ctx, modex);
  if (rc) return rc;
  rc = check_border(ctx, modey);
  if (rc) return rc;
  dim3 grid((w + 2 * px + 63) / 64, (h + 2 * py + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  switch (dtype) {
    case IPA_U8:
      hipLaunchKernelGGL((extend_kernel<uint8_t>), grid, block, 0, ctx->stream,
                         (const uint8_t*)d_src, h, w, src_pitch, px, py, modex, modey,
                         (uint8_t*)d_dst, dst_pitch);
      break;
    case IPA_U16:
      hipLaunchKernelGGL((extend_kernel<uint16_t>), grid, block, 0, ctx->stream,
                         (const uint16_t*)d_src, h, w, src_pitch, px, py, modex, modey,
                         (uint16_t*)d_dst, dst_pitch);
      break;
    case IPA_F32:
      hipLaunchKernelGGL((extend_kernel<float>), grid, block, 0, ctx->stream, (const float*)d_src,
                         h, w, src_pitch, px, py, modex, modey, (float*)d_dst, dst_pitch);
      break;
    case IPA_F64:
      hipLaunchKernelGGL((extend_kernel<double>), grid, block, 0, ctx->stream,
                         (const double*)d_src, h, w, src_pitch, px, py, modex, modey,
                         (double*)d_dst, dst_pitch);
      break;
    default: IPA_REQUIRE(ctx, false, "unknown dtype %d", dtype);
  }
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_deinterleave_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, int channels,
                         long src_pitch, void* d_dst, long dst_pitch, long dst_plane_stride) {
  return channels_launch<true>(ctx, d_src, dtype, h, w, channels, src_pitch, dst_pitch, dst_plane_stride, d_dst);
}
int ipa_interleave_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, int channels,
                       long src_pitch, long src_plane_stride, void* d_dst, long dst_pitch) {
  return channels_launch<false>(ctx, d_src, dtype, h, w, channels, dst_pitch, src_pitch, src_plane_stride, d_dst);
}

// ---------------------------------------------------- host-pointer variants --
int ipa_conv2d(ipa_ctx* ctx, const void* src, int dtype, int h, int w, const double* kernel,
               int kh, int kw, const uint8_t* mask, void* dst, int n_frames, int border_x,
               int border_y, double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, src && dst && h > 0 && w > 0 && n_frames >= 1, "bad arguments");
  size_t es = ipa_dtype_size(dtype);
  IPA_REQUIRE(ctx, es, "unknown dtype");
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t ib = (size_t)h * w * es * n_frames, mb = mask ? (size_t)h * w : 0;
  int rc = ipa_ws_reserve(ctx, 2 * up(ib) + up(mb));
  if (rc) return rc;
  char* d_in = (char*)ctx->ws;
  char* d_out = d_in + up(ib);
  uint8_t* d_m = mask ? (uint8_t*)(d_out + up(ib)) : nullptr;
  IPA_HIP(ctx, hipMemcpyAsync(d_in, src, ib, hipMemcpyHostToDevice, ctx->stream));
  if (mask) IPA_HIP(ctx, hipMemcpyAsync(d_m, mask, mb, hipMemcpyHostToDevice, ctx->stream));
  rc = ipa_conv2d_dev(ctx, d_in, dtype, h, w, w, kernel, kh, kw, d_m, w, d_out, w, n_frames,
                      (long)h * w, (long)h * w, border_x, border_y, border_value);
  if (rc) return rc;
  IPA_HIP(ctx, hipMemcpyAsync(dst, d_out, ib, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

int ipa_sepconv2d(ipa_ctx* ctx, const void* src, int dtype, int h, int w, const double* ky,
                  int nky, const double* kx, int nkx, void* dst, int n_frames, int border_y,
                  int border_x, double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, src && dst && h > 0 && w > 0 && n_frames >= 1, "bad arguments");
  size_t es = ipa_dtype_size(dtype);
  IPA_REQUIRE(ctx, es, "unknown dtype");
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t ib = (size_t)h * w * es * n_frames;
  int rc = ipa_ws_reserve(ctx, 2 * up(ib));
  if (rc) return rc;
  char* d_in = (char*)ctx->ws;
  char* d_out = d_in + up(ib);
  IPA_HIP(ctx, hipMemcpyAsync(d_in, src, ib, hipMemcpyHostToDevice, ctx->stream));
  rc = ipa_sepconv2d_dev(ctx, d_in, dtype, h, w, w, ky, nky, kx, nkx, d_out, w, n_frames,
                         (long)h * w, (long)h * w, border_y, border_x, border_value);
  if (rc) return rc;
  IPA_HIP(ctx, hipMemcpyAsync(dst, d_out, ib, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

}  // extern "C"
