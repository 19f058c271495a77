// stencils.hip — the secondary filters/ stencils of the hot path on gfx950:
//   ipa_conv_ydep*   filters/varYSizeGaussianFilter.py:53-68 (_2dConvolutionYdependentKernel):
//                    a dense k0 x k1 correlation whose coefficients depend on the ROW,
//                    NaN pixels skipped; borders resolved on the fly (the reference pads
//                    with extendArrayForConvolution, default modex='wrap', modey='reflect')
//   ipa_local_std*   filters/standardDeviation.py:34-70 (_calc): local standard deviation
//                    around a given (Gaussian-blurred) mean, reference quirks included
//   ipa_masked_mean* filters/maskedFilter.py:43-72 (_calcMean): mean of the unmasked pixels of
//                    the clipped window, for the masked (fill) or the unmasked pixels
//   ipa_nan_max*     filters/nan_maximum_filter.py:17-37: NaN-ignoring window maximum
//
// All are one-output-pixel-per-lane kernels: a wave covers 64 consecutive
// pixels of one row, so the per-row coefficient table of conv_ydep is
// wave-uniform (scalar loads) and the window reads of neighbouring lanes
// coalesce in L1.  Accumulation is in double like the reference's numba code
// (float64 coefficient tables, `val` promoted to float64).
#include "common.hpp"

namespace ipa {

template <typename T>
__global__ void __launch_bounds__(256)
conv_ydep_kernel(const T* __restrict__ src, int h, int w, long spitch,
                 const double* __restrict__ kernels, int k0, int k1, int bx, int by,
                 T* __restrict__ dst, long dpitch) {
  const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y * 4 + threadIdx.y;
  if (c >= w || r >= h) return;
  const double* kr = kernels + (long)r * k0 * k1;  // wave-uniform
  double v = 0.0;
  for (int ii = 0; ii < k0; ii++) {
    int yy = resolve_idx(r + ii - k0 / 2, h, by);
    for (int jj = 0; jj < k1; jj++) {
      int xx = resolve_idx(c + jj - k1 / 2, w, bx);
      double a = (yy < 0 || xx < 0) ? 0.0 : (double)src[(long)yy * spitch + xx];
      if (a == a) v += kr[ii * k1 + jj] * a;  // NaN-aware: skip, no renormalisation
    }
  }
  dst[(long)r * dpitch + c] = (T)v;
}

// window [i-hkx, min(i+hkx, gx)) x [j-hky, min(j+hky, gy)), clipped at 0;
// divisor = (rows-1)*(cols-1): the reference divides by its last loop indices
template <typename T>
__global__ void __launch_bounds__(256)
local_std_kernel(const T* __restrict__ img, const T* __restrict__ blurred, int gx, int gy,
                 long pitch, long bpitch, int hkx, int hky, T* __restrict__ out, long opitch) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= gx || j >= gy) return;
  int xmn = i - hkx < 0 ? 0 : i - hkx, xmx = i + hkx > gx ? gx : i + hkx;
  int ymn = j - hky < 0 ? 0 : j - hky, ymx = j + hky > gy ? gy : j + hky;
  double mean = (double)blurred[(long)i * bpitch + j], val = 0.0;
  for (int ii = xmn; ii < xmx; ii++)
    for (int jj = ymn; jj < ymx; jj++) {
      double d = (double)img[(long)ii * pitch + jj] - mean;
      val += d * d;
    }
  double npx = (double)((xmx - xmn - 1) * (ymx - ymn - 1));
  out[(long)i * opitch + j] = (T)sqrt(val / npx);
}

// filters/maskedFilter.py:43-72 (_calcMean).  FILL: pixels with mask != 0 get the mean of the
// mask == 0 pixels in the clipped window (left untouched when there are none) — dst may be
// src, written pixels are never read.  !FILL: pixels with mask == 0 get that mean, the others
// NaN (the reference's np.full_like(arr, nan) output).
template <typename T, bool FILL>
__global__ void __launch_bounds__(256)
masked_mean_kernel(const T* src, const unsigned char* __restrict__ mask, int gx, int gy,
                   long pitch, long mpitch, int k, T* dst, long dpitch) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= gx || j >= gy) return;
  const bool masked = mask[(long)i * mpitch + j] != 0;
  if (masked != FILL) {
    if constexpr (!FILL) dst[(long)i * dpitch + j] = (T)__builtin_nan("");
    return;
  }
  int xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
  int ymn = j - k < 0 ? 0 : j - k, ymx = j + k > gy ? gy : j + k;
  double val = 0.0;
  int n = 0;
  for (int ii = xmn; ii < xmx; ii++)
    for (int jj = ymn; jj < ymx; jj++)
      if (!mask[(long)ii * mpitch + jj]) {
        val += (double)src[(long)ii * pitch + jj];
        n++;
      }
  if (n > 0) dst[(long)i * dpitch + j] = (T)(val / (double)n);
}

// filters/nan_maximum_filter.py:17-37: np.nanmax over the clipped window (NaN when all NaN)
template <typename T>
__global__ void __launch_bounds__(256)
nan_max_kernel(const T* __restrict__ src, int gx, int gy, long pitch, int k, T* __restrict__ dst,
               long dpitch) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= gx || j >= gy) return;
  int xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
  int ymn = j - k < 0 ? 0 : j - k, ymx = j + k > gy ? gy : j + k;
  T m = (T)__builtin_nan("");
  for (int ii = xmn; ii < xmx; ii++)
    for (int jj = ymn; jj < ymx; jj++) {
      T v = src[(long)ii * pitch + jj];
      if (v == v && !(m >= v)) m = v;
    }
  dst[(long)i * dpitch + j] = m;
}

}  // namespace ipa

using namespace ipa;

extern "C" {

int ipa_masked_mean_dev(ipa_ctx* ctx, const void* d_arr, int dtype, const unsigned char* d_mask,
                        int h, int w, long pitch, long mask_pitch, int ksize, int fill_mask,
                        void* d_out, long out_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_arr && d_mask && d_out, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && ksize >= 2, "empty image or ksize < 2");
  IPA_REQUIRE(ctx, pitch >= w && mask_pitch >= w && out_pitch >= w, "pitch smaller than width");
  IPA_REQUIRE(ctx, fill_mask || d_arr != d_out, "fill_mask=0 cannot run in place");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "masked_mean supports float32/float64 (got dtype %d)", dtype);
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
#define IPA_MM(T, FILL)                                                                          \
  hipLaunchKernelGGL((masked_mean_kernel<T, FILL>), grid, block, 0, ctx->stream, (const T*)d_arr, \
                     d_mask, h, w, pitch, mask_pitch, ksize / 2, (T*)d_out, out_pitch)
  if (dtype == IPA_F32) {
    if (fill_mask) IPA_MM(float, true); else IPA_MM(float, false);
  } else {
    if (fill_mask) IPA_MM(double, true); else IPA_MM(double, false);
  }
#undef IPA_MM
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_nan_max_dev(ipa_ctx* ctx, const void* d_arr, int dtype, int h, int w, long pitch,
                    int ksize, void* d_out, long out_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_arr && d_out, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && ksize >= 2, "empty image or ksize < 2");
  IPA_REQUIRE(ctx, pitch >= w && out_pitch >= w, "pitch smaller than width");
  IPA_REQUIRE(ctx, d_arr != d_out, "nan_max cannot run in place");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "nan_max supports float32/float64 (got dtype %d)", dtype);
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  if (dtype == IPA_F32)
    hipLaunchKernelGGL((nan_max_kernel<float>), grid, block, 0, ctx->stream, (const float*)d_arr,
                       h, w, pitch, ksize / 2, (float*)d_out, out_pitch);
  else
    hipLaunchKernelGGL((nan_max_kernel<double>), grid, block, 0, ctx->stream,
                       (const double*)d_arr, h, w, pitch, ksize / 2, (double*)d_out, out_pitch);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_conv_ydep_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                      const double* d_kernels, int k0, int k1, int border_x, int border_y,
                      void* d_dst, long dst_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_src && d_dst && d_kernels, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && k0 > 0 && k1 > 0 && (k0 & 1) && (k1 & 1),
              "empty image or even kernel size");
  IPA_REQUIRE(ctx, src_pitch >= w && dst_pitch >= w, "pitch smaller than width");
  IPA_REQUIRE(ctx, border_x >= 0 && border_x <= IPA_BORDER_REFLECT101 && border_y >= 0 &&
                       border_y <= IPA_BORDER_REFLECT101, "unknown border mode");
  IPA_REQUIRE(ctx, d_src != d_dst, "conv_ydep cannot run in place");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "conv_ydep supports float32/float64 (got dtype %d)", dtype);
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  if (dtype == IPA_F32)
    hipLaunchKernelGGL((conv_ydep_kernel<float>), grid, block, 0, ctx->stream, (const float*)d_src,
                       h, w, src_pitch, d_kernels, k0, k1, border_x, border_y, (float*)d_dst,
                       dst_pitch);
  else
    hipLaunchKernelGGL((conv_ydep_kernel<double>), grid, block, 0, ctx->stream,
                       (const double*)d_src, h, w, src_pitch, d_kernels, k0, k1, border_x,
                       border_y, (double*)d_dst, dst_pitch);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_local_std_dev(ipa_ctx* ctx, const void* d_img, const void* d_blurred, int dtype, int h,
                      int w, long pitch, long blurred_pitch, int ksize_x, int ksize_y,
                      void* d_out, long out_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_img && d_blurred && d_out, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && ksize_x >= 2 && ksize_y >= 2, "empty image or ksize < 2");
  IPA_REQUIRE(ctx, pitch >= w && blurred_pitch >= w && out_pitch >= w, "pitch smaller than width");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "local_std supports float32/float64 (got dtype %d)", dtype);
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  if (dtype == IPA_F32)
    hipLaunchKernelGGL((local_std_kernel<float>), grid, block, 0, ctx->stream, (const float*)d_img,
                       (const float*)d_blurred, h, w, pitch, blurred_pitch, ksize_x / 2,
                       ksize_y / 2, (float*)d_out, out_pitch);
  else
    hipLaunchKernelGGL((local_std_kernel<double>), grid, block, 0, ctx->stream,
                       (const double*)d_img, (const double*)d_blurred, h, w, pitch, blurred_pitch,
                       ksize_x / 2, ksize_y / 2, (double*)d_out, out_pitch);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

}  // extern "C"
