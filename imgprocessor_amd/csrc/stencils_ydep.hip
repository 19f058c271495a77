// stencils_ydep.hip — the two LDS-tiled secondary stencils with 4 pixels per lane:
// varYSizeGaussianFilter entirely on the device (reference: filters/varYSizeGaussianFilter.py:9-68)
// and the local standard deviation (filters/standardDeviation.py:34-70, at the end of the file).
//
//   ydep_gauss_cols_kernel   the per-row y responses: row r of the reference's table is
//                            gaussian_filter(delta_(ky,kx), (stdys[r], stdx)) (:40-46) and the
//                            delta is separable, so kernels[r][ii][jj] = cols[r][ii] * rowk[jj]
//                            with cols[r] = gaussian_filter1d(delta_ky, stdys[r], 'reflect') -
//                            scipy's weights exp(-x^2 / 2 sigma^2) over |x| <= int(4 sigma + 0.5),
//                            normalised, summed over every tap that lands on the delta after
//                            reflection.  stdys = numpy.linspace(mn, mx, H) is evaluated here
//                            too: the host uploads nothing but the kx doubles of rowk.
//   conv_ydep_sep_kernel     the NaN-skipping row-dependent correlation (:53-68) with the
//                            coefficient formed on the fly.  One workgroup = 256 px x RB rows:
//                            the source window and the RB rows of `cols` are staged in LDS once
//                            (border resolved per element, rows shared by the k0 output rows
//                            that use them), every lane owns 4 pixels, coefficients are LDS
//                            broadcasts.  The generic kernel of stencils.hip fetched every
//                            coefficient with its own scalar load per 64 pixels and was bound by
//                            that latency (196 us per 4K frame, 11 x 3 taps).
// Summation order per pixel: ii = 0..k0-1, jj = 0..k1-1, accumulation in double like the
// reference's numba loop; the products cols * rowk are the doubles numpy's outer product holds.
#include <type_traits>

#include "common.hpp"

namespace ipa {

__global__ void __launch_bounds__(256)
ydep_gauss_cols_kernel(int h, int ky, double mn, double mx, double truncate, double* cols) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= h * ky) return;
  const int r = idx / ky, ii = idx - r * ky;
  // numpy.linspace(mn, mx, h)[r]
  double sigma;
  if (h == 1) sigma = mn;
  else if (r == h - 1) sigma = mx;
  else sigma = (double)r * ((mx - mn) / (double)(h - 1)) + mn;
  const int c = ky / 2;
  if (!(sigma > 1e-15)) {  // gaussian_filter1d leaves the delta untouched
    cols[idx] = ii == c ? 1.0 : 0.0;
    return;
  }
  const int radius = (int)(truncate * sigma + 0.5);
  double sum = 0.0;
  for (int m = -radius; m <= radius; m++) {
    const double t = (double)m / sigma;
    sum += exp(-0.5 * (t * t));
  }
  double acc = 0.0;
  const int p = 2 * ky;
  for (int m = -radius; m <= radius; m++) {
    int k = (ii + m) % p;  // position on the symmetric ('reflect') extension of the delta row
    if (k < 0) k += p;
    k = k < ky ? k : p - 1 - k;
    if (k == c) {
      const double t = (double)m / sigma;
      acc += exp(-0.5 * (t * t)) / sum;
    }
  }
  cols[idx] = acc;
}

// kernels[r][ii][jj] = cols[r][ii] * rowk[jj]: the reference's whole table, for windows the LDS
// tile of conv_ydep_sep_kernel cannot hold (the generic kernels of stencils.hip read it)
__global__ void __launch_bounds__(256)
ydep_outer_kernel(const double* __restrict__ cols, const double* __restrict__ rowk, long n_cols,
                  int kx, double* __restrict__ table) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_cols * kx) return;
  const long rc = idx / kx;
  table[idx] = cols[rc] * rowk[idx - rc * kx];
}

constexpr int kYdepTW = 256;  // output pixels per workgroup row (4 per lane)

// K1 = kx when it is 1, 3 or 5 (window of a lane read once per tile row, loops unrolled),
// 0 = any kx
template <typename T, int K1>
__global__ void __launch_bounds__(256)
conv_ydep_sep_kernel(const T* __restrict__ src, int h, int w, long spitch,
                     const double* __restrict__ cols, int k0, const double* __restrict__ rowk,
                     int k1, int bx, int by, int rb, T* __restrict__ dst, long dpitch,
                     int vec_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ydep_lds[];
  const int tw = kYdepTW + k1 - 1 + 3;  // + 3: the last lane's 4-px window reads
  const int th = rb + k0 - 1;
  double* lcols = reinterpret_cast<double*>(ydep_lds);              // rb * k0
  double* lrow = lcols + rb * k0;                                    // k1
  T* tile = reinterpret_cast<T*>(lrow + k1);                         // th * tw
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x0 = blockIdx.x * kYdepTW, y0 = blockIdx.y * rb;

  for (int e = threadIdx.x; e < rb * k0; e += 256) {
    const int rl = e / k0;
    lcols[e] = y0 + rl < h ? cols[(long)(y0 + rl) * k0 + (e - rl * k0)] : 0.0;
  }
  for (int e = threadIdx.x; e < k1; e += 256) lrow[e] = rowk[e];
  // columns of the tile inside the image: no border resolution (a division per element for
  // 'wrap') in the blocks away from the left / right edge
  const bool xin = x0 - k1 / 2 >= 0 && x0 - k1 / 2 + tw <= w;
  for (int ty = wave; ty < th; ty += 4) {
    const int yy = resolve_idx(y0 - k0 / 2 + ty, h, by);  // wave-uniform
    const T* srow = src + (long)(yy < 0 ? 0 : yy) * spitch;
    // NaN pixels are skipped by the reference (:62-64): staged as 0 they add k * 0 = +-0 to a
    // sum that starts at +0 and therefore never is -0 - the same value as not adding at all
    if (xin && yy >= 0) {
      const T* sp = srow + (x0 - k1 / 2);
      for (int tx = lane; tx < tw; tx += 64) {
        const T a = sp[tx];
        tile[ty * tw + tx] = a == a ? a : (T)0;
      }
    } else {
      for (int tx = lane; tx < tw; tx += 64) {
        const int xx = resolve_idx(x0 - k1 / 2 + tx, w, bx);
        const T a = (yy < 0 || xx < 0) ? (T)0 : srow[xx];
        tile[ty * tw + tx] = a == a ? a : (T)0;
      }
    }
  }
  __syncthreads();

  for (int rl = wave; rl < rb; rl += 4) {
    const int r = y0 + rl;
    if (r >= h) break;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (K1 > 0) {
      double rk[K1];
#pragma unroll
      for (int jj = 0; jj < K1; jj++) rk[jj] = lrow[jj];
#pragma unroll 2
      for (int ii = 0; ii < k0; ii++) {
        const double cy = lcols[rl * k0 + ii];  // LDS broadcast
        const T* tp = tile + (rl + ii) * tw + 4 * lane;
        double a[K1 + 3];
#pragma unroll
        for (int q = 0; q < K1 + 3; q++) a[q] = (double)tp[q];
#pragma unroll
        for (int jj = 0; jj < K1; jj++) {
          const double k = cy * rk[jj];
#pragma unroll
          for (int q = 0; q < 4; q++)
            v[q] += k * a[jj + q];
        }
      }
    } else {
      for (int ii = 0; ii < k0; ii++) {
        const double cy = lcols[rl * k0 + ii];  // LDS broadcast
        const T* tp = tile + (rl + ii) * tw + 4 * lane;
        for (int jj = 0; jj < k1; jj++) {
          const double k = cy * lrow[jj];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            v[q] += k * (double)tp[jj + q];  // (NaN staged as 0: skipped, no renormalisation)
          }
        }
      }
    }
    const int c0 = x0 + 4 * lane;
    T* orow = dst + (long)r * dpitch + c0;
    if (vec_out && c0 + 4 <= w) {
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(orow) = float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
      } else {
        reinterpret_cast<double2*>(orow)[0] = double2{v[0], v[1]};
        reinterpret_cast<double2*>(orow)[1] = double2{v[2], v[3]};
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (c0 + q < w) orow[q] = (T)v[q];
    }
  }
}


// ---------------------------------------------------------------------------------------
// local standard deviation (filters/standardDeviation.py:34-70, _calc) for square windows:
// std[i,j] = sqrt( sum over the clipped window [i-hk, i+hk) x [j-hk, j+hk) of (img - blurred[i,j])^2
//                  / ((rows - 1) * (cols - 1)) )      - the reference's divisor, as written.
// One workgroup = 256 px x RB rows; the window rows are staged in LDS once, every lane owns 4
// pixels and reads the 2 hk + 3 values of a tile row once for its 4 windows; loops unrolled for
// the window size.  Same summation order as the reference (rows, then columns) in double.
template <typename T, int HK>
__global__ void __launch_bounds__(256)
local_std_wave_kernel(const T* __restrict__ img, const T* __restrict__ blurred, int gx, int gy,
                      long pitch, long bpitch, int rb, T* __restrict__ out, long opitch,
                      int vec_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char std_lds2[];
  T* tile = reinterpret_cast<T*>(std_lds2);
  constexpr int tw = kYdepTW + 2 * HK + 3;
  const int th = rb + 2 * HK;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j0 = blockIdx.x * kYdepTW, i0 = blockIdx.y * rb;
  for (int ty = wave; ty < th; ty += 4) {
    int ii = i0 - HK + ty;
    ii = ii < 0 ? 0 : (ii >= gx ? gx - 1 : ii);  // outside the image: never summed
    const T* srow = img + (long)ii * pitch;
    for (int tx = lane; tx < tw; tx += 64) {
      int jj = j0 - HK + tx;
      jj = jj < 0 ? 0 : (jj >= gy ? gy - 1 : jj);
      tile[ty * tw + tx] = srow[jj];
    }
  }
  __syncthreads();
  // every window of the block inside the image? (then no tap needs a bounds test)
  const bool inner = i0 - HK >= 0 && i0 + rb - 1 + HK <= gx && j0 - HK >= 0 &&
                     j0 + kYdepTW - 1 + HK <= gy;
  for (int rl = wave; rl < rb; rl += 4) {
    const int i = i0 + rl;
    if (i >= gx) break;
    const int jb = j0 + 4 * lane;
    double mean[4], val[4] = {0.0, 0.0, 0.0, 0.0};
    const T* brow = blurred + (long)i * bpitch + jb;
    if (sizeof(T) == 4 && vec_out && jb + 4 <= gy && ((reinterpret_cast<uintptr_t>(brow) & 15) == 0)) {
      const float4 b4 = *reinterpret_cast<const float4*>(brow);
      mean[0] = (double)b4.x; mean[1] = (double)b4.y; mean[2] = (double)b4.z; mean[3] = (double)b4.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) mean[q] = jb + q < gy ? (double)brow[q] : 0.0;
    }
    const int xmn = i - HK < 0 ? 0 : i - HK, xmx = i + HK > gx ? gx : i + HK;
    auto rows = [&](auto Inner) {
      constexpr bool kInner = decltype(Inner)::value;
#pragma unroll
      for (int di = 0; di < 2 * HK; di++) {
        const int ii = i - HK + di;
        const bool row_ok = ii >= xmn && ii < xmx;
        const T* tp = tile + (rl + di) * tw + 4 * lane;
        double a[2 * HK + 3];
#pragma unroll
        for (int q = 0; q < 2 * HK + 3; q++) a[q] = (double)tp[q];
#pragma unroll
        for (int dj = 0; dj < 2 * HK; dj++) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const double d = a[dj + q] - mean[q];
            if constexpr (kInner) {
              val[q] += d * d;
            } else {
              const int jj = jb + q - HK + dj;
              if (row_ok && jj >= 0 && jj < gy) val[q] += d * d;
            }
          }
        }
      }
    };
    if (inner) rows(std::true_type{});
    else rows(std::false_type{});
    T res[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int j = jb + q;
      const int ymn = j - HK < 0 ? 0 : j - HK, ymx = j + HK > gy ? gy : j + HK;
      const double npx = (double)((xmx - xmn - 1) * (ymx - ymn - 1));
      // float32 images: the double sum is rounded once, quotient and root in float32 (the
      // stored result is float32; 1e-7 relative against the double expression)
      if constexpr (sizeof(T) == 4) res[q] = sqrtf((float)val[q] / (float)npx);
      else res[q] = (T)sqrt(val[q] / npx);
    }
    T* orow = out + (long)i * opitch + jb;
    if (vec_out && jb + 4 <= gy) {
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(orow) = float4{res[0], res[1], res[2], res[3]};
      } else {
        reinterpret_cast<double2*>(orow)[0] = double2{res[0], res[1]};
        reinterpret_cast<double2*>(orow)[1] = double2{res[2], res[3]};
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (jb + q < gy) orow[q] = res[q];
    }
  }
}

template <typename T, int HK>
static void local_std_wave_launch(ipa_ctx* ctx, const void* img, const void* blurred, int h, int w,
                                  long pitch, long bpitch, void* out, long opitch) {
  // rows per block, 4K float32 frame: 4: 45.7 us, 8: 41.7, 12: 41.5, 16: 44.8, 24: 54.0, 32: 49.8
  const int rb = sizeof(T) == 4 ? 12 : 16;
  const size_t lds = (size_t)(rb + 2 * HK) * (kYdepTW + 2 * HK + 3) * sizeof(T);
  const int vec_out = (((uintptr_t)out) % 16 == 0) && ((opitch * (long)sizeof(T)) % 16 == 0);
  dim3 grid((w + kYdepTW - 1) / kYdepTW, (h + rb - 1) / rb), block(256);
  hipLaunchKernelGGL((local_std_wave_kernel<T, HK>), grid, block, lds, ctx->stream, (const T*)img,
                     (const T*)blurred, h, w, pitch, bpitch, rb, (T*)out, opitch, vec_out);
}

}  // namespace ipa

using namespace ipa;

// square windows of ksize 2..11 (hk = 1..5); 1 = not covered (the generic kernels of
// stencils.hip run instead)
int ipa_local_std_wave_launch(ipa_ctx* ctx, const void* img, const void* blurred, int dtype, int h,
                              int w, long pitch, long bpitch, int hkx, int hky, void* out,
                              long opitch) {
  if (hkx != hky || hkx < 1 || hkx > 5) return 1;
#define IPA_STD_CASE(HK)                                                                          \
  case HK:                                                                                        \
    if (dtype == IPA_F32) local_std_wave_launch<float, HK>(ctx, img, blurred, h, w, pitch, bpitch, out, opitch); \
    else local_std_wave_launch<double, HK>(ctx, img, blurred, h, w, pitch, bpitch, out, opitch);  \
    return 0;
  switch (hkx) {
    IPA_STD_CASE(1) IPA_STD_CASE(2) IPA_STD_CASE(3) IPA_STD_CASE(4) IPA_STD_CASE(5)
  }
#undef IPA_STD_CASE
  return 1;
}

extern "C" {

// filters/varYSizeGaussianFilter.py:9-50 in one call: stdys = linspace(sig_min, sig_max, h),
// per-row Gaussian tables of ky x kx taps (truncate 4.0, reflecting delta), NaN-skipping
// correlation with borders (border_x, border_y).  `rowk` = the kx x-responses (host, tiny).
int ipa_var_y_gauss_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                        double sig_min, double sig_max, int ky, const double* rowk, int kx,
                        int border_x, int border_y, void* d_dst, long dst_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_src && d_dst && rowk, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && ky > 0 && kx > 0 && (ky & 1) && (kx & 1),
              "empty image or even kernel size");
  IPA_REQUIRE(ctx, kx <= 255 && ky <= 4095, "kernel too large");
  IPA_REQUIRE(ctx, src_pitch >= w && dst_pitch >= w, "pitch smaller than width");
  IPA_REQUIRE(ctx, border_x >= 0 && border_x <= IPA_BORDER_REFLECT101 && border_y >= 0 &&
                       border_y <= IPA_BORDER_REFLECT101, "unknown border mode");
  IPA_REQUIRE(ctx, d_src != d_dst, "varYSizeGaussianFilter cannot run in place");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "varYSizeGaussianFilter supports float32/float64 (got dtype %d)", dtype);
  const size_t esz = dtype == IPA_F32 ? 4 : 8;
  // rows per workgroup: ~2 x the kernel height (read amplification <= 1.5) while the tile stays
  // below ~30 KB (5 workgroups per CU), never above 60 KB
  const int tw = kYdepTW + kx - 1 + 3;
  auto lds_of = [&](int r) {
    return (size_t)r * ky * 8 + (size_t)kx * 8 + (size_t)(r + ky - 1) * tw * esz;
  };
  int rb = 2 * ky < 16 ? 16 : (2 * ky + 3) / 4 * 4;
  if (rb > 64) rb = 64;
  while (rb > 8 && lds_of(rb) > 30 * 1024) rb -= 4;
  while (rb > 4 && lds_of(rb) > 60 * 1024) rb -= 4;
  // windows beyond the tile (stdyrange above ~23 for float32, ~11 for float64): the whole
  // h x ky x kx table is expanded on the device and the generic entry point runs it - any
  // stdyrange the reference accepts works, as before the tiled kernel existed
  const bool tiled = lds_of(rb) <= 64 * 1024;
  // tables in the context's plan scratch: cols (h * ky doubles) + rowk (kx doubles) [+ table]
  const size_t cols_b = (size_t)h * ky * sizeof(double);
  const size_t table_b = tiled ? 0 : cols_b * kx;
  int rc = ipa_plan_reserve(ctx, cols_b + (size_t)kx * sizeof(double) + table_b);
  if (rc) return rc;
  double* d_cols = reinterpret_cast<double*>(ctx->plan);
  double* d_rowk = d_cols + (size_t)h * ky;
  double* d_table = d_rowk + kx;
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  IPA_HIP(ctx, hipMemcpyAsync(d_rowk, rowk, (size_t)kx * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  // rowk is caller memory: the copy above must have read it before we return
  hipLaunchKernelGGL(ydep_gauss_cols_kernel, dim3((h * ky + 255) / 256), dim3(256), 0, ctx->stream,
                     h, ky, sig_min, sig_max, 4.0, d_cols);
  if (!tiled) {
    const long n = (long)h * ky * kx;
    hipLaunchKernelGGL(ydep_outer_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_cols, d_rowk, (long)h * ky, kx, d_table);
    IPA_HIP(ctx, hipGetLastError());
    rc = ipa_conv_ydep_dev(ctx, d_src, dtype, h, w, src_pitch, d_table, ky, kx, border_x,
                           border_y, d_dst, dst_pitch);
    if (rc) return rc;
    IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // rowk (host) may be reused by the caller
    return IPA_OK;
  }
  const int vec_out = (((uintptr_t)d_dst) % 16 == 0) && ((dst_pitch * (long)esz) % 16 == 0);
  dim3 grid((w + kYdepTW - 1) / kYdepTW, (h + rb - 1) / rb), block(256);
#define IPA_YDEP_LAUNCH(T, K1)                                                                 \
  hipLaunchKernelGGL((conv_ydep_sep_kernel<T, K1>), grid, block, lds_of(rb), ctx->stream,         \
                     (const T*)d_src, h, w, src_pitch, d_cols, ky, d_rowk, kx, border_x, border_y, \
                     rb, (T*)d_dst, dst_pitch, vec_out)
  if (dtype == IPA_F32) {
    if (kx == 1) IPA_YDEP_LAUNCH(float, 1);
    else if (kx == 3) IPA_YDEP_LAUNCH(float, 3);
    else if (kx == 5) IPA_YDEP_LAUNCH(float, 5);
    else IPA_YDEP_LAUNCH(float, 0);
  } else {
    if (kx == 1) IPA_YDEP_LAUNCH(double, 1);
    else if (kx == 3) IPA_YDEP_LAUNCH(double, 3);
    else if (kx == 5) IPA_YDEP_LAUNCH(double, 5);
    else IPA_YDEP_LAUNCH(double, 0);
  }
#undef IPA_YDEP_LAUNCH
  IPA_HIP(ctx, hipGetLastError());
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // rowk (host) may be reused by the caller
  return IPA_OK;
}

}  // extern "C"
