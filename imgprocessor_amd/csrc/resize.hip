// resize.hip — the two "fast" large-kernel filters of the reference's filters/ package on gfx950:
//   ipa_fast_filter_stat*  filters/fastFilter.py:52-122   strided window median / mean (NaN-aware)
//   ipa_resize*            cv2.resize at filters/fastFilter.py:47-48 (INTER_LANCZOS4, float64
//                          grid of statistics -> image size) and filters/fastMean.py:14-19
//                          (INTER_AREA down, INTER_LINEAR back up)
//
// cv2.resize is restated from OpenCV's published algorithm (imgproc/src/resize.cpp) for
// single-channel float32 / float64 images; like every cv2 mode of this build it is UNPINNED (no
// cv2 in the build container) - the test suite's CPU checker restates it a second time, the tests
// compare the two and check the algorithm's identities against numpy:
//   * scale = 1 / ((double)dsize / ssize); position of destination index d:
//     f = (float)((d + 0.5) scale - 0.5), s = floor(f), f -= s  (float32)
//   * bilinear: along x a position outside the row is clamped to the edge pixel with fraction 0,
//     along y the two rows are clipped; bicubic (a = -0.75) / Lanczos4: tap indices clipped
//   * float32 coefficients; work type = the image's type; horizontal pass first, its rows rounded
//     to the work type; products summed left to right, no fused multiply-add
//   * INTER_AREA, integer scale: the block summed in groups of four, times (float)(1 / area);
//     otherwise computeResizeAreaTab's decimation tables (float32 weights), row sums first.
//     INTER_AREA upscaling (a bilinear variant in OpenCV) is not built.
#include <cfloat>
#include <cmath>
#include <vector>

#include "common.hpp"

#define IPA_NO_FMA _Pragma("clang fp contract(off)")

namespace ipa {

// ------------------------------------------------------------- separable kernels --
template <typename T, int KS>
__global__ void __launch_bounds__(256)
hresize_kernel(const T* __restrict__ src, long spitch, int sh, int sw, T* __restrict__ tmp, int dw,
               const int* __restrict__ xofs, const float* __restrict__ alpha, int xmax) {
  IPA_NO_FMA
  const int dx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (dx >= dw) return;
  const T* S = src + (long)y * spitch;
  const float* a = alpha + (long)dx * KS;
  const int sx = xofs[dx];
  T v;
  if constexpr (KS == 2) {
    if (dx >= xmax) v = S[sx] * (T)1;
    else v = S[sx] * (T)a[0] + S[sx + 1] * (T)a[1];
  } else {
    v = 0;
#pragma unroll
    for (int j = 0; j < KS; j++) {
      int sxj = sx - (KS / 2 - 1) + j;
      sxj = sxj < 0 ? 0 : (sxj >= sw ? sw - 1 : sxj);
      v += S[sxj] * (T)a[j];
    }
  }
  tmp[(long)y * dw + dx] = v;
}

template <typename T, int KS>
__global__ void __launch_bounds__(256)
vresize_kernel(const T* __restrict__ tmp, int sh, int dw, T* __restrict__ dst, long dpitch,
               const int* __restrict__ yofs, const float* __restrict__ beta) {
  IPA_NO_FMA
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw) return;
  const float* b = beta + (long)dy * KS;
  const int sy0 = yofs[dy];
  T v = 0;
#pragma unroll
  for (int k = 0; k < KS; k++) {
    int sy = sy0 - KS / 2 + 1 + k;
    sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;
    const T t = tmp[(long)sy * dw + dx] * (T)b[k];
    v = k == 0 ? t : v + t;
  }
  dst[(long)dy * dpitch + dx] = v;
}

// four result pixels per lane (rows of the intermediate and of the result 4-element aligned):
// the same per-element arithmetic on vector loads / stores
template <typename T, int KS>
__global__ void __launch_bounds__(256)
vresize4_kernel(const T* __restrict__ tmp, int sh, int dw, T* __restrict__ dst, long dpitch,
                const int* __restrict__ yofs, const float* __restrict__ beta) {
  IPA_NO_FMA
  typedef T V __attribute__((ext_vector_type(4)));
  const int dx = (blockIdx.x * 256 + threadIdx.x) * 4, dy = blockIdx.y;
  if (dx >= dw) return;
  const float* b = beta + (long)dy * KS;
  const int sy0 = yofs[dy];
  V v = 0;
#pragma unroll
  for (int k = 0; k < KS; k++) {
    int sy = sy0 - KS / 2 + 1 + k;
    sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;
    const V t = *reinterpret_cast<const V*>(tmp + (long)sy * dw + dx) * (T)b[k];
    v = k == 0 ? t : v + t;
  }
  *reinterpret_cast<V*>(dst + (long)dy * dpitch + dx) = v;
}

// ------------------------------------------------------------- INTER_AREA --
template <typename T>
__global__ void __launch_bounds__(256)
area_fast_kernel(const T* __restrict__ src, long spitch, int sh, int sw, T* __restrict__ dst,
                 long dpitch, int dh, int dw, int isx, int isy) {
  IPA_NO_FMA
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw) return;
  const float fscale = 1.f / (float)(isx * isy);
  const long sy0 = (long)dy * isy, sx0 = (long)dx * isx;
  int w = sy0 + isy <= sh ? sw / isx : 0;
  w = w < dw ? w : dw;
  T out;
  if (sy0 >= sh) {
    out = 0;
  } else if (dx < w) {
    T sum = 0;
    const int area = isx * isy;
    int k = 0;
    auto px = [&](int e) { return src[(sy0 + e / isx) * spitch + sx0 + e % isx]; };
    for (; k <= area - 4; k += 4) {
      const T a = px(k), b = px(k + 1), c = px(k + 2), d = px(k + 3);
      sum += a + b + c + d;
    }
    for (; k < area; k++) sum += px(k);
    out = (T)(sum * fscale);
  } else if (sx0 >= sw) {
    out = 0;
  } else {
    T sum = 0;
    int count = 0;
    for (int sy = 0; sy < isy && sy0 + sy < sh; sy++)
      for (int sx = 0; sx < isx && sx0 + sx < sw; sx++) {
        sum += src[(sy0 + sy) * spitch + sx0 + sx];
        count++;
      }
    out = (T)((float)sum / count);
  }
  dst[(long)dy * dpitch + dx] = out;
}

struct AreaTab { int si, di; float alpha; };

// one lane per destination pixel: its rows of the y table, per row its entries of the x table
template <typename T>
__global__ void __launch_bounds__(256)
area_kernel(const T* __restrict__ src, long spitch, T* __restrict__ dst, long dpitch, int dw,
            const AreaTab* __restrict__ xt, const int* __restrict__ xstart,
            const AreaTab* __restrict__ yt, const int* __restrict__ ystart) {
  IPA_NO_FMA
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw) return;
  const int k0 = xstart[dx], k1 = xstart[dx + 1];
  T sum = 0;
  for (int j = ystart[dy]; j < ystart[dy + 1]; j++) {
    const T* S = src + (long)yt[j].si * spitch;
    const T beta = (T)yt[j].alpha;
    T buf = 0;
    for (int k = k0; k < k1; k++) buf += S[xt[k].si] * (T)xt[k].alpha;
    sum += beta * buf;
  }
  dst[(long)dy * dpitch + dx] = sum;
}

// ------------------------------------------------------------- fastFilter's statistics --
constexpr int kStatMax = 4096;   // window elements a wave keeps in LDS (doubles)

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one wave per output cell.  fn: 0 median, 1 nanmedian, 2 mean, 3 nanmean.
template <typename T>
__global__ void __launch_bounds__(64)
fast_filter_stat_kernel(const T* __restrict__ arr, long pitch, int gx, int gy, int ksize, int every,
                        int fn, double* __restrict__ out, int n1) {
  __shared__ double buf[kStatMax];
  const int lane = threadIdx.x;
  const int jj = blockIdx.x, ii = blockIdx.y;
  const int i = ii * every, j = jj * every;
  const int xmn = i - ksize < 0 ? 0 : i - ksize, xmx = i + ksize > gx ? gx : i + ksize;
  const int ymn = j - ksize < 0 ? 0 : j - ksize, ymx = j + ksize > gy ? gy : j + ksize;
  const int nx = (xmx - xmn + every - 1) / every, ny = (ymx - ymn + every - 1) / every;
  const int total = nx * ny;
  // the finite values into LDS (order does not matter for the statistics), NaNs counted
  int n = 0, nans = 0;
  double s = 0.0;
  for (int base = 0; base < total; base += 64) {
    const int t = base + lane;
    double v = 0.0;
    bool live = t < total, isn = false;
    if (live) {
      const int a = t / ny;
      v = (double)arr[(long)(xmn + a * every) * pitch + (ymn + (t - a * ny) * every)];
      isn = v != v;
    }
    const unsigned long long keep = __ballot(live && !isn);
    nans += __popcll(__ballot(live && isn));
    if (live && !isn) {
      const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
      buf[n + __popcll(keep & below)] = v;
      s += v;
    }
    n += __popcll(keep);
  }
  __builtin_amdgcn_wave_barrier();
  double r;
  const bool plain = fn == 0 || fn == 2;
  if ((plain && nans) || n == 0) {
    r = __builtin_nan("");
  } else if (fn >= 2) {
    r = wave_sum_d(s) / (double)n;
  } else {
    // the order statistics of rank (n-1)/2 and n/2 by counting: a value v has rank r iff
    // #(x < v) <= r < #(x < v) + #(x == v)
    const int r0 = (n - 1) / 2, r1 = n / 2;
    double m0 = 0.0, m1 = 0.0;
    bool f0 = false, f1 = false;
    for (int c = lane; c < n; c += 64) {
      const double v = buf[c];
      int less = 0, eq = 0;
      for (int q = 0; q < n; q++) {
        const double x = buf[q];
        less += x < v ? 1 : 0;
        eq += x == v ? 1 : 0;
      }
      if (less <= r0 && r0 < less + eq) { m0 = v; f0 = true; }
      if (less <= r1 && r1 < less + eq) { m1 = v; f1 = true; }
    }
    // any lane that found the rank holds the same value
    const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
    m0 = __shfl(m0, __ffsll((long long)b0) - 1, 64);
    m1 = __shfl(m1, __ffsll((long long)b1) - 1, 64);
    r = (n & 1) ? m0 : (m0 + m1) / 2.0;
  }
  if (lane == 0) out[(long)ii * n1 + jj] = r;
}

// ------------------------------------------------------------- host: coefficient tables --
static void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

// OpenCV's interpolateLanczos4: float coefficients from double sines, normalised in float
static void lanczos4_coeffs(float x, float* c) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0},  {-s45, -s45}, {0, 1},  {s45, -s45},
                                 {-1, 0}, {s45, s45},   {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; i++) c[i] = 0;
    c[3] = 1;
    return;
  }
  float sum = 0;
  const double y0 = -(x + 3) * M_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
  for (int i = 0; i < 8; i++) {
    const double y = -(x + 3 - i) * M_PI * 0.25;
    c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += c[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; i++) c[i] *= sum;
}

static void axis_tables(int ssize, int dsize, double scale, int interp, int ks, bool clamp_x,
                        std::vector<int>& ofs, std::vector<float>& coef, int* pmax) {
  ofs.resize(dsize);
  coef.resize((size_t)dsize * ks);
  int xmax = dsize;
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (clamp_x) {
      if (s < 0 && interp == IPA_RESIZE_LINEAR) { f = 0; s = 0; }
      if (s + ks / 2 >= ssize) {
        xmax = xmax < d ? xmax : d;
        if (s >= ssize - 1 && interp == IPA_RESIZE_LINEAR) { f = 0; s = ssize - 1; }
      }
    }
    ofs[d] = s;
    float* c = coef.data() + (size_t)d * ks;
    if (interp == IPA_RESIZE_CUBIC) cubic_coeffs(f, c);
    else if (interp == IPA_RESIZE_LANCZOS4) lanczos4_coeffs(f, c);
    else { c[0] = 1.f - f; c[1] = f; }
  }
  if (pmax) *pmax = xmax;
}

static void area_tables(int ssize, int dsize, double scale, std::vector<AreaTab>& tab,
                        std::vector<int>& start) {
  tab.clear();
  start.assign(dsize + 1, 0);
  for (int dx = 0; dx < dsize; dx++) {
    start[dx] = (int)tab.size();
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    if (sx1 - fsx1 > 1e-3) tab.push_back(AreaTab{sx1 - 1, dx, (float)((sx1 - fsx1) / cell)});
    for (int sx = sx1; sx < sx2; sx++) tab.push_back(AreaTab{sx, dx, (float)(1.0 / cell)});
    if (fsx2 - sx2 > 1e-3) {
      double a = fsx2 - sx2 < 1.0 ? fsx2 - sx2 : 1.0;
      a = a < cell ? a : cell;
      tab.push_back(AreaTab{sx2, dx, (float)(a / cell)});
    }
  }
  start[dsize] = (int)tab.size();
}

template <typename T>
static void launch_separable(ipa_ctx* ctx, int ks, const T* src, long spitch, int sh, int sw, T* tmp,
                             T* dst, long dpitch, int dh, int dw, const int* xofs,
                             const float* alpha, int xmax, const int* yofs, const float* beta) {
  dim3 block(256), gh((unsigned)((dw + 255) / 256), (unsigned)sh), gv((unsigned)((dw + 255) / 256), (unsigned)dh);
  dim3 gv4((unsigned)((dw / 4 + 255) / 256), (unsigned)dh);
  const bool vec4 = dw % 4 == 0 && dpitch % 4 == 0 && (uintptr_t)dst % (4 * sizeof(T)) == 0 &&
                    (uintptr_t)tmp % (4 * sizeof(T)) == 0;
#define IPA_RS(KS)                                                                                 \
  hipLaunchKernelGGL((hresize_kernel<T, KS>), gh, block, 0, ctx->stream, src, spitch, sh, sw, tmp,  \
                     dw, xofs, alpha, xmax);                                                        \
  if (vec4)                                                                                        \
    hipLaunchKernelGGL((vresize4_kernel<T, KS>), gv4, block, 0, ctx->stream, (const T*)tmp, sh, dw,  \
                       dst, dpitch, yofs, beta);                                                    \
  else                                                                                             \
    hipLaunchKernelGGL((vresize_kernel<T, KS>), gv, block, 0, ctx->stream, (const T*)tmp, sh, dw,   \
                       dst, dpitch, yofs, beta)
  if (ks == 2) { IPA_RS(2); } else if (ks == 4) { IPA_RS(4); } else { IPA_RS(8); }
#undef IPA_RS
}

}  // namespace ipa

using namespace ipa;

extern "C" {

int ipa_resize_dev(ipa_ctx* ctx, const void* d_src, int dtype, int sh, int sw, long src_pitch,
                   void* d_dst, int dh, int dw, long dst_pitch, int interp) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_src && d_dst, "null pointer");
  IPA_REQUIRE(ctx, sh > 0 && sw > 0 && dh > 0 && dw > 0 && src_pitch >= sw && dst_pitch >= dw &&
                       dh <= 65535 && sh <= 65535, "bad shape");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "resize is built for float32 / float64 images (cv2's 8-bit fixed-point "
                         "paths differ between OpenCV versions); got dtype %d", dtype);
  const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
  const size_t es = ipa_dtype_size(dtype);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  dim3 block(256), grid((unsigned)((dw + 255) / 256), (unsigned)dh);
  // OpenCV's rule (resize.cpp): INTER_LINEAR at an exact 2 x 2 reduction IS the area average
  // ("interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2")
  if (interp == IPA_RESIZE_LINEAR && sw == 2 * dw && sh == 2 * dh) interp = IPA_RESIZE_AREA;
  if (interp != IPA_RESIZE_LINEAR && interp != IPA_RESIZE_CUBIC && interp != IPA_RESIZE_AREA &&
      interp != IPA_RESIZE_LANCZOS4)
    IPA_UNSUPPORTED(ctx, "resize: interpolation %d is not built (linear 1, cubic 2, area 3, "
                         "lanczos4 4; INTER_NEAREST and the exact / bit-exact variants are not)", interp);
  if (interp == IPA_RESIZE_AREA) {
    if (!(scale_x >= 1 && scale_y >= 1))
      IPA_UNSUPPORTED(ctx, "INTER_AREA is built for downscaling (OpenCV switches to a bilinear "
                           "variant when enlarging)");
    const int isx = (int)nearbyint(scale_x), isy = (int)nearbyint(scale_y);
    if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON) {
      if (dtype == IPA_F32)
        hipLaunchKernelGGL((area_fast_kernel<float>), grid, block, 0, ctx->stream,
                           (const float*)d_src, src_pitch, sh, sw, (float*)d_dst, dst_pitch, dh, dw,
                           isx, isy);
      else
        hipLaunchKernelGGL((area_fast_kernel<double>), grid, block, 0, ctx->stream,
                           (const double*)d_src, src_pitch, sh, sw, (double*)d_dst, dst_pitch, dh,
                           dw, isx, isy);
      IPA_HIP(ctx, hipGetLastError());
      return IPA_OK;
    }
    std::vector<AreaTab> xt, yt;
    std::vector<int> xs, ys;
    area_tables(sw, dw, scale_x, xt, xs);
    area_tables(sh, dh, scale_y, yt, ys);
    const size_t b0 = up(xt.size() * sizeof(AreaTab)), b1 = up(xs.size() * 4),
                 b2 = up(yt.size() * sizeof(AreaTab)), b3 = up(ys.size() * 4);
    std::vector<char> blob(b0 + b1 + b2 + b3);
    memcpy(blob.data(), xt.data(), xt.size() * sizeof(AreaTab));
    memcpy(blob.data() + b0, xs.data(), xs.size() * 4);
    memcpy(blob.data() + b0 + b1, yt.data(), yt.size() * sizeof(AreaTab));
    memcpy(blob.data() + b0 + b1 + b2, ys.data(), ys.size() * 4);
    void* d = nullptr;
    int rc = ipa_tab_upload(ctx, blob.data(), blob.size(), &d);
    if (rc) return rc;
    const char* t = (const char*)d;
    if (dtype == IPA_F32)
      hipLaunchKernelGGL((area_kernel<float>), grid, block, 0, ctx->stream, (const float*)d_src,
                         src_pitch, (float*)d_dst, dst_pitch, dw, (const AreaTab*)t,
                         (const int*)(t + b0), (const AreaTab*)(t + b0 + b1),
                         (const int*)(t + b0 + b1 + b2));
    else
      hipLaunchKernelGGL((area_kernel<double>), grid, block, 0, ctx->stream, (const double*)d_src,
                         src_pitch, (double*)d_dst, dst_pitch, dw, (const AreaTab*)t,
                         (const int*)(t + b0), (const AreaTab*)(t + b0 + b1),
                         (const int*)(t + b0 + b1 + b2));
    IPA_HIP(ctx, hipGetLastError());
    return IPA_OK;
  }
  if (interp != IPA_RESIZE_LINEAR && interp != IPA_RESIZE_CUBIC && interp != IPA_RESIZE_LANCZOS4)
    IPA_UNSUPPORTED(ctx, "resize: interpolation %d (INTER_LINEAR 1, INTER_CUBIC 2, INTER_AREA 3, "
                         "INTER_LANCZOS4 4 are built)", interp);
  const int ks = interp == IPA_RESIZE_LINEAR ? 2 : (interp == IPA_RESIZE_CUBIC ? 4 : 8);
  // the tables of the last resize are still on the device (a sequence of frames of one shape:
  // building OpenCV's Lanczos4 coefficients for an 8K result costs the host ~0.2 ms)
  const long key[5] = {sw, dw, sh, dh, interp};
  const size_t b0 = up((size_t)dw * 4), b1 = up((size_t)dw * ks * 4), b2 = up((size_t)dh * 4),
               b3 = up((size_t)dh * ks * 4);
  int xmax = dw;
  void* d = nullptr;
  int rc = 0;
  if (ctx->tab && ctx->resize_serial == ctx->tab_serial && ctx->resize_serial &&
      memcmp(key, ctx->resize_key, sizeof key) == 0) {
    d = ctx->tab;
    xmax = ctx->resize_xmax;
  } else {
    std::vector<int> xofs, yofs;
    std::vector<float> alpha, beta;
    axis_tables(sw, dw, scale_x, interp, ks, true, xofs, alpha, &xmax);
    axis_tables(sh, dh, scale_y, interp, ks, false, yofs, beta, nullptr);
    std::vector<char> blob(b0 + b1 + b2 + b3);
    memcpy(blob.data(), xofs.data(), xofs.size() * 4);
    memcpy(blob.data() + b0, alpha.data(), alpha.size() * 4);
    memcpy(blob.data() + b0 + b1, yofs.data(), yofs.size() * 4);
    memcpy(blob.data() + b0 + b1 + b2, beta.data(), beta.size() * 4);
    ctx->resize_serial = 0;
    rc = ipa_tab_upload(ctx, blob.data(), blob.size(), &d);
    if (rc) return rc;
    memcpy(ctx->resize_key, key, sizeof key);
    ctx->resize_serial = ctx->tab_serial;
    ctx->resize_xmax = xmax;
  }
  rc = ipa_plan_reserve(ctx, (size_t)sh * dw * es);   // the horizontally resized rows
  if (rc) return rc;
  const char* t = (const char*)d;
  if (dtype == IPA_F32)
    launch_separable<float>(ctx, ks, (const float*)d_src, src_pitch, sh, sw, (float*)ctx->plan,
                            (float*)d_dst, dst_pitch, dh, dw, (const int*)t, (const float*)(t + b0),
                            xmax, (const int*)(t + b0 + b1), (const float*)(t + b0 + b1 + b2));
  else
    launch_separable<double>(ctx, ks, (const double*)d_src, src_pitch, sh, sw, (double*)ctx->plan,
                             (double*)d_dst, dst_pitch, dh, dw, (const int*)t,
                             (const float*)(t + b0), xmax, (const int*)(t + b0 + b1),
                             (const float*)(t + b0 + b1 + b2));
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_resize(ipa_ctx* ctx, const void* src, int dtype, int sh, int sw, void* dst, int dh, int dw,
               int interp) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0, "bad arguments");
  const size_t es = ipa_dtype_size(dtype);
  IPA_REQUIRE(ctx, es, "unknown dtype");
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t sb = (size_t)sh * sw * es, db = (size_t)dh * dw * es;
  int rc = ipa_ws_reserve(ctx, up(sb) + up(db));
  if (rc) return rc;
  char* ds = (char*)ctx->ws;
  char* dd = ds + up(sb);
  IPA_HIP(ctx, hipMemcpyAsync(ds, src, sb, hipMemcpyHostToDevice, ctx->stream));
  rc = ipa_resize_dev(ctx, ds, dtype, sh, sw, sw, dd, dh, dw, dw, interp);
  if (rc) return rc;
  IPA_HIP(ctx, hipMemcpyAsync(dst, dd, db, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

int ipa_fast_filter_stat_dev(ipa_ctx* ctx, const void* d_arr, int dtype, int h, int w, long pitch,
                             int ksize, int every, int fn, double* d_out) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_arr && d_out, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && pitch >= w && ksize >= 1 && every >= 1 && fn >= 0 && fn <= 3,
              "bad arguments");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "fast_filter_stat supports float32/float64 arrays (got dtype %d)", dtype);
  const long per_axis = (2L * ksize + every - 1) / every;
  if (per_axis * per_axis > kStatMax)
    IPA_UNSUPPORTED(ctx, "fast_filter_stat: %ld x %ld window samples exceed the %d a wave keeps in "
                         "LDS (raise `every`)", per_axis, per_axis, kStatMax);
  const int n0 = (h + every - 1) / every, n1 = (w + every - 1) / every;
  IPA_REQUIRE(ctx, n0 <= 65535, "too many rows of cells");
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  dim3 grid((unsigned)n1, (unsigned)n0), block(64);
  if (dtype == IPA_F32)
    hipLaunchKernelGGL((fast_filter_stat_kernel<float>), grid, block, 0, ctx->stream,
                       (const float*)d_arr, pitch, h, w, ksize, every, fn, d_out, n1);
  else
    hipLaunchKernelGGL((fast_filter_stat_kernel<double>), grid, block, 0, ctx->stream,
                       (const double*)d_arr, pitch, h, w, ksize, every, fn, d_out, n1);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_fast_filter_stat(ipa_ctx* ctx, const void* arr, int dtype, int h, int w, int ksize,
                         int every, int fn, double* out) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, arr && out && h > 0 && w > 0 && every >= 1, "bad arguments");
  const size_t es = ipa_dtype_size(dtype);
  IPA_REQUIRE(ctx, es, "unknown dtype");
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t sb = (size_t)h * w * es;
  const size_t ob = (size_t)((h + every - 1) / every) * ((w + every - 1) / every) * 8;
  int rc = ipa_ws_reserve(ctx, up(sb) + up(ob));
  if (rc) return rc;
  char* ds = (char*)ctx->ws;
  double* dd = (double*)(ds + up(sb));
  IPA_HIP(ctx, hipMemcpyAsync(ds, arr, sb, hipMemcpyHostToDevice, ctx->stream));
  rc = ipa_fast_filter_stat_dev(ctx, ds, dtype, h, w, w, ksize, every, fn, dd);
  if (rc) return rc;
  IPA_HIP(ctx, hipMemcpyAsync(out, dd, ob, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

}  // extern "C"
