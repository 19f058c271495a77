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
//   ipa_closest_distance*  render/closestDirectDistance.py:17-41
//   ipa_pos_intensity_unc* uncertainty/positionToIntensityUncertainty.py:7-49
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

// The same correlation with the source window of a 64 x 4 output block staged in LDS once:
// the border mode is resolved per tile element ((4+k0-1) x (64+k1-1) of them) instead of twice
// per tap and pixel, and the taps are LDS reads.  Summation order as above -> identical results.
template <typename T>
__global__ void __launch_bounds__(256)
conv_ydep_tile_kernel(const T* __restrict__ src, int h, int w, long spitch,
                      const double* __restrict__ kernels, int k0, int k1, int bx, int by,
                      T* __restrict__ dst, long dpitch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ydep_lds[];
  T* tile = reinterpret_cast<T*>(ydep_lds);
  const int tw = 64 + k1 - 1, th = 4 + k0 - 1;
  const int x0 = blockIdx.x * 64 - k1 / 2, y0 = blockIdx.y * 4 - k0 / 2;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int e = tid; e < tw * th; e += 256) {
    const int ty = e / tw, tx = e - ty * tw;
    const int yy = resolve_idx(y0 + ty, h, by), xx = resolve_idx(x0 + tx, w, bx);
    tile[e] = (yy < 0 || xx < 0) ? (T)0 : src[(long)yy * spitch + xx];
  }
  __syncthreads();
  const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y * 4 + threadIdx.y;
  if (c >= w || r >= h) return;
  const double* kr = kernels + (long)r * k0 * k1;  // wave-uniform
  const T* tp = tile + threadIdx.y * tw + threadIdx.x;
  double v = 0.0;
  for (int ii = 0; ii < k0; ii++)
    for (int jj = 0; jj < k1; jj++) {
      const double a = (double)tp[ii * tw + jj];
      if (a == a) v += kr[ii * k1 + jj] * a;  // NaN-aware: skip, no renormalisation
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

// The same reduction with the (4 + 2 hkx) x (64 + 2 hky) source window of a 64 x 4 output block
// staged in LDS; the loops still visit exactly the clipped window, in the same order.
template <typename T>
__global__ void __launch_bounds__(256)
local_std_tile_kernel(const T* __restrict__ img, const T* __restrict__ blurred, int gx, int gy,
                      long pitch, long bpitch, int hkx, int hky, T* __restrict__ out,
                      long opitch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char std_lds[];
  T* tile = reinterpret_cast<T*>(std_lds);
  const int tw = 64 + 2 * hky, th = 4 + 2 * hkx;
  const int j0 = blockIdx.x * 64 - hky, i0 = blockIdx.y * 4 - hkx;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int e = tid; e < tw * th; e += 256) {
    const int ty = e / tw, tx = e - ty * tw;
    int ii = i0 + ty, jj = j0 + tx;
    ii = ii < 0 ? 0 : (ii >= gx ? gx - 1 : ii);  // outside the image: never read back
    jj = jj < 0 ? 0 : (jj >= gy ? gy - 1 : jj);
    tile[e] = img[(long)ii * pitch + jj];
  }
  __syncthreads();
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= gx || j >= gy) return;
  int xmn = i - hkx < 0 ? 0 : i - hkx, xmx = i + hkx > gx ? gx : i + hkx;
  int ymn = j - hky < 0 ? 0 : j - hky, ymx = j + hky > gy ? gy : j + hky;
  double mean = (double)blurred[(long)i * bpitch + j], val = 0.0;
  for (int ii = xmn; ii < xmx; ii++)
    for (int jj = ymn; jj < ymx; jj++) {
      double d = (double)tile[(ii - i0) * tw + (jj - j0)] - mean;
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

// The fill case (pixels with mask != 0 are usually sparse, each costs up to ksize^2 visits) with
// the WAVE as the unit of work, like the IDW kernels: a wave owns 64 consecutive pixels of a
// row, ballots the masked ones and spreads its 64 lanes over the window of one masked pixel at
// a time (coalesced row reads), then reduces sum and count with shuffles.  The float64 sum is
// formed in a different order than the reference's loop (relative difference ~1e-16).
template <typename T>
__global__ void __launch_bounds__(256)
masked_mean_fill_wave_kernel(T* grid, const unsigned char* __restrict__ mask, int gx, int gy,
                             long pitch, long mpitch, int k, int segs_x) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long seg = (long)blockIdx.x * 4 + wave;
  const int i = (int)(seg / segs_x);
  if (i >= gx) return;
  const int js = (int)(seg - (long)i * segs_x) * 64;
  const int jl = js + lane;
  unsigned long long todo = __ballot(jl < gy && mask[(long)i * mpitch + jl] != 0);
  const int xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
  while (todo) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int j = js + b;
    const int ymn = j - k < 0 ? 0 : j - k, ymx = j + k > gy ? gy : j + k;
    const int ww = ymx - ymn, ntap = (xmx - xmn) * ww;
    double val = 0.0;
    int n = 0;
    if (ww <= 64) {
      // lanes over the columns of a window row (two rows per pass when the window is at most
      // 32 wide): no per-tap division, and mask + value of EIGHT rows are loaded back to back
      // before any of them is looked at - the loop was bound by the latency of one dependent
      // mask -> value load pair per pass
      const bool two = ww <= 32;
      const int half = two ? lane >> 5 : 0, step = two ? 2 : 1;
      const int cl = two ? (lane & 31) : lane;
      const bool col_ok = cl < ww;
      const int jj = ymn + (col_ok ? cl : 0);
      for (int ii = xmn + half; ii < xmx; ii += 8 * step) {
        unsigned char m[8];
        T g[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int r = ii + u * step < xmx ? ii + u * step : xmx - 1;
          m[u] = mask[(long)r * mpitch + jj];
          g[u] = grid[(long)r * pitch + jj];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (col_ok && ii + u * step < xmx && !m[u]) {
            val += (double)g[u];
            n++;
          }
        }
      }
    } else {
      for (int t = lane; t < ntap; t += 64) {
        const int dy = t / ww, ii = xmn + dy, jj = ymn + (t - dy * ww);
        if (!mask[(long)ii * mpitch + jj]) {
          val += (double)grid[(long)ii * pitch + jj];
          n++;
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      val += __shfl_xor(val, off, 64);
      n += __shfl_xor(n, off, 64);
    }
    if (lane == 0 && n > 0) grid[(long)i * pitch + j] = (T)(val / (double)n);
  }
}

// The same fill for windows up to 64 columns wide (k <= 32): the masked pixels of a 64-px
// segment share their window rows and overlap in columns, so the wave first forms, per column
// of the segment's 64 + 2k columns, the sum and count of the unmasked pixels over the window
// rows - two columns per lane, every load independent of every other - and parks them in its
// LDS slice; a masked pixel then sums its <= 2k column entries with one LDS read per lane and a
// shuffle reduction.  Per segment 4 loads per window row instead of 2 dependent passes per
// masked pixel (4K, 5 % masked, k = 15: 332 -> see profiles/r02_micro.txt).
template <typename T>
__global__ void __launch_bounds__(256)
masked_mean_fill_cols_kernel(T* grid, const unsigned char* __restrict__ mask, int gx, int gy,
                             long pitch, long mpitch, int k, int segs_x) {
  __shared__ double cs[4][128];
  __shared__ int cn[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long seg = (long)blockIdx.x * 4 + wave;
  const int i = (int)(seg / segs_x);
  if (i >= gx) return;
  const int js = (int)(seg - (long)i * segs_x) * 64;
  const int jl = js + lane;
  unsigned long long todo = __ballot(jl < gy && mask[(long)i * mpitch + jl] != 0);
  if (!todo) return;
  const int xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
  const int c0 = js - k;
  const int ca = c0 + lane, cb = c0 + 64 + lane;
  const bool oka = ca >= 0 && ca < gy, okb = cb < js + 64 + k && cb >= 0 && cb < gy;
  const long oa = oka ? ca : 0, ob = okb ? cb : 0;
  double sa = 0.0, sb = 0.0;
  int na = 0, nb = 0;
  for (int ii = xmn; ii < xmx; ii += 4) {
    unsigned char ma[4], mb[4];
    T ga[4], gb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = ii + u < xmx ? ii + u : xmx - 1;
      ma[u] = mask[(long)r * mpitch + oa];
      mb[u] = mask[(long)r * mpitch + ob];
      ga[u] = grid[(long)r * pitch + oa];
      gb[u] = grid[(long)r * pitch + ob];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool live = ii + u < xmx;
      if (live && oka && !ma[u]) { sa += (double)ga[u]; na++; }
      if (live && okb && !mb[u]) { sb += (double)gb[u]; nb++; }
    }
  }
  cs[wave][lane] = sa; cs[wave][64 + lane] = sb;
  cn[wave][lane] = na; cn[wave][64 + lane] = nb;
  __builtin_amdgcn_wave_barrier();
  while (todo) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int j = js + b;
    const int ymn = j - k < 0 ? 0 : j - k, ymx = j + k > gy ? gy : j + k;
    const int e = ymn - c0 + lane;
    const bool in = lane < ymx - ymn;
    double val = in ? cs[wave][e] : 0.0;
    int n = in ? cn[wave][e] : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      val += __shfl_xor(val, off, 64);
      n += __shfl_xor(n, off, 64);
    }
    if (lane == 0 && n > 0) grid[(long)i * pitch + j] = (T)(val / (double)n);
  }
}

// filters/maskedFilter.py:76-102 (_calcMedian): np.median of the mask == 0 pixels of the clipped
// window, wave-cooperative like the fill above.  The wave collects the window values as
// order-preserving integer keys in its LDS buffer (ballot-ranked append), then finds the middle
// order statistic(s) by a most-significant-bit-first radix descent: per bit one counting pass
// over the buffer and one shuffle reduction.  Pure selection: bit-identical to sorting.
template <typename T> struct key_of;
template <> struct key_of<float> {
  using type = unsigned;
  static __device__ __forceinline__ unsigned enc(float v) {
    unsigned u = __float_as_uint(v);
    return (u >> 31) ? ~u : (u | 0x80000000u);
  }
  static __device__ __forceinline__ float dec(unsigned k) {
    return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k);
  }
};
template <> struct key_of<double> {
  using type = unsigned long long;
  static __device__ __forceinline__ unsigned long long enc(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
  }
  static __device__ __forceinline__ double dec(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
  }
};

// Order statistic k (0-based) of the n keys in `cur`, and - for the median of an even count - the
// next one (rank k + 1) in *next.  MSB-first radix descent over the bits in which the keys differ
// at all, with the candidates PARTITIONED in every pass: one sweep writes the keys with a 0 at
// the bit to the front of the other buffer and those with a 1 to its back, the count of zeros
// (ballots, no shuffle reduction) picks the half that holds rank k, and the next pass sweeps only
// that half.  About 2 n key visits in all instead of 32 n (64 n with both middle ranks): the
// default maskedFilter (median, k = 15, 5 % masked) 4.8 -> see profiles/r02_micro.txt.
// The next rank is the answer again when it has duplicates left, else the smallest key that
// was discarded as greater.  Pure selection: bit-identical to sorting.
template <typename K>
__device__ __forceinline__ K wave_select2(K* cur, K* oth, int n, int k, int lane, K* next) {
  constexpr int BITS = sizeof(K) * 8;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  // bits in which the keys differ
  K all_or = 0, all_and = ~(K)0;
  for (int t = lane; t < n; t += 64) {
    const K x = cur[t];
    all_or |= x;
    all_and &= x;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    all_or |= __shfl_xor(all_or, off, 64);
    all_and &= __shfl_xor(all_and, off, 64);
  }
  K diff = all_or ^ all_and;
  K greater = ~(K)0;   // per lane: smallest discarded key above the answer
  bool any_greater = false;
  int na = n;
  while (diff != 0 && na > 1) {
    const int b = BITS - 1 - (sizeof(K) == 8 ? __builtin_clzll((unsigned long long)diff)
                                             : __builtin_clz((unsigned)diff));
    diff &= ~((K)1 << b);
    int zeros = 0;
    K ones_min = ~(K)0;
    bool ones_any = false;
    for (int base = 0; base < na; base += 64) {
      const int t = base + lane;
      const bool live = t < na;
      const K x = live ? cur[t] : (K)0;
      const bool one = live && ((x >> b) & 1);
      const bool zero = live && !one;
      const unsigned long long zm = __ballot(zero), om = __ballot(one);
      const int ones_before = base - zeros;  // ones written by earlier iterations
      if (zero) oth[zeros + __popcll(zm & below)] = x;
      if (one) {
        oth[na - 1 - (ones_before + __popcll(om & below))] = x;
        ones_min = x < ones_min ? x : ones_min;
        ones_any = true;
      }
      zeros += __popcll(zm);
    }
    __builtin_amdgcn_wave_barrier();
    K* nxt = oth;
    if (k < zeros) {  // the answer has a 0 here: the ones are all greater
      if (ones_any) {
        greater = ones_min < greater ? ones_min : greater;
        any_greater = true;
      }
      na = zeros;
    } else {          // it has a 1: the zeros are all smaller
      k -= zeros;
      nxt = oth + zeros;
      na -= zeros;
    }
    oth = cur;
    cur = nxt;
    // (the buffers swap roles: the region the next pass writes is the one read two passes ago,
    // which held at least as many keys as are left now)
  }
  const K ans = cur[0];  // every remaining key is equal (no differing bit left) or na == 1
  if (next) {
    K g = any_greater ? greater : ~(K)0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const K o = __shfl_xor(g, off, 64);
      g = o < g ? o : g;
    }
    *next = (k + 1 < na) ? ans : g;
  }
  return ans;
}

template <typename T, bool FILL>
__global__ void __launch_bounds__(256)
masked_median_wave_kernel(const T* src, const unsigned char* __restrict__ mask, int gx, int gy,
                          long pitch, long mpitch, int k, int cap, int segs_x, T* dst,
                          long dpitch) {
  using KT = typename key_of<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char median_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  KT* keys = reinterpret_cast<KT*>(median_lds) + (long)wave * 2 * cap;
  KT* keys2 = keys + cap;
  const long seg = (long)blockIdx.x * 4 + wave;
  const int i = (int)(seg / segs_x);
  if (i >= gx) return;
  const int js = (int)(seg - (long)i * segs_x) * 64;
  const int jl = js + lane;
  const bool inside = jl < gy;
  const bool masked = inside && mask[(long)i * mpitch + jl] != 0;
  if (!FILL && masked) dst[(long)i * dpitch + jl] = (T)__builtin_nan("");
  unsigned long long todo = __ballot(inside && masked == FILL);
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
  while (todo) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int j = js + b;
    const int ymn = j - k < 0 ? 0 : j - k, ymx = j + k > gy ? gy : j + k;
    const int ww = ymx - ymn, ntap = (xmx - xmn) * ww;
    int n = 0;
    bool has_nan = false;
    if (ww <= 64) {
      // lanes over the columns of a window row (two rows per pass when the window is at most
      // 32 wide); mask and value of 8 passes are loaded back to back, the value whether or not
      // it is masked: no division per tap, no dependent load pair per pass
      const bool two = ww <= 32;
      const int half = two ? lane >> 5 : 0, step = two ? 2 : 1;
      const int cl = two ? (lane & 31) : lane;
      const bool col_ok = cl < ww;
      const int jj = ymn + (col_ok ? cl : 0);
      for (int ii = xmn + half; ii - half < xmx; ii += 8 * step) {
        unsigned char m[8];
        T g[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int r = ii + u * step < xmx ? ii + u * step : xmx - 1;
          m[u] = mask[(long)r * mpitch + jj];
          g[u] = src[(long)r * pitch + jj];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const bool use = col_ok && ii + u * step < xmx && !m[u];
          const unsigned long long um = __ballot(use);
          has_nan |= __ballot(use && g[u] != g[u]) != 0ull;
          if (use) keys[n + __popcll(um & below)] = key_of<T>::enc(g[u]);
          n += __popcll(um);
        }
      }
    } else {
      for (int base = 0; base < ntap; base += 64) {
        const int t = base + lane;
        bool use = false;
        T v = (T)0;
        if (t < ntap) {
          const int dy = t / ww, ii = xmn + dy, jj = ymn + (t - dy * ww);
          use = mask[(long)ii * mpitch + jj] == 0;
          if (use) v = src[(long)ii * pitch + jj];
        }
        const unsigned long long um = __ballot(use);
        has_nan |= __ballot(use && v != v) != 0ull;
        if (use) keys[n + __popcll(um & below)] = key_of<T>::enc(v);
        n += __popcll(um);
      }
    }
    if (n == 0) continue;
    __builtin_amdgcn_wave_barrier();  // wave-private buffer: in-order ds_write / ds_read
    T med;
    if (has_nan) {
      med = (T)__builtin_nan("");
    } else {
      KT nk;
      const T a = key_of<T>::dec(wave_select2<KT>(keys, keys2, n, (n - 1) / 2, lane, &nk));
      const T c = (n & 1) ? a : key_of<T>::dec(nk);
      med = (a + c) * (T)0.5;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) dst[(long)i * dpitch + j] = med;
  }
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

// The same in two separable passes through LDS (nanmax over a rectangle = nanmax over its rows
// of the row-wise nanmax; the scan order row-major, first maximum kept, is preserved): a block
// of 64 x RB outputs stages its (RB + 2k) x (64 + 2k) source tile once, reduces every tile row
// horizontally into a second plane, then every output column vertically.  2k + 2k LDS reads per
// output instead of 4 k^2 global loads (4K frame, ksize 7: 150 -> see profiles/r02_micro.txt).
template <typename T>
__global__ void __launch_bounds__(256)
nan_max_sep_kernel(const T* __restrict__ src, int gx, int gy, long pitch, int k, int rb,
                   T* __restrict__ dst, long dpitch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nanmax_lds[];
  const int tw = 64 + 2 * k, th = rb + 2 * k;
  T* tile = reinterpret_cast<T*>(nanmax_lds);  // th x tw source values
  T* hmax = tile + th * tw;                    // th x 64 row-wise maxima
  const int j0 = blockIdx.x * 64, i0 = blockIdx.y * rb;
  const T nanv = (T)__builtin_nan("");
  for (int e = threadIdx.x; e < th * tw; e += 256) {
    const int ty = e / tw, tx = e - ty * tw;
    const int ii = i0 - k + ty, jj = j0 - k + tx;
    tile[e] = (ii >= 0 && ii < gx && jj >= 0 && jj < gy) ? src[(long)ii * pitch + jj] : nanv;
  }
  __syncthreads();
  // row-wise: output column j0 + c covers source columns [j - k, j + k) = tile columns [c, c + 2k)
  for (int e = threadIdx.x; e < th * 64; e += 256) {
    const int ty = e >> 6, c = e & 63;
    const T* tp = tile + ty * tw + c;
    T m = nanv;
    for (int q = 0; q < 2 * k; q++) {
      const T v = tp[q];
      if (v == v && !(m >= v)) m = v;
    }
    hmax[e] = m;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < rb * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const int i = i0 + r, j = j0 + c;
    if (i >= gx || j >= gy) continue;
    T m = nanv;
    for (int q = 0; q < 2 * k; q++) {  // source rows [i - k, i + k) = tile rows [r, r + 2k)
      const T v = hmax[(r + q) * 64 + c];
      if (v == v && !(m >= v)) m = v;
    }
    dst[(long)i * dpitch + j] = m;
  }
}

// render/closestDirectDistance.py:17-41: distance to the closest non-zero pixel within the
// +-ksize window (centre excluded), 2*ksize when there is none, 0 on non-zero pixels.  The
// minimum is taken over the integer squared distances; one sqrt at the end gives the same
// float64 as the reference's running minimum of sqrt values.
template <typename OT>
__global__ void __launch_bounds__(256)
closest_distance_kernel(const unsigned char* __restrict__ arr, int s0, int s1, long pitch,
                        int ksize, OT* __restrict__ out, long opitch) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= s0 || j >= s1) return;
  double md = 0.0;
  if (!arr[(long)i * pitch + j]) {
    const int big = 0x7fffffff;
    int best = big;
    const int xmn = i - ksize < 0 ? 0 : i - ksize, xmx = i + ksize >= s0 ? s0 - 1 : i + ksize;
    const int ymn = j - ksize < 0 ? 0 : j - ksize, ymx = j + ksize >= s1 ? s1 - 1 : j + ksize;
    for (int xi = xmn; xi <= xmx; xi++)
      for (int yi = ymn; yi <= ymx; yi++)
        if (arr[(long)xi * pitch + yi]) {
          const int d2 = (xi - i) * (xi - i) + (yi - j) * (yi - j);
          best = d2 < best ? d2 : best;
        }
    md = 2.0 * (double)ksize;
    if (best != big) {
      const double d = sqrt((double)best);
      if (d < md) md = d;
    }
  }
  out[(long)i * opitch + j] = (OT)md;  // uint16: truncation, like numba's store
}

// The same minimum in two passes: g(i, j) = distance along row i from column j to the nearest
// non-zero pixel within +-ksize (255 = none), then d^2(i, j) = min over the rows i + dy of
// dy^2 + g(i + dy, j)^2 - the minimum over a row of dy^2 + dx^2 is reached at the smallest |dx|.
// 2 ksize + 1 byte reads per pixel and pass instead of (2 ksize + 1)^2; integer squared
// distances, so the result is the same number (4K frame, ksize 30: 6.8 ms -> see r02_micro.txt).
__global__ void __launch_bounds__(256)
closest_rowdist_kernel(const unsigned char* __restrict__ arr, int s0, int s1, long pitch, int ksize,
                       unsigned char* __restrict__ g) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= s0 || j >= s1) return;
  const unsigned char* row = arr + (long)i * pitch;
  int best = 255;
  for (int d = 0; d <= ksize; d++) {
    const bool l = j - d >= 0 && row[j - d] != 0, r = j + d < s1 && row[j + d] != 0;
    if (l || r) {
      best = d;
      break;
    }
  }
  g[(long)i * s1 + j] = (unsigned char)best;
}

template <typename OT>
__global__ void __launch_bounds__(256)
closest_coldist_kernel(const unsigned char* __restrict__ arr, const unsigned char* __restrict__ g,
                       int s0, int s1, long pitch, int ksize, OT* __restrict__ out, long opitch) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= s0 || j >= s1) return;
  double md = 0.0;
  if (!arr[(long)i * pitch + j]) {
    const int big = 0x7fffffff;
    int best = big;
    const int xmn = i - ksize < 0 ? 0 : i - ksize, xmx = i + ksize >= s0 ? s0 - 1 : i + ksize;
    for (int xi = xmn; xi <= xmx; xi++) {
      const int gd = g[(long)xi * s1 + j];
      if (gd != 255) {
        const int d2 = (xi - i) * (xi - i) + gd * gd;
        best = d2 < best ? d2 : best;
      }
    }
    md = 2.0 * (double)ksize;
    if (best != big) {
      const double d = sqrt((double)best);
      if (d < md) md = d;
    }
  }
  out[(long)i * opitch + j] = (OT)md;  // uint16: truncation, like numba's store
}

// uncertainty/positionToIntensityUncertainty.py:7-49: square root of the PSF-weighted mean of
// the squared differences to the centre pixel.  The Gaussian is equations/numbaGaussian2d.py
// as that file is called there (first sigma on the row axis), evaluated per pixel: the constant
// case then simply repeats the same table.  Pixels closer than ksize to the frame stay 0, NaN
// centres are skipped.
template <typename T>
__global__ void __launch_bounds__(256)
pos_intensity_unc_kernel(const T* __restrict__ img, int s0, int s1, long pitch,
                         const double* __restrict__ sxm, const double* __restrict__ sym,
                         long spitch, double sx0, double sy0, int ksize,
                         double* __restrict__ sint, long opitch) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= s0 || j >= s1) return;
  double res = 0.0;
  const double cpx = (double)img[(long)i * pitch + j];
  if (i >= ksize && i < s0 - ksize && j >= ksize && j < s1 - ksize && cpx == cpx) {
    const double v0 = sxm ? sxm[(long)i * spitch + j] : sx0;
    const double v1 = sym ? sym[(long)i * spitch + j] : sy0;
    const double ss_row = 2 * v0 * v0, ss_col = 2 * v1 * v1;
    const int a = 2 * ksize + 1, c = a / 2;
    double tot = 0.0;
    for (int ii = 0; ii < a; ii++)
      for (int jj = 0; jj < a; jj++)
        tot += exp(-((double)((ii - c) * (ii - c)) / ss_row +
                     (double)((jj - c) * (jj - c)) / ss_col));
    double sdev = 0.0;
    for (int ii = 0; ii < a; ii++)
      for (int jj = 0; jj < a; jj++) {
        const double e = exp(-((double)((ii - c) * (ii - c)) / ss_row +
                               (double)((jj - c) * (jj - c)) / ss_col));
        const double d = (double)img[(long)(i - ii + c) * pitch + (j - jj + c)] - cpx;
        sdev += (e / tot) * (d * d);
      }
    res = sqrt(sdev);
  }
  sint[(long)i * opitch + j] = res;
}

// The same with the Gaussian taken apart: exp(-(a / ssr + b / ssc)) = exp(-a / ssr) exp(-b / ssc),
// so a pixel evaluates 2 (2k + 1) exponentials instead of 2 (2k + 1)^2 - the column factors go to
// the thread's LDS column, the row factor is formed per window row - and multiplies by 1 / tot
// instead of dividing every tap by tot (relative difference ~1e-16 per tap; the tests compare
// at 1e-12).  4K frame, ksize 9: 8.5 ms -> see profiles/r02_micro.txt.
template <typename T>
__global__ void __launch_bounds__(256)
pos_intensity_unc_sep_kernel(const T* __restrict__ img, int s0, int s1, long pitch,
                             const double* __restrict__ sxm, const double* __restrict__ sym,
                             long spitch, double sx0, double sy0, int ksize,
                             double* __restrict__ sint, long opitch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char piu_lds[];
  double* ecol = reinterpret_cast<double*>(piu_lds);  // [2k + 1][256]: column factors per thread
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  if (i >= s0 || j >= s1) return;
  double res = 0.0;
  const double cpx = (double)img[(long)i * pitch + j];
  if (i >= ksize && i < s0 - ksize && j >= ksize && j < s1 - ksize && cpx == cpx) {
    const double v0 = sxm ? sxm[(long)i * spitch + j] : sx0;
    const double v1 = sym ? sym[(long)i * spitch + j] : sy0;
    const double ss_row = 2 * v0 * v0, ss_col = 2 * v1 * v1;
    const int a = 2 * ksize + 1, c = a / 2;
    double csum = 0.0, rsum = 0.0;
    for (int jj = 0; jj < a; jj++) {
      const double e = exp(-((double)((jj - c) * (jj - c)) / ss_col));
      ecol[jj * 256 + tid] = e;
      csum += e;
    }
    for (int ii = 0; ii < a; ii++) rsum += exp(-((double)((ii - c) * (ii - c)) / ss_row));
    const double itot = 1.0 / (rsum * csum);
    double sdev = 0.0;
    for (int ii = 0; ii < a; ii++) {
      const double er = exp(-((double)((ii - c) * (ii - c)) / ss_row)) * itot;
      const T* row = img + (long)(i - ii + c) * pitch + (j + c);
      for (int jj = 0; jj < a; jj++) {
        const double d = (double)row[-jj] - cpx;
        sdev += (er * ecol[jj * 256 + tid]) * (d * d);
      }
    }
    res = sqrt(sdev);
  }
  sint[(long)i * opitch + j] = res;
}

// ---------------------------------------------------------------------------
// 3x3 median + relative threshold (filters/medianThreshold.py:7-30), optionally behind the
// dark-current / flat-field stages of CameraCalibration.correct
// (camera/CameraCalibration.py:416-437): the stage-corrected pixel is computed ONCE per pixel
// into an LDS tile (block 64 x 4 outputs + 1-px symmetric halo), the median is a 19-exchange
// selection network on the 9 LDS values.
template <typename T> struct finite_max;
template <> struct finite_max<float> { static constexpr float value = 3.402823466e+38f; };
template <> struct finite_max<double> { static constexpr double value = 1.7976931348623157e+308; };

template <typename T> __device__ __forceinline__ void order2(T& a, T& b) {
  const bool lt = a < b;
  const T lo = lt ? a : b, hi = lt ? b : a;
  a = lo;
  b = hi;
}

template <typename T> __device__ __forceinline__ T median_of_9(T (&p)[9]) {
  // exchanges of the classic median-of-9 selection network; p[4] ends as the median
  constexpr int net[19][2] = {{1, 2}, {4, 5}, {7, 8}, {0, 1}, {3, 4}, {6, 7}, {1, 2},
                              {4, 5}, {7, 8}, {0, 3}, {5, 8}, {4, 7}, {3, 6}, {1, 4},
                              {2, 5}, {4, 7}, {4, 2}, {6, 4}, {4, 2}};
#pragma unroll
  for (int i = 0; i < 19; i++) order2(p[net[i][0]], p[net[i][1]]);
  return p[4];
}

template <typename T, bool CALIB>
__global__ void __launch_bounds__(256)
median_threshold_kernel(const T* __restrict__ img, const T* __restrict__ bg,
                        const T* __restrict__ ff, int h, int w, long pitch, long bgpitch,
                        long ffpitch, double threshold, int cond_less, T* __restrict__ out,
                        long opitch, unsigned char* __restrict__ indices, long ipitch) {
  constexpr int TW = 66, TH = 6;
  __shared__ T tile[TH][TW + 1];
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int e = tid; e < TW * TH; e += 256) {
    const int ty = e / TW, tx = e - ty * TW;
    // scipy 'reflect' == edge pixel repeated; the clamp only keeps far-outside halo
    // elements of edge tiles (never read by an in-range pixel) inside the image
    int yy = resolve_idx(y0 + ty - 1, h, IPA_BORDER_REFLECT);
    int xx = resolve_idx(x0 + tx - 1, w, IPA_BORDER_REFLECT);
    T v = img[(long)yy * pitch + xx];
    if constexpr (CALIB) {
      if (bg) v -= bg[(long)yy * bgpitch + xx];
      if (ff) {
        const T d = ff[(long)yy * ffpitch + xx];
        if (d != (T)0) v /= d;
      }
      if (threshold > 0) {  // np.nan_to_num belongs to the artefact stage
        if (v != v) v = (T)0;
        else if (v > finite_max<T>::value) v = finite_max<T>::value;
        else if (v < -finite_max<T>::value) v = -finite_max<T>::value;
      }
    }
    tile[ty][tx] = v;
  }
  __syncthreads();
  const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
  if (x >= w || y >= h) return;
  const T a = tile[threadIdx.y + 1][threadIdx.x + 1];
  T res = a;
  bool hit = false;
  if (threshold > 0) {
    T p[9];
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++) p[dy * 3 + dx] = tile[threadIdx.y + dy][threadIdx.x + dx];
    const T blur = median_of_9(p);
    const double rel = fabs(((double)a - (double)blur) / (double)blur);
    hit = cond_less ? rel < threshold : rel > threshold;
    if (hit) res = blur;
  }
  out[(long)y * opitch + x] = res;
  if (indices) indices[(long)y * ipitch + x] = hit ? 1 : 0;
}

// filters/medianThreshold.py:7-30 with ANY size (round 4): scipy.ndimage.median_filter(img,
// size) = the element of rank size*size / 2 of the size x size window at offsets -size/2 ..
// size - 1 - size/2, edge pixels repeated ('reflect').  A block of 64 x 4 outputs stages its
// window tile in LDS once; every lane then finds ITS window's rank element by counting: the
// candidate v is the answer when (values < v) <= rank < (values <= v) - what sorting would put at
// that rank, for any size, without a size-specific selection network (3x3 keeps its own).
template <typename T>
__global__ void __launch_bounds__(256)
median_threshold_any_kernel(const T* __restrict__ img, int h, int w, long pitch, int size,
                            double threshold, int cond_less, T* __restrict__ out, long opitch,
                            unsigned char* __restrict__ indices, long ipitch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mt_lds[];
  T* tile = reinterpret_cast<T*>(mt_lds);
  const int lo = size / 2, TW = 64 + size - 1, TH = 4 + size - 1, TP = TW | 1;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int e = tid; e < TW * TH; e += 256) {
    const int ty = e / TW, tx = e - ty * TW;
    const int yy = resolve_idx(y0 + ty - lo, h, IPA_BORDER_REFLECT);
    const int xx = resolve_idx(x0 + tx - lo, w, IPA_BORDER_REFLECT);
    tile[ty * TP + tx] = img[(long)yy * pitch + xx];
  }
  __syncthreads();
  const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
  if (x >= w || y >= h) return;
  const T* win = tile + threadIdx.y * TP + threadIdx.x;   // window origin of this pixel
  const T a = win[lo * TP + lo];
  const int rank = (size * size) / 2;
  T blur = a;
  bool found = false;
  for (int cy = 0; cy < size && !found; cy++)
    for (int cx = 0; cx < size && !found; cx++) {
      const T v = win[cy * TP + cx];
      int less = 0, leq = 0;
      for (int by = 0; by < size; by++)
        for (int bx = 0; bx < size; bx++) {
          const T u = win[by * TP + bx];
          less += u < v ? 1 : 0;
          leq += u <= v ? 1 : 0;
        }
      if (less <= rank && rank < leq) {
        blur = v;
        found = true;
      }
    }
  const double rel = fabs(((double)a - (double)blur) / (double)blur);
  const bool hit = cond_less ? rel < threshold : rel > threshold;
  out[(long)y * opitch + x] = hit ? blur : a;
  if (indices) indices[(long)y * ipitch + x] = hit ? 1 : 0;
}

}  // namespace ipa

using namespace ipa;

int ipa_local_std_wave_launch(ipa_ctx* ctx, const void* img, const void* blurred, int dtype, int h,
                              int w, long pitch, long bpitch, int hkx, int hky, void* out,
                              long opitch);  // stencils_ydep.hip

extern "C" {

static int median_threshold_launch(ipa_ctx* ctx, const void* d_img, int dtype, const void* d_bg,
                                   const void* d_ff, bool calib, int h, int w, long pitch,
                                   long bg_pitch, long ff_pitch, double threshold, int cond_less,
                                   void* d_out, long out_pitch, unsigned char* d_indices,
                                   long idx_pitch) {
  IPA_REQUIRE(ctx, d_img && d_out, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0, "empty image");
  IPA_REQUIRE(ctx, pitch >= w && out_pitch >= w && (!d_bg || bg_pitch >= w) &&
                       (!d_ff || ff_pitch >= w) && (!d_indices || idx_pitch >= w),
              "pitch smaller than width");
  IPA_REQUIRE(ctx, d_img != d_out, "the 3x3 median cannot run in place");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "median threshold supports float32/float64 (got dtype %d)", dtype);
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
#define IPA_MT(T, CALIB)                                                                        \
  hipLaunchKernelGGL((median_threshold_kernel<T, CALIB>), grid, block, 0, ctx->stream,          \
                     (const T*)d_img, (const T*)d_bg, (const T*)d_ff, h, w, pitch, bg_pitch,     \
                     ff_pitch, threshold, cond_less, (T*)d_out, out_pitch, d_indices, idx_pitch)
  if (dtype == IPA_F32) {
    if (calib) IPA_MT(float, true); else IPA_MT(float, false);
  } else {
    if (calib) IPA_MT(double, true); else IPA_MT(double, false);
  }
#undef IPA_MT
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_median_threshold_dev(ipa_ctx* ctx, const void* d_img, int dtype, int h, int w, long pitch,
                             double threshold, int cond_less, void* d_out, long out_pitch,
                             unsigned char* d_indices, long idx_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, threshold > 0, "threshold must be > 0 (the reference returns the input as is)");
  return median_threshold_launch(ctx, d_img, dtype, nullptr, nullptr, false, h, w, pitch, 0, 0,
                                 threshold, cond_less, d_out, out_pitch, d_indices, idx_pitch);
}

int ipa_median_threshold_size_dev(ipa_ctx* ctx, const void* d_img, int dtype, int h, int w,
                                  long pitch, int size, double threshold, int cond_less,
                                  void* d_out, long out_pitch, unsigned char* d_indices,
                                  long idx_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, threshold > 0, "threshold must be > 0 (the reference returns the input as is)");
  IPA_REQUIRE(ctx, size >= 1, "size must be >= 1");
  if (size == 3)
    return median_threshold_launch(ctx, d_img, dtype, nullptr, nullptr, false, h, w, pitch, 0, 0,
                                   threshold, cond_less, d_out, out_pitch, d_indices, idx_pitch);
  IPA_REQUIRE(ctx, d_img && d_out, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0, "empty image");
  IPA_REQUIRE(ctx, pitch >= w && out_pitch >= w && (!d_indices || idx_pitch >= w),
              "pitch smaller than width");
  IPA_REQUIRE(ctx, d_img != d_out, "the median cannot run in place");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "median threshold supports float32/float64 (got dtype %d)", dtype);
  const size_t es = dtype == IPA_F32 ? 4 : 8;
  const size_t lds = (size_t)(4 + size - 1) * ((64 + size - 1) | 1) * es;
  if (lds > 64 * 1024)
    IPA_UNSUPPORTED(ctx, "median threshold: a %dx%d window does not fit the LDS tile", size, size);
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  if (dtype == IPA_F32)
    hipLaunchKernelGGL((median_threshold_any_kernel<float>), grid, block, lds, ctx->stream,
                       (const float*)d_img, h, w, pitch, size, threshold, cond_less, (float*)d_out,
                       out_pitch, d_indices, idx_pitch);
  else
    hipLaunchKernelGGL((median_threshold_any_kernel<double>), grid, block, lds, ctx->stream,
                       (const double*)d_img, h, w, pitch, size, threshold, cond_less,
                       (double*)d_out, out_pitch, d_indices, idx_pitch);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_calib_prefilter_dev(ipa_ctx* ctx, const void* d_img, int dtype, const void* d_bg,
                            const void* d_ff, int h, int w, long pitch, long bg_pitch,
                            long ff_pitch, double threshold, void* d_out, long out_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  return median_threshold_launch(ctx, d_img, dtype, d_bg, d_ff, true, h, w, pitch, bg_pitch,
                                 ff_pitch, threshold, 0, d_out, out_pitch, nullptr, 0);
}

int ipa_closest_distance_dev(ipa_ctx* ctx, const unsigned char* d_arr, int h, int w, long pitch,
                             int ksize, void* d_out, int out_dtype, long out_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_arr && d_out, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && ksize >= 1 && ksize < 20000, "empty image or bad ksize");
  IPA_REQUIRE(ctx, pitch >= w && out_pitch >= w, "pitch smaller than width");
  if (out_dtype != IPA_U16 && out_dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "closest_distance writes uint16 or float64 (got dtype %d)", out_dtype);
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  if (ksize <= 254) {  // (row distances fit a byte; the distances go through the context workspace)
    int rc = ipa_ws_reserve(ctx, (size_t)h * w);
    if (rc) return rc;
    unsigned char* g = (unsigned char*)ctx->ws;
    hipLaunchKernelGGL(closest_rowdist_kernel, grid, block, 0, ctx->stream, d_arr, h, w, pitch, ksize,
                       g);
    if (out_dtype == IPA_U16)
      hipLaunchKernelGGL((closest_coldist_kernel<unsigned short>), grid, block, 0, ctx->stream, d_arr,
                         g, h, w, pitch, ksize, (unsigned short*)d_out, out_pitch);
    else
      hipLaunchKernelGGL((closest_coldist_kernel<double>), grid, block, 0, ctx->stream, d_arr, g, h, w,
                         pitch, ksize, (double*)d_out, out_pitch);
    IPA_HIP(ctx, hipGetLastError());
    return IPA_OK;
  }
  if (out_dtype == IPA_U16)
    hipLaunchKernelGGL((closest_distance_kernel<unsigned short>), grid, block, 0, ctx->stream,
                       d_arr, h, w, pitch, ksize, (unsigned short*)d_out, out_pitch);
  else
    hipLaunchKernelGGL((closest_distance_kernel<double>), grid, block, 0, ctx->stream, d_arr, h, w,
                       pitch, ksize, (double*)d_out, out_pitch);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_pos_intensity_unc_dev(ipa_ctx* ctx, const void* d_img, int dtype, int h, int w, long pitch,
                              const double* d_sx, const double* d_sy, long sigma_pitch, double sx,
                              double sy, int ksize, double* d_sint, long out_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_img && d_sint, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && ksize >= 1, "empty image or ksize < 1");
  IPA_REQUIRE(ctx, pitch >= w && out_pitch >= w, "pitch smaller than width");
  IPA_REQUIRE(ctx, (d_sx == nullptr) == (d_sy == nullptr), "give both sigma maps or neither");
  IPA_REQUIRE(ctx, !d_sx || sigma_pitch >= w, "sigma map pitch smaller than width");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "pos_intensity_unc supports float32/float64 (got dtype %d)", dtype);
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  {
    const size_t lds = (size_t)(2 * ksize + 1) * 256 * sizeof(double);
    if (lds <= 60 * 1024) {
      if (dtype == IPA_F32)
        hipLaunchKernelGGL((pos_intensity_unc_sep_kernel<float>), grid, block, lds, ctx->stream,
                           (const float*)d_img, h, w, pitch, d_sx, d_sy, sigma_pitch, sx, sy, ksize,
                           d_sint, out_pitch);
      else
        hipLaunchKernelGGL((pos_intensity_unc_sep_kernel<double>), grid, block, lds, ctx->stream,
                           (const double*)d_img, h, w, pitch, d_sx, d_sy, sigma_pitch, sx, sy, ksize,
                           d_sint, out_pitch);
      IPA_HIP(ctx, hipGetLastError());
      return IPA_OK;
    }
  }
  if (dtype == IPA_F32)
    hipLaunchKernelGGL((pos_intensity_unc_kernel<float>), grid, block, 0, ctx->stream,
                       (const float*)d_img, h, w, pitch, d_sx, d_sy, sigma_pitch, sx, sy, ksize,
                       d_sint, out_pitch);
  else
    hipLaunchKernelGGL((pos_intensity_unc_kernel<double>), grid, block, 0, ctx->stream,
                       (const double*)d_img, h, w, pitch, d_sx, d_sy, sigma_pitch, sx, sy, ksize,
                       d_sint, out_pitch);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

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
  if (fill_mask && d_out == d_arr) {
    // the in-place fill: wave-cooperative kernel (masked pixels are sparse)
    const int segs_x = (w + 63) / 64;
    const long segs = (long)segs_x * h;
    dim3 wgrid((unsigned)((segs + 3) / 4)), wblock(256);
    const bool cols = ksize / 2 >= 1 && ksize / 2 <= 32;
    if (dtype == IPA_F32) {
      if (cols)
        hipLaunchKernelGGL((masked_mean_fill_cols_kernel<float>), wgrid, wblock, 0, ctx->stream,
                           (float*)d_out, d_mask, h, w, pitch, mask_pitch, ksize / 2, segs_x);
      else
        hipLaunchKernelGGL((masked_mean_fill_wave_kernel<float>), wgrid, wblock, 0, ctx->stream,
                           (float*)d_out, d_mask, h, w, pitch, mask_pitch, ksize / 2, segs_x);
    } else {
      if (cols)
        hipLaunchKernelGGL((masked_mean_fill_cols_kernel<double>), wgrid, wblock, 0, ctx->stream,
                           (double*)d_out, d_mask, h, w, pitch, mask_pitch, ksize / 2, segs_x);
      else
        hipLaunchKernelGGL((masked_mean_fill_wave_kernel<double>), wgrid, wblock, 0, ctx->stream,
                           (double*)d_out, d_mask, h, w, pitch, mask_pitch, ksize / 2, segs_x);
    }
  } else if (dtype == IPA_F32) {
    if (fill_mask) IPA_MM(float, true); else IPA_MM(float, false);
  } else {
    if (fill_mask) IPA_MM(double, true); else IPA_MM(double, false);
  }
#undef IPA_MM
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_masked_median_dev(ipa_ctx* ctx, const void* d_arr, int dtype, const unsigned char* d_mask,
                          int h, int w, long pitch, long mask_pitch, int ksize, int fill_mask,
                          void* d_out, long out_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_arr && d_mask && d_out, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && ksize >= 2, "empty image or ksize < 2");
  IPA_REQUIRE(ctx, pitch >= w && mask_pitch >= w && out_pitch >= w, "pitch smaller than width");
  IPA_REQUIRE(ctx, fill_mask || d_arr != d_out, "fill_mask=0 cannot run in place");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "masked_median supports float32/float64 (got dtype %d)", dtype);
  const int k = ksize / 2, cap = 4 * k * k;
  const size_t lds = (size_t)4 * 2 * cap * (dtype == IPA_F32 ? 4 : 8);  // two buffers per wave
  if (lds > 64 * 1024)
    IPA_UNSUPPORTED(ctx, "masked_median: a %dx%d window does not fit the per-wave LDS buffers",
                    2 * k, 2 * k);
  const int segs_x = (w + 63) / 64;
  const long segs = (long)segs_x * h;
  dim3 grid((unsigned)((segs + 3) / 4)), block(256);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
#define IPA_MMED(T, FILL)                                                                       \
  hipLaunchKernelGGL((masked_median_wave_kernel<T, FILL>), grid, block, lds, ctx->stream,       \
                     (const T*)d_arr, d_mask, h, w, pitch, mask_pitch, k, cap, segs_x, (T*)d_out, \
                     out_pitch)
  if (dtype == IPA_F32) {
    if (fill_mask) IPA_MMED(float, true); else IPA_MMED(float, false);
  } else {
    if (fill_mask) IPA_MMED(double, true); else IPA_MMED(double, false);
  }
#undef IPA_MMED
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
  {
    const int k = ksize / 2, rb = 32;
    const size_t esz = dtype == IPA_F32 ? 4 : 8;
    const size_t lds = ((size_t)(rb + 2 * k) * (64 + 2 * k) + (size_t)(rb + 2 * k) * 64) * esz;
    if (k >= 1 && lds <= 60 * 1024) {
      dim3 sgrid((w + 63) / 64, (h + rb - 1) / rb), sblock(256);
      if (dtype == IPA_F32)
        hipLaunchKernelGGL((nan_max_sep_kernel<float>), sgrid, sblock, lds, ctx->stream,
                           (const float*)d_arr, h, w, pitch, k, rb, (float*)d_out, out_pitch);
      else
        hipLaunchKernelGGL((nan_max_sep_kernel<double>), sgrid, sblock, lds, ctx->stream,
                           (const double*)d_arr, h, w, pitch, k, rb, (double*)d_out, out_pitch);
      IPA_HIP(ctx, hipGetLastError());
      return IPA_OK;
    }
  }
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
  const size_t esz = dtype == IPA_F32 ? 4 : 8;
  const size_t lds = (size_t)(64 + k1 - 1) * (4 + k0 - 1) * esz;
  if (lds <= 48 * 1024) {  // the window of a block fits in LDS: staged version
    if (dtype == IPA_F32)
      hipLaunchKernelGGL((conv_ydep_tile_kernel<float>), grid, block, lds, ctx->stream,
                         (const float*)d_src, h, w, src_pitch, d_kernels, k0, k1, border_x,
                         border_y, (float*)d_dst, dst_pitch);
    else
      hipLaunchKernelGGL((conv_ydep_tile_kernel<double>), grid, block, lds, ctx->stream,
                         (const double*)d_src, h, w, src_pitch, d_kernels, k0, k1, border_x,
                         border_y, (double*)d_dst, dst_pitch);
  } else if (dtype == IPA_F32)
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
  const int hkx = ksize_x / 2, hky = ksize_y / 2;
  // square windows up to 11: 256-px tiles, 4 pixels per lane (stencils_ydep.hip)
  if (ipa_local_std_wave_launch(ctx, d_img, d_blurred, dtype, h, w, pitch, blurred_pitch, hkx, hky,
                                d_out, out_pitch) == 0) {
    IPA_HIP(ctx, hipGetLastError());
    return IPA_OK;
  }
  const size_t lds = (size_t)(64 + 2 * hky) * (4 + 2 * hkx) * (dtype == IPA_F32 ? 4 : 8);
  if (lds <= 48 * 1024) {  // the block's window fits in LDS: staged version
    if (dtype == IPA_F32)
      hipLaunchKernelGGL((local_std_tile_kernel<float>), grid, block, lds, ctx->stream,
                         (const float*)d_img, (const float*)d_blurred, h, w, pitch, blurred_pitch,
                         hkx, hky, (float*)d_out, out_pitch);
    else
      hipLaunchKernelGGL((local_std_tile_kernel<double>), grid, block, lds, ctx->stream,
                         (const double*)d_img, (const double*)d_blurred, h, w, pitch,
                         blurred_pitch, hkx, hky, (double*)d_out, out_pitch);
  } else if (dtype == IPA_F32)
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
