// idw.hip — the interpolate/ stencils of the hot path on gfx950:
//   ipa_idw_fill*       interpolate/interpolate2dStructuredIDW.py:26-65
//   ipa_fast_idw_fill*  interpolate/interpolate2dStructuredFastIDW.py:29-63
//
// Masked pixels are usually sparse, and each costs up to (2k+1)^2 = 961
// neighbour visits, so the unit of work is the WAVE, not the lane: a wave64
// owns 64 consecutive pixels of a row, ballots their mask bits and then
// processes one masked pixel at a time with all 64 lanes spread over the
// window (coalesced reads of grid / mask / weight-table rows), finishing with
// a wave-level shuffle reduction of the two float64 sums.  Reads touch only
// unmasked pixels and writes only masked ones, so running in place is race
// free — the same property that makes the reference loop order-independent.
#include <vector>

#include "common.hpp"

namespace ipa {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ROWS: lanes over the columns of a window row with batched loads (windows of 17..64 columns);
// otherwise lanes strided over the taps (small windows: fewer idle lanes; wide ones: any size)
template <typename T, bool ROWS>
__global__ void __launch_bounds__(256)
idw_kernel(T* __restrict__ grid, const uint8_t* __restrict__ mask, int h, int w, long pitch,
           int ksize, const double* __restrict__ weights, int segs_x) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long seg = (long)blockIdx.x * 4 + wave;
  const int row = (int)(seg / segs_x);
  if (row >= h) return;
  const int xs = (int)(seg - (long)row * segs_x) * 64;
  const int x = xs + lane;
  unsigned long long todo = __ballot(x < w && mask[(long)row * w + x] != 0);
  const int kw = 2 * ksize + 1, ntap = kw * kw;
  while (todo) {
    int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int j = xs + b;  // masked pixel (row, j)
    double sw = 0.0, sv = 0.0;
    if constexpr (ROWS) {
      // lanes over the columns of a window row (two rows per pass when the window is at most
      // 32 wide); mask, value and weight of 8 passes are loaded back to back, the value whether
      // or not it is used: no division per tap, no dependent mask -> weight -> value chain per
      // pass (4K, 5 % masked, kernel 15: 947 -> see profiles/r02_micro.txt)
      const bool two = kw <= 32;
      const int half = two ? lane >> 5 : 0, step = two ? 2 : 1;
      const int dx = two ? (lane & 31) : lane;
      const bool col_ok = dx < kw;
      const int xx = j + dx - ksize;
      const bool col_in = col_ok && xx >= 0 && xx < w;
      const int xxc = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
      const int dxc = col_ok ? dx : 0;
      for (int dy0 = half; dy0 - half < kw; dy0 += 8 * step) {
        uint8_t m[8];
        T g[8];
        double wt[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int dy = dy0 + u * step < kw ? dy0 + u * step : kw - 1;
          const int yy = row + dy - ksize;
          const int yyc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
          m[u] = mask[(long)yyc * w + xxc];
          g[u] = grid[(long)yyc * pitch + xxc];
          wt[u] = weights[dy * kw + dxc];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int dy = dy0 + u * step;
          const int yy = row + dy - ksize;
          if (col_in && dy < kw && yy >= 0 && yy < h && !(dy == ksize && dx == ksize) && m[u] == 0) {
            sw += wt[u];
            sv += wt[u] * (double)g[u];
          }
        }
      }
    } else {
      for (int t = lane; t < ntap; t += 64) {
        int dy = t / kw, dx = t - dy * kw;
        int yy = row + dy - ksize, xx = j + dx - ksize;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w && !(dy == ksize && dx == ksize) &&
            mask[(long)yy * w + xx] == 0) {
          double wi = weights[t];
          sw += wi;
          sv += wi * (double)grid[(long)yy * pitch + xx];
        }
      }
    }
    sw = wave_sum(sw);
    sv = wave_sum(sv);
    if (lane == 0 && sw != 0.0) grid[(long)row * pitch + j] = (T)(sv / sw);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
fast_idw_kernel(T* __restrict__ grid, const uint8_t* __restrict__ mask, int h, int w, long pitch,
                const int* __restrict__ offs, const double* __restrict__ weights, int n,
                int minnvals, int segs_x) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long seg = (long)blockIdx.x * 4 + wave;
  const int row = (int)(seg / segs_x);
  if (row >= h) return;
  const int xs = (int)(seg - (long)row * segs_x) * 64;
  const int x = xs + lane;
  unsigned long long todo = __ballot(x < w && mask[(long)row * w + x] != 0);
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  while (todo) {
    int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int j = xs + b;
    double sw = 0.0, sv = 0.0;
    int c = 0;  // hits accepted so far (the reference's running counter)
    for (int base = 0; base < n; base += 64) {
      int t = base + lane;
      bool live = t < n, hit = false, far = false;
      int yy = 0, xx = 0;
      if (live) {
        yy = row + offs[2 * t];
        xx = j + offs[2 * t + 1];
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) hit = mask[(long)yy * w + xx] == 0;
        else far = (yy < -1 || yy > h + 1) && (xx < -1 || xx > w + 1);
      }
      unsigned long long hitm = __ballot(hit);
      int before = c + __popcll(hitm & below);
      // positions where the sequential loop would stop: the (minnvals+1)-th hit
      // (included), or a far-outside neighbour once at least one hit exists (excluded)
      unsigned long long stop_hit = __ballot(hit && before == minnvals);
      unsigned long long stop_far = __ballot(far && before > 0);
      int ph = stop_hit ? __ffsll((long long)stop_hit) - 1 : 64;
      int pf = stop_far ? __ffsll((long long)stop_far) - 1 : 64;
      int stop = ph < pf ? ph : pf;
      bool take = hit && (lane < stop || (lane == stop && ph <= pf));
      if (take) {
        double wi = weights[t];
        sw += wi;
        sv += wi * (double)grid[(long)yy * pitch + xx];
      }
      if (stop < 64) break;
      c += __popcll(hitm);
    }
    sw = wave_sum(sw);
    sv = wave_sum(sv);
    if (lane == 0 && sw != 0.0) grid[(long)row * pitch + j] = (T)(sv / sw);
  }
}

}  // namespace ipa

using namespace ipa;

extern "C" {

int ipa_idw_fill_dev(ipa_ctx* ctx, void* d_grid, int dtype, const uint8_t* d_mask, int h, int w,
                     long pitch, int ksize, const double* weights) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_grid && d_mask && weights, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && pitch >= w && ksize >= 1 && ksize <= 512, "bad shape/ksize");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "idw_fill supports float32/float64 grids (got dtype %d)", dtype);
  size_t nt = (size_t)(2 * ksize + 1) * (2 * ksize + 1);
  void* dw = nullptr;
  int rc = ipa_tab_upload(ctx, weights, nt * sizeof(double), &dw);
  if (rc) return rc;
  int segs_x = (w + 63) / 64;
  long segs = (long)segs_x * h;
  dim3 grid((unsigned)((segs + 3) / 4)), block(256);
  const int kw = 2 * ksize + 1;
  const bool rows = kw > 16 && kw <= 64;
#define IPA_IDW(T, R)                                                                          \
  hipLaunchKernelGGL((idw_kernel<T, R>), grid, block, 0, ctx->stream, (T*)d_grid, d_mask, h, w, \
                     pitch, ksize, (const double*)dw, segs_x)
  if (dtype == IPA_F32) {
    if (rows) IPA_IDW(float, true); else IPA_IDW(float, false);
  } else {
    if (rows) IPA_IDW(double, true); else IPA_IDW(double, false);
  }
#undef IPA_IDW
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_fast_idw_fill_dev(ipa_ctx* ctx, void* d_grid, int dtype, const uint8_t* d_mask, int h,
                          int w, long pitch, const int32_t* offsets, const double* weights, int n,
                          int minnvals) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_grid && d_mask && offsets && weights, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && pitch >= w && n >= 1 && minnvals >= 0, "bad arguments");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "fast_idw_fill supports float32/float64 grids (got dtype %d)", dtype);
  // one table: [weights (n doubles)][offsets (2n int32)]
  std::vector<char> tab((size_t)n * 16);
  memcpy(tab.data(), weights, (size_t)n * 8);
  memcpy(tab.data() + (size_t)n * 8, offsets, (size_t)n * 8);
  void* d = nullptr;
  int rc = ipa_tab_upload(ctx, tab.data(), tab.size(), &d);
  if (rc) return rc;
  const double* dwt = (const double*)d;
  const int* doff = (const int*)((char*)d + (size_t)n * 8);
  int segs_x = (w + 63) / 64;
  long segs = (long)segs_x * h;
  dim3 grid((unsigned)((segs + 3) / 4)), block(256);
  if (dtype == IPA_F32)
    hipLaunchKernelGGL((fast_idw_kernel<float>), grid, block, 0, ctx->stream, (float*)d_grid,
                       d_mask, h, w, pitch, doff, dwt, n, minnvals, segs_x);
  else
    hipLaunchKernelGGL((fast_idw_kernel<double>), grid, block, 0, ctx->stream, (double*)d_grid,
                       d_mask, h, w, pitch, doff, dwt, n, minnvals, segs_x);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

static int idw_host(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                    char** d_grid, uint8_t** d_mask, size_t* gb) {
  IPA_REQUIRE(ctx, grid && mask && h > 0 && w > 0, "bad arguments");
  size_t es = ipa_dtype_size(dtype);
  IPA_REQUIRE(ctx, es, "unknown dtype");
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  *gb = (size_t)h * w * es;
  int rc = ipa_ws_reserve(ctx, up(*gb) + up((size_t)h * w));
  if (rc) return rc;
  *d_grid = (char*)ctx->ws;
  *d_mask = (uint8_t*)(*d_grid + up(*gb));
  IPA_HIP(ctx, hipMemcpyAsync(*d_grid, grid, *gb, hipMemcpyHostToDevice, ctx->stream));
  IPA_HIP(ctx, hipMemcpyAsync(*d_mask, mask, (size_t)h * w, hipMemcpyHostToDevice, ctx->stream));
  return IPA_OK;
}

int ipa_idw_fill(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                 int ksize, const double* weights) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  char* dg; uint8_t* dm; size_t gb;
  int rc = idw_host(ctx, grid, dtype, mask, h, w, &dg, &dm, &gb);
  if (rc) return rc;
  rc = ipa_idw_fill_dev(ctx, dg, dtype, dm, h, w, w, ksize, weights);
  if (rc) return rc;
  IPA_HIP(ctx, hipMemcpyAsync(grid, dg, gb, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

int ipa_fast_idw_fill(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                      const int32_t* offsets, const double* weights, int n, int minnvals) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  char* dg; uint8_t* dm; size_t gb;
  int rc = idw_host(ctx, grid, dtype, mask, h, w, &dg, &dm, &gb);
  if (rc) return rc;
  rc = ipa_fast_idw_fill_dev(ctx, dg, dtype, dm, h, w, w, offsets, weights, n, minnvals);
  if (rc) return rc;
  IPA_HIP(ctx, hipMemcpyAsync(grid, dg, gb, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

}  // extern "C"
