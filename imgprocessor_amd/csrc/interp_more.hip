// interp_more.hip — the other three fills of the reference's interpolate/ package on gfx950:
//   ipa_unstructured_idw*   interpolate/interpolate2dUnstructuredIDW.py:7-38
//   ipa_circular_idw_fill*  interpolate/interpolateCircular2dStructuredIDW.py:7-69
//   ipa_cross_avg_fill*     interpolate/interpolate2dStructuredCrossAvg.py:7-115
//   ipa_point_spread_idw*   interpolate/interpolate2dStructuredPointSpreadIDW.py:7-141 (round 4)
//
// All three read only unmasked pixels and write only masked ones (the scattered-point fill
// writes every pixel and reads none), so they run in place without a copy.  The arithmetic is
// float64 as in the reference (numba types the accumulators float64); the unit of work is the
// wave for the two masked fills (all 64 lanes over one masked pixel's window, as in idw.hip)
// and the lane for the scattered-point fill (every pixel loops over the same points).
#include <cmath>
#include <vector>

#include "common.hpp"

// the reference's float64 expressions operation for operation: no fused multiply-add (hipcc
// contracts by default; the __dmul_rn / __dadd_rn intrinsics do not help - they are inline
// functions of the HIP headers and carry THEIR translation mode); the pragma has to open the
// function body to reach a template's instantiations
#define IPA_NO_FMA _Pragma("clang fp contract(off)")

namespace ipa {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// 1 / d2^(power/2): the two common exponents without pow (pow(x, 1.0) == x exactly)
template <int PW> __device__ __forceinline__ double inv_dist_pow(double d2, double half_power) {
  IPA_NO_FMA
  if constexpr (PW == 2) return 1.0 / d2;
  else if constexpr (PW == 1) return 1.0 / sqrt(d2);
  else return 1.0 / pow(d2, half_power);
}

// ---------------------------------------------------------------- scattered points --
// pts = [x(n) | y(n) | v(n)] doubles; x is the ROW coordinate (the reference indexes
// grid[i, j] with i against x).  The loads are wave-uniform (scalar unit); the sums run in
// point order like the reference's, a pixel ON a point takes the first such point's value.
template <typename T, int PW>
__global__ void __launch_bounds__(256)
unstructured_idw_kernel(T* __restrict__ grid, int h, int w, long pitch,
                        const double* __restrict__ pts, int n, double half_power) {
  IPA_NO_FMA
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= h || j >= w) return;
  const double fi = (double)i, fj = (double)j;
  double sw = 0.0, sv = 0.0, hit = 0.0;
  bool over = false;
  for (int k = 0; k < n; k++) {
    const double x = pts[k], y = pts[n + k], v = pts[2 * n + k];
    if (x == fi && y == fj) {
      hit = v;
      over = true;
      break;
    }
    const double dx = x - fi, dy = y - fj;
    const double wi = inv_dist_pow<PW>(dx * dx + dy * dy, half_power);
    sw += wi;
    sv += wi * v;
  }
  grid[(long)i * pitch + j] = (T)(over ? hit : sv / sw);
}

// ---------------------------------------------------------------- polar-distance IDW --
// As written in the reference: rows AND columns run to shape[0] (gy = grid.shape[0]), the
// window is [i-k, min(i+k, gx)) x [j-k, min(j+k, gx)) with the upper end exclusive, and the
// distance is the SQUARE of (fr dr)^2 + (fphi dphi)^2.
// radius and angle of every pixel of the g x g domain about (cx, cy), once per call: a window
// position then costs two loads instead of a float64 sqrt and atan2 (every unmasked pixel sits in
// the windows of many masked ones; 2160^2 with 2 % scattered + a hole, kernel 15: 874 -> 686 us).  The same expressions as at the masked pixel itself: same bits.
__global__ void __launch_bounds__(256)
polar_table_kernel(int g, double cx, double cy, double2* __restrict__ tab) {
  IPA_NO_FMA
  const int yi = blockIdx.x * 256 + threadIdx.x, xi = blockIdx.y;
  if (yi >= g) return;
  const double ni = (double)xi - cx, nj = (double)yi - cy;
  tab[(long)xi * g + yi] = make_double2(sqrt(ni * ni + nj * nj), atan2(nj, ni));
}

template <typename T, int PW>
__global__ void __launch_bounds__(256)
circular_idw_kernel(T* __restrict__ grid, const uint8_t* __restrict__ mask, int g, int w,
                    long pitch, int ksize, double half_power, double fr, double fphi, double cx,
                    double cy, int segs_x, const double2* __restrict__ polar) {
  IPA_NO_FMA
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long seg = (long)blockIdx.x * 4 + wave;
  const int row = (int)(seg / segs_x);
  if (row >= g) return;
  const int xs = (int)(seg - (long)row * segs_x) * 64;
  const int x = xs + lane;
  unsigned long long todo = __ballot(x < g && mask[(long)row * w + x] != 0);
  const double kTwoPi = 6.283185307179586476925286766559;
  while (todo) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int j = xs + b;
    const int xmn = row - ksize < 0 ? 0 : row - ksize, xmx = row + ksize > g ? g : row + ksize;
    const int ymn = j - ksize < 0 ? 0 : j - ksize, ymx = j + ksize > g ? g : j + ksize;
    const int nx = xmx - xmn, ny = ymx - ymn;
    const double di = (double)row - cx, dj = (double)j - cy;
    const double R = sqrt(di * di + dj * dj), PHI = atan2(dj, di);
    double sw = 0.0, sv = 0.0;
    // (t / ny by a multiplication, exact while t ny < 2^20 - see cross_local_avg_kernel)
    const bool fastdiv = ny > 0 && (long)nx * ny * ny < (1l << 20);
    const unsigned M = ny > 0 ? ((1u << 20) + (unsigned)ny - 1u) / (unsigned)ny : 0u;
    for (int t = lane; t < nx * ny; t += 64) {
      const int a = fastdiv ? (int)(((unsigned long long)(unsigned)t * M) >> 20) : t / ny;
      const int xi = xmn + a, yi = ymn + (t - a * ny);
      // (mask, table entry and value loaded side by side: one round trip per pass)
      const uint8_t mk = mask[(long)xi * w + yi];
      const double2 rp = polar[(long)xi * g + yi];
      const T gv = grid[(long)xi * pitch + yi];
      if ((xi != row || yi != j) && mk == 0) {
        const double nR = rp.x;
        const double dr = R - nR, midR = 0.5 * (R + nR);
        const double d = fabs(PHI - rp.y), e = kTwoPi - d;
        const double dphi = (e < d ? e : d) * midR;
        const double p = fr * dr, q = fphi * dphi;
        const double s = p * p + q * q;
        const double wi = inv_dist_pow<PW>(s * s, half_power);
        sw += wi;
        sv += wi * (double)gv;
      }
    }
    sw = wsum(sw);
    sv = wsum(sv);
    if (lane == 0 && sw != 0.0) grid[(long)row * pitch + j] = (T)(sv / sw);
  }
}

// ---------------------------------------------------------------- cross average --
// pixels of a row one wave looks after: the wave works through ITS masked pixels one after the
// other, so a hole costs the launch the time of the wave with the most hole pixels (4K with a
// 200 x 400 hole + 2 % scattered, kernel 5: 64 per wave 906 us, 16 per wave 852)
#ifndef IPA_CROSS_SEG
#define IPA_CROSS_SEG 16
#endif
constexpr int kCrossSeg = IPA_CROSS_SEG;
// Pass 1: _localAvg (:21-44) at every unmasked pixel that can be the end of a search - one with
// a masked 4-neighbour - stored in the grid's dtype (the reference's `vals` array).
template <typename T>
__global__ void __launch_bounds__(256)
cross_local_avg_kernel(const T* __restrict__ grid, const uint8_t* __restrict__ mask, int h, int w,
                       long pitch, int ksize, T* __restrict__ avg, int segs_x) {
  IPA_NO_FMA
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long seg = (long)blockIdx.x * 4 + wave;
  const int row = (int)(seg / segs_x);
  if (row >= h) return;
  const int xs = (int)(seg - (long)row * segs_x) * kCrossSeg;
  const int x = xs + lane;
  bool flag = false;
  if (lane < kCrossSeg && x < w && mask[(long)row * w + x] == 0) {
    flag = (row > 0 && mask[(long)(row - 1) * w + x]) || (row < h - 1 && mask[(long)(row + 1) * w + x]) ||
           (x > 0 && mask[(long)row * w + x - 1]) || (x < w - 1 && mask[(long)row * w + x + 1]);
  }
  unsigned long long todo = __ballot(flag);
  while (todo) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int j = xs + b;
    // (the reference clamps to gx / gy and reads that index - out of bounds; clamped to the
    // array here, identical wherever the reference is defined)
    const int xmn = row - ksize < 0 ? 0 : row - ksize, xmx = row + ksize > h - 1 ? h - 1 : row + ksize;
    const int ymn = j - ksize < 0 ? 0 : j - ksize, ymx = j + ksize > w - 1 ? w - 1 : j + ksize;
    const int ny = ymx - ymn + 1, nt = (xmx - xmn + 1) * ny;
    double sv = 0.0, sn = 0.0;
    // t / ny by a multiplication while t * ny < 2^20 (exact there: M = ceil(2^20 / ny) errs by
    // less than one part in 2^20 / ny) - the integer division was a third of the loop
    const bool fastdiv = (long)nt * ny < (1l << 20);
    const unsigned M = ((1u << 20) + (unsigned)ny - 1u) / (unsigned)ny;
    // two positions per pass, mask and value loaded side by side (the value of a masked
    // position is dropped): one memory round trip per 128 positions instead of two per 64 - the
    // loop waits for its loads and nothing else.  Per-lane order of the sums unchanged.
    auto at = [&](int t, long& mi, long& gi) {
      const int a = fastdiv ? (int)(((unsigned long long)(unsigned)t * M) >> 20) : t / ny;
      const int xi = xmn + a, yi = ymn + (t - a * ny);
      mi = (long)xi * w + yi;
      gi = (long)xi * pitch + yi;
    };
    for (int t = lane; t < nt; t += 128) {
      long m0i, g0i, m1i = 0, g1i = 0;
      at(t, m0i, g0i);
      const bool two = t + 64 < nt;
      if (two) at(t + 64, m1i, g1i);
      const uint8_t m0 = mask[m0i];
      const T g0 = grid[g0i];
      const uint8_t m1 = two ? mask[m1i] : (uint8_t)1;
      const T g1 = two ? grid[g1i] : (T)0;
      if (m0 == 0) {
        sv += (double)g0;
        sn += 1.0;
      }
      if (m1 == 0) {
        sv += (double)g1;
        sn += 1.0;
      }
    }
    sv = wsum(sv);
    sn = wsum(sn);
    if (lane == 0) avg[(long)row * w + j] = (T)(sv / sn);
  }
}

// Pass 2: per row the last masked pixel with an unmasked pixel somewhere to its left (-1: none)
__global__ void __launch_bounds__(256)
cross_row_last_kernel(const uint8_t* __restrict__ mask, int h, int w, int* __restrict__ rowlast) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= h) return;
  const uint8_t* m = mask + (long)row * w;
  int first = w;  // first unmasked column
  for (int c = 0; c < w; c += 64) {
    const unsigned long long un = __ballot(c + lane < w && m[c + lane] == 0);
    if (un) {
      first = c + __ffsll((long long)un) - 1;
      break;
    }
  }
  int last = -1;
  for (int c = ((w - 1) / 64) * 64; c >= 0 && c + 63 > first; c -= 64) {
    const unsigned long long ms = __ballot(c + lane < w && c + lane > first && m[c + lane] != 0);
    if (ms) {
      last = c + 63 - __clzll((long long)ms);
      break;
    }
  }
  if (lane == 0) rowlast[row] = last;
}

// Pass 3 (one wave): prev[i] = the last row before i whose rowlast is >= 0 (-1: none)
__global__ void __launch_bounds__(64)
cross_prev_row_kernel(const int* __restrict__ rowlast, int h, int* __restrict__ prev) {
  const int lane = threadIdx.x;
  int carry = -1;
  for (int c = 0; c < h; c += 64) {
    const unsigned long long has = __ballot(c + lane < h && rowlast[c + lane] >= 0);
    const unsigned long long below = lane == 0 ? 0ull : has & (~0ull >> (64 - lane));
    if (c + lane < h) prev[c + lane] = below ? c + 63 - __clzll((long long)below) : carry;
    if (has) carry = c + 63 - __clzll((long long)has);
  }
}

// distance (1-based) from (row, col) along (dr, dc) to the nearest unmasked pixel within
// `count` steps, 0 when there is none: 64 positions per pass
__device__ __forceinline__ int cross_search(const uint8_t* __restrict__ mask, int w, int row,
                                            int col, int dr, int dc, int count, int lane) {
  for (int c = 0; c < count; c += 64) {
    const int t = c + lane + 1;
    const bool un = t <= count && mask[(long)(row + t * dr) * w + (col + t * dc)] == 0;
    const unsigned long long hit = __ballot(un);
    if (hit) return c + __ffsll((long long)hit);
  }
  return 0;
}

// Pass 4: the fill.  Slots as in the reference: 0 = the search towards row 0, 2 = towards
// column 0 (valid when THAT or the search towards the last row succeeded - the source raises
// valid[2] there and never uses the value it found; with only that search successful slot 2
// still holds what the last earlier pixel in raster order left there), 3 = towards the last
// column (only when i < gy - 1, as written).  dist is uint16, the weights float32.
template <typename T>
__global__ void __launch_bounds__(256)
cross_fill_kernel(T* __restrict__ grid, const uint8_t* __restrict__ mask, int h, int w,
                  long pitch, double half_power, const T* __restrict__ avg,
                  const int* __restrict__ rowlast, const int* __restrict__ prev, int segs_x) {
  IPA_NO_FMA
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long seg = (long)blockIdx.x * 4 + wave;
  const int row = (int)(seg / segs_x);
  if (row >= h) return;
  const int xs = (int)(seg - (long)row * segs_x) * kCrossSeg;
  const int x = xs + lane;
  unsigned long long todo = __ballot(lane < kCrossSeg && x < w && mask[(long)row * w + x] != 0);
  while (todo) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int j = xs + b;
    // the first 8 steps of all four searches in ONE load (lanes 8 d .. 8 d + 7 = direction d): an
    // isolated masked pixel ends every search there - one memory round trip instead of four
    // (4K, 2 % scattered + a 200 x 400 hole, kernel 5: 615 -> 566 us for the whole fill)
    const int c0 = row, c1 = h - 1 - row, c2n = j, c3 = row < w - 1 ? w - 1 - j : 0;
    const int dir = lane >> 3, t = (lane & 7) + 1;
    const int cnt = dir == 0 ? c0 : (dir == 1 ? c1 : (dir == 2 ? c2n : c3));
    const int pr = row + (dir == 0 ? -t : (dir == 1 ? t : 0)), pc = j + (dir == 2 ? -t : (dir == 3 ? t : 0));
    const unsigned long long hit8 = __ballot(lane < 32 && t <= cnt && mask[(long)pr * w + pc] == 0);
    auto near = [&](int d, int count, int dr, int dc) {
      const unsigned hd = (unsigned)(hit8 >> (8 * d)) & 0xffu;
      if (hd) return (int)__ffs((int)hd);
      return count > 8 ? cross_search(mask, w, row, j, dr, dc, count, lane) : 0;
    };
    const int d0 = near(0, c0, -1, 0);
    const int d1 = near(1, c1, 1, 0);
    int d2 = near(2, c2n, 0, -1);
    const int d3 = near(3, c3, 0, 1);
    int r2 = row, c2 = j - d2;
    bool v2 = d2 > 0;
    if (!v2 && d1 > 0) {  // the stale slot
      r2 = prev[row];
      if (r2 >= 0) {
        const int jl = rowlast[r2];
        d2 = cross_search(mask, w, r2, jl, 0, -1, jl, lane);
        c2 = jl - d2;
        v2 = true;
      }
    }
    if (lane == 0) {
      double val[3];
      float wt[3];
      bool ok[3] = {d0 > 0, v2, d3 > 0};
      const int dist[3] = {d0, d2, d3};
      val[0] = ok[0] ? (double)avg[(long)(row - d0) * w + j] : 0.0;
      val[1] = ok[1] ? (double)avg[(long)r2 * w + c2] : 0.0;
      val[2] = ok[2] ? (double)avg[(long)row * w + j + d3] : 0.0;
      float wsumf = 0.f;
#pragma unroll
      for (int s = 0; s < 3; s++) {
        const double dd = (double)(unsigned short)dist[s];
        // (power 2: pow(d, 1.0) is d itself, exactly - and the library call is most of what
        // lane 0 does for a pixel)
        wt[s] = ok[s] ? (float)(1.0 / (half_power == 1.0 ? dd : pow(dd, half_power))) : 0.f;
        if (ok[s]) wsumf += wt[s];
      }
      if constexpr (sizeof(T) == 4) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 3; s++)
          if (ok[s]) acc += (float)val[s] * (wt[s] / wsumf);
        grid[(long)row * pitch + j] = acc;
      } else {
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < 3; s++)
          if (ok[s]) acc += val[s] * (double)(wt[s] / wsumf);
        grid[(long)row * pitch + j] = acc;
      }
    }
  }
}


// ------------------------------------------------------------- point spread IDW --
// interpolate/interpolate2dStructuredPointSpreadIDW.py.  _createBorder (:31-63) is a pair of
// scans whose only state is "the previous pixel" - in raster order for the row scan, in column-major
// order for the column scan, carried ACROSS the ends of rows / columns - and whose only effect
// is to SET flags: every pixel can apply both rules by itself.
__global__ void __launch_bounds__(256)
ps_border_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ border, int gx, int gy,
                 unsigned* __restrict__ any) {
  const long n = (long)gx * gy;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  bool found = false;
  if (p < n) {
    const int i = (int)(p / gy), j = (int)(p - (long)i * gy);
    const uint8_t val = mask[p];
    if (p > 0) {   // row-major predecessor: (i, j - 1), or the last pixel of the row above
      const uint8_t last = mask[p - 1];
      if (val != last) {
        found = true;
        if (val) border[p] = 1;
        else border[(long)i * gy + (j > 0 ? j - 1 : gy - 1)] = 1;   // (index -1: the row's last pixel)
      }
    }
    if (i > 0 || j > 0) {   // column-major predecessor: (i - 1, j), or the last pixel of column j - 1
      const uint8_t last = i > 0 ? mask[p - gy] : mask[(long)(gx - 1) * gy + (j - 1)];
      if (val != last) {
        found = true;
        if (val) border[p] = 1;
        else border[(long)(i > 0 ? i - 1 : gx - 1) * gy + j] = 1;
      }
    }
  }
  if (__builtin_amdgcn_ballot_w64(found) != 0 && (threadIdx.x & 63) == 0) atomicOr(any, 1u);
}

// One sweep of _calc (:75-135): the border pixels in raster order, each filled from the unmasked
// pixels of its window - the mask AS THE SWEEP HAS LEFT IT SO FAR - and then unmasked itself, so
// a pixel depends on every border pixel before it inside its window and must not be overtaken by
// a later one that would unmask a pixel it still has to see masked.  One workgroup of 16 waves:
// wave w takes the rows w, w + 16, ... in order, its 64 lanes share a pixel's window; a pixel
// (i, j) starts when each of the k rows above has finished its pixels left of the window's end + 1
// (their progress, in LDS): what lies inside its window in those rows is final, and no pixel of a
// row BELOW can have been filled inside it (that one waits for this row to pass ITS column + k).
// The sums run over the lanes of a wave, not in the source's raster order: float64, equal to the
// last bits only.
template <typename T, int PW>
__global__ void __launch_bounds__(1024)
ps_sweep_kernel(T* grid, uint8_t* mask, uint8_t* border, int gx, int gy, long pitch, int k,
                double half_power) {
  IPA_NO_FMA
  extern __shared__ int ps_prog[];
  volatile int* prog = ps_prog;
  for (int r = threadIdx.x; r < gx; r += 1024) ps_prog[r] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const volatile uint8_t* vmask = mask;
  const volatile T* vgrid = grid;
  for (int i = wave; i < gx; i += 16) {
    for (int j0 = 0; j0 < gy; j0 += 64) {
      const int jl = j0 + lane;
      unsigned long long bits = __builtin_amdgcn_ballot_w64(jl < gy && border[(long)i * gy + jl] != 0);
      while (bits) {
        const int q = __builtin_ctzll(bits);
        bits &= bits - 1;
        const int j = j0 + q;
        const int xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
        const int ymn = j - k < 0 ? 0 : j - k;
        int ymx = j + k;
        if (ymx > gx) ymx = gy;   // (as written: the column limit against the ROW count)
        if (ymx > gy) ymx = gy;   // (out of bounds in the source)
        // the rows above inside the window: finished left of the window's end + 1
        const int need = ymx + 1 < gy ? ymx + 1 : gy;
        for (int r0 = i - 1; r0 >= 0 && r0 >= i - k; r0 -= 64) {
          const int r = r0 - lane;
          for (;;) {
            const bool ok = r < 0 || r < i - k || prog[r] >= need;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
            __builtin_amdgcn_s_sleep(2);
          }
        }
        double sw = 0.0, sv = 0.0;
        for (int xi = xmn; xi < xmx; xi++) {
          const double dx2 = (double)((xi - i) * (xi - i));
          for (int yi = ymn + lane; yi < ymx; yi += 64) {
            if ((xi != i || yi != j) && vmask[(long)xi * gy + yi] == 0) {
              const double wi = inv_dist_pow<PW>(dx2 + (double)((yi - j) * (yi - j)), half_power);
              sw += wi;
              sv += wi * (double)vgrid[(long)xi * pitch + yi];
            }
          }
        }
        sw = wsum(sw);
        sv = wsum(sv);
        if (sw != 0.0) {
          if (lane == 0) {
            grid[(long)i * pitch + j] = (T)(sv / sw);
            border[(long)i * gy + j] = 0;
            mask[(long)i * gy + j] = 0;
          }
          __threadfence();   // the fill is visible before the progress says so
        }
        if (lane == 0) prog[i] = j + 1;
      }
      if (lane == 0) prog[i] = j0 + 64 < gy ? j0 + 64 : gy;
    }
    if (lane == 0) prog[i] = gy;
  }
}

}  // namespace ipa

using namespace ipa;

extern "C" {

int ipa_unstructured_idw_dev(ipa_ctx* ctx, void* d_grid, int dtype, int h, int w, long pitch,
                             const double* x, const double* y, const double* v, int n,
                             double power) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_grid && x && y && v, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && pitch >= w && n >= 1, "bad shape / no points");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "unstructured_idw supports float32/float64 grids (got dtype %d)", dtype);
  std::vector<double> pts((size_t)3 * n);
  memcpy(pts.data(), x, (size_t)n * 8);
  memcpy(pts.data() + n, y, (size_t)n * 8);
  memcpy(pts.data() + 2 * (size_t)n, v, (size_t)n * 8);
  void* d = nullptr;
  int rc = ipa_tab_upload(ctx, pts.data(), pts.size() * 8, &d);
  if (rc) return rc;
  dim3 grid((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4)), block(256);
  const int pw = power == 2.0 ? 2 : (power == 1.0 ? 1 : 0);
#define IPA_UIDW(T, PW)                                                                        \
  hipLaunchKernelGGL((unstructured_idw_kernel<T, PW>), grid, block, 0, ctx->stream, (T*)d_grid, \
                     h, w, pitch, (const double*)d, n, 0.5 * power)
  if (dtype == IPA_F32) {
    if (pw == 2) IPA_UIDW(float, 2); else if (pw == 1) IPA_UIDW(float, 1); else IPA_UIDW(float, 0);
  } else {
    if (pw == 2) IPA_UIDW(double, 2); else if (pw == 1) IPA_UIDW(double, 1); else IPA_UIDW(double, 0);
  }
#undef IPA_UIDW
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_circular_idw_fill_dev(ipa_ctx* ctx, void* d_grid, int dtype, const uint8_t* d_mask, int h,
                              int w, long pitch, int ksize, double power, double fr, double fphi,
                              double cx, double cy) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_grid && d_mask, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && pitch >= w && ksize >= 1, "bad shape/ksize");
  // (:16-17 take both extents from shape[0]: fewer columns than rows index out of bounds there)
  IPA_REQUIRE(ctx, w >= h, "circular IDW: the reference runs rows and columns to shape[0]; "
                           "%d columns < %d rows is out of bounds there", w, h);
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "circular_idw_fill supports float32/float64 grids (got dtype %d)", dtype);
  const int segs_x = (h + 63) / 64;
  const long segs = (long)segs_x * h;
  dim3 grid((unsigned)((segs + 3) / 4)), block(256);
  const int pw = power == 2.0 ? 2 : (power == 1.0 ? 1 : 0);
  int rc = ipa_plan_reserve(ctx, (size_t)h * h * sizeof(double2));
  if (rc) return rc;
  double2* polar = (double2*)ctx->plan;
  hipLaunchKernelGGL(polar_table_kernel, dim3((unsigned)((h + 255) / 256), (unsigned)h), block, 0,
                     ctx->stream, h, cx, cy, polar);
#define IPA_CIDW(T, PW)                                                                          \
  hipLaunchKernelGGL((circular_idw_kernel<T, PW>), grid, block, 0, ctx->stream, (T*)d_grid, d_mask, \
                     h, w, pitch, ksize, 0.5 * power, fr, fphi, cx, cy, segs_x, polar)
  if (dtype == IPA_F32) {
    if (pw == 2) IPA_CIDW(float, 2); else if (pw == 1) IPA_CIDW(float, 1); else IPA_CIDW(float, 0);
  } else {
    if (pw == 2) IPA_CIDW(double, 2); else if (pw == 1) IPA_CIDW(double, 1); else IPA_CIDW(double, 0);
  }
#undef IPA_CIDW
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_cross_avg_fill_dev(ipa_ctx* ctx, void* d_grid, int dtype, const uint8_t* d_mask, int h,
                           int w, long pitch, int ksize, double power) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_grid && d_mask, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && pitch >= w && ksize >= 0, "bad shape/ksize");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "cross_avg_fill supports float32/float64 grids (got dtype %d)", dtype);
  const size_t es = ipa_dtype_size(dtype);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t avg_b = up((size_t)h * w * es), row_b = up((size_t)h * 4);
  int rc = ipa_plan_reserve(ctx, avg_b + 2 * row_b);
  if (rc) return rc;
  char* avg = (char*)ctx->plan;
  int* rowlast = (int*)(avg + avg_b);
  int* prev = (int*)(avg + avg_b + row_b);
  const int segs_x = (w + kCrossSeg - 1) / kCrossSeg;
  const long segs = (long)segs_x * h;
  dim3 grid((unsigned)((segs + 3) / 4)), block(256);
  hipLaunchKernelGGL(cross_row_last_kernel, dim3((unsigned)((h + 3) / 4)), block, 0, ctx->stream,
                     d_mask, h, w, rowlast);
  hipLaunchKernelGGL(cross_prev_row_kernel, dim3(1), dim3(64), 0, ctx->stream, rowlast, h, prev);
  if (dtype == IPA_F32) {
    hipLaunchKernelGGL((cross_local_avg_kernel<float>), grid, block, 0, ctx->stream,
                       (const float*)d_grid, d_mask, h, w, pitch, ksize, (float*)avg, segs_x);
    hipLaunchKernelGGL((cross_fill_kernel<float>), grid, block, 0, ctx->stream, (float*)d_grid,
                       d_mask, h, w, pitch, 0.5 * power, (const float*)avg, rowlast, prev, segs_x);
  } else {
    hipLaunchKernelGGL((cross_local_avg_kernel<double>), grid, block, 0, ctx->stream,
                       (const double*)d_grid, d_mask, h, w, pitch, ksize, (double*)avg, segs_x);
    hipLaunchKernelGGL((cross_fill_kernel<double>), grid, block, 0, ctx->stream, (double*)d_grid,
                       d_mask, h, w, pitch, 0.5 * power, (const double*)avg, rowlast, prev, segs_x);
  }
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

// host-pointer forms: grid (and mask) staged through the context's workspace
static int fill_host(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                     char** d_grid, uint8_t** d_mask, size_t* gb) {
  IPA_REQUIRE(ctx, grid && h > 0 && w > 0, "bad arguments");
  const size_t es = ipa_dtype_size(dtype);
  IPA_REQUIRE(ctx, es, "unknown dtype");
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  *gb = (size_t)h * w * es;
  int rc = ipa_ws_reserve(ctx, up(*gb) + up((size_t)h * w));
  if (rc) return rc;
  *d_grid = (char*)ctx->ws;
  *d_mask = (uint8_t*)(*d_grid + up(*gb));
  IPA_HIP(ctx, hipMemcpyAsync(*d_grid, grid, *gb, hipMemcpyHostToDevice, ctx->stream));
  if (mask)
    IPA_HIP(ctx, hipMemcpyAsync(*d_mask, mask, (size_t)h * w, hipMemcpyHostToDevice, ctx->stream));
  return IPA_OK;
}

static int fill_back(ipa_ctx* ctx, void* grid, const char* dg, size_t gb) {
  IPA_HIP(ctx, hipMemcpyAsync(grid, dg, gb, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

int ipa_unstructured_idw(ipa_ctx* ctx, void* grid, int dtype, int h, int w, const double* x,
                         const double* y, const double* v, int n, double power) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, grid && h > 0 && w > 0, "bad arguments");
  const size_t es = ipa_dtype_size(dtype);
  IPA_REQUIRE(ctx, es, "unknown dtype");
  const size_t gb = (size_t)h * w * es;
  int rc = ipa_ws_reserve(ctx, gb);  // every pixel is written: nothing to upload
  if (rc) return rc;
  rc = ipa_unstructured_idw_dev(ctx, ctx->ws, dtype, h, w, w, x, y, v, n, power);
  if (rc) return rc;
  return fill_back(ctx, grid, (const char*)ctx->ws, gb);
}

int ipa_circular_idw_fill(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                          int ksize, double power, double fr, double fphi, double cx, double cy) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, mask, "null mask");
  char* dg; uint8_t* dm; size_t gb;
  int rc = fill_host(ctx, grid, dtype, mask, h, w, &dg, &dm, &gb);
  if (rc) return rc;
  rc = ipa_circular_idw_fill_dev(ctx, dg, dtype, dm, h, w, w, ksize, power, fr, fphi, cx, cy);
  if (rc) return rc;
  return fill_back(ctx, grid, dg, gb);
}

int ipa_cross_avg_fill(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                       int ksize, double power) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, mask, "null mask");
  char* dg; uint8_t* dm; size_t gb;
  int rc = fill_host(ctx, grid, dtype, mask, h, w, &dg, &dm, &gb);
  if (rc) return rc;
  rc = ipa_cross_avg_fill_dev(ctx, dg, dtype, dm, h, w, w, ksize, power);
  if (rc) return rc;
  return fill_back(ctx, grid, dg, gb);
}

int ipa_point_spread_idw_dev(ipa_ctx* ctx, void* d_grid, int dtype, uint8_t* d_mask, int h, int w,
                             long pitch, int ksize, double power, long max_iter) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_grid && d_mask, "null pointer");
  IPA_REQUIRE(ctx, h > 0 && w > 0 && pitch >= w && ksize >= 0, "bad shape/ksize");
  IPA_REQUIRE(ctx, h <= 16000, "point spread IDW keeps one progress word per row in LDS: at most 16000 rows");
  if (dtype != IPA_F32 && dtype != IPA_F64)
    IPA_UNSUPPORTED(ctx, "point_spread_idw supports float32/float64 grids (got dtype %d)", dtype);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t bb = up((size_t)h * w);
  int rc = ipa_plan_reserve(ctx, bb + 256);
  if (rc) return rc;
  uint8_t* border = (uint8_t*)ctx->plan;
  unsigned* any = (unsigned*)((char*)ctx->plan + bb);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  IPA_HIP(ctx, hipMemsetAsync(border, 0, bb + 256, ctx->stream));
  const unsigned nb = (unsigned)(((long)h * w + 255) / 256);
  const double hp = 0.5 * power;
  const size_t lds = (size_t)h * sizeof(int);
  for (long n = 0;; n++) {
    // _createBorder; its return value decides whether another sweep runs
    hipLaunchKernelGGL(ps_border_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_mask, border, h, w, any);
    unsigned found = 0;
    IPA_HIP(ctx, hipMemcpyAsync(&found, any, sizeof(found), hipMemcpyDeviceToHost, ctx->stream));
    IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!found || n >= max_iter) break;
    IPA_HIP(ctx, hipMemsetAsync(any, 0, sizeof(unsigned), ctx->stream));
#define IPA_PS_LAUNCH(T, PW)                                                                     \
  hipLaunchKernelGGL((ps_sweep_kernel<T, PW>), dim3(1), dim3(1024), lds, ctx->stream, (T*)d_grid, \
                     d_mask, border, h, w, pitch, ksize, hp)
    if (dtype == IPA_F32) {
      if (power == 2.0) IPA_PS_LAUNCH(float, 2);
      else if (power == 1.0) IPA_PS_LAUNCH(float, 1);
      else IPA_PS_LAUNCH(float, 0);
    } else {
      if (power == 2.0) IPA_PS_LAUNCH(double, 2);
      else if (power == 1.0) IPA_PS_LAUNCH(double, 1);
      else IPA_PS_LAUNCH(double, 0);
    }
#undef IPA_PS_LAUNCH
  }
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_point_spread_idw(ipa_ctx* ctx, void* grid, int dtype, uint8_t* mask, int h, int w,
                         int ksize, double power, long max_iter) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, mask, "null mask");
  char* dg; uint8_t* dm; size_t gb;
  int rc = fill_host(ctx, grid, dtype, mask, h, w, &dg, &dm, &gb);
  if (rc) return rc;
  rc = ipa_point_spread_idw_dev(ctx, dg, dtype, dm, h, w, w, ksize, power, max_iter);
  if (rc) return rc;
  // the mask is modified too (filled pixels are unmasked)
  IPA_HIP(ctx, hipMemcpyAsync(mask, dm, (size_t)h * w, hipMemcpyDeviceToHost, ctx->stream));
  return fill_back(ctx, grid, dg, gb);
}

}  // extern "C"
