/*
 * oracle.c — CPU restatement of imgProcessor's per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under imgprocessor_amd/ may import, link
 * or execute this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / CPU comparator.
 *
 * What it restates (reference file:line, all relative to /root/reference):
 *   - imgProcessor/filters/_extendArrayForConvolution.py:5-97  -> orc_extend_array
 *   - imgProcessor/filters/maskedConvolve.py:24-43             -> orc_masked_convolve
 *   - scipy.ndimage.correlate / gaussian_filter as the reference calls them
 *     (filters/standardDeviation.py:23, filters/fastFilter.py:42,
 *      filters/varYSizeGaussianFilter.py:46)                   -> orc_conv2d, orc_sepconv2d,
 *                                                                 orc_gaussian_kernel1d
 *   - imgProcessor/filters/varYSizeGaussianFilter.py:53-68     -> orc_conv_ydep
 *   - imgProcessor/filters/standardDeviation.py:34-70          -> orc_std2d
 *   - imgProcessor/filters/maskedFilter.py:43-72 (mean)        -> orc_masked_mean
 *   - imgProcessor/filters/nan_maximum_filter.py:17-37         -> orc_nan_max
 *   - imgProcessor/filters/medianThreshold.py:7-30, camera/CameraCalibration.py:416-437
 *                                                              -> orc_median_threshold, orc_calib_prefilter
 *   - imgProcessor/render/closestDirectDistance.py:17-41       -> orc_closest_distance
 *   - imgProcessor/uncertainty/positionToIntensityUncertainty.py:7-49
 *                                                              -> orc_pos_to_intensity_unc
 *   - imgProcessor/interpolate/interpolate2dStructuredIDW.py:26-65      -> orc_idw
 *   - imgProcessor/interpolate/interpolate2dStructuredFastIDW.py:29-63  -> orc_fast_idw
 *   - imgProcessor/interpolate/interpolate2dStructuredPointSpreadIDW.py:31-141 -> orc_point_spread_idw
 *   - imgProcessor/camera/LensDistortion.py:316-330,342-358 (cv2.remap,
 *     cv2.initUndistortRectifyMap)                             -> orc_remap, orc_build_undistort_map
 *   - imgProcessor/camera/PerspectiveCorrection.py:374-406 (cv2.warpPerspective)
 *                                                              -> orc_warp_perspective
 *
 * Pinning status:
 *   - in-tree stencils: pinned against the reference itself, imported here
 *     through a numba identity shim (tests/golden/gen_golden.py, fixtures in
 *     tests/golden/).
 *   - remap INTER_LINEAR (exact coordinates): pinned against
 *     scipy.ndimage.map_coordinates(order=1, mode='grid-constant') and
 *     skimage.transform.warp(order=1).  Keys bicubic a=-0.5: pinned against
 *     skimage.transform.warp(order=3).
 *   - OpenCV-specific modes (1/32-px coordinate quantisation "q5", bicubic
 *     a=-0.75, Lanczos4, the uint8 fixed-point path): cv2 is a third-party,
 *     un-vendored, un-pinned dependency (setup.py:40) that is not importable
 *     in the build container -> these follow OpenCV's published algorithm
 *     (imgproc/imgwarp.cpp: remapBilinear/remapBicubic/remapLanczos4,
 *     interpolateCubic, interpolateLanczos4, initInterTab1D) and are
 *     **parity unpinned**.
 *
 * Arithmetic: every floating-point accumulation is done in double (that is
 * what numba's type unification, scipy.ndimage and skimage's _warp_fast do),
 * then rounded once to the output dtype.  The HIP path computes float32
 * images in float32; tests compare at the 1e-5 relative tolerance of
 * BASELINE.json.  The uint8->uint8 path is integer fixed point and bit-exact.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_U8 0
#define ORC_U16 1
#define ORC_F32 2
#define ORC_F64 3

/* interpolation ids follow cv2's numbering where one exists */
#define ORC_NEAREST 0
#define ORC_LINEAR 1
#define ORC_CUBIC_CV 2   /* Keys a = -0.75 (cv2.INTER_CUBIC) */
#define ORC_LANCZOS4 4   /* cv2.INTER_LANCZOS4, always q5 */
#define ORC_CUBIC_KEYS 5 /* Keys a = -0.5 (skimage order=3) */
#define ORC_Q5 0x100     /* coordinates rounded to 1/32 px like cv2 */

/* border ids follow cv2 */
#define ORC_CONSTANT 0
#define ORC_REPLICATE 1
#define ORC_REFLECT 2 /* fedcba|abcdef  == numpy 'symmetric' == scipy 'reflect' */
#define ORC_WRAP 3
#define ORC_REFLECT101 4 /* scipy 'mirror' */

static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static inline size_t dtype_size(int dt) {
  switch (dt) {
    case ORC_U8: return 1;
    case ORC_U16: return 2;
    case ORC_F32: return 4;
    case ORC_F64: return 8;
  }
  return 0;
}

static inline double load_px(const void* p, int dt, long idx) {
  switch (dt) {
    case ORC_U8: return ((const uint8_t*)p)[idx];
    case ORC_U16: return ((const uint16_t*)p)[idx];
    case ORC_F32: return ((const float*)p)[idx];
    default: return ((const double*)p)[idx];
  }
}

/* cv::saturate_cast<uchar/ushort>(double) = round-half-even, then clamp */
static inline void store_px(void* p, int dt, long idx, double v) {
  switch (dt) {
    case ORC_U8: {
      double r = nearbyint(v);
      if (!(r > 0)) r = 0; /* NaN -> 0 */
      if (r > 255) r = 255;
      ((uint8_t*)p)[idx] = (uint8_t)r;
      break;
    }
    case ORC_U16: {
      double r = nearbyint(v);
      if (!(r > 0)) r = 0;
      if (r > 65535) r = 65535;
      ((uint16_t*)p)[idx] = (uint16_t)r;
      break;
    }
    case ORC_F32: ((float*)p)[idx] = (float)v; break;
    default: ((double*)p)[idx] = v; break;
  }
}

/* index resolution for an out-of-range coordinate; -1 means "use cval" */
static inline long resolve_idx(long i, long n, int mode) {
  if (i >= 0 && i < n) return i;
  switch (mode) {
    case ORC_CONSTANT: return -1;
    case ORC_REPLICATE: return i < 0 ? 0 : n - 1;
    case ORC_REFLECT: {
      if (n == 1) return 0;
      long p = 2 * n;
      long m = i % p;
      if (m < 0) m += p;
      return m < n ? m : p - 1 - m;
    }
    case ORC_REFLECT101: {
      if (n == 1) return 0;
      long p = 2 * n - 2;
      long m = i % p;
      if (m < 0) m += p;
      return m < n ? m : p - m;
    }
    case ORC_WRAP: {
      long m = i % n;
      if (m < 0) m += n;
      return m;
    }
  }
  return -1;
}

/* ------------------------------------------------------------------ */
/* interpolation weights                                               */
/* ------------------------------------------------------------------ */

/* OpenCV interpolateCubic (imgwarp.cpp), generalised over A */
static void cubic_weights(double t, double A, double* w) {
  w[0] = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A;
  w[1] = ((A + 2) * t - (A + 3)) * t * t + 1;
  w[2] = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1;
  w[3] = 1.0 - w[0] - w[1] - w[2];
}

/* OpenCV interpolateLanczos4 (imgwarp.cpp): float coefficients */
void orc_lanczos4_weights(float x, float* coeffs) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0},  {-s45, -s45}, {0, 1},  {s45, -s45},
                                 {-1, 0}, {s45, s45},   {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; i++) coeffs[i] = 0;
    coeffs[3] = 1;
    return;
  }
  float sum = 0;
  double y0 = -(x + 3) * M_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
  for (int i = 0; i < 8; i++) {
    double y = -(x + 3 - i) * M_PI * 0.25;
    coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += coeffs[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

typedef struct {
  const void* src;
  int dt;
  long h, w, pitch;
  int interp; /* base id without Q5 */
  int q5;
  int border;
  double cval;
} sampler_t;

static inline int ntaps_of(int interp) {
  switch (interp) {
    case ORC_NEAREST: return 1;
    case ORC_LINEAR: return 2;
    case ORC_LANCZOS4: return 8;
    default: return 4;
  }
}

/* split a coordinate into first-tap index and fractional weights */
static inline void axis_weights(const sampler_t* s, double c, long* i0, double* w) {
  int interp = s->interp;
  double fl, t;
  if (s->q5 || interp == ORC_LANCZOS4) {
    /* cvRound(c * INTER_TAB_SIZE): round-half-even at 1/32 px */
    double q = nearbyint(c * 32.0);
    if (q > 2147483647.0) q = 2147483647.0;
    if (q < -2147483648.0) q = -2147483648.0;
    long qi = (long)q;
    long ip = qi >> 5; /* arithmetic shift == floor */
    fl = (double)ip;
    t = (double)(qi & 31) / 32.0;
  } else {
    fl = floor(c);
    t = c - fl;
  }
  switch (interp) {
    case ORC_NEAREST:
      /* cv2.remap INTER_NEAREST: cvRound(coordinate) */
      *i0 = (long)nearbyint(c);
      w[0] = 1.0;
      break;
    case ORC_LINEAR:
      *i0 = (long)fl;
      w[0] = 1.0 - t;
      w[1] = t;
      break;
    case ORC_CUBIC_CV:
      *i0 = (long)fl - 1;
      cubic_weights(t, -0.75, w);
      break;
    case ORC_CUBIC_KEYS:
      *i0 = (long)fl - 1;
      cubic_weights(t, -0.5, w);
      break;
    case ORC_LANCZOS4: {
      float cf[8];
      orc_lanczos4_weights((float)t, cf);
      *i0 = (long)fl - 3;
      for (int k = 0; k < 8; k++) w[k] = cf[k];
      break;
    }
  }
}

static double sample(const sampler_t* s, double sx, double sy) {
  /* NaN / absurd coordinates behave like "far outside" */
  if (!(sx > -1e6 && sx < 1e6 && sy > -1e6 && sy < 1e6)) {
    if (s->border == ORC_CONSTANT || sx != sx || sy != sy) return s->cval;
    sx = sx < -1e6 ? -1e6 : (sx > 1e6 ? 1e6 : sx);
    sy = sy < -1e6 ? -1e6 : (sy > 1e6 ? 1e6 : sy);
  }
  int n = ntaps_of(s->interp);
  long ix0 = 0, iy0 = 0;
  double wx[8], wy[8];
  axis_weights(s, sx, &ix0, wx);
  axis_weights(s, sy, &iy0, wy);
  if (s->border == ORC_CONSTANT &&
      (ix0 >= s->w || ix0 + n <= 0 || iy0 >= s->h || iy0 + n <= 0))
    return s->cval; /* whole footprint outside */
  double out = 0.0;
  for (int r = 0; r < n; r++) {
    long yy = resolve_idx(iy0 + r, s->h, s->border);
    double rs = 0.0;
    for (int c = 0; c < n; c++) {
      long xx = resolve_idx(ix0 + c, s->w, s->border);
      double v = (yy < 0 || xx < 0) ? s->cval : load_px(s->src, s->dt, yy * s->pitch + xx);
      rs += wx[c] * v;
    }
    out += wy[r] * rs;
  }
  return out;
}

/* OpenCV's uint8 bilinear: q5 coordinates, 15-bit integer weights
 * (BilinearTab_i = (32-fx)(32-fy)*32 etc., exact), rounded shift. */
static uint8_t sample_u8_fixed(const sampler_t* s, double sx, double sy, uint8_t cv8) {
  if (!(sx > -1e6 && sx < 1e6 && sy > -1e6 && sy < 1e6)) {
    if (s->border == ORC_CONSTANT || sx != sx || sy != sy) return cv8;
    sx = sx < -1e6 ? -1e6 : (sx > 1e6 ? 1e6 : sx);
    sy = sy < -1e6 ? -1e6 : (sy > 1e6 ? 1e6 : sy);
  }
  long qx = (long)nearbyint(sx * 32.0), qy = (long)nearbyint(sy * 32.0);
  long ix = qx >> 5, iy = qy >> 5;
  int fx = (int)(qx & 31), fy = (int)(qy & 31);
  int w[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
  int acc = 0;
  for (int r = 0; r < 2; r++) {
    long yy = resolve_idx(iy + r, s->h, s->border);
    for (int c = 0; c < 2; c++) {
      long xx = resolve_idx(ix + c, s->w, s->border);
      int v = (yy < 0 || xx < 0) ? cv8 : ((const uint8_t*)s->src)[yy * s->pitch + xx];
      acc += v * w[r * 2 + c];
    }
  }
  int o = (acc + (1 << 14)) >> 15;
  return (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
}

/* OpenCV's uint8 bicubic (A = -0.75) and Lanczos4 (imgwarp.cpp: initInterTab1D / initInterTab2D with
 * fixpt = true, remapBicubic / remapLanczos4 with FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS>),
 * restated from the published algorithm - cv2 cannot be installed here, so this is unpinned like
 * the other cv2 modes:
 *   1-D weights in float32 for the 32 fractions (interpolateCubic / interpolateLanczos4);
 *   2-D weights  saturate_cast<short>(wy[k1] * wx[k2] * 32768)  (float products, cvRound);
 *   if their sum is not 32768 the difference goes to ONE weight of the 2 x 2 block starting at
 *   (ksize/2, ksize/2): the largest of the block when the sum is too small, the smallest when
 *   it is too large (first such entry in row-major order);
 *   integer accumulation, (sum + 2^14) >> 15, saturate to uint8. */
static float g_tab_cubic[32][4], g_tab_lanczos[32][8];
static int g_tabs_ready = 0;

static void interpolate_cubic_f32(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

static void init_fixed_tabs(void) {
  if (g_tabs_ready) return;
  const float scale = 1.f / 32;
  for (int i = 0; i < 32; i++) {
    interpolate_cubic_f32(i * scale, g_tab_cubic[i]);
    orc_lanczos4_weights(i * scale, g_tab_lanczos[i]);
  }
  g_tabs_ready = 1;
}

/* the 1-D float tables (for the tests): which = 0 bicubic (32 x 4), 1 Lanczos4 (32 x 8) */
void orc_fixed_tab1d(int which, float* out) {
  init_fixed_tabs();
  if (which == 0) memcpy(out, g_tab_cubic, sizeof(g_tab_cubic));
  else memcpy(out, g_tab_lanczos, sizeof(g_tab_lanczos));
}

static inline int sat_short(float v) {
  double r = nearbyint((double)v); /* cvRound: half to even */
  if (r < -32768.0) r = -32768.0;
  if (r > 32767.0) r = 32767.0;
  return (int)r;
}

/* the ks x ks integer weights of fraction pair (fy, fx) */
void orc_fixed_weights_2d(int ks, int fy, int fx, int* itab) {
  init_fixed_tabs();
  const float* ty = ks == 4 ? g_tab_cubic[fy] : g_tab_lanczos[fy];
  const float* tx = ks == 4 ? g_tab_cubic[fx] : g_tab_lanczos[fx];
  int isum = 0;
  for (int k1 = 0; k1 < ks; k1++) {
    float vy = ty[k1];
    for (int k2 = 0; k2 < ks; k2++) {
      float v = vy * tx[k2];
      isum += itab[k1 * ks + k2] = sat_short(v * 32768.f);
    }
  }
  if (isum != 32768) {
    int diff = isum - 32768;
    int h = ks / 2, Mk1 = h, Mk2 = h, mk1 = h, mk2 = h;
    for (int k1 = h; k1 < h + 2; k1++)
      for (int k2 = h; k2 < h + 2; k2++) {
        if (itab[k1 * ks + k2] < itab[mk1 * ks + mk2]) { mk1 = k1; mk2 = k2; }
        else if (itab[k1 * ks + k2] > itab[Mk1 * ks + Mk2]) { Mk1 = k1; Mk2 = k2; }
      }
    if (diff < 0) itab[Mk1 * ks + Mk2] = (short)(itab[Mk1 * ks + Mk2] - diff);
    else itab[mk1 * ks + mk2] = (short)(itab[mk1 * ks + mk2] - diff);
  }
}

static uint8_t sample_u8_tab(const sampler_t* s, double sx, double sy, uint8_t cv8) {
  const int ks = s->interp == ORC_LANCZOS4 ? 8 : 4;
  if (!(sx > -1e6 && sx < 1e6 && sy > -1e6 && sy < 1e6)) {
    if (s->border == ORC_CONSTANT || sx != sx || sy != sy) return cv8;
    sx = sx < -1e6 ? -1e6 : (sx > 1e6 ? 1e6 : sx);
    sy = sy < -1e6 ? -1e6 : (sy > 1e6 ? 1e6 : sy);
  }
  long qx = (long)nearbyint(sx * 32.0), qy = (long)nearbyint(sy * 32.0);
  long ix0 = (qx >> 5) - (ks / 2 - 1), iy0 = (qy >> 5) - (ks / 2 - 1);
  if (s->border == ORC_CONSTANT &&
      (ix0 >= s->w || ix0 + ks <= 0 || iy0 >= s->h || iy0 + ks <= 0))
    return cv8; /* whole footprint outside */
  int itab[64];
  orc_fixed_weights_2d(ks, (int)(qy & 31), (int)(qx & 31), itab);
  int acc = 0;
  for (int r = 0; r < ks; r++) {
    long yy = resolve_idx(iy0 + r, s->h, s->border);
    for (int c = 0; c < ks; c++) {
      long xx = resolve_idx(ix0 + c, s->w, s->border);
      int v = (yy < 0 || xx < 0) ? cv8 : ((const uint8_t*)s->src)[yy * s->pitch + xx];
      acc += v * itab[r * ks + c];
    }
  }
  int o = (acc + (1 << 14)) >> 15;
  return (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
}

/* OpenCV's remap arithmetic on CV_16U (cv2.remap / warpPerspective on the camera's uint16 frames:
 * camera/LensDistortion.py:323-326, camera/PerspectiveCorrection.py:401-405), restated from the
 * published algorithm - imgwarp.cpp remapBilinear / remapBicubic / remapLanczos4 with
 * Cast<float, ushort> and float weights (cv2 itself is absent here: unpinned, and the expression
 * order below is from memory of the OpenCV 4.x source):
 *   coordinates to 1/32 px; 1-D float32 weights of the 32 fractions; the 2-D weight of a tap the
 *   float32 product wy[r] * wx[c]; float32 accumulation (every product and sum rounded, no fma):
 *     bilinear         v00 w00 + v01 w01 + v10 w10 + v11 w11 left to right, taps outside the frame
 *                      replaced by the border value;
 *     bicubic inside   the 16 products summed left to right in row-major order;
 *     Lanczos4 inside  per tap row the 8 products summed left to right, the row sums added;
 *     bicubic / Lanczos4 with taps outside:  sum = cv;  sum += (S - cv) * w  for every tap that
 *                      exists after the border mode;
 *   saturate_cast<ushort>: round half to even, clamp. */
static uint16_t sample_u16_cv(const sampler_t* s, double sx, double sy, uint16_t cv16) {
  init_fixed_tabs();
  const int ks = s->interp == ORC_LANCZOS4 ? 8 : (s->interp == ORC_LINEAR ? 2 : 4);
  if (!(sx > -1e6 && sx < 1e6 && sy > -1e6 && sy < 1e6)) {
    if (s->border == ORC_CONSTANT || sx != sx || sy != sy) return cv16;
    sx = sx < -1e6 ? -1e6 : (sx > 1e6 ? 1e6 : sx);
    sy = sy < -1e6 ? -1e6 : (sy > 1e6 ? 1e6 : sy);
  }
  long qx = (long)nearbyint(sx * 32.0), qy = (long)nearbyint(sy * 32.0);
  long ix0 = (qx >> 5) - (ks / 2 - 1), iy0 = (qy >> 5) - (ks / 2 - 1);
  if (s->border == ORC_CONSTANT &&
      (ix0 >= s->w || ix0 + ks <= 0 || iy0 >= s->h || iy0 + ks <= 0))
    return cv16;
  float wx[8], wy[8];
  if (ks == 2) {
    float fx = (float)(qx & 31) * 0.03125f, fy = (float)(qy & 31) * 0.03125f;
    wx[0] = 1.f - fx; wx[1] = fx; wy[0] = 1.f - fy; wy[1] = fy;
  } else {
    const float* tx = ks == 4 ? g_tab_cubic[qx & 31] : g_tab_lanczos[qx & 31];
    const float* ty = ks == 4 ? g_tab_cubic[qy & 31] : g_tab_lanczos[qy & 31];
    for (int k = 0; k < ks; k++) { wx[k] = tx[k]; wy[k] = ty[k]; }
  }
  const uint16_t* src = (const uint16_t*)s->src;
  const float cv = (float)cv16;
  const int inside = ix0 >= 0 && iy0 >= 0 && ix0 + ks <= s->w && iy0 + ks <= s->h;
  float sum;
  if (ks == 2) {
    float v[2][2];
    for (int r = 0; r < 2; r++) {
      long yy = resolve_idx(iy0 + r, s->h, s->border);
      for (int c = 0; c < 2; c++) {
        long xx = resolve_idx(ix0 + c, s->w, s->border);
        v[r][c] = (yy < 0 || xx < 0) ? cv : (float)src[yy * s->pitch + xx];
      }
    }
    sum = v[0][0] * (wy[0] * wx[0]);
    sum = sum + v[0][1] * (wy[0] * wx[1]);
    sum = sum + v[1][0] * (wy[1] * wx[0]);
    sum = sum + v[1][1] * (wy[1] * wx[1]);
  } else if (inside) {
    sum = 0.f;
    for (int r = 0; r < ks; r++) {
      float rs = 0.f;
      for (int c = 0; c < ks; c++) {
        float pr = (float)src[(iy0 + r) * s->pitch + ix0 + c] * (wy[r] * wx[c]);
        if (ks == 4) sum = (r == 0 && c == 0) ? pr : sum + pr;
        else rs = c == 0 ? pr : rs + pr;
      }
      if (ks != 4) sum = sum + rs;
    }
  } else {
    sum = cv;
    for (int r = 0; r < ks; r++) {
      long yy = resolve_idx(iy0 + r, s->h, s->border);
      if (yy < 0) continue;
      for (int c = 0; c < ks; c++) {
        long xx = resolve_idx(ix0 + c, s->w, s->border);
        if (xx >= 0) sum = sum + ((float)src[yy * s->pitch + xx] - cv) * (wy[r] * wx[c]);
      }
    }
  }
  double r = nearbyint((double)sum);
  if (!(r > 0)) r = 0;
  if (r > 65535) r = 65535;
  return (uint16_t)r;
}

static inline uint16_t sat_u16cv(double v) {
  double r = nearbyint(v);
  if (!(r > 0)) r = 0;
  if (r > 65535) r = 65535;
  return (uint16_t)r;
}

/* integer frames in cv2's own arithmetic: 0 = floating point path, 1 = uint8 bilinear fixed point,
 * 2 = uint8 table fixed point, 3 = uint16 float tables (the cv2 modes: 1/32-px coordinates) */
static int fixed_kind(const sampler_t* s, int src_dt, int dst_dt) {
  if (src_dt == ORC_U16 && dst_dt == ORC_U16) {
    if (s->interp == ORC_LANCZOS4 ||
        (s->q5 && (s->interp == ORC_LINEAR || s->interp == ORC_CUBIC_CV)))
      return 3;
    return 0;
  }
  if (src_dt != ORC_U8 || dst_dt != ORC_U8) return 0;
  if (s->interp == ORC_LINEAR) return 1;
  if (s->interp == ORC_CUBIC_CV || s->interp == ORC_LANCZOS4) {
    init_fixed_tabs();
    return 2;
  }
  return 0;
}

static inline uint8_t sat_u8(double v) {
  double r = nearbyint(v);
  if (!(r > 0)) r = 0;
  if (r > 255) r = 255;
  return (uint8_t)r;
}

static void init_sampler(sampler_t* s, const void* src, int dt, long h, long w, long pitch,
                         int interp, int border, double cval) {
  s->src = src; s->dt = dt; s->h = h; s->w = w; s->pitch = pitch;
  s->interp = interp & 0xff;
  s->q5 = (interp & ORC_Q5) != 0;
  s->border = border; s->cval = cval;
}

/* cv2.remap(src, mapx, mapy, interp, borderMode, borderValue) —
 * camera/LensDistortion.py:323-326 */
int orc_remap(const void* src, int src_dt, long sh, long sw, long src_pitch, const float* mapx,
              const float* mapy, long map_pitch, void* dst, int dst_dt, long dh, long dw,
              long dst_pitch, int interp, int border, double cval) {
  sampler_t s;
  init_sampler(&s, src, src_dt, sh, sw, src_pitch, interp, border, cval);
  int fixed = fixed_kind(&s, src_dt, dst_dt);
  uint8_t cv8 = sat_u8(cval);
  uint16_t cv16 = sat_u16cv(cval);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long v = 0; v < dh; v++) {
    for (long u = 0; u < dw; u++) {
      double sx = mapx[v * map_pitch + u], sy = mapy[v * map_pitch + u];
      if (fixed == 3)
        ((uint16_t*)dst)[v * dst_pitch + u] = sample_u16_cv(&s, sx, sy, cv16);
      else if (fixed)
        ((uint8_t*)dst)[v * dst_pitch + u] =
            fixed == 1 ? sample_u8_fixed(&s, sx, sy, cv8) : sample_u8_tab(&s, sx, sy, cv8);
      else
        store_px(dst, dst_dt, v * dst_pitch + u, sample(&s, sx, sy));
    }
  }
  return 0;
}

/* 3x3 inverse (double) */
static int inv3(const double* m, double* o) {
  double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + b * B + c * C;
  if (det == 0) return -1;
  double id = 1.0 / det;
  o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
  o[3] = B * id; o[4] = (a * i - c * g) * id;  o[5] = -(a * f - c * d) * id;
  o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
  return 0;
}

/* the distortion model of cv2.initUndistortRectifyMap with R = I
 * (camera/LensDistortion.py:355-357); dist = [k1,k2,p1,p2,k3] (:370,380) */
static inline void undistort_coord(const double* K, const double* d, const double* ir, long u,
                                   long v, double* sx, double* sy) {
  double _x = ir[0] * u + ir[1] * v + ir[2];
  double _y = ir[3] * u + ir[4] * v + ir[5];
  double _w = ir[6] * u + ir[7] * v + ir[8];
  double iw = 1.0 / _w;
  double x = _x * iw, y = _y * iw;
  double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
  double kr = 1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2;
  double xd = x * kr + d[2] * _2xy + d[3] * (r2 + 2 * x2);
  double yd = y * kr + d[2] * (r2 + 2 * y2) + d[3] * _2xy;
  *sx = K[0] * xd + K[2];
  *sy = K[4] * yd + K[5];
}

int orc_build_undistort_map(const double* K, const double* dist5, const double* newK, long h,
                            long w, float* mapx, float* mapy, long map_pitch) {
  double ir[9];
  if (inv3(newK, ir)) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long v = 0; v < h; v++)
    for (long u = 0; u < w; u++) {
      double sx, sy;
      undistort_coord(K, dist5, ir, u, v, &sx, &sy);
      mapx[v * map_pitch + u] = (float)sx;
      mapy[v * map_pitch + u] = (float)sy;
    }
  return 0;
}

/* analytic undistort = build float32 map value on the fly, then remap */
int orc_undistort(const void* src, int src_dt, long sh, long sw, long src_pitch, const double* K,
                  const double* dist5, const double* newK, void* dst, int dst_dt, long dh, long dw,
                  long dst_pitch, int interp, int border, double cval) {
  sampler_t s;
  init_sampler(&s, src, src_dt, sh, sw, src_pitch, interp, border, cval);
  double ir[9];
  if (inv3(newK, ir)) return -1;
  int fixed = fixed_kind(&s, src_dt, dst_dt);
  uint8_t cv8 = sat_u8(cval);
  uint16_t cv16 = sat_u16cv(cval);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long v = 0; v < dh; v++)
    for (long u = 0; u < dw; u++) {
      double sx, sy;
      undistort_coord(K, dist5, ir, u, v, &sx, &sy);
      sx = (double)(float)sx; /* CV_32FC1 map storage */
      sy = (double)(float)sy;
      if (fixed == 3)
        ((uint16_t*)dst)[v * dst_pitch + u] = sample_u16_cv(&s, sx, sy, cv16);
      else if (fixed)
        ((uint8_t*)dst)[v * dst_pitch + u] =
            fixed == 1 ? sample_u8_fixed(&s, sx, sy, cv8) : sample_u8_tab(&s, sx, sy, cv8);
      else
        store_px(dst, dst_dt, v * dst_pitch + u, sample(&s, sx, sy));
    }
  return 0;
}

/* cv2.warpPerspective: dst(x,y) = src(M·(x,y,1)) with M the dst->src matrix
 * (inverse of H unless WARP_INVERSE_MAP) — camera/PerspectiveCorrection.py:377-378,401-405.
 * Coordinates are kept in double (skimage _warp_fast semantics); the q5 flag
 * applies cv2's 1/32-px rounding. */
int orc_warp_perspective(const void* src, int src_dt, long sh, long sw, long src_pitch,
                         const double* M, void* dst, int dst_dt, long dh, long dw, long dst_pitch,
                         int interp, int border, double cval) {
  sampler_t s;
  init_sampler(&s, src, src_dt, sh, sw, src_pitch, interp, border, cval);
  int fixed = fixed_kind(&s, src_dt, dst_dt);
  uint8_t cv8 = sat_u8(cval);
  uint16_t cv16 = sat_u16cv(cval);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long v = 0; v < dh; v++)
    for (long u = 0; u < dw; u++) {
      double X = M[0] * u + M[1] * v + M[2];
      double Y = M[3] * u + M[4] * v + M[5];
      double W = M[6] * u + M[7] * v + M[8];
      double sx, sy;
      if (W != 0) { double iw = 1.0 / W; sx = X * iw; sy = Y * iw; }
      else { sx = 0; sy = 0; } /* cv2: W ? 1/W : 0 */
      if (fixed == 3)
        ((uint16_t*)dst)[v * dst_pitch + u] = sample_u16_cv(&s, sx, sy, cv16);
      else if (fixed)
        ((uint8_t*)dst)[v * dst_pitch + u] =
            fixed == 1 ? sample_u8_fixed(&s, sx, sy, cv8) : sample_u8_tab(&s, sx, sy, cv8);
      else
        store_px(dst, dst_dt, v * dst_pitch + u, sample(&s, sx, sy));
    }
  return 0;
}

/* ------------------------------------------------------------------ */
/* filters                                                             */
/* ------------------------------------------------------------------ */

/* filters/_extendArrayForConvolution.py:5-97.  kx,ky are the *kernel sizes*
 * along x (columns) and y (rows); padding is k//2 per side.  'reflect'
 * repeats the edge pixel (numpy 'symmetric'); modex may be 'wrap'.
 * out has shape (h + 2*(ky/2), w + 2*(kx/2)), dense. */
int orc_extend_array(const void* arr, int dt, long h, long w, long kx, long ky, int modex,
                     int modey, void* out) {
  long px = kx / 2, py = ky / 2;
  if (!(py < h && px < w)) return -2; /* the reference asserts */
  if (modey != ORC_REFLECT) return -3; /* modey=='wrap' raises in the reference (:57 typo) */
  if (modex != ORC_REFLECT && modex != ORC_WRAP) return -3;
  long ow = w + 2 * px, oh = h + 2 * py;
  size_t es = dtype_size(dt);
  for (long y = 0; y < oh; y++) {
    long sy = resolve_idx(y - py, h, modey);
    for (long x = 0; x < ow; x++) {
      long sx = resolve_idx(x - px, w, modex);
      memcpy((char*)out + (y * ow + x) * es, (const char*)arr + (sy * w + sx) * es, es);
    }
  }
  return 0;
}

/* centred correlation, scipy.ndimage.correlate semantics (origin 0, centre at
 * k//2), per-axis border mode, optional mask (unmasked -> 0), double accumulate */
int orc_conv2d(const void* src, int dt, long h, long w, long pitch, const double* kern, long kh,
               long kw, const uint8_t* mask, long mask_pitch, void* dst, long dst_pitch,
               int border_x, int border_y, double cval) {
  long cy = kh / 2, cx = kw / 2;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long y = 0; y < h; y++)
    for (long x = 0; x < w; x++) {
      if (mask && !mask[y * mask_pitch + x]) { store_px(dst, dt, y * dst_pitch + x, 0.0); continue; }
      double acc = 0.0;
      for (long i = 0; i < kh; i++) {
        long yy = resolve_idx(y + i - cy, h, border_y);
        for (long j = 0; j < kw; j++) {
          long xx = resolve_idx(x + j - cx, w, border_x);
          double v = (yy < 0 || xx < 0) ? cval : load_px(src, dt, yy * pitch + xx);
          acc += kern[i * kw + j] * v;
        }
      }
      store_px(dst, dt, y * dst_pitch + x, acc);
    }
  return 0;
}

/* filters/maskedConvolve.py:13-43 as written: pad with extendArray, then
 * kernel[ii,jj] with ii,jj in [-h..h] — NEGATIVE indices wrap (python), so
 * the effective centred kernel is np.fft.fftshift(kernel).  Square odd kernels only. */
int orc_masked_convolve(const void* arr, int dt, long h, long w, const double* kern, long k,
                        const uint8_t* mask, int mode, void* out) {
  if (k % 2 == 0) return -2;
  long hk = k / 2;
  /* the wrapper passes modex=modey=mode and modey=='wrap' raises (:57 typo):
   * 'reflect' is the only mode the reference function can run with */
  if (mode != ORC_REFLECT) return -3;
  int my = mode, mx = mode;
  long ow = w + 2 * hk, oh = h + 2 * hk;
  size_t es = dtype_size(dt);
  void* ext = malloc((size_t)ow * oh * es);
  if (!ext) return -1;
  int rc = orc_extend_array(arr, dt, h, w, k, k, mx, my, ext);
  if (rc) { free(ext); return rc; }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long i = hk; i < oh - hk; i++)
    for (long j = hk; j < ow - hk; j++) {
      long oi = i - hk, oj = j - hk;
      if (!mask[oi * w + oj]) { store_px(out, dt, oi * w + oj, 0.0); continue; }
      double val = 0;
      for (long ii = -hk; ii <= hk; ii++)
        for (long jj = -hk; jj <= hk; jj++) {
          long ki = ii < 0 ? ii + k : ii, kj = jj < 0 ? jj + k : jj; /* python negative index */
          val += kern[ki * k + kj] * load_px(ext, dt, (i + ii) * ow + (j + jj));
        }
      store_px(out, dt, oi * w + oj, val);
    }
  free(ext);
  return 0;
}

/* scipy.ndimage._filters._gaussian_kernel1d(sigma, order=0, radius) */
int orc_gaussian_kernel1d(double sigma, long radius, double* out) {
  double s2 = sigma * sigma, sum = 0;
  for (long i = -radius; i <= radius; i++) {
    out[i + radius] = exp(-0.5 / s2 * (double)(i * i));
    sum += out[i + radius];
  }
  for (long i = 0; i <= 2 * radius; i++) out[i] /= sum;
  return 0;
}

/* scipy.ndimage.gaussian_filter-style separable correlation: axis 0 (y) first
 * with ky, result stored in the image dtype, then axis 1 (x) with kx.
 * nky==0 or nkx==0 skips that axis (scipy skips sigma<=1e-15). */
int orc_sepconv2d(const void* src, int dt, long h, long w, long pitch, const double* ky, long nky,
                  const double* kx, long nkx, void* dst, long dst_pitch, int border_y,
                  int border_x, double cval) {
  size_t es = dtype_size(dt);
  void* tmp = malloc((size_t)h * w * es);
  if (!tmp) return -1;
  long cy = nky / 2, cx = nkx / 2;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long y = 0; y < h; y++)
    for (long x = 0; x < w; x++) {
      double acc;
      if (nky > 0) {
        acc = 0;
        for (long i = 0; i < nky; i++) {
          long yy = resolve_idx(y + i - cy, h, border_y);
          acc += ky[i] * (yy < 0 ? cval : load_px(src, dt, yy * pitch + x));
        }
      } else acc = load_px(src, dt, y * pitch + x);
      store_px(tmp, dt, y * w + x, acc);
    }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long y = 0; y < h; y++)
    for (long x = 0; x < w; x++) {
      double acc;
      if (nkx > 0) {
        acc = 0;
        for (long j = 0; j < nkx; j++) {
          long xx = resolve_idx(x + j - cx, w, border_x);
          acc += kx[j] * (xx < 0 ? cval : load_px(tmp, dt, y * w + xx));
        }
      } else acc = load_px(tmp, dt, y * w + x);
      store_px(dst, dt, y * dst_pitch + x, acc);
    }
  free(tmp);
  return 0;
}

/* filters/varYSizeGaussianFilter.py:53-68 restricted to its well-defined
 * output range: out[r,c] = sum_{ii<k0,jj<k1} kernels[r,ii,jj]*ext[r+ii,c+jj],
 * NaN pixels skipped.  ext is the (h+2*(k0/2)) x (w+2*(k1/2)) padded array. */
int orc_conv_ydep(const void* ext, int dt, long h, long w, const double* kernels, long k0, long k1,
                  void* out) {
  long ow = w + 2 * (k1 / 2);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long r = 0; r < h; r++)
    for (long c = 0; c < w; c++) {
      double v = 0;
      for (long ii = 0; ii < k0; ii++)
        for (long jj = 0; jj < k1; jj++) {
          double a = load_px(ext, dt, (r + ii) * ow + (c + jj));
          if (a == a) v += kernels[(r * k0 + ii) * k1 + jj] * a;
        }
      store_px(out, dt, r * w + c, v);
    }
  return 0;
}

/* filters/standardDeviation.py:34-70, quirks included: window
 * [i-h, min(i+h,gx)) x [j-h, min(j+h,gy)), divisor = (n_i-1)*(n_j-1) (last loop indices) */
int orc_std2d(const void* img, int dt, long gx, long gy, long ksx, long ksy, const void* blurred,
              void* std) {
  long hkx = ksx / 2, hky = ksy / 2;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long i = 0; i < gx; i++)
    for (long j = 0; j < gy; j++) {
      long xmn = i - hkx < 0 ? 0 : i - hkx, xmx = i + hkx > gx ? gx : i + hkx;
      long ymn = j - hky < 0 ? 0 : j - hky, ymx = j + hky > gy ? gy : j + hky;
      double val = 0, mean = load_px(blurred, dt, i * gy + j);
      for (long ii = 0; ii < xmx - xmn; ii++)
        for (long jj = 0; jj < ymx - ymn; jj++) {
          double d = load_px(img, dt, (xmn + ii) * gy + (ymn + jj)) - mean;
          val += d * d;
        }
      double npx = (double)((xmx - xmn - 1) * (ymx - ymn - 1));
      store_px(std, dt, i * gy + j, sqrt(val / npx));
    }
  return 0;
}

/* interpolate/interpolate2dStructuredIDW.py:26-65.  In-place on grid.
 * The reference clamps xmx to gx (not gx-1) and then reads index gx: out of
 * bounds (UB under numba).  Here the window is clamped to the array, which is
 * identical wherever the reference is well defined. */
int orc_idw(void* grid, int dt, const uint8_t* mask, long gx, long gy, long kernel,
            const double* weights) {
  long kw = 2 * kernel + 1;
  size_t es = dtype_size(dt);
  void* in = malloc((size_t)gx * gy * es);
  if (!in) return -1;
  memcpy(in, grid, (size_t)gx * gy * es); /* masked px are never read: copy == in-place */
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
  for (long i = 0; i < gx; i++)
    for (long j = 0; j < gy; j++) {
      if (!mask[i * gy + j]) continue;
      long xmn = i - kernel < 0 ? 0 : i - kernel, xmx = i + kernel > gx - 1 ? gx - 1 : i + kernel;
      long ymn = j - kernel < 0 ? 0 : j - kernel, ymx = j + kernel > gy - 1 ? gy - 1 : j + kernel;
      double sumWi = 0, value = 0;
      for (long xi = xmn; xi <= xmx; xi++)
        for (long yi = ymn; yi <= ymx; yi++)
          if ((xi != i || yi != j) && !mask[xi * gy + yi]) {
            double wi = weights[(xi - i + kernel) * kw + (yi - j + kernel)];
            sumWi += wi;
            value += wi * load_px(in, dt, xi * gy + yi);
          }
      if (sumWi != 0) store_px(grid, dt, i * gy + j, value / sumWi);
    }
  free(in);
  return 0;
}

/* interpolate/interpolate2dStructuredFastIDW.py:29-63.  indices: (n,2) int64
 * offsets in growing-distance order, weights[n]; minnvals already decremented
 * by the wrapper (it passes minnvals-1). */
int orc_fast_idw(void* grid, int dt, const uint8_t* mask, long s0, long s1, const int64_t* indices,
                 const double* weights, long n, long minnvals) {
  size_t es = dtype_size(dt);
  void* in = malloc((size_t)s0 * s1 * es);
  if (!in) return -1;
  memcpy(in, grid, (size_t)s0 * s1 * es);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
  for (long i = 0; i < s0; i++)
    for (long j = 0; j < s1; j++) {
      if (!mask[i * s1 + j]) continue;
      double sumWi = 0, value = 0;
      long c = 0;
      for (long k = 0; k < n; k++) {
        long iii = i + indices[2 * k], jjj = j + indices[2 * k + 1];
        if (iii >= 0 && iii < s0 && jjj >= 0 && jjj < s1) {
          if (!mask[iii * s1 + jjj]) {
            double wi = weights[k];
            sumWi += wi;
            value += wi * load_px(in, dt, iii * s1 + jjj);
            if (c == minnvals) break;
            c++;
          }
        } else if (c > 0 && (iii < -1 || iii > s0 + 1) && (jjj < -1 || jjj > s1 + 1))
          break;
      }
      if (sumWi != 0) store_px(grid, dt, i * s1 + j, value / sumWi);
    }
  free(in);
  return 0;
}

/* filters/maskedFilter.py:43-72 (_calcMean): for every pixel with sel[i,j] != 0 the mean of
 * the pixels with use[ii,jj] != 0 in [i-k, min(i+k,gx)) x [j-k, min(j+k,gy)); written only
 * when at least one such pixel exists.  out may alias arr (fill_mask=True): pixels that are
 * written are never read. */
int orc_masked_mean(const void* arr, int dt, const uint8_t* sel, const uint8_t* use, long gx,
                    long gy, long k, void* out) {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
  for (long i = 0; i < gx; i++)
    for (long j = 0; j < gy; j++) {
      if (!sel[i * gy + j]) continue;
      long xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
      long ymn = j - k < 0 ? 0 : j - k, ymx = j + k > gy ? gy : j + k;
      double val = 0;
      long n = 0;
      for (long ii = xmn; ii < xmx; ii++)
        for (long jj = ymn; jj < ymx; jj++)
          if (use[ii * gy + jj]) { val += load_px(arr, dt, ii * gy + jj); n++; }
      if (n > 0) store_px(out, dt, i * gy + j, val / (double)n);
    }
  return 0;
}

static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* filters/maskedFilter.py:76-102 (_calcMedian): as orc_masked_mean with np.median of the
 * selected window values (mean of the two middle ones for an even count, formed in the array
 * dtype; NaN if any value is NaN) */
int orc_masked_median(const void* arr, int dt, const uint8_t* sel, const uint8_t* use, long gx,
                      long gy, long k, void* out) {
#pragma omp parallel num_threads(g_threads)
  {
    double* buf = (double*)malloc((size_t)(4 * k * k + 1) * sizeof(double));
#pragma omp for schedule(dynamic, 4)
    for (long i = 0; i < gx; i++)
      for (long j = 0; j < gy; j++) {
        if (!sel[i * gy + j]) continue;
        long xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
        long ymn = j - k < 0 ? 0 : j - k, ymx = j + k > gy ? gy : j + k;
        long n = 0;
        int has_nan = 0;
        for (long ii = xmn; ii < xmx; ii++)
          for (long jj = ymn; jj < ymx; jj++)
            if (use[ii * gy + jj]) {
              double v = load_px(arr, dt, ii * gy + jj);
              has_nan |= v != v;
              buf[n++] = v;
            }
        if (n == 0) continue;
        double med;
        if (has_nan) {
          med = NAN;
        } else {
          qsort(buf, (size_t)n, sizeof(double), cmp_double);
          double a = buf[(n - 1) / 2], b = buf[n / 2];
          if (dt == ORC_F32) med = (double)(((float)a + (float)b) * 0.5f);
          else med = (a + b) * 0.5;
        }
        store_px(out, dt, i * gy + j, med);
      }
    free(buf);
  }
  return 0;
}

/* filters/nan_maximum_filter.py:17-37: np.nanmax over [i-k, min(i+k,gx)) x [j-k, min(j+k,gy));
 * an all-NaN window gives NaN */
int orc_nan_max(const void* arr, int dt, long gx, long gy, long k, void* out) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long i = 0; i < gx; i++)
    for (long j = 0; j < gy; j++) {
      long xmn = i - k < 0 ? 0 : i - k, xmx = i + k > gx ? gx : i + k;
      long ymn = j - k < 0 ? 0 : j - k, ymx = j + k > gy ? gy : j + k;
      double m = NAN;
      for (long ii = xmn; ii < xmx; ii++)
        for (long jj = ymn; jj < ymx; jj++) {
          double v = load_px(arr, dt, ii * gy + jj);
          if (v == v && !(m >= v)) m = v;
        }
      store_px(out, dt, i * gy + j, m);
    }
  return 0;
}

/* scipy.ndimage.median_filter(img, size=3) (default mode='reflect' = edge pixel repeated): the
 * middle of the 9 window values; selection only, so the value is exact in any dtype */
static double median3x3(const void* img, int dt, long h, long w, long y, long x) {
  double v[9];
  int n = 0;
  for (long dy = -1; dy <= 1; dy++)
    for (long dx = -1; dx <= 1; dx++)
      v[n++] = load_px(img, dt, resolve_idx(y + dy, h, ORC_REFLECT) * w +
                                    resolve_idx(x + dx, w, ORC_REFLECT));
  for (int i = 1; i < 9; i++) { /* insertion sort */
    double t = v[i];
    int j = i - 1;
    while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; j--; }
    v[j + 1] = t;
  }
  return v[4];
}

/* scipy.ndimage.median_filter(img, size=n): the element of rank n*n/2 of the n x n window at
 * offsets -n/2 .. n-1-n/2 (scipy's origin rule for even sizes), edge pixels repeated */
static double median_nxn(const void* img, int dt, long h, long w, long y, long x, int n, double* v) {
  int c = 0;
  const long lo = n / 2;
  for (long dy = -lo; dy < n - lo; dy++)
    for (long dx = -lo; dx < n - lo; dx++)
      v[c++] = load_px(img, dt, resolve_idx(y + dy, h, ORC_REFLECT) * w +
                                    resolve_idx(x + dx, w, ORC_REFLECT));
  for (int i = 1; i < c; i++) { /* insertion sort */
    double t = v[i];
    int j = i - 1;
    while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; j--; }
    v[j + 1] = t;
  }
  return v[c / 2];
}

/* filters/medianThreshold.py:7-30: blur = float64(median_filter(img, size)); indices =
 * |(img - blur) / blur| > threshold ('<' when cond_less); out = indices ? blur : img.
 * IEEE semantics as numpy under errstate(ignore): blur == 0 gives inf (replaced for '>') or
 * NaN (0/0: comparison false, kept).  indices may be NULL. */
int orc_median_threshold_size(const void* img, int dt, long h, long w, int size, double threshold,
                              int cond_less, void* out, uint8_t* indices) {
  if (dt != ORC_F32 && dt != ORC_F64) return -2;
  if (size < 1 || size > 64) return -3;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long y = 0; y < h; y++) {
    double v[64 * 64];
    for (long x = 0; x < w; x++) {
      double a = load_px(img, dt, y * w + x);
      double blur = size == 3 ? median3x3(img, dt, h, w, y, x) : median_nxn(img, dt, h, w, y, x, size, v);
      double rel = fabs((a - blur) / blur);
      int hit = cond_less ? rel < threshold : rel > threshold;
      if (indices) indices[y * w + x] = (uint8_t)hit;
      store_px(out, dt, y * w + x, hit ? blur : a);
    }
  }
  return 0;
}
int orc_median_threshold(const void* img, int dt, long h, long w, double threshold, int cond_less,
                         void* out, uint8_t* indices) {
  return orc_median_threshold_size(img, dt, h, w, 3, threshold, cond_less, out, indices);
}

/* camera/CameraCalibration.py:416-437 (stages 2-4 of correct()): image -= bg (:505);
 * image[ff != 0] /= ff[ff != 0] (:527-528); image = np.nan_to_num(image) (:566);
 * medianThreshold(image, threshold, copy=False) (:567) when threshold > 0 (:430).
 * bg / ff may be NULL (stage skipped); arithmetic in the image dtype. */
int orc_calib_prefilter(const void* img, int dt, const void* bg, const void* ff, long h, long w,
                        double threshold, void* out) {
  if (dt != ORC_F32 && dt != ORC_F64) return -2;
  void* tmp = malloc((size_t)h * w * (dt == ORC_F32 ? 4 : 8));
  if (!tmp) return -3;
  const double big = dt == ORC_F32 ? (double)FLT_MAX : DBL_MAX;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long i = 0; i < h * w; i++) {
    double v = load_px(img, dt, i);
    if (bg) {
      v -= load_px(bg, dt, i);
      if (dt == ORC_F32) v = (double)(float)v;
    }
    if (ff) {
      double d = load_px(ff, dt, i);
      if (d != 0) {
        v /= d;
        if (dt == ORC_F32) v = (double)(float)v;
      }
    }
    if (threshold > 0) { /* np.nan_to_num is part of _correctArtefacts, skipped with it */
      if (v != v) v = 0.0;
      else if (v > big) v = big;
      else if (v < -big) v = -big;
    }
    store_px(tmp, dt, i, v);
  }
  int rc = 0;
  if (threshold > 0) rc = orc_median_threshold(tmp, dt, h, w, threshold, 0, out, NULL);
  else memcpy(out, tmp, (size_t)h * w * (dt == ORC_F32 ? 4 : 8));
  free(tmp);
  return rc;
}

/* render/closestDirectDistance.py:17-41 (_calc): distance to the closest non-zero pixel of
 * `arr` (uint8 / bool) inside the +-ksize window (centre excluded), 2*ksize when there is none,
 * 0 on the non-zero pixels; out is uint16 (numba stores the float by truncation) or float64 */
int orc_closest_distance(const uint8_t* arr, long s0, long s1, long ksize, void* out, int out_dt) {
  if (out_dt != ORC_U16 && out_dt != ORC_F64) return -2;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
  for (long i = 0; i < s0; i++)
    for (long j = 0; j < s1; j++) {
      double md = 0.0;
      if (!arr[i * s1 + j]) {
        md = 2.0 * (double)ksize;
        for (long ii = -ksize; ii <= ksize; ii++)
          for (long jj = -ksize; jj <= ksize; jj++) {
            if (ii == 0 && jj == 0) continue;
            long xi = i + ii, yi = j + jj;
            if (xi >= 0 && xi < s0 && yi >= 0 && yi < s1 && arr[xi * s1 + yi]) {
              double d = sqrt((double)(ii * ii + jj * jj));
              if (d < md) md = d;
            }
          }
      }
      if (out_dt == ORC_U16) ((uint16_t*)out)[i * s1 + j] = (uint16_t)md; /* truncation */
      else ((double*)out)[i * s1 + j] = md;
    }
  return 0;
}

/* uncertainty/positionToIntensityUncertainty.py:7-49 (_calc_constPSF / _calc_variPSF) with
 * equations/numbaGaussian2d.py:10-23 as it is CALLED there: numbaGaussian2d(psf, sx, sy) binds
 * its (sy, sx) parameters in that order, i.e. the first value scales the ROW axis.
 *   sint[i,j] = sqrt( sum_{ii,jj} psf[ii,jj] * (image[i-ii+c, j-jj+c] - image[i,j])^2 ),  c = a/2
 * for i, j in [ksize, n-ksize), NaN centres skipped, everything else stays 0.  a = 2*ksize+1.
 * sx / sy: per-pixel float64 maps (vari != 0) or a single value each (vari == 0). */
int orc_pos_to_intensity_unc(const void* image, int dt, long s0, long s1, const double* sx,
                             const double* sy, int vari, long ksize, double* sint) {
  const long a = 2 * ksize + 1, c = a / 2;
  memset(sint, 0, (size_t)s0 * s1 * sizeof(double));
#pragma omp parallel num_threads(g_threads)
  {
    double* psf = (double*)malloc((size_t)a * a * sizeof(double));
#pragma omp for schedule(dynamic, 4)
    for (long i = ksize; i < s0 - ksize; i++)
      for (long j = ksize; j < s1 - ksize; j++) {
        double cpx = load_px(image, dt, i * s1 + j);
        if (cpx != cpx) continue;
        double v0 = vari ? sx[i * s1 + j] : sx[0], v1 = vari ? sy[i * s1 + j] : sy[0];
        double ss_row = 2 * v0 * v0, ss_col = 2 * v1 * v1, tot = 0.0;
        for (long ii = 0; ii < a; ii++)
          for (long jj = 0; jj < a; jj++) {
            double e = exp(-((double)((ii - c) * (ii - c)) / ss_row +
                             (double)((jj - c) * (jj - c)) / ss_col));
            psf[ii * a + jj] = e;
            tot += e;
          }
        double sdev = 0.0;
        for (long ii = 0; ii < a; ii++)
          for (long jj = 0; jj < a; jj++) {
            double d = load_px(image, dt, (i - ii + c) * s1 + (j - jj + c)) - cpx;
            sdev += (psf[ii * a + jj] / tot) * (d * d);
          }
        sint[i * s1 + j] = sqrt(sdev);
      }
    free(psf);
  }
  return 0;
}

/* interpolate/interpolate2dUnstructuredIDW.py:7-38.  n scattered points (x = ROW coordinate,
 * y = column, as the reference indexes grid[i, j] with i against x), every grid pixel gets
 * sum(w v) / sum(w), w = 1 / (dx^2 + dy^2)^(power/2), summed in point order in float64; a pixel
 * that IS a point takes the value of the first such point. */
int orc_unstructured_idw(void* grid, int dt, long gx, long gy, const double* x, const double* y,
                         const double* v, long n, double power) {
  if (n < 1) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long i = 0; i < gx; i++)
    for (long j = 0; j < gy; j++) {
      int over = 0;
      double sumWi = 0.0, value = 0.0;
      for (long k = 0; k < n; k++) {
        if (x[k] == (double)i && y[k] == (double)j) {
          store_px(grid, dt, i * gy + j, v[k]);
          over = 1;
          break;
        }
        double dx = x[k] - (double)i, dy = y[k] - (double)j;
        double wi = 1.0 / pow(dx * dx + dy * dy, 0.5 * power);
        sumWi += wi;
        value += wi * v[k];
      }
      if (!over) store_px(grid, dt, i * gy + j, value / sumWi);
    }
  return 0;
}

/* interpolate/interpolateCircular2dStructuredIDW.py:7-69, as written: the column count is
 * taken from shape[0] too (gy = grid.shape[0], :16-17), so columns >= shape[0] are neither
 * filled nor read (s1 < s0 indexes out of bounds in the reference: -2 here); the window is
 * [i-k, min(i+k, gx)) x [j-k, min(j+k, gx)) - the upper end EXCLUSIVE, unlike the plain IDW;
 * dist is the SQUARE of the squared polar distance (:56).  dist == 0 (two pixels at one polar
 * position under the weights given) divides by zero in the reference and is not covered. */
int orc_circular_idw(void* grid, int dt, const uint8_t* mask, long s0, long s1, long kernel,
                     double power, double fr, double fphi, double cx, double cy) {
  if (s1 < s0) return -2;
  const long gx = s0, gy = s0;
  size_t es = dtype_size(dt);
  void* in = malloc((size_t)s0 * s1 * es);
  if (!in) return -1;
  memcpy(in, grid, (size_t)s0 * s1 * es); /* masked px are never read: copy == in-place */
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
  for (long i = 0; i < gx; i++)
    for (long j = 0; j < gy; j++) {
      if (!mask[i * s1 + j]) continue;
      long xmn = i - kernel < 0 ? 0 : i - kernel, xmx = i + kernel > gx ? gx : i + kernel;
      long ymn = j - kernel < 0 ? 0 : j - kernel, ymx = j + kernel > gx ? gy : j + kernel;
      double sumWi = 0.0, value = 0.0;
      double di = (double)i - cx, dj = (double)j - cy;
      double R = pow(di * di + dj * dj, 0.5), PHI = atan2(dj, di);
      for (long xi = xmn; xi < xmx; xi++)
        for (long yi = ymn; yi < ymx; yi++)
          if ((xi != i || yi != j) && !mask[xi * s1 + yi]) {
            double ni = (double)xi - cx, nj = (double)yi - cy;
            double nR = pow(ni * ni + nj * nj, 0.5);
            double dr = R - nR, midR = 0.5 * (R + nR);
            double nphi = atan2(nj, ni);
            double a = fabs(PHI - nphi), b = 2 * M_PI - a;
            double dphi = (b < a ? b : a) * midR;
            double q = (fr * dr) * (fr * dr) + (fphi * dphi) * (fphi * dphi);
            double dist = q * q;
            double wi = 1.0 / pow(dist, 0.5 * power);
            sumWi += wi;
            value += wi * load_px(in, dt, xi * s1 + yi);
          }
      if (sumWi != 0) store_px(grid, dt, i * s1 + j, value / sumWi);
    }
  free(in);
  return 0;
}

/* interpolate/interpolate2dStructuredCrossAvg.py:7-115, as written.  Every masked pixel looks
 * down / up / left / right along its column and row for the nearest unmasked pixel and takes
 * the local average (_localAvg :21-44: unmasked pixels within +-kernel) there; the values are
 * blended with weights 1 / distance^(power/2), normalised.  Quirks kept:
 *  - "look up" (:73-83) stores vals[1] / dist[1] but raises valid[2], so the upward value is
 *    never used, and slot 2 (the leftward value) counts as valid whenever EITHER the upward or
 *    the leftward search succeeded; when only the upward one did, slot 2 still holds what the
 *    LAST earlier pixel (raster order) with a leftward hit left there.  Before any such pixel
 *    the slot is np.empty garbage in the reference: dropped here.
 *  - "look right" runs only if i < gy - 1 (the ROW index against the column count, :96).
 *  - dist is uint16, weights float32 (normalised in float32), vals has the grid's dtype.
 *  - _localAvg clamps its window to gx / gy, not gx-1 / gy-1, and reads that index: out of
 *    bounds (UB under numba); clamped to the array here like orc_idw.
 * Sequential (the stale slot makes the raster order part of the result). */
static double cross_local_avg(const void* in, int dt, const uint8_t* mask, long i, long j, long gx,
                              long gy, long kernel) {
  long xmn = i - kernel < 0 ? 0 : i - kernel, xmx = i + kernel > gx - 1 ? gx - 1 : i + kernel;
  long ymn = j - kernel < 0 ? 0 : j - kernel, ymx = j + kernel > gy - 1 ? gy - 1 : j + kernel;
  double val = 0;
  long n = 0;
  for (long xi = xmn; xi <= xmx; xi++)
    for (long yi = ymn; yi <= ymx; yi++)
      if (!mask[xi * gy + yi]) {
        val += load_px(in, dt, xi * gy + yi);
        n++;
      }
  return val / (double)n;
}

/* interpolate/interpolate2dStructuredPointSpreadIDW.py:31-63 (_createBorder) and :65-141 (_calc),
 * as written.  _createBorder: a row-major scan, then a column-major scan, each carrying its
 * "previous value" ACROSS the ends of rows / columns; a masked pixel that follows an unmasked one
 * becomes border, an unmasked pixel that follows a masked one marks ITS PREDECESSOR IN THE SAME
 * ROW / COLUMN - index j - 1 (i - 1), which for j = 0 (i = 0) is numpy's -1: the last pixel of
 * the row (column).  Flags are only ever set here.  _calc: sweeps over the border pixels in
 * raster order, each filled from the unmasked pixels of its window [i-k, i+k) x [j-k, j+k) - the
 * mask as the sweep has left it so far -, then unmasked itself; a sweep ends with a new border
 * pass; the loop ends when a border pass finds no transition or after maxIter sweeps.  The
 * column limit is clamped to gy only when it exceeds the ROW count gx (:89-90); where it would
 * leave the array (gy < gx) it is clamped to the array here.  mask is modified. */
static int ps_create_border(const uint8_t* mask, uint8_t* border, long gx, long gy) {
  int any = 0;
  uint8_t last = mask[0];
  for (long i = 0; i < gx; i++)
    for (long j = 0; j < gy; j++) {
      uint8_t val = mask[i * gy + j];
      if (val != last) {
        if (val) border[i * gy + j] = 1;
        else border[i * gy + (j > 0 ? j - 1 : gy - 1)] = 1;
        any = 1;
      }
      last = val;
    }
  last = mask[0];
  for (long j = 0; j < gy; j++)
    for (long i = 0; i < gx; i++) {
      uint8_t val = mask[i * gy + j];
      if (val != last) {
        if (val) border[i * gy + j] = 1;
        else border[(i > 0 ? i - 1 : gx - 1) * gy + j] = 1;
        any = 1;
      }
      last = val;
    }
  return any;
}

int orc_point_spread_idw(void* grid, int dt, uint8_t* mask, long gx, long gy, long kernel,
                         double power, long max_iter) {
  uint8_t* border = (uint8_t*)calloc((size_t)gx * gy, 1);
  if (!border) return -1;
  int any = ps_create_border(mask, border, gx, gy);
  long n = 0;
  while (n < max_iter && any) {
    for (long i = 0; i < gx; i++)
      for (long j = 0; j < gy; j++) {
        if (!border[i * gy + j]) continue;
        long xmn = i - kernel < 0 ? 0 : i - kernel, xmx = i + kernel > gx ? gx : i + kernel;
        long ymn = j - kernel < 0 ? 0 : j - kernel, ymx = j + kernel;
        if (ymx > gx) ymx = gy;
        if (ymx > gy) ymx = gy; /* (out of bounds in the source) */
        double sumWi = 0.0, value = 0.0;
        for (long xi = xmn; xi < xmx; xi++)
          for (long yi = ymn; yi < ymx; yi++)
            if (!(xi == i && yi == j) && !mask[xi * gy + yi]) {
              double d2 = (double)((xi - i) * (xi - i) + (yi - j) * (yi - j));
              double wi = 1.0 / pow(d2, 0.5 * power);
              sumWi += wi;
              value += wi * load_px(grid, dt, xi * gy + yi);
            }
        if (sumWi != 0.0) {
          store_px(grid, dt, i * gy + j, value / sumWi);
          border[i * gy + j] = 0;
          mask[i * gy + j] = 0;
        }
      }
    any = ps_create_border(mask, border, gx, gy);
    n++;
  }
  free(border);
  return 0;
}

int orc_cross_avg(void* grid, int dt, const uint8_t* mask, long gx, long gy, long kernel,
                  double power) {
  size_t es = dtype_size(dt);
  void* in = malloc((size_t)gx * gy * es);
  if (!in) return -1;
  memcpy(in, grid, (size_t)gx * gy * es); /* only unmasked px are read: copy == in-place */
  double vals[4] = {0, 0, 0, 0};
  uint16_t dist[4] = {0, 0, 0, 0};
  int slot2_set = 0;
  for (long i = 0; i < gx; i++)
    for (long j = 0; j < gy; j++) {
      if (!mask[i * gy + j]) continue;
      int valid[4] = {0, 0, 0, 0};
      for (long i0 = i - 1; i0 >= 0; i0--) /* look down (:60-71) */
        if (!mask[i0 * gy + j]) {
          vals[0] = cross_local_avg(in, dt, mask, i0, j, gx, gy, kernel);
          dist[0] = (uint16_t)(i - i0);
          valid[0] = 1;
          break;
        }
      for (long i0 = i + 1; i0 < gx; i0++) /* look up (:72-83): sets valid[2] */
        if (!mask[i0 * gy + j]) {
          vals[1] = cross_local_avg(in, dt, mask, i0, j, gx, gy, kernel);
          dist[1] = (uint16_t)(i0 - i);
          valid[2] = 1;
          break;
        }
      for (long j0 = j - 1; j0 >= 0; j0--) /* look left (:84-95) */
        if (!mask[i * gy + j0]) {
          vals[2] = cross_local_avg(in, dt, mask, i, j0, gx, gy, kernel);
          dist[2] = (uint16_t)(j - j0);
          valid[2] = 1;
          slot2_set = 1;
          break;
        }
      if (i < gy - 1) /* look right (:96-107) */
        for (long j0 = j + 1; j0 < gy; j0++)
          if (!mask[i * gy + j0]) {
            vals[3] = cross_local_avg(in, dt, mask, i, j0, gx, gy, kernel);
            dist[3] = (uint16_t)(j0 - j);
            valid[3] = 1;
            break;
          }
      if (valid[2] && !slot2_set) valid[2] = 0;
      /* vals has the grid's dtype */
      if (dt == ORC_F32)
        for (int s = 0; s < 4; s++) vals[s] = (double)(float)vals[s];
      float w[4], wsum = 0.f;
      for (int s = 0; s < 4; s++)
        if (valid[s]) {
          w[s] = (float)(1.0 / pow((double)dist[s], 0.5 * power));
          wsum += w[s];
        }
      if (dt == ORC_F32) {
        float acc = 0.f;
        for (int s = 0; s < 4; s++)
          if (valid[s]) acc += (float)vals[s] * (w[s] / wsum);
        store_px(grid, dt, i * gy + j, acc);
      } else {
        double acc = 0.0;
        for (int s = 0; s < 4; s++)
          if (valid[s]) acc += vals[s] * (double)(w[s] / wsum);
        store_px(grid, dt, i * gy + j, acc);
      }
    }
  free(in);
  return 0;
}

/* ------------------------------------------------------------------ */
/* filters/fastFilter.py, filters/fastMean.py                          */
/* ------------------------------------------------------------------ */

/* filters/fastFilter.py:52-122 (_iter + _calcMedian / _calcNanMedian / _calcMean / _calcNanMean):
 * for i in range(0, gx, every), j in range(0, gy, every) the statistic of
 * arr[max(i-k,0) : min(i+k,gx) : every, max(j-k,0) : min(j+k,gy) : every] goes to out[ii, jj].
 * fn: 0 median (NaN if the window holds a NaN, like np.median), 1 nanmedian, 2 mean,
 * 3 nanmean (NaN for an all-NaN window).  out: n0 x n1 doubles, n0 = ceil(gx / every),
 * n1 = ceil(gy / every); the wrapper applies the reference's crop to [:n0-1, :n1-1] (:36-37:
 * the loops return their LAST indices, which are then used as sizes). */
int orc_fast_filter_stat(const void* arr, int dt, long gx, long gy, long ksize, long every, int fn,
                         double* out) {
  if (every < 1 || ksize < 1) return -1;
  long n0 = (gx + every - 1) / every, n1 = (gy + every - 1) / every;
  long cap = (2 * ksize / every + 2) * (2 * ksize / every + 2);
#pragma omp parallel num_threads(g_threads)
  {
    double* buf = (double*)malloc((size_t)cap * sizeof(double));
#pragma omp for schedule(dynamic, 4)
    for (long ii = 0; ii < n0; ii++)
      for (long jj = 0; jj < n1; jj++) {
        long i = ii * every, j = jj * every;
        long xmn = i - ksize < 0 ? 0 : i - ksize, xmx = i + ksize > gx ? gx : i + ksize;
        long ymn = j - ksize < 0 ? 0 : j - ksize, ymx = j + ksize > gy ? gy : j + ksize;
        long n = 0, nn = 0;
        double sum = 0;
        for (long x = xmn; x < xmx; x += every)
          for (long y = ymn; y < ymx; y += every) {
            double v = load_px(arr, dt, x * gy + y);
            if (v != v) {
              nn++;
              continue;
            }
            buf[n++] = v;
            sum += v;
          }
        double r;
        if ((fn == 0 || fn == 2) && nn) r = NAN;
        else if (n == 0) r = NAN;
        else if (fn >= 2) r = sum / (double)n;
        else {
          qsort(buf, (size_t)n, sizeof(double), cmp_double);
          r = (n & 1) ? buf[n / 2] : (buf[n / 2 - 1] + buf[n / 2]) / 2.0;
        }
        out[ii * n1 + jj] = r;
      }
    free(buf);
  }
  return 0;
}

/* cv::resize for single-channel float32 / float64 images (OpenCV imgproc/src/resize.cpp, restated
 * from the published algorithm - cv2 is not available here: UNPINNED like the other cv2 modes).
 * interp: 1 INTER_LINEAR, 2 INTER_CUBIC (a = -0.75), 3 INTER_AREA, 4 INTER_LANCZOS4.
 *  - scale = 1 / ((double)dsize / ssize); source position of destination index d:
 *    f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s  (float)
 *  - linear: along x a position left of the first / right of the last pixel is clamped to it with
 *    fraction 0; along y the two ROWS are clipped instead;  cubic / Lanczos4: tap indices clipped
 *  - coefficients are float32 (interpolateCubic / interpolateLanczos4); the work type is float32 for
 *    float32 images and float64 for float64 ones; horizontal pass first (its rows rounded to the
 *    work type), products summed left to right
 *  - area, integer scale: the block's pixels summed in groups of four ((a+b+c+d) added to the
 *    running sum - CV_ENABLE_UNROLLED), times (float)(1/area); otherwise the decimation tables of
 *    computeResizeAreaTab with float32 weights, row sums first.  Upscaling with INTER_AREA (OpenCV
 *    switches to a bilinear variant there) is not covered: -2. */
static void cv_cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

typedef struct { int si, di; float alpha; } area_tab_t;

static int area_tab(int ssize, int dsize, double scale, area_tab_t* tab) {
  int k = 0;
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    if (sx1 - fsx1 > 1e-3) {
      tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell);
    }
    for (int sx = sx1; sx < sx2; sx++) {
      tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell);
    }
    if (fsx2 - sx2 > 1e-3) {
      double a = fsx2 - sx2 < 1.0 ? fsx2 - sx2 : 1.0;
      a = a < cell ? a : cell;
      tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(a / cell);
    }
  }
  return k;
}

#define ORC_RESIZE_BODY(T, WT)                                                                      \
  const T* S = (const T*)src;                                                                       \
  T* D = (T*)dst;                                                                                   \
  if (interp == 3) {                                                                                \
    if (area_fast) {                                                                                \
      const float fscale = 1.f / (float)(isx * isy);                                               \
      const long w1 = sw / isx;                                                                     \
      for (long dy = 0; dy < dh; dy++) {                                                            \
        long sy0 = dy * isy, w = sy0 + isy <= sh ? w1 : 0;                                          \
        if (w > dw) w = dw;                                                                         \
        for (long dx = 0; dx < dw; dx++) {                                                          \
          long sx0 = dx * isx;                                                                      \
          if (sy0 >= sh) { D[dy * dw + dx] = 0; continue; }                                         \
          if (dx < w) {                                                                             \
            WT sum = 0;                                                                             \
            long k = 0, area = (long)isx * isy;                                                     \
            for (; k <= area - 4; k += 4) {                                                         \
              WT a = S[(sy0 + k / isx) * sw + sx0 + k % isx];                                       \
              WT b = S[(sy0 + (k + 1) / isx) * sw + sx0 + (k + 1) % isx];                           \
              WT c = S[(sy0 + (k + 2) / isx) * sw + sx0 + (k + 2) % isx];                           \
              WT d = S[(sy0 + (k + 3) / isx) * sw + sx0 + (k + 3) % isx];                           \
              sum += a + b + c + d;                                                                 \
            }                                                                                       \
            for (; k < area; k++) sum += S[(sy0 + k / isx) * sw + sx0 + k % isx];                   \
            D[dy * dw + dx] = (T)(sum * fscale);                                                    \
          } else {                                                                                  \
            WT sum = 0;                                                                             \
            long count = 0;                                                                         \
            if (sx0 >= sw) { D[dy * dw + dx] = 0; continue; }                                       \
            for (long sy = 0; sy < isy && sy0 + sy < sh; sy++)                                      \
              for (long sx = 0; sx < isx && sx0 + sx < sw; sx++) {                                  \
                sum += S[(sy0 + sy) * sw + sx0 + sx];                                               \
                count++;                                                                            \
              }                                                                                     \
            D[dy * dw + dx] = (T)((float)sum / count);                                              \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
    } else {                                                                                        \
      area_tab_t* xt = (area_tab_t*)malloc(sizeof(area_tab_t) * (size_t)(sw * 2 + 2 * dw + 4));    \
      area_tab_t* yt = (area_tab_t*)malloc(sizeof(area_tab_t) * (size_t)(sh * 2 + 2 * dh + 4));    \
      int nx = area_tab((int)sw, (int)dw, scale_x, xt), ny = area_tab((int)sh, (int)dh, scale_y, yt); \
      WT* buf = (WT*)malloc(sizeof(WT) * (size_t)dw);                                               \
      WT* sum = (WT*)calloc((size_t)dw, sizeof(WT));                                                \
      int prev_dy = yt[0].di;                                                                       \
      for (int j = 0; j < ny; j++) {                                                                \
        WT beta = yt[j].alpha;                                                                      \
        int dy = yt[j].di, sy = yt[j].si;                                                           \
        for (long dx = 0; dx < dw; dx++) buf[dx] = 0;                                               \
        for (int k = 0; k < nx; k++) buf[xt[k].di] += S[(long)sy * sw + xt[k].si] * (WT)xt[k].alpha; \
        if (dy != prev_dy) {                                                                        \
          for (long dx = 0; dx < dw; dx++) {                                                        \
            D[(long)prev_dy * dw + dx] = (T)sum[dx];                                                \
            sum[dx] = beta * buf[dx];                                                               \
          }                                                                                         \
          prev_dy = dy;                                                                             \
        } else {                                                                                    \
          for (long dx = 0; dx < dw; dx++) sum[dx] += beta * buf[dx];                               \
        }                                                                                           \
      }                                                                                             \
      for (long dx = 0; dx < dw; dx++) D[(long)prev_dy * dw + dx] = (T)sum[dx];                     \
      free(xt); free(yt); free(buf); free(sum);                                                     \
    }                                                                                               \
  } else {                                                                                          \
    WT* tmp = (WT*)malloc(sizeof(WT) * (size_t)sh * dw);                                            \
    for (long y = 0; y < sh; y++)                                                                   \
      for (long dx = 0; dx < dw; dx++) {                                                            \
        const float* a = alpha + dx * ks;                                                           \
        long sx = xofs[dx];                                                                         \
        WT v;                                                                                       \
        if (ks == 2) {                                                                              \
          if (dx >= xmax) v = (WT)S[y * sw + sx] * (WT)1;                                           \
          else v = S[y * sw + sx] * (WT)a[0] + S[y * sw + sx + 1] * (WT)a[1];                       \
        } else {                                                                                    \
          v = 0;                                                                                    \
          for (int j = 0; j < ks; j++) {                                                            \
            long sxj = sx - (ks / 2 - 1) + j;                                                       \
            sxj = sxj < 0 ? 0 : (sxj >= sw ? sw - 1 : sxj);                                         \
            v += S[y * sw + sxj] * (WT)a[j];                                                        \
          }                                                                                         \
        }                                                                                           \
        tmp[y * dw + dx] = v;                                                                       \
      }                                                                                             \
    for (long dy = 0; dy < dh; dy++)                                                                \
      for (long dx = 0; dx < dw; dx++) {                                                            \
        const float* b = beta + dy * ks;                                                            \
        WT v = 0;                                                                                   \
        for (int k = 0; k < ks; k++) {                                                              \
          long sy = yofs[dy] - (ks / 2) + 1 + k;                                                    \
          sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;                                               \
          WT t = tmp[sy * dw + dx] * (WT)b[k];                                                      \
          v = k == 0 ? t : v + t;                                                                   \
        }                                                                                           \
        D[dy * dw + dx] = (T)v;                                                                     \
      }                                                                                             \
    free(tmp);                                                                                      \
  }

int orc_resize(const void* src, int dt, long sh, long sw, void* dst, long dh, long dw, int interp) {
  if (dt != ORC_F32 && dt != ORC_F64) return -1;
  if (sh < 1 || sw < 1 || dh < 1 || dw < 1) return -1;
  const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
  /* resize.cpp: INTER_LINEAR at an exact 2 x 2 reduction is computed as INTER_AREA */
  if (interp == 1 && sw == 2 * dw && sh == 2 * dh) interp = 3;
  if (interp < 1 || interp > 4) return -3;
  int ks = interp == 1 ? 2 : (interp == 2 ? 4 : 8);
  int isx = (int)nearbyint(scale_x), isy = (int)nearbyint(scale_y);
  int area_fast = 0;
  long* xofs = NULL; long* yofs = NULL;
  float* alpha = NULL; float* beta = NULL;
  long xmax = dw;
  if (interp == 3) {
    if (!(scale_x >= 1 && scale_y >= 1)) return -2;
    area_fast = fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON;
  } else if (interp == 1 || interp == 2 || interp == 4) {
    xofs = (long*)malloc(sizeof(long) * (size_t)dw);
    yofs = (long*)malloc(sizeof(long) * (size_t)dh);
    alpha = (float*)malloc(sizeof(float) * (size_t)dw * ks);
    beta = (float*)malloc(sizeof(float) * (size_t)dh * ks);
    for (long dx = 0; dx < dw; dx++) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      long sx = (long)floorf(fx);
      fx -= sx;
      if (sx < ks / 2 - 1 && sx < 0 && interp == 1) { fx = 0; sx = 0; }
      if (sx + ks / 2 >= sw) {
        xmax = xmax < dx ? xmax : dx;
        if (sx >= sw - 1 && interp == 1) { fx = 0; sx = sw - 1; }
      }
      xofs[dx] = sx;
      float* c = alpha + dx * ks;
      if (interp == 2) cv_cubic_coeffs(fx, c);
      else if (interp == 4) orc_lanczos4_weights(fx, c);
      else { c[0] = 1.f - fx; c[1] = fx; }
    }
    for (long dy = 0; dy < dh; dy++) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      long sy = (long)floorf(fy);
      fy -= sy;
      yofs[dy] = sy;
      float* c = beta + dy * ks;
      if (interp == 2) cv_cubic_coeffs(fy, c);
      else if (interp == 4) orc_lanczos4_weights(fy, c);
      else { c[0] = 1.f - fy; c[1] = fy; }
    }
  } else {
    return -1;
  }
  if (dt == ORC_F32) { ORC_RESIZE_BODY(float, float) } else { ORC_RESIZE_BODY(double, double) }
  free(xofs); free(yofs); free(alpha); free(beta);
  return 0;
}

/* headline chain for the CPU baseline: map-based undistort then K x K filter */
int orc_remap_conv2d(const void* src, int src_dt, long h, long w, const float* mapx,
                     const float* mapy, const double* kern, long kh, long kw, void* tmp, void* dst,
                     int dst_dt, int interp, int border, double cval, int cborder_x,
                     int cborder_y) {
  int rc = orc_remap(src, src_dt, h, w, w, mapx, mapy, w, tmp, dst_dt, h, w, w, interp, border, cval);
  if (rc) return rc;
  return orc_conv2d(tmp, dst_dt, h, w, w, kern, kh, kw, NULL, 0, dst, w, cborder_x, cborder_y, 0.0);
}
