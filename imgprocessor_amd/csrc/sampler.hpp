// sampler.hpp — device-side source sampling for the remap family.
//
// A sample is:  coordinate -> (first tap index, per-axis weights) -> NT x NT
// taps fetched through a raw buffer descriptor (hardware range check on the
// frame, 32-bit byte offsets) -> separable weighted sum with fma chains.
// Interior footprints use one wide buffer load per tap row (dwordx2 for
// bilinear f32, dwordx4 for bicubic f32 ...); footprints that touch the image
// edge resolve every tap through the border mode.
//
// Semantics restated from (reference = /root/reference):
//   camera/LensDistortion.py:323-326   cv2.remap INTER_LINEAR, BORDER_CONSTANT
//   camera/PerspectiveCorrection.py:377-378, 401-405  INTER_CUBIC / INTER_LANCZOS4
// with cv2's interpolation kernels as published in OpenCV imgwarp.cpp
// (interpolateCubic, interpolateLanczos4, INTER_BITS = 5) and the exact-
// coordinate forms that scipy.ndimage.map_coordinates(order=1) and
// skimage.transform.warp(order=1/3) implement.
#pragma once

#include "common.hpp"

namespace ipa {

enum : int { kNearest = 0, kLinear = 1, kCubic = 2, kLanczos4 = 4 };

template <int INTERP> struct ntaps { static constexpr int value = 4; };
template <> struct ntaps<kNearest> { static constexpr int value = 1; };
template <> struct ntaps<kLinear> { static constexpr int value = 2; };
template <> struct ntaps<kLanczos4> { static constexpr int value = 8; };

// read-only view of one source frame
struct SrcView {
  __amdgpu_buffer_rsrc_t rsrc;
  int h, w, pitch;        // pitch in elements
  int border;             // ipa_border
  int q5;                 // 1: cv2-style 1/32-px coordinate rounding
  float cubic_a;          // Keys parameter (-0.75 cv2, -0.5 skimage)
  const float* lanczos;   // [32][8] table (device), only for kLanczos4
  // 1: neighbouring lanes sample neighbouring pixels (the lane-interleaved order of the fused
  // kernels and of remap_kernel's aligned segments): bilinear float tap rows then load as two
  // dword gathers - which cost by the cache lines the wave touches - instead of one dwordx2
  // (16 clocks whatever its addresses).  0: one lane = 4 consecutive pixels, the dwordx2 wins.
  int pair_split = 0;
  // byte offset of pixel (0, 0) from the descriptor's base (sample_u8_lanczos_lds only): the
  // descriptor starts on the dword below a frame that does not start on one
  int org = 0;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  // 0x00020000: raw buffer, DST_SEL identity / 32-bit data format for gfx9-family MUBUF
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// ------------------------------------------------------------- tap loads --
// NOTE: never __builtin_bit_cast a vector ELEMENT expression (r[1]): clang 22
// (ROCm 7.2) reads element 0 for every index.  Pass the element by value.
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ double u2d(unsigned lo, unsigned hi) {
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <typename ST, typename CT> struct TapLoad;

template <> struct TapLoad<float, float> {
  static __device__ __forceinline__ float one(const SrcView& s, int e) {
    return u2f(__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, e << 2, 0, 0));
  }
  template <int N> static __device__ __forceinline__ void row(const SrcView& s, int e, float (&v)[N]) {
    if constexpr (N == 1) {
      v[0] = one(s, e);
    } else if constexpr (N == 2) {
      // two dwords (the second through the instruction's immediate offset) instead of one
      // dwordx2: a 64-bit gather issues at 4 lanes/clk whatever its addresses are, a dword
      // gather by the cache lines it touches (tools/ta_micro.hip) - and the lane-interleaved
      // samplers make neighbouring lanes hit neighbouring pixels.  Measured +1.5 % on the
      // fused 4K kernel and the bilinear remap (profiles/r01_micro.txt).
      // (the "4" is hidden from the compiler, which otherwise fuses the pair into the dwordx2
      // again, as it silently did for most of round 2: 64 x 4K fused 5x5 1.481 -> 1.387 ms on
      // one box.  Through a second descriptor at base + 4 instead: 1.573 ms; with the sc0 scope
      // bit on one of the two: level with this form)
      if (s.pair_split) {
        int four = 4;
        asm("" : "+s"(four));
        v[0] = u2f(__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, e << 2, 0, 0));
        v[1] = u2f(__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, (e << 2) + four, 0, 0));
      } else {
        auto r = __builtin_amdgcn_raw_buffer_load_b64(s.rsrc, e << 2, 0, 0);
        v[0] = u2f(r[0]);
        v[1] = u2f(r[1]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < N; k += 4) {
        auto r = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (e + k) << 2, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) v[k + j] = u2f(r[j]);
      }
    }
  }
};

template <> struct TapLoad<double, double> {
  static __device__ __forceinline__ double one(const SrcView& s, int e) {
    auto r = __builtin_amdgcn_raw_buffer_load_b64(s.rsrc, e << 3, 0, 0);
    return u2d(r[0], r[1]);
  }
  template <int N> static __device__ __forceinline__ void row(const SrcView& s, int e, double (&v)[N]) {
    if constexpr (N == 1) {
      v[0] = one(s, e);
    } else {
#pragma unroll
      for (int k = 0; k < N; k += 2) {
        auto r = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (e + k) << 3, 0, 0);
        v[k] = u2d(r[0], r[1]);
        v[k + 1] = u2d(r[2], r[3]);
      }
    }
  }
};

template <> struct TapLoad<uint8_t, float> {
  static __device__ __forceinline__ float one(const SrcView& s, int e) {
    return (float)__builtin_amdgcn_raw_buffer_load_b8(s.rsrc, e, 0, 0);
  }
  // tap rows as one ushort / dword / dwordx2 at the byte offset of the first tap (served at
  // any alignment, tools/unaligned_probe.hip)
  template <int N> static __device__ __forceinline__ void row(const SrcView& s, int e, float (&v)[N]) {
    if constexpr (N == 1) {
      v[0] = one(s, e);
    } else if constexpr (N == 2) {
      const unsigned r = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(s.rsrc, e, 0, 0);
      v[0] = (float)(r & 0xffu);
      v[1] = (float)(r >> 8);
    } else {
#pragma unroll
      for (int k = 0; k < N; k += 4) {
        const unsigned r = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, e + k, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) v[k + j] = (float)((r >> (8 * j)) & 0xffu);
      }
    }
  }
};

// uint16 tap rows come in as ONE dword / dwordx2 / dwordx4 at the 2-byte-aligned byte offset of
// the first tap (gfx950 serves buffer loads at any 2-byte offset: tools/unaligned_probe.hip);
// per-tap ushort loads would double the gather instructions of a camera-frame remap.
template <> struct TapLoad<uint16_t, float> {
  static __device__ __forceinline__ float one(const SrcView& s, int e) {
    return (float)__builtin_amdgcn_raw_buffer_load_b16(s.rsrc, e << 1, 0, 0);
  }
  static __device__ __forceinline__ void unpack(unsigned r, float& lo, float& hi) {
    lo = (float)(r & 0xffffu);
    hi = (float)(r >> 16);
  }
  template <int N> static __device__ __forceinline__ void row(const SrcView& s, int e, float (&v)[N]) {
    if constexpr (N == 1) {
      v[0] = one(s, e);
    } else if constexpr (N == 2) {
      unpack(__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, e << 1, 0, 0), v[0], v[1]);
    } else if constexpr (N == 4) {
      auto r = __builtin_amdgcn_raw_buffer_load_b64(s.rsrc, e << 1, 0, 0);
      unpack(r[0], v[0], v[1]);
      unpack(r[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int k = 0; k < N; k += 8) {
        auto r = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (e + k) << 1, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) unpack(r[j], v[k + 2 * j], v[k + 2 * j + 1]);
      }
    }
  }
};

// --------------------------------------------------------------- weights --
__device__ __forceinline__ float ipa_floor(float x) { return floorf(x); }
__device__ __forceinline__ double ipa_floor(double x) { return floor(x); }
__device__ __forceinline__ float ipa_abs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double ipa_abs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float ipa_rint(float x) { return rintf(x); }
__device__ __forceinline__ double ipa_rint(double x) { return rint(x); }

// OpenCV interpolateCubic generalised over A:
//   w0 = ((A t1 - 5 A) t1 + 8 A) t1 - 4 A,  w1 = ((A + 2) t - (A + 3)) t t + 1,  w2 = the same in u = 1 - t,
//   w3 = 1 - w0 - w1 - w2,  t1 = t + 1.
// Written with EXPLICIT fused multiply-adds, in the grouping the compiler's own contraction gave the scalar float
// form through round 5 (so those bits stay): round 6 evaluates the weights of TWO pixels in the halves of packed
// instructions on the tile kernel (CT = v2f) and a second contraction decision must not be able to differ from
// the first.  ipa_fma has float, double and v2f overloads; every other operation is a lone add / sub / mul.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f ipa_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
template <typename CT> __device__ __forceinline__ CT ipa_splat(float v) { return (CT)v; }
template <> __device__ __forceinline__ v2f ipa_splat<v2f>(float v) { return v2f{v, v}; }
template <typename CT> __device__ __forceinline__ void cubic_weights(CT t, CT A, CT (&w)[4]) {
  const CT one = ipa_splat<CT>(1.f);
  const CT t1 = t + one, u = one - t;
  const CT a2 = A + ipa_splat<CT>(2.f), a3n = -(A + ipa_splat<CT>(3.f));
  CT p = ipa_fma(A, t1, -(ipa_splat<CT>(5.f) * A));
  p = t1 * p;
  p = ipa_fma(A, ipa_splat<CT>(8.f), p);
  w[0] = ipa_fma(t1, p, -(ipa_splat<CT>(4.f) * A));
  CT x = ipa_fma(a2, t, a3n);
  x = t * x;
  w[1] = ipa_fma(t, x, one);
  CT y = ipa_fma(a2, u, a3n);
  y = u * y;
  w[2] = ipa_fma(u, y, one);
  w[3] = ((one - w[0]) - w[1]) - w[2];
}

// coordinate (float: map / undistort; double: homography) -> first tap + weights
template <int INTERP, typename CT, typename C>
__device__ __forceinline__ void axis_split(const SrcView& s, C c, int& i0,
                                           CT (&w)[ntaps<INTERP>::value]) {
  int ip;
  CT t;
  int k = 0;
  if (INTERP == kLanczos4 || s.q5) {
    int qi = (int)ipa_rint(c * (C)32);  // cvRound(coordinate * INTER_TAB_SIZE)
    ip = qi >> 5;
    k = qi & 31;
    t = (CT)k * (CT)0.03125;
  } else {
    C fl = ipa_floor(c);
    ip = (int)fl;
    // fraction in the wider of (coordinate, compute) types: c - floor(c) is not
    // exact in float for c in (-1, 0)
    if constexpr (sizeof(CT) > sizeof(C)) t = (CT)c - (CT)fl;
    else t = (CT)(c - fl);
  }
  if constexpr (INTERP == kNearest) {
    i0 = (int)ipa_rint(c);
    w[0] = (CT)1;
  } else if constexpr (INTERP == kLinear) {
    i0 = ip;
    w[0] = (CT)1 - t;
    w[1] = t;
  } else if constexpr (INTERP == kCubic) {
    i0 = ip - 1;
    cubic_weights<CT>(t, (CT)s.cubic_a, w);
  } else {
    i0 = ip - 3;
    const float4* row = reinterpret_cast<const float4*>(s.lanczos + k * 8);
    float4 a = row[0], b = row[1];
    w[0] = (CT)a.x; w[1] = (CT)a.y; w[2] = (CT)a.z; w[3] = (CT)a.w;
    w[4] = (CT)b.x; w[5] = (CT)b.y; w[6] = (CT)b.z; w[7] = (CT)b.w;
  }
}

constexpr float kCoordLimit = 1.0e6f;  // beyond this a coordinate is "far outside"

// One interpolated sample in the compute type CT (float, or double for f64 images)
template <typename ST, int INTERP, typename C>
__device__ __forceinline__ typename compute_of<ST>::type sample(const SrcView& s, C sx, C sy,
                                                                typename compute_of<ST>::type cval) {
  using CT = typename compute_of<ST>::type;
  constexpr int NT = ntaps<INTERP>::value;
  if (!(sx > (C)-kCoordLimit && sx < (C)kCoordLimit && sy > (C)-kCoordLimit &&
        sy < (C)kCoordLimit)) {
    if (s.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cval;
    sx = sx < (C)-kCoordLimit ? (C)-kCoordLimit : (sx > (C)kCoordLimit ? (C)kCoordLimit : sx);
    sy = sy < (C)-kCoordLimit ? (C)-kCoordLimit : (sy > (C)kCoordLimit ? (C)kCoordLimit : sy);
  }
  int ix0, iy0;
  CT wx[NT], wy[NT];
  axis_split<INTERP, CT, C>(s, sx, ix0, wx);
  axis_split<INTERP, CT, C>(s, sy, iy0, wy);

  CT out = (CT)0;
  if (ix0 >= 0 && iy0 >= 0 && ix0 + NT <= s.w && iy0 + NT <= s.h) {
    // interior: one wide load per tap row
    int e = iy0 * s.pitch + ix0;
#pragma unroll
    for (int r = 0; r < NT; r++) {
      CT v[NT];
      TapLoad<ST, CT>::template row<NT>(s, e + r * s.pitch, v);
      CT rs = wx[0] * v[0];
#pragma unroll
      for (int c = 1; c < NT; c++) rs = ipa_fma(wx[c], v[c], rs);
      out = r == 0 ? wy[0] * rs : ipa_fma(wy[r], rs, out);
    }
    return out;
  }
  if (s.border == IPA_BORDER_CONSTANT &&
      (ix0 >= s.w || ix0 + NT <= 0 || iy0 >= s.h || iy0 + NT <= 0))
    return cval;  // whole footprint outside
#pragma unroll
  for (int r = 0; r < NT; r++) {
    int yy = resolve_idx(iy0 + r, s.h, s.border);
    CT rs = (CT)0;
#pragma unroll
    for (int c = 0; c < NT; c++) {
      int xx = resolve_idx(ix0 + c, s.w, s.border);
      CT v = (yy < 0 || xx < 0) ? cval : TapLoad<ST, CT>::one(s, yy * s.pitch + xx);
      rs = c == 0 ? wx[0] * v : ipa_fma(wx[c], v, rs);
    }
    out = r == 0 ? wy[0] * rs : ipa_fma(wy[r], rs, out);
  }
  return out;
}

// One interpolated sample for INTEGER destinations: every product and sum in double, no fused
// multiply-add, taps in row-major order (rs += wx[c] * v; out += wy[r] * rs) - the plain
// double arithmetic of a scalar CPU implementation built without fp contraction, operation
// for operation, so that the value handed to the round-half-even store is the same
// double and uint8 / uint16 results agree bit for bit (cv::saturate_cast semantics of
// cv2.remap on integer images, camera/LensDistortion.py:323-326).
template <typename ST, int INTERP, typename C>
__device__ __forceinline__ double sample_exact(const SrcView& s, C sx, C sy, double cval) {
#pragma clang fp contract(off)
  constexpr int NT = ntaps<INTERP>::value;
  if (!(sx > (C)-kCoordLimit && sx < (C)kCoordLimit && sy > (C)-kCoordLimit &&
        sy < (C)kCoordLimit)) {
    if (s.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cval;
    sx = sx < (C)-kCoordLimit ? (C)-kCoordLimit : (sx > (C)kCoordLimit ? (C)kCoordLimit : sx);
    sy = sy < (C)-kCoordLimit ? (C)-kCoordLimit : (sy > (C)kCoordLimit ? (C)kCoordLimit : sy);
  }
  int i0[2];
  double w[2][NT];
  const C cc[2] = {sx, sy};
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const double c = (double)cc[a];
    double fl, t;
    int k = 0;
    if (INTERP == kLanczos4 || s.q5) {
      const int qi = (int)rint(c * 32.0);  // cvRound(coordinate * INTER_TAB_SIZE)
      fl = (double)(qi >> 5);
      k = qi & 31;
      t = (double)k / 32.0;
    } else {
      fl = floor(c);
      t = c - fl;
    }
    if constexpr (INTERP == kNearest) {
      i0[a] = (int)rint(c);
      w[a][0] = 1.0;
    } else if constexpr (INTERP == kLinear) {
      i0[a] = (int)fl;
      w[a][0] = 1.0 - t;
      w[a][1] = t;
    } else if constexpr (INTERP == kCubic) {
      i0[a] = (int)fl - 1;
      const double A = (double)s.cubic_a;
      w[a][0] = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A;
      w[a][1] = ((A + 2) * t - (A + 3)) * t * t + 1;
      w[a][2] = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1;
      w[a][3] = 1.0 - w[a][0] - w[a][1] - w[a][2];
    } else {
      i0[a] = (int)fl - 3;
#pragma unroll
      for (int j = 0; j < 8; j++) w[a][j] = (double)s.lanczos[k * 8 + j];
    }
  }
  const int ix0 = i0[0], iy0 = i0[1];
  if (s.border == IPA_BORDER_CONSTANT &&
      (ix0 >= s.w || ix0 + NT <= 0 || iy0 >= s.h || iy0 + NT <= 0))
    return cval;  // whole footprint outside
  const bool interior = ix0 >= 0 && iy0 >= 0 && ix0 + NT <= s.w && iy0 + NT <= s.h;
  double out = 0.0;
  if (interior) {
    // one wide load per tap row (a Lanczos4 sample of a uint8 image: 8 dwordx2 instead of 64
    // byte loads); integer taps are exact in float, the sums below are the same operations
    using CT = typename compute_of<ST>::type;
    const int e = iy0 * s.pitch + ix0;
#pragma unroll
    for (int r = 0; r < NT; r++) {
      CT v[NT];
      TapLoad<ST, CT>::template row<NT>(s, e + r * s.pitch, v);
      double rs = 0.0;
      if constexpr (INTERP == kLanczos4 && !std::is_same<ST, double>::value) {
        // table weights are float32 values and the taps have at most 24 significant bits: the
        // products are exact in double, so the fused form rounds exactly like rs + w * v
#pragma unroll
        for (int c = 0; c < NT; c++) rs = __builtin_fma(w[0][c], (double)v[c], rs);
      } else {
#pragma unroll
        for (int c = 0; c < NT; c++) rs = rs + w[0][c] * (double)v[c];
      }
      out = out + w[1][r] * rs;
    }
    return out;
  }
#pragma unroll
  for (int r = 0; r < NT; r++) {
    const int yy = resolve_idx(iy0 + r, s.h, s.border);
    double rs = 0.0;
#pragma unroll
    for (int c = 0; c < NT; c++) {
      const int xx = resolve_idx(ix0 + c, s.w, s.border);
      const double v = (yy < 0 || xx < 0)
                           ? cval
                           : (double)TapLoad<ST, typename compute_of<ST>::type>::one(s, yy * s.pitch + xx);
      rs = rs + w[0][c] * v;
    }
    out = out + w[1][r] * rs;
  }
  return out;
}

// N samples at once, laid out for memory-level parallelism: the tap loads of
// the WHOLE batch are issued back to back (range-checked buffer loads cannot
// fault, so they are issued unconditionally at a clamped offset) before any
// value is consumed; only then are they blended.  Footprints that touch the
// image border (rare) are redone through sample().  Same arithmetic and
// summation order as sample(): results are bit-identical.
// Constant border mode (cv2.remap's default, and what the reference's alpha = 1 undistortion
// leaves along the rim of every picture): a footprint that touches the border keeps the fast
// path - its tap rows are loaded at their real offsets (range-checked: never a fault, a tap
// outside the source is just the wrong pixel), the taps outside are replaced by the border value
// in the blend, no tap inside gives the border value itself: sample()'s arithmetic in sample()'s
// order.  Only a footprint whose row loads start before the frame or run past its end (the
// range check may drop such a load whole) is redone through sample().
template <typename ST, int INTERP, int N, typename C>
__device__ __forceinline__ void sample_batch(const SrcView& s, const C (&sx)[N], const C (&sy)[N],
                                             typename compute_of<ST>::type cval,
                                             typename compute_of<ST>::type (&out)[N]) {
  using CT = typename compute_of<ST>::type;
  constexpr int NT = ntaps<INTERP>::value;
  int e[N];
  bool interior[N], redo[N];
  unsigned rmask[N], cmask[N];   // rows / columns of the footprint inside the source
  CT wx[N][NT], wy[N][NT];
#pragma unroll
  for (int k = 0; k < N; k++) {
    bool ok = sx[k] > (C)-kCoordLimit && sx[k] < (C)kCoordLimit && sy[k] > (C)-kCoordLimit &&
              sy[k] < (C)kCoordLimit;
    int ix0, iy0;
    axis_split<INTERP, CT, C>(s, ok ? sx[k] : (C)0, ix0, wx[k]);
    axis_split<INTERP, CT, C>(s, ok ? sy[k] : (C)0, iy0, wy[k]);
    interior[k] = ok && ix0 >= 0 && iy0 >= 0 && ix0 + NT <= s.w && iy0 + NT <= s.h;
    e[k] = interior[k] ? iy0 * s.pitch + ix0 : 0;
    redo[k] = !interior[k];
    rmask[k] = cmask[k] = (1u << NT) - 1u;
    if (!interior[k] && s.border == IPA_BORDER_CONSTANT) {
      rmask[k] = cmask[k] = 0u;
      if (ok) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
          rmask[k] |= (unsigned)(iy0 + t) < (unsigned)s.h ? 1u << t : 0u;
          cmask[k] |= (unsigned)(ix0 + t) < (unsigned)s.w ? 1u << t : 0u;
        }
      }
      if (rmask[k] == 0u || cmask[k] == 0u) {
        rmask[k] = cmask[k] = 0u;   // nothing of it inside: the border value
        redo[k] = false;
      } else {
        // first and last row that exist: their loads must lie inside the frame (element offsets;
        // a frame narrower than the footprint has rows that start before it further down)
        const int r0 = iy0 < 0 ? 0 : iy0, r1 = iy0 + NT - 1 < s.h - 1 ? iy0 + NT - 1 : s.h - 1;
        const long first = (long)r0 * s.pitch + ix0, last = (long)r1 * s.pitch + ix0 + NT;
        const bool before = first < 0, past = last > (long)(s.h - 1) * s.pitch + s.w;
        if (!before && !past) {
          e[k] = __mul24(iy0, s.pitch) + ix0;
          redo[k] = false;
        }
      }
    }
  }
  CT v[N][NT][NT];
#pragma unroll
  for (int k = 0; k < N; k++)
#pragma unroll
    for (int r = 0; r < NT; r++) TapLoad<ST, CT>::template row<NT>(s, e[k] + r * s.pitch, v[k][r]);
#pragma unroll
  for (int k = 0; k < N; k++) {
    if (!interior[k] && !redo[k]) {   // (rare) taps outside the source: the border value
#pragma unroll
      for (int r = 0; r < NT; r++)
#pragma unroll
        for (int c = 0; c < NT; c++)
          v[k][r][c] = ((rmask[k] >> r) & (cmask[k] >> c) & 1u) ? v[k][r][c] : cval;
    }
    CT o = (CT)0;
#pragma unroll
    for (int r = 0; r < NT; r++) {
      CT rs = wx[k][0] * v[k][r][0];
#pragma unroll
      for (int c = 1; c < NT; c++) rs = ipa_fma(wx[k][c], v[k][r][c], rs);
      o = r == 0 ? wy[k][0] * rs : ipa_fma(wy[k][r], rs, o);
    }
    out[k] = (!interior[k] && !redo[k] && rmask[k] == 0u) ? cval : o;
  }
#pragma unroll
  for (int k = 0; k < N; k++)
    if (redo[k]) out[k] = sample<ST, INTERP, C>(s, sx[k], sy[k], cval);
}

// ---- split form of sample_batch for software-pipelined callers (wave_stencil.hpp):
// batch_issue() computes the footprints and ISSUES every tap load of the batch,
// batch_blend() (called later, after more loads were queued) consumes them.
// Only the fractions are kept between the two halves; bilinear / bicubic only.
template <typename ST, int INTERP, int N> struct BatchTaps {
  using CT = typename compute_of<ST>::type;
  static constexpr int NT = ntaps<INTERP>::value;
  CT v[N][NT][NT];
  CT tx[N], ty[N];
  unsigned interior;  // bit k: footprint k fully inside the source
  // bilinear, constant border mode: which of the 4 taps of a footprint that is NOT interior lie
  // inside the source (bits 4 k .. 4 k + 3: x0y0, x1y0, x0y1, x1y1) - batch_blend_border() blends
  // it from the taps issued, the others replaced by the border value; bit 31: a footprint whose
  // row loads start before the frame or run past its end (sample() redoes the lane's)
  unsigned tapbits;
};

// QM selects the coordinate rule at compile time where the caller can (the per-strip loops
// of wave_stencil.hpp): -1 = s.q5 decides at run time, 0 = exact, 1 = 1/32-px rounding
template <int INTERP, typename CT, typename C, int QM = -1>
__device__ __forceinline__ void axis_frac(const SrcView& s, C c, int& i0, CT& t) {
  int ip;
  const bool q5 = QM < 0 ? s.q5 != 0 : QM == 1;
  if (q5) {
    int qi = (int)ipa_rint(c * (C)32);
    ip = qi >> 5;
    t = (CT)(qi & 31) * (CT)0.03125;
  } else {
    C fl = ipa_floor(c);
    ip = (int)fl;
    if constexpr (sizeof(CT) > sizeof(C)) t = (CT)c - (CT)fl;
    else t = (CT)(c - fl);
  }
  i0 = INTERP == kLinear ? ip : ip - 1;
}

template <int INTERP, typename CT>
__device__ __forceinline__ void weights_from_frac(const SrcView& s, CT t,
                                                  CT (&w)[ntaps<INTERP>::value]) {
  if constexpr (INTERP == kLinear) {
    w[0] = (CT)1 - t;
    w[1] = t;
  } else {
    cubic_weights<CT>(t, (CT)s.cubic_a, w);
  }
}

template <typename ST, int INTERP, int N, int QM = -1, typename C>
__device__ __forceinline__ void batch_issue(const SrcView& s, const C (&sx)[N], const C (&sy)[N],
                                            BatchTaps<ST, INTERP, N>& b) {
  using CT = typename compute_of<ST>::type;
  constexpr int NT = ntaps<INTERP>::value;
  static_assert(INTERP == kLinear || INTERP == kCubic, "split sampling: bilinear/bicubic only");
  b.interior = 0;
  b.tapbits = 0;
  int e[N];
  const unsigned xlim = s.w - NT + 1 > 0 ? (unsigned)(s.w - NT + 1) : 0u;
  const unsigned ylim = s.h - NT + 1 > 0 ? (unsigned)(s.h - NT + 1) : 0u;
#pragma unroll
  for (int k = 0; k < N; k++) {
    // the same decisions with fewer instructions (this loop is a third of the fused kernel's
    // VALU work): |c| < limit is one compare with a source modifier and false for NaN; a
    // footprint lies inside when (unsigned)i0 < extent - NT + 1
    const bool ok = ipa_abs(sx[k]) < (C)kCoordLimit && ipa_abs(sy[k]) < (C)kCoordLimit;
    int ix0, iy0;
    axis_frac<INTERP, CT, C, QM>(s, ok ? sx[k] : (C)0, ix0, b.tx[k]);
    axis_frac<INTERP, CT, C, QM>(s, ok ? sy[k] : (C)0, iy0, b.ty[k]);
    const bool in = ok && (unsigned)ix0 < xlim && (unsigned)iy0 < ylim;
    b.interior |= in ? (1u << k) : 0u;
    if constexpr (INTERP == kLinear) {
      if (!in && s.border == IPA_BORDER_CONSTANT) {   // (rare)
        const bool x0 = (unsigned)ix0 < (unsigned)s.w, x1 = (unsigned)(ix0 + 1) < (unsigned)s.w;
        const bool y0 = (unsigned)iy0 < (unsigned)s.h, y1 = (unsigned)(iy0 + 1) < (unsigned)s.h;
        const unsigned tb = ((y0 && x0) ? 1u : 0u) | ((y0 && x1) ? 2u : 0u) | ((y1 && x0) ? 4u : 0u) |
                            ((y1 && x1) ? 8u : 0u);
        b.tapbits |= (ok ? tb : 0u) << (4 * k);
        const bool before = ix0 < 0 && (iy0 == 0 || iy0 == -1);
        const bool past = ix0 + 2 > s.w && (iy0 == s.h - 1 || iy0 == s.h - 2);
        if (ok && tb && (before || past)) b.tapbits |= 1u << 31;
      }
    }
    // |iy0| <= kCoordLimit < 2^23 and pitch < 2^23 (checked at the entry points): one
    // full-rate v_mul_i32_i24 (+ add) instead of a quarter-rate 32-bit multiply.  Footprints
    // that are not inside load from wherever this lands - range-checked buffer loads return 0
    // beyond the frame - and are blended by batch_blend_border() or redone by sample()
    e[k] = __mul24(iy0, s.pitch) + ix0;
  }
#pragma unroll
  for (int k = 0; k < N; k++)
#pragma unroll
    for (int r = 0; r < NT; r++)
      TapLoad<ST, CT>::template row<NT>(s, e[k] + r * s.pitch, b.v[k][r]);
}

// The two halves of batch_issue() for callers that sample SEVERAL frames at the same
// coordinates (wave_pair.hpp): footprints once, the tap loads per frame.  Bilinear float frames
// in the lane-interleaved order (dword pairs).
template <int N, int QM, typename C>
__device__ __forceinline__ void batch_footprint_linear(const SrcView& s, const C (&sx)[N],
                                                       const C (&sy)[N], float (&tx)[N],
                                                       float (&ty)[N], int (&e)[N],
                                                       unsigned& interior) {
  constexpr int NT = 2;
  interior = 0;
  const unsigned xlim = s.w - NT + 1 > 0 ? (unsigned)(s.w - NT + 1) : 0u;
  const unsigned ylim = s.h - NT + 1 > 0 ? (unsigned)(s.h - NT + 1) : 0u;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const bool ok = ipa_abs(sx[k]) < (C)kCoordLimit && ipa_abs(sy[k]) < (C)kCoordLimit;
    int ix0, iy0;
    axis_frac<kLinear, float, C, QM>(s, ok ? sx[k] : (C)0, ix0, tx[k]);
    axis_frac<kLinear, float, C, QM>(s, ok ? sy[k] : (C)0, iy0, ty[k]);
    const bool in = ok && (unsigned)ix0 < xlim && (unsigned)iy0 < ylim;
    interior |= in ? (1u << k) : 0u;
    e[k] = __mul24(iy0, s.pitch) + ix0;
  }
}
template <int N>
__device__ __forceinline__ void batch_loads_linear(const SrcView& s, const int (&e)[N],
                                                   BatchTaps<float, kLinear, N>& b) {
  int four = 4;
  asm("" : "+s"(four));  // (see TapLoad<float, float>::row: keeps the two dwords apart)
#pragma unroll
  for (int k = 0; k < N; k++)
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int o = (e[k] + r * s.pitch) << 2;
      b.v[k][r][0] = u2f(__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, o, 0, 0));
      b.v[k][r][1] = u2f(__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, o + four, 0, 0));
    }
}

// blend footprint k of an issued batch (same arithmetic / order as sample())
template <typename ST, int INTERP, int N>
__device__ __forceinline__ typename compute_of<ST>::type batch_blend_one(
    const SrcView& s, const BatchTaps<ST, INTERP, N>& b, int k) {
  using CT = typename compute_of<ST>::type;
  constexpr int NT = ntaps<INTERP>::value;
  CT wx[NT], wy[NT];
  weights_from_frac<INTERP, CT>(s, b.tx[k], wx);
  weights_from_frac<INTERP, CT>(s, b.ty[k], wy);
  CT o = (CT)0;
#pragma unroll
  for (int r = 0; r < NT; r++) {
    CT rs = wx[0] * b.v[k][r][0];
#pragma unroll
    for (int c = 1; c < NT; c++) rs = ipa_fma(wx[c], b.v[k][r][c], rs);
    o = r == 0 ? wy[0] * rs : ipa_fma(wy[r], rs, o);
  }
  return o;
}

// footprint k of an issued batch that touches the source border: true + its value where the taps
// issued can give it (bilinear, constant border mode: the taps outside the source replaced by the
// border value, none inside = the border value; sample()'s arithmetic), false where sample() has to
template <typename ST, int INTERP, int N>
__device__ __forceinline__ bool batch_blend_border(const SrcView& s, const BatchTaps<ST, INTERP, N>& b,
                                                   int k, typename compute_of<ST>::type cval,
                                                   typename compute_of<ST>::type& out) {
  using CT = typename compute_of<ST>::type;
  if constexpr (INTERP != kLinear) {
    return false;
  } else {
    if (s.border != IPA_BORDER_CONSTANT || (b.tapbits >> 31)) return false;
    const unsigned tb = (b.tapbits >> (4 * k)) & 15u;
    const CT v00 = (tb & 1u) ? b.v[k][0][0] : cval, v01 = (tb & 2u) ? b.v[k][0][1] : cval;
    const CT v10 = (tb & 4u) ? b.v[k][1][0] : cval, v11 = (tb & 8u) ? b.v[k][1][1] : cval;
    CT wx[2], wy[2];
    weights_from_frac<kLinear, CT>(s, b.tx[k], wx);
    weights_from_frac<kLinear, CT>(s, b.ty[k], wy);
    CT r0 = wx[0] * v00;
    r0 = ipa_fma(wx[1], v01, r0);
    CT o = wy[0] * r0;
    CT r1 = wx[0] * v10;
    r1 = ipa_fma(wx[1], v11, r1);
    o = ipa_fma(wy[1], r1, o);
    out = tb ? o : cval;
    return true;
  }
}

// batch width per interpolation: bounded by the tap registers (N * NT^2)
template <int INTERP> struct batch_of { static constexpr int value = 4; };
template <> struct batch_of<kCubic> { static constexpr int value = 2; };
template <> struct batch_of<kLanczos4> { static constexpr int value = 1; };

// cv2's uint8 bilinear: q5 coordinates, exact 15-bit integer weights
// ((32-fx)(32-fy)*32 ...), rounded shift.  Integer arithmetic: bit-exact.
template <typename C>
__device__ __forceinline__ uint8_t sample_u8_fixed(const SrcView& s, C sx, C sy, uint8_t cv8) {
  if (!(sx > (C)-kCoordLimit && sx < (C)kCoordLimit && sy > (C)-kCoordLimit &&
        sy < (C)kCoordLimit)) {
    if (s.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cv8;
    sx = sx < (C)-kCoordLimit ? (C)-kCoordLimit : (sx > (C)kCoordLimit ? (C)kCoordLimit : sx);
    sy = sy < (C)-kCoordLimit ? (C)-kCoordLimit : (sy > (C)kCoordLimit ? (C)kCoordLimit : sy);
  }
  int qx = (int)ipa_rint(sx * (C)32), qy = (int)ipa_rint(sy * (C)32);
  int ix = qx >> 5, iy = qy >> 5, fx = qx & 31, fy = qy & 31;
  int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
  int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  int v[4];
  if (ix >= 0 && iy >= 0 && ix + 2 <= s.w && iy + 2 <= s.h) {
    // interior footprint: each tap row is one ushort load (any byte offset)
    const unsigned r0 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
        s.rsrc, iy * s.pitch + ix, 0, 0);
    const unsigned r1 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
        s.rsrc, (iy + 1) * s.pitch + ix, 0, 0);
    v[0] = (int)(r0 & 0xffu); v[1] = (int)(r0 >> 8);
    v[2] = (int)(r1 & 0xffu); v[3] = (int)(r1 >> 8);
  } else {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      int yy = resolve_idx(iy + r, s.h, s.border);
#pragma unroll
      for (int c = 0; c < 2; c++) {
        int xx = resolve_idx(ix + c, s.w, s.border);
        v[r * 2 + c] = (yy < 0 || xx < 0) ? (int)cv8
                                          : (int)__builtin_amdgcn_raw_buffer_load_b8(
                                                s.rsrc, yy * s.pitch + xx, 0, 0);
      }
    }
  }
  int acc = v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11;
  int o = (acc + (1 << 14)) >> 15;
  return (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
}

// cv2's uint8 bicubic (A = -0.75) / Lanczos4 (imgwarp.cpp remapBicubic / remapLanczos4 with
// FixedPtCast<int, uchar, 15>): q5 coordinates; the ks x ks short weights of the fraction pair,
// saturate_cast<short>(wy[k1] * wx[k2] * 2^15) from the float32 1-D tables (s.lanczos: Lanczos4
// rows at 0, bicubic rows at 256), their sum forced to 2^15 on one entry of the 2 x 2 block at
// (ks/2, ks/2) as OpenCV's initInterTab2D does; integer accumulation, rounded shift.  The weights
// are formed per sample: fetching them from the 32 / 128 KB 2-D tables costs a gather of 64
// different cache lines per wave instruction (measured 13 % slower).  Bit-exact.
template <int INTERP, typename C>
__device__ __forceinline__ uint8_t sample_u8_tab(const SrcView& s, C sx, C sy, uint8_t cv8) {
  constexpr int NT = ntaps<INTERP>::value;
  constexpr int H = NT / 2;
  if (!(sx > (C)-kCoordLimit && sx < (C)kCoordLimit && sy > (C)-kCoordLimit &&
        sy < (C)kCoordLimit)) {
    if (s.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cv8;
    sx = sx < (C)-kCoordLimit ? (C)-kCoordLimit : (sx > (C)kCoordLimit ? (C)kCoordLimit : sx);
    sy = sy < (C)-kCoordLimit ? (C)-kCoordLimit : (sy > (C)kCoordLimit ? (C)kCoordLimit : sy);
  }
  const int qx = (int)ipa_rint(sx * (C)32), qy = (int)ipa_rint(sy * (C)32);
  const int ix0 = (qx >> 5) - (H - 1), iy0 = (qy >> 5) - (H - 1);
  if (s.border == IPA_BORDER_CONSTANT &&
      (ix0 >= s.w || ix0 + NT <= 0 || iy0 >= s.h || iy0 + NT <= 0))
    return cv8;  // whole footprint outside
  const float* tab = s.lanczos + (INTERP == kLanczos4 ? 0 : 256);
  float wx[NT], wy[NT];
#pragma unroll
  for (int k = 0; k < NT; k++) {
    wx[k] = tab[(qx & 31) * NT + k];
    wy[k] = tab[(qy & 31) * NT + k];
  }
  int it[NT][NT];
  int isum = 0;
#pragma unroll
  for (int k1 = 0; k1 < NT; k1++)
#pragma unroll
    for (int k2 = 0; k2 < NT; k2++) {
      // cvRound(fl(wy * wx) * 2^15): the scaling is exact, so one fused multiply-add onto
      // 1.5 * 2^23 rounds it half-to-even into the low bits of the sum
      const float v = wy[k1] * wx[k2];
      const float r = __builtin_fmaf(v, 32768.f, 12582912.f);
      int q = (int)__float_as_uint(r) - 0x4B400000;
      // the weights are <= 1 in magnitude: only 1 * 1 (the centre tap at fraction 0) reaches
      // 2^15 and saturates
      if (k1 == H - 1 && k2 == H - 1) q = q > 32767 ? 32767 : q;
      it[k1][k2] = q;
      isum += q;
    }
  {
    // the 2 x 2 block at (H, H) in row-major order: first minimum, first maximum
    const int diff = isum - 32768;
    const int b[4] = {it[H][H], it[H][H + 1], it[H + 1][H], it[H + 1][H + 1]};
    int mi = 0, Mi = 0, mv = b[0], Mv = b[0];
#pragma unroll
    for (int i = 1; i < 4; i++) {
      if (b[i] < mv) { mv = b[i]; mi = i; }
      else if (b[i] > Mv) { Mv = b[i]; Mi = i; }
    }
    const int target = diff < 0 ? Mi : mi;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int& e = it[H + (i >> 1)][H + (i & 1)];
      e = (i == target && diff != 0) ? (int)(short)(e - diff) : e;
    }
  }
  int acc = 0;
  if (ix0 >= 0 && iy0 >= 0 && ix0 + NT <= s.w && iy0 + NT <= s.h) {
    // interior footprint: a tap row is one / two dword loads at any byte offset
    unsigned taps[NT][NT / 4];
#pragma unroll
    for (int r = 0; r < NT; r++)
#pragma unroll
      for (int q = 0; q < NT / 4; q++)
        taps[r][q] = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, (iy0 + r) * s.pitch + ix0 + 4 * q,
                                                          0, 0);
#pragma unroll
    for (int r = 0; r < NT; r++)
#pragma unroll
      for (int c = 0; c < NT; c++)
        acc += (int)((taps[r][c >> 2] >> (8 * (c & 3))) & 0xffu) * it[r][c];
  } else {
#pragma unroll
    for (int r = 0; r < NT; r++) {
      const int yy = resolve_idx(iy0 + r, s.h, s.border);
#pragma unroll
      for (int c = 0; c < NT; c++) {
        const int xx = resolve_idx(ix0 + c, s.w, s.border);
        const int v = (yy < 0 || xx < 0) ? (int)cv8
                                         : (int)__builtin_amdgcn_raw_buffer_load_b8(
                                               s.rsrc, yy * s.pitch + xx, 0, 0);
        acc += v * it[r][c];
      }
    }
  }
  const int o = (acc + (1 << 14)) >> 15;
  return (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
}

// The same for bicubic with the 1024 x 16 short weights of OpenCV's table resident in LDS
// (tab2d: row fy * 32 + fx = 8 dwords {w0 | w2 << 16, w1 | w3 << 16} per tap row): two 16-byte LDS
// reads and 8 v_dot2_i32_i16 per sample instead of forming 16 weights (the formed-per-sample
// version is VALU-bound at 184 instructions per sample).
typedef short v2s __attribute__((ext_vector_type(2)));
#ifndef IPA_U8_CUBIC_ROW
#define IPA_U8_CUBIC_ROW 2
#endif
constexpr int kU8CubicRow = IPA_U8_CUBIC_ROW;  // int4 per fraction pair in LDS (2 used)
template <typename C>
__device__ __forceinline__ uint8_t sample_u8_cubic_lds(const SrcView& s, const int4* tab2d, C sx,
                                                       C sy, uint8_t cv8) {
  if (!(sx > (C)-kCoordLimit && sx < (C)kCoordLimit && sy > (C)-kCoordLimit &&
        sy < (C)kCoordLimit)) {
    if (s.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cv8;
    sx = sx < (C)-kCoordLimit ? (C)-kCoordLimit : (sx > (C)kCoordLimit ? (C)kCoordLimit : sx);
    sy = sy < (C)-kCoordLimit ? (C)-kCoordLimit : (sy > (C)kCoordLimit ? (C)kCoordLimit : sy);
  }
  const int qx = (int)ipa_rint(sx * (C)32), qy = (int)ipa_rint(sy * (C)32);
  const int ix0 = (qx >> 5) - 1, iy0 = (qy >> 5) - 1;
  if (s.border == IPA_BORDER_CONSTANT && (ix0 >= s.w || ix0 + 4 <= 0 || iy0 >= s.h || iy0 + 4 <= 0))
    return cv8;  // whole footprint outside
  const int4* wrow = tab2d + (((qy & 31) << 5) | (qx & 31)) * kU8CubicRow;
  const int4 wa = wrow[0], wb = wrow[1];
  const int wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
  unsigned taps[4];
  if (ix0 >= 0 && iy0 >= 0 && ix0 + 4 <= s.w && iy0 + 4 <= s.h) {
#pragma unroll
    for (int r = 0; r < 4; r++)
      taps[r] = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, (iy0 + r) * s.pitch + ix0, 0, 0);
  } else {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int yy = resolve_idx(iy0 + r, s.h, s.border);
      unsigned t = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int xx = resolve_idx(ix0 + c, s.w, s.border);
        const unsigned b = (yy < 0 || xx < 0) ? (unsigned)cv8
                                              : (unsigned)__builtin_amdgcn_raw_buffer_load_b8(
                                                    s.rsrc, yy * s.pitch + xx, 0, 0) & 0xffu;
        t |= b << (8 * c);
      }
      taps[r] = t;
    }
  }
  int acc = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const unsigned even = taps[r] & 0x00ff00ffu, odd = (taps[r] >> 8) & 0x00ff00ffu;
    acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, even), __builtin_bit_cast(v2s, wv[2 * r]), acc,
                                 false);
    acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, odd), __builtin_bit_cast(v2s, wv[2 * r + 1]),
                                 acc, false);
  }
  const int o = (acc + (1 << 14)) >> 15;
  return (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
}

// uint8 Lanczos4 with OpenCV's 1024 x 64 short weights resident in LDS (remap_u8_lz_kernel: 128 KB,
// row fy * 32 + fx = 8 int4, one per tap row): eight 16-byte LDS reads, 16 tap dwords and 32
// v_dot2_i32_i16 per sample instead of forming 64 weights.
constexpr int kU8LzRow = 9;  // int4 per fraction pair in LDS
template <typename C>
__device__ __forceinline__ uint8_t sample_u8_lanczos_lds(const SrcView& s, const int4* tab2d, C sx,
                                                         C sy, uint8_t cv8) {
  if (!(sx > (C)-kCoordLimit && sx < (C)kCoordLimit && sy > (C)-kCoordLimit &&
        sy < (C)kCoordLimit)) {
    if (s.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cv8;
    sx = sx < (C)-kCoordLimit ? (C)-kCoordLimit : (sx > (C)kCoordLimit ? (C)kCoordLimit : sx);
    sy = sy < (C)-kCoordLimit ? (C)-kCoordLimit : (sy > (C)kCoordLimit ? (C)kCoordLimit : sy);
  }
  const int qx = (int)ipa_rint(sx * (C)32), qy = (int)ipa_rint(sy * (C)32);
  const int ix0 = (qx >> 5) - 3, iy0 = (qy >> 5) - 3;
  if (s.border == IPA_BORDER_CONSTANT && (ix0 >= s.w || ix0 + 8 <= 0 || iy0 >= s.h || iy0 + 8 <= 0))
    return cv8;  // whole footprint outside
  // rows of the LDS table are 9 int4 apart (8 used): fraction pairs then start on 16 different
  // bank groups instead of 2
  const int4* wrow = tab2d + (((qy & 31) << 5) | (qx & 31)) * kU8LzRow;
  // pr[r][j] = taps 2j, 2j+1 of row r as two 16-bit lanes (the table packs its shorts the same way)
  unsigned pr[8][4];
  // (the three dwords of a tap row reach up to 3 bytes past the footprint: in the LAST row of the
  // frame that is past the frame's buffer descriptor, which answers the whole dword with 0 -
  // such footprints take the byte path.  Found by tests/fuzz_oracle.py: one pixel in the
  // bottom-right corner off by one level.)
  const int e7 = s.org + (iy0 + 7) * s.pitch + ix0;   // first tap of the last tap row
  const bool tail_ok = iy0 + 8 < s.h || (e7 & ~3) + 12 <= s.org + (s.h - 1) * s.pitch + s.w;
  if (ix0 >= 0 && iy0 >= 0 && ix0 + 8 <= s.w && iy0 + 8 <= s.h && tail_ok) {
    // three ALIGNED dwords per tap row (two dwords at the sample's odd byte offset cost the
    // texture addresser a third more); one v_perm_b32 per tap pair picks bytes sh + 2j, sh + 2j + 1
    // out of two of them and widens them to 16 bits.  Aligned in MEMORY, row by row: a pitch that
    // is no multiple of 4 (an odd width) shifts every row differently (16 x 2160 x 3838: 1318 us
    // with the dwords aligned to the row's start only, 937 in memory; 919 at width 3840), and an
    // odd frame size every other frame (s.org; 16 x 1079 x 1919: 449 -> 259 us)
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int e = s.org + (iy0 + r) * s.pitch + ix0;
      const int a = e & ~3;
      const unsigned sh = (unsigned)e & 3u;
      const unsigned sel01 = 0x0c010c00u + sh * 0x00010001u, sel23 = sel01 + 0x00020002u;
      const unsigned d0 = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, a, 0, 0);
      const unsigned d1 = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, a + 4, 0, 0);
      const unsigned d2 = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, a + 8, 0, 0);
      pr[r][0] = __builtin_amdgcn_perm(d1, d0, sel01);
      pr[r][1] = __builtin_amdgcn_perm(d1, d0, sel23);
      pr[r][2] = __builtin_amdgcn_perm(d2, d1, sel01);
      pr[r][3] = __builtin_amdgcn_perm(d2, d1, sel23);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int yy = resolve_idx(iy0 + r, s.h, s.border);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        unsigned t = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int xx = resolve_idx(ix0 + 4 * q + c, s.w, s.border);
          const unsigned b = (yy < 0 || xx < 0) ? (unsigned)cv8
                                                : (unsigned)__builtin_amdgcn_raw_buffer_load_b8(
                                                      s.rsrc, s.org + yy * s.pitch + xx, 0, 0) & 0xffu;
          t |= b << (8 * c);
        }
        pr[r][2 * q] = (t & 0xffu) | ((t & 0xff00u) << 8);
        pr[r][2 * q + 1] = ((t >> 16) & 0xffu) | ((t >> 24) << 16);
      }
    }
  }
  int acc = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int4 wq = wrow[r];
    const int wv[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
    for (int j = 0; j < 4; j++)
      acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, pr[r][j]), __builtin_bit_cast(v2s, wv[j]), acc,
                                   false);
  }
  const int o = (acc + (1 << 14)) >> 15;
  return (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
}

// OpenCV's remap arithmetic on CV_16U frames (what cv2.remap / warpPerspective do on the camera's
// uint16 frames in LensDistortion.correct, camera/LensDistortion.py:323-326, and
// PerspectiveCorrection.correct, camera/PerspectiveCorrection.py:401-405), restated from the
// published algorithm (imgwarp.cpp: remapBilinear / remapBicubic / remapLanczos4 instantiated
// with Cast<float, ushort>, weights float):
//   * coordinates rounded to 1/32 px; the 1-D weights of the 32 fractions in float32
//     (interpolateLinear / interpolateCubic / interpolateLanczos4), the 2-D weight of a tap the
//     float32 PRODUCT wy[r] * wx[c] (initInterTab2D without fixed point);
//   * float32 accumulation, every product and every sum rounded, no fused multiply-add:
//       bilinear          v00 w00 + v01 w01 + v10 w10 + v11 w11, left to right (taps outside the
//                         frame replaced by the border value first);
//       bicubic, inside   the 16 products summed left to right in row-major order;
//       Lanczos4, inside  per tap row the 8 products summed left to right, the row sums added;
//       bicubic / Lanczos4 with taps outside the frame:  sum = cv, then for every tap that
//                         exists (rows, columns through the border mode)  sum += (S - cv) * w;
//   * saturate_cast<ushort>(sum) = round half to even, clamp to [0, 65535].
// `tab`: the float32 1-D tables ([0, 256) Lanczos4 rows, [256, 384) bicubic rows) of
// ipa_lanczos_table.  Unpinned against cv2 like the other cv2 modes: the
// expression order is restated from memory of the OpenCV 4.x source.
template <int INTERP, typename C>
__device__ __forceinline__ uint16_t sample_u16_cv(const SrcView& s, const float* tab, C sx, C sy,
                                                  uint16_t cv16) {
#pragma clang fp contract(off)
  constexpr int ks = ntaps<INTERP>::value;
  if (!(sx > (C)-kCoordLimit && sx < (C)kCoordLimit && sy > (C)-kCoordLimit &&
        sy < (C)kCoordLimit)) {
    if (s.border == IPA_BORDER_CONSTANT || sx != sx || sy != sy) return cv16;
    sx = sx < (C)-kCoordLimit ? (C)-kCoordLimit : (sx > (C)kCoordLimit ? (C)kCoordLimit : sx);
    sy = sy < (C)-kCoordLimit ? (C)-kCoordLimit : (sy > (C)kCoordLimit ? (C)kCoordLimit : sy);
  }
  const int qx = (int)ipa_rint(sx * (C)32), qy = (int)ipa_rint(sy * (C)32);
  const int ix0 = (qx >> 5) - (ks / 2 - 1), iy0 = (qy >> 5) - (ks / 2 - 1);
  if (s.border == IPA_BORDER_CONSTANT &&
      (ix0 >= s.w || ix0 + ks <= 0 || iy0 >= s.h || iy0 + ks <= 0))
    return cv16;  // whole footprint outside
  float wx[ks], wy[ks];
  if constexpr (INTERP == kLinear) {
    const float fx = (float)(qx & 31) * 0.03125f, fy = (float)(qy & 31) * 0.03125f;
    wx[0] = 1.f - fx; wx[1] = fx;
    wy[0] = 1.f - fy; wy[1] = fy;
  } else {
    const float* t = tab + (INTERP == kCubic ? 256 : 0);
#pragma unroll
    for (int k = 0; k < ks; k++) {
      wx[k] = t[(qx & 31) * ks + k];
      wy[k] = t[(qy & 31) * ks + k];
    }
  }
  const float cv = (float)cv16;
  const bool inside = ix0 >= 0 && iy0 >= 0 && ix0 + ks <= s.w && iy0 + ks <= s.h;
  auto tap = [&](int yy, int xx) -> float {
    return (float)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(s.rsrc, (yy * s.pitch + xx) << 1, 0, 0);
  };
  float sum;
  if constexpr (INTERP == kLinear) {
    float v[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int yy = inside ? iy0 + r : resolve_idx(iy0 + r, s.h, s.border);
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int xx = inside ? ix0 + c : resolve_idx(ix0 + c, s.w, s.border);
        v[r][c] = (yy < 0 || xx < 0) ? cv : tap(yy < 0 ? 0 : yy, xx < 0 ? 0 : xx);
      }
    }
    sum = v[0][0] * (wy[0] * wx[0]);
    sum = sum + v[0][1] * (wy[0] * wx[1]);
    sum = sum + v[1][0] * (wy[1] * wx[0]);
    sum = sum + v[1][1] * (wy[1] * wx[1]);
  } else if (inside) {
    sum = 0.f;
#pragma unroll
    for (int r = 0; r < ks; r++) {
      float rs = 0.f;
#pragma unroll
      for (int c = 0; c < ks; c++) {
        const float pr = tap(iy0 + r, ix0 + c) * (wy[r] * wx[c]);
        if constexpr (INTERP == kCubic) sum = (r == 0 && c == 0) ? pr : sum + pr;   // one flat sum
        else rs = c == 0 ? pr : rs + pr;                                            // row sums
      }
      if constexpr (INTERP != kCubic) sum = sum + rs;
    }
  } else {
    sum = cv;
    for (int r = 0; r < ks; r++) {
      const int yy = resolve_idx(iy0 + r, s.h, s.border);
      if (yy < 0) continue;
      for (int c = 0; c < ks; c++) {
        const int xx = resolve_idx(ix0 + c, s.w, s.border);
        if (xx >= 0) sum = sum + (tap(yy, xx) - cv) * (wy[r] * wx[c]);
      }
    }
  }
  float r = rintf(sum);
  r = r > 0.f ? r : 0.f;  // NaN -> 0
  r = r < 65535.f ? r : 65535.f;
  return (uint16_t)r;
}

// ------------------------------------------------------- coordinate sources --
// Each provides  coord_t  and  get(u, v, sx, sy)  for destination pixel (u,v).

struct MapCoord {
  using coord_t = float;
  const float* mx;
  const float* my;
  long pitch;
  __device__ __forceinline__ void get(int u, int v, float& sx, float& sy) const {
    long o = (long)v * pitch + u;
    sx = mx[o];
    sy = my[o];
  }
};

// coordinates a planning pass stored in the coordinate source's own type (ring_stencil.hpp):
// the frames of a batch then read them instead of evaluating the source again
template <typename T> struct StoredCoord {
  using coord_t = T;
  const T* mx;
  const T* my;
  long pitch;
  __device__ __forceinline__ void get(int u, int v, T& sx, T& sy) const {
    long o = (long)v * pitch + u;
    sx = mx[o];
    sy = my[o];
  }
};
template <typename Coord> struct coord_is_table : std::false_type {};
template <> struct coord_is_table<MapCoord> : std::true_type {};
template <typename T> struct coord_is_table<StoredCoord<T>> : std::true_type {};

// cv2.initUndistortRectifyMap with R = I evaluated per pixel in double and
// rounded to the float32 a CV_32FC1 map stores (camera/LensDistortion.py:355-357).
// No fp contraction here: the map builder, the analytic kernels and the CPU
// oracle must produce the same float32 coordinate bit for bit.
struct UndistortCoord {
  using coord_t = float;
  double ir[9];  // inv(newK)
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
  int affine;  // ir[6]==ir[7]==0 && ir[8]==1  ->  _w == 1 exactly
  __device__ __forceinline__ void get(int u, int v, float& sx, float& sy) const {
#pragma clang fp contract(off)
    double du = (double)u, dv = (double)v;
    double _x = ir[0] * du + ir[1] * dv + ir[2];
    double _y = ir[3] * du + ir[4] * dv + ir[5];
    double x = _x, y = _y;
    if (!affine) {
      double _w = ir[6] * du + ir[7] * dv + ir[8];
      double iw = 1.0 / _w;
      x = _x * iw;
      y = _y * iw;
    }
    double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
    double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
    double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
    double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
    sx = (float)(fx * xd + cx);
    sy = (float)(fy * yd + cy);
  }
};

// cv2.warpPerspective: source = M * (u, v, 1), M = dst->src matrix, double
// coordinates (skimage _warp_fast keeps them in double too).
struct HomographyCoord {
  using coord_t = double;
  double m[9];
  __device__ __forceinline__ void get(int u, int v, double& sx, double& sy) const {
#pragma clang fp contract(off)
    double du = (double)u, dv = (double)v;
    double X = m[0] * du + m[1] * dv + m[2];
    double Y = m[3] * du + m[4] * dv + m[5];
    double W = m[6] * du + m[7] * dv + m[8];
    if (W != 0.0) {
      double iw = 1.0 / W;
      sx = X * iw;
      sy = Y * iw;
    } else {
      sx = 0.0;
      sy = 0.0;
    }
  }
};

}  // namespace ipa
