// remap.hip — C-ABI entry points of the gather half of the hot path
// (kernels + dispatch: remap_impl.hpp, one translation unit per coordinate source).
#include <math.h>

#define IPA_REMAP_API_TU
#include "remap_impl.hpp"

int ipa_remap_launch_map(ipa_ctx*, const RemapCall&, const MapCoord&, int map_vec);
int ipa_remap_launch_undistort(ipa_ctx*, const RemapCall&, const UndistortCoord&);
int ipa_remap_launch_homography(ipa_ctx*, const RemapCall&, const HomographyCoord&);

// ---------------------------------------------------------------- host side --
static int inv3(const double* m, double* o) {
  double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + b * B + c * C;
  if (det == 0 || det != det) return -1;
  double id = 1.0 / det;
  o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
  o[3] = B * id; o[4] = (a * i - c * g) * id;  o[5] = -(a * f - c * d) * id;
  o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
  return 0;
}

static int make_undistort_coord(ipa_ctx* ctx, const double* K, const double* d, const double* newK,
                                UndistortCoord* c) {
  IPA_REQUIRE(ctx, K && d && newK, "K, dist5 and newK must be given");
  IPA_REQUIRE(ctx, inv3(newK, c->ir) == 0, "newK is singular");
  c->fx = K[0]; c->fy = K[4]; c->cx = K[2]; c->cy = K[5];
  c->k1 = d[0]; c->k2 = d[1]; c->p1 = d[2]; c->p2 = d[3]; c->k3 = d[4];
  c->affine = (c->ir[6] == 0.0 && c->ir[7] == 0.0 && c->ir[8] == 1.0) ? 1 : 0;
  return IPA_OK;
}

// OpenCV interpolateLanczos4 evaluated at k/32, k = 0..31 (the rows of cv2's
// Lanczos4 interpolation table)
static void lanczos4_row(float x, float* coeffs) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0},  {-s45, -s45}, {0, 1},  {s45, -s45},
                                 {-1, 0}, {s45, s45},   {0, -1}, {-s45, s45}};
  if (x < 1.1920929e-07f) {
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

// OpenCV interpolateCubic (A = -0.75) in float32, the rows of cv2's bicubic table
static void cubic_row_f32(float x, float* c) {
#pragma clang fp contract(off)
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

static std::mutex g_lz_mu;
static float* g_lanczos_dev[64] = {nullptr};

int ipa_lanczos_table(ipa_ctx* ctx, const float** out) {
  std::lock_guard<std::mutex> lk(g_lz_mu);
  int dev = ctx->device;
  IPA_REQUIRE(ctx, dev >= 0 && dev < 64, "device id out of range");
  if (!g_lanczos_dev[dev]) {
    // [0, 256): Lanczos4 rows; [256, 384): bicubic rows (the uint8 fixed-point path)
    float tab[32 * 8 + 32 * 4];
    for (int k = 0; k < 32; k++) lanczos4_row((float)k / 32.f, tab + k * 8);
    for (int k = 0; k < 32; k++) cubic_row_f32((float)k * (1.f / 32), tab + 256 + k * 4);
    float* d = nullptr;
    IPA_HIP(ctx, hipSetDevice(dev));
    IPA_HIP(ctx, hipMalloc((void**)&d, sizeof(tab)));
    IPA_HIP(ctx, hipMemcpy(d, tab, sizeof(tab), hipMemcpyHostToDevice));
    g_lanczos_dev[dev] = d;
  }
  *out = g_lanczos_dev[dev];
  return IPA_OK;
}

// OpenCV's 8U bicubic weights as a table (imgwarp.cpp initInterTab2D, fixpt): per fraction pair
// (fy, fx) the 4 x 4 shorts saturate_cast<short>(wy[k1] * wx[k2] * 2^15), their sum forced to 2^15
// on one entry of the 2 x 2 block at (2, 2).  Device layout: row fy * 32 + fx = 8 dwords, per tap
// row {w0 | w2 << 16, w1 | w3 << 16} - the operands of v_dot2_i32_i16 against the tap bytes
// (b0, b2) and (b1, b3).  32 KB: remap_kernel keeps it in LDS.
static int sat_short_f(float v) {
  double r = nearbyint((double)v);  // cvRound: half to even
  if (r < -32768.0) r = -32768.0;
  if (r > 32767.0) r = 32767.0;
  return (int)r;
}

static int* g_cubic2d_dev[64] = {};

int ipa_u8_cubic_tab2d(ipa_ctx* ctx, const int** out) {
#pragma clang fp contract(off)
  std::lock_guard<std::mutex> lk(g_lz_mu);
  const int dev = ctx->device;
  IPA_REQUIRE(ctx, dev >= 0 && dev < 64, "device id out of range");
  if (!g_cubic2d_dev[dev]) {
    float t1[32][4];
    for (int k = 0; k < 32; k++) cubic_row_f32((float)k * (1.f / 32), t1[k]);
    static int packed[1024 * 8];
    for (int fy = 0; fy < 32; fy++)
      for (int fx = 0; fx < 32; fx++) {
        int itab[16], isum = 0;
        for (int k1 = 0; k1 < 4; k1++) {
          const float vy = t1[fy][k1];
          for (int k2 = 0; k2 < 4; k2++) {
            const float v = vy * t1[fx][k2];
            isum += itab[k1 * 4 + k2] = sat_short_f(v * 32768.f);
          }
        }
        if (isum != 32768) {
          const int diff = isum - 32768;
          int Mk1 = 2, Mk2 = 2, mk1 = 2, mk2 = 2;
          for (int k1 = 2; k1 < 4; k1++)
            for (int k2 = 2; k2 < 4; k2++) {
              if (itab[k1 * 4 + k2] < itab[mk1 * 4 + mk2]) { mk1 = k1; mk2 = k2; }
              else if (itab[k1 * 4 + k2] > itab[Mk1 * 4 + Mk2]) { Mk1 = k1; Mk2 = k2; }
            }
          if (diff < 0) itab[Mk1 * 4 + Mk2] = (short)(itab[Mk1 * 4 + Mk2] - diff);
          else itab[mk1 * 4 + mk2] = (short)(itab[mk1 * 4 + mk2] - diff);
        }
        int* row = packed + (fy * 32 + fx) * 8;
        for (int r = 0; r < 4; r++) {
          const int* w = itab + r * 4;
          row[r * 2 + 0] = (w[0] & 0xffff) | (int)((unsigned)w[2] << 16);
          row[r * 2 + 1] = (w[1] & 0xffff) | (int)((unsigned)w[3] << 16);
        }
      }
    int* d = nullptr;
    IPA_HIP(ctx, hipSetDevice(dev));
    IPA_HIP(ctx, hipMalloc((void**)&d, sizeof(packed)));
    IPA_HIP(ctx, hipMemcpy(d, packed, sizeof(packed), hipMemcpyHostToDevice));
    g_cubic2d_dev[dev] = d;
  }
  *out = g_cubic2d_dev[dev];
  return IPA_OK;
}

// The same table for Lanczos4: per fraction pair 8 x 8 shorts = 32 dwords, per tap row
// {w0 | w2 << 16, w1 | w3 << 16, w4 | w6 << 16, w5 | w7 << 16}.  128 KB: remap_u8_lz_kernel keeps
// it in the LDS of a 1024-thread workgroup.
static int* g_lz2d_dev[64] = {};

int ipa_u8_lanczos_tab2d(ipa_ctx* ctx, const int** out) {
#pragma clang fp contract(off)
  std::lock_guard<std::mutex> lk(g_lz_mu);
  const int dev = ctx->device;
  IPA_REQUIRE(ctx, dev >= 0 && dev < 64, "device id out of range");
  if (!g_lz2d_dev[dev]) {
    float t1[32][8];
    for (int k = 0; k < 32; k++) lanczos4_row((float)k * (1.f / 32), t1[k]);
    static int packed[1024 * 32];
    for (int fy = 0; fy < 32; fy++)
      for (int fx = 0; fx < 32; fx++) {
        int itab[64], isum = 0;
        for (int k1 = 0; k1 < 8; k1++) {
          const float vy = t1[fy][k1];
          for (int k2 = 0; k2 < 8; k2++) {
            const float v = vy * t1[fx][k2];
            isum += itab[k1 * 8 + k2] = sat_short_f(v * 32768.f);
          }
        }
        if (isum != 32768) {
          const int diff = isum - 32768;
          int Mk1 = 4, Mk2 = 4, mk1 = 4, mk2 = 4;
          for (int k1 = 4; k1 < 6; k1++)
            for (int k2 = 4; k2 < 6; k2++) {
              if (itab[k1 * 8 + k2] < itab[mk1 * 8 + mk2]) { mk1 = k1; mk2 = k2; }
              else if (itab[k1 * 8 + k2] > itab[Mk1 * 8 + Mk2]) { Mk1 = k1; Mk2 = k2; }
            }
          if (diff < 0) itab[Mk1 * 8 + Mk2] = (short)(itab[Mk1 * 8 + Mk2] - diff);
          else itab[mk1 * 8 + mk2] = (short)(itab[mk1 * 8 + mk2] - diff);
        }
        int* row = packed + (fy * 32 + fx) * 32;
        for (int r = 0; r < 8; r++)
          for (int q = 0; q < 2; q++) {
            const int* w = itab + r * 8 + 4 * q;
            row[r * 4 + q * 2 + 0] = (w[0] & 0xffff) | (int)((unsigned)w[1] << 16);
            row[r * 4 + q * 2 + 1] = (w[2] & 0xffff) | (int)((unsigned)w[3] << 16);
          }
      }
    int* d = nullptr;
    IPA_HIP(ctx, hipSetDevice(dev));
    IPA_HIP(ctx, hipMalloc((void**)&d, sizeof(packed)));
    IPA_HIP(ctx, hipMemcpy(d, packed, sizeof(packed), hipMemcpyHostToDevice));
    g_lz2d_dev[dev] = d;
  }
  *out = g_lz2d_dev[dev];
  return IPA_OK;
}

int ipa_check_interp_border(ipa_ctx* ctx, int interp, int border) {
  int base = interp & 0xff;
  IPA_REQUIRE(ctx, (interp & ~(0xff | IPA_INTER_Q5)) == 0, "unknown interpolation flags 0x%x",
              interp);
  IPA_REQUIRE(ctx,
              base == IPA_INTER_NEAREST || base == IPA_INTER_LINEAR ||
                  base == IPA_INTER_CUBIC_CV || base == IPA_INTER_LANCZOS4 ||
                  base == IPA_INTER_CUBIC_KEYS,
              "unknown interpolation %d", base);
  IPA_REQUIRE(ctx, border >= IPA_BORDER_CONSTANT && border <= IPA_BORDER_REFLECT101,
              "unknown border mode %d", border);
  return IPA_OK;
}

// host-pointer staging shared by the three remap flavours
struct Staged {
  char* d_src; char* d_dst; float* d_mx; float* d_my;
  size_t src_bytes, dst_bytes, map_bytes;
};

static int stage_in(ipa_ctx* ctx, const void* src, int src_dt, int sh, int sw, int dst_dt, int dh,
                    int dw, int n_frames, const float* mapx, const float* mapy, Staged* st) {
  IPA_REQUIRE(ctx, src, "null source");
  IPA_REQUIRE(ctx, sh > 0 && sw > 0 && dh > 0 && dw > 0 && n_frames >= 1, "bad shape");
  size_t ss = ipa_dtype_size(src_dt), ds = ipa_dtype_size(dst_dt);
  IPA_REQUIRE(ctx, ss && ds, "unknown dtype");
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  st->src_bytes = (size_t)sh * sw * ss * n_frames;
  st->dst_bytes = (size_t)dh * dw * ds * n_frames;
  st->map_bytes = mapx ? (size_t)dh * dw * 4 : 0;
  size_t total = up(st->src_bytes) + up(st->dst_bytes) + 2 * up(st->map_bytes);
  int rc = ipa_ws_reserve(ctx, total);
  if (rc) return rc;
  char* b = (char*)ctx->ws;
  st->d_src = b; b += up(st->src_bytes);
  st->d_dst = b; b += up(st->dst_bytes);
  st->d_mx = (float*)b; b += up(st->map_bytes);
  st->d_my = (float*)b;
  IPA_HIP(ctx, hipMemcpyAsync(st->d_src, src, st->src_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (mapx) {
    IPA_HIP(ctx, hipMemcpyAsync(st->d_mx, mapx, st->map_bytes, hipMemcpyHostToDevice, ctx->stream));
    IPA_HIP(ctx, hipMemcpyAsync(st->d_my, mapy, st->map_bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  return IPA_OK;
}

static int stage_out(ipa_ctx* ctx, void* dst, const Staged& st) {
  IPA_HIP(ctx, hipMemcpyAsync(dst, st.d_dst, st.dst_bytes, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

// the float32 maps of a lens model, kept in the context until another model or size is asked for
int ipa_lens_map_cached(ipa_ctx* ctx, const double* K, const double* dist5, const double* newK,
                        int h, int w, float** mx, float** my) {
  double key[25];
  for (int i = 0; i < 9; i++) key[i] = K[i];
  for (int i = 0; i < 5; i++) key[9 + i] = dist5[i];
  for (int i = 0; i < 9; i++) key[14 + i] = newK[i];
  key[23] = (double)h;
  key[24] = (double)w;
  const size_t mb = ((size_t)h * w * 4 + 255) & ~(size_t)255;
  const bool hit = ctx->lens_key_n == 25 && memcmp(ctx->lens_key, key, sizeof(key)) == 0;
  if (!hit) {
    ctx->lens_key_n = 0;
    if (ctx->lens_map_bytes < 2 * mb) {
      IPA_HIP(ctx, hipSetDevice(ctx->device));
      IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // earlier calls may still read the old maps
      if (ctx->lens_map) {
        IPA_HIP(ctx, hipFree(ctx->lens_map));
        ctx->lens_map = nullptr;
        ctx->lens_map_bytes = 0;
      }
      IPA_HIP(ctx, hipMalloc(&ctx->lens_map, 2 * mb));
      ctx->lens_map_bytes = 2 * mb;
    }
    int rc = ipa_build_undistort_map_dev(ctx, K, dist5, newK, h, w, (float*)ctx->lens_map,
                                         (float*)((char*)ctx->lens_map + mb), w);
    if (rc) return rc;
    memcpy(ctx->lens_key, key, sizeof(key));
    ctx->lens_key_n = 25;
  }
  *mx = (float*)ctx->lens_map;
  *my = (float*)((char*)ctx->lens_map + mb);
  return IPA_OK;
}

extern "C" {

int ipa_build_undistort_map_dev(ipa_ctx* ctx, const double* K, const double* dist5,
                                const double* newK, int h, int w, float* d_mapx, float* d_mapy,
                                long map_pitch) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_mapx && d_mapy && h > 0 && w > 0 && map_pitch >= w, "bad map arguments");
  UndistortCoord c;
  int rc = make_undistort_coord(ctx, K, dist5, newK, &c);
  if (rc) return rc;
  int vec = aligned_rows(d_mapx, map_pitch, 0, 1, 4, IPA_VEC_ALIGN) &&
            aligned_rows(d_mapy, map_pitch, 0, 1, 4, IPA_VEC_ALIGN);
  dim3 grid((w + 255) / 256, (h + 3) / 4), block(64, 4);
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(build_map_kernel, grid, block, 0, ctx->stream, c, h, w, d_mapx, d_mapy,
                     map_pitch, vec);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

int ipa_build_undistort_map(ipa_ctx* ctx, const double* K, const double* dist5,
                            const double* newK, int h, int w, float* mapx, float* mapy) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, mapx && mapy && h > 0 && w > 0, "bad map arguments");
  size_t mb = ((size_t)h * w * 4 + 255) & ~(size_t)255;
  int rc = ipa_ws_reserve(ctx, 2 * mb);
  if (rc) return rc;
  float* dx = (float*)ctx->ws;
  float* dy = (float*)((char*)ctx->ws + mb);
  rc = ipa_build_undistort_map_dev(ctx, K, dist5, newK, h, w, dx, dy, w);
  if (rc) return rc;
  IPA_HIP(ctx, hipMemcpyAsync(mapx, dx, (size_t)h * w * 4, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipMemcpyAsync(mapy, dy, (size_t)h * w * 4, hipMemcpyDeviceToHost, ctx->stream));
  IPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IPA_OK;
}

// Round 6 (knob strip_remap): bilinear remaps of uint16 frames INTO float32 - camera frames as
// transformations.toFloatArray ingests them (transformations.py:78-87), the element types of BASELINE C4 - run on
// the marching strips of the fused chains with NO filter (wave_sep_kernel, K = 1: 256-px strips, no halo, both
// passes the identity; fused_sep_c.hip): the gather kernels these calls took stream 64 x 4K in 1.28 ms, the strips in
// 0.90 (maps; -30 %), lens model 1.42 -> 0.89, homography 1.16 -> 1.07; identical bits (tools/strip_remap_probe.py).
// Batches of a multiple of 4 frames (the shared-footprint loop; from 7 frames on any count: whole workgroups + the last
// four frames again) for maps and homographies that do not turn the picture; the lens model by value at any count (its
// map is evaluated once and cached).  uint8 frames (8-bit cameras) the same with maps: 1.28 -> 0.93 ms.  float32 frames
// stay where they are: the tile kernel is level with the strips on maps and 15 - 19 % faster on homographies.
// (The knob sep_u16 is required too: with it off the chain entry would come back here through its two-launch form.)
static bool strip_remap_takes(const ipa_ctx* ctx, const void* d_src, const void* d_dst, int src_dtype, int dst_dtype,
                              int sh, int sw, long src_pitch, int dh, int dw, long dst_pitch, int n_frames,
                              int interp, bool maps = false) {   // (maps: uint8 frames are built with the map pair only)
  const ipa_tuning& t = ctx->tune;
  if (!t.strip_remap || !t.sep_u16 || !t.frames_wg || !t.frames_inner || !t.pipe) return false;
  if ((src_dtype != IPA_U16 && !(src_dtype == IPA_U8 && maps)) || dst_dtype != IPA_F32 || !d_src || !d_dst) return false;
  if ((interp & 0xff) != IPA_INTER_LINEAR || (interp & ~(0xff | IPA_INTER_Q5)) != 0) return false;
  // (what the chain kernels' 32-bit offsets hold; anything else stays with the gather kernels and their checks)
  if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || src_pitch < sw || dst_pitch < dw || src_pitch >= (1l << 23)) return false;
  if (((size_t)(sh - 1) * src_pitch + sw) * ipa_dtype_size(src_dtype) >= (1ull << 31) || n_frames < 1 || n_frames > 65535) return false;
  return true;
}
// how far the source row moves along one output row (px per px), at 9 points of the picture
static double warp_row_drift(const double* m, int dh, int dw) {
  double drift = 0;
  for (int py = 0; py < 3; py++)
    for (int px = 0; px < 3; px++) {
      const double u = (dw - 2) * 0.5 * px, v = (dh - 2) * 0.5 * py;
      double y[2];
      for (int i = 0; i < 2; i++) {
        const double W = m[6] * (u + i) + m[7] * v + m[8];
        y[i] = (m[3] * (u + i) + m[4] * v + m[5]) * (W != 0.0 ? 1.0 / W : 0.0);
      }
      if (!(fabs(y[1] - y[0]) < 1e6)) return 1e6;
      drift = fabs(y[1] - y[0]) > drift ? fabs(y[1] - y[0]) : drift;
    }
  return drift;
}
static const double kOneTap = 1.0;

int ipa_remap_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw, long src_pitch,
                  const float* d_mapx, const float* d_mapy, long map_pitch, void* d_dst,
                  int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                  long src_frame_stride, long dst_frame_stride, int interp, int border_mode,
                  double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_mapx && d_mapy && map_pitch >= dw, "bad map arguments");
  if ((src_dtype == IPA_U16 || src_dtype == IPA_U8) && dst_dtype == src_dtype) {
    // ... and INTO the frames' own integer type with cv2's arithmetic (what LensDistortion.correct returns for camera
    // frames): the same strips, the blend of sampler.hpp::sample_u16_cv / sample_u8_fixed
    // (fused.hip::ipa_strip_remap_int; 1: not a call it covers)
    int rc = ipa_strip_remap_int(ctx, src_dtype, d_src, sh, sw, src_pitch, d_mapx, d_mapy, map_pitch, d_dst, dh, dw,
                                 dst_pitch, n_frames, src_frame_stride, dst_frame_stride, interp, border_mode,
                                 border_value);
    if (rc <= 0) return rc;
  }
  // (a multiple of 4 frames, or from 7 on: the chain then runs whole workgroups + the last four frames again)
  if ((n_frames % 4 == 0 || n_frames >= 7) && strip_remap_takes(ctx, d_src, d_dst, src_dtype, dst_dtype, sh, sw, src_pitch,
                                                                dh, dw, dst_pitch, n_frames, interp, true)) {
    ctx->strip_remaps++;
    return ipa_remap_sepconv2d_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, d_mapx, d_mapy, map_pitch, &kOneTap, 1,
                                   &kOneTap, 1, d_dst, dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                                   dst_frame_stride, interp, border_mode, border_value, IPA_BORDER_REFLECT,
                                   IPA_BORDER_REFLECT);
  }
  MapCoord c{d_mapx, d_mapy, map_pitch};
  int map_vec = aligned_rows(d_mapx, map_pitch, 0, 1, 4, IPA_VEC_ALIGN) &&
            aligned_rows(d_mapy, map_pitch, 0, 1, 4, IPA_VEC_ALIGN);
  RemapCall a{d_src, src_dtype, sh, sw, src_pitch, d_dst, dst_dtype, dh, dw, dst_pitch,
              n_frames, src_frame_stride, dst_frame_stride, interp, border_mode, border_value};
  return ipa_remap_launch_map(ctx, a, c, map_vec);
}

int ipa_undistort_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                      long src_pitch, const double* K, const double* dist5, const double* newK,
                      void* d_dst, int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                      long src_frame_stride, long dst_frame_stride, int interp, int border_mode,
                      double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  UndistortCoord c;
  int rc = make_undistort_coord(ctx, K, dist5, newK, &c);
  if (rc) return rc;
  if (ctx->tune.lens_cache && strip_remap_takes(ctx, d_src, d_dst, src_dtype, dst_dtype, sh, sw, src_pitch, dh, dw,
                                                dst_pitch, n_frames, interp)) {   // (see ipa_remap_dev)
    ctx->strip_remaps++;
    return ipa_undistort_sepconv2d_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, K, dist5, newK, &kOneTap, 1, &kOneTap,
                                       1, d_dst, dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                                       dst_frame_stride, interp, border_mode, border_value, IPA_BORDER_REFLECT,
                                       IPA_BORDER_REFLECT);
  }
  // integer frames, batches: the model's float32 coordinates are the same for every frame - through the cached
  // map (bit for bit what the per-pixel evaluation gives; the chains have used it since round 2) instead of
  // evaluating the model per pixel and frame: 64 x 4K uint16 -> uint16 1.77 -> 1.31 ms (float32 frames: level, stay)
  if (ctx->tune.lens_cache && n_frames >= 4 && (src_dtype == IPA_U8 || src_dtype == IPA_U16) && dh > 0 && dw > 0) {
    float *mx = nullptr, *my = nullptr;
    rc = ipa_lens_map_cached(ctx, K, dist5, newK, dh, dw, &mx, &my);
    if (rc) return rc;
    return ipa_remap_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, mx, my, dw, d_dst, dst_dtype, dh, dw, dst_pitch,
                         n_frames, src_frame_stride, dst_frame_stride, interp, border_mode, border_value);
  }
  RemapCall a{d_src, src_dtype, sh, sw, src_pitch, d_dst, dst_dtype, dh, dw, dst_pitch,
              n_frames, src_frame_stride, dst_frame_stride, interp, border_mode, border_value};
  return ipa_remap_launch_undistort(ctx, a, c);
}

int ipa_warp_perspective_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                             long src_pitch, const double* M, void* d_dst, int dst_dtype, int dh,
                             int dw, long dst_pitch, int n_frames, long src_frame_stride,
                             long dst_frame_stride, int interp, int border_mode,
                             double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, M, "null matrix");
  if ((n_frames % 4 == 0 || n_frames >= 7) && strip_remap_takes(ctx, d_src, d_dst, src_dtype, dst_dtype, sh, sw, src_pitch,
                                                                dh, dw, dst_pitch, n_frames, interp) &&
      warp_row_drift(M, dh, dw) < 0.2) {   // (see ipa_remap_dev; pictures that turn stay with the gather kernels)
    ctx->strip_remaps++;
    return ipa_warp_perspective_sepconv2d_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, M, &kOneTap, 1, &kOneTap, 1,
                                              d_dst, dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                                              dst_frame_stride, interp, border_mode, border_value,
                                              IPA_BORDER_REFLECT, IPA_BORDER_REFLECT);
  }
  HomographyCoord c;
  for (int i = 0; i < 9; i++) c.m[i] = M[i];
  RemapCall a{d_src, src_dtype, sh, sw, src_pitch, d_dst, dst_dtype, dh, dw, dst_pitch,
              n_frames, src_frame_stride, dst_frame_stride, interp, border_mode, border_value};
  return ipa_remap_launch_homography(ctx, a, c);
}

int ipa_remap(ipa_ctx* ctx, const void* src, int src_dtype, int sh, int sw, const float* mapx,
              const float* mapy, void* dst, int dst_dtype, int dh, int dw, int n_frames,
              int interp, int border_mode, double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, mapx && mapy && dst, "null pointer");
  Staged st;
  int rc = stage_in(ctx, src, src_dtype, sh, sw, dst_dtype, dh, dw, n_frames, mapx, mapy, &st);
  if (rc) return rc;
  rc = ipa_remap_dev(ctx, st.d_src, src_dtype, sh, sw, sw, st.d_mx, st.d_my, dw, st.d_dst,
                     dst_dtype, dh, dw, dw, n_frames, (long)sh * sw, (long)dh * dw, interp,
                     border_mode, border_value);
  if (rc) return rc;
  return stage_out(ctx, dst, st);
}

int ipa_undistort(ipa_ctx* ctx, const void* src, int src_dtype, int sh, int sw, const double* K,
                  const double* dist5, const double* newK, void* dst, int dst_dtype, int dh,
                  int dw, int n_frames, int interp, int border_mode, double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, dst, "null pointer");
  Staged st;
  int rc = stage_in(ctx, src, src_dtype, sh, sw, dst_dtype, dh, dw, n_frames, nullptr, nullptr, &st);
  if (rc) return rc;
  rc = ipa_undistort_dev(ctx, st.d_src, src_dtype, sh, sw, sw, K, dist5, newK, st.d_dst, dst_dtype,
                         dh, dw, dw, n_frames, (long)sh * sw, (long)dh * dw, interp, border_mode,
                         border_value);
  if (rc) return rc;
  return stage_out(ctx, dst, st);
}

int ipa_warp_perspective(ipa_ctx* ctx, const void* src, int src_dtype, int sh, int sw,
                         const double* M, void* dst, int dst_dtype, int dh, int dw, int n_frames,
                         int interp, int border_mode, double border_value) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, dst, "null pointer");
  Staged st;
  int rc = stage_in(ctx, src, src_dtype, sh, sw, dst_dtype, dh, dw, n_frames, nullptr, nullptr, &st);
  if (rc) return rc;
  rc = ipa_warp_perspective_dev(ctx, st.d_src, src_dtype, sh, sw, sw, M, st.d_dst, dst_dtype, dh,
                                dw, dw, n_frames, (long)sh * sw, (long)dh * dw, interp,
                                border_mode, border_value);
  if (rc) return rc;
  return stage_out(ctx, dst, st);
}

}  // extern "C"
