// fused.hip — C-ABI entry points of the fused remap -> K x K filter chain
// (kernels: fused_impl.hpp, one translation unit per K).
#include "fused_sep_impl.hpp"

using namespace ipa;

void ipa_fused_sep_launch_a(ipa_ctx*, const FusedCall&, const FusedSep&);  // 3, 5 taps
void ipa_fused_sep_launch_b(ipa_ctx*, const FusedCall&, const FusedSep&);  // 7, 9 taps
void ipa_fused_sep_launch_c(ipa_ctx*, const FusedCall&, const FusedSep&);  // 1 tap: the remap alone
void ipa_fused_sep_launch_c16(ipa_ctx*, const FusedCall&);                   // ... uint16 into uint16 (cv2's 16U arithmetic)
void ipa_fused_sep_launch_c8(ipa_ctx*, const FusedCall&);                    // ... uint8 into uint8 (cv2's 8U fixed point)

int ipa_fused_launch_k3(ipa_ctx*, const FusedCall&);
int ipa_fused_launch_k5(ipa_ctx*, const FusedCall&);
int ipa_fused_launch_k7(ipa_ctx*, const FusedCall&);
int ipa_fused_big_launch(ipa_ctx*, const FusedCall&, int K);  // fused_big.hip; 1 = not covered
int ipa_check_interp_border(ipa_ctx* ctx, int interp, int border);  // remap.hip
#if IPA_WITH_TILE_CHAIN
// tile_chain.hip: 0 = launched, 1 = not a chain for that kernel
int ipa_tile_chain_launch(ipa_ctx* ctx, const void* d_src, int sh, int sw, long src_pitch, const double* M,
                          const double* ky, const double* kx, int K, void* d_dst, int dh, int dw,
                          long dst_pitch, int n_frames, long src_frame_stride, long dst_frame_stride,
                          int interp, int border_mode, double border_value, int cby, int cbx);
#endif

static int inv3f(const double* m, double* o) {
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

// Dense K x K kernels that are an outer product ky (x) kx - the bench's 5x5 is outer(g, g), and the reference itself
// obtains its Gaussians separably (scipy.ndimage.gaussian_filter: filters/standardDeviation.py:23,
// filters/fastFilter.py:42) - run on the separable K + K chain when that chain is ONE kernel for the call
// (fused_sep_common's one_kernel: float32 frames, bilinear taps, 3 / 5 / 7 / 9 taps).  64 x 4K maps + 5x5: K + K
// = 10 instead of K * K = 25 multiply-adds per pixel on the same strips.  Knob rank1_sep bit 0; the two loops
// differ by the order of a float32 sum only (both within 1e-5 of the oracle's double sum).
static bool rank1_chain(ipa_ctx* ctx, const double* kernel, int kh, int kw, int src_dtype, int dst_dtype,
                        int interp, double* ky, double* kx, bool maps = false, bool u8_maps = false) {
  if (!(ctx->tune.rank1_sep & 1) || !kernel || kh != kw) return false;
  if (!(kh == 3 || kh == 5 || kh == 7 || kh == 9)) return false;
  // (uint16 frames: where the separable chain is one kernel for them - maps, homographies; knob sep_u16)
  const bool src_ok = src_dtype == IPA_F32 || ((src_dtype == IPA_U16 || (src_dtype == IPA_U8 && u8_maps)) && maps && ctx->tune.sep_u16 != 0);
  if (!src_ok || dst_dtype != IPA_F32 || (interp & 0xff) != IPA_INTER_LINEAR) return false;
  return ipa_rank1_factor(kernel, kh, kw, ky, kx);
}

// K = 9, 11: map-based bilinear remaps of float32 frames run in one kernel (fused_big.hip);
// for the rest (bicubic, analytic coordinates, uint16 frames) the sampling source plus 9 / 11
// running rows exceed the VGPR budget that pays: the chain runs as two launches through the
// context workspace: remap kernel -> 9x9 / 11x11 filter.
// Round 6: so does every combination the standalone entry points accept and no fused kernel is
// built for - Lanczos4 / nearest taps, uint8 frames, uint16 frames with a homography or bicubic
// taps, rectangular or larger kernels (they returned IPA_ERR_UNSUPPORTED before): a caller of the
// chain gets what remap + filter give, in whatever number of launches.
static bool dense_chain_built(int src_dtype, int coord_kind, int interp, int kh, int kw) {
  const int base = interp & 0xff;
  if (kh != kw || !(kh == 3 || kh == 5 || kh == 7)) return false;
  if (src_dtype == IPA_F32)
    return base == IPA_INTER_LINEAR || base == IPA_INTER_CUBIC_CV || base == IPA_INTER_CUBIC_KEYS;
  if (src_dtype == IPA_U16) return base == IPA_INTER_LINEAR && coord_kind != 2;
  if (src_dtype == IPA_U8) return base == IPA_INTER_LINEAR && coord_kind == 0;   // (8-bit camera frames, maps)
  return false;
}
static int big_kernel_tmp(ipa_ctx* ctx, int src_dtype, int coord_kind, int interp, int kh, int kw, int dst_dtype,
                          int dh, int dw, int n_frames, void** tmp) {
  const bool big = kh == kw && (kh == 9 || kh == 11);
  if (!big) {
    if (dense_chain_built(src_dtype, coord_kind, interp, kh, kw)) return 1;   // one kernel
    // (anything the two launches would reject themselves goes on to the fused path's own checks)
    if (dst_dtype != IPA_F32 || dh <= 0 || dw <= 0 || n_frames < 1 || kh < 1 || kw < 1) return 1;
  }
  IPA_REQUIRE(ctx, dst_dtype == IPA_F32, "fused remap+filter writes float32");
  int rc = ipa_ws_reserve(ctx, (size_t)n_frames * dh * dw * 4);
  if (rc) return rc;
  *tmp = ctx->ws;
  return IPA_OK;
}

// validation + everything of a FusedCall that does not depend on the filter
static int fused_fill(ipa_ctx* ctx, FusedCall& f, const void* d_src, int src_dtype, int sh, int sw,
                      long src_pitch, void* d_dst, int dst_dtype, int dh, int dw, long dst_pitch,
                      int n_frames, long src_frame_stride, long dst_frame_stride, int interp,
                      int border_mode, double border_value, int cbx, int cby) {
  IPA_REQUIRE(ctx, d_src && d_dst, "null pointer");
  IPA_REQUIRE(ctx, sh > 0 && sw > 0 && dh > 0 && dw > 0, "empty image");
  IPA_REQUIRE(ctx, src_pitch >= sw && dst_pitch >= dw, "pitch smaller than width");
  IPA_REQUIRE(ctx, src_pitch < (1l << 23), "source pitch must be below 2^23 elements");  // mul24
  IPA_REQUIRE(ctx, n_frames >= 1 && n_frames <= 65535, "n_frames must be in [1,65535]");
  int rc = ipa_check_interp_border(ctx, interp, border_mode);
  if (rc) return rc;
  IPA_REQUIRE(ctx, cbx >= 0 && cbx <= IPA_BORDER_REFLECT101 && cby >= 0 && cby <= IPA_BORDER_REFLECT101,
              "unknown filter border mode");
  size_t ss = ipa_dtype_size(src_dtype), ds = ipa_dtype_size(dst_dtype);
  IPA_REQUIRE(ctx, ss && ds, "unknown dtype");
  size_t frame_bytes = ((size_t)(sh - 1) * src_pitch + sw) * ss;
  IPA_REQUIRE(ctx, frame_bytes < (1ull << 31), "source frame too large for 32-bit offsets");
  int base = interp & 0xff;
  WaveParams& p = f.p;
  p.dst = (char*)d_dst;
  p.dst_frame_elems = dst_frame_stride;
  p.dh = dh; p.dw = dw; p.dpitch = dst_pitch;
  p.cbx = cbx; p.cby = cby;
  p.vec_out = (((uintptr_t)d_dst) % IPA_VEC_ALIGN == 0) && ((dst_pitch * (long)ds) % IPA_VEC_ALIGN == 0) &&
              (n_frames == 1 || (dst_frame_stride * (long)ds) % IPA_VEC_ALIGN == 0);
  f.src = (const char*)d_src;
  f.src_frame_bytes = src_frame_stride * (long)ss;
  f.src_bytes = (unsigned)frame_bytes;
  f.sh = sh; f.sw = sw; f.spitch = (int)src_pitch;
  f.border = border_mode; f.q5 = (interp & IPA_INTER_Q5) ? 1 : 0;
  f.cubic_a = base == IPA_INTER_CUBIC_KEYS ? -0.5f : -0.75f;
  f.cval = border_value;
  f.conv_cval = 0.0;
  if (f.coord_kind == 0)
    f.map_vec = (((uintptr_t)f.map.mx) % IPA_VEC_ALIGN == 0) && (((uintptr_t)f.map.my) % IPA_VEC_ALIGN == 0) &&
                ((f.map.pitch * 4) % IPA_VEC_ALIGN == 0);
  else
    f.map_vec = 0;
  f.src_dt = src_dtype; f.dst_dt = dst_dtype; f.interp_base = base; f.n_frames = n_frames;
  f.kernel = nullptr;
  return IPA_OK;
}

static int fused_common(ipa_ctx* ctx, FusedCall& f, const void* d_src, int src_dtype, int sh,
                        int sw, long src_pitch, const double* kernel, int kh, int kw, void* d_dst,
                        int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                        long src_frame_stride, long dst_frame_stride, int interp, int border_mode,
                        double border_value, int cbx, int cby) {
  IPA_REQUIRE(ctx, kernel, "null pointer");
  if (kh != kw || !(kh == 3 || kh == 5 || kh == 7 || kh == 9 || kh == 11))
    IPA_UNSUPPORTED(ctx, "fused remap+filter is built for square 3/5/7/9/11 kernels (got %dx%d); "
                         "use ipa_remap_dev + ipa_conv2d_dev", kh, kw);
  int rc = fused_fill(ctx, f, d_src, src_dtype, sh, sw, src_pitch, d_dst, dst_dtype, dh, dw,
                      dst_pitch, n_frames, src_frame_stride, dst_frame_stride, interp, border_mode,
                      border_value, cbx, cby);
  if (rc) return rc;
  f.kernel = kernel;
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  switch (kh) {
    case 3: rc = ipa_fused_launch_k3(ctx, f); break;
    case 5: rc = ipa_fused_launch_k5(ctx, f); break;
    case 7: rc = ipa_fused_launch_k7(ctx, f); break;
    default: rc = ipa_fused_launch_k7(ctx, f); break;
  }
  if (rc) return rc;
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

// remap -> separable filter.  One kernel where wave_sep_kernel covers the case; otherwise
// `two(tmp)` materialises the remap in the context workspace and the plain separable filter
// follows (same results: the fused kernel rounds the remapped row to float32 as well).
template <typename TwoLaunch>
static int fused_sep_common(ipa_ctx* ctx, FusedCall& f, TwoLaunch two, const void* d_src,
                            int src_dtype, int sh, int sw, long src_pitch, const double* ky, int nky,
                            const double* kx, int nkx, void* d_dst, int dst_dtype, int dh, int dw,
                            long dst_pitch, int n_frames, long src_frame_stride,
                            long dst_frame_stride, int interp, int border_mode, double border_value,
                            int cby, int cbx, bool prefer_two = false) {
  IPA_REQUIRE(ctx, ky && kx && nky > 0 && nkx > 0 && (nky & 1) && (nkx & 1),
              "ky / kx must be given with odd lengths");
  IPA_REQUIRE(ctx, dst_dtype == IPA_F32, "remap + separable filter writes float32");
  const int base = interp & 0xff;
  // bicubic: built and correct, but 16 taps per sample on the K-1 extra halo rows of every
  // strip make it slower than two launches (4K, 9 taps: 813 vs 694 us) -> two launches
  // (uint16 frames: with maps or a homography - f.coord_kind is set by the caller before it comes here)
  // (one tap - the remap alone - is built for uint16 frames: remap.hip::strip_remap_takes)
  // (uint8 frames: with maps only)
  const bool u8_maps = src_dtype == IPA_U8 && f.coord_kind == 0 && ctx->tune.sep_u16 != 0;
  const bool one_kernel = nky == nkx && (nky == 3 || nky == 5 || nky == 7 || nky == 9 || (nky == 1 && src_dtype != IPA_F32)) &&
                          (src_dtype == IPA_F32 || u8_maps || (src_dtype == IPA_U16 && f.coord_kind != 1 && ctx->tune.sep_u16 != 0)) &&
                          base == IPA_INTER_LINEAR && !prefer_two;
  if (!one_kernel) {
    IPA_REQUIRE(ctx, dh > 0 && dw > 0 && n_frames >= 1, "empty image");
    int rc = ipa_ws_reserve(ctx, (size_t)n_frames * dh * dw * 4);
    if (rc) return rc;
    rc = two(ctx->ws);
    if (rc) return rc;
    return ipa_sepconv2d_dev(ctx, ctx->ws, IPA_F32, dh, dw, dw, ky, nky, kx, nkx, d_dst, dst_pitch,
                             n_frames, (long)dh * dw, dst_frame_stride, cby, cbx, 0.0);
  }
  int rc = fused_fill(ctx, f, d_src, src_dtype, sh, sw, src_pitch, d_dst, dst_dtype, dh, dw,
                      dst_pitch, n_frames, src_frame_stride, dst_frame_stride, interp, border_mode,
                      border_value, cbx, cby);
  if (rc) return rc;
  FusedSep q{ky, kx, nky, 0.0f};
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  if (nky == 1) ipa_fused_sep_launch_c(ctx, f, q);
  else if (nky <= 5) ipa_fused_sep_launch_a(ctx, f, q);
  else ipa_fused_sep_launch_b(ctx, f, q);
  IPA_HIP(ctx, hipGetLastError());
  return IPA_OK;
}

// A homography whose output rows drift across source rows (a rotation of a few degrees) is where
// the fused kernels lose: their gathers pay per cache line a wave touches (16 x 4K, perspective
// warp + separable 9+9: 0.33 ms at no rotation, 0.47 at 7 degrees, 0.72 at 15, 1.52 at 45).  For batches
// the tile warp kernel takes (remap_impl.hpp: tile_warp_pays) the chain then runs as two
// launches - tile warp into the workspace, filter - whose time hardly depends on the angle
// (0.52 - 0.66 ms); same results (the fused kernels round the remapped rows to float32 too).
static bool rotated_warp_in_two_launches(const ipa_ctx* ctx, const double* m, int src_dtype,
                                         int dst_dtype, int interp, int dh, int dw, int n_frames) {
  if (!ctx->tune.tile_warp || src_dtype != IPA_F32 || dst_dtype != IPA_F32) return false;
  if ((interp & 0xff) != IPA_INTER_LINEAR) return false;
  if (n_frames < 8 || (double)n_frames * dh * dw < 64e6) return false;
  auto at = [&](double u, double v, double& sx, double& sy) {
    const double W = m[6] * u + m[7] * v + m[8], iw = W != 0.0 ? 1.0 / W : 0.0;
    sx = (m[0] * u + m[1] * v + m[2]) * iw;
    sy = (m[3] * u + m[4] * v + m[5]) * iw;
  };
  double drift = 0;
  for (int py = 0; py < 3; py++)
    for (int px = 0; px < 3; px++) {
      const double u = (dw - 2) * 0.5 * px, v = (dh - 2) * 0.5 * py;
      double x0, y0, x1, y1;
      at(u, v, x0, y0);
      at(u + 1, v, x1, y1);
      if (!(fabs(y1 - y0) < 1e6)) return false;
      drift = fabs(y1 - y0) > drift ? fabs(y1 - y0) : drift;
    }
  return drift >= (ctx->tune.tile_warp > 1 ? 0.0 : 0.2);
}

// The strip remap of integer frames INTO their own type (remap.hip::ipa_remap_dev): cv2.remap's bilinear as it computes
// it on 16U (float32 product weights at 1/32-px coordinates) and 8U (15-bit fixed point) images - what
// LensDistortion.correct returns for camera frames - on the shared-record loop.  Returns 1 when the call is not one the
// loop covers on EVERY strip (the caller then takes the gather kernel), 0 when launched.
int ipa_strip_remap_int(ipa_ctx* ctx, int dtype, const void* d_src, int sh, int sw, long src_pitch, const float* d_mapx,
                        const float* d_mapy, long map_pitch, void* d_dst, int dh, int dw, long dst_pitch, int n_frames,
                        long src_frame_stride, long dst_frame_stride, int interp, int border_mode,
                        double border_value) {
  const ipa_tuning& t = ctx->tune;
  if (!t.strip_remap || !t.sep_u16 || !t.frames_wg || !t.frames_inner || !t.pipe) return 1;
  // uint16: cv2's arithmetic is what 'linear_cv_q5' names ('linear' = exact coordinates in double: the gather kernel);
  // uint8: every bilinear remap is cv2's fixed point
  if (dtype == IPA_U16 ? interp != (IPA_INTER_LINEAR | IPA_INTER_Q5)
                       : (dtype != IPA_U8 || (interp & 0xff) != IPA_INTER_LINEAR || (interp & ~(0xff | IPA_INTER_Q5)) != 0))
    return 1;
  // (counts that are no multiple of 4: from 7 frames on as a head of whole workgroups + the LAST four frames again -
  //  as the chains do, fused_impl.hpp::fused_split_tail; 5 and 6 frames stay with the gather kernel)
  if (n_frames < 4 || (n_frames % 4 != 0 && n_frames < 7) || n_frames > 65535) return 1;
  if (!d_src || !d_dst || !d_mapx || !d_mapy || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || (dw & 3) != 0) return 1;
  if (src_pitch < sw || dst_pitch < dw || map_pitch < dw || src_pitch >= (1l << 23)) return 1;
  if (((size_t)(sh - 1) * src_pitch + sw) * ipa_dtype_size(dtype) >= (1ull << 31)) return 1;
  if ((unsigned long)(((dw + 255) / 256) * ((dh + 15) / 16)) * (unsigned long)n_frames >= (1ul << 31)) return 1;
  FusedCall f;
  f.coord_kind = 0;
  f.map = MapCoord{d_mapx, d_mapy, map_pitch};
  int rc = fused_fill(ctx, f, d_src, dtype, sh, sw, src_pitch, d_dst, dtype, dh, dw, dst_pitch, n_frames,
                      src_frame_stride, dst_frame_stride, interp, border_mode, border_value, IPA_BORDER_REFLECT,
                      IPA_BORDER_REFLECT);
  if (rc) return rc;
  if (!f.p.vec_out || !f.map_vec) return 1;   // (rows of the result / of the maps that are no whole 16-byte vectors)
  // the border value as cv2 casts it: saturate_cast
  const double r = rint(border_value), top = dtype == IPA_U16 ? 65535.0 : 255.0;
  f.cval = r > 0 ? (r < top ? r : top) : 0;
  IPA_HIP(ctx, hipSetDevice(ctx->device));
  auto launch = [&](const FusedCall& g) {
    if (dtype == IPA_U16) ipa_fused_sep_launch_c16(ctx, g);
    else ipa_fused_sep_launch_c8(ctx, g);
  };
  if (n_frames % 4 == 0) {
    launch(f);
  } else {
    FusedCall head = f, tail = f;
    head.n_frames = n_frames - n_frames % 4;
    tail.n_frames = 4;
    tail.src = f.src + (long)(n_frames - 4) * f.src_frame_bytes;
    tail.p.dst = f.p.dst + (long)(n_frames - 4) * f.p.dst_frame_elems * (long)ipa_dtype_size(dtype);
    launch(head);
    launch(tail);
  }
  IPA_HIP(ctx, hipGetLastError());
  ctx->strip_remaps++;
  return IPA_OK;
}

extern "C" {

int ipa_remap_sepconv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                            long src_pitch, const float* d_mapx, const float* d_mapy,
                            long map_pitch, const double* ky, int nky, const double* kx, int nkx,
                            void* d_dst, int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                            long src_frame_stride, long dst_frame_stride, int interp,
                            int border_mode, double border_value, int conv_border_y,
                            int conv_border_x) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_mapx && d_mapy && map_pitch >= dw, "bad map arguments");
  FusedCall f;
  f.coord_kind = 0;
  f.map = MapCoord{d_mapx, d_mapy, map_pitch};
  auto two = [&](void* tmp) {
    return ipa_remap_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, d_mapx, d_mapy, map_pitch, tmp,
                         IPA_F32, dh, dw, dw, n_frames, src_frame_stride, (long)dh * dw, interp,
                         border_mode, border_value);
  };
  return fused_sep_common(ctx, f, two, d_src, src_dtype, sh, sw, src_pitch, ky, nky, kx, nkx, d_dst,
                          dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                          dst_frame_stride, interp, border_mode, border_value, conv_border_y,
                          conv_border_x);
}

int ipa_undistort_sepconv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                                long src_pitch, const double* K, const double* dist5,
                                const double* newK, const double* ky, int nky, const double* kx,
                                int nkx, void* d_dst, int dst_dtype, int dh, int dw, long dst_pitch,
                                int n_frames, long src_frame_stride, long dst_frame_stride,
                                int interp, int border_mode, double border_value, int conv_border_y,
                                int conv_border_x) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, K && dist5 && newK, "K, dist5 and newK must be given");
  if (ctx->tune.lens_cache) {
    float *mx = nullptr, *my = nullptr;
    int rc = ipa_lens_map_cached(ctx, K, dist5, newK, dh, dw, &mx, &my);
    if (rc) return rc;
    return ipa_remap_sepconv2d_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, mx, my, dw, ky, nky, kx,
                                   nkx, d_dst, dst_dtype, dh, dw, dst_pitch, n_frames,
                                   src_frame_stride, dst_frame_stride, interp, border_mode,
                                   border_value, conv_border_y, conv_border_x);
  }
  FusedCall f;
  f.coord_kind = 1;
  UndistortCoord& c = f.und;
  IPA_REQUIRE(ctx, inv3f(newK, c.ir) == 0, "newK is singular");
  c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
  c.k1 = dist5[0]; c.k2 = dist5[1]; c.p1 = dist5[2]; c.p2 = dist5[3]; c.k3 = dist5[4];
  c.affine = (c.ir[6] == 0.0 && c.ir[7] == 0.0 && c.ir[8] == 1.0) ? 1 : 0;
  auto two = [&](void* tmp) {
    return ipa_undistort_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, K, dist5, newK, tmp, IPA_F32,
                             dh, dw, dw, n_frames, src_frame_stride, (long)dh * dw, interp,
                             border_mode, border_value);
  };
  return fused_sep_common(ctx, f, two, d_src, src_dtype, sh, sw, src_pitch, ky, nky, kx, nkx, d_dst,
                          dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                          dst_frame_stride, interp, border_mode, border_value, conv_border_y,
                          conv_border_x);
}

int ipa_warp_perspective_sepconv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh,
                                       int sw, long src_pitch, const double* M, const double* ky,
                                       int nky, const double* kx, int nkx, void* d_dst,
                                       int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                                       long src_frame_stride, long dst_frame_stride, int interp,
                                       int border_mode, double border_value, int conv_border_y,
                                       int conv_border_x) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, M, "null matrix");
  FusedCall f;
  f.coord_kind = 2;
  for (int i = 0; i < 9; i++) f.hom.m[i] = M[i];
  auto two = [&](void* tmp) {
    return ipa_warp_perspective_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, M, tmp, IPA_F32, dh,
                                    dw, dw, n_frames, src_frame_stride, (long)dh * dw, interp,
                                    border_mode, border_value);
  };
  const bool rotated = dh > 0 && dw > 0 &&
                       rotated_warp_in_two_launches(ctx, M, src_dtype, dst_dtype, interp, dh, dw, n_frames);
#if IPA_WITH_TILE_CHAIN   // experiment builds only (tools/tile_chain): the one-launch chain, slower than the two launches
  // knob tile_chain = 1: the chains that take two launches - bicubic warps, bilinear warps that rotate
  // the picture - in ONE launch on the tile skeleton (tile_chain.hpp).  Built for the review of round
  // 4, bit-identical, and slower than the two launches (16 x 4K + 9 + 9: bicubic 0.80 against 0.55 ms,
  // rotated bilinear 0.87 against 0.56): the warp kernel is bound by its vector work, not by the 8 B/px
  // of workspace traffic the fusion saves, and the filter passes join it in the same waves instead of
  // running at stream rate in a kernel of their own.  Off by default; only calls the two launches
  // would accept go there (anything else falls through to their checks).
  {
    const int base = interp & 0xff;
    const bool cubic = base == IPA_INTER_CUBIC_CV || base == IPA_INTER_CUBIC_KEYS;
    auto mode_ok = [](int b) { return b >= IPA_BORDER_CONSTANT && b <= IPA_BORDER_REFLECT101; };
    const bool valid = d_src && d_dst && d_src != d_dst && ky && kx && nky == nkx && src_dtype == IPA_F32 &&
                       dst_dtype == IPA_F32 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && src_pitch >= sw &&
                       dst_pitch >= dw && src_pitch < (1l << 23) && n_frames >= 1 && n_frames <= 65535 &&
                       (interp & ~(0xff | IPA_INTER_Q5)) == 0 && mode_ok(border_mode) &&
                       mode_ok(conv_border_y) && mode_ok(conv_border_x);
    // (the two launches may write over their source - the warp has read it all by then; one launch may not)
    auto span = [](const void* p0, long frames, long stride, long pitch, int h, int w) {
      const char* lo = (const char*)p0;
      return std::pair<const char*, const char*>(lo, lo + ((frames - 1) * stride + (long)(h - 1) * pitch + w) * 4);
    };
    bool apart = false;
    if (valid && src_frame_stride >= 0 && dst_frame_stride >= 0) {
      const auto a = span(d_src, n_frames, src_frame_stride, src_pitch, sh, sw);
      const auto b = span(d_dst, n_frames, dst_frame_stride, dst_pitch, dh, dw);
      apart = a.second <= b.first || b.second <= a.first;
    }
    // (1: the chains that take two launches - bicubic, rotated bilinear; 2: every chain the kernel covers, upright
    // bilinear ones included, which the fused strip kernel already runs in one launch - the tests' value)
    if (ctx->tune.tile_chain && valid && apart &&
        (cubic || (base == IPA_INTER_LINEAR && (rotated || ctx->tune.tile_chain >= 2)))) {
      const int rc = ipa_tile_chain_launch(ctx, d_src, sh, sw, src_pitch, M, ky, kx, nky, d_dst, dh, dw,
                                           dst_pitch, n_frames, src_frame_stride, dst_frame_stride, interp,
                                           border_mode, border_value, conv_border_y, conv_border_x);
      if (rc < 0) return rc;
      if (rc == 0) {
        IPA_HIP(ctx, hipGetLastError());
        return IPA_OK;
      }
    }
  }
#endif
  return fused_sep_common(ctx, f, two, d_src, src_dtype, sh, sw, src_pitch, ky, nky, kx, nkx, d_dst,
                          dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                          dst_frame_stride, interp, border_mode, border_value, conv_border_y,
                          conv_border_x, rotated);
}

int ipa_remap_conv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                         long src_pitch, const float* d_mapx, const float* d_mapy, long map_pitch,
                         const double* kernel, int kh, int kw, void* d_dst, int dst_dtype, int dh,
                         int dw, long dst_pitch, int n_frames, long src_frame_stride,
                         long dst_frame_stride, int interp, int border_mode, double border_value,
                         int conv_border_x, int conv_border_y) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, d_mapx && d_mapy && map_pitch >= dw, "bad map arguments");
  {
    double ky[9], kx[9];
    if (rank1_chain(ctx, kernel, kh, kw, src_dtype, dst_dtype, interp, ky, kx, true, true)) {
      ctx->rank1_routed++;
      return ipa_remap_sepconv2d_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, d_mapx, d_mapy, map_pitch, ky, kh,
                                     kx, kw, d_dst, dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                                     dst_frame_stride, interp, border_mode, border_value, conv_border_y,
                                     conv_border_x);
    }
  }
  // 9x9 / 11x11 on float32 frames: one kernel (fused_big.hip); big_fused = 0 is the tuning
  // knob that sends them through the two launches below instead
  const bool big_fused = ctx->tune.big_fused != 0;
  // 7x7 as well: with the sampling source's scalar state, 49 resident coefficients overflow
  // the SGPR file (331 spills); streamed, the 4K chain measured 489 -> 449 us (float32) and
  // 493 -> 460 us (uint16 frames); 5x5 measured slower streamed (0.427 vs 0.399 ms, 16 frames).
  // stream_k = 9 is the tuning knob that puts 7x7 back on the resident form.
  const int stream_k = ctx->tune.stream_k;
  // round 3: batches of uint16 frames whose frames share map rows through LDS
  // (wave_run_strip_shared: bilinear, n_frames a multiple of the workgroup's waves) keep the 7x7
  // coefficients resident as op_sel pairs on the hand-scheduled loop: C4 64 x 4K 1.475 -> 1.333 ms
  // (knob pipe7 = 0: streamed).  float32 frames measure the same either way (1.484 / 1.484: two
  // more tap registers per footprint, 141 VGPRs) and stay on the streamed kernel.
  const bool shared7 = kh == 7 && kw == 7 && ctx->tune.pipe7 != 0 && ctx->tune.frames_wg != 0 &&
                       ctx->tune.frames_inner != 0 && (interp & 0xff) == IPA_INTER_LINEAR &&
                       n_frames % 4 == 0 && dst_dtype == IPA_F32 &&
                       src_dtype == IPA_U16;
  const bool streamed = kh >= stream_k && kh >= 7 && kh <= 11 && !shared7;
  if (big_fused && kh == kw && streamed &&
      (src_dtype == IPA_F32 || (kh == 7 && src_dtype == IPA_U16)) && dst_dtype == IPA_F32 &&
      kernel) {
    FusedCall f;
    f.coord_kind = 0;
    f.map = MapCoord{d_mapx, d_mapy, map_pitch};
    int rc = fused_fill(ctx, f, d_src, src_dtype, sh, sw, src_pitch, d_dst, dst_dtype, dh, dw,
                        dst_pitch, n_frames, src_frame_stride, dst_frame_stride, interp,
                        border_mode, border_value, conv_border_x, conv_border_y);
    if (rc) return rc;
    f.kernel = kernel;
    IPA_HIP(ctx, hipSetDevice(ctx->device));
    rc = ipa_fused_big_launch(ctx, f, kh);
    if (rc < 0) return rc;
    if (rc == 0) {
      IPA_HIP(ctx, hipGetLastError());
      return IPA_OK;
    }
  }
  void* tmp = nullptr;
  int big = big_kernel_tmp(ctx, src_dtype, 0, interp, kh, kw, dst_dtype, dh, dw, n_frames, &tmp);
  if (big < 0) return big;
  if (big == 0) {
    int rc = ipa_remap_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, d_mapx, d_mapy, map_pitch, tmp,
                           IPA_F32, dh, dw, dw, n_frames, src_frame_stride, (long)dh * dw, interp,
                           border_mode, border_value);
    if (rc) return rc;
    return ipa_conv2d_dev(ctx, tmp, IPA_F32, dh, dw, dw, kernel, kh, kw, nullptr, 0, d_dst,
                          dst_pitch, n_frames, (long)dh * dw, dst_frame_stride, conv_border_x,
                          conv_border_y, 0.0);
  }
  FusedCall f;
  f.coord_kind = 0;
  f.map = MapCoord{d_mapx, d_mapy, map_pitch};
  return fused_common(ctx, f, d_src, src_dtype, sh, sw, src_pitch, kernel, kh, kw, d_dst, dst_dtype,
                      dh, dw, dst_pitch, n_frames, src_frame_stride, dst_frame_stride, interp,
                      border_mode, border_value, conv_border_x, conv_border_y);
}

int ipa_undistort_conv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                             long src_pitch, const double* K, const double* dist5,
                             const double* newK, const double* kernel, int kh, int kw, void* d_dst,
                             int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                             long src_frame_stride, long dst_frame_stride, int interp,
                             int border_mode, double border_value, int conv_border_x,
                             int conv_border_y) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, K && dist5 && newK, "K, dist5 and newK must be given");
  {
    double ky[9], kx[9];
    if (rank1_chain(ctx, kernel, kh, kw, src_dtype, dst_dtype, interp, ky, kx, ctx->tune.lens_cache != 0, ctx->tune.lens_cache != 0)) {
      ctx->rank1_routed++;
      return ipa_undistort_sepconv2d_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, K, dist5, newK, ky, kh, kx, kw,
                                         d_dst, dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                                         dst_frame_stride, interp, border_mode, border_value, conv_border_y,
                                         conv_border_x);
    }
  }
  if (ctx->tune.lens_cache) {
    // the model's float32 coordinates are the same for every frame and every call with these
    // parameters: evaluate them once (bit for bit what the per-pixel evaluation gives) and run
    // the map-based kernels, 9x9 / 11x11 in one kernel included
    float *mx = nullptr, *my = nullptr;
    int rc = ipa_lens_map_cached(ctx, K, dist5, newK, dh, dw, &mx, &my);
    if (rc) return rc;
    return ipa_remap_conv2d_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, mx, my, dw, kernel, kh, kw,
                                d_dst, dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                                dst_frame_stride, interp, border_mode, border_value, conv_border_x,
                                conv_border_y);
  }
  void* tmp = nullptr;
  int big = big_kernel_tmp(ctx, src_dtype, 1, interp, kh, kw, dst_dtype, dh, dw, n_frames, &tmp);
  if (big < 0) return big;
  if (big == 0) {
    int rc = ipa_undistort_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, K, dist5, newK, tmp,
                               IPA_F32, dh, dw, dw, n_frames, src_frame_stride, (long)dh * dw,
                               interp, border_mode, border_value);
    if (rc) return rc;
    return ipa_conv2d_dev(ctx, tmp, IPA_F32, dh, dw, dw, kernel, kh, kw, nullptr, 0, d_dst,
                          dst_pitch, n_frames, (long)dh * dw, dst_frame_stride, conv_border_x,
                          conv_border_y, 0.0);
  }
  FusedCall f;
  f.coord_kind = 1;
  UndistortCoord& c = f.und;
  IPA_REQUIRE(ctx, inv3f(newK, c.ir) == 0, "newK is singular");
  c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
  c.k1 = dist5[0]; c.k2 = dist5[1]; c.p1 = dist5[2]; c.p2 = dist5[3]; c.k3 = dist5[4];
  c.affine = (c.ir[6] == 0.0 && c.ir[7] == 0.0 && c.ir[8] == 1.0) ? 1 : 0;
  return fused_common(ctx, f, d_src, src_dtype, sh, sw, src_pitch, kernel, kh, kw, d_dst, dst_dtype,
                      dh, dw, dst_pitch, n_frames, src_frame_stride, dst_frame_stride, interp,
                      border_mode, border_value, conv_border_x, conv_border_y);
}

int ipa_warp_perspective_conv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                                    long src_pitch, const double* M, const double* kernel, int kh,
                                    int kw, void* d_dst, int dst_dtype, int dh, int dw,
                                    long dst_pitch, int n_frames, long src_frame_stride,
                                    long dst_frame_stride, int interp, int border_mode,
                                    double border_value, int conv_border_x, int conv_border_y) {
  if (!ctx) return IPA_ERR_BAD_ARG;
  IPA_REQUIRE(ctx, M, "null matrix");
  {
    double ky[9], kx[9];
    if (rank1_chain(ctx, kernel, kh, kw, src_dtype, dst_dtype, interp, ky, kx, true)) {
      ctx->rank1_routed++;
      return ipa_warp_perspective_sepconv2d_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, M, ky, kh, kx, kw, d_dst,
                                                dst_dtype, dh, dw, dst_pitch, n_frames, src_frame_stride,
                                                dst_frame_stride, interp, border_mode, border_value,
                                                conv_border_y, conv_border_x);
    }
  }
  void* tmp = nullptr;
  int big = big_kernel_tmp(ctx, src_dtype, 2, interp, kh, kw, dst_dtype, dh, dw, n_frames, &tmp);
  if (big < 0) return big;
  if (big != 0 && kernel && dh > 0 && dw > 0 &&
      rotated_warp_in_two_launches(ctx, M, src_dtype, dst_dtype, interp, dh, dw, n_frames)) {
    int rc = ipa_ws_reserve(ctx, (size_t)n_frames * dh * dw * 4);
    if (rc) return rc;
    tmp = ctx->ws;
    big = 0;
  }
  if (big == 0) {
    int rc = ipa_warp_perspective_dev(ctx, d_src, src_dtype, sh, sw, src_pitch, M, tmp, IPA_F32, dh,
                                      dw, dw, n_frames, src_frame_stride, (long)dh * dw, interp,
                                      border_mode, border_value);
    if (rc) return rc;
    return ipa_conv2d_dev(ctx, tmp, IPA_F32, dh, dw, dw, kernel, kh, kw, nullptr, 0, d_dst,
                          dst_pitch, n_frames, (long)dh * dw, dst_frame_stride, conv_border_x,
                          conv_border_y, 0.0);
  }
  FusedCall f;
  f.coord_kind = 2;
  for (int i = 0; i < 9; i++) f.hom.m[i] = M[i];
  return fused_common(ctx, f, d_src, src_dtype, sh, sw, src_pitch, kernel, kh, kw, d_dst, dst_dtype,
                      dh, dw, dst_pitch, n_frames, src_frame_stride, dst_frame_stride, interp,
                      border_mode, border_value, conv_border_x, conv_border_y);
}

}  // extern "C"
