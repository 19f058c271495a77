/*
 * imgproc_hip.h — C ABI of libimgproc_hip.so, the MI355X (gfx950) implementation
 * of imgProcessor's per-pixel hot path (undistort / perspective remap, K x K
 * filters, IDW stencils).
 *
 * The reference (radjkarl/imgProcessor, pure Python) has no FFI layer; its
 * boundary for this path is a set of Python call sites into cv2 / numba /
 * scipy.  Each entry point below names the reference call it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes
 * binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns an ipa_status (0 = ok).
 *   - images are row-major [y][x], single channel; pitches are in ELEMENTS.
 *   - `*_dev` functions take DEVICE pointers and enqueue on the context's
 *     stream without synchronising (use ipa_ctx_synchronize / events).
 *     The un-suffixed variants take HOST pointers, stage H2D/D2H through the
 *     context's workspace and return after the result is in host memory.
 *   - batches: n_frames images, consecutive frames `*_frame_stride` ELEMENTS
 *     apart; maps / kernels / matrices are shared by all frames.
 *   - there is NO CPU fallback in this library.  Without a usable gfx950
 *     device ipa_ctx_create fails with IPA_ERR_NO_DEVICE.
 */
#ifndef IMGPROC_HIP_H
#define IMGPROC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPA_VERSION 100 /* 0.1.0 */

typedef enum {
  IPA_OK = 0,
  IPA_ERR_BAD_ARG = -1,
  IPA_ERR_UNSUPPORTED = -2, /* dtype / mode combination not implemented */
  IPA_ERR_HIP = -3,         /* a HIP runtime call failed; see ipa_last_error */
  IPA_ERR_OOM = -4,
  IPA_ERR_NO_DEVICE = -5
} ipa_status;

/* pixel types (transformations.py:78-87 toFloatArray: u8,u16 -> f32) */
typedef enum { IPA_U8 = 0, IPA_U16 = 1, IPA_F32 = 2, IPA_F64 = 3 } ipa_dtype;

/* interpolation; numbering follows cv2.INTER_* where one exists */
typedef enum {
  IPA_INTER_NEAREST = 0,
  IPA_INTER_LINEAR = 1,     /* cv2.INTER_LINEAR   — LensDistortion.py:323 */
  IPA_INTER_CUBIC_CV = 2,   /* cv2.INTER_CUBIC (Keys a=-0.75) — PerspectiveCorrection.py:378 */
  IPA_INTER_LANCZOS4 = 4,   /* cv2.INTER_LANCZOS4 — PerspectiveCorrection.py:404 (always q5) */
  IPA_INTER_CUBIC_KEYS = 5, /* Keys a=-0.5 == skimage.transform.warp(order=3) */
  IPA_INTER_Q5 = 0x100      /* OR-able flag: round coordinates to 1/32 px like cv2 (INTER_BITS=5) */
} ipa_interp;

/* border modes; numbering follows cv2.BORDER_* */
typedef enum {
  IPA_BORDER_CONSTANT = 0,  /* cv2.BORDER_CONSTANT / scipy 'constant' (per-tap blend) */
  IPA_BORDER_REPLICATE = 1, /* scipy 'nearest' */
  IPA_BORDER_REFLECT = 2,   /* fedcba|abcdef: scipy 'reflect', numpy 'symmetric',
                               extendArrayForConvolution 'reflect' */
  IPA_BORDER_WRAP = 3,      /* scipy 'wrap' / 'grid-wrap', extendArray modex='wrap' */
  IPA_BORDER_REFLECT101 = 4 /* scipy 'mirror' */
} ipa_border;

typedef struct ipa_ctx ipa_ctx;     /* one per device; owns a stream + workspace */
typedef struct ipa_event ipa_event; /* HIP event on the context's stream */

/* ---------------------------------------------------------------- runtime */
int ipa_version(void);
const char* ipa_status_string(int status);
/* last error text of this context (or of the calling thread when ctx==NULL) */
const char* ipa_last_error(const ipa_ctx* ctx);
int ipa_device_count(int* count);
/* PCI address "dddd:bb:dd.f" of HIP device `device_id` as this process numbers it (honours
 * HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES) - for host-side placement of the feeding threads
 * (sharding.numa_cpus_of_device reads /sys/bus/pci/devices/<address>/numa_node).  len >= 16. */
int ipa_device_pci_bus_id(int device_id, char* buf, size_t len);
int ipa_ctx_create(int device_id, ipa_ctx** ctx);
int ipa_ctx_destroy(ipa_ctx* ctx);
int ipa_ctx_synchronize(ipa_ctx* ctx);
/* Launch-shape knobs of a context (csrc/runtime.hip::kTuneNames is the list):
 *   "strip_h" (0 = by launch size), "frames_inner", "frames_wg", "frame_major", "big_wave",
 *   "big_fused", "stream_k", "pipe7", "ring_remap", "ring_min", "lens_cache", "u8_lz_lds",
 *   "stored_coords" (smallest batch whose homography / lens coordinates are evaluated once
 *   for all frames, 0 = never), "pipe", "tile_warp" (perspective warps of float32 frames with an
 *   output tile's source box in LDS: 0 never, 1 where it pays, 2 whenever the homography fits).
 *   "rank1_sep" (dense K x K kernels that are an outer product ky (x) kx on the separable K + K loops: bit 0
 *   the remap -> filter chains (float32 frames, bilinear taps, 3 / 5 / 7 / 9 taps), bit 1 the plain 9 x 9
 *   filter; default 3; the reference obtains its Gaussians separably: scipy.ndimage.gaussian_filter,
 *   filters/fastFilter.py:42).
 *   "strip_remap" (standalone bilinear remaps of camera frames - uint16 into float32, uint16 into uint16 with cv2's 16U
 *   arithmetic (IPA_INTER_LINEAR | IPA_INTER_Q5), uint8 into uint8 with its 8U fixed point; ipa_remap_dev, ipa_undistort_dev,
 *   ipa_warp_perspective_dev - on the marching strips of the chains with no filter: default 1; read-only counter
 *   "strip_remaps").
 *   "sep_u16" (uint16 frames, bilinear remap by maps or a homography -> separable 3 / 5 / 7 / 9-tap filter in one kernel: default 1;
 *   0 = two launches through the workspace).
 *   "tail_rows" (chunked batches on the shared-record loop end every XCD's share of the launch on short strips -
 *   the workgroups that run while the launch drains: -1 = the measured rule by taps and launch size, 0 = uniform
 *   strips, n = short strips of n rows; same bits).
 *   ipa_ctx_get_tuning also answers the read-only names "tail_rows_used" (height of the short strips of the last such
 *   launch, 0 = uniform), "rank1_routed" (dense calls sent to the separable loops so far) and "group_chunk_used" (frame
 *   groups per chunk of the last launch on the shared-record loop; "group_chunk" values that do not divide
 *   the group count go to the nearest divisor).
 * Values are range-checked (IPA_ERR_BAD_ARG).  ipa_ctx_create reads the IPA_* environment
 * defaults once; no launch path consults the environment.  The reference has no counterpart
 * (its numba / cv2 calls take no launch parameters). */
int ipa_ctx_set_tuning(ipa_ctx* ctx, const char* name, int value);
int ipa_ctx_get_tuning(ipa_ctx* ctx, const char* name, int* value);
/* name (e.g. "gfx950...") and compute-unit count of the context's device */
int ipa_ctx_device_info(ipa_ctx* ctx, char* name, size_t name_len, int* cu_count,
                        size_t* total_mem_bytes);
/* free and total bytes of the context's device right now (hipMemGetInfo): what the Python layer's
 * block pool bounds its optional second candidate allocation by (imgprocessor_amd/device.py; the
 * reference has no device memory - this is library plumbing, not a reference call site) */
int ipa_mem_info(ipa_ctx* ctx, size_t* free_bytes, size_t* total_bytes);

/* device / pinned-host memory owned by the caller until freed */
int ipa_malloc(ipa_ctx* ctx, size_t bytes, void** dptr);
int ipa_free(ipa_ctx* ctx, void* dptr);
int ipa_host_alloc(ipa_ctx* ctx, size_t bytes, void** hptr);
int ipa_host_free(ipa_ctx* ctx, void* hptr);
int ipa_memcpy_h2d(ipa_ctx* ctx, void* dptr, const void* hptr, size_t bytes); /* synchronous */
int ipa_memcpy_d2h(ipa_ctx* ctx, void* hptr, const void* dptr, size_t bytes); /* synchronous */
int ipa_memcpy_d2d(ipa_ctx* ctx, void* dst, const void* src, size_t bytes);   /* stream-ordered */
int ipa_memset(ipa_ctx* ctx, void* dptr, int value, size_t bytes);            /* stream-ordered */

/* HIP events recorded on the stream the kernels run on (bench.py timing) */
int ipa_event_create(ipa_ctx* ctx, ipa_event** ev);
int ipa_event_destroy(ipa_ctx* ctx, ipa_event* ev);
int ipa_event_record(ipa_ctx* ctx, ipa_event* ev);
int ipa_event_elapsed_ms(ipa_ctx* ctx, ipa_event* start, ipa_event* stop, float* ms);

/* ------------------------------------------------------------ map builder */
/* replaces cv2.initUndistortRectifyMap(K, dist, None, newK, (w,h), CV_32FC1)
 * at camera/LensDistortion.py:355-357.  K,newK: 3x3 row-major double;
 * dist5 = [k1,k2,p1,p2,k3] (LensDistortion.py:370,380).  Maps are float32. */
int ipa_build_undistort_map_dev(ipa_ctx* ctx, const double* K, const double* dist5,
                                const double* newK, int h, int w, float* d_mapx, float* d_mapy,
                                long map_pitch);
int ipa_build_undistort_map(ipa_ctx* ctx, const double* K, const double* dist5,
                            const double* newK, int h, int w, float* mapx, float* mapy);

/* ------------------------------------------------------------------ remap */
/* replaces cv2.remap(image, mapx, mapy, interp, borderMode, borderValue) at
 * camera/LensDistortion.py:323-326,339-340 (and transform/polarTransform.py:66,105).
 * dst dtype may equal the src dtype or be IPA_F32 (fused toFloatArray ingest).
 * u8 -> u8 INTER_LINEAR uses cv2's exact fixed-point arithmetic. */
int ipa_remap_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw, long src_pitch,
                  const float* d_mapx, const float* d_mapy, long map_pitch, void* d_dst,
                  int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                  long src_frame_stride, long dst_frame_stride, int interp, int border_mode,
                  double border_value);
int ipa_remap(ipa_ctx* ctx, const void* src, int src_dtype, int sh, int sw, const float* mapx,
              const float* mapy, void* dst, int dst_dtype, int dh, int dw, int n_frames,
              int interp, int border_mode, double border_value);

/* LensDistortion.correct without materialised maps: the distortion model of
 * initUndistortRectifyMap is evaluated per pixel in double, rounded to the
 * float32 a CV_32FC1 map would hold, then sampled exactly like ipa_remap.
 * Bit-identical to ipa_build_undistort_map_dev + ipa_remap_dev. */
int ipa_undistort_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                      long src_pitch, const double* K, const double* dist5, const double* newK,
                      void* d_dst, int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                      long src_frame_stride, long dst_frame_stride, int interp, int border_mode,
                      double border_value);
int ipa_undistort(ipa_ctx* ctx, const void* src, int src_dtype, int sh, int sw, const double* K,
                  const double* dist5, const double* newK, void* dst, int dst_dtype, int dh,
                  int dw, int n_frames, int interp, int border_mode, double border_value);

/* replaces cv2.warpPerspective at camera/PerspectiveCorrection.py:241-242,
 * 377-378,401-405 (and transform/simplePerspectiveTransform.py:27-31).
 * M is the 3x3 row-major double matrix mapping DESTINATION (x,y,1) to source
 * coordinates: pass inv(H) for a plain call, H itself for WARP_INVERSE_MAP. */
int ipa_warp_perspective_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                             long src_pitch, const double* M, void* d_dst, int dst_dtype, int dh,
                             int dw, long dst_pitch, int n_frames, long src_frame_stride,
                             long dst_frame_stride, int interp, int border_mode,
                             double border_value);
int ipa_warp_perspective(ipa_ctx* ctx, const void* src, int src_dtype, int sh, int sw,
                         const double* M, void* dst, int dst_dtype, int dh, int dw, int n_frames,
                         int interp, int border_mode, double border_value);

/* ---------------------------------------------------------------- filters */
/* dense kh x kw centred correlation (== scipy.ndimage.correlate, origin 0):
 *   dst[y,x] = sum_{i,j} kernel[i,j] * src[y+i-kh/2, x+j-kw/2]
 * replaces filters/maskedConvolve.py:24-43 (_calc) + the padding of
 * filters/_extendArrayForConvolution.py:5-97 (border_x / border_y per axis),
 * and the cv2.blur / scipy.ndimage.convolve call sites
 * (camera/lens/estimateSystematicErrorLensCorrection.py:206-207, features/hog.py:62-63).
 * kernel: HOST pointer, kh*kw doubles.  d_mask: optional uint8 (H,W) device
 * array; where it is 0 the output is 0 (maskedConvolve), NULL = everywhere.
 * dtype: IPA_F32 or IPA_F64 (src and dst). */
int ipa_conv2d_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                   const double* kernel, int kh, int kw, const uint8_t* d_mask, long mask_pitch,
                   void* d_dst, long dst_pitch, int n_frames, long src_frame_stride,
                   long dst_frame_stride, int border_x, int border_y, double border_value);
int ipa_conv2d(ipa_ctx* ctx, const void* src, int dtype, int h, int w, const double* kernel,
               int kh, int kw, const uint8_t* mask, void* dst, int n_frames, int border_x,
               int border_y, double border_value);

/* separable correlation, scipy.ndimage.gaussian_filter order: axis 0 (y) with
 * ky[nky] first, intermediate rounded to the image dtype, then axis 1 (x) with
 * kx[nkx].  nky==0 / nkx==0 skips that axis.  Replaces the gaussian_filter call
 * sites filters/standardDeviation.py:23, filters/fastFilter.py:42,
 * camera/flatField/flatField.py:47.  Single pass over HBM. */
int ipa_sepconv2d_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                      const double* ky, int nky, const double* kx, int nkx, void* d_dst,
                      long dst_pitch, int n_frames, long src_frame_stride, long dst_frame_stride,
                      int border_y, int border_x, double border_value);
int ipa_sepconv2d(ipa_ctx* ctx, const void* src, int dtype, int h, int w, const double* ky,
                  int nky, const double* kx, int nkx, void* dst, int n_frames, int border_y,
                  int border_x, double border_value);

/* replaces filters/varYSizeGaussianFilter.py:53-68 (_2dConvolutionYdependentKernel):
 *   dst[r,c] = sum_{ii<k0, jj<k1} kernels[r][ii][jj] * src[r+ii-k0/2, c+jj-k1/2]
 * with NaN pixels skipped (no renormalisation) and borders resolved on the fly
 * (the reference pads first; its defaults are modex='wrap', modey='reflect').
 * d_kernels: DEVICE array of h*k0*k1 doubles (one k0 x k1 table per row). */
int ipa_conv_ydep_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                      const double* d_kernels, int k0, int k1, int border_x, int border_y,
                      void* d_dst, long dst_pitch);
/* filters/varYSizeGaussianFilter.py:9-50 in one call: the per-row Gaussian tables
 * (gaussian_filter(delta, (stdys[r], stdx)) with stdys = linspace(sig_min, sig_max, h), :22-46) are
 * built on the device as separable factors - `rowk` = the kx x-responses of the delta (host,
 * kx doubles) - and the NaN-skipping row-dependent correlation (:53-68) forms the coefficients
 * on the fly.  Returns after the launch has consumed `rowk`. */
int ipa_var_y_gauss_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                        double sig_min, double sig_max, int ky, const double* rowk, int kx,
                        int border_x, int border_y, void* d_dst, long dst_pitch);

/* replaces filters/standardDeviation.py:34-70 (_calc): local standard deviation of
 * img around blurred[i,j] over the window [i-kx/2, min(i+kx/2, h)) x [j-ky/2, min(j+ky/2, w)),
 * divided by (rows-1)*(cols-1) exactly as the reference does. */
int ipa_local_std_dev(ipa_ctx* ctx, const void* d_img, const void* d_blurred, int dtype, int h,
                      int w, long pitch, long blurred_pitch, int ksize_x, int ksize_y,
                      void* d_out, long out_pitch);

/* replaces filters/maskedFilter.py:43-72 (_calcMean, reached from maskedFilter(fn='mean'),
 * :12-37): mean of the pixels with mask == 0 inside the window
 * [i-ksize/2, min(i+ksize/2, h)) x [j-ksize/2, min(j+ksize/2, w)).
 *   fill_mask != 0: written for the pixels with mask != 0 that have at least one such
 *                   neighbour, everything else of d_out untouched; d_out may be d_arr
 *                   (the reference's in-place fill);
 *   fill_mask == 0: written for the pixels with mask == 0, NaN elsewhere; not in place.
 * d_mask: DEVICE uint8 h x w (non-zero = masked).  float32/float64, double accumulation. */
int ipa_masked_mean_dev(ipa_ctx* ctx, const void* d_arr, int dtype, const unsigned char* d_mask,
                        int h, int w, long pitch, long mask_pitch, int ksize, int fill_mask,
                        void* d_out, long out_pitch);

/* replaces filters/maskedFilter.py:76-102 (_calcMedian, maskedFilter(fn='median')): as
 * ipa_masked_mean_dev with np.median of the window's mask == 0 pixels (mean of the two middle
 * values for an even count; NaN if any of them is NaN).  The window (ksize/2*2)^2 must fit the
 * kernel's per-wave LDS buffer: ksize <= 126 for float32, 90 for float64. */
int ipa_masked_median_dev(ipa_ctx* ctx, const void* d_arr, int dtype, const unsigned char* d_mask,
                          int h, int w, long pitch, long mask_pitch, int ksize, int fill_mask,
                          void* d_out, long out_pitch);

/* replaces filters/nan_maximum_filter.py:17-37 (_calc): np.nanmax over the same clipped
 * window; NaN where the whole window is NaN. */
int ipa_nan_max_dev(ipa_ctx* ctx, const void* d_arr, int dtype, int h, int w, long pitch,
                    int ksize, void* d_out, long out_pitch);

/* replaces render/closestDirectDistance.py:17-41 (_calc): for every zero pixel of d_arr (uint8,
 * non-zero = set) the distance to the closest set pixel within the +-ksize window, 2*ksize when
 * there is none; 0 on set pixels.  out_dtype IPA_U16 (the reference's default dtype, value
 * truncated) or IPA_F64. */
int ipa_closest_distance_dev(ipa_ctx* ctx, const unsigned char* d_arr, int h, int w, long pitch,
                             int ksize, void* d_out, int out_dtype, long out_pitch);

/* replaces uncertainty/positionToIntensityUncertainty.py:7-49 (_calc_constPSF / _calc_variPSF):
 *   sint[i,j] = sqrt( sum psf[ii,jj] * (img[i-ii+c, j-jj+c] - img[i,j])^2 ),  psf a (2*ksize+1)^2
 * Gaussian normalised to 1 with the FIRST sigma on the row axis (numbaGaussian2d as called at
 * :14,:39), pixels within ksize of the frame and NaN centres left 0.  d_sx / d_sy: per-pixel
 * float64 sigma maps, or both NULL to use the scalars sx / sy.  img float32/float64, d_sint
 * float64. */
int ipa_pos_intensity_unc_dev(ipa_ctx* ctx, const void* d_img, int dtype, int h, int w, long pitch,
                              const double* d_sx, const double* d_sy, long sigma_pitch, double sx,
                              double sy, int ksize, double* d_sint, long out_pitch);

/* replaces filters/medianThreshold.py:7-30 with size=3:
 *   blur = scipy.ndimage.median_filter(img, size=3)   (mode 'reflect': edge pixel repeated)
 *   hit  = |(img - blur) / blur| > threshold           ('<' when cond_less != 0), float64, IEEE
 *   out  = hit ? blur : img;   d_indices (uint8 h x w, may be NULL) = hit
 * threshold must be > 0 (the reference returns its input untouched otherwise).  Not in place. */
int ipa_median_threshold_dev(ipa_ctx* ctx, const void* d_img, int dtype, int h, int w, long pitch,
                             double threshold, int cond_less, void* d_out, long out_pitch,
                             unsigned char* d_indices, long idx_pitch);
/* the same for any window size (filters/medianThreshold.py:7-30 passes `size` to
 * scipy.ndimage.median_filter): blur = the element of rank size*size/2 of the size x size window
 * at offsets -size/2 .. size-1-size/2, edge pixels repeated; size = 3 takes the kernel above. */
int ipa_median_threshold_size_dev(ipa_ctx* ctx, const void* d_img, int dtype, int h, int w,
                                  long pitch, int size, double threshold, int cond_less,
                                  void* d_out, long out_pitch, unsigned char* d_indices,
                                  long idx_pitch);

/* replaces stages 2-4 of CameraCalibration.correct (camera/CameraCalibration.py:416-437) in one
 * pass over the frame:
 *   v = img - bg                      (:505  _correctDarkCurrent;   d_bg NULL = stage skipped)
 *   v = ff != 0 ? v / ff : v          (:527-528 _correctVignetting; d_ff NULL = stage skipped)
 *   if threshold > 0: v = nan_to_num(v); medianThreshold(v, threshold, size 3, '>')  (:566-567)
 * d_bg / d_ff: DEVICE arrays of the image dtype (float32/float64).  Not in place. */
int ipa_calib_prefilter_dev(ipa_ctx* ctx, const void* d_img, int dtype, const void* d_bg,
                            const void* d_ff, int h, int w, long pitch, long bg_pitch,
                            long ff_pitch, double threshold, void* d_out, long out_pitch);

/* replaces filters/_extendArrayForConvolution.py:5-97 for callers that want the
 * padded array itself (the filters above resolve borders while staging and do
 * not need it): dst is (h + 2*(ky/2)) x (w + 2*(kx/2)), kx/ky = kernel size
 * along x/y.  Any dtype (pure copy). */
int ipa_extend_array_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, long src_pitch,
                         int kx, int ky, int modex, int modey, void* d_dst, long dst_pitch);

/* (H, W, C) images - what cv2.remap / cv2.warpPerspective take in LensDistortion.correct
 * (camera/LensDistortion.py:323-326) and PerspectiveCorrection.correct
 * (camera/PerspectiveCorrection.py:401-405; the reference's own demo warps a colour PNG,
 * :858-900) - against the C planes of (H, W) every entry point above works on (n_frames = C):
 * device-side layout copies, so that a colour frame goes host -> device -> host once, without a
 * transposed copy on the host.  Pitches in elements (src_pitch of the interleaved image >= w * C),
 * any dtype. */
int ipa_deinterleave_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, int channels,
                         long src_pitch, void* d_dst, long dst_pitch, long dst_plane_stride);
int ipa_interleave_dev(ipa_ctx* ctx, const void* d_src, int dtype, int h, int w, int channels,
                       long src_pitch, long src_plane_stride, void* d_dst, long dst_pitch);

/* --------------------------------------------- fused remap -> K x K filter */
/* the headline chain (LensDistortion.correct followed by a K x K filter, the
 * in-tree archetype being estimateSystematicErrorLensCorrection.py:199-207):
 * remapped pixels (incl. the filter halo, resolved with conv_border_*) are
 * produced into LDS and filtered there; the intermediate image never touches
 * HBM.  dst dtype is IPA_F32 (src may be u8/u16/f32) or IPA_F64 (src f64).
 * One kernel is built for float32 frames (bilinear, both bicubics), uint16 frames (bilinear;
 * maps or the lens model) and uint8 frames (bilinear, maps; 3 .. 7) with square 3 .. 11 kernels; every OTHER combination the standalone
 * entry points accept (Lanczos4 / nearest taps, uint8 frames, uint16 frames under a homography or
 * with bicubic taps, rectangular / even / larger kernels) runs as what it is - remap into the
 * context workspace, then ipa_conv2d_dev - with the bits of the two calls (round 6; these
 * returned IPA_ERR_UNSUPPORTED before).  Holds for the three *_conv2d_dev entry points. */
int ipa_remap_conv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                         long src_pitch, const float* d_mapx, const float* d_mapy, long map_pitch,
                         const double* kernel, int kh, int kw, void* d_dst, int dst_dtype, int dh,
                         int dw, long dst_pitch, int n_frames, long src_frame_stride,
                         long dst_frame_stride, int interp, int border_mode, double border_value,
                         int conv_border_x, int conv_border_y);
int ipa_undistort_conv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                             long src_pitch, const double* K, const double* dist5,
                             const double* newK, const double* kernel, int kh, int kw, void* d_dst,
                             int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                             long src_frame_stride, long dst_frame_stride, int interp,
                             int border_mode, double border_value, int conv_border_x,
                             int conv_border_y);
int ipa_warp_perspective_conv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                                    long src_pitch, const double* M, const double* kernel, int kh,
                                    int kw, void* d_dst, int dst_dtype, int dh, int dw,
                                    long dst_pitch, int n_frames, long src_frame_stride,
                                    long dst_frame_stride, int interp, int border_mode,
                                    double border_value, int conv_border_x, int conv_border_y);

/* The same chain with a SEPARABLE filter (BASELINE config C3: PerspectiveCorrection remap +
 * separable 9-tap Gaussian; in the reference cv2.remap / cv2.warpPerspective followed by
 * scipy.ndimage.gaussian_filter, e.g. PerspectiveCorrection.py:401-405 then
 * filters/fastFilter.py:42): axis 0 with ky[nky], the float32 intermediate, then axis 1 with
 * kx[nkx], exactly as ipa_sepconv2d_dev on the materialised remap result.  One kernel for
 * float32 sources, INTER_LINEAR and nky == nkx in {3,5,7,9}; any other combination runs
 * as remap + ipa_sepconv2d_dev through the context workspace (same results).  dst is
 * IPA_F32; a constant filter border uses 0. */
int ipa_remap_sepconv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                            long src_pitch, const float* d_mapx, const float* d_mapy,
                            long map_pitch, const double* ky, int nky, const double* kx, int nkx,
                            void* d_dst, int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                            long src_frame_stride, long dst_frame_stride, int interp,
                            int border_mode, double border_value, int conv_border_y,
                            int conv_border_x);
int ipa_undistort_sepconv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh, int sw,
                                long src_pitch, const double* K, const double* dist5,
                                const double* newK, const double* ky, int nky, const double* kx,
                                int nkx, void* d_dst, int dst_dtype, int dh, int dw, long dst_pitch,
                                int n_frames, long src_frame_stride, long dst_frame_stride,
                                int interp, int border_mode, double border_value, int conv_border_y,
                                int conv_border_x);
int ipa_warp_perspective_sepconv2d_dev(ipa_ctx* ctx, const void* d_src, int src_dtype, int sh,
                                       int sw, long src_pitch, const double* M, const double* ky,
                                       int nky, const double* kx, int nkx, void* d_dst,
                                       int dst_dtype, int dh, int dw, long dst_pitch, int n_frames,
                                       long src_frame_stride, long dst_frame_stride, int interp,
                                       int border_mode, double border_value, int conv_border_y,
                                       int conv_border_x);

/* ------------------------------------------------------------ interpolate */
/* replaces interpolate/interpolate2dStructuredIDW.py:26-65 (_calc): every
 * pixel with mask!=0 becomes the weighted mean of the unmasked pixels in the
 * (2*ksize+1)^2 window; weights: HOST (2*ksize+1)^2 doubles (table built by the
 * caller exactly as :16-21).  In place on d_grid (F32 or F64). */
int ipa_idw_fill_dev(ipa_ctx* ctx, void* d_grid, int dtype, const uint8_t* d_mask, int h, int w,
                     long pitch, int ksize, const double* weights);
int ipa_idw_fill(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                 int ksize, const double* weights);

/* replaces interpolate/interpolate2dStructuredFastIDW.py:29-63: neighbours are
 * visited in the order of offsets[n][2] (int32 dy,dx pairs, HOST), stopping
 * after `minnvals`+1 hits exactly as the reference loop does. */
int ipa_fast_idw_fill_dev(ipa_ctx* ctx, void* d_grid, int dtype, const uint8_t* d_mask, int h,
                          int w, long pitch, const int32_t* offsets, const double* weights, int n,
                          int minnvals);
int ipa_fast_idw_fill(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                      const int32_t* offsets, const double* weights, int n, int minnvals);

/* replaces interpolate/interpolate2dUnstructuredIDW.py:7-38: every pixel of the grid
 * becomes sum(w v) / sum(w) over the n scattered points (x = ROW coordinate, y = column - the
 * reference indexes grid[i, j] with i against x), w = 1 / ((x-i)^2 + (y-j)^2)^(power/2), summed
 * in point order in float64; a pixel that is a point takes the first such point's value.
 * x, y, v: HOST doubles.  Writes every pixel of d_grid (F32 or F64). */
int ipa_unstructured_idw_dev(ipa_ctx* ctx, void* d_grid, int dtype, int h, int w, long pitch,
                             const double* x, const double* y, const double* v, int n,
                             double power);
int ipa_unstructured_idw(ipa_ctx* ctx, void* grid, int dtype, int h, int w, const double* x,
                         const double* y, const double* v, int n, double power);

/* replaces interpolate/interpolateCircular2dStructuredIDW.py:7-69 as written: IDW over the
 * window [i-k, min(i+k, h)) x [j-k, min(j+k, h)) (upper ends exclusive) with the distance
 * measured in polar coordinates about (cx, cy): ((fr dr)^2 + (fphi dphi midR)^2)^2; rows AND
 * columns run to shape[0] (:16-17), so w >= h is required and columns >= h stay untouched.
 * In place on d_grid (F32 or F64). */
int ipa_circular_idw_fill_dev(ipa_ctx* ctx, void* d_grid, int dtype, const uint8_t* d_mask, int h,
                              int w, long pitch, int ksize, double power, double fr, double fphi,
                              double cx, double cy);
int ipa_circular_idw_fill(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                          int ksize, double power, double fr, double fphi, double cx, double cy);

/* replaces interpolate/interpolate2dStructuredCrossAvg.py:7-115 as written: every masked
 * pixel blends the local averages (unmasked pixels within +-ksize) at the nearest unmasked
 * pixel towards row 0, towards column 0 and towards the last column with weights
 * 1 / distance^(power/2) (float32, normalised); the source's slot quirks (the search towards
 * the last row only validates the column-0 slot, which then keeps the previous pixel's value)
 * are reproduced.  In place on d_grid (F32 or F64). */
int ipa_cross_avg_fill_dev(ipa_ctx* ctx, void* d_grid, int dtype, const uint8_t* d_mask, int h,
                           int w, long pitch, int ksize, double power);
int ipa_cross_avg_fill(ipa_ctx* ctx, void* grid, int dtype, const uint8_t* mask, int h, int w,
                       int ksize, double power);

/* replaces interpolate/interpolate2dStructuredPointSpreadIDW.py:7-141 as written: sweeps over the
 * border pixels of the masked areas (_createBorder :31-63: row and column scans that carry their
 * previous value across row / column ends - index -1 marks the last pixel of the row / column),
 * each filled in raster order from the unmasked pixels within [i-k, i+k) x [j-k, j+k) with weights
 * 1 / distance^power and unmasked at once, so later pixels of the sweep see it (:75-135); repeated
 * until a border pass finds no transition or max_iter sweeps have run.  In place on grid (F32 or
 * F64) AND mask (filled pixels become 0). */
int ipa_point_spread_idw_dev(ipa_ctx* ctx, void* d_grid, int dtype, uint8_t* d_mask, int h, int w,
                             long pitch, int ksize, double power, long max_iter);
int ipa_point_spread_idw(ipa_ctx* ctx, void* grid, int dtype, uint8_t* mask, int h, int w,
                         int ksize, double power, long max_iter);

/* ---------------------------------------------------------------- filters: fast* */
/* replaces cv2.resize(img, (dw, dh), interpolation=...) at filters/fastFilter.py:47-48
 * (INTER_LANCZOS4 on the float64 grid of statistics) and filters/fastMean.py:14-19 (INTER_AREA
 * down, INTER_LINEAR up), single-channel IPA_F32 / IPA_F64.  OpenCV's published algorithm
 * (coordinates (d + 0.5) scale - 0.5 in float32, float32 coefficients, horizontal pass first in
 * the image's type; INTER_AREA by block sums / decimation tables, downscaling only). */
typedef enum {
  IPA_RESIZE_LINEAR = 1, IPA_RESIZE_CUBIC = 2, IPA_RESIZE_AREA = 3, IPA_RESIZE_LANCZOS4 = 4
} ipa_resize_interp;   /* cv2's INTER_* numbers */
int ipa_resize_dev(ipa_ctx* ctx, const void* d_src, int dtype, int sh, int sw, long src_pitch,
                   void* d_dst, int dh, int dw, long dst_pitch, int interp);
int ipa_resize(ipa_ctx* ctx, const void* src, int dtype, int sh, int sw, void* dst, int dh, int dw,
               int interp);

/* replaces filters/fastFilter.py:52-122 (_iter + _calcMedian / _calcNanMedian / _calcMean /
 * _calcNanMean): out[ii, jj] = statistic of arr[max(i-k,0) : min(i+k,h) : every,
 * max(j-k,0) : min(j+k,w) : every] at i = ii every, j = jj every; out is ceil(h / every) x
 * ceil(w / every) doubles (the caller applies the reference's crop of the last row and column).
 * fn: 0 median (NaN when the window holds one), 1 nanmedian, 2 mean, 3 nanmean. */
int ipa_fast_filter_stat_dev(ipa_ctx* ctx, const void* d_arr, int dtype, int h, int w, long pitch,
                             int ksize, int every, int fn, double* d_out);
int ipa_fast_filter_stat(ipa_ctx* ctx, const void* arr, int dtype, int h, int w, int ksize,
                         int every, int fn, double* out);

#ifdef __cplusplus
}
#endif
#endif /* IMGPROC_HIP_H */
